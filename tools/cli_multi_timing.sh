#!/bin/bash
# End-to-end run of the command-line front end on SEVERAL large records in flight at once (each large enough for the batched
# GPU alignment path), with and without that path: wall time and BED identity.  Usage: bash tools/cli_multi_timing.sh <records> <bases per record>
set -e
N=${1:-4}; LEN=${2:-30000000}
python3 - <<PY
import sys; sys.path.insert(0, '.')
from ribbit_amd.simulate import simulate_sequence, write_fasta
write_fasta('/tmp/multi_in.fa', [(f'chr{i}', simulate_sequence($LEN, 60 + i, 2, 100, n_block_rate=0.1)[0]) for i in range($N)])
PY
for MODE in 0 auto; do
  if [ $MODE = auto ]; then unset RIBBIT_GPU_SSW; else export RIBBIT_GPU_SSW=$MODE; fi
  START=$(date +%s.%N)
  ./ribbit_amd/ribbit-hip -i /tmp/multi_in.fa -o /tmp/multi_$MODE.bed -m 2 -M 100 2> /tmp/multi_$MODE.err || { tail -5 /tmp/multi_$MODE.err; exit 1; }
  END=$(date +%s.%N)
  python3 -c "print(f'RIBBIT_GPU_SSW=$MODE: wall {$END - $START:.2f} s, {$N * $LEN / ($END - $START) / 1e6:.1f} Mbases/s')"
  grep -c "GPU alignment batches skipped" /tmp/multi_$MODE.err || true
done
cmp /tmp/multi_0.bed /tmp/multi_auto.bed && echo "BED identical with and without the GPU alignment path ($(wc -l < /tmp/multi_auto.bed) rows)"
