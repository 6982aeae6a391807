#!/bin/bash
# End-to-end timing on several long records (chromosome shape) on the GPU box.
# Usage: bash tools/cli_chroms_timing.sh <records> <bases per record> [M]
set -e
N=${1:-4}; LEN=${2:-20000000}; M=${3:-100}
python3 - <<PY
import sys; sys.path.insert(0, '.')
from ribbit_amd.simulate import simulate_sequence, write_fasta
recs = []
for i in range($N):
    seq, _ = simulate_sequence($LEN, 20 + i, 2, min($M, 100))
    recs.append((f'chr{i}', seq))
write_fasta('/tmp/chroms_in.fa', recs)
PY
START=$(date +%s.%N)
RIBBIT_PROFILE=1 ./ribbit_amd/ribbit-hip -i /tmp/chroms_in.fa -o /tmp/chroms_out.bed -m 2 -M $M 2> /tmp/chroms_err.log || true
END=$(date +%s.%N)
grep stages /tmp/chroms_err.log | tail -1
python3 -c "
t = $END - $START
print(f'wall {t:.2f} s  {$N * $LEN / t / 1e6:.1f} Mbases/s')"
wc -l /tmp/chroms_out.bed
