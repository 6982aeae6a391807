#!/bin/bash
# End-to-end timing of the command-line front end on MANY short records (simulated long reads; BASELINE.json
# configs[4] shape) on the GPU box.   Usage: bash tools/cli_reads_timing.sh <records> <bases per record> [M]
set -e
N=${1:-400}; LEN=${2:-50000}; M=${3:-100}
python3 - <<PY
import sys; sys.path.insert(0, '.')
from ribbit_amd.simulate import simulate_sequence, write_fasta
seq, _ = simulate_sequence($N * $LEN, 5, 2, min($M, 100))
write_fasta('/tmp/reads_in.fa', [(f'read{i}', seq[i * $LEN:(i + 1) * $LEN]) for i in range($N)])
PY
START=$(date +%s.%N)
RIBBIT_PROFILE=1 ./ribbit_amd/ribbit-hip -i /tmp/reads_in.fa -o /tmp/reads_out.bed -m 2 -M $M 2> /tmp/reads_err.log || true
END=$(date +%s.%N)
grep -c "Processing sequence" /tmp/reads_err.log || true
tail -3 /tmp/reads_err.log
python3 -c "
t = $END - $START
print(f'wall {t:.2f} s  {$N * $LEN / t / 1e6:.1f} Mbases/s  {$N / t:.0f} records/s')"
wc -l /tmp/reads_out.bed
