#!/bin/bash
# End-to-end timing of the command-line front end on a synthetic FASTA (GPU box).
# Usage: bash tools/cli_timing.sh <bases> [M]
set -e
BASES=${1:-5000000}; M=${2:-100}
python3 - <<PY
import sys; sys.path.insert(0, '.')
from ribbit_amd.simulate import simulate_sequence, write_fasta
seq, _ = simulate_sequence($BASES, 2, 2, $M)
write_fasta('/tmp/cli_in.fa', [('sim', seq)])
PY
START=$(date +%s.%N)
./ribbit_amd/ribbit-hip -i /tmp/cli_in.fa -o /tmp/cli_out.bed -m 2 -M $M 2> /tmp/cli_err.log || true
END=$(date +%s.%N)
grep -E "Time elapsed|failed" /tmp/cli_err.log || true
echo "wall seconds: $(echo "$END - $START" | bc -l 2>/dev/null || python3 -c "print($END - $START)")"
wc -l /tmp/cli_out.bed
