#!/bin/bash
# Refinement against the number of host threads (no profile: the timers perturb it).  Usage (GPU box): bash tools/refine_threads_probe.sh [bases]
B=${1:-64000000}
for T in 8 12 14 16 20 24; do
  echo "RIBBIT_THREADS=$T: $(RIBBIT_THREADS=$T python tools/refine_timing.py $B 2>/dev/null | tail -1)"
done
