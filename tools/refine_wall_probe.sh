#!/bin/bash
# Wall time of every refine_to_bed call (the workers' phase of a slice) against the thread count, without per-seed timers.
# Usage (GPU box): bash tools/refine_wall_probe.sh [bases]
B=${1:-64000000}
for T in 4 8 16; do
  echo "== RIBBIT_THREADS=$T"
  RIBBIT_REFINE_WALL=1 RIBBIT_THREADS=$T python tools/refine_timing.py $B 2>&1 | grep -E "^\[refine\] call|^pass" | tail -12 | awk '/call/ {s+=$(NF-2); n++; print} /pass/ {print; print "   sum of calls so far", s, "ms in", n, "calls"}'
done
