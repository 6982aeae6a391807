#!/bin/bash
# Where a slice's fixed cost goes: the [refine_bed] profile line of the second pass for a few slicings.
# Usage (GPU box, repo root): bash tools/refine_slices_probe.sh [bases]
B=${1:-64000000}
for S in 2 8 16; do
  echo "== slices $S"
  RIBBIT_PROFILE=1 RIBBIT_SSW_SLICES=$S python tools/refine_timing.py $B 2>&1 | grep -E "^\[refine_bed\] [0-9]|^pass" | tail -3
done
