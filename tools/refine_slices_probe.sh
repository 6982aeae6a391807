#!/bin/bash
# Where a slice's fixed cost goes: the [refine_bed] profile line of the second pass for a few slicings, with and without
# the long alignment classes on the GPU.  Usage (GPU box, repo root): bash tools/refine_slices_probe.sh [bases]
B=${1:-64000000}
for CFG in "2 1" "8 1" "2 0" "8 0" "16 0"; do
  set -- $CFG
  echo "== slices $1 large classes on the GPU: $2"
  RIBBIT_PROFILE=1 RIBBIT_SSW_SLICES=$1 RIBBIT_SSW_LARGE=$2 python tools/refine_timing.py $B 2>&1 | grep -E "^\[refine_bed\] [0-9]|^pass" | tail -3
done
