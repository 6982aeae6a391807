#!/bin/bash
# Refinement wall time against the slicing of the GPU alignment batches (RIBBIT_SSW_SLICES x RIBBIT_SSW_SLICE_GROWTH), one record.
# Usage (GPU box, repo root): bash tools/refine_slices_sweep.sh [bases] > gpurun_out/slices.log
B=${1:-64000000}
for S in 2 3 4 6 8 12; do for G in 1 1.5 2; do
  echo "slices $S growth $G: $(RIBBIT_SSW_SLICES=$S RIBBIT_SSW_SLICE_GROWTH=$G python tools/refine_timing.py $B 2>/dev/null | tail -1)"
done; done
