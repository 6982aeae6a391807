#!/bin/bash
# Refinement wall time against the number of slices of the small-job pipeline (RIBBIT_SSW_SLICES), one record.
# Usage (GPU box, repo root): bash tools/refine_slices_sweep.sh [bases]
B=${1:-64000000}
for S in 2 4 6 8 12 17 24 32 48; do
  echo "slices $S: $(RIBBIT_SSW_SLICES=$S python tools/refine_timing.py $B 2>/dev/null | tail -1)"
done
