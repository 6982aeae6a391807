#!/bin/bash
# Refinement of one record on the host threads against the record's own GPU alignment pipeline, by record size: where the
# switch (GPU_SSW_MIN_SEEDS in api_refine_bed.cpp) belongs.  Usage (GPU box): bash tools/refine_threshold_probe.sh
for B in 1000000 2000000 5000000 10000000 20000000 40000000; do
  for F in 0 1; do
    echo "bases $B RIBBIT_GPU_SSW=$F: $(RIBBIT_GPU_SSW=$F python tools/refine_timing.py $B 2>/dev/null | tail -1)"
  done
done
