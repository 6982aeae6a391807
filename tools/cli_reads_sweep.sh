#!/bin/bash
# 400 reads of 50 kb through ribbit-hip: records in flight x shared alignment batches.  Usage (GPU box): bash tools/cli_reads_sweep.sh
for J in 8 16; do
  echo "== --jobs $J, no shared batches: $(RIBBIT_JOBS=$J RIBBIT_SHARED_SSW=0 bash tools/cli_reads_timing.sh 400 50000 2>&1 | grep wall)"
  for P in 1 0; do
    echo "== --jobs $J, shared batches, paths on the GPU $P: $(RIBBIT_JOBS=$J RIBBIT_BATCH_PATHS=$P bash tools/cli_reads_timing.sh 400 50000 2>&1 | grep -E 'wall|shared' | tr '\n' ' ')"
  done
done
