#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate
# PMC passes (gfx950: they do not fit in one pass).  Output under gpurun_out/<tag>/; copy the
# summaries you want judged into profiles/.
# Usage (on the GPU box, from the repo root):  bash tools/profile_bench.sh <tag> ["extra bench.py arguments"]
#   e.g. bash tools/profile_bench.sh r04_m500 "--max-motif 500"   (the three scan kernels at BASELINE.json configs[4]'s 499 motif sizes)
set -e
TAG=${1:-prof}
EXTRA=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --depth 1: one batch at a time, so that a kernel's duration in the trace is its own (with several batches in flight the
# result copies -- blit kernels on this stack -- share the CUs with the next batch's scan under the profiler)
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --calibrate --depth 1 --stage-kernels $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err
# instruction mix / VALU utilisation of the scan kernel (own passes; a counter this build of rocprofv3 does not
# know just leaves its pass empty)
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "VALUBusy" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  D=$OUT/pmc_$(echo $SET | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --depth 1 --stage-kernels $EXTRA > $D.json 2> $D.err || echo "pass [$SET] failed"
done
find $OUT -name "*.csv" | head -40
