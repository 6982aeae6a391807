#!/bin/bash
# SQ_INSTS_VALU of scan_perfect_kernel for the product build and the two ablated variants (tools/build_variant.sh), on the
# same 100-Mbp record: the measured weights behind profiles/isa_mix.json.  Usage (GPU box, repo root): bash tools/variant_pmc.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/variant_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in product nostage nochain; do
  if [ $V = product ]; then unset RIBBIT_HIP_LIBRARY; else export RIBBIT_HIP_LIBRARY=$R/variants/libribbit_$V.so; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/$V -- python3 $R/tools/perfect_probe.py 100000000 3 > $OUT/$V.log 2> $OUT/$V.err
  echo "$V done: $(tail -1 $OUT/$V.log)"
done
