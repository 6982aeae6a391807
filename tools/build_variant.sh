#!/bin/bash
# Builds an experimental variant of the library: kernels.hip recompiled with extra flags, the other objects taken from
# the regular build.  Usage: tools/build_variant.sh NAME "-DRB_SOMETHING=1 ..."  ->  variants/libribbit_NAME.so
# (select it with RIBBIT_HIP_LIBRARY; variants/ is not tracked).  Used for the ablation timings in DESIGN.md §4.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; flags=$2
mkdir -p "$root/variants"
src=$root/ribbit_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread $flags -I"$root/include" -I"$src" -c -o "$root/variants/kernels_$name.o" "$src/kernels.hip"
objs=$(ls "$src"/build/*.o | grep -v '/kernels.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -pthread -shared -o "$root/variants/libribbit_$name.so" "$root/variants/kernels_$name.o" $objs
echo "$root/variants/libribbit_$name.so"
