#!/bin/bash
# Builds a variant of libdeff_amd.so with extra compiler flags into tools/ab/<name>.so (git-ignored, travels with gpurun):
#   tools/build_variant.sh fence2 -DTB_FENCE_EVERY=2
# run it against the in-tree build with DEFF_AMD_LIB=tools/ab/<name>.so python tools/kbench.py ...
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/effectivediffusivityfvm_amd/csrc
out=$root/tools/ab
mkdir -p "$out/obj_$name"
for tu in api_core api_solve api_slab; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden \
      -Wno-unused-function "$@" -c -o "$out/obj_$name/$tu.o" "$src/$tu.hip" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$out/$name.so" "$out/obj_$name"/*.o -L/opt/rocm/lib -lrccl
rm -rf "$out/obj_$name"
echo "$out/$name.so"
