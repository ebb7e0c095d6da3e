#!/bin/bash
# Builds a variant of libdeff_amd.so with extra compiler flags into tools/ab/<name>.so (git-ignored, travels with gpurun):
#   tools/build_variant.sh fence2 -DTB_FENCE_EVERY=2
# run it against the in-tree build with DEFF_AMD_LIB=tools/ab/<name>.so python tools/kbench.py ...
# Only api_solve.hip holds the sweep kernels: the other objects are taken from the in-tree build (csrc/build/).
# Prints the VGPR / scratch budget of the streaming kernel's instantiations as a by-product.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/effectivediffusivityfvm_amd/csrc
out=$root/tools/ab
mkdir -p "$out/obj_$name"
make -s -C "$src" build/api_core.o build/api_slab.o build/api_residual.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden \
    -Wno-unused-function -Rpass-analysis=kernel-resource-usage "$@" -c -o "$out/obj_$name/api_solve.o" "$src/api_solve.hip" 2> "$out/$name.usage.txt"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$out/$name.so" "$out/obj_$name/api_solve.o" "$src/build/api_core.o" "$src/build/api_slab.o" "$src/build/api_residual.o" -L/opt/rocm/lib -lrccl
rm -rf "$out/obj_$name"
python3 - "$out/$name.usage.txt" <<'PY'
import re, sys
cur = None
for line in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        continue
    if cur and "k_sweep_matfree_tbILi8ELb0ELb0E" in cur:
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m:
            print(" ", m.group(1), m.group(2))
PY
echo "$out/$name.so"
