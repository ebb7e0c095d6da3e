#!/bin/bash
# Round-3 evidence run (GPU box): SQ counters of the streaming kernel in three builds (default, fewer VALU instructions +
# mid-level fence, no global memory at all) and the reference's as-shipped input.txt on its own image, uncapped.
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03d
mkdir -p "$out"
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
B="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
for v in ${R03_VARIANTS-default bufsplitmid fake}; do
  if [ $v = default ]; then unset DEFF_AMD_LIB; else export DEFF_AMD_LIB=$root/tools/ab/$v.so; fi
  "$root/tools/pmc_kernel.sh" ${v}_a "$A" > "$out/pmc_${v}_a.log" 2>&1 || { echo "pmc $v a failed"; tail -5 "$out/pmc_${v}_a.log"; exit 1; }
  "$root/tools/pmc_kernel.sh" ${v}_b "$B" > "$out/pmc_${v}_b.log" 2>&1 || { echo "pmc $v b failed"; tail -5 "$out/pmc_${v}_b.log"; exit 1; }
  echo "counters $v done"
done
unset DEFF_AMD_LIB
# the as-shipped configuration, uncapped (input.txt:2-18 on 00042.jpg)
w=$(mktemp -d)
cp "$root/tests/golden/00042.jpg" "$w/"
printf 'Input File:\nPhases: 3\nDs: 0\nDf: 1\nDg: 1237500\nMeshAmpX: 1\nMeshAmpY: 1\nInputName: 00042.jpg\nCR: 1\nCL: 0\nOutputName: singleTest.csv\nprintCMap: 1\nCMapName: CMAP_00042.csv\nConvergence: 1e-5\nMaxIter: 5e5\nVerbose: 1\nRunBatch: 0\nNumImages: 500\n' > "$w/input.txt"
cd "$w"
t0=$(date +%s.%N)
"$root/effectivediffusivityfvm_amd/deff2d" --json res.json --field-bin field > "$out/as_shipped_00042.stdout" 2> "$out/as_shipped_00042.stderr"
echo "deff2d rc $? wall $(echo "$(date +%s.%N) - $t0" | bc) s" | tee "$out/as_shipped_00042.wall"
cp res.json "$out/as_shipped_00042.json"; cp singleTest.csv "$out/as_shipped_00042.csv"
sha256sum field_00000_1002x2007.f64 > "$out/as_shipped_00042_field.sha256"
head -3 CMAP_00042.csv > "$out/as_shipped_00042_cmap_head.csv"; wc -l CMAP_00042.csv >> "$out/as_shipped_00042_cmap_head.csv"
cat res.json
