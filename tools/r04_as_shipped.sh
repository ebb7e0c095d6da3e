#!/bin/bash
# The configuration the reference ships (input.txt:2-18 on 00042.jpg), uncapped, through deff2d on the GPU box; the field's
# SHA-256 must be round 3's (profiles/r03_as_shipped_00042.json): the kernels changed, the bits must not.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r04_as_shipped
mkdir -p "$out"
w=$(mktemp -d)
cp "$root/tests/golden/00042.jpg" "$w/"
printf 'Input File:\nPhases: 3\nDs: 0\nDf: 1\nDg: 1237500\nMeshAmpX: 1\nMeshAmpY: 1\nInputName: 00042.jpg\nCR: 1\nCL: 0\nOutputName: singleTest.csv\nprintCMap: 1\nCMapName: CMAP_00042.csv\nConvergence: 1e-5\nMaxIter: 5e5\nVerbose: 1\nRunBatch: 0\nNumImages: 500\n' > "$w/input.txt"
cd "$w"
SECONDS=0
"$root/effectivediffusivityfvm_amd/deff2d" --json res.json --field-bin field > "$out/stdout.txt" 2> "$out/stderr.txt"
echo "deff2d rc $? wall ${SECONDS} s" | tee "$out/wall.txt"
cp res.json "$out/as_shipped_00042.json"; cp singleTest.csv "$out/as_shipped_00042.csv"
sha256sum field_00000_1002x2007.f64 | tee "$out/field.sha256"
