#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> "<counters>" [bench args]   (runs on the GPU box)
set -o pipefail
tag=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc $ctrs --output-format csv -d "$out" -- python3 "$root/bench.py" --steps 1 --warmup 0 --sweeps-per-step 24 --no-cpu-baseline --no-small-image --no-live-traffic --no-live-stats --no-iters-to-tol --explicit-sweeps 0 "$@" > "$out/bench.log" 2>&1 || { echo "pmc run failed"; tail -5 "$out/bench.log"; exit 1; }
python3 - "$out" <<'PY'
import csv,glob,sys,collections
f=max(glob.glob(sys.argv[1]+"/*/*_counter_collection.csv"))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "sweep" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    print(k)
    for c,v in d.items(): print("   %-28s %.4g (n=%d)"%(c,sum(v)/len(v),len(v)))
PY
