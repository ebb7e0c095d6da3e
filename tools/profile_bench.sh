#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + HBM PMC passes of bench.py.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-small-image --no-live-traffic --no-live-stats --no-iters-to-tol --explicit-sweeps 0 "$@" > "$out/bench_kt.log" 2>&1 || { echo "kernel-trace run failed"; tail -5 "$out/bench_kt.log"; exit 1; }
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d "$out/pmc_$ctr" -- python3 "$root/bench.py" --steps 1 --warmup 0 --sweeps-per-step 24 --no-cpu-baseline --no-small-image --no-live-traffic --no-live-stats --no-iters-to-tol --explicit-sweeps 0 "$@" > "$out/bench_pmc_$ctr.log" 2>&1 || { echo "pmc $ctr run failed"; tail -5 "$out/bench_pmc_$ctr.log"; exit 1; }
done
find "$out" -name '*.csv' | head -20
