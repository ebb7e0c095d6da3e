#!/bin/bash
# Round-4 bundle (run on the GPU box through gpurun): the default bench line with its live rocprofv3 children, the residual
# kernels' per-kernel times and HBM counters (separate --pmc passes, the program itself after `--`).  Everything lands in gpurun_out/r04_evidence/.
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r04_evidence
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
python3 "$root/bench.py" > "$out/bench_default.log" 2> "$out/bench_default.err" || { echo "bench failed"; tail -5 "$out/bench_default.err"; exit 1; }
cp "$root/gpurun_out/bench_live_kernel_stats.csv" "$out/tb_kernel_stats.csv" 2>/dev/null
cp "$root/gpurun_out/bench_live_explicit_kernel_stats.csv" "$out/explicit_kernel_stats.csv" 2>/dev/null
for n in 4096 16384; do
  rm -rf /tmp/rp_res$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_res$n -- python3 "$root/tools/residual_bench.py" $n > "$out/residual_${n}.log" 2>&1 || { echo "residual stats run failed"; exit 1; }
  f=$(ls /tmp/rp_res$n/*/*kernel_stats.csv | head -1); grep -i "Name\|residual" "$f" > "$out/residual_${n}_kernel_stats.csv"
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_pmc_$ctr
  rocprofv3 --pmc $ctr --output-format csv -d /tmp/rp_pmc_$ctr -- python3 "$root/tools/residual_bench.py" 4096 > /dev/null 2>&1 || { echo "pmc run failed"; exit 1; }
  f=$(ls /tmp/rp_pmc_$ctr/*/*counter_collection.csv | head -1)
  python3 - "$f" $ctr >> "$out/residual_4096_traffic.txt" <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "k_residual_classes" in r["Kernel_Name"]]
print(f"{sys.argv[2]} k_residual_classes mean {sum(v)/len(v):.1f} KiB over {len(v)} launches")
PY
done
cat "$out/residual_4096_traffic.txt"; grep "^{" "$out/residual_4096.log" | cut -c1-200; cat "$out/residual_4096_kernel_stats.csv" | cut -c1-160
