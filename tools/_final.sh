cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r04zz_bench.json 2> gpurun_out/r04zz_bench.err; tail -c 600 gpurun_out/r04zz_bench.json
for WL in cfg2_dnerf_100k_800x800 cfg3_hypernerf_300k_536x960 cfg5_stress_2M_2048x2048; do
  timeout 600 python bench.py --workload $WL --steps 10 --warmup 3 --repeats 5 --cpu-frames 1 > gpurun_out/r04zz_bench_$WL.json 2> gpurun_out/r04zz_bench_$WL.err
done
timeout 300 python bench.py --scene shell --steps 10 --warmup 3 --repeats 5 --cpu-frames 1 --no-extras > gpurun_out/r04zz_bench_shell.json 2> gpurun_out/r04zz_bench_shell.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04zz_prof_shell -o trace --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --scene shell --steps 20 --warmup 4 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/r04zz_prof_shell/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    out = ["%-60s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct")]
    for r in rows[:28]:
        out.append("%-60s %8s %12.3f %10.2f %7s" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    open("gpurun_out/r04zz_kernel_stats_shell.txt", "w").write("\n".join(out) + "\n")
PY
rm -rf gpurun_out/r04zz_prof_shell
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -rA 2>&1 | grep -E "kink rows|passed|failed" | cut -c1-400
