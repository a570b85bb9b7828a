#!/bin/bash
# tools/gpu_round.sh TAG -- one GPU-box pass: parity tests, bench line (with CPU baseline), rocprofv3 kernel stats.
# Outputs land in gpurun_out/TAG_*; copy what should be judged into profiles/.
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
(timeout 600 python -m pytest tests -m gpu -x -q -rA 2>&1 | tail -80) > gpurun_out/${TAG}_pytest.log 2>&1
tail -3 gpurun_out/${TAG}_pytest.log
timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 3000 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o trace --output-format csv -- python $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline > $R/gpurun_out/${TAG}_prof_bench.json 2> $R/gpurun_out/${TAG}_prof.err
cd $R
ls gpurun_out/${TAG}_prof | head
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/${TAG}_prof/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    out = ["%-60s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct")]
    for r in rows[:28]:
        out.append("%-60s %8s %12.3f %10.2f %7s" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    open("gpurun_out/${TAG}_kernel_stats.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out[:14]))
PY
