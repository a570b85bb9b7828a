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
# optional: SQ counters (MFMA busy / stalls) and HBM-side traffic counters, each in its own --pmc pass
if [ -n "$PMC" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_pmc_sq -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/${TAG}_pmc_$C -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  done
  cd $R
  python tools/pmc_summary.py gpurun_out/${TAG}_pmc_sq gpurun_out/${TAG}_pmc_sq.txt > /dev/null
  python tools/pmc_summary.py gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_fetch.txt > /dev/null
  python tools/pmc_summary.py gpurun_out/${TAG}_pmc_WRITE_SIZE gpurun_out/${TAG}_pmc_write.txt > /dev/null
  head -8 gpurun_out/${TAG}_pmc_sq.txt | cut -c1-230
fi
if [ -n "$EXTRA_WORKLOADS" ]; then
  for WL in $EXTRA_WORKLOADS; do
    timeout 300 python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_$WL.json 2> gpurun_out/${TAG}_bench_$WL.err
    python -c "
import json,sys
d=json.load(open('gpurun_out/${TAG}_bench_$WL.json')); print('$WL', round(d['value'],1), 'frames/s', d['config']['num_rendered'], d['kernels_ms_per_step'])" 2>&1 | cut -c1-400
  done
fi
