#!/bin/bash
# tools/gpu_round.sh TAG -- one GPU-box pass: parity tests, bench line (with CPU baseline + oracle parity), rocprofv3 kernel
# stats; optional (PMC=1) SQ counters and per-workload HBM traffic passes; optional EXTRA_WORKLOADS bench lines; optional (SHELL_SCENE=1)
# bench line + kernel stats of the shell scene.
# Outputs land in gpurun_out/TAG_*; copy what should be judged into profiles/.
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
if [ -z "$SKIP_TESTS" ]; then
  timeout ${TEST_TIMEOUT:-900} python -m pytest ${PYTEST_TARGET:-tests} -m gpu -q -rA ${PYTEST_ARGS:-} > gpurun_out/${TAG}_pytest.log 2>&1
  tail -3 gpurun_out/${TAG}_pytest.log
fi
if [ -z "$SKIP_BENCH" ]; then
  timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
  tail -c 4000 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
fi
prof_one() {  # $1 = workload, $2 = suffix
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof$2 -o trace --output-format csv -- python $R/bench.py --workload $1 --steps 20 --warmup 4 --repeats 1 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_prof_bench$2.json 2> $R/gpurun_out/${TAG}_prof$2.err
  cd $R
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/${TAG}_prof$2/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    out = ["%-60s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct")]
    for r in rows[:32]:
        out.append("%-60s %8s %12.3f %10.2f %7s" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    open("gpurun_out/${TAG}_kernel_stats$2.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out[:16]))
PY
}
pmc_one() {  # $1 = workload, $2 = suffix: FETCH_SIZE and WRITE_SIZE in separate passes (TCC counters do not fit one pass)
  cd /tmp && export TMPDIR=/tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/${TAG}_pmc_${C}$2 -o pmc --output-format csv -- python $R/bench.py --workload $1 --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  done
  cd $R
  ST=gpurun_out/${TAG}_kernel_stats$2.txt
  python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_FETCH_SIZE$2 gpurun_out/${TAG}_pmc_WRITE_SIZE$2 $1 gpurun_out/${TAG}_pmc_traffic$2.json $ST
}
WL0=cfg4_dynerf_300k_1352x1014
if [ -z "$SKIP_PROF" ]; then prof_one $WL0 ""; fi
if [ -n "$PMC" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_pmc_sq -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  cd $R
  python tools/pmc_summary.py gpurun_out/${TAG}_pmc_sq gpurun_out/${TAG}_pmc_sq.txt > /dev/null
  head -8 gpurun_out/${TAG}_pmc_sq.txt | cut -c1-230
  # second set: where the vector-memory / LDS issue cycles and instruction counts of each kernel go (round 4: the forward kernel's idle MFMA cycles)
  cd /tmp
  timeout 240 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_pmc_sq2 -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  cd $R
  python tools/pmc_summary.py gpurun_out/${TAG}_pmc_sq2 gpurun_out/${TAG}_pmc_sq2.txt > /dev/null
  head -6 gpurun_out/${TAG}_pmc_sq2.txt | cut -c1-230
  pmc_one $WL0 ""
fi
for WL in $EXTRA_WORKLOADS; do
  # (one CPU frame: every workload's line carries its own parity against the oracle chain)
  timeout 600 python bench.py --workload $WL --steps 10 --warmup 3 --repeats 5 --cpu-frames 1 > gpurun_out/${TAG}_bench_$WL.json 2> gpurun_out/${TAG}_bench_$WL.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/${TAG}_bench_$WL.json')); print('$WL', round(d['value'],1), 'frames/s', d['config']['num_rendered'], d['kernels_ms_per_step'])" 2>&1 | cut -c1-500
  # (PMC_EXTRA=1: counter passes for every extra workload; PMC_EXTRA=<substring>: only for the workloads whose name contains it)
  if [ -n "$PMC_EXTRA" ] && { [ "$PMC_EXTRA" = "1" ] || [[ "$WL" == *"$PMC_EXTRA"* ]]; }; then prof_one $WL _$WL > /dev/null; pmc_one $WL _$WL; fi
done
if [ -n "$SHELL_SCENE" ]; then
  # the second scene statistic (bench.py --scene shell): its own bench line with oracle parity + rocprofv3 kernel stats
  timeout 300 python bench.py --scene shell --steps 10 --warmup 3 --repeats 5 --cpu-frames 1 --no-extras > gpurun_out/${TAG}_bench_shell.json 2> gpurun_out/${TAG}_bench_shell.err
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_shell -o trace --output-format csv -- python $R/bench.py --scene shell --steps 20 --warmup 4 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  cd $R
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/${TAG}_prof_shell/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    out = ["%-60s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct")]
    for r in rows[:28]:
        out.append("%-60s %8s %12.3f %10.2f %7s" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    open("gpurun_out/${TAG}_kernel_stats_shell.txt", "w").write("\n".join(out) + "\n")
PY
  if [ -n "$PMC" ]; then   # the two SQ counter sets on the shell scene (long un-terminated lists: where the blending kernels spend their cycles)
    cd /tmp
    timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_pmc_sq_shell -o pmc --output-format csv -- python $R/bench.py --scene shell --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
    timeout 240 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_pmc_sq2_shell -o pmc --output-format csv -- python $R/bench.py --scene shell --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
    cd $R
    python tools/pmc_summary.py gpurun_out/${TAG}_pmc_sq_shell gpurun_out/${TAG}_pmc_sq_shell.txt > /dev/null
    python tools/pmc_summary.py gpurun_out/${TAG}_pmc_sq2_shell gpurun_out/${TAG}_pmc_sq2_shell.txt > /dev/null
    head -6 gpurun_out/${TAG}_pmc_sq_shell.txt | cut -c1-230
  fi
fi
# drop the bulky raw traces from what travels back (summaries stay)
rm -rf gpurun_out/${TAG}_prof*/ gpurun_out/${TAG}_pmc_FETCH_SIZE* gpurun_out/${TAG}_pmc_WRITE_SIZE* gpurun_out/${TAG}_pmc_sq/ gpurun_out/${TAG}_pmc_sq2/ gpurun_out/${TAG}_pmc_sq_shell/ gpurun_out/${TAG}_pmc_sq2_shell/ 2>/dev/null
exit 0
