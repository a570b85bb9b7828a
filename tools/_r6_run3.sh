mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_zz_gpu_reference_train_step.py -x -q -m gpu -s > gpurun_out/t3.log 2>&1; echo "rc=$?" >> gpurun_out/t3.log
tail -15 gpurun_out/t3.log
for m in auto capacity exact; do FDGS_BINNING=$m timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$m.json 2> gpurun_out/bench_$m.err; echo "$m rc=$?"; done
python - <<'PY'
import json
for m in ("auto","capacity","exact"):
    try:
        d=json.loads(open(f"gpurun_out/bench_{m}.json").read().strip().splitlines()[-1])
        print(m, d["value"], d["ms_per_step"], d["binning"]["mode"], d["binning"]["capacity_reruns"], d.get("roofline"))
    except Exception as e: print(m, "ERR", e)
PY
