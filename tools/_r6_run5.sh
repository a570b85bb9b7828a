mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_render_branches.py -x -q -m gpu > gpurun_out/t5.log 2>&1; echo "rc=$?" >> gpurun_out/t5.log
tail -12 gpurun_out/t5.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_io.json 2> gpurun_out/bench_io.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_io.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
for k in ("random_order","random_order_implicit_permutation_off","shell_scene","all_tiles_backward"):
    print(k, {kk:vv for kk,vv in d[k].items() if kk in ("frames_per_s","ms_per_step","kernels_ms_per_step")})
PY
