cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > gpurun_out/r4h_bench.json 2> gpurun_out/r4h_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r4h_bench.json')); print(round(d['value'],1), round(d['ms_per_step'],4), d['kernels_ms_per_step'], d['loss'])"
tail -3 gpurun_out/r4h_bench.err
timeout 600 python -m pytest tests/test_gpu_loss.py -m gpu -q -x 2>&1 | tail -2
