cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_deform.py tests/test_gpu_fullsize.py -m gpu -q -x -rA > gpurun_out/r4c_pytest.log 2>&1; tail -5 gpurun_out/r4c_pytest.log
for F in 16 32; do
FDGS_D1_FORM=$F timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > gpurun_out/r4c_bench_form$F.json 2> gpurun_out/r4c_bench_form$F.err
python -c "
import json; d=json.load(open('gpurun_out/r4c_bench_form$F.json')); print('FORM $F', round(d['value'],1), d['ms_per_step'], d['kernels_ms_per_step'])"
done
