mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_deform.py tests/test_gpu_fullsize.py -x -q -m gpu -k "dead_tile or fullsize or end_to_end or forward_backward_parity" > gpurun_out/d2ws_t7.log 2>&1; tail -3 gpurun_out/d2ws_t7.log
bash tools/_r6_d2ab.sh ${1:-r06c_d2g}
FDGS_LIB=$PWD/tools/_variants/libfdgs_d2wsprof.so timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 4 --scene shell 2>&1 | grep -A9 "D2-ws profile" | grep -v metric
