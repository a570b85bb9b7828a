mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/glds_probe.hip -o /tmp/glds_probe 2>/dev/null && /tmp/glds_probe > gpurun_out/glds_probe.txt 2>&1; cat gpurun_out/glds_probe.txt
timeout 1200 python -m pytest tests/test_gpu_deform.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/d2ws_t1.log 2>&1; tail -30 gpurun_out/d2ws_t1.log
