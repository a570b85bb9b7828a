mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_deform.py -x -q -m gpu > gpurun_out/t4.log 2>&1; echo "rc=$?" >> gpurun_out/t4.log
tail -25 gpurun_out/t4.log
