mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s > gpurun_out/t6.log 2>&1; echo "rc=$?" >> gpurun_out/t6.log
grep -E "RAW vs|passed|failed|Error|assert" gpurun_out/t6.log | cut -c1-700 | head -30
