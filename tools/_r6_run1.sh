set -x
mkdir -p gpurun_out
for i in 1 2 3 4; do
  FDGS_TRAIN_STEP_JSON=gpurun_out/ts_$i.json timeout 900 python -m pytest tests/test_zz_gpu_reference_train_step.py -x -q -m gpu -s > gpurun_out/ts_$i.log 2>&1
  echo "run $i rc=$?" >> gpurun_out/ts_summary.txt
done
timeout 600 python -m pytest tests/test_gpu_densify.py -x -q -m gpu > gpurun_out/densify.log 2>&1; echo "densify rc=$?" >> gpurun_out/ts_summary.txt
