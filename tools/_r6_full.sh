mkdir -p gpurun_out
n=${1:-1}
for i in $(seq 1 $n); do
  timeout 1700 python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_full_$i.log 2>&1; echo "run $i rc=$?" >> gpurun_out/gpu_full_summary.txt
  tail -3 gpurun_out/gpu_full_$i.log
done
cat gpurun_out/gpu_full_summary.txt
