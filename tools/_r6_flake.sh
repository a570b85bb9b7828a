# flake hunt: the whole GPU suite (no -x: every failure is seen) + the long harness file on its own, repeated
mkdir -p gpurun_out; rm -f gpurun_out/flake_summary.txt
nf=${1:-2}; nz=${2:-4}
for i in $(seq 1 $nf); do
  timeout 1700 python -m pytest tests/ -q -m gpu --durations=8 > gpurun_out/flake_full_$i.log 2>&1; echo "full $i rc=$?" >> gpurun_out/flake_summary.txt
  tail -2 gpurun_out/flake_full_$i.log
done
for i in $(seq 1 $nz); do
  timeout 900 python -m pytest tests/test_zz_gpu_reference_train_step.py tests/test_gpu_fullsize.py -q -m gpu -k "train_step or implicit" > gpurun_out/flake_zz_$i.log 2>&1; echo "zz $i rc=$?" >> gpurun_out/flake_summary.txt
  tail -2 gpurun_out/flake_zz_$i.log
done
cat gpurun_out/flake_summary.txt
