mkdir -p gpurun_out
timeout 1500 python tools/train_step_noise_floor.py gpurun_out/r06_train_step_noise_floor.json 6 > gpurun_out/noise_floor.log 2>&1; echo "rc=$?" >> gpurun_out/noise_floor.log
tail -30 gpurun_out/noise_floor.log
