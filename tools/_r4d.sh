cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for i in 1 2; do for V in dp0 dp1; do FDGS_LIB=$PWD/tools/_variants/libfdgs_$V.so python tools/d1_ab.py; done; done
} > gpurun_out/r4d_d1_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4d_d1_ab.txt
timeout 600 python -m pytest tests/test_gpu_deform.py -m gpu -q -x 2>&1 | tail -3
