cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
FDGS_D1_FORM=32 FDGS_D1_PACK32=1 python tools/d1_ab.py
FDGS_D1_FORM=32 FDGS_D1_PACK32=0 python tools/d1_ab.py
FDGS_D1_FORM=16 python tools/d1_ab.py
FDGS_D1_FORM=32 FDGS_D1_PACK32=1 python tools/d1_ab.py
FDGS_D1_FORM=32 FDGS_D1_PACK32=0 python tools/d1_ab.py
} > gpurun_out/r4d_d1_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4d_d1_ab.txt
timeout 600 python -m pytest tests/test_gpu_deform.py -m gpu -q -x 2>&1 | tail -3
FDGS_D1_FORM=16 timeout 600 python -m pytest tests/test_gpu_deform.py -m gpu -q -x 2>&1 | tail -3
