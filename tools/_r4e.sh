R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for F in 16 32; do
FDGS_D1_FORM=$F FDGS_D16_SKEW=0 timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/r4e_pmc_$F -o pmc --output-format csv -- python $R/tools/d1_ab.py > /dev/null 2>&1
FDGS_D1_FORM=$F FDGS_D16_SKEW=0 timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA -d $R/gpurun_out/r4e_pmc2_$F -o pmc --output-format csv -- python $R/tools/d1_ab.py > /dev/null 2>&1
FDGS_D1_FORM=$F FDGS_D16_SKEW=0 timeout 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $R/gpurun_out/r4e_pmc3_$F -o pmc --output-format csv -- python $R/tools/d1_ab.py > /dev/null 2>&1
FDGS_D1_FORM=$F FDGS_D16_SKEW=0 timeout 240 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum -d $R/gpurun_out/r4e_pmc4_$F -o pmc --output-format csv -- python $R/tools/d1_ab.py > /dev/null 2>&1
cd $R
for P in pmc pmc2 pmc3 pmc4; do python tools/pmc_summary.py gpurun_out/r4e_${P}_$F gpurun_out/r4e_${P}_$F.txt > /dev/null 2>&1; grep -E "kernel|deform_fwd" gpurun_out/r4e_${P}_$F.txt | cut -c1-260; done
cd /tmp
done
rm -rf $R/gpurun_out/r4e_pmc*_16 $R/gpurun_out/r4e_pmc*_32
