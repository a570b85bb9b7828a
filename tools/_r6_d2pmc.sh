# SQ counters of the weight-stationary backward-data kernel on the shell scene (two passes; --pmc alone with --kernel-trace)
TAG=${1:-r06c_d2pmc}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out
B="python $R/bench.py --scene shell --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $R/gpurun_out/${TAG}_a -o pmc --output-format csv -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU -d $R/gpurun_out/${TAG}_b -o pmc --output-format csv -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_WAIT_IFETCH -d $R/gpurun_out/${TAG}_c -o pmc --output-format csv -- $B > /dev/null 2>&1
cd $R
for p in a b c; do python tools/pmc_summary.py gpurun_out/${TAG}_$p gpurun_out/${TAG}_$p.txt > /dev/null 2>&1; head -1 gpurun_out/${TAG}_$p.txt | cut -c1-250; grep "deform_bwd_data\|deform_mlp_ws_kernel<2, 2, true" gpurun_out/${TAG}_$p.txt | cut -c1-250; done
rm -rf gpurun_out/${TAG}_a gpurun_out/${TAG}_b gpurun_out/${TAG}_c
