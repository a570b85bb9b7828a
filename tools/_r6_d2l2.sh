# L2-side request counters of the backward-data kernel, weight-stationary vs 32-row form, shell scene
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out
for f in 0 32; do
  FDGS_D2_FORM=$f timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $R/gpurun_out/d2l2_$f -o pmc --output-format csv -- python $R/bench.py --scene shell --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  cd $R; python tools/pmc_summary.py gpurun_out/d2l2_$f gpurun_out/d2l2_$f.txt > /dev/null 2>&1; echo "== d2_form $f"; head -1 gpurun_out/d2l2_$f.txt | cut -c1-220; grep "deform_bwd_data\|deform_mlp_ws_kernel<2, 2, true" gpurun_out/d2l2_$f.txt | cut -c1-220; rm -rf gpurun_out/d2l2_$f; cd /tmp
done
