mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_deform.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/hm_t2.log 2>&1; tail -2 gpurun_out/hm_t2.log
for f in 0 32; do for a in "--workload cfg3_hypernerf_300k_536x960" "--scene shell" "--scene cube"; do
  FDGS_D2_FORM=$f timeout 300 python bench.py $a --no-extras --no-cpu-baseline --steps 20 --warmup 6 --repeats 8 > /tmp/v.json 2>/tmp/v.err
  python - "d2_form=$f $a" <<'PY'
import json, sys
d = json.loads(open("/tmp/v.json").read().strip().splitlines()[-1]); k = d["kernels_ms_per_step"]
print("%-52s D2 %.4f D1 %.4f frame %.4f  %.1f fps" % (sys.argv[1], k.get("deform_bwd_data", 0), k.get("deform_fwd", 0), d["ms_per_step"], d["value"]))
PY
done; done
