mkdir -p gpurun_out
V=$PWD/tools/_variants/libfdgs_${1}.so
FDGS_LIB=$V timeout 900 python -m pytest tests/test_gpu_deform.py tests/test_gpu_fullsize.py -x -q -m gpu -k "parity or golden or fullsize or end_to_end" > gpurun_out/d1ab_t.log 2>&1; tail -2 gpurun_out/d1ab_t.log
for r in 1 2 3; do for v in "" $V; do
  FDGS_LIB=$v timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 6 --repeats 8 > /tmp/v.json 2>/tmp/v.err
  python - "${v:-base}" <<'PY'
import json, sys
d = json.loads(open("/tmp/v.json").read().strip().splitlines()[-1]); k = d["kernels_ms_per_step"]
print("%-40s D1 %.4f  frame %.4f  %.1f fps  frac %.3f" % (sys.argv[1][-28:], k.get("deform_fwd", 0), d["ms_per_step"], d["value"], d["roofline"]["frac"]))
PY
done; done
