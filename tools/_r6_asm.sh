mkdir -p gpurun_out
V=$PWD/tools/_variants/libfdgs_${1:-asm2}.so
FDGS_LIB=$V timeout 600 python -m pytest tests/test_gpu_deform.py tests/test_gpu_fullsize.py -x -q -m gpu -k "dead_tile or fullsize" > gpurun_out/asm_t.log 2>&1; tail -2 gpurun_out/asm_t.log
for r in 1 2; do for sc in shell cube; do for v in $PWD/tools/_variants/libfdgs_asm2.so $V; do
  FDGS_LIB=$v timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 6 --repeats 8 --scene $sc > /tmp/v.json 2>/tmp/v.err
  python - "${v:-base}" $sc <<'PY'
import json, sys
d = json.loads(open("/tmp/v.json").read().strip().splitlines()[-1]); k = d["kernels_ms_per_step"]
print("%-50s %-6s D2 %.4f  frame %.4f  %.1f fps" % (sys.argv[1][-30:], sys.argv[2], k.get("deform_bwd_data", 0), d["ms_per_step"], d["value"]))
PY
done; done; done
