for r in 1 2 3; do for v in "" $PWD/tools/_variants/libfdgs_${1}.so; do
  FDGS_LIB=$v timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 6 --repeats 8 > /tmp/v.json 2>/tmp/v.err
  python - "${v:-base}" <<'PY'
import json, sys
d = json.loads(open("/tmp/v.json").read().strip().splitlines()[-1]); k = d["kernels_ms_per_step"]
print("%-40s D1 %.4f  frame %.4f  %.1f fps" % (sys.argv[1][-28:], k.get("deform_fwd", 0), d["ms_per_step"], d["value"]))
PY
done; done
