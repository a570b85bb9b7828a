# timing experiments: development variants of the weight-stationary D2 with pieces switched off (results wrong, time informative)
mkdir -p gpurun_out; OUT=gpurun_out/${1:-r06c_d2x}.txt; : > $OUT
for v in "" $(ls tools/_variants/libfdgs_x_*.so); do
  for sc in shell; do
    FDGS_LIB=${v:+$PWD/$v} timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 6 --repeats 6 --scene $sc > /tmp/v.json 2>/tmp/v.err
    python - "${v:-base}" $sc >> $OUT <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/v.json")); k = d["kernels_ms_per_step"]
    print("%-44s %-6s D2 %.4f  frame %.4f" % (sys.argv[1], sys.argv[2], k.get("deform_bwd_data", 0), d["ms_per_step"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done
done
cat $OUT
