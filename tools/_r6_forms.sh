mkdir -p gpurun_out
for WL in cfg2_dnerf_100k_800x800 cfg3_hypernerf_300k_536x960 cfg5_stress_2M_2048x2048; do
  for F in 8 16; do
    FDGS_D1_FORM=$F timeout 600 python bench.py --workload $WL --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-extras > gpurun_out/forms_${WL}_$F.json 2> gpurun_out/forms_${WL}_$F.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/forms_${WL}_$F.json").read().strip().splitlines()[-1])
k=d["kernels_ms_per_step"]
print("$WL form $F:", round(d["value"],1), "frames/s; deform_fwd", k.get("deform_fwd"), "gather", k.get("deform_gather"), "pack", k.get("pack_weights"))
PY
  done
done
