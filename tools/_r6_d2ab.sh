# A/B of the backward-data kernel forms in the frame: d2_form 0 (weight-stationary where it applies) vs 32, cube and shell, config 4 and 5
TAG=${1:-r06c_d2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_forms_ab.txt
: > $OUT
run() {
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 6 --repeats 12 "$@" > /tmp/v.json 2> /tmp/v.err || { echo "$name FAILED: $(tail -2 /tmp/v.err)" >> $OUT; return; }
  python - "$name" "$@" >> $OUT <<'PY'
import json, sys
d = json.load(open("/tmp/v.json"))
k = d["kernels_ms_per_step"]
print("%-14s %-44s %8.1f fps  %.4f ms  p10 %.4f p90 %.4f | D1 %.4f D2 %.4f D3 %.4f D4 %.4f rbwd %.4f rfwd %.4f rowlist %.4f" % (
    sys.argv[1], " ".join(sys.argv[2:]), d["value"], d["ms_per_step"], d["p10_ms_per_step"], d["p90_ms_per_step"], k.get("deform_fwd", 0), k.get("deform_bwd_data", 0),
    k.get("deform_wgrad", 0), k.get("deform_plane_grad", 0), k.get("render_bwd", 0), k.get("render_fwd", 0), k.get("row_list", 0)))
PY
}
for round in 1 2; do
  for scene in cube shell; do
    run d2_ws FDGS_D2_FORM=0 -- --scene $scene
    run d2_32 FDGS_D2_FORM=32 -- --scene $scene
  done
done
run d2_ws FDGS_D2_FORM=0 -- --workload cfg5_stress_2M_2048x2048
run d2_32 FDGS_D2_FORM=32 -- --workload cfg5_stress_2M_2048x2048
cat $OUT
