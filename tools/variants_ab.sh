#!/bin/bash
# tools/variants_ab.sh TAG -- A/B of development builds (tools/_variants/libfdgs_*.so, tools/build_variant.sh) and load-time knobs IN THE FRAME
# (bench.py --no-extras --no-cpu-baseline), cube and shell scene, two interleaved rounds on one box.  Output: gpurun_out/TAG_variants.txt
TAG=${1:-ab}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_variants.txt
: > $OUT
run() {  # name, env assignments..., -- bench args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 6 --repeats 12 "$@" > /tmp/v.json 2> /tmp/v.err || { echo "$name FAILED: $(tail -2 /tmp/v.err)" >> $OUT; return; }
  python - "$name" "$@" >> $OUT <<'PY'
import json, sys
d = json.load(open("/tmp/v.json"))
k = d["kernels_ms_per_step"]
print("%-34s %-14s %8.1f fps  %.4f ms  p10 %.4f p90 %.4f | D1 %.4f D2 %.4f D3 %.4f D4 %.4f rbwd %.4f rfwd %.4f" % (
    sys.argv[1], " ".join(sys.argv[2:]), d["value"], d["ms_per_step"], d["p10_ms_per_step"], d["p90_ms_per_step"], k.get("deform_fwd", 0), k.get("deform_bwd_data", 0),
    k.get("deform_wgrad", 0), k.get("deform_plane_grad", 0), k.get("render_bwd", 0), k.get("render_fwd", 0)))
PY
}
for round in 1 2; do
  for scene in cube shell; do
    run base FDGS_X=0 -- --scene $scene
    for v in $(ls tools/_variants/libfdgs_*.so 2>/dev/null); do
      run "$(basename $v .so)" FDGS_LIB=$R/$v -- --scene $scene
    done
    run rbwd_ppl2 FDGS_RBWD_PPL=2 -- --scene $scene
    run d1_form32 FDGS_D1_FORM=32 -- --scene $scene
    run binning_exact FDGS_BINNING=exact -- --scene $scene
  done
done
cat $OUT
