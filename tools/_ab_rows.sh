cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; OUT=gpurun_out/r05_row_compact_ab.txt; : > $OUT
timeout 600 python -m pytest tests/test_gpu_deform.py tests/test_gpu_render_branches.py -x -q 2>&1 | tail -15 >> $OUT
one() { env "$2" timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 6 --repeats 12 $3 > /tmp/v.json 2>/tmp/v.err || { echo "$1 FAILED $(tail -2 /tmp/v.err)" >> $OUT; return; }
python - $1 $3 >> $OUT <<'PY'
import json,sys
d=json.load(open("/tmp/v.json")); k=d["kernels_ms_per_step"]
print("%-8s %-14s %8.1f fps %.4f ms | D2 %.4f D3 %.4f D4 %.4f compact %.4f gather %.4f K8 %.4f | live %s" % (sys.argv[1], " ".join(sys.argv[2:]), d["value"], d["ms_per_step"], k.get("deform_bwd_data",0), k.get("deform_wgrad",0), k.get("deform_plane_grad",0), k.get("tile_compact",0), k.get("row_gather",0), k.get("preprocess_bwd",0), d.get("mlp",{}).get("backward_live_tiles")))
PY
}
for r in 1 2; do
  one rows1 FDGS_ROW_COMPACT=1 ""; one rows0 FDGS_ROW_COMPACT=0 ""
  one rows1 FDGS_ROW_COMPACT=1 "--scene shell"; one rows0 FDGS_ROW_COMPACT=0 "--scene shell"
done
cat $OUT
