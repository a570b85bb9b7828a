# tools/frame_timeline.sh -- rocprofv3 --kernel-trace of a short bench run -> gpurun_out/r05x_frame_timeline_cfg4.txt (tools/frame_timeline.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tl_prof -o trace --output-format csv -- python $R/bench.py --steps 20 --warmup 6 --repeats 2 --no-cpu-baseline --no-extras > $R/gpurun_out/tl_bench.json 2> $R/gpurun_out/tl.err
cd $R
python tools/frame_timeline.py gpurun_out/tl_prof gpurun_out/r05x_frame_timeline_cfg4.txt
rm -rf gpurun_out/tl_prof
python -c "
import json; d=json.load(open('gpurun_out/tl_bench.json')); print(d['value'], d['ms_per_step'], d['kernels_ms_per_step'])"
