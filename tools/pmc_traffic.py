"""Merge the rocprofv3 FETCH_SIZE and WRITE_SIZE passes of ONE bench command into profiles-ready JSON, keyed by the
workload and by the build (sha256 of libfdgs.so) they were collected with -- bench.py refuses artefacts whose keys differ.

usage: python tools/pmc_traffic.py FETCH_DIR WRITE_DIR WORKLOAD OUT.json [KERNEL_STATS.csv]
With a kernel-stats CSV (rocprofv3 --kernel-trace --stats of the same command) the per-kernel measured HBM GB/s
((2 x FETCH + WRITE) / average duration) and the whole-frame figure are added (north_star config 5: "rocprof GB/s reported").
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALIAS = {"radix_digit_scan": "radix_scan", "scan_chunk": "scan_tiles", "scan_add": "scan_tiles", "adam": "adam_step",
         "deform_plane_grad_mfma": "deform_plane_grad", "deform_fwd16": "deform_fwd", "deform_mlp_ws": "deform_fwd", "deform_bwd_data_ws": "deform_bwd_data", "pack_weights16": "pack_weights"}


def src_sha16():
    """Hash of every kernel source the library is built from: the key that ties a counter artefact to a build (a hash of the
    binary would change with the build path; bench.py computes the same value)."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "4dgaussians_amd", "csrc", "*"))) + [os.path.join(ROOT, "include", "fdgs.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("fdgs::", "")
    name = re.sub(r"<.*", "", name)
    name = re.sub(r"_kernel$", "", name)
    return ALIAS.get(name, name)


def per_launch(d, counter):
    agg, calls = defaultdict(float), defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = short(row.get("Kernel_Name", ""))
            agg[k] += float(row["Counter_Value"])
            calls[k].add(row.get("Dispatch_Id"))
    return {k: (v / max(len(calls[k]), 1), len(calls[k])) for k, v in agg.items()}


def main():
    fetch_dir, write_dir, workload, out = sys.argv[1:5]
    stats = sys.argv[5] if len(sys.argv) > 5 else None
    if fetch_dir == "--merge":      # add durations to an existing artefact: pmc_traffic.py --merge IN.json WORKLOAD OUT.json STATS
        old = json.load(open(write_dir))
        fe = {k: (v["FETCH_SIZE_KB_per_launch"], v.get("launches_profiled", 0)) for k, v in old.items() if not k.startswith("_")}
        wr = {k: (v["WRITE_SIZE_KB_per_launch"], v.get("launches_profiled", 0)) for k, v in old.items() if not k.startswith("_")}
        fe = {ALIAS.get(k, k): v for k, v in fe.items()}
        wr = {ALIAS.get(k, k): v for k, v in wr.items()}
    else:
        fe, wr = per_launch(fetch_dir, "FETCH_SIZE"), per_launch(write_dir, "WRITE_SIZE")
    res = {"_workload": workload, "_src_sha16": src_sha16(),
           "_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --workload "
                      + workload + " --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras, MI355X",
           "_units": "KB per launch as reported by rocprofv3 (TCC_EA0 request counters x 64 B); gfx950 reports HALF of the bytes of "
                     "16-B/lane streaming reads (MI355X_MICROARCH.md, HBM section): consumers double FETCH_SIZE"}
    dur = {}
    if stats and os.path.exists(stats) and stats.endswith(".csv"):
        for r in csv.DictReader(open(stats)):
            k = short(r["Name"])
            c, t = dur.get(k, (0, 0.0))
            dur[k] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
    elif stats and os.path.exists(stats):          # the text summary tools/gpu_round.sh writes (kernel, calls, total_ms, avg_us, pct)
        for line in open(stats).read().splitlines()[1:]:
            name, rest = line[:60], line[60:].split()
            if len(rest) < 3:
                continue
            k = short(name.strip())
            c, t = dur.get(k, (0, 0.0))
            dur[k] = (c + int(rest[0]), t + float(rest[1]) * 1e6)
    frame_bytes = frame_ns = 0.0
    for k in sorted(set(fe) | set(wr)):
        f, nf = fe.get(k, (0.0, 0))
        w, nw = wr.get(k, (0.0, 0))
        e = {"FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "launches_profiled": max(nf, nw)}
        if k in dur and dur[k][0]:
            avg_ns = dur[k][1] / dur[k][0]
            hbm = (2 * f + w) * 1024
            e["avg_us"] = avg_ns / 1e3
            e["hbm_GBps"] = hbm / avg_ns
        res[k] = e
    if dur:
        # whole frame: bytes and time of every kernel weighted by its launches per profiled step
        steps = None
        for probe in ("render_fwd", "preprocess_fwd", "deform_fwd"):
            if probe in dur:
                steps = dur[probe][0]
                break
        if steps:
            for k, e in res.items():
                if k.startswith("_") or k not in dur:
                    continue
                lps = dur[k][0] / steps
                frame_bytes += (2 * e["FETCH_SIZE_KB_per_launch"] + e["WRITE_SIZE_KB_per_launch"]) * 1024 * lps
                frame_ns += dur[k][1] / steps
            res["_frame"] = {"hbm_bytes_per_frame": frame_bytes, "kernel_ms_per_frame": frame_ns / 1e6,
                             "hbm_GBps_over_kernel_time": frame_bytes / frame_ns if frame_ns else None,
                             "frac_of_8TBps": frame_bytes / frame_ns / 8000.0 if frame_ns else None}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res.get("_frame", {})))


if __name__ == "__main__":
    main()
