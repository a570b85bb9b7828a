#!/usr/bin/env python
"""tools/tune.py -- development sweep of libfdgs' environment tuning knobs on one GPU (not part of the product).

Builds the bench workload once, then for every knob setting runs a few timed frames with the per-kernel HIP-event
timing of libfdgs and prints one table row per setting.   python tools/tune.py [--workload NAME] [--steps K]
"""
import argparse
import ctypes
import importlib
import itertools
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SWEEPS = [
    ("default", {}),
    ("wg_trunk=1", {"FDGS_WGRAD_TRUNK": "1"}),
    ("wg_trunk=2", {"FDGS_WGRAD_TRUNK": "2"}),
    ("wg_trunk=4", {"FDGS_WGRAD_TRUNK": "4"}),
    ("wg_trunk=6", {"FDGS_WGRAD_TRUNK": "6"}),
    ("small_heads=0", {"FDGS_SMALL_HEADS": "0"}),
    ("pg_lds=0", {"FDGS_PG_LDS": "0"}),
    ("pg_wgs=256", {"FDGS_PG_WGS": "256"}),
    ("pg_wgs=1024", {"FDGS_PG_WGS": "1024"}),
    ("pg_wgs=2048", {"FDGS_PG_WGS": "2048"}),
    ("d2_wgs=128", {"FDGS_D2_WGS": "128"}),
    ("d2_wgs=512", {"FDGS_D2_WGS": "512"}),
    ("wgrad_wgs=64", {"FDGS_WGRAD_WGS": "64"}),
    ("wgrad_wgs=96", {"FDGS_WGRAD_WGS": "96"}),
    ("wgrad_wgs=128", {"FDGS_WGRAD_WGS": "128"}),
    ("wgrad_wgs=192", {"FDGS_WGRAD_WGS": "192"}),
    ("wgrad_wgs=512", {"FDGS_WGRAD_WGS": "512"}),
    ("d2_wgs=192", {"FDGS_D2_WGS": "192"}),
    ("d1_wgs=0", {"FDGS_D1_WGS": "0"}),
    ("default_again", {}),
]
KNOBS = ("FDGS_PG_LDS", "FDGS_PG_WGS", "FDGS_D2_WGS", "FDGS_WGRAD_WGS", "FDGS_WGRAD_TRUNK", "FDGS_SMALL_HEADS", "FDGS_D1_WGS")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg4_dynerf_300k_1352x1014")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import bench
    fdgs = importlib.import_module("4dgaussians_amd")
    syn = fdgs.synthetic
    dev = torch.device("cuda:0")
    L = fdgs._lib.lib()
    N, W, H, dcfg = bench.WORKLOADS[args.workload]
    pc = syn.SynthModel(N, dcfg, seed=6666, device=dev)
    fdgs.densify.spatial_reorder(pc)          # the order the train loop keeps (live-tile lists, windowed plane gradient)
    pipe = syn.PipelineParams()
    bg = torch.zeros(3, device=dev)
    cams = [c.to(dev) for c in syn.orbit_cameras(W, H, n=160)]
    target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(6666)).to(dev)
    params = [p for p in pc.parameters() if p.requires_grad]
    acc = torch.zeros(3, device=dev)
    dimg = torch.empty(3, H, W, device=dev)
    st, ptr = fdgs._lib.stream_ptr, fdgs._lib.ptr

    def step(i):
        cam = cams[i % len(cams)]
        for p_ in params:
            p_.grad = None
        res = fdgs.render(cam, pc, pipe, bg, stage="fine")
        img = res["render"]
        acc.zero_()
        fdgs._lib.check(L.fdgs_l1_stats(st(), img.numel(), ptr(img), ptr(target), 1.0 / img.numel(), ptr(dimg), ptr(acc)))
        img.backward(dimg)

    rows = []
    for name, env in SWEEPS:
        if args.only and args.only not in name:
            continue
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(3 + i)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.steps * 1e3
        L.fdgs_timing_enable(1)
        for i in range(args.steps):
            step(3 + i)
        buf = ctypes.create_string_buffer(1 << 16)
        fdgs._lib.check(L.fdgs_timing_report(buf, len(buf), 1))
        L.fdgs_timing_enable(0)
        kern = {}
        for line in buf.value.decode().strip().splitlines():
            nm, cnt, tot = line.split()
            kern[nm] = float(tot) / args.steps
        rows.append((name, wall, kern))
        print(json.dumps({"setting": name, "ms_per_frame": round(wall, 3), "fps": round(1e3 / wall, 1),
                          "kernels_ms": {k: round(v, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])}}), flush=True)
    keys = ["deform_fwd", "deform_bwd_prep", "deform_bwd_data", "deform_wgrad", "deform_plane_grad", "render_fwd", "render_bwd",
            "radix_scan", "scan_tiles", "radix_scatter"]
    print("\n%-16s %8s " % ("setting", "frame") + " ".join("%10s" % k[-10:] for k in keys))
    for name, wall, kern in rows:
        print("%-16s %8.3f " % (name, wall) + " ".join("%10.4f" % kern.get(k, float("nan")) for k in keys))


if __name__ == "__main__":
    main()
