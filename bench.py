#!/usr/bin/env python
"""bench.py -- train-step frames/s of the MI355X-native 4D-Gaussian render path (BASELINE.json metric).

One step = one frame: render() forward (HexPlane + deformation MLP -> projection -> binning/sort -> alpha blending) and
the full backward to every Gaussian parameter and every deformation parameter, for the L1-loss gradient against a
fixed random target, at BASELINE config 4 shape (DyNeRF cook_spinach: 300k Gaussians, 1352x1014, dynerf deformation
config, all five heads).  N GPUs = N independent frames per step (weak scaling) + one RCCL all-reduce of the loss
statistics.  Synthetic scene: no datasets/checkpoints exist offline.

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or, when
WORLD_SIZE is unset, bench.py spawns its N ranks itself (parallel.spawn_local: one process per GPU, rank r -> cuda:r, RCCL
rendezvous on 127.0.0.1) and fails with a clear message when fewer than N GPUs are visible.
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import math
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12        # B/s spec (MI355X_MICROARCH.md); 6.29e12 measured copy
MFMA_F32_PEAK = 157.3e12  # FLOP/s, v_mfma_f32_32x32x2_f32

WORKLOADS = {
    # name: (N gaussians, W, H, deformation config)
    "cfg4_dynerf_300k_1352x1014": (300_000, 1352, 1014, "dynerf_default"),
    "cfg2_dnerf_100k_800x800": (100_000, 800, 800, "dnerf_bouncingballs"),
    "cfg3_hypernerf_300k_536x960": (300_000, 536, 960, "hypernerf_default"),
    "cfg5_stress_2M_2048x2048": (2_000_000, 2048, 2048, "dynerf_default"),
    "tiny": (20_000, 400, 400, "dynerf_default"),
}


def mlp_flops_fwd(cfg):
    F = cfg["kplanes_config"]["output_coordinate_dim"] * len(cfg["multires"])
    W = cfg["net_width"]
    ks = [k for k, off in zip((3, 3, 4, 1, 48), (cfg["no_dx"], cfg["no_ds"], cfg["no_dr"], cfg["no_do"], cfg["no_dshs"])) if not off]
    return 2 * F * W + sum(2 * W * W + 2 * W * k for k in ks)


def algorithmic_bytes(N, R, Px, G, d):
    """SURVEY.md 8(d) per-stage algorithmic HBM bytes of one fwd+bwd frame."""
    p = 6
    return dict(deform_fwd=N * (236 + 44 + 192 * d) + G, deform_bwd=N * (236 + 44 + 192 * d) + N * 236 + 2 * G,
                preprocess_fwd=N * (236 + 76), binning=N * 8 + R * 12 + R * 12 * 2 * p + R * 8,
                render_fwd=R * 44 + Px * 24, render_bwd=R * 44 + Px * 20 + R * 36 * 2, preprocess_bwd=N * (36 + 76 + 236 + 236))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="cfg4_dynerf_300k_1352x1014", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=4)
    ap.add_argument("--no-train-step", action="store_true", help="skip the secondary full-iteration measurement")
    ap.add_argument("--order", default="hilbert", choices=["hilbert", "morton", "random"],
                    help="order of the Gaussian set: hilbert = as fdgs.densify.spatial_reorder leaves it after every densification "
                         "(the order the train loop runs in), random = the generator's order")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of exactly --steps steps each; value = median")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: spawn the ranks ourselves (one process per GPU); every child re-enters run() with the torchrun
        # environment contract, rank 0 prints the JSON line
        par = importlib.import_module("4dgaussians_amd.parallel")
        n_vis = torch.cuda.device_count()
        if n_vis < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_vis} GPU(s) visible on this node "
                             "(one rank per GPU; there is no CPU fallback and ranks never share a device)")
        par.spawn_local(args.gpus, _spawned, (vars(args),))
        return
    run(args)


def _spawned(argdict):
    run(argparse.Namespace(**argdict))


def run(args):
    fdgs = importlib.import_module("4dgaussians_amd")
    par, syn = fdgs.parallel, fdgs.synthetic
    rank, world, dev = par.init_from_env()
    if dev.type != "cuda":
        raise SystemExit("bench.py needs a GPU (the render path has no CPU fallback)")
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    ranks_seen = par.ranks_seen(dev)          # all-reduce of ones over RCCL: every rank really joined the job
    if ranks_seen != world:
        raise SystemExit(f"RCCL saw {ranks_seen} ranks, expected {world}")
    L = fdgs._lib.lib()
    N, W, H, dcfg = WORKLOADS[args.workload]
    pc = syn.SynthModel(N, dcfg, seed=6666, device=dev)
    if args.order != "random":
        fdgs.densify.spatial_reorder(pc, curve=args.order)
    pipe = syn.PipelineParams()
    bg = torch.zeros(3, device=dev)
    cams = [c.to(dev) for c in syn.orbit_cameras(W, H, n=160)]
    target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(6666)).to(dev)
    params = [p for p in pc.parameters() if p.requires_grad]
    acc = torch.zeros(3, device=dev)
    dimg = torch.empty(3, H, W, device=dev)
    st = fdgs._lib.stream_ptr
    ptr = fdgs._lib.ptr
    info = {}

    def step(i):
        cam = cams[par.frames_for_rank(len(cams), i, rank, world)]
        for p_ in params:
            p_.grad = None
        res = fdgs.render(cam, pc, pipe, bg, stage="fine")
        img = res["render"]
        acc.zero_()
        fdgs._lib.check(L.fdgs_l1_stats(st(), img.numel(), ptr(img), ptr(target), 1.0 / img.numel(), ptr(dimg), ptr(acc)))
        img.backward(dimg)
        par.allreduce_loss_stats(acc)   # the only cross-GPU exchange of the path
        info["radii"], info["vsp"], info["vis"] = res["radii"], res["viewspace_points"], res["visibility_filter"]
        return acc

    for i in range(args.warmup):
        step(i)
    # `repeats` timed regions of EXACTLY `steps` steps each, every one bracketed by barrier + synchronize on both sides and
    # reduced with MAX over ranks; value = median region (a single 20-step region is 80 ms -- too thin a sample to be robust
    # against one clock ramp or one stray host interrupt); all regions are listed in the JSON line
    regions, local_regions = [], []
    for r_ in range(max(args.repeats, 1)):
        par.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + r_ * args.steps + i)
        par.barrier(); torch.cuda.synchronize()
        local_regions.append(time.perf_counter() - t0)
        regions.append(par.max_over_ranks(local_regions[-1], dev))
    dt = sorted(regions)[len(regions) // 2]
    per_rank_ms = [x / args.steps * 1e3 for x in par.gather_floats(sorted(local_regions)[len(local_regions) // 2], dev)]
    l1, psnr = par.loss_from_stats(acc.clone())
    fps = world * args.steps / dt

    # ---- per-kernel timing with HIP events on the launch stream (separate instrumented pass over the same steps)
    kern = {}
    L.fdgs_timing_enable(1)
    for i in range(args.steps):
        step(args.warmup + i)
    import ctypes
    buf = ctypes.create_string_buffer(1 << 16)
    fdgs._lib.check(L.fdgs_timing_report(buf, len(buf), 1))
    L.fdgs_timing_enable(0)
    for line in buf.value.decode().strip().splitlines():
        name, cnt, tot = line.split()
        kern[name] = dict(launches_per_step=int(cnt) / args.steps, ms_per_step=float(tot) / args.steps,
                          avg_ms=float(tot) / int(cnt))
    # frame statistics for the byte model: one more forward to read num_rendered / visible count
    with torch.no_grad():
        cam = cams[par.frames_for_rank(len(cams), args.warmup, rank, world)]
        out = fdgs.deformation.deform(pc._deformation, pc._xyz, pc._scaling, pc._rotation, pc._opacity,
                                      shs_dc=pc._features_dc, shs_rest=pc._features_rest, time=cam.time, activate=True)
        rs = fdgs.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
                                                cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
        _, radii, _, state = fdgs.rasterizer.rasterize_forward(rs, out[0], out[4], None, out[3], out[1], out[2], None)
        R, V = int(state.num_rendered), int((radii > 0).sum())
    cfg = syn.DEFORM_CONFIGS[dcfg]
    G = sum(p_.numel() for n_, p_ in pc._deformation.named_parameters() if "grids" in n_) * 4
    d_sh = 0 if cfg["no_dshs"] else 1
    stage_bytes = algorithmic_bytes(N, R, W * H, G, d_sh)
    B_frame = sum(stage_bytes.values())
    flops_fwd = mlp_flops_fwd(cfg) * N
    ms_step = dt / args.steps * 1e3
    # dominant kernel and its roofline
    dom = max(kern, key=lambda k: kern[k]["ms_per_step"]) if kern else None
    # backward-data FLOPs: with saved activations (default) the kernel only does the backward products proper --
    # dh1 = W2^T G (2 W k), dW2 = G^T h1 (2 W k), dhid += W1^T dh1 (2 W^2) per head, dfeat = W0^T dhid (2 F W);
    # without them it also recomputes the forward (trunk + the heads' hidden layers)
    Fd, Wd = cfg["kplanes_config"]["output_coordinate_dim"] * len(cfg["multires"]), cfg["net_width"]
    ks_on = [k for k, off in zip((3, 3, 4, 1, 48), (cfg["no_dx"], cfg["no_ds"], cfg["no_dr"], cfg["no_do"], cfg["no_dshs"])) if not off]
    bwd_core = N * (sum(2 * Wd * Wd + 4 * Wd * k for k in ks_on) + 2 * Fd * Wd)
    recompute = N * (2 * Fd * Wd + sum(2 * Wd * Wd for _ in ks_on))
    saved_on = bool(fdgs.deformation.SAVE_ACTIVATIONS)
    mfma_flops = {"deform_fwd": flops_fwd, "deform_bwd_data": bwd_core if saved_on else bwd_core + recompute, "deform_wgrad": flops_fwd}
    hbm_bytes = {"render_bwd": stage_bytes["render_bwd"], "render_fwd": stage_bytes["render_fwd"],
                 "preprocess_fwd": stage_bytes["preprocess_fwd"], "preprocess_bwd": stage_bytes["preprocess_bwd"],
                 "deform_plane_grad": N * 12 + N * (cfg["kplanes_config"]["output_coordinate_dim"] * len(cfg["multires"])) * 4 + 2 * G,
                 "radix_scatter": R * 16, "radix_hist": R * 4, "expand_pairs": R * 8 + N * 16, "deform_bwd_prep": N * (59 * 8 + 256)}
    roof = None
    if dom:
        t_dom = kern[dom]["avg_ms"] * 1e-3
        lps = max(kern[dom]["launches_per_step"], 1e-9)
        if dom in mfma_flops:
            a = mfma_flops[dom] / lps / t_dom
            roof = dict(kernel=dom, bound="mfma", achieved=a / 1e12, peak=MFMA_F32_PEAK / 1e12, unit="TFLOP/s",
                        frac=a / MFMA_F32_PEAK, traffic=None, avg_launch_ms=kern[dom]["avg_ms"], launches_per_step=lps,
                        algorithmic_flops_per_launch=mfma_flops[dom] / lps,
                        note=("saved activations: backward products only" if dom == "deform_bwd_data" and saved_on else None))
        else:
            a = hbm_bytes.get(dom, 0) / lps / t_dom
            roof = dict(kernel=dom, bound="hbm", achieved=a / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=a / HBM_PEAK,
                        traffic=None, avg_launch_ms=kern[dom]["avg_ms"], launches_per_step=lps)

    if roof is not None:
        roof.update(pmc_traffic(dom, args.workload, lib_sha16(fdgs)))

    # ---- secondary measurement: one full fine-stage iteration of train.py (180-292) without data loading / densification:
    # render fwd+bwd + L1 statistics + HexPlane regulariser fwd+bwd (train.py:208-211) + optimizer step (:291-292), the
    # last two through the fused kernels of the "next" rows.  lr = 0 keeps the scene identical from step to step.
    train = None
    if not args.no_train_step:
        opt = fdgs.FusedAdam(pc.optimizer_groups(lr=0.0), lr=0.0, eps=1e-15)
        dstat = types.SimpleNamespace(xyz_gradient_accum=torch.zeros(N, 1, device=dev), denom=torch.zeros(N, 1, device=dev),
                                      max_radii2D=torch.zeros(N, device=dev))

        def train_iter(i):
            step(i)
            fdgs.densify.add_densification_stats(dstat, info["vsp"].grad, info["vis"], info["radii"])   # train.py:259-262
            reg = fdgs.compute_regulation(pc, 0.01, 0.0001, 0.0001)     # arguments/__init__.py:85-87 defaults
            reg.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)

        for i in range(args.warmup):
            train_iter(i)
        par.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            train_iter(args.warmup + i)
        par.barrier(); torch.cuda.synchronize()
        dt_tr = par.max_over_ranks(time.perf_counter() - t0, dev)
        L.fdgs_timing_enable(1)
        for i in range(4):
            train_iter(i)
        buf2 = ctypes.create_string_buffer(1 << 16)
        fdgs._lib.check(L.fdgs_timing_report(buf2, len(buf2), 1))
        L.fdgs_timing_enable(0)
        extra = {}
        for line in buf2.value.decode().strip().splitlines():
            name, cnt, tot = line.split()
            if name in ("plane_regulation", "adam_step", "densification_stats"):
                extra[name] = round(float(tot) / 4, 4)
        train = {"iterations_per_s": world * args.steps / dt_tr, "ms_per_iteration": dt_tr / args.steps * 1e3,
                 "includes": "render fwd+bwd, L1 stats, densification statistics, HexPlane regulariser fwd+bwd, FusedAdam step over all 8 parameter groups (lr = 0)",
                 "extra_kernels_ms_per_iteration": extra}
    # ---- secondary measurement: the views of one optimizer step behind one autograd node (fdgs.render_views; the reference's batch loop,
    # train.py:180-201, with batch_size = 2 as arguments/dynerf/cook_spinach.py:3 sets it): frames/s with the step's gradient arena shared
    batched = None
    if not args.no_train_step:
        B = 2
        dimgs = [torch.empty(3, H, W, device=dev) for _ in range(B)]

        def batch_step(i):
            for p_ in params:
                p_.grad = None
            vc = [cams[par.frames_for_rank(len(cams), B * i + v, rank, world)] for v in range(B)]
            res = fdgs.render_views(vc, pc, pipe, bg, stage="fine")
            acc.zero_()
            for v in range(B):
                img = res[v]["render"]
                fdgs._lib.check(L.fdgs_l1_stats(st(), img.numel(), ptr(img), ptr(target), 1.0 / (B * img.numel()), ptr(dimgs[v]), ptr(acc)))
            torch.autograd.backward([r_["render"] for r_ in res], dimgs)

        for i in range(max(args.warmup // 2, 2)):
            batch_step(i)
        par.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = max(args.steps // 2, 1)
        for i in range(nb):
            batch_step(i)
        par.barrier(); torch.cuda.synchronize()
        dt_b = par.max_over_ranks(time.perf_counter() - t0, dev)
        batched = {"views_per_step": B, "frames_per_s": world * B * nb / dt_b, "ms_per_step": dt_b / nb * 1e3, "ms_per_frame": dt_b / nb / B * 1e3,
                   "api": "fdgs.render_views (one autograd node and one gradient arena per optimizer step)"}
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cam_p = cams[args.warmup % len(cams)]
        cpu, ref = cpu_baseline(fdgs, syn, pc, cam_p, target, dcfg, args.cpu_frames)
        parity = parity_vs_oracle(fdgs, pc, cam_p, pipe, bg, params, ref)

    if rank == 0:
        out = {
            "metric": "train-step frames/sec (fwd+bwd raster+deform)", "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "src_sha16": lib_sha16(fdgs),
            "config": {"workload": args.workload, "gaussians": N, "image": [W, H], "deformation": dcfg,
                       "frames_per_step": world, "gaussian_order": args.order, "parallelism": f"frame-parallel x{world}", "num_rendered": R, "visible": V},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "train_iteration": train, "batched_step": batched,
            "ranks_seen": ranks_seen, "per_rank_ms_per_step": [round(x, 4) for x in per_rank_ms],
            "timed_regions_ms_per_step": [round(x / args.steps * 1e3, 4) for x in regions], "value_is": "median of the timed regions",
            "frame_hbm": {"algorithmic_bytes": B_frame, "achieved_GBps": B_frame / (dt / args.steps / 1) / 1e9 if world == 1 else None,
                          "frac_of_8TBps": (B_frame / (dt / args.steps)) / HBM_PEAK if world == 1 else None,
                          "note": "working set < 256 MiB Infinity Cache at this size: the HBM fraction is structurally small"},
            "mlp": {"fwd_bwd_flops": 3 * flops_fwd,
                    "achieved_TFLOPs_in_mfma_kernels": (3 * flops_fwd / (sum(kern[k]["ms_per_step"] for k in mfma_flops if k in kern) * 1e-3) / 1e12)
                    if kern else None, "peak_TFLOPs": MFMA_F32_PEAK / 1e12},
            "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms_per_step"])},
            "gpu_kernel_ms_per_step": round(sum(v["ms_per_step"] for v in kern.values()), 4),
            "loss": {"l1": float(l1), "psnr": float(psnr)},
        }
        print(json.dumps(out))


def lib_sha16(fdgs):
    """Hash of the kernel sources libfdgs.so is built from (the same value tools/pmc_traffic.py stamps on a counter artefact)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "4dgaussians_amd", "csrc", "*"))) + [os.path.join(ROOT, "include", "fdgs.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, workload, lib_sha):
    """HBM-side bytes per launch of `kernel` from a committed rocprofv3 PMC artefact (profiles/*_pmc_traffic*.json, written by
    tools/pmc_traffic.py from separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes over this same bench command).  An
    artefact is only used when it was collected for THIS workload with THIS build of libfdgs.so (`_workload`, `_src_sha16`
    keys: a hash of the kernel sources): anything else gives traffic = null plus the reason -- a counter from another workload or build is not a
    measurement of this run.  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KB; on
    gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads (what these kernels issue), so it is
    doubled; WRITE_SIZE is uncalibrated there and taken as reported."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic*.json")))
    reasons = []
    for f in reversed(files):
        d = json.load(open(f))
        if d.get("_workload") != workload:
            reasons.append(f"{os.path.basename(f)}: workload {d.get('_workload')!r}")
            continue
        if d.get("_src_sha16") != lib_sha:
            reasons.append(f"{os.path.basename(f)}: other build ({d.get('_src_sha16')})")
            continue
        k = d.get(kernel)
        if not k:
            return {"traffic": None, "traffic_source": os.path.basename(f) + " (kernel not profiled)"}
        fetch, write = k.get("FETCH_SIZE_KB_per_launch", 0.0) * 1024, k.get("WRITE_SIZE_KB_per_launch", 0.0) * 1024
        return {"traffic": 2 * fetch + write, "traffic_unit": "bytes/launch",
                "traffic_detail": {"FETCH_SIZE_bytes_raw": fetch, "FETCH_SIZE_bytes_corrected_x2": 2 * fetch, "WRITE_SIZE_bytes": write},
                "traffic_source": "profiles/" + os.path.basename(f)}
    return {"traffic": None, "traffic_source": None,
            "traffic_refused": ("no PMC artefact for this workload and build (kernel sources sha16 " + lib_sha + "); not used: " + "; ".join(reasons[:4]))
            if reasons else "no PMC artefact under profiles/"}


def cpu_baseline(fdgs, syn, pc, cam, target, dcfg, frames):
    """The same frame on the host cores: oracle deformation (torch CPU) -> oracle rasterizer (C, OpenMP), fwd + bwd.
    kind = "port": the rasterizer restatement is ours (the reference has no CPU rasterizer); the deformation oracle is
    pinned to the reference's modules (tests/test_oracle_deform.py).  Bounded sample: `frames` full frame(s) of the bench
    workload on min(cores, 64) threads (more threads only add scheduling overhead to these memory-bound loops)."""
    import numpy as np
    from oracle import deform_oracle as DO
    from oracle import raster_oracle as RO
    cores = os.cpu_count() or 1
    threads = min(cores, int(os.environ.get("FDGS_CPU_THREADS", "64")))
    torch.set_num_threads(threads)
    RO.set_threads(threads)
    sd = {k: v.detach().cpu().contiguous().clone().requires_grad_(v.dtype.is_floating_point and "poc" not in k and "aabb" not in k)
          for k, v in pc._deformation.state_dict().items()}
    leaves = {k: getattr(pc, k).detach().cpu().clone().requires_grad_(True)
              for k in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")}
    flags = syn.deform_args(dcfg)
    cam = cam.to("cpu")
    tgt = target.cpu().numpy()
    n = leaves["_xyz"].shape[0]
    stage = {"deform_fwd": 0.0, "raster_fwd": 0.0, "raster_bwd": 0.0, "deform_bwd": 0.0}
    ref = None
    t0 = time.perf_counter()
    for fi in range(frames):
        ta = time.perf_counter()
        shs = torch.cat([leaves["_features_dc"], leaves["_features_rest"]], 1)
        outs = DO.deform_forward(sd, flags, leaves["_xyz"], leaves["_scaling"], leaves["_rotation"], leaves["_opacity"], shs,
                                 torch.full((n, 1), cam.time), activate=True)
        tb = time.perf_counter()
        f = lambda x: np.ascontiguousarray(x.detach().numpy())
        o = RO.RasterOracle(means3D=f(outs[0]), scales=f(outs[1]), rotations=f(outs[2]), opacities=f(outs[3]), shs=f(outs[4]),
                            viewmatrix=f(cam.world_view_transform), projmatrix=f(cam.full_proj_transform), campos=f(cam.camera_center),
                            bg=np.zeros(3, np.float32), image_height=cam.image_height, image_width=cam.image_width,
                            tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), sh_degree=3)
        tc = time.perf_counter()
        dc = (np.sign(o.color - tgt) / o.color.size).astype(np.float32)
        g = o.backward(dc)
        td = time.perf_counter()
        gouts = [torch.tensor(g["means3D"]), torch.tensor(g["scales"]), torch.tensor(g["rotations"]),
                 torch.tensor(g["opacities"]).reshape(outs[3].shape), torch.tensor(g["shs"]).reshape(outs[4].shape)]
        wanted = list(leaves.values()) + [v for v in sd.values() if v.requires_grad]
        gref = torch.autograd.grad(list(outs), wanted, grad_outputs=gouts, allow_unused=True)
        te = time.perf_counter()
        if fi == 0:   # the oracle's image, depth and every gradient of this frame: the checker of parity_vs_oracle()
            names = list(leaves.keys()) + ["_deformation." + k for k, v in sd.items() if v.requires_grad]
            ref = dict(color=o.color.copy(), depth=o.depth.copy(), dc=dc, means2D=g["means2D"].copy(), radii=o.radii.copy(),
                       grads={k: (None if v is None else v.numpy()) for k, v in zip(names, gref)})
        o.close()
        for k_, v_ in zip(stage, (tb - ta, tc - tb, td - tc, te - td)):
            stage[k_] += v_
    dt = time.perf_counter() - t0
    return ({"value": frames / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{frames} full frame(s) of the same workload (fwd+bwd), {dt:.1f} s wall on {threads} threads of {cores} host cores; "
                      "deformation = oracle pinned to the reference modules (torch CPU), rasterizer = our C restatement (OpenMP)",
            "stage_seconds": {k_: round(v_, 3) for k_, v_ in stage.items()}}, ref)


def parity_vs_oracle(fdgs, pc, cam, pipe, bg, params, ref):
    """The headline checks itself: the frame the CPU leg just computed with the oracle chain (deformation oracle pinned to
    the reference's modules -> C rasterizer restatement, forward + analytic backward -> torch-CPU autograd) is rendered
    through the HIP path with the SAME upstream image gradient, and image, depth, radii and every parameter gradient are
    compared.  The oracle is the checker here, never the thing measured."""
    import numpy as np

    def rel(a, b):
        a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
        d = np.linalg.norm(b)
        return float(np.linalg.norm(a - b) / d) if d > 0 else float(np.linalg.norm(a))

    for p_ in params:
        p_.grad = None
    res = fdgs.render(cam, pc, pipe, bg, stage="fine")
    img = res["render"]
    img.backward(torch.tensor(ref["dc"], device=img.device))
    torch.cuda.synchronize()
    im, dp = img.detach().cpu().numpy(), res["depth"].detach().cpu().numpy()
    mse = float(((im.astype(np.float64) - ref["color"]) ** 2).mean())
    named = dict(pc.named_parameters())
    groups = {"xyz": ["_xyz"], "scaling": ["_scaling"], "rotation": ["_rotation"], "opacity": ["_opacity"], "f_dc": ["_features_dc"],
              "f_rest": ["_features_rest"],
              "planes": [k for k in ref["grads"] if "grids" in k and ref["grads"][k] is not None],
              "mlp": [k for k in ref["grads"] if k.startswith("_deformation.") and "grids" not in k and ref["grads"][k] is not None]}
    grad_rel = {}
    for gname, keys in groups.items():
        a = np.concatenate([named[k].grad.detach().cpu().numpy().ravel() for k in keys])
        b = np.concatenate([ref["grads"][k].ravel() for k in keys])
        grad_rel[gname] = rel(a, b)
    worst_tensor = max(((k, rel(named[k].grad.detach().cpu().numpy(), v)) for k, v in ref["grads"].items()
                        if v is not None and float(np.abs(v).max()) > 0), key=lambda kv: kv[1])
    radii = res["radii"].cpu().numpy()
    return {"frame": "the cpu_baseline frame (same camera, same upstream image gradient)",
            "image_psnr_dB": 10 * math.log10(1.0 / max(mse, 1e-20)), "image_mean_abs": float(np.abs(im - ref["color"]).mean()),
            "image_max_abs": float(np.abs(im - ref["color"]).max()), "depth_mean_abs": float(np.abs(dp - ref["depth"]).mean()),
            "radii_mismatch_frac": float((radii != ref["radii"]).mean()),
            "grad_rel_l2": {k: float(f"{v:.3e}") for k, v in grad_rel.items()},
            "viewspace_rel_l2": float(f"{rel(res['viewspace_points'].grad.cpu().numpy(), ref['means2D']):.3e}"),
            "worst_single_tensor": {"name": worst_tensor[0], "rel_l2": float(f"{worst_tensor[1]:.3e}")},
            "tolerance": {"image_psnr_dB": ">= 80", "grad_rel_l2": "<= 1e-3 (north_star)"}}


if __name__ == "__main__":
    main()
