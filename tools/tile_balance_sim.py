"""tools/tile_balance_sim.py [cube|shell] -- how well does the blending backward's "one wave per tile" fill the chip?  (CPU, ~20 s)

Per-tile work of the bench frame (BASELINE config 4: 300 k Gaussians, 1352 x 1014) from the C rasterizer oracle -- entries walked by the
backward = max n_contrib over the tile's pixels; list length for the forward -- fed to a dispatch model of csrc/render.hip: block b runs on
XCD b % 8 and maps to a tile through unit_of_block; an XCD hands its blocks out IN ORDER to the SIMD (backward: 128 per XCD, 4 wave slots
each at 110 VGPRs) or CU (forward: 32 per XCD, 8 workgroups each) with the fewest resident units; resident units of a SIMD share its issue
slots equally.  Prints makespan vs perfect balance for the image order and for heaviest-first orders.  What it said in round 5
(profiles/r05_tile_balance_sim.txt): backward 0.62 (cube) / 0.65 (shell) of perfect balance in image order, 0.82 / 0.84 heaviest-first per
XCD by the exact walk length -- and 0.58 on the cube when ordered by LIST length (the cube's lists are 6x longer than what its saturating
pixels walk: correlation 0.22), which is why the forward leaves the walk length of every tile behind for the backward's order."""
import importlib
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import deform_oracle as DO  # noqa: E402
from oracle.raster_oracle import RasterOracle  # noqa: E402


def unit_order(gx, gy, bh=2):
    """tiles in dispatch order per XCD (csrc/render.hip: unit_of_block)"""
    groups = (gy + bh - 1) // bh
    nb = 8 * ((groups + 7) // 8) * bh * gx
    per_xcd = [[] for _ in range(8)]
    for b in range(nb):
        xcd, idx = b & 7, b >> 3
        g, rem = divmod(idx, bh * gx)
        row = (g * 8 + xcd) * bh + rem // gx
        if row < gy:
            per_xcd[xcd].append((row, rem % gx))
    return per_xcd


def simulate(seqs, work, slots, procs, overhead=8.0):
    """per XCD: in-order dispatch to the processor with the fewest resident units; processor sharing inside one.  (makespan, ideal)"""
    worst = total = 0.0
    for seq in seqs:
        jobs = [work[r, c] + overhead for (r, c) in seq]
        total += sum(jobs)
        res = [[] for _ in range(procs)]
        t, qi = 0.0, 0

        def fill():
            nonlocal qi
            while qi < len(jobs):
                s = min(range(procs), key=lambda i: len(res[i]))
                if len(res[s]) >= slots:
                    break
                res[s].append(jobs[qi])
                qi += 1
        fill()
        while any(res):
            dt = min(min(r) * len(r) for r in res if r)
            t += dt
            for r in res:
                if r:
                    k = len(r)
                    r[:] = [x - dt / k for x in r if x - dt / k > 1e-9]
            fill()
        worst = max(worst, t)
    return worst, total / (8 * procs)


def main():
    fd = importlib.import_module("4dgaussians_amd")
    syn = fd.synthetic
    for scene in (sys.argv[1:] or ["cube", "shell"]):
        N, W, H = 300000, 1352, 1014
        pc = syn.SynthModel(N, "dynerf_default", seed=6666, scene=scene)
        cam = syn.orbit_cameras(W, H, n=160)[8]
        with torch.no_grad():
            shs = torch.cat([pc._features_dc, pc._features_rest], 1)
            m3, sc, rot, op, sh = DO.deform_forward(pc._deformation.state_dict(), pc._deformation.args, pc._xyz, pc._scaling, pc._rotation, pc._opacity,
                                                    shs, torch.full((N, 1), cam.time), activate=True)
        f = lambda x: np.ascontiguousarray(x.detach().numpy())
        o = RasterOracle(means3D=f(m3), scales=f(sc), rotations=f(rot), opacities=f(op), shs=f(sh), viewmatrix=f(cam.world_view_transform),
                         projmatrix=f(cam.full_proj_transform), campos=f(cam.camera_center), bg=np.zeros(3, np.float32), image_height=H, image_width=W,
                         tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), sh_degree=3)
        _, nc = o.image_state()
        gy, gx = (H + 15) // 16, (W + 15) // 16
        pad = np.zeros((gy * 16, gx * 16), np.int64)
        pad[:H, :W] = nc
        todo = pad.reshape(gy, 16, gx, 16).max(axis=(1, 3)).astype(np.float64)
        keys, _ = o.pairs()
        length = np.bincount(keys.astype(np.int64), minlength=gx * gy).reshape(gy, gx).astype(np.float64)
        print(f"[{scene}] {gx * gy} tiles; backward walk length per tile: mean {todo.mean():.0f} p50 {np.median(todo):.0f} p90 {np.percentile(todo, 90):.0f} "
              f"max {todo.max():.0f}; list length (un-culled oracle lists) mean {length.mean():.0f}; correlation {np.corrcoef(length.ravel(), todo.ravel())[0, 1]:.2f}")
        seqs = unit_order(gx, gy)
        for name, key in (("image order", None), ("heaviest-first per XCD, by walk length", todo), ("heaviest-first per XCD, by list length", length)):
            s2 = seqs if key is None else [sorted(s, key=lambda rc: -key[rc]) for s in seqs]
            mk, ideal = simulate(s2, todo, slots=4, procs=128)
            print(f"   backward (one wave per tile, 128 SIMDs x 4 slots per XCD), {name:42s}: {ideal / mk:.2f} of perfect balance")
        fw = np.minimum(np.ceil(np.maximum(todo, 1) / 256) * 256, np.maximum(length, 1))
        for name, key in (("image order", None), ("heaviest-first per XCD, by list length", length)):
            s2 = seqs if key is None else [sorted(s, key=lambda rc: -key[rc]) for s in seqs]
            mk, ideal = simulate(s2, fw, slots=8, procs=32)
            print(f"   forward (one 256-thread workgroup per tile, 32 CUs x 8 per XCD), {name:38s}: {ideal / mk:.2f} of perfect balance")


if __name__ == "__main__":
    main()
