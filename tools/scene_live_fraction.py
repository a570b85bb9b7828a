#!/usr/bin/env python
"""tools/scene_live_fraction.py -- CPU (oracle chain): what fraction of a synthetic scene's visible Gaussians receives a non-zero gradient row
from the rasterizer, i.e. how much of the deformation backward a frame of that scene can skip.  Used to choose the parameters of
synthetic.make_gaussians(scene="shell") (development measurement; the oracle is the tool here, nothing is timed).

    python tools/scene_live_fraction.py [--scene shell] [--n 300000] [--size 1352x1014] [--cams 8,60]
"""
import argparse, importlib, json, math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import deform_oracle as DO
from oracle.raster_oracle import RasterOracle

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="shell"); ap.add_argument("--n", type=int, default=300000); ap.add_argument("--size", default="1352x1014")
ap.add_argument("--cams", default="8"); ap.add_argument("--cfg", default="dynerf_default")
ap.add_argument("--radii", default=None); ap.add_argument("--opacity", default=None); ap.add_argument("--scale", type=float, default=None)
a = ap.parse_args()
fd = importlib.import_module("4dgaussians_amd"); syn = fd.synthetic
if a.radii: syn.SHELL_RADII = tuple(float(x) for x in a.radii.split(","))
if a.opacity: syn.SHELL_OPACITY = tuple(float(x) for x in a.opacity.split(","))
if a.scale: syn.SHELL_SCALE = a.scale
W, H = (int(x) for x in a.size.split("x"))
pc = syn.SynthModel(a.n, a.cfg, seed=6666, scene=a.scene)
fd.densify.spatial_reorder(pc, curve="hilbert")
for ci in [int(c) for c in a.cams.split(",")]:
    cam = syn.orbit_cameras(W, H, n=160)[ci]
    with torch.no_grad():
        outs = DO.deform_forward(pc._deformation.state_dict(), pc._deformation.args, pc._xyz, pc._scaling, pc._rotation, pc._opacity,
                                 torch.cat([pc._features_dc, pc._features_rest], 1), torch.full((a.n, 1), cam.time), activate=True)
    f = lambda x: np.ascontiguousarray(x.detach().numpy())
    o = RasterOracle(means3D=f(outs[0]), scales=f(outs[1]), rotations=f(outs[2]), opacities=f(outs[3]), shs=f(outs[4]), viewmatrix=f(cam.world_view_transform),
                     projmatrix=f(cam.full_proj_transform), campos=f(cam.camera_center), bg=np.zeros(3, np.float32), image_height=H, image_width=W,
                     tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), sh_degree=3)
    tgt = np.random.default_rng(0).random(o.color.shape).astype(np.float32)
    g = o.backward((np.sign(o.color - tgt) / o.color.size).astype(np.float32))
    nz = np.zeros(a.n, bool)
    for k in ("means3D", "scales", "rotations", "opacities", "shs"):
        nz |= (g[k].reshape(a.n, -1) != 0).any(1)
    vis = o.radii > 0
    npad = (a.n + 31) // 32 * 32
    z = np.zeros(npad, bool); z[:a.n] = nz
    fT, nc = o.image_state()
    print(json.dumps({"scene": a.scene, "cam": ci, "N": a.n, "visible": int(vis.sum()), "num_rendered": int(o.num_rendered), "nonzero_rows": int(nz.sum()),
                      "nonzero_of_visible": round(float(nz.sum() / max(vis.sum(), 1)), 4), "live_tiles32_frac": round(float(z.reshape(-1, 32).any(1).mean()), 4),
                      "mean_n_contrib": round(float(nc.mean()), 1), "mean_final_T": round(float(fT.mean()), 4), "image_mean": round(float(o.color.mean()), 4)}), flush=True)
    o.close()
