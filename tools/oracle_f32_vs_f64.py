"""cfg5: is the f32 CPU oracle itself within 1e-3 of a float64 evaluation of the same deformation backward?"""
import importlib, math, sys, time
import numpy as np, torch
R = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
from scenes import rel_l2
from oracle import deform_oracle as DO
from oracle.raster_oracle import RasterOracle
fd = importlib.import_module("4dgaussians_amd")
syn = fd.synthetic
N, W, H, dcfg = (int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000), 2048, 2048, "dynerf_default"
pc = syn.SynthModel(N, dcfg, seed=6666)
fd.densify.spatial_reorder(pc, curve="hilbert")
cam = syn.orbit_cameras(W, H, n=160)[8]
n = N
def chain(dt, go=None):
    sd = {k: (v.detach().clone().to(dt) if v.dtype.is_floating_point else v.detach().clone()) for k, v in pc._deformation.state_dict().items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "poc" not in k and "aabb" not in k:
            v.requires_grad_(True)
    leaves = {k: getattr(pc, k).detach().clone().to(dt).requires_grad_(True) for k in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")}
    shs = torch.cat([leaves["_features_dc"], leaves["_features_rest"]], 1)
    t0 = time.time()
    m3, sc, rot, op, sh = DO.deform_forward(sd, pc._deformation.args, leaves["_xyz"], leaves["_scaling"], leaves["_rotation"], leaves["_opacity"], shs,
                                            torch.full((n, 1), cam.time, dtype=dt), activate=True)
    print(dt, "deform fwd", time.time() - t0, flush=True)
    if go is None:
        f = lambda x: np.ascontiguousarray(x.detach().float().numpy())
        o = RasterOracle(means3D=f(m3), scales=f(sc), rotations=f(rot), opacities=f(op), shs=f(sh), viewmatrix=f(cam.world_view_transform),
                         projmatrix=f(cam.full_proj_transform), campos=f(cam.camera_center), bg=np.zeros(3, np.float32), image_height=H,
                         image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), sh_degree=3)
        rng = np.random.default_rng(3)
        target = rng.random(o.color.shape).astype(np.float32)
        dc = (np.sign(o.color - target) / o.color.size).astype(np.float32)
        go = o.backward(dc, None)
        print("raster done", time.time() - t0, flush=True)
    gouts = [torch.tensor(go["means3D"]).to(dt), torch.tensor(go["scales"]).to(dt), torch.tensor(go["rotations"]).to(dt),
             torch.tensor(go["opacities"]).reshape(op.shape).to(dt), torch.tensor(go["shs"]).reshape(sh.shape).to(dt)]
    wanted = list(leaves.values()) + [v for v in sd.values() if v.requires_grad]
    wnames = list(leaves.keys()) + ["_deformation." + k for k, v in sd.items() if v.requires_grad]
    g = torch.autograd.grad([m3, sc, rot, op, sh], wanted, grad_outputs=gouts, allow_unused=True)
    print(dt, "bwd done", time.time() - t0, flush=True)
    return go, {k: (None if x is None else x.double().numpy()) for k, x in zip(wnames, g)}
go, g32 = chain(torch.float32)
np.savez("/tmp/go.npz", **{k: v for k, v in go.items()})
_, g64 = chain(torch.float64, go)
rows = []
for k in g32:
    if g32[k] is None or np.abs(g64[k]).max() == 0: continue
    rows.append((rel_l2(g32[k], g64[k]), k))
rows.sort(reverse=True)
for r, k in rows[:12]: print(f"{k:60s} f32 vs f64 rel-L2 {r:.2e}")
grp = lambda pred: rel_l2(np.concatenate([g32[k].ravel() for k in g32 if g32[k] is not None and pred(k)]), np.concatenate([g64[k].ravel() for k in g32 if g32[k] is not None and pred(k)]))
print("planes", grp(lambda k: "grids" in k), "mlp", grp(lambda k: k.startswith("_deformation.") and "grids" not in k), "xyz", grp(lambda k: k == "_xyz"))
np.savez("/tmp/g64.npz", **{k: v for k, v in g64.items() if v is not None})
