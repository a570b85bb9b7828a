"""Development diagnostic (GPU box): the 2 M / 2048^2 frame stage by stage against the oracles -- HIP deformation outputs vs the deformation
oracle, then the HIP rasterizer on the ORACLE's deformed Gaussians vs the C rasterizer oracle; where do the differing pixels sit?"""
import importlib, math, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import deform_oracle as DO
from oracle.raster_oracle import RasterOracle
fd = importlib.import_module("4dgaussians_amd"); syn = fd.synthetic
dev = torch.device("cuda:0")
N, W, H = int(os.environ.get("N", 2_000_000)), int(os.environ.get("W", 2048)), int(os.environ.get("H", 2048))
pc = syn.SynthModel(N, "dynerf_default", seed=6666)
fd.densify.spatial_reorder(pc, curve="hilbert")
cam = syn.orbit_cameras(W, H, n=160)[8]
with torch.no_grad():
    shs = torch.cat([pc._features_dc, pc._features_rest], 1)
    ref = DO.deform_forward(pc._deformation.state_dict(), pc._deformation.args, pc._xyz, pc._scaling, pc._rotation, pc._opacity, shs,
                            torch.full((N, 1), cam.time), activate=True)
f = lambda x: np.ascontiguousarray(x.detach().cpu().numpy())
o = RasterOracle(means3D=f(ref[0]), scales=f(ref[1]), rotations=f(ref[2]), opacities=f(ref[3]), shs=f(ref[4]), viewmatrix=f(cam.world_view_transform),
                 projmatrix=f(cam.full_proj_transform), campos=f(cam.camera_center), bg=np.zeros(3, np.float32), image_height=H, image_width=W,
                 tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), sh_degree=3)
pcg = pc.to(dev)
camg = cam.to(dev)
bg = torch.zeros(3, device=dev)
for grad in (False, True):
    ctx = torch.enable_grad() if grad else torch.no_grad()
    with ctx:
        out = fd.deformation.deform(pcg._deformation, pcg._xyz, pcg._scaling, pcg._rotation, pcg._opacity, shs_dc=pcg._features_dc, shs_rest=pcg._features_rest,
                                    time=cam.time, activate=True)
    torch.cuda.synchronize()
    for name, a, b in zip(("xyz", "scales", "rot", "opacity", "shs"), out, ref):
        d = (a.detach().cpu().reshape(N, -1) - b.reshape(N, -1)).abs().max(1).values
        print(f"deform (grad mode {grad}) {name:8s} max abs diff {float(d.max()):.3e}  rows over 1e-4: {int((d > 1e-4).sum())}  first: {torch.nonzero(d > 1e-4).flatten()[:8].tolist()}")
gx = (W + 15) // 16
def report(tag, im):
    d = np.abs(im - o.color).max(0)
    bad = d > 1e-4
    ys, xs = np.nonzero(bad)
    t = (ys // 16) * gx + xs // 16
    cnt = np.bincount(t, minlength=1)
    top = np.argsort(-cnt)[:8]
    print(f"{tag}: psnr {10 * math.log10(1.0 / max(float(((im.astype(np.float64) - o.color) ** 2).mean()), 1e-20)):.1f} dB  max {d.max():.3e}  pixels over 1e-4: {int(bad.sum())}  "
          f"tiles with bad pixels: {int((cnt > 0).sum())}  heaviest (tile: pixels): {[(int(i), int(cnt[i])) for i in top if cnt[i] > 0]}")
rs = fd.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0, camg.world_view_transform, camg.full_proj_transform, 3, camg.camera_center, False, False)
g = [x.to(dev) for x in ref]
color, radii, depth, st = fd.rasterizer.rasterize_forward(rs, g[0], g[4], None, g[3], g[1], g[2], None)
torch.cuda.synchronize()
report("HIP rasterizer on the ORACLE's deformed Gaussians", color.cpu().numpy())
res = fd.render(camg, pcg, syn.PipelineParams(), bg, stage="fine")
torch.cuda.synchronize()
report("fd.render (grad mode on)", res["render"].detach().cpu().numpy())
with torch.no_grad():
    res2 = fd.render(camg, pcg, syn.PipelineParams(), bg, stage="fine")
report("fd.render (no grad)", res2["render"].detach().cpu().numpy())
rad = res2["radii"].cpu().numpy()
mism = np.nonzero(rad != o.radii)[0]
xy, dep = o.field("xy"), o.field("depth")
print("radii mismatches:", len(mism))
for i in mism[:12]:
    print(f"   Gaussian {i}: oracle radius {o.radii[i]} HIP radius {rad[i]}  oracle depth {dep[i]:.6f}  screen xy ({xy[i,0]:.1f}, {xy[i,1]:.1f}) -> tile {int(xy[i,1]) // 16 * gx + int(xy[i,0]) // 16}  "
          f"HIP view depth {float((res2['depth'] * 0).sum()):.0f}")
# view-space depth of these Gaussians from both sets of positions
V = cam.world_view_transform.double()
for name, P in (("oracle xyz", ref[0]), ("HIP xyz", out[0].detach().cpu())):
    p = torch.cat([P[mism].double(), torch.ones(len(mism), 1, dtype=torch.float64)], 1) @ V
    print("   view z from", name, [float(f"{z:.7f}") for z in p[:, 2].tolist()][:12])
print("---- which deformed tensor matters? (HIP rasterizer; h = HIP deformation output, o = oracle's)")
hip = [x.detach() for x in out]
names = ("xyz", "scales", "rot", "opacity", "shs")
def raster(ts, tag):
    c, r, d_, s_ = fd.rasterizer.rasterize_forward(rs, ts[0], ts[4], None, ts[3], ts[1], ts[2], None)
    torch.cuda.synchronize()
    report(tag, c.cpu().numpy())
raster(hip, "all HIP")
for k in range(5):
    ts = list(g); ts[k] = hip[k]
    raster(ts, f"oracle's, except {names[k]} from HIP")
# rows of the HIP tensors that differ most, and Gaussians projected into the bad tiles
bad_tile = 9150
ty, tx = divmod(bad_tile, gx)
inside = (np.abs(xy[:, 0] - (tx * 16 + 24)) < 60) & (np.abs(xy[:, 1] - (ty * 16 + 16)) < 50) & (o.radii > 0)
idx = np.nonzero(inside)[0]
print("Gaussians near the bad tiles:", len(idx))
big = idx[np.argsort(-o.radii[idx])[:6]]
for i in big:
    print(f"   Gaussian {i}: radius {o.radii[i]} depth {dep[i]:.4f} xy ({xy[i,0]:.1f},{xy[i,1]:.1f}) opacity oracle {float(ref[3][i]):.6f} HIP {float(hip[3][i]):.6f} "
          f"scale oracle {ref[1][i].tolist()} HIP {hip[1][i].cpu().tolist()}")
