"""Development diagnostic (GPU): are all tiles of the 2 M / 2048^2 frame rendered?  Output buffers are pre-filled with NaN; any NaN left = a tile
no workgroup took.  Compared with the tile order on / off."""
import importlib, math, sys, os, torch
sys.path.insert(0, os.getcwd())
fd = importlib.import_module("4dgaussians_amd"); syn = fd.synthetic
dev = torch.device("cuda:0")
N, W, H = int(os.environ.get("N", 2_000_000)), int(os.environ.get("W", 2048)), int(os.environ.get("H", 2048))
pc = syn.SynthModel(N, "dynerf_default", seed=6666)
fd.densify.spatial_reorder(pc, curve="hilbert")
pc = pc.to(dev)
cam = syn.orbit_cameras(W, H, n=160)[8].to(dev)
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    out = fd.deformation.deform(pc._deformation, pc._xyz, pc._scaling, pc._rotation, pc._opacity, shs_dc=pc._features_dc, shs_rest=pc._features_rest, time=cam.time, activate=True)
rs = fd.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
imgs = {}
for order in (0, 1, 1, 0):
    fd._lib.tuning_set("tile_order", order)
    color = torch.full((3, H, W), float("nan"), device=dev); depth = torch.full((1, H, W), float("nan"), device=dev); radii = torch.zeros(N, dtype=torch.int32, device=dev)
    _, _, _, st = fd.rasterizer.rasterize_forward(rs, out[0], out[4], None, out[3], out[1], out[2], None, out=(color, radii, depth))
    torch.cuda.synchronize()
    bad = torch.isnan(color).any(0)
    ys, xs = torch.nonzero(bad, as_tuple=True)
    gx = (W + 15) // 16
    tiles = sorted(set(((ys // 16) * gx + xs // 16).tolist()))
    print(f"tile_order={order}: pairs {st.num_rendered}, NaN pixels {int(bad.sum())}, tiles not rendered: {len(tiles)} {tiles[:16]}")
    if order in imgs:
        print("   same order again: max diff", float((torch.nan_to_num(color) - torch.nan_to_num(imgs[order])).abs().max()))
    imgs[order] = color
print("order1 vs order0 max diff (NaN -> 0):", float((torch.nan_to_num(imgs[1]) - torch.nan_to_num(imgs[0])).abs().max()))
