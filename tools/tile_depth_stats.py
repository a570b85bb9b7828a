"""Distribution of the blending kernels' per-tile walk length (max n_contrib over the tile's pixels) on the bench scene, and its layout over
the image: what bounds the critical path of a one-wave-per-tile kernel.  usage (GPU box): python tools/tile_depth_stats.py [workload]"""
import ctypes, importlib, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
fdgs = importlib.import_module("4dgaussians_amd")
syn = fdgs.synthetic
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4_dynerf_300k_1352x1014"
N, W, H, dcfg = bench.WORKLOADS[wl]
pc = syn.SynthModel(N, dcfg, seed=6666, device=dev)
fdgs.densify.spatial_reorder(pc)
cams = [c.to(dev) for c in syn.orbit_cameras(W, H, n=160)]
L = fdgs._lib.lib()
ptr = fdgs._lib.ptr
for ci in (8, 40, 100):
    cam = cams[ci]
    with torch.no_grad():
        out = fdgs.deformation.deform(pc._deformation, pc._xyz, pc._scaling, pc._rotation, pc._opacity, shs_dc=pc._features_dc,
                                      shs_rest=pc._features_rest, time=cam.time, activate=True)
        rs = fdgs.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev), 1.0,
                                                cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
        _, radii, _, state = fdgs.rasterizer.rasterize_forward(rs, out[0], out[4], None, out[3], out[1], out[2], None)
        fp = ctypes.c_void_p()
        fdgs._lib.check(L.fdgs_img_field(ptr(state.img), W, H, 1, ctypes.byref(fp)))
        off = fp.value - state.img.data_ptr()
        nc = state.img[off:off + W * H * 4].view(torch.int32).view(H, W)
        fdgs._lib.check(L.fdgs_img_field(ptr(state.img), W, H, 2, ctypes.byref(fp)))
        off = fp.value - state.img.data_ptr()
        gy, gx = (H + 15) // 16, (W + 15) // 16
        rng = state.img[off:off + gx * gy * 8].view(torch.int32).view(gy, gx, 2)
        length = (rng[..., 1] - rng[..., 0]).float()
        pad = torch.zeros(gy * 16, gx * 16, dtype=torch.int32, device=dev)
        pad[:H, :W] = nc
        tmax = pad.view(gy, 16, gx, 16).amax(dim=(1, 3)).float()
        q = lambda t, p: float(torch.quantile(t.flatten(), p))
        print(f"cam {ci}: pairs {int(state.num_rendered)}  walk(max n_contrib per tile): mean {float(tmax.mean()):.1f} p50 {q(tmax, .5):.0f} p90 {q(tmax, .9):.0f} "
              f"p99 {q(tmax, .99):.0f} max {float(tmax.max()):.0f} | list length: mean {float(length.mean()):.1f} p99 {q(length, .99):.0f} max {float(length.max()):.0f}")
        rows = tmax.sum(dim=1)
        print("   per tile-row sums (top to bottom, /1000): " + " ".join(f"{float(v) / 1000:.1f}" for v in rows))
