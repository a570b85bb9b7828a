"""Every branch of the reference's render() (gaussian_renderer/__init__.py:18-138) executed through the HIP path.

The reference holds two formulations of two pieces of rasterizer arithmetic in-tree: the python SH evaluation
(`pipe.convert_SHs_python`, :106-111) and the python 3-D covariance (`pipe.compute_cov3D_python`, :74).  Rendering with the
python formulation (colours / covariances precomputed with torch ops, handed to the rasterizer as *_precomp) must equal the
native path (SH / covariance evaluated inside the HIP preprocess kernel) -- image to 1e-6, gradients to 1e-4: this ties the
kernel's arithmetic to the reference's own python, the only rasterizer maths the reference has in-tree.
Also: override_color (:113), cam_type == "PanopticSports" (:53-55), a foreign deformation module (non-fused fine stage),
and the HIP preprocess kernel's cov3D / RGB against tests/golden/raster_pins.npz (reference python, see its generator)."""
import importlib
import math
import os

import numpy as np
import pytest

from conftest import set_knob
import torch

from scenes import rel_l2

pytestmark = pytest.mark.gpu
synthetic = importlib.import_module("4dgaussians_amd.synthetic")


def _fd():
    return importlib.import_module("4dgaussians_amd")


def _model(n=6000, cfg="dynerf_default", seed=17, boost=1.0):
    pc = synthetic.SynthModel(n, cfg, seed=seed)
    with torch.no_grad():
        pc._scaling.add_(boost)
    return pc.to(torch.device("cuda:0"))


def _render_and_grads(pc, cam, pipe, stage, w, **kw):
    fd = _fd()
    dev = torch.device("cuda:0")
    for p in pc.parameters():
        p.grad = None
    res = fd.render(cam, pc, pipe, torch.tensor([0.1, 0.2, 0.3], device=dev), stage=stage, **kw)
    (res["render"] * w).sum().backward()
    grads = {k: p.grad.clone() for k, p in pc.named_parameters() if p.grad is not None}
    return res, grads, res["viewspace_points"].grad.clone()


class _Pipe:
    def __init__(self, sh=False, cov=False):
        self.convert_SHs_python, self.compute_cov3D_python, self.debug = sh, cov, False


@pytest.mark.parametrize("sh,cov", [(True, False), (False, True), (True, True)])
def test_python_sh_and_python_cov3d_equal_native_path_coarse(sh, cov):
    dev = torch.device("cuda:0")
    pc = _model()
    cam = synthetic.make_camera(240, 180, theta_deg=55.0, time=0.4).to(dev)
    w = torch.randn(3, 180, 240, generator=torch.Generator().manual_seed(2)).to(dev)
    a, ga, va = _render_and_grads(pc, cam, _Pipe(), "coarse", w)
    b, gb, vb = _render_and_grads(pc, cam, _Pipe(sh, cov), "coarse", w)
    assert (a["radii"] > 0).sum() > 1000
    mism = float((a["radii"] != b["radii"]).float().mean())     # radius = ceil(3 sigma): a last-bit cov3D difference may move it by 1
    d = (a["render"] - b["render"]).abs()
    print(f"[sh={sh} cov={cov}] radii mismatch {mism:.2e}, image max diff {float(d.max()):.2e} mean {float(d.mean()):.2e}")
    assert mism < 1e-3
    # a last-bit difference in cov3D can move one splat across the alpha >= 1/255 cut at a pixel (a step of <= 1/255 there);
    # everywhere else the two formulations agree to float rounding
    assert float(d.max()) < 1.0 / 255 + 1e-5 and float(d.mean()) < 1e-6 and float(torch.quantile(d.flatten()[::3], 0.9999)) < 2e-5
    dd = (a["depth"] - b["depth"]).abs()
    assert float(dd.mean()) < 1e-6 and float(torch.quantile(dd.flatten(), 0.9999)) < 1e-4
    errs = {k: rel_l2(gb[k].cpu().numpy(), ga[k].cpu().numpy()) for k in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")}
    print("   grads: " + ", ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    for k, e in errs.items():
        assert e < 1e-4, (k, e)
    assert rel_l2(vb.cpu().numpy(), va.cpu().numpy()) < 1e-4


def test_python_sh_fine_stage_runs_and_uses_canonical_features():
    """In the fine stage the reference's python-SH branch evaluates the CANONICAL features at the CANONICAL positions
    (pc.get_features / pc.get_xyz, :107-110) -- not the deformed ones.  Same here: image = native render with the SH head's
    and position head's effect on the colour removed, i.e. equal to a render with override_color = those colours."""
    fd = _fd()
    dev = torch.device("cuda:0")
    pc = _model(4000)
    cam = synthetic.make_camera(200, 152, theta_deg=10.0, time=0.7).to(dev)
    bg = torch.zeros(3, device=dev)
    a = fd.render(cam, pc, _Pipe(sh=True), bg, stage="fine")
    feats = pc.get_features
    dirs = pc.get_xyz - cam.camera_center.repeat(feats.shape[0], 1)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    cols = torch.clamp_min(fd.sh.eval_sh(3, feats.transpose(1, 2).view(-1, 3, 16), dirs) + 0.5, 0.0)
    b = fd.render(cam, pc, _Pipe(), bg, stage="fine", override_color=cols)
    assert torch.equal(a["render"], b["render"])
    a["render"].sum().backward()
    assert pc._features_rest.grad is not None and float(pc._features_rest.grad.abs().max()) > 0
    assert float(pc._deformation.deformation_net.pos_deform[3].weight.grad.abs().max()) > 0


def test_override_color_matches_direct_rasterizer_call():
    fd = _fd()
    dev = torch.device("cuda:0")
    pc = _model(3000)
    cam = synthetic.make_camera(160, 120, theta_deg=-20.0).to(dev)
    bg = torch.tensor([1.0, 1.0, 1.0], device=dev)
    cols = torch.rand(3000, 3, generator=torch.Generator().manual_seed(0)).to(dev).requires_grad_(True)
    res = fd.render(cam, pc, _Pipe(), bg, stage="coarse", override_color=cols)
    rs = fd.GaussianRasterizationSettings(120, 160, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0, cam.world_view_transform,
                                          cam.full_proj_transform, 3, cam.camera_center, False, False)
    img, radii, depth = fd.GaussianRasterizer(rs)(means3D=pc._xyz, means2D=torch.zeros_like(pc._xyz), opacities=torch.sigmoid(pc._opacity),
                                                  colors_precomp=cols, scales=torch.exp(pc._scaling),
                                                  rotations=torch.nn.functional.normalize(pc._rotation))
    assert torch.equal(res["render"], img) and torch.equal(res["radii"], radii) and torch.equal(res["depth"], depth)
    res["render"].sum().backward()
    assert cols.grad is not None and float(cols.grad.abs().sum()) > 0 and pc._features_dc.grad is None


def test_panoptic_sports_dict_camera_equals_camera_object():
    fd = _fd()
    dev = torch.device("cuda:0")
    pc = _model(3000)
    cam = synthetic.make_camera(160, 120, theta_deg=75.0, time=0.25).to(dev)
    bg = torch.zeros(3, device=dev)
    a = fd.render(cam, pc, _Pipe(), bg, stage="fine")
    rs = fd.GaussianRasterizationSettings(120, 160, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0, cam.world_view_transform,
                                          cam.full_proj_transform, pc.active_sh_degree, cam.camera_center, False, False)
    b = fd.render({"camera": rs, "time": 0.25}, pc, _Pipe(), bg, stage="fine", cam_type="PanopticSports")
    for k in ("render", "depth", "radii"):
        assert torch.equal(a[k], b[k]), k


class _Foreign(torch.nn.Module):
    """A deformation module render() does not know (stands for the reference's own scene.deformation.deform_network):
    called as the reference calls it -- (means3D, scales, rotations, opacity, shs, time [N,1]) -> raw outputs."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, point, scales=None, rotations=None, opacity=None, shs=None, times_sel=None):
        assert times_sel.shape == (point.shape[0], 1)
        return self.inner(point, scales, rotations, opacity, shs, times_sel)


def test_foreign_deformation_module_non_fused_fine_stage():
    dev = torch.device("cuda:0")
    pc = _model(5000)
    cam = synthetic.make_camera(200, 152, theta_deg=140.0, time=0.55).to(dev)
    w = torch.randn(3, 152, 200, generator=torch.Generator().manual_seed(3)).to(dev)
    a, ga, va = _render_and_grads(pc, cam, _Pipe(), "fine", w)
    inner = pc._deformation
    pc._deformation = _Foreign(inner)                          # not an instance of fdgs.deform_network -> non-fused branch
    b, gb, vb = _render_and_grads(pc, cam, _Pipe(), "fine", w)
    assert torch.equal(a["radii"], b["radii"])
    # the two branches round the activations differently (fused exp / normalize / sigmoid in the kernel vs torch ops): the images agree to
    # rounding except where that moves an alpha across the 1/255 threshold of the blend -- single pixels, bounded by one blended contribution
    dimg = (a["render"] - b["render"]).abs()
    # an entry whose alpha crosses 1/255 contributes alpha * T * colour <= colour / 255 to a pixel channel: the bound on a single outlier is the
    # largest SH colour of a visible Gaussian (no upper clamp, utils/sh_utils.py:113) over 255 -- not a free constant -- and an outlier must be
    # isolated (one decision, one Gaussian footprint: a handful of pixels), which a real blending error would not be
    import ctypes
    fd = _fd()
    with torch.no_grad():
        out = fd.deformation.deform(inner, pc._xyz, pc._scaling, pc._rotation, pc._opacity, shs_dc=pc._features_dc, shs_rest=pc._features_rest,
                                    time=0.55, activate=True)
        rs = fd.GaussianRasterizationSettings(152, 200, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev), 1.0,
                                              cam.world_view_transform, cam.full_proj_transform, pc.active_sh_degree, cam.camera_center, False, False)
        _, radii_, _, st_ = fd.rasterizer.rasterize_forward(rs, out[0], out[4], None, out[3], out[1], out[2], None)
        pp = ctypes.c_void_p()
        assert fd._lib.lib().fdgs_geom_field(ctypes.c_void_p(st_.geom.data_ptr()), 5000, 3, ctypes.byref(pp)) == 0
        rgb = torch.empty(5000 * 4, device=dev)
        assert ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(rgb.data_ptr()), pp, ctypes.c_size_t(5000 * 16), ctypes.c_int(3)) == 0
        cmax = max(1.0, float(rgb.reshape(5000, 4)[radii_ > 0, :3].max()))
    n_out = int((dimg > 1e-5).sum())
    assert float(dimg.mean()) < 1e-6 and n_out <= 12 and float(dimg.max()) <= 1.05 * cmax / 255.0, (float(dimg.mean()), n_out, float(dimg.max()), cmax)
    for k, v in ga.items():
        k2 = k.replace("_deformation.", "_deformation.inner.")
        assert rel_l2(gb[k2].cpu().numpy(), v.cpu().numpy()) < 1e-4, k
    assert rel_l2(vb.cpu().numpy(), va.cpu().numpy()) < 1e-4


def test_hip_preprocess_cov3d_and_sh_colour_match_reference_python():
    """geom fields of fdgs_preprocess_fwd against the fixture computed by the REFERENCE's python functions."""
    import ctypes
    fd = _fd()
    dev = torch.device("cuda:0")
    Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "raster_pins.npz"))
    g = synthetic.make_gaussians(512, seed=31)
    cam = synthetic.make_camera(400, 400, theta_deg=30.0).to(dev)
    q = torch.nn.functional.normalize(g["rotation"]).to(dev)
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1).to(dev)
    hip = ctypes.CDLL("libamdhip64.so")
    L = fd._lib.lib()
    for mod in (1.0, 0.7):
        for deg in (3, 1):
            rs = fd.GaussianRasterizationSettings(400, 400, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev), mod,
                                                  cam.world_view_transform, cam.full_proj_transform, deg, cam.camera_center, False, False)
            _, radii, _, st = fd.rasterizer.rasterize_forward(rs, g["xyz"].to(dev), shs, None, torch.sigmoid(g["opacity"]).to(dev),
                                                              torch.tensor(Z["cov.scales"], device=dev), q, None)
            torch.cuda.synchronize()
            vis = (radii > 0).cpu().numpy()
            assert vis.sum() > 300
            out = {}
            for which, width in ((4, 6), (3, 4)):
                p = ctypes.c_void_p()
                assert L.fdgs_geom_field(ctypes.c_void_p(st.geom.data_ptr()), 512, which, ctypes.byref(p)) == 0
                t = torch.empty(512 * width, device=dev)
                assert hip.hipMemcpy(ctypes.c_void_p(t.data_ptr()), p, ctypes.c_size_t(512 * width * 4), ctypes.c_int(3)) == 0
                out[which] = t.reshape(512, width).cpu().numpy()
            ref = Z[f"cov3D.mod{mod}"]
            scale = np.abs(ref[vis]).max(axis=1, keepdims=True)
            assert (np.abs(out[4][vis] - ref[vis]) / scale).max() < 1e-5
            np.testing.assert_allclose(out[3][vis, :3], Z[f"sh.colors.deg{deg}"][vis], rtol=0, atol=2e-6)


@pytest.mark.parametrize("cfg", ["dynerf_default", "dnerf_bouncingballs"])
def test_fused_backward_node_equals_two_node_form(cfg, monkeypatch):
    """render()'s fine stage as ONE autograd node (rasterizer backward writing straight into the deformation backward's packed rows,
    fdgs_raster_deform_epilogue) against the two-node form (five gradient tensors through autograd + the packing kernel): same
    image bit for bit, every gradient equal to summation noise; also with a gradient on the depth image and under no_grad."""
    fd = _fd()
    dev = torch.device("cuda:0")
    cam = synthetic.make_camera(240, 180, theta_deg=-35.0, time=0.45).to(dev)
    gen = torch.Generator().manual_seed(4)
    w = torch.randn(3, 180, 240, generator=gen).to(dev)
    wd = torch.randn(1, 180, 240, generator=gen).to(dev)
    out = []
    for fused in (True, False):
        monkeypatch.setattr(fd.renderer, "FUSED_BACKWARD", fused)
        pc = _model(7001, cfg, seed=23)           # (odd count: the last 128-row block of the packed rows is partly padding)
        res = fd.render(cam, pc, _Pipe(), torch.zeros(3, device=dev), stage="fine")
        ((res["render"] * w).sum() + (res["depth"] * wd).sum()).backward()
        out.append((res, {k: p.grad.clone() for k, p in pc.named_parameters() if p.grad is not None}, res["viewspace_points"].grad.clone()))
        with torch.no_grad():
            r2 = fd.render(cam, pc, _Pipe(), torch.zeros(3, device=dev), stage="fine")
        assert torch.equal(r2["render"], res["render"]) and r2["render"].grad_fn is None
    (ra, ga, va), (rb, gb, vb) = out
    assert torch.equal(ra["render"], rb["render"]) and torch.equal(ra["depth"], rb["depth"]) and torch.equal(ra["radii"], rb["radii"])
    assert set(ga) == set(gb)
    worst = max((rel_l2(ga[k].cpu().numpy(), gb[k].cpu().numpy()), k) for k in ga)
    print(f"[{cfg}] fused vs two-node: worst gradient rel-L2 {worst[0]:.1e} ({worst[1]})")
    assert worst[0] < 2e-5
    assert rel_l2(va.cpu().numpy(), vb.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("n", [2, 70, 129, 1000])   # (one point alone makes the bounding-box aabb degenerate: 0/0 in any implementation)
@pytest.mark.parametrize("assign", [True, False])
def test_fused_backward_small_and_ragged_counts_both_epilogue_modes(n, assign, monkeypatch):
    """Gaussian counts that do not fill a 64-row wave block / a 128-row padded block, and both modes of the rasterizer backward's
    deformation epilogue (assign into an uninitialised arena / accumulate into a zero-filled one), against the two-node form."""
    fd = _fd()
    dev = torch.device("cuda:0")
    cam = synthetic.make_camera(96, 80, theta_deg=15.0, time=0.3).to(dev)
    w = torch.randn(3, 80, 96, generator=torch.Generator().manual_seed(9)).to(dev)
    grads = []
    for fused in (True, False):
        monkeypatch.setattr(fd.renderer, "FUSED_BACKWARD", fused)
        monkeypatch.setattr(fd.renderer, "EPILOGUE_ASSIGN", assign)
        pc = _model(n, "dynerf_default", seed=31, boost=2.5)
        # poison the caching allocator's free blocks: an "assign" epilogue that forgot a row would hand back garbage, not zeros
        junk = torch.full((4_000_000,), float("nan"), device=dev)
        del junk
        res = fd.render(cam, pc, _Pipe(), torch.zeros(3, device=dev), stage="fine")
        (res["render"] * w).sum().backward()
        grads.append({k: p.grad.clone() for k, p in pc.named_parameters() if p.grad is not None})
    a, b = grads
    assert set(a) == set(b)
    for k in a:
        assert torch.isfinite(a[k]).all(), k
        assert rel_l2(a[k].cpu().numpy(), b[k].cpu().numpy()) < 2e-5, (k, n, assign)


@pytest.mark.parametrize("nviews", [1, 3])
def test_render_views_equals_per_view_render_with_summed_loss(nviews):
    """fdgs.render_views: the views of one optimizer step behind one autograd node and one gradient arena, against the reference's
    loop of render() calls whose losses are summed before one backward (train.py:180-219): identical images / radii / depth, every
    parameter gradient and every view's viewspace gradient equal to summation noise."""
    fd = _fd()
    dev = torch.device("cuda:0")
    cams = [synthetic.make_camera(160, 120, theta_deg=-60.0 + 70.0 * v, time=0.15 + 0.3 * v).to(dev) for v in range(nviews)]
    ws = [torch.randn(3, 120, 160, generator=torch.Generator().manual_seed(40 + v)).to(dev) for v in range(nviews)]
    bg = torch.zeros(3, device=dev)
    pc = _model(5003, "dynerf_default", seed=12)
    res_a = fd.render_views(cams, pc, _Pipe(), bg, stage="fine")
    sum((r["render"] * w).sum() for r, w in zip(res_a, ws)).backward()
    ga = {k: p.grad.clone() for k, p in pc.named_parameters() if p.grad is not None}
    va = [r["viewspace_points"].grad.clone() for r in res_a]
    for p in pc.parameters():
        p.grad = None
    res_b = [fd.render(c, pc, _Pipe(), bg, stage="fine") for c in cams]
    sum((r["render"] * w).sum() for r, w in zip(res_b, ws)).backward()
    gb = {k: p.grad.clone() for k, p in pc.named_parameters() if p.grad is not None}
    for ra, rb in zip(res_a, res_b):
        for key in ("render", "depth", "radii", "visibility_filter"):
            assert torch.equal(ra[key], rb[key]), key
    assert set(ga) == set(gb)
    worst = max((rel_l2(ga[k].cpu().numpy(), gb[k].cpu().numpy()), k) for k in ga)
    print(f"[{nviews} views] batched node vs per-view graphs: worst gradient rel-L2 {worst[0]:.1e} ({worst[1]})")
    assert worst[0] < 2e-5
    for v in range(nviews):
        assert rel_l2(va[v].cpu().numpy(), res_b[v]["viewspace_points"].grad.cpu().numpy()) < 1e-6


def test_backward_skips_tiles_the_rasterizer_gave_no_gradient(monkeypatch):
    """A camera that sees well under half of a spatially ordered set: the Gaussians outside the frustum (and the occluded ones) get
    all-zero rows from fdgs_raster_bwd's epilogue, which also leaves the per-tile flags; the deformation backward then walks only the
    live tiles.  Gradients must equal the all-tiles run (FDGS_SKIP_DEAD=0) to the re-association of the sums."""
    fd = _fd()
    dev = torch.device("cuda:0")
    pc = synthetic.SynthModel(40_000, "dynerf_default", seed=23)
    fd.densify.spatial_reorder(pc)
    pc = pc.to(dev)
    cam = synthetic.make_camera(320, 240, theta_deg=20.0, time=0.6, radius=1.2).to(dev)     # inside the cloud: most of it is behind / beside the camera
    w = torch.randn(3, 240, 320, generator=torch.Generator().manual_seed(4)).to(dev)
    monkeypatch.setattr(fd.deformation, "COUNT_LIVE_TILES", True)
    runs = {}
    for skip in ("1", "0"):
        set_knob("skip_dead", skip)
        res, g, v = _render_and_grads(pc, cam, _Pipe(), "fine", w)
        runs[skip] = (res, g, v, fd.deformation.last_live_tiles)
    vis = float((runs["1"][0]["radii"] > 0).float().mean())
    live, total = runs["1"][3][0], runs["1"][3][1]
    print(f"visible {vis:.2f} of the set; live tiles {live} of {total}")
    assert 0.02 < vis < 0.6 and live < 0.6 * total and runs["0"][3][0] == total
    assert torch.equal(runs["1"][0]["render"], runs["0"][0]["render"])
    for k in runs["0"][1]:
        e = rel_l2(runs["1"][1][k].cpu().numpy(), runs["0"][1][k].cpu().numpy())
        assert e < 2e-6, (k, e)
    assert rel_l2(runs["1"][2].cpu().numpy(), runs["0"][2].cpu().numpy()) < 2e-6      # (blending atomics: order differs run to run)
    # packed_rows_ready = 1 (epilogue tile_flags = 0: rows without flags) and 2 (flags, every row written): the C-ABI's other two modes.
    # Mode 1 once read per-tile flags nothing had written: the scratch block is recycled by the caching allocator, so a
    # backward of ANOTHER camera in skip mode first leaves that camera's 0 / 1 flags where mode 1 would have looked
    set_knob("skip_dead", "1")
    other = synthetic.make_camera(320, 240, theta_deg=200.0, time=0.6, radius=1.2).to(dev)
    for mode in (0, 1):
        _render_and_grads(pc, other, _Pipe(), "fine", w)
        monkeypatch.setattr(fd.renderer, "EPILOGUE_TILE_FLAGS", mode)
        res, g, v = _render_and_grads(pc, cam, _Pipe(), "fine", w)
        monkeypatch.setattr(fd.renderer, "EPILOGUE_TILE_FLAGS", None)
        if mode == 0:
            assert fd.deformation.last_live_tiles[0] == total
        for k in runs["0"][1]:
            e = rel_l2(g[k].cpu().numpy(), runs["0"][1][k].cpu().numpy())
            assert e < 2e-6, (mode, k, e)
