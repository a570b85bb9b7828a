"""GPU parity tests of the HIP rasterizer (through the C-ABI) against the CPU oracle. Run with -m gpu on MI355X."""
import importlib
import math

import numpy as np
import pytest

from conftest import set_knob
import torch

from oracle.raster_oracle import RasterOracle
from scenes import raster_scene, rel_l2

pytestmark = pytest.mark.gpu


def _mod():
    return importlib.import_module("4dgaussians_amd")


def _settings(sc, dev, debug=True):
    R = _mod().rasterizer
    return R.GaussianRasterizationSettings(
        image_height=sc["image_height"], image_width=sc["image_width"], tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"],
        bg=torch.tensor(sc["bg"], device=dev), scale_modifier=1.0, viewmatrix=torch.tensor(sc["viewmatrix"], device=dev),
        projmatrix=torch.tensor(sc["projmatrix"], device=dev), sh_degree=sc["sh_degree"],
        campos=torch.tensor(sc["campos"], device=dev), prefiltered=False, debug=debug)


def _dev_field(ptr_fn, shape, dtype, dev):
    """Wrap a device pointer returned by the C-ABI accessors into a torch tensor copy (via ctypes + hipMemcpy)."""
    import ctypes
    n = int(np.prod(shape))
    t = torch.empty(n, dtype=dtype, device=dev)
    p = ptr_fn()
    nbytes = n * t.element_size()
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipMemcpy(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(p), ctypes.c_size_t(nbytes), ctypes.c_int(3))
    assert rc == 0
    return t.reshape(shape).cpu().numpy()


def _geom(state, which, shape, dtype, dev):
    import ctypes
    L = _mod()._lib.lib()

    def f():
        out = ctypes.c_void_p()
        assert L.fdgs_geom_field(ctypes.c_void_p(state.geom.data_ptr()), state.params.P, which, ctypes.byref(out)) == 0
        return out.value
    return _dev_field(f, shape, dtype, dev)


def _binning(state, which, dev):
    import ctypes
    L = _mod()._lib.lib()

    def f():
        out = ctypes.c_void_p()
        assert L.fdgs_binning_field(ctypes.c_void_p(state.binning.data_ptr()), state.capacity, state.params.W,      # (the buffer is LAID OUT for the capacity)
                                    state.params.H, which, ctypes.byref(out)) == 0
        return out.value
    return _dev_field(f, (state.num_rendered,), torch.int32, dev).view(np.uint32)


def _img(state, which, shape, dtype, dev):
    import ctypes
    L = _mod()._lib.lib()

    def f():
        out = ctypes.c_void_p()
        assert L.fdgs_img_field(ctypes.c_void_p(state.img.data_ptr()), state.params.W, state.params.H, which, ctypes.byref(out)) == 0
        return out.value
    return _dev_field(f, shape, dtype, dev)


def _run_forward(sc, dev):
    R = _mod().rasterizer
    t = {k: torch.tensor(sc[k], device=dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    return R.rasterize_forward(_settings(sc, dev), t["means3D"], t["shs"], None, t["opacities"], t["scales"], t["rotations"], None), t


CASES = [
    dict(n=20000, width=400, height=400, seed=0, theta=30.0),                    # BASELINE config 1 shape
    dict(n=3000, width=201, height=77, seed=1, theta=-100.0, scale_boost=4.0),   # ragged image edge, big splats
    dict(n=500, width=64, height=48, seed=2, theta=170.0, sh_degree=1, scale_boost=10.0),
    dict(n=5000, width=320, height=240, seed=3, theta=0.0, sh_degree=0, extent=3.0),  # many culled / behind camera
    dict(n=300000, width=1352, height=1014, seed=6666, theta=-60.0),             # BASELINE config 4 shape, VALUES compared
    dict(n=100000, width=800, height=800, seed=11, theta=100.0, scale_boost=2.0),  # BASELINE config 2 shape
]


def _culled_pairs_contribute_nothing(o, listed_gid, listed_tile, ok, W, H):
    """Every (Gaussian, tile) pair of the reference's list (all tiles of the 3-sigma square) that the device list lacks must
    be unable to reach alpha >= 1/255 on any pixel of the tile (float64 re-evaluation of the per-pixel rule, Appendix B.3)."""
    gx = (W + 15) // 16
    rect = o.field("rect").astype(np.int64)
    co = o.field("conic_opacity").astype(np.float64)
    xy = o.field("xy").astype(np.float64)
    have = set(zip(listed_gid.tolist(), listed_tile.tolist()))
    miss_g, miss_t = [], []
    for gi in np.nonzero(ok)[0]:
        x0, y0, x1, y1 = rect[gi]
        for ty in range(y0, y1):
            for tx in range(x0, x1):
                if (gi, ty * gx + tx) not in have:
                    miss_g.append(gi); miss_t.append(ty * gx + tx)
    if not miss_g:
        return 0
    g, t = np.array(miss_g), np.array(miss_t)
    px = ((t % gx) * 16)[:, None, None] + np.arange(16)[None, None, :]
    py = ((t // gx) * 16)[:, None, None] + np.arange(16)[None, :, None]
    dx, dy = xy[g, 0][:, None, None] - px, xy[g, 1][:, None, None] - py
    power = -0.5 * (co[g, 0][:, None, None] * dx * dx + co[g, 2][:, None, None] * dy * dy) - co[g, 1][:, None, None] * dx * dy
    alpha = co[g, 3][:, None, None] * np.exp(np.minimum(power, 0))
    live = (power <= 0) & (alpha >= 1.0 / 255) & (px < W) & (py < H)
    assert not live.any(), f"{int(live.reshape(len(g), -1).any(1).sum())} culled pairs have contributing pixels"
    return len(g)


@pytest.mark.parametrize("cull", [False, True])
@pytest.mark.parametrize("case", CASES)
def test_stagewise_forward_parity(case, cull, monkeypatch):
    set_knob("tile_cull", "1" if cull else "0")
    dev = torch.device("cuda:0")
    sc = raster_scene(**case)
    o = RasterOracle(**sc)
    (color, radii, depth, st), _ = _run_forward(sc, dev)
    torch.cuda.synchronize()
    P, W, H = st.params.P, sc["image_width"], sc["image_height"]
    # --- K1: per-Gaussian projection
    radii = radii.cpu().numpy()
    mism = (radii != o.radii)
    assert mism.mean() < 2e-4, f"radii mismatch fraction {mism.mean()}"
    ok = ~mism & (o.radii > 0)
    recA = _geom(st, 1, (P, 4), torch.float32, dev)
    recB = _geom(st, 2, (P, 4), torch.float32, dev)
    recC = _geom(st, 3, (P, 4), torch.float32, dev)
    tiles = _geom(st, 5, (P,), torch.int32, dev)
    if cull:    # exact tile culling: a subset of the reference's list ...
        assert np.all(tiles[ok] <= o.field("tiles_touched")[ok])
    else:       # ... or the reference's list itself
        assert np.array_equal(tiles[ok], o.field("tiles_touched")[ok])
    np.testing.assert_allclose(recA[ok, :2], o.field("xy")[ok], rtol=0, atol=2e-3)
    co = o.field("conic_opacity")
    assert rel_l2(recA[ok, 2:4], co[ok, 0:2]) < 1e-5 and rel_l2(recB[ok, 0], co[ok, 2]) < 1e-5
    np.testing.assert_allclose(recB[ok, 1], co[ok, 3], rtol=0, atol=0)
    np.testing.assert_allclose(recB[ok, 2], o.field("depth")[ok], rtol=1e-6)
    np.testing.assert_allclose(recC[ok, :3], o.field("rgb")[ok], rtol=0, atol=1e-5)
    # --- K2-K5: binning invariants on the device's own state (bit-exact integer work)
    Rn = st.num_rendered
    assert Rn == int(tiles.astype(np.int64).sum())
    gid = _binning(st, 0, dev)
    tile = _binning(st, 1, dev)
    dep = recB[:, 2].view(np.uint32)[gid].astype(np.uint64)
    key = (tile.astype(np.uint64) << np.uint64(32)) | dep
    assert np.all(key[1:] >= key[:-1]), "pair list not sorted by (tile, depth)"
    same = key[1:] == key[:-1]
    assert np.all(gid[1:][same] > gid[:-1][same]), "ties must keep Gaussian-index order (stable sort)"
    rect = _geom(st, 7, (P, 2), torch.int32, dev).view(np.uint32)
    gx = (W + 15) // 16
    tx, ty = tile % gx, tile // gx
    assert np.all((tx >= (rect[gid, 0] & 0xFFFF)) & (tx < (rect[gid, 1] & 0xFFFF)) & (ty >= (rect[gid, 0] >> 16)) & (ty < (rect[gid, 1] >> 16)))
    cnt = np.bincount(gid, minlength=P)
    assert np.array_equal(cnt, tiles)
    if cull and P <= 5000:   # ... whose missing pairs cannot contribute to any pixel (brute force, small scenes)
        dropped = _culled_pairs_contribute_nothing(o, gid, tile, ok, W, H)
        print(f"[{case}] culled {dropped} of {int(o.field('tiles_touched')[ok].sum())} pairs")
        assert dropped > 0 or case.get("scale_boost", 1.0) >= 10.0      # (screen-filling splats reach every tile)
    ranges = _img(st, 2, (gx * ((H + 15) // 16), 2), torch.int32, dev).view(np.uint32)
    tc = np.bincount(tile, minlength=ranges.shape[0])
    assert np.array_equal(ranges[:, 1] - ranges[:, 0], tc)
    nz = tc > 0
    assert np.all(tile[ranges[nz, 0]] == np.nonzero(nz)[0]) and np.all(tile[ranges[nz, 1] - 1] == np.nonzero(nz)[0])
    # --- K6: image
    color, depth = color.cpu().numpy(), depth.cpu().numpy()
    d = np.abs(color - o.color)
    mse = float(((color - o.color) ** 2).mean())
    psnr = 10 * math.log10(1.0 / max(mse, 1e-20))
    print(f"[{case}] R={Rn} max|dC|={d.max():.3e} mean|dC|={d.mean():.3e} psnr={psnr:.1f} dB")
    assert d.mean() < 2e-6 and np.quantile(d, 0.9999) < 5e-5 and psnr > 80.0
    dd = np.abs(depth - o.depth)
    assert dd.mean() < 2e-5 and np.quantile(dd, 0.9999) < 5e-4
    fT, nc = o.image_state()
    nc_g = _img(st, 1, (H, W), torch.int32, dev)
    if not cull:   # (n_contrib is a position in the pair list: only comparable when the lists are the reference's)
        assert (nc_g != nc.astype(np.int32)).mean() < 1e-3
    else:
        assert np.all(nc_g <= nc.astype(np.int32) + 1)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("with_depth", [False, True])
def test_backward_parity(case, with_depth):
    dev = torch.device("cuda:0")
    sc = raster_scene(**case)
    o = RasterOracle(**sc)
    R = _mod().rasterizer
    t = {k: torch.tensor(sc[k], device=dev, requires_grad=True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    means2D = torch.zeros_like(t["means3D"], requires_grad=True)
    rast = R.GaussianRasterizer(_settings(sc, dev))
    color, radii, depth = rast(means3D=t["means3D"], means2D=means2D, shs=t["shs"], colors_precomp=None,
                               opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    rng = np.random.default_rng(7)
    target = rng.random(o.color.shape).astype(np.float32)
    dc = np.sign(o.color - target).astype(np.float32) / o.color.size  # L1-loss gradient (utils/loss_utils.py:20-21)
    dd = (rng.standard_normal(o.depth.shape).astype(np.float32) / o.depth.size) if with_depth else None
    loss = (color * torch.tensor(dc, device=dev)).sum()
    if with_depth:
        loss = loss + (depth * torch.tensor(dd, device=dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    g = o.backward(dc, dd)
    names = dict(means3D="means3D", shs="shs", opacities="opacities", scales="scales", rotations="rotations")
    errs = {k: rel_l2(t[k].grad.cpu().numpy(), g[v].reshape(t[k].shape)) for k, v in names.items()}
    errs["means2D"] = rel_l2(means2D.grad.cpu().numpy(), g["means2D"])
    print(f"[{case} depth={with_depth}] grad rel-L2: " + ", ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    for k, v in errs.items():
        assert v < 1e-3, (k, v)


def test_colors_precomp_and_cov3d_precomp_paths():
    dev = torch.device("cuda:0")
    sc = raster_scene(2000, 160, 120, seed=4, scale_boost=3.0)
    rot = sc["rotations"].astype(np.float64); s = sc["scales"].astype(np.float64)
    r, x, y, z = rot.T
    Rm = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                   2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    Lm = Rm * s[:, None, :]
    S = Lm @ Lm.transpose(0, 2, 1)
    cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1).astype(np.float32)
    cols = np.random.default_rng(0).random((2000, 3)).astype(np.float32)
    sc2 = {k: v for k, v in sc.items() if k not in ("shs", "scales", "rotations")}
    o = RasterOracle(**sc2, colors_precomp=cols, cov3D_precomp=cov)
    R = _mod().rasterizer
    t = dict(means3D=torch.tensor(sc["means3D"], device=dev, requires_grad=True), cols=torch.tensor(cols, device=dev, requires_grad=True),
             cov=torch.tensor(cov, device=dev, requires_grad=True), op=torch.tensor(sc["opacities"], device=dev, requires_grad=True))
    m2 = torch.zeros(2000, 3, device=dev, requires_grad=True)
    color, radii, depth = R.GaussianRasterizer(_settings(sc, dev))(means3D=t["means3D"], means2D=m2, colors_precomp=t["cols"],
                                                                   opacities=t["op"], cov3D_precomp=t["cov"])
    assert np.abs(color.detach().cpu().numpy() - o.color).mean() < 2e-6
    dc = np.random.default_rng(1).standard_normal(o.color.shape).astype(np.float32)
    (color * torch.tensor(dc, device=dev)).sum().backward()
    g = o.backward(dc)
    assert rel_l2(t["cols"].grad.cpu().numpy(), g["colors"]) < 1e-3
    assert rel_l2(t["cov"].grad.cpu().numpy(), g["cov3D"]) < 1e-3
    assert rel_l2(t["means3D"].grad.cpu().numpy(), g["means3D"]) < 1e-3


def test_edge_cases_empty_and_all_culled():
    dev = torch.device("cuda:0")
    sc = raster_scene(64, 50, 34, seed=5)
    R = _mod().rasterizer
    rs = _settings(sc, dev)
    z = lambda *s: torch.zeros(*s, device=dev)
    color, radii, depth = R.GaussianRasterizer(rs)(means3D=z(0, 3), means2D=z(0, 3), shs=z(0, 16, 3), opacities=z(0, 1), scales=z(0, 3),
                                                   rotations=z(0, 4))
    assert color.shape == (3, 34, 50) and torch.allclose(color, torch.ones_like(color)) and radii.numel() == 0
    far = torch.tensor(sc["means3D"], device=dev) * 0 + 50.0
    far.requires_grad_(True)
    color, radii, depth = R.GaussianRasterizer(rs)(means3D=far, means2D=z(64, 3), shs=torch.tensor(sc["shs"], device=dev),
                                                   opacities=torch.tensor(sc["opacities"], device=dev), scales=torch.tensor(sc["scales"], device=dev),
                                                   rotations=torch.tensor(sc["rotations"], device=dev))
    assert (radii == 0).all() and torch.allclose(color, torch.ones_like(color)) and (depth == 0).all()
    color.sum().backward()
    assert (far.grad == 0).all()
    with pytest.raises(Exception):
        R.GaussianRasterizer(rs)(means3D=far, means2D=z(64, 3), opacities=z(64, 1), scales=z(64, 3), rotations=z(64, 4))
    vis = R.GaussianRasterizer(rs).markVisible(torch.tensor(sc["means3D"], device=dev))
    pv = sc["means3D"] @ sc["viewmatrix"].reshape(4, 4)[:3, 2] + sc["viewmatrix"].reshape(4, 4)[3, 2]
    assert np.array_equal(vis.cpu().numpy(), pv > 0.2)


def test_full_size_properties_config4_shape():
    """BASELINE config 4 shape (300k Gaussians, 1352x1014): size-independent properties only (the oracle takes ~10 s)."""
    dev = torch.device("cuda:0")
    sc = raster_scene(300000, 1352, 1014, seed=6666, theta=-60.0)
    (color, radii, depth, st), t = _run_forward(sc, dev)
    torch.cuda.synchronize()
    P = st.params.P
    tiles = _geom(st, 5, (P,), torch.int32, dev)
    assert st.num_rendered == int(tiles.astype(np.int64).sum())
    assert not np.any((tiles > 0) & (radii.cpu().numpy() <= 0))      # (culling may leave a visible Gaussian without tiles)
    gid, tile = _binning(st, 0, dev), _binning(st, 1, dev)
    recB = _geom(st, 2, (P, 4), torch.float32, dev)
    key = (tile.astype(np.uint64) << np.uint64(32)) | recB[:, 2].view(np.uint32)[gid].astype(np.uint64)
    assert np.all(key[1:] >= key[:-1])
    assert np.array_equal(np.bincount(gid, minlength=P), tiles)
    c = color.cpu().numpy()
    assert np.isfinite(c).all() and c.min() >= 0.0
    # idempotence: a second run gives the bit-identical image (forward has no atomics)
    (color2, _, _, _), _ = _run_forward(sc, dev)
    assert torch.equal(color, color2)
    # permutation invariance up to equal-depth ties
    perm = torch.randperm(P, generator=torch.Generator().manual_seed(0)).numpy()
    sc_p = dict(sc)
    for k in ("means3D", "shs", "opacities", "scales", "rotations"):
        sc_p[k] = np.ascontiguousarray(sc[k][perm])
    (color3, _, _, _), _ = _run_forward(sc_p, dev)
    assert (color3 - color).abs().max().item() < 1e-5


@pytest.mark.parametrize("case", CASES[:3])
def test_tile_culling_is_exact(case, monkeypatch):
    """Forward outputs with and without exact tile culling are BIT-identical (a dropped pair never touches T or an
    accumulator); backward gradients agree to atomic-ordering noise; the pair list shrinks."""
    dev = torch.device("cuda:0")
    sc = raster_scene(**case)
    R = _mod().rasterizer
    outs = []
    for flag in ("0", "1"):
        set_knob("tile_cull", flag)
        t = {k: torch.tensor(sc[k], device=dev).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        m2d = torch.zeros(t["means3D"].shape[0], 3, device=dev, requires_grad=True)
        color, radii, depth = R.GaussianRasterizer(_settings(sc, dev))(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"],
                                                                    scales=t["scales"], rotations=t["rotations"])
        gen = torch.Generator().manual_seed(1)
        w = torch.randn(color.shape, generator=gen).to(dev)
        (color * w).sum().backward()
        (_, _, _, st), _ = _run_forward(sc, dev)
        outs.append((color.detach(), radii, depth.detach(), {k: v.grad.clone() for k, v in t.items()}, m2d.grad.clone(), st.num_rendered))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for k in a[3]:
        assert rel_l2(b[3][k].cpu().numpy(), a[3][k].cpu().numpy()) < 1e-5, k
    assert rel_l2(b[4].cpu().numpy(), a[4].cpu().numpy()) < 1e-5
    assert b[5] <= a[5] and (b[5] < a[5] or case.get("scale_boost", 1.0) >= 10.0)
    print(f"[{case}] pairs {a[5]} -> {b[5]} ({100.0 * b[5] / a[5]:.1f} %)")


@pytest.mark.parametrize("with_depth", [False, True])
def test_blending_kernel_forms_agree(with_depth, monkeypatch):
    """The blending kernels exist in several shapes selected by development knobs -- backward: one wave per tile with four pixels per
    lane (rbwd_ppl = 4: what images of more than 4 096 tiles take by default), two waves with two pixels per lane (rbwd_ppl = 2: the default
    below that, i.e. for every image of this file), the 256-thread form of rounds 1-2 (rbwd_ppl = 0).  All of them
    must produce the same gradients up to the association of the sums (and the forward, which has one form, the same image every time)."""
    dev = torch.device("cuda:0")
    sc = raster_scene(6000, 232, 152, seed=11, scale_boost=2.0)
    R = _mod().rasterizer
    rng = np.random.default_rng(3)
    wc = torch.tensor(rng.standard_normal((3, 152, 232)).astype(np.float32), device=dev)
    wd = torch.tensor(rng.standard_normal((1, 152, 232)).astype(np.float32), device=dev)
    outs = {}
    for name, ppl in (("default", 4), ("bwd2", 2), ("bwd0", 0)):      # ("default": the reference leg of the comparison below)
        set_knob("rbwd_ppl", ppl)
        t = {k: torch.tensor(sc[k], device=dev, requires_grad=True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        means2D = torch.zeros_like(t["means3D"], requires_grad=True)
        rast = R.GaussianRasterizer(_settings(sc, dev))
        color, radii, depth = rast(means3D=t["means3D"], means2D=means2D, shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
                                   scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
        loss = (color * wc).sum() + ((depth * wd).sum() if with_depth else 0.0)
        loss.backward()
        torch.cuda.synchronize()
        outs[name] = (color.detach().cpu().numpy(), depth.detach().cpu().numpy(),
                      {k: v.grad.cpu().numpy() for k, v in t.items()} | {"means2D": means2D.grad.cpu().numpy()})
    c0, d0, g0 = outs["default"]
    for name, (c, d, g) in outs.items():
        assert np.abs(c - c0).max() <= 1e-6 and np.abs(d - d0).max() <= 1e-5, name       # (forward forms: same arithmetic, fma order only)
        for k in g0:
            assert rel_l2(g[k], g0[k]) < 5e-6, (name, k, rel_l2(g[k], g0[k]))


def test_capacity_paths_equal_the_blocking_exact_path(monkeypatch):
    """Once a pair count has been seen for (image size, Gaussian count) rasterize_forward sizes the binning buffer from a predicted count and
    queues the whole forward with ONE C call (fdgs_raster_fwd_capacity: the kernels read the true count on the device); the default mode
    then verifies the count before it returns, the opt-in "capacity" mode does not.  Same image, depth, radii and per-pixel bookkeeping bit
    for bit, same sorted lists and ranges, same gradients as the blocking exact path, training frame or evaluation frame."""
    dev = torch.device("cuda:0")
    R = _mod().rasterizer
    sc = raster_scene(30000, 640, 480, seed=21, scale_boost=1.5)
    rng = np.random.default_rng(5)
    wc = torch.tensor(rng.standard_normal((3, 480, 640)).astype(np.float32), device=dev)
    outs = {}
    for mode in ("exact", "auto", "capacity"):
        monkeypatch.setattr(R, "BINNING", mode)
        if mode != "exact":
            assert (dev.index, 640, 480, 30000) in R._seen    # the exact frame above fed the predictor
        t = {k: torch.tensor(sc[k], device=dev, requires_grad=True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        color, radii, depth, st = R.rasterize_forward(_settings(sc, dev, debug=False), t["means3D"], t["shs"], None, t["opacities"], t["scales"],
                                                      t["rotations"], None, expect_backward=True)
        if mode == "auto":
            # the verified path: the count is known when the call returns, the buffer is laid out for the predicted capacity
            assert st.count.R is not None and st.count.capacity is None and st.capacity >= st.count.R > 0 and st.capacity % 4096 == 0
            # ... and an evaluation frame (no backward expected) takes the same verified path
            c_eval, _, d_eval, st_eval = R.rasterize_forward(_settings(sc, dev, debug=False), t["means3D"], t["shs"], None, t["opacities"], t["scales"],
                                                             t["rotations"], None)
            assert st_eval.count.R == st.count.R and torch.equal(c_eval, color) and torch.equal(d_eval, depth)
        elif mode == "capacity":
            assert st.count.capacity is not None and st.capacity >= st.num_rendered > 0 and st.capacity % 4096 == 0
        else:
            assert st.count.capacity is None and st.capacity == st.num_rendered
        g = R.rasterize_backward(st, wc)
        torch.cuda.synchronize()
        n = st.num_rendered
        outs[mode] = dict(color=color.clone(), radii=radii.clone(), depth=depth.clone(), n=n, gid=_binning(st, 0, dev)[:n], tile=_binning(st, 1, dev)[:n],
                          ranges=_img(st, 2, (((480 + 15) // 16) * ((640 + 15) // 16), 2), torch.int32, dev),
                          ncontrib=_img(st, 1, (480, 640), torch.int32, dev), grads={k: v.clone() for k, v in g.items() if v is not None})
    a = outs["exact"]
    for mode in ("auto", "capacity"):
        b = outs[mode]
        assert a["n"] == b["n"]
        for k in ("color", "radii", "depth"):
            assert torch.equal(a[k], b[k]), (mode, k)
        for k in ("gid", "tile", "ranges", "ncontrib"):
            assert np.array_equal(a[k], b[k]), (mode, k)
        for k in a["grads"]:
            assert rel_l2(b["grads"][k].cpu().numpy(), a["grads"][k].cpu().numpy()) < 5e-6, (mode, k)


def test_capacity_overflow_in_the_default_mode_returns_the_exact_frame(monkeypatch):
    """The default binning mode never returns an image that differs from the exact path: a frame that lists more pairs than its speculative
    binning buffer holds is finished exactly (fdgs_bin_sort + fdgs_render_fwd on a buffer of the true size) BEFORE rasterize_forward returns
    -- same image, depth, radii, lists, per-pixel bookkeeping and gradients as the blocking path -- and counted in `capacity_reruns`."""
    dev = torch.device("cuda:0")
    R = _mod().rasterizer
    sc = raster_scene(20000, 416, 304, seed=22, scale_boost=2.0)
    rng = np.random.default_rng(6)
    wc = torch.tensor(rng.standard_normal((3, 304, 416)).astype(np.float32), device=dev)
    t = {k: torch.tensor(sc[k], device=dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    args = (t["means3D"], t["shs"], None, t["opacities"], t["scales"], t["rotations"], None)
    key = (dev.index, 416, 304, 20000)
    tiles = ((304 + 15) // 16) * ((416 + 15) // 16)

    def frame():
        color, radii, depth, st = R.rasterize_forward(_settings(sc, dev, debug=False), *args, expect_backward=True)
        g = R.rasterize_backward(st, wc)
        torch.cuda.synchronize()
        n = st.num_rendered
        return dict(color=color.clone(), radii=radii.clone(), depth=depth.clone(), n=n, gid=_binning(st, 0, dev)[:n], tile=_binning(st, 1, dev)[:n],
                    ranges=_img(st, 2, (tiles, 2), torch.int32, dev), ncontrib=_img(st, 1, (304, 416), torch.int32, dev),
                    grads={k: v.clone() for k, v in g.items() if v is not None}), st

    monkeypatch.setattr(R, "BINNING", "exact")
    want, st0 = frame()
    assert want["n"] > 40000
    monkeypatch.setattr(R, "BINNING", "auto")
    for too_low in (want["n"] // 4, 1):                                                              # a predictor that is far too low
        R._seen[key] = [too_low, 20000]
        before = R.capacity_reruns
        got, st1 = frame()
        assert R.capacity_reruns == before + 1 and R._seen[key][0] == want["n"]                      # detected, finished exactly, learnt
        assert st1.capacity == want["n"] == got["n"] and st1.count.capacity is None
        for k in ("color", "radii", "depth"):
            assert torch.equal(want[k], got[k]), k
        for k in ("gid", "tile", "ranges", "ncontrib"):
            assert np.array_equal(want[k], got[k]), k
        for k in want["grads"]:
            assert rel_l2(got["grads"][k].cpu().numpy(), want["grads"][k].cpu().numpy()) < 5e-6, k
    before = R.capacity_reruns
    got, st2 = frame()                                                                               # the predictor has learnt: no re-run
    assert R.capacity_reruns == before and st2.capacity >= want["n"] and torch.equal(got["color"], want["color"])


def test_opt_in_capacity_mode_reports_an_overflow_and_grows(monkeypatch):
    """BINNING = "capacity" (opt-in, never waits): a frame that lists more pairs than its binning buffer was sized for drops its FARTHEST
    pairs (never writes out of bounds), is reported with a RuntimeWarning when its count arrives, counted in `capacity_overflows`, and the
    next frame's capacity covers it."""
    import warnings
    dev = torch.device("cuda:0")
    R = _mod().rasterizer
    sc = raster_scene(20000, 416, 304, seed=22, scale_boost=2.0)
    monkeypatch.setattr(R, "BINNING", "capacity")
    t = {k: torch.tensor(sc[k], device=dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    args = (t["means3D"], t["shs"], None, t["opacities"], t["scales"], t["rotations"], None)
    key = (dev.index, 416, 304, 20000)
    R._seen.pop(key, None)
    color0, radii0, depth0, st0 = R.rasterize_forward(_settings(sc, dev, debug=False), *args, expect_backward=True)     # first frame of this (size, count): exact
    true_n = st0.num_rendered
    assert st0.count.capacity is None and true_n > 40000
    R._seen[key] = [true_n // 4, 20000]                                                                 # a predictor that is far too low
    before = R.capacity_overflows
    color1, radii1, depth1, st1 = R.rasterize_forward(_settings(sc, dev, debug=False), *args, expect_backward=True)
    assert st1.capacity < true_n
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert st1.num_rendered == true_n                                                               # (waits for the count, which reports the overflow)
    assert R.capacity_overflows == before + 1 and any(issubclass(x.category, RuntimeWarning) for x in w)
    torch.cuda.synchronize()
    assert torch.equal(radii1, radii0)
    assert bool(torch.isfinite(color1).all()) and float((color1 - color0).abs().max()) > 0.0            # far pairs are missing, nothing else broke
    n_kept = int(_img(st1, 2, (((304 + 15) // 16) * ((416 + 15) // 16), 2), torch.int32, dev)[:, 1].max())
    assert n_kept <= st1.capacity
    color2, radii2, depth2, st2 = R.rasterize_forward(_settings(sc, dev, debug=False), *args, expect_backward=True)     # the predictor has learnt
    assert st2.capacity >= true_n and st2.num_rendered == true_n
    assert torch.equal(color2, color0) and torch.equal(depth2, depth0)


@pytest.mark.parametrize("ppl", [4, 2, 0])
def test_heaviest_first_tile_order_changes_nothing_but_the_dispatch_order(ppl):
    """The blending kernels take their tiles heaviest-first per XCD (tuning knob tile_order, csrc/render.hip: tile_order_kernel): the order
    lists are permutations of each XCD's tiles of the image-order map, sorted by descending work to the 256 levels of the counting sort
    (forward: list length; backward: the walk length the forward left per tile), and forward outputs are bit-identical with and without it;
    gradients agree to the association of the atomics."""
    import ctypes
    dev = torch.device("cuda:0")
    R = _mod().rasterizer
    L = _mod()._lib.lib()
    Wd, Ht = 420, 300
    sc = raster_scene(24000, Wd, Ht, seed=31, scale_boost=1.6)
    rng = np.random.default_rng(7)
    wc = torch.tensor(rng.standard_normal((3, Ht, Wd)).astype(np.float32), device=dev)
    set_knob("rbwd_ppl", ppl)
    outs = {}
    for on in (1, 0):
        set_knob("tile_order", on)
        t = {k: torch.tensor(sc[k], device=dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        color, radii, depth, st = R.rasterize_forward(_settings(sc, dev, debug=False), t["means3D"], t["shs"], None, t["opacities"], t["scales"],
                                                      t["rotations"], None)
        g = R.rasterize_backward(st, wc)
        torch.cuda.synchronize()
        gx, gy = (Wd + 15) // 16, (Ht + 15) // 16
        per_xcd = -(-(-(-gy // 2)) // 8) * 2 * gx
        outs[on] = dict(color=color.clone(), depth=depth.clone(), ncontrib=_img(st, 1, (Ht, Wd), torch.int32, dev), grads={k: v.clone() for k, v in g.items() if v is not None},
                        ranges=_img(st, 2, (gx * gy, 2), torch.int32, dev), todo=_img(st, 3, (gx * gy,), torch.int32, dev),
                        order_f=_img(st, 4, (8, per_xcd), torch.int32, dev).view(np.uint32), order_b=_img(st, 5, (8, per_xcd), torch.int32, dev).view(np.uint32))
    a, b = outs[1], outs[0]
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["depth"], b["depth"]) and np.array_equal(a["ncontrib"], b["ncontrib"])
    for k in a["grads"]:
        assert rel_l2(a["grads"][k].cpu().numpy(), b["grads"][k].cpu().numpy()) < 5e-6, k
    # the forward's per-tile walk length = max n_contrib over the tile's pixels
    pad = np.zeros((gy * 16, gx * 16), np.int64)
    pad[:Ht, :Wd] = a["ncontrib"]
    assert np.array_equal(a["todo"].astype(np.int64), pad.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1))
    # the order lists: per XCD a permutation of its tiles, non-increasing in the sort's 256 levels
    length = (a["ranges"][:, 1].astype(np.int64) - a["ranges"][:, 0].astype(np.int64))
    for name, key in (("order_f", length), ("order_b", a["todo"].astype(np.int64))):
        seen = []
        for x in range(8):
            row = a[name][x]
            tiles = row[row != 0xFFFFFFFF].astype(np.int64)
            # image-order map of this XCD: groups of two tile rows, group G belongs to XCD G % 8
            mine = [ty * gx + tx for ty in range(gy) if (ty // 2) % 8 == x for tx in range(gx)]
            assert sorted(tiles.tolist()) == sorted(mine), (name, x)
            assert (row == 0xFFFFFFFF).sum() == per_xcd - len(mine)      # (padding slots of the map share the last level with empty tiles)
            mx = max(int(key[tiles].max()), 1) if len(tiles) else 1
            level = 255 - np.minimum((key[tiles].astype(np.float32) * np.float32(255.0 / mx)).astype(np.int64), 255)
            assert np.all(np.diff(level) >= -1), (name, x)         # (non-increasing work, to the sort's 256 levels +- one level of float rounding)
            seen += tiles.tolist()
        assert sorted(seen) == list(range(gx * gy))
