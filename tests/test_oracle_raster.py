"""Trust-building tests for oracle/raster_oracle.c (the reference rasterizer is an un-vendored CUDA submodule, so
no reference golden vector exists: parity unpinned).  (i) analytic backward of the C restatement == float64 autograd
of an independent dense PyTorch restatement; (ii) closed-form single-Gaussian cases; (iii) properties."""
import math

import numpy as np
import pytest
import torch

from oracle.raster_oracle import RasterOracle
from oracle import raster_torch
from scenes import raster_scene, rel_l2


def _torch_inputs(sc, dtype=torch.float64, grad=True):
    t = {}
    for k in ("means3D", "scales", "rotations", "opacities", "shs"):
        t[k] = torch.tensor(sc[k], dtype=dtype, requires_grad=grad)
    for k in ("viewmatrix", "projmatrix", "campos", "bg"):
        t[k] = torch.tensor(sc[k], dtype=dtype)
    return t


@pytest.mark.parametrize("seed,deg,boost", [(0, 3, 1.0), (1, 2, 3.0), (2, 0, 0.5), (3, 1, 8.0)])
def test_c_backward_matches_float64_autograd(seed, deg, boost):
    sc = raster_scene(60, 40, 36, seed=seed, sh_degree=deg, scale_boost=boost * 6.0, dtype=np.float64)
    o = RasterOracle(**sc, dtype=np.float64)
    t = _torch_inputs(sc)
    means2D = torch.zeros(60, 3, dtype=torch.float64, requires_grad=True)
    color, depth, radii = raster_torch.rasterize(means3D=t["means3D"], opacities=t["opacities"], viewmatrix=t["viewmatrix"],
                                                 projmatrix=t["projmatrix"], campos=t["campos"], bg=t["bg"],
                                                 image_height=sc["image_height"], image_width=sc["image_width"],
                                                 tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], sh_degree=deg, shs=t["shs"],
                                                 scales=t["scales"], rotations=t["rotations"], means2D=means2D)
    assert np.array_equal(radii.numpy(), o.radii)
    assert (o.radii > 0).sum() > 10
    np.testing.assert_allclose(color.detach().numpy(), o.color, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(depth.detach().numpy(), o.depth, rtol=1e-9, atol=1e-10)
    rng = np.random.default_rng(seed)
    dc = rng.standard_normal(o.color.shape)
    dd = rng.standard_normal(o.depth.shape)
    g = o.backward(dc, dd)
    loss = (color * torch.tensor(dc)).sum() + (depth * torch.tensor(dd)).sum()
    gt = torch.autograd.grad(loss, [t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], means2D])
    for name, a in zip(("means3D", "scales", "rotations", "opacities", "shs", "means2D"), gt):
        assert rel_l2(g[name].reshape(-1), a.numpy().reshape(-1)) < 1e-6, name


def test_float32_build_close_to_float64():
    sc = raster_scene(400, 64, 48, seed=5, scale_boost=3.0)
    o32 = RasterOracle(**sc, dtype=np.float32)
    o64 = RasterOracle(**sc, dtype=np.float64)
    assert np.abs(o32.color - o64.color).max() < 2e-5
    dc = np.random.default_rng(0).standard_normal(o32.color.shape).astype(np.float32)
    g32, g64 = o32.backward(dc), o64.backward(dc)
    for k in ("means3D", "scales", "rotations", "opacities", "shs"):
        assert rel_l2(g32[k], g64[k]) < 1e-3, k


def _single(px_offset=0.0, opacity=0.8, s=0.05, W=65, H=65, z=4.0, bg=(0.0, 0.0, 0.0)):
    # camera at origin looking down +z: identity view; projection with tanfov=0.5
    tanf = 0.5
    view = np.eye(4, dtype=np.float64)
    P = np.zeros((4, 4))
    zn, zf = 0.01, 100.0
    P[0, 0] = 1 / tanf; P[1, 1] = 1 / tanf; P[3, 2] = 1.0; P[2, 2] = zf / (zf - zn); P[2, 3] = -(zf * zn) / (zf - zn)
    proj = (view.T @ P.T)
    fx = W / (2 * tanf)
    # place so that the projected centre lands on pixel (W/2 - 0.5 + px_offset) -> ndc = (2*px+1)/W - 1
    px = (W - 1) / 2 + px_offset  # odd W: the optical axis passes through the centre of pixel (W-1)/2
    ndc = (2 * px + 1) / W - 1
    x = ndc * tanf * z
    return dict(means3D=np.array([[x, x, z]]), scales=np.full((1, 3), s), rotations=np.array([[1.0, 0, 0, 0]]),
                opacities=np.array([opacity]), colors_precomp=np.array([[0.2, 0.5, 0.9]]), viewmatrix=view.T.copy(),
                projmatrix=proj, campos=np.zeros(3), bg=np.array(bg), image_height=H, image_width=W, tanfovx=tanf,
                tanfovy=tanf, sh_degree=0), fx, int(px)


def test_single_isotropic_gaussian_closed_form():
    sc, fx, px = _single()
    o = RasterOracle(**sc, dtype=np.float64)
    sigma2 = (0.05 * fx / 4.0) ** 2 + 0.3
    # isotropic: mid^2-det = 0 -> the 0.1 floor under the sqrt applies
    assert o.radii[0] == math.ceil(3 * math.sqrt(sigma2 + math.sqrt(0.1)))
    # centre pixel: alpha = 0.8, colour = 0.8*c ; depth = 0.8*z
    np.testing.assert_allclose(o.color[:, px, px], 0.8 * np.array([0.2, 0.5, 0.9]), rtol=1e-6)
    np.testing.assert_allclose(o.depth[0, px, px], 0.8 * 4.0, rtol=1e-6)
    # 2 px to the right: alpha = 0.8*exp(-0.5*4/sigma2)
    a = 0.8 * math.exp(-0.5 * 4 / sigma2)
    np.testing.assert_allclose(o.color[1, px, px + 2], a * 0.5, rtol=1e-6)


def test_alpha_clamp_and_background():
    sc, fx, px = _single(opacity=1.0, bg=(1.0, 1.0, 1.0))
    o = RasterOracle(**sc, dtype=np.float64)
    np.testing.assert_allclose(o.color[0, px, px], 0.99 * 0.2 + 0.01 * 1.0, rtol=1e-9)
    # a pixel outside the 3-sigma tile rect sees only background
    assert o.color[0, 0, 0] == 1.0 and o.depth[0, 0, 0] == 0.0


def test_near_plane_cull():
    for z, vis in ((0.19, False), (0.21, True)):
        sc, _, _ = _single(z=z, s=0.001)
        o = RasterOracle(**sc, dtype=np.float64)
        assert (o.radii[0] > 0) == vis


def test_depth_order_of_two_overlapping():
    sc, fx, px = _single()
    sc["means3D"] = np.array([[sc["means3D"][0, 0], sc["means3D"][0, 1], 4.0], [sc["means3D"][0, 0] * 5 / 4, sc["means3D"][0, 1] * 5 / 4, 5.0]])
    sc["scales"] = np.full((2, 3), 0.05); sc["rotations"] = np.array([[1.0, 0, 0, 0]] * 2)
    sc["opacities"] = np.array([0.5, 0.5]); sc["colors_precomp"] = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    o = RasterOracle(**sc, dtype=np.float64)
    np.testing.assert_allclose(o.color[:, px, px], [0.5, 0.25, 0.0], atol=1e-9)
    # swap the order of the inputs: identical image (sorted by depth)
    for k in ("means3D", "scales", "rotations", "colors_precomp"):
        sc[k] = sc[k][::-1].copy()
    o2 = RasterOracle(**sc, dtype=np.float64)
    np.testing.assert_allclose(o2.color, o.color, atol=1e-12)


def test_zero_opacity_contributes_nothing_and_all_culled_is_background():
    sc = raster_scene(50, 32, 32, seed=9, scale_boost=4.0, dtype=np.float64)
    base = RasterOracle(**sc, dtype=np.float64).color
    sc2 = dict(sc)
    sc2["means3D"] = np.concatenate([sc["means3D"], sc["means3D"][:7] * 0.9])
    sc2["scales"] = np.concatenate([sc["scales"], sc["scales"][:7]])
    sc2["rotations"] = np.concatenate([sc["rotations"], sc["rotations"][:7]])
    sc2["shs"] = np.concatenate([sc["shs"], sc["shs"][:7]])
    sc2["opacities"] = np.concatenate([sc["opacities"], np.zeros((7, 1))])
    np.testing.assert_allclose(RasterOracle(**sc2, dtype=np.float64).color, base, atol=1e-12)
    sc3 = dict(sc); sc3["means3D"] = sc["means3D"] + np.array([0, 0, 100.0])  # behind / far outside
    sc3["viewmatrix"] = sc["viewmatrix"].copy()
    o3 = RasterOracle(**{**sc, "means3D": sc["means3D"] * 0 + np.array([50.0, 50.0, 50.0])}, dtype=np.float64)
    assert (o3.radii == 0).all() and np.allclose(o3.color, 1.0)


def test_sh_matches_reference_python_formula():
    """utils/sh_utils.py:57-112 of the reference (in-tree python path, gaussian_renderer/__init__.py:107-111)."""
    from conftest import have_reference
    if not have_reference():
        pytest.skip("/root/reference not present")
    import sys
    sys.path.insert(0, "/root/reference")
    from utils.sh_utils import eval_sh
    rng = np.random.default_rng(0)
    sh = torch.tensor(rng.standard_normal((20, 16, 3)))
    d = torch.nn.functional.normalize(torch.tensor(rng.standard_normal((20, 3))), dim=1)
    for deg in range(4):
        ref = eval_sh(deg, sh.transpose(1, 2), d)
        np.testing.assert_allclose(raster_torch.sh_to_rgb(deg, sh, d).numpy(), ref.numpy(), rtol=1e-12, atol=1e-12)
