"""Partial pin of the (otherwise unpinned) rasterizer oracle: every piece of the rasterizer's arithmetic that the reference
ALSO holds in-tree as python is compared with the reference's own functions -- live where /root/reference exists, and through
the committed fixture tests/golden/raster_pins.npz (generator: tests/golden/make_raster_pins_golden.py) everywhere:
  cov3D from scale/rotation (utils/general_utils.py:63-116 + scene/gaussian_model.py:29-34), world-to-view / projection /
  camera centre (utils/graphics_utils.py:38-71, scene/cameras.py:59-64), SH -> colour (utils/sh_utils.py:57-112 as render()
  uses it, gaussian_renderer/__init__.py:106-111).
What stays unpinned: EWA projection, radius/tile rule, blending and their backward (CUDA-only in the reference)."""
import importlib
import math
import os

import numpy as np
import pytest
import torch

from conftest import have_reference
from oracle import raster_torch
from oracle.raster_oracle import RasterOracle

synthetic = importlib.import_module("4dgaussians_amd.synthetic")
sh_mod = importlib.import_module("4dgaussians_amd.sh")
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "raster_pins.npz"))


def _cov_scene(dtype, mod):
    g = synthetic.make_gaussians(512, seed=31)
    cam = synthetic.make_camera(400, 400, theta_deg=30.0)
    f = lambda t: np.ascontiguousarray(t.numpy().astype(dtype))
    q = torch.nn.functional.normalize(g["rotation"])       # render() hands the rasterizer normalised quaternions
    return dict(means3D=f(g["xyz"]), scales=np.ascontiguousarray(Z["cov.scales"].astype(dtype)), rotations=f(q),
                opacities=f(torch.sigmoid(g["opacity"])), shs=f(torch.cat([g["features_dc"], g["features_rest"]], 1)),
                viewmatrix=f(cam.world_view_transform), projmatrix=f(cam.full_proj_transform), campos=f(cam.camera_center),
                bg=np.zeros(3, dtype), image_height=400, image_width=400, tanfovx=math.tan(cam.FoVx * 0.5),
                tanfovy=math.tan(cam.FoVy * 0.5), sh_degree=3, scale_modifier=mod, dtype=dtype)


@pytest.mark.parametrize("mod", [1.0, 0.7])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 2e-6), (np.float32, 5e-6)])
def test_c_oracle_cov3d_matches_reference_python(dtype, tol, mod):
    o = RasterOracle(**_cov_scene(dtype, mod))
    vis = o.radii > 0
    assert vis.sum() > 300
    ref = Z[f"cov3D.mod{mod}"]                      # float32 result of the reference functions
    got = o.field("cov3D")
    scale = np.abs(ref[vis]).max(axis=1, keepdims=True)
    assert np.abs(got[vis] - ref[vis]).max() / 1.0 < tol * max(1.0, float(scale.max())) and \
        (np.abs(got[vis] - ref[vis]) / scale).max() < 50 * tol


def test_torch_restatement_rotation_matches_reference_cov3d():
    q = torch.nn.functional.normalize(torch.tensor(Z["cov.rotations_raw"], dtype=torch.float64))
    s = torch.tensor(Z["cov.scales"], dtype=torch.float64)
    R = raster_torch.quat_to_rot(q)
    L = R * s[:, None, :]
    S = L @ L.transpose(1, 2)
    cov = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1).numpy()
    np.testing.assert_allclose(cov, Z["cov3D.mod1.0"], rtol=2e-5, atol=1e-8)


def test_synth_model_get_covariance_matches_reference():
    """SynthModel.get_covariance (what render() calls with pipe.compute_cov3D_python) == the reference's covariance_activation."""
    pc = synthetic.SynthModel(512, "dynerf_default", seed=31)
    with torch.no_grad():
        pc._scaling.copy_(torch.log(torch.tensor(Z["cov.scales"])))
        pc._rotation.copy_(torch.tensor(Z["cov.rotations_raw"]))
    for mod in (1.0, 0.7):
        np.testing.assert_allclose(pc.get_covariance(mod).detach().numpy(), Z[f"cov3D.mod{mod}"], rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize("i", range(4))
def test_camera_matrices_match_reference(i):
    W, H, th, ph, rad = Z[f"cam{i}.pose"]
    cam = synthetic.make_camera(int(W), int(H), theta_deg=float(th), phi_deg=float(ph), radius=float(rad))
    np.testing.assert_allclose(synthetic.world_to_view(Z[f"cam{i}.R"], Z[f"cam{i}.T"]), Z[f"cam{i}.w2v"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(synthetic.projection_matrix(0.01, 100.0, cam.FoVx, cam.FoVy), Z[f"cam{i}.proj"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(cam.world_view_transform.numpy(), Z[f"cam{i}.world_view_transform"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(cam.full_proj_transform.numpy(), Z[f"cam{i}.full_proj_transform"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(cam.camera_center.numpy(), Z[f"cam{i}.camera_center"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("deg", range(4))
def test_sh_colours_match_reference(deg):
    shs, xyz, campos = torch.tensor(Z["sh.shs"]), torch.tensor(Z["sh.xyz"]), torch.tensor(Z["sh.campos"])
    dirs = xyz - campos
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    ours = torch.clamp_min(sh_mod.eval_sh(deg, shs.transpose(1, 2).reshape(-1, 3, 16), dirs) + 0.5, 0.0)
    np.testing.assert_allclose(ours.numpy(), Z[f"sh.colors.deg{deg}"], rtol=1e-5, atol=1e-6)
    # the C oracle's SH -> RGB (its `rgb` field) on the same Gaussians: visible ones only
    sc = _cov_scene(np.float32, 1.0)
    sc["sh_degree"] = deg
    o = RasterOracle(**sc)
    vis = o.radii > 0
    np.testing.assert_allclose(o.field("rgb")[vis], Z[f"sh.colors.deg{deg}"][vis], rtol=0, atol=2e-6)


@pytest.mark.skipif(not have_reference(), reason="/root/reference not present (GPU box)")
def test_fixture_is_what_the_reference_computes_now():
    """Live re-evaluation of the reference functions (guards the committed fixture against drift)."""
    from oracle import densify_oracle as DN
    DN.import_reference_gaussian_model()
    import utils.general_utils as gu
    from utils.graphics_utils import getProjectionMatrix, getWorld2View2
    s, q = torch.tensor(Z["cov.scales"]), torch.tensor(Z["cov.rotations_raw"])
    with DN.reference_on_cpu():
        L = gu.build_scaling_rotation(0.7 * s, q)
        cov = gu.strip_symmetric(L @ L.transpose(1, 2))
    np.testing.assert_array_equal(cov.numpy(), Z["cov3D.mod0.7"])
    np.testing.assert_array_equal(getWorld2View2(Z["cam1.R"], Z["cam1.T"]), Z["cam1.w2v"])
    cam = synthetic.make_camera(1352, 1014, theta_deg=-60.0)
    np.testing.assert_array_equal(getProjectionMatrix(0.01, 100.0, cam.FoVx, cam.FoVy).numpy(), Z["cam1.proj"])
