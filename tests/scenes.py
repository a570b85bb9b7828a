"""Shared seeded test scenes (numpy dicts ready for the oracle / the HIP path)."""
import importlib
import math

import numpy as np
import torch

synthetic = importlib.import_module("4dgaussians_amd.synthetic")


def raster_scene(n, width, height, seed=0, theta=30.0, sh_degree=3, scale_boost=1.0, extent=1.3, dtype=np.float32,
                 opacity_lo=0.05):
    """Activated Gaussians + camera, as the rasterizer receives them (gaussian_renderer/__init__.py:120-128)."""
    g = synthetic.make_gaussians(n, seed=seed, extent=extent)
    cam = synthetic.make_camera(width, height, theta_deg=theta, time=0.0)
    scales = torch.exp(g["scaling"]) * scale_boost
    rot = torch.nn.functional.normalize(g["rotation"])
    op = torch.sigmoid(g["opacity"]).clamp_min(opacity_lo)
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1)
    f = lambda t: np.ascontiguousarray(t.numpy().astype(dtype))
    return dict(means3D=f(g["xyz"]), scales=f(scales), rotations=f(rot), opacities=f(op), shs=f(shs),
                viewmatrix=f(cam.world_view_transform), projmatrix=f(cam.full_proj_transform), campos=f(cam.camera_center),
                bg=np.array([1.0, 1.0, 1.0], dtype), image_height=height, image_width=width,
                tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), sh_degree=sh_degree)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / d) if d > 0 else float(np.linalg.norm(a))


from oracle.chain import oracle_render_chain  # noqa: E402,F401  (the chain lives with the oracles: smoke() and bench.py use it too)
