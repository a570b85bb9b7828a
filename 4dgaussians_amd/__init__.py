"""MI355X-native 4D-Gaussian render path (hand-written HIP for gfx950 behind the reference's Python API).

The directory name starts with a digit, so import it by string:
    fdgs = importlib.import_module("4dgaussians_amd")          (or `import fdgs`, the root-level alias)
Sub-modules: rasterizer (drop-in for `diff_gaussian_rasterization`), deformation (drop-in `deform_network`),
renderer (`render()` of gaussian_renderer/__init__.py:18), regulation (`compute_regulation`, the HexPlane regulariser), losses (`l1_loss`, `ssim`, `psnr`, fused `image_loss`), densify (densification statistics, clone / split / prune with optimizer-state surgery), optim (`FusedAdam`, the multi-tensor optimizer step), knn (`distCUDA2` of simple-knn), io (PLY point clouds and deformation checkpoints in the reference's formats), parallel (frame-parallel driver), synthetic (test scenes).
The HIP library is loaded lazily on first use and there is no CPU fallback.
"""
from . import _lib, deformation, densify, io, knn, losses, optim, parallel, rasterizer, regulation, renderer, sh, synthetic  # noqa: F401
from .deformation import deform_network  # noqa: F401
from .renderer import render, render_views  # noqa: F401
from .regulation import compute_regulation  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "deform_network", "render", "render_views", "compute_regulation", "FusedAdam", "rasterizer", "regulation", "optim", "losses", "densify",
           "deformation", "renderer", "synthetic", "_lib"]
