"""Name-compatible shim: `from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
(gaussian_renderer/__init__.py:14 of the reference) resolves to the MI355X-native implementation."""
import importlib as _il

_impl = _il.import_module("4dgaussians_amd.rasterizer")
GaussianRasterizationSettings = _impl.GaussianRasterizationSettings
GaussianRasterizer = _impl.GaussianRasterizer
rasterize_gaussians = _impl.rasterize_gaussians
__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
