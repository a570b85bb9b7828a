"""TEST-ONLY stand-in for the two device entry points of 4dgaussians_amd/rasterizer.py, backed by the C rasterizer oracle, so that the
Python layer above them (GaussianRasterizer, the autograd node, render() -- ours AND the reference's own render() source) can be executed on a
machine without a GPU (BASELINE.json configs[0]: "PyTorch CPU rasterizer reference path (plumbing, no GPU)").  The product has no CPU path:
this module lives under tests/ and is installed by a pytest monkeypatch only."""
import contextlib
import math

import numpy as np
import torch

from oracle.raster_oracle import RasterOracle


class _State:
    pass


def rasterize_forward(settings, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, out=None, expect_backward=False):
    f = lambda t: None if (t is None or t.numel() == 0) else np.ascontiguousarray(t.detach().cpu().float().numpy())
    o = RasterOracle(means3D=f(means3D), opacities=f(opacities), viewmatrix=f(settings.viewmatrix), projmatrix=f(settings.projmatrix),
                     campos=f(settings.campos), bg=f(settings.bg), image_height=int(settings.image_height), image_width=int(settings.image_width),
                     tanfovx=float(settings.tanfovx), tanfovy=float(settings.tanfovy), sh_degree=int(settings.sh_degree), shs=f(shs),
                     colors_precomp=f(colors_precomp), scales=f(scales), rotations=f(rotations), cov3D_precomp=f(cov3D_precomp),
                     scale_modifier=float(settings.scale_modifier))
    st = _State()
    st.o = o
    st.params = type("P", (), {"H": o.H, "W": o.W, "P": o.P})()
    st.had_sh = shs is not None and shs.numel() > 0
    return torch.tensor(o.color), torch.tensor(o.radii), torch.tensor(o.depth), st


def rasterize_backward(state, grad_color, grad_depth=None):
    g = state.o.backward(grad_color.detach().cpu().numpy(), None if grad_depth is None else grad_depth.detach().cpu().numpy())
    t = lambda a: None if a is None else torch.tensor(a)
    return dict(means2D=t(g["means2D"]), means3D=t(g["means3D"]), opacities=t(g["opacities"]).reshape(-1, 1), colors=t(g["colors"]),
                cov3D=t(g["cov3D"]), shs=t(g["shs"]) if state.had_sh else None, scales=t(g["scales"]), rotations=t(g["rotations"]))


@contextlib.contextmanager
def installed(monkeypatch):
    """Route the shim's two device calls to the oracle and make `.cuda()` / `device="cuda"` no-ops (the reference's render() hard-codes them,
    gaussian_renderer/__init__.py:27,45-48)."""
    import importlib
    R = importlib.import_module("4dgaussians_amd.rasterizer")
    monkeypatch.setattr(R, "rasterize_forward", rasterize_forward)
    monkeypatch.setattr(R, "rasterize_backward", rasterize_backward)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    real_zeros_like = torch.zeros_like

    def zeros_like(x, *a, **k):
        if str(k.get("device", "")) == "cuda":
            k.pop("device")
        return real_zeros_like(x, *a, **k)

    monkeypatch.setattr(torch, "zeros_like", zeros_like)
    yield
