"""CPU: the Python surface mirrors the reference's by name -- same parameter names, order and defaults as the functions /
methods it replaces (checked against the reference sources where /root/reference is present, against the recorded
parameter lists everywhere)."""
import importlib
import inspect
import os

import pytest

fdgs = importlib.import_module("4dgaussians_amd")
REF = "/root/reference"

# (our callable, recorded parameter list of the reference callable, reference "module:qualname" or None)
def _cases():
    L, D, R = fdgs.losses, fdgs.densify, fdgs.rasterizer
    return [
        (L.l1_loss, ["network_output", "gt"], "utils.loss_utils:l1_loss"),
        (L.l2_loss, ["network_output", "gt"], "utils.loss_utils:l2_loss"),
        (L.ssim, ["img1", "img2", "window_size=11", "size_average=True"], "utils.loss_utils:ssim"),
        (L.psnr, ["img1", "img2", "mask=None"], "utils.image_utils:psnr"),
        (L.mse, ["img1", "img2"], "utils.image_utils:mse"),
        (fdgs.render, ["viewpoint_camera", "pc", "pipe", "bg_color", "scaling_modifier=1.0", "override_color=None", "stage='fine'", "cam_type=None"],
         None),   # gaussian_renderer/__init__.py:18 (its module imports the CUDA wheel: recorded list only)
        # GaussianModel methods: ours take the model as the first argument instead of self
        (D.add_densification_stats, ["pc", "viewspace_point_tensor", "update_filter", "radii=None"], None),
        # (trailing `reorder=None`: our optional extra -- back onto the Hilbert curve after the set was rebuilt)
        (D.prune_points, ["pc", "mask", "reorder=None"], None),
        (D.prune, ["pc", "max_grad", "min_opacity", "extent", "max_screen_size", "reorder=None"], None),
        (D.reset_opacity, ["pc"], None),
        (fdgs.compute_regulation, ["pc_or_net", "time_smoothness_weight", "l1_time_planes_weight", "plane_tv_weight"], None),
        (fdgs.io.save_ply, ["pc", "path"], None), (fdgs.io.save_deformation, ["pc", "path"], None),
    ]


def _sig(fn):
    out = []
    for p in inspect.signature(fn).parameters.values():
        out.append(p.name if p.default is inspect.Parameter.empty else f"{p.name}={p.default!r}")
    return out


@pytest.mark.parametrize("ours,recorded,ref", _cases(), ids=lambda v: getattr(v, "__name__", None))
def test_signature_matches_recorded_reference_signature(ours, recorded, ref):
    assert _sig(ours) == recorded


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_recorded_signatures_match_the_reference_sources():
    from oracle import densify_oracle, loss_oracle
    lu = loss_oracle.import_reference_loss_utils()
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_image_utils", os.path.join(REF, "utils", "image_utils.py"))
    iu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(iu)
    L, D = fdgs.losses, fdgs.densify
    for ours, theirs in ((L.l1_loss, lu.l1_loss), (L.l2_loss, lu.l2_loss), (L.ssim, lu.ssim), (L.psnr, iu.psnr), (L.mse, iu.mse)):
        assert _sig(ours) == _sig(theirs), ours.__name__
    GM = densify_oracle.import_reference_gaussian_model()
    # method(self, ...) <-> function(pc, ...): same names after the first; our extra trailing keyword arguments are optional
    for ours, theirs in ((D.add_densification_stats, GM.add_densification_stats), (D.prune_points, GM.prune_points), (D.prune, GM.prune),
                         (D.reset_opacity, GM.reset_opacity), (D.densify, GM.densify), (fdgs.io.save_ply, GM.save_ply),
                         (fdgs.io.load_ply, GM.load_ply), (fdgs.io.save_deformation, GM.save_deformation), (fdgs.io.load_model, GM.load_model)):
        a, b = _sig(ours)[1:], _sig(theirs)[1:]
        # same names in the same order; a default of the reference is our default too (ours may add defaults, never drop one)
        assert [x.split("=")[0] for x in a[:len(b)]] == [x.split("=")[0] for x in b], (ours.__name__, a, b)
        assert all(x == y for x, y in zip(a, b) if "=" in y), (ours.__name__, a, b)
        assert all("=" in extra for extra in a[len(b):]), (ours.__name__, a[len(b):])
    assert _sig(fdgs.compute_regulation)[1:] == _sig(GM.compute_regulation)[1:]
    # the settings tuple and the rasterizer call of the un-vendored CUDA wheel, as the reference uses them by keyword
    # (gaussian_renderer/__init__.py:38-58, 120-128)
    assert list(fdgs.GaussianRasterizationSettings._fields) == ["image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                                                               "projmatrix", "sh_degree", "campos", "prefiltered", "debug"]
    assert _sig(fdgs.GaussianRasterizer.forward)[1:] == ["means3D", "means2D", "opacities", "shs=None", "colors_precomp=None", "scales=None",
                                                        "rotations=None", "cov3D_precomp=None"]
