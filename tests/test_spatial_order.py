"""fdgs.densify.spatial_reorder: host-side checks (CPU) -- the curve keys, the permutation's consistency over every
per-Gaussian array and the optimizer state, chunk compactness (what the plane-gradient kernel's texel windows rely on)."""
import importlib
import types

import numpy as np
import pytest
import torch

fdgs = importlib.import_module("4dgaussians_amd")
D = fdgs.densify


def test_hilbert_curve_is_a_bijection_with_unit_steps():
    """Every cell of an 8^3 grid gets a distinct key, and cells that are consecutive along the curve are face neighbours."""
    b = 3
    g = torch.stack(torch.meshgrid(*[torch.arange(2 ** b)] * 3, indexing="ij"), -1).reshape(-1, 3).float() + 0.5
    k = D.hilbert_keys(g, lo=[0, 0, 0], hi=[2 ** b] * 3, bits=b)
    assert torch.unique(k).numel() == g.shape[0]
    # keys use the 10-bit spread layout: ranks are what matters
    order = torch.argsort(k)
    step = (g[order][1:] - g[order][:-1]).abs().sum(1)
    assert torch.all(step == 1.0)


def test_morton_keys_sorted_axis_bits():
    pts = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]]) * 0.999 + 0.0005
    k = D.morton_keys(pts, lo=[0, 0, 0], hi=[1, 1, 1], bits=1)
    assert k.tolist() == [0, 1, 2, 4, 7]


@pytest.mark.parametrize("curve", ["hilbert", "morton"])
def test_reorder_permutes_everything_consistently(curve):
    syn = fdgs.synthetic
    n = 3000
    pc = syn.SynthModel(n, "dynerf_default", seed=5)
    tag = torch.arange(n, dtype=torch.float32)
    with torch.no_grad():
        pc._opacity[:, 0] = tag                     # a recognisable row id carried by a parameter
    opt = torch.optim.Adam(pc.optimizer_groups(lr=0.0), lr=0.0, eps=1e-15)
    for grp in opt.param_groups:                     # give every per-Gaussian group Adam moments that carry the row id
        if grp["name"] in D.GROUPS:
            q = grp["params"][0]
            opt.state[q] = {"step": torch.tensor(3.0), "exp_avg": tag.reshape(-1, *[1] * (q.dim() - 1)).expand_as(q).clone(),
                            "exp_avg_sq": 2 * tag.reshape(-1, *[1] * (q.dim() - 1)).expand_as(q).clone()}
    pc.optimizer = opt
    pc.xyz_gradient_accum, pc.denom = tag.reshape(n, 1).clone(), tag.reshape(n, 1).clone() + 1
    pc.max_radii2D, pc._deformation_accum = tag.clone() * 3, tag.reshape(n, 1).expand(n, 3).clone()
    pc._deformation_table = (tag % 2 == 0)
    before = {k: getattr(pc, a).detach().clone() for k, a in D.ATTR.items()}
    perm = D.spatial_reorder(pc, curve=curve)
    assert sorted(perm.tolist()) == list(range(n))
    ids = pc._opacity.detach()[:, 0]
    assert torch.equal(ids, tag[perm])
    for k, a in D.ATTR.items():
        now = getattr(pc, a)
        assert isinstance(now, torch.nn.Parameter) and now.requires_grad
        if k != "opacity":
            assert torch.equal(now.detach(), before[k][perm]), k
        grp = [g_ for g_ in opt.param_groups if g_["name"] == k][0]
        assert grp["params"][0] is now                                   # the optimizer steps the new Parameter
        st = opt.state[now]
        assert float(st["step"]) == 3.0
        assert torch.equal(st["exp_avg"].reshape(n, -1)[:, 0], ids) and torch.equal(st["exp_avg_sq"].reshape(n, -1)[:, 0], 2 * ids)
    assert len(opt.state) == 6
    assert torch.equal(pc.xyz_gradient_accum[:, 0], ids) and torch.equal(pc.denom[:, 0], ids + 1)
    assert torch.equal(pc.max_radii2D, ids * 3) and torch.equal(pc._deformation_accum[:, 2], ids)
    assert torch.equal(pc._deformation_table, ids % 2 == 0)
    # neighbours in the array are now neighbours in space
    x = pc._xyz.detach()
    assert float((x[1:] - x[:-1]).norm(dim=1).mean()) < 0.25 * float((before["xyz"][1:] - before["xyz"][:-1]).norm(dim=1).mean())


def test_hilbert_chunks_fit_the_texel_window():
    """The property csrc/deform.hip's D4 relies on for speed (never for correctness): at BASELINE density, most Gaussians of a
    128-chunk lie within 15 texels of the chunk's minimum on the 128-texel planes."""
    syn = fdgs.synthetic
    g = syn.make_gaussians(100_000, seed=1)
    pc = types.SimpleNamespace(_xyz=torch.nn.Parameter(g["xyz"]), _features_dc=torch.nn.Parameter(g["features_dc"]),
                               _features_rest=torch.nn.Parameter(g["features_rest"]), _opacity=torch.nn.Parameter(g["opacity"]),
                               _scaling=torch.nn.Parameter(g["scaling"]), _rotation=torch.nn.Parameter(g["rotation"]))
    D.spatial_reorder(pc)
    x = pc._xyz.detach()
    lo, hi = x.min(0).values, x.max(0).values
    pix = ((x - lo) / (hi - lo) * 63).floor().long()[: 100_000 // 128 * 128].reshape(-1, 128, 3)
    inside = ((pix - pix.min(1, keepdim=True).values) <= 14).all(2).float().mean()
    assert float(inside) > 0.97


def test_spatial_order_hint_is_measured_from_the_positions():
    """deformation.spatial_order_hint (host side, any device): random order -> False, curve order -> True, cached per tensor object."""
    syn = fdgs.synthetic
    x = syn.make_gaussians(20_000, seed=4)["xyz"]
    hint = fdgs.deformation.spatial_order_hint
    assert hint(x) is False
    xs = x[torch.argsort(D.hilbert_keys(x))].contiguous()
    assert hint(xs) is True and hint(xs) is True
    xm = x[torch.argsort(D.morton_keys(x))].contiguous()
    assert hint(xm) is True
    assert hint(xs[:50].contiguous()) is False          # too few rows to tell: the conservative answer
    # four concatenated curve-ordered runs (what densify leaves behind: kept originals, clones, first and second children) still count
    # as ordered: only the three run boundaries jump
    runs = torch.cat([xs[0::4], xs[1::4], xs[2::4], xs[3::4]])
    assert hint(runs.contiguous()) is True
