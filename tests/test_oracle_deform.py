"""Pins oracle/deform_oracle.py against the reference's own modules (imported from /root/reference when present,
SURVEY.md Appendix E) and against committed golden vectors generated from them (tests/golden/)."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import have_reference
from oracle import deform_oracle as DO

synthetic = importlib.import_module("4dgaussians_amd.synthetic")
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _inputs(n, seed=1):
    g = synthetic.make_gaussians(n, seed=seed)
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1)
    return g["xyz"], g["scaling"], g["rotation"], g["opacity"], shs


@pytest.mark.skipif(not have_reference(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("cfg", ["dnerf_bouncingballs", "hypernerf_default", "dynerf_default"])
def test_matches_reference_modules_fwd_bwd(cfg):
    deform_network = DO.import_reference_deform_network()
    torch.manual_seed(6666)
    args = synthetic.deform_args(cfg)
    net = deform_network(args)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if "grids" in name:
                p.add_(0.1 * torch.randn_like(p))
    n = 257
    xyz, sc, rot, op, shs = [x.clone().requires_grad_(True) for x in _inputs(n)]
    net.deformation_net.set_aabb(xyz.max(0).values.tolist(), xyz.min(0).values.tolist())
    t = torch.rand(n, 1)
    ref = net(xyz, sc, rot, op, shs, t)
    sd = {k: v for k, v in net.named_parameters()}
    sd.update({k: v for k, v in net.named_buffers()})
    ora = DO.deform_forward(sd, args, xyz, sc, rot, op, shs, t)
    for a, b in zip(ref, ora):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    ws = [torch.randn_like(a) for a in ref]
    params = [p for p in net.parameters() if p.requires_grad]
    gr = torch.autograd.grad(sum((a * w).sum() for a, w in zip(ref, ws)), [xyz, sc, rot, op, shs] + params, allow_unused=True)
    go = torch.autograd.grad(sum((a * w).sum() for a, w in zip(ora, ws)), [xyz, sc, rot, op, shs] + params, allow_unused=True)
    for a, b in zip(gr, go):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), float((a - b).abs().max())


@pytest.mark.parametrize("cfg", ["dnerf_bouncingballs", "hypernerf_default", "dynerf_default"])
def test_matches_golden(cfg):
    z = np.load(os.path.join(GOLD, f"deform_{cfg}.npz"))
    sd = {k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("sd.")}
    args = synthetic.deform_args(cfg)
    ins = [torch.tensor(z["in." + k]) for k in ("xyz", "scales", "rot", "opacity", "shs", "t")]
    outs = DO.deform_forward(sd, args, *ins)
    for k, o in zip(("xyz", "scales", "rot", "opacity", "shs"), outs):
        assert torch.allclose(o, torch.tensor(z["out." + k]), rtol=1e-5, atol=1e-6), k
    # backward: the reference modules' gradients of sum(out * w) stored next to the outputs
    for v in sd.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    ins = [x.requires_grad_(True) for x in ins[:5]] + [ins[5]]
    outs = DO.deform_forward(sd, args, *ins)
    loss = sum((o * torch.tensor(z["w." + k])).sum() for k, o in zip(("xyz", "scales", "rot", "opacity", "shs"), outs))
    gkeys = [k for k in z.files if k.startswith("grad.")]
    wanted = [dict(zip(("in.xyz", "in.scales", "in.rot", "in.opacity", "in.shs"), ins))[k[5:]] if k.startswith("grad.in.") else sd[k[8:]]
              for k in gkeys]
    for k, g in zip(gkeys, torch.autograd.grad(loss, wanted, allow_unused=True)):
        assert g is not None, k
        assert torch.allclose(g, torch.tensor(z[k]), rtol=1e-4, atol=1e-5), (k, float((g - torch.tensor(z[k])).abs().max()))


def test_border_and_flip_semantics():
    # xyz = aabb max maps to -1 (flipped axis) and samples column 0; t is used un-normalised
    C = 4
    sd = {"deformation_net.grid.aabb": torch.tensor([[1.0, 1.0, 1.0], [-1.0, -1.0, -1.0]])}
    for k in range(6):
        sd[f"deformation_net.grid.grids.0.{k}"] = torch.ones(1, C, 5, 7)
    ramp = torch.arange(7.0).view(1, 1, 1, 7).expand(1, C, 5, 7).clone()
    sd["deformation_net.grid.grids.0.0"] = ramp  # plane (x,y): width indexes x
    f = DO.hexplane_features(sd, torch.tensor([[1.0, 0.0, 0.0], [-1.0, 0.0, 0.0], [5.0, 0.0, 0.0]]), torch.zeros(3, 1), 1)
    assert torch.allclose(f[:, 0], torch.tensor([0.0, 6.0, 0.0]))


def test_float64_backward_of_live_rows_matches_the_float32_autograd():
    """oracle.deform_oracle.backward_float64 (the gradient reference of the full-size parity checks) = the float32 autograd of the same
    oracle up to float32 rounding, with rows whose upstream gradients are all zero left out of the evaluation."""
    import importlib
    import numpy as np
    syn = importlib.import_module("4dgaussians_amd.synthetic")
    pc = syn.SynthModel(600, "dynerf_default", seed=5)
    n = 600
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "poc" not in k and "aabb" not in k)
          for k, v in pc._deformation.state_dict().items()}
    leaves = {k: getattr(pc, k).detach().clone().requires_grad_(True)
              for k in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")}
    shs = torch.cat([leaves["_features_dc"], leaves["_features_rest"]], 1)
    outs = DO.deform_forward(sd, pc._deformation.args, leaves["_xyz"], leaves["_scaling"], leaves["_rotation"], leaves["_opacity"], shs,
                             torch.full((n, 1), 0.37), activate=True)
    g = torch.Generator().manual_seed(9)
    mask = (torch.rand(n, generator=g) < 0.3).float()                 # 70 % of the rows get no gradient at all
    gouts = [torch.randn(o.shape, generator=g) * mask.reshape([-1] + [1] * (o.dim() - 1)) for o in outs]
    wanted = list(leaves.values()) + [v for v in sd.values() if v.requires_grad]
    names = list(leaves.keys()) + ["_deformation." + k for k, v in sd.items() if v.requires_grad]
    g32 = dict(zip(names, torch.autograd.grad(list(outs), wanted, grad_outputs=gouts, allow_unused=True)))
    g64 = DO.backward_float64(sd, pc._deformation.args, leaves, 0.37, gouts)
    assert set(g64) == set(g32)
    for k, a in g32.items():
        b = g64[k]
        if a is None:
            assert b is None or float(np.abs(b).max()) == 0.0, k
            continue
        a = a.double().numpy()
        assert b.shape == a.shape, k
        den = np.linalg.norm(b)
        assert np.linalg.norm(a - b) <= 2e-5 * den + 1e-12, (k, np.linalg.norm(a - b) / max(den, 1e-30))
    dead = mask == 0
    assert float(np.abs(g64["_xyz"][dead.numpy()]).max()) == 0.0
