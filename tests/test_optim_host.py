"""Host-side (no GPU) checks of FusedAdam: it is a torch.optim.Optimizer with the reference's constructor call, the same
param_groups / state_dict structure as torch.optim.Adam (what densification and checkpoints rely on), and it refuses the
options the kernel does not implement."""
import importlib

import pytest
import torch

fdgs = importlib.import_module("4dgaussians_amd")


def _groups():
    a, b = torch.nn.Parameter(torch.rand(5, 3)), torch.nn.Parameter(torch.rand(7))
    return [{"params": [a], "lr": 1e-3, "name": "xyz"}, {"params": [b], "lr": 0.0, "name": "opacity"}], a, b


def test_constructor_and_structure_match_torch_adam():
    g1, a, b = _groups()
    opt = fdgs.FusedAdam(g1, lr=0.0, eps=1e-15)         # scene/gaussian_model.py:184
    ref = torch.optim.Adam(_groups()[0], lr=0.0, eps=1e-15)
    assert isinstance(opt, torch.optim.Optimizer)
    assert [g["name"] for g in opt.param_groups] == ["xyz", "opacity"]
    for go, gr in zip(opt.param_groups, ref.param_groups):
        for k in ("lr", "betas", "eps", "weight_decay", "amsgrad"):
            assert go[k] == gr[k], k
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and sd["state"] == {}
    # a torch.optim.Adam checkpoint loads (densification / resume path of the reference)
    a2 = _groups()
    ref2 = torch.optim.Adam(a2[0], lr=0.0, eps=1e-15)
    for p in (a2[1], a2[2]):
        p.grad = torch.ones_like(p)
    ref2.step()
    opt.load_state_dict(ref2.state_dict())
    st = opt.state[a]
    assert set(st) == {"step", "exp_avg", "exp_avg_sq"} and float(st["step"]) == 1.0


def test_unsupported_options_raise():
    g, _, _ = _groups()
    with pytest.raises(NotImplementedError):
        fdgs.FusedAdam(g, weight_decay=0.1)
    with pytest.raises(NotImplementedError):
        fdgs.FusedAdam(g, amsgrad=True)


def test_step_needs_the_gpu_library():
    g, a, b = _groups()
    opt = fdgs.FusedAdam(g, lr=0.0, eps=1e-15)
    a.grad = torch.ones_like(a)
    with pytest.raises(fdgs._lib.FdgsError):
        opt.step()                      # CPU parameters: no fallback


def test_dense_layout_detection():
    from importlib import import_module
    optim = import_module("4dgaussians_amd.optim")
    t = torch.rand(1, 8, 5, 6)
    assert optim._dense(t) and optim._dense(t.contiguous(memory_format=torch.channels_last)) and optim._dense(t.permute(0, 2, 3, 1))
    assert not optim._dense(t[..., ::2]) and not optim._dense(t[:, :, :3])
