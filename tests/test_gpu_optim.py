"""GPU parity of FusedAdam (csrc/adam.hip through fdgs_adam_step) against torch.optim.Adam on the CPU -- which IS the
reference's optimizer (scene/gaussian_model.py:184) -- over several steps with per-group learning rates rewritten every
step (update_learning_rate), odd tensor sizes, channels_last HexPlane parameters, a parameter without gradient, the
densification-style state surgery and a state_dict round trip."""
import importlib

import pytest
import torch

from scenes import rel_l2

pytestmark = pytest.mark.gpu
fdgs = importlib.import_module("4dgaussians_amd")


def _make(dev, seed=0):
    gen = torch.Generator().manual_seed(seed)
    shapes = {"xyz": (1001, 3), "f_dc": (1001, 1, 3), "f_rest": (1001, 15, 3), "opacity": (1001, 1), "scaling": (1001, 3),
              "rotation": (1001, 4), "bias": (7,), "one": (1,), "plane": (1, 16, 9, 11), "unused": (5,)}
    cpu = {k: torch.nn.Parameter(torch.randn(*s, generator=gen)) for k, s in shapes.items()}
    gpu = {k: torch.nn.Parameter(v.detach().clone().to(dev)) for k, v in cpu.items()}
    with torch.no_grad():
        gpu["plane"] = torch.nn.Parameter(gpu["plane"].detach().contiguous(memory_format=torch.channels_last))
    groups = lambda d: [{"params": [d[k]], "lr": 1e-3 * (i + 1), "name": k} for i, k in enumerate(shapes)]
    return cpu, gpu, groups, gen


def _set_grads(cpu, gpu, gen, scale=1.0):
    for k in cpu:
        if k == "unused":
            continue
        g = torch.randn(cpu[k].shape, generator=gen) * scale
        cpu[k].grad = g.clone()
        gpu[k].grad = g.to(gpu[k].device)


def test_matches_torch_adam_over_steps_with_lr_schedule():
    dev = torch.device("cuda:0")
    cpu, gpu, groups, gen = _make(dev)
    ref = torch.optim.Adam(groups(cpu), lr=0.0, eps=1e-15)
    opt = fdgs.FusedAdam(groups(gpu), lr=0.0, eps=1e-15)
    for it in range(6):
        _set_grads(cpu, gpu, gen, scale=10.0 ** (it - 3))
        for gi, (gr, go) in enumerate(zip(ref.param_groups, opt.param_groups)):   # update_learning_rate rewrites "lr" in place
            gr["lr"] = go["lr"] = 1e-3 / (1 + it) * (1 + gi)
        ref.step(); opt.step()
    torch.cuda.synchronize()
    for k in cpu:
        assert rel_l2(gpu[k].detach().cpu().numpy(), cpu[k].detach().numpy()) < 1e-6, k
        if k == "unused":
            assert len(opt.state[gpu[k]]) == 0
            continue
        so, sr = opt.state[gpu[k]], ref.state[cpu[k]]
        assert float(so["step"]) == float(sr["step"]) == 6.0
        assert rel_l2(so["exp_avg"].cpu().numpy(), sr["exp_avg"].numpy()) < 1e-6, k
        assert rel_l2(so["exp_avg_sq"].cpu().numpy(), sr["exp_avg_sq"].numpy()) < 1e-6, k
    assert gpu["plane"].is_contiguous(memory_format=torch.channels_last)


def test_state_surgery_and_state_dict_round_trip():
    """What densification does (scene/gaussian_model.py:331-347 _prune_optimizer): new Parameter, sliced state tensors,
    state re-keyed; then a checkpoint round trip through state_dict()."""
    dev = torch.device("cuda:0")
    cpu, gpu, groups, gen = _make(dev, seed=1)
    ref = torch.optim.Adam(groups(cpu), lr=0.0, eps=1e-15)
    opt = fdgs.FusedAdam(groups(gpu), lr=0.0, eps=1e-15)
    _set_grads(cpu, gpu, gen)
    ref.step(); opt.step()
    mask = torch.rand(1001, generator=gen) > 0.3
    for o, params in ((ref, cpu), (opt, gpu)):
        for group in o.param_groups:
            if group["name"] not in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
                continue
            old = group["params"][0]
            m = mask.to(old.device)
            st = o.state.get(old, None)
            st["exp_avg"] = st["exp_avg"][m]
            st["exp_avg_sq"] = st["exp_avg_sq"][m]
            del o.state[old]
            group["params"][0] = torch.nn.Parameter(old[m].detach().requires_grad_(True))
            o.state[group["params"][0]] = st
            params[group["name"]] = group["params"][0]
    sd = opt.state_dict()
    opt2 = fdgs.FusedAdam([{"params": [gpu[g["name"]]], "lr": g["lr"], "name": g["name"]} for g in opt.param_groups], lr=0.0, eps=1e-15)
    opt2.load_state_dict(sd)
    for it in range(2):
        _set_grads(cpu, gpu, gen)
        ref.step(); opt2.step()
    torch.cuda.synchronize()
    for k in cpu:
        assert gpu[k].shape == cpu[k].shape
        assert rel_l2(gpu[k].detach().cpu().numpy(), cpu[k].detach().numpy()) < 1e-6, k
    assert float(opt2.state[gpu["xyz"]]["step"]) == 3.0


def test_abi_rejects_bad_descriptors():
    L = fdgs._lib
    lib = L.lib()
    dev = torch.device("cuda:0")
    p = torch.zeros(8, device=dev)
    arr = (L.AdamTensor * 1)()
    arr[0].param, arr[0].grad, arr[0].exp_avg, arr[0].exp_avg_sq = p.data_ptr(), p.data_ptr(), p.data_ptr(), p.data_ptr()
    arr[0].n, arr[0].lr, arr[0].step = 8, 0.1, 0
    assert lib.fdgs_adam_step(L.stream_ptr(), 1, arr, 0.9, 0.999, 1e-15) != 0 and b"step" in lib.fdgs_last_error()
    arr[0].step, arr[0].grad = 1, None
    assert lib.fdgs_adam_step(L.stream_ptr(), 1, arr, 0.9, 0.999, 1e-15) != 0
    assert lib.fdgs_adam_step(L.stream_ptr(), 0, None, 0.9, 0.999, 1e-15) == 0
