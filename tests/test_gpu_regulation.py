"""GPU parity of the fused HexPlane regulariser (csrc/regulation.hip through fdgs_plane_regulation) against the CPU
oracle (pinned to the reference's function): value and every plane gradient, golden vector, full-size grids of the
BASELINE configs, and the weight / shape edge cases."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import regulation_oracle as RO
from scenes import rel_l2

pytestmark = pytest.mark.gpu
synthetic = importlib.import_module("4dgaussians_amd.synthetic")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "regulation_dynerf.npz")


def _fdgs():
    return importlib.import_module("4dgaussians_amd")


def _net(cfg, seed):
    fd = _fdgs()
    torch.manual_seed(seed)
    net = fd.deform_network(synthetic.deform_args(cfg))
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if "grids" in name:
                p.add_(0.2 * torch.randn(p.shape, generator=gen))
    return net


@pytest.mark.parametrize("cfg", ["dynerf_default", "dnerf_bouncingballs", "hypernerf_default"])
@pytest.mark.parametrize("weights", [(0.01, 0.0001, 0.0001), (0.0, 0.5, 2.0), (1.0, 0.0, 0.0)])
def test_value_and_gradients_match_oracle_full_size(cfg, weights):
    dev = torch.device("cuda:0")
    fd = _fdgs()
    net = _net(cfg, 5)
    tsw, l1w, tvw = weights
    grids = net.deformation_net.grid.grids
    cpu_levels = [[p.detach().clone().double().requires_grad_(True) for p in level] for level in grids]
    ref = RO.compute_regulation(cpu_levels, tsw, l1w, tvw)
    flat = [p for level in cpu_levels for p in level]
    g_ref = torch.autograd.grad(ref * 3.0, flat, allow_unused=True)   # upstream gradient 3.0
    net = net.to(dev)
    loss = fd.compute_regulation(net, tsw, l1w, tvw)
    assert loss.dim() == 0 and loss.requires_grad
    assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-5 * abs(float(ref.detach())) + 1e-12
    (loss * 3.0).backward()
    gp = [p for level in net.deformation_net.grid.grids for p in level]
    for i, (p, b) in enumerate(zip(gp, g_ref)):
        if b is None or float(b.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, i
            continue
        assert rel_l2(p.grad.cpu().numpy(), b.float().numpy()) < 2e-6, i


def test_golden_vector_from_reference_function():
    dev = torch.device("cuda:0")
    L = _fdgs()._lib
    g = np.load(GOLD)
    tsw, l1w, tvw = [float(x) for x in g["weights"]]
    planes = [torch.tensor(g[f"plane{i}"]).to(dev).contiguous(memory_format=torch.channels_last) for i in range(12)]
    grads = [torch.zeros_like(p) for p in planes]     # channels_last preserved
    arr = (L.RegPlane * 12)()
    for i, (p, gr) in enumerate(zip(planes, grads)):
        assert gr.is_contiguous(memory_format=torch.channels_last)
        k = i % 6
        arr[i].plane, arr[i].grad_opt = p.data_ptr(), gr.data_ptr()
        arr[i].H, arr[i].W, arr[i].C = p.shape[2], p.shape[3], p.shape[1]
        arr[i].w_smooth = tvw if k in (0, 1, 3) else tsw
        arr[i].w_l1 = l1w if k in (2, 4, 5) else 0.0
    loss = torch.zeros(1, device=dev)
    L.check(L.lib().fdgs_plane_regulation(L.stream_ptr(), 12, arr, 1.0, None, L.ptr(loss)))
    torch.cuda.synchronize()
    assert abs(float(loss) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    for i, gr in enumerate(grads):
        assert rel_l2(gr.cpu().numpy(), g[f"grad{i}"]) < 2e-6, i


def test_edge_cases_bad_arguments_and_accumulation():
    dev = torch.device("cuda:0")
    L = _fdgs()._lib
    lib = L.lib()
    arr = (L.RegPlane * 1)()
    p = torch.rand(1, 8, 3, 5, device=dev).contiguous(memory_format=torch.channels_last)   # H = 3: a single second difference
    gr = torch.ones_like(p)
    arr[0].plane, arr[0].grad_opt, arr[0].H, arr[0].W, arr[0].C, arr[0].w_smooth, arr[0].w_l1 = p.data_ptr(), gr.data_ptr(), 3, 5, 8, 2.0, 0.0
    loss = torch.full((1,), 10.0, device=dev)
    L.check(lib.fdgs_plane_regulation(L.stream_ptr(), 1, arr, 0.5, None, L.ptr(loss)))
    pc = p.cpu().double().requires_grad_(True)
    ref = 2.0 * RO.compute_plane_smoothness(pc)
    gref, = torch.autograd.grad(ref, pc)
    assert abs(float(loss) - 10.0 - float(ref)) < 1e-5            # the value is accumulated into loss_acc
    assert rel_l2((gr.cpu() - 1.0).numpy(), 0.5 * gref.float().numpy()) < 2e-6   # gradients accumulate, scaled by grad_scale
    assert lib.fdgs_plane_regulation(L.stream_ptr(), 0, None, 1.0, None, None) == 0      # empty list is a no-op
    arr[0].C = 6
    assert lib.fdgs_plane_regulation(L.stream_ptr(), 1, arr, 1.0, None, L.ptr(loss)) != 0  # C % 4 != 0 is rejected loudly
    assert b"channels-last" in lib.fdgs_last_error()
    assert lib.fdgs_plane_regulation(L.stream_ptr(), 25, arr, 1.0, None, None) != 0        # more than 4 levels x 6 planes
