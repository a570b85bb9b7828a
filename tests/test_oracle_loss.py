"""CPU: the image-loss oracle (oracle/loss_oracle.py) against the committed golden vector generated from the reference's own
utils/loss_utils.py (tests/golden/make_loss_golden.py) and, where /root/reference is present, against that code directly."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle as LO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_ssim.npz")


def test_oracle_matches_golden_vector():
    g = np.load(GOLD)
    img = torch.from_numpy(g["img"]).requires_grad_(True)
    gt = torch.from_numpy(g["gt"])
    lam = float(g["lambda_dssim"])
    l1, s = LO.l1_loss(img, gt), LO.ssim(img, gt)
    loss = l1 + lam * (1 - s)
    (grad,) = torch.autograd.grad(loss, img)
    assert abs(l1.item() - float(g["l1"])) < 1e-7
    assert abs(s.item() - float(g["ssim"])) < 2e-6          # separable row/column filter vs the reference's 2-D window
    assert abs(loss.item() - float(g["loss"])) < 2e-6
    rel = (grad - torch.from_numpy(g["grad"])).norm() / torch.from_numpy(g["grad"]).norm()
    assert rel < 1e-5, rel
    items = LO.ssim(img.detach(), gt, size_average=False)
    assert np.allclose(items.numpy(), g["ssim_items"], atol=2e-6)


def test_oracle_float64_agrees_with_float32():
    g = np.load(GOLD)
    a32, b32 = torch.from_numpy(g["img"]), torch.from_numpy(g["gt"])
    assert abs(LO.ssim(a32, b32).item() - LO.ssim(a32.double(), b32.double()).item()) < 5e-6
    assert abs(LO.ssim(b32.double(), b32.double()).item() - 1.0) < 1e-12   # identical images


@pytest.mark.skipif(not os.path.exists("/root/reference/utils/loss_utils.py"), reason="reference tree not present")
@pytest.mark.parametrize("shape", [(1, 3, 11, 11), (2, 3, 40, 64), (1, 1, 5, 70)])
def test_oracle_matches_reference_functions(shape):
    ref = LO.import_reference_loss_utils()
    gen = torch.Generator().manual_seed(sum(shape))
    a = torch.rand(*shape, generator=gen).requires_grad_(True)
    b = torch.rand(*shape, generator=gen)
    for avg in (True, False):
        r, o = ref.ssim(a, b, size_average=avg), LO.ssim(a, b, size_average=avg)
        assert torch.allclose(r, o, atol=3e-6), (r, o)
    gr, = torch.autograd.grad(ref.ssim(a, b), a)
    go, = torch.autograd.grad(LO.ssim(a, b), a)
    assert (gr - go).norm() / gr.norm() < 1e-5
    assert abs(ref.l1_loss(a, b).item() - LO.l1_loss(a, b).item()) < 1e-7
