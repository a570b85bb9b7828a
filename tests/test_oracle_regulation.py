"""Pins oracle/regulation_oracle.py: (i) its compute_plane_smoothness against the reference's own function imported from
/root/reference (where present), (ii) the combined regulariser value and every plane gradient against the golden vector
generated from the reference's function (tests/golden/make_regulation_golden.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import have_reference
from oracle import regulation_oracle as RO

GOLD = os.path.join(os.path.dirname(__file__), "golden", "regulation_dynerf.npz")


@pytest.mark.skipif(not have_reference(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("shape", [(1, 16, 75, 64), (1, 32, 3, 5), (2, 4, 10, 7)])
def test_plane_smoothness_matches_reference_function(shape):
    ref = RO.import_reference_plane_smoothness()
    t = torch.rand(*shape, generator=torch.Generator().manual_seed(3)).requires_grad_(True)
    a, b = ref(t), RO.compute_plane_smoothness(t)
    assert torch.equal(a, b)
    ga, = torch.autograd.grad(a, t)
    gb, = torch.autograd.grad(b, t)
    assert torch.equal(ga, gb)


def test_regulariser_matches_golden_value_and_gradients():
    g = np.load(GOLD)
    planes = [torch.tensor(g[f"plane{i}"]).requires_grad_(True) for i in range(12)]
    tsw, l1w, tvw = [float(x) for x in g["weights"]]
    loss = RO.compute_regulation([planes[:6], planes[6:]], tsw, l1w, tvw)
    assert abs(float(loss) - float(g["loss"])) <= 1e-7 * abs(float(g["loss"]))
    grads = torch.autograd.grad(loss, planes)
    for i, gr in enumerate(grads):
        assert np.allclose(gr.numpy(), g[f"grad{i}"], rtol=1e-6, atol=1e-12), i


def test_three_plane_grids_are_skipped_like_the_reference():
    lv = [[torch.rand(1, 4, 5, 6) for _ in range(3)]]
    assert float(RO.compute_regulation(lv, 1.0, 1.0, 1.0)) == 0.0
