"""GPU parity of distCUDA2 (csrc/knn.hip) against the exact CPU k-d tree restatement; import shim; edge cases."""
import importlib

import numpy as np
import pytest
import torch

from oracle import knn_oracle as KO
from scenes import rel_l2

pytestmark = pytest.mark.gpu
fdgs = importlib.import_module("4dgaussians_amd")


@pytest.mark.parametrize("n,seed", [(5000, 0), (1025, 1), (257, 2), (20000, 3)])
def test_matches_exact_knn(n, seed):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    pts = (rng.random((n, 3)) * 2.6 - 1.3).astype(np.float32)          # the reference's random init cloud
    if seed == 1:
        pts[10] = pts[500]; pts[11] = pts[500]                          # coincident points count with distance 0
    ref = KO.dist2_mean3(pts)
    from simple_knn._C import distCUDA2                                 # the reference's import line
    out = distCUDA2(torch.from_numpy(pts).float().to(dev)).cpu().numpy()
    assert out.shape == (n,)
    assert rel_l2(out, ref.astype(np.float32)) < 1e-5
    assert np.allclose(out, ref, rtol=2e-4, atol=1e-9)


def test_small_inputs_and_errors():
    dev = torch.device("cuda:0")
    assert fdgs.knn.distCUDA2(torch.zeros(0, 3, device=dev)).shape == (0,)
    out = fdgs.knn.distCUDA2(torch.tensor([[0., 0, 0], [1, 0, 0], [0, 2, 0]], device=dev)).cpu().numpy()
    big = np.finfo(np.float32).max
    assert np.allclose(out, [(1 + 4 + big) / 3, (1 + 5 + big) / 3, (4 + 5 + big) / 3], rtol=1e-6)   # unfilled slot stays FLT_MAX
    with pytest.raises(fdgs._lib.FdgsError):
        fdgs.knn.distCUDA2(torch.zeros(4, 3))
    with pytest.raises(ValueError):
        fdgs.knn.distCUDA2(torch.zeros(4, 2, device=dev))
