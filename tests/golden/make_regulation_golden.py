"""Generates tests/golden/regulation_dynerf.npz from the REFERENCE's own compute_plane_smoothness (scene/regulation.py:22-28)
combined exactly as GaussianModel.compute_regulation does (scene/gaussian_model.py:538-577).  Run in the build container
(needs /root/reference):  python tests/golden/make_regulation_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import regulation_oracle as RO  # noqa: E402

ref_smooth = RO.import_reference_plane_smoothness()
gen = torch.Generator().manual_seed(20240)
C, res, tres = 16, (8, 12, 10), 9     # small 4-D grid, two levels (x1, x2 spatial multiplier), planes in reference order
levels = []
for mult in (1, 2):
    r = [res[0] * mult, res[1] * mult, res[2] * mult, tres]
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    levels.append([(torch.rand(1, C, r[j], r[i], generator=gen) * 0.8 + 0.3).requires_grad_(True) for (i, j) in pairs])
tsw, l1w, tvw = 0.01, 0.0001, 0.0002
plane = sum(ref_smooth(g[k]) for g in levels for k in (0, 1, 3))
time = sum(ref_smooth(g[k]) for g in levels for k in (2, 4, 5))
l1 = sum(torch.abs(1 - g[k]).mean() for g in levels for k in (2, 4, 5))
loss = tvw * plane + tsw * time + l1w * l1
grads = torch.autograd.grad(loss, [p for g in levels for p in g])
out = {"loss": np.float64(loss.item()), "weights": np.array([tsw, l1w, tvw])}
for i, (p, g) in enumerate(zip([p for gr in levels for p in gr], grads)):
    out[f"plane{i}"] = p.detach().numpy()
    out[f"grad{i}"] = g.numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "regulation_dynerf.npz"), **out)
print("wrote regulation_dynerf.npz, loss =", loss.item())
