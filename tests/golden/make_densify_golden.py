"""Generates tests/golden/densify_small.npz by running the REFERENCE's own GaussianModel.densify / prune / reset_opacity /
add_densification_stats (scene/gaussian_model.py) on CPU: `device="cuda"` allocations redirected, torch.normal fed the
recorded standard-normal samples (oracle/densify_oracle.py: reference_on_cpu).  Run in the build container:
    python tests/golden/make_densify_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import densify_oracle as DO  # noqa: E402

N, SEED, PD, EXTENT, MAX_GRAD = 257, 31, 0.01, 3.0, 0.0002
st = DO.random_state(N, SEED, sh_rest=3, extent=EXTENT, percent_dense=PD)
g = torch.Generator().manual_seed(99)
normals = torch.randn(2 * N, 3, generator=g)
vgrad = torch.randn(N, 3, generator=g) * 1e-3
radii = torch.randint(0, 60, (N,), generator=g, dtype=torch.int32)
vis = radii > 10
out = {"meta": np.array([N, SEED, PD, EXTENT, MAX_GRAD])}


def dump(prefix, s):
    for k, v in s.items():
        if isinstance(v, dict):
            for n, t in v.items():
                out[f"{prefix}.{k}.{n}"] = t.numpy()
        else:
            out[f"{prefix}.{k}"] = v.numpy()


dump("in", st)
out["normals"], out["vgrad"], out["radii"], out["vis"] = normals.numpy(), vgrad.numpy(), radii.numpy(), vis.numpy()
m = DO.reference_model_from_state(st, PD)
with DO.reference_on_cpu(normals):
    # train.py:261-262
    m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], radii[vis])
    m.add_densification_stats(vgrad, vis)
    dump("stats", DO.state_from_reference_model(m))
    m.densify(MAX_GRAD, 0.005, EXTENT, 20, 5, 5)
    dump("densified", DO.state_from_reference_model(m))
    m.max_radii2D = torch.rand(m.get_xyz.shape[0], generator=g) * 40
    out["radii_after"] = m.max_radii2D.numpy().copy()
    m.prune(MAX_GRAD, 0.05, EXTENT, 20)
    dump("pruned", DO.state_from_reference_model(m))
    m.reset_opacity()
    dump("reset", DO.state_from_reference_model(m))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "densify_small.npz"), **out)
print("wrote densify_small.npz:", N, "->", out["densified.param.xyz"].shape[0], "->", out["pruned.param.xyz"].shape[0])
