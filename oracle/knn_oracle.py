"""TEST INFRASTRUCTURE ONLY -- CPU restatement of simple_knn's distCUDA2: mean of the squared distances to the three
nearest OTHER points (exact, scipy cKDTree in float64).  "Parity unpinned": the submodule source (submodules/simple-knn,
.gitmodules:1-3) is not in the mount and the reference holds no test or fixture for it; the semantics are restated from
the call site scene/gaussian_model.py:148 and the published kernel (self excluded by index, three best squared distances
averaged).  Only tests/ may import this module."""
import numpy as np
from scipy.spatial import cKDTree


def dist2_mean3(points):
    pts = np.asarray(points, dtype=np.float64)
    n = pts.shape[0]
    k = min(4, n)
    d, idx = cKDTree(pts).query(pts, k=k)
    d = d.reshape(n, k) ** 2
    idx = idx.reshape(n, k)
    out = np.empty(n)
    for i in range(n):
        others = d[i][idx[i] != i] if (idx[i] == i).any() else d[i][1:]   # coincident points: drop exactly one zero (self)
        others = np.sort(others)[:3]
        vals = list(others) + [np.finfo(np.float32).max] * (3 - len(others))
        out[i] = sum(vals) / 3.0
    return out
