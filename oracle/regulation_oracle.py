"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch, float32 or float64) of the reference's HexPlane regulariser.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows, line by line:
  scene/regulation.py:22-28           compute_plane_smoothness(t): second difference along dim 2 of [B,C,H,W], mean of squares
  scene/gaussian_model.py:538-549     _plane_regulation : sum of compute_plane_smoothness over planes [0,1,3] of every level
  scene/gaussian_model.py:550-561     _time_regulation  : same over planes [2,4,5]
  scene/gaussian_model.py:562-575     _l1_regulation    : sum of mean|1 - plane| over planes [2,4,5]
  scene/gaussian_model.py:576-577     compute_regulation: plane_tv_weight*_plane + time_smoothness_weight*_time + l1_time_planes_weight*_l1
Pinned by tests/test_oracle_regulation.py against the reference's own compute_plane_smoothness (imported from
/root/reference where present) and a committed golden vector (tests/golden/regulation_*.npz).
"""
import torch


def compute_plane_smoothness(t):
    h = t.shape[2]
    first = t[..., 1:, :] - t[..., :h - 1, :]
    second = first[..., 1:, :] - first[..., :h - 2, :]
    return torch.square(second).mean()


def compute_regulation(levels, time_smoothness_weight, l1_time_planes_weight, plane_tv_weight):
    """levels: list (per resolution level) of six tensors [1,C,H,W] in the reference's plane order."""
    plane = time = l1 = 0.0
    for grids in levels:
        if len(grids) == 3:
            continue
        for k in (0, 1, 3):
            plane = plane + compute_plane_smoothness(grids[k])
        for k in (2, 4, 5):
            time = time + compute_plane_smoothness(grids[k])
            l1 = l1 + torch.abs(1 - grids[k]).mean()
    return plane_tv_weight * plane + time_smoothness_weight * time + l1_time_planes_weight * l1


def import_reference_plane_smoothness():
    """The reference's own function (needs /root/reference; scene/__init__.py is bypassed like in deform_oracle)."""
    import sys
    import types
    REF = "/root/reference"
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if "scene" not in sys.modules or not hasattr(sys.modules["scene"], "__path__") or \
            REF + "/scene" not in list(sys.modules["scene"].__path__):
        pkg = types.ModuleType("scene")
        pkg.__path__ = [REF + "/scene"]
        sys.modules["scene"] = pkg
    from scene.regulation import compute_plane_smoothness as ref_fn
    return ref_fn
