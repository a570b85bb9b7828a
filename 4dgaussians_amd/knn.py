"""`distCUDA2` of simple-knn on the MI355X (csrc/knn.hip through fdgs_knn3_mean_dist2): for every point the mean of the
squared distances to its three nearest neighbours.  Call site in the reference: scene/gaussian_model.py:148
(`dist2 = torch.clamp_min(distCUDA2(points.float().cuda()), 0.0000001)`).  GPU only, no fallback."""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def distCUDA2(points):
    if points.device.type != "cuda":
        raise _lib.FdgsError("distCUDA2 runs on the GPU only (the reference passes a .cuda() tensor)")
    pts = points.detach().float().contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("points must be [N,3]")
    out = torch.empty(pts.shape[0], dtype=torch.float32, device=pts.device)
    check(_lib.lib().fdgs_knn3_mean_dist2(stream_ptr(), pts.shape[0], ptr(pts), ptr(out)))
    return out
