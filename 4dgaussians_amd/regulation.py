"""Drop-in for `GaussianModel.compute_regulation` of the reference (scene/gaussian_model.py:538-577; evaluated at
train.py:208-211 every fine iteration) as ONE fused HIP launch over all 6*L HexPlane planes (csrc/regulation.hip).

    tv_loss = fdgs.compute_regulation(gaussians, hyper.time_smoothness_weight, hyper.l1_time_planes, hyper.plane_tv_weight)

Same value (a 0-dim tensor with a grad_fn) and same gradients to the plane Parameters as
    plane_tv_weight * _plane_regulation() + time_smoothness_weight * _time_regulation() + l1_time_planes_weight * _l1_regulation()
where `_plane_regulation` sums `compute_plane_smoothness` (scene/regulation.py:22-28) over planes (0,1,3) of every level
and `_time_regulation` / `_l1_regulation` run over planes (2,4,5).  No CPU fallback: the kernel library must be present.
"""
import torch

from . import _lib
from ._lib import RegPlane, check, ptr, stream_ptr

SPATIAL, TEMPORAL = (0, 1, 3), (2, 4, 5)


def _planes_of(obj):
    net = obj._deformation if hasattr(obj, "_deformation") else obj
    grids = net.deformation_net.grid.grids
    for level in grids:
        if len(level) != 6:
            raise NotImplementedError("only 4-D (six-plane) HexPlane grids are supported (the reference skips 3-plane grids)")
    return [g for level in grids for g in level]


def compute_regulation(pc_or_net, time_smoothness_weight, l1_time_planes_weight, plane_tv_weight):
    planes = _planes_of(pc_or_net)
    return _Regulation.apply(float(time_smoothness_weight), float(l1_time_planes_weight), float(plane_tv_weight), *planes)


def _cl(p):
    return p if p.is_contiguous(memory_format=torch.channels_last) and p.dtype == torch.float32 else \
        p.float().contiguous(memory_format=torch.channels_last)


def _descriptors(planes, grads, tsw, l1w, tvw):
    arr = (RegPlane * len(planes))()
    for i, pl in enumerate(planes):
        k = i % 6
        d = arr[i]
        d.plane, d.grad_opt = pl.data_ptr(), (grads[i].data_ptr() if grads is not None and grads[i] is not None else None)
        d.H, d.W, d.C = pl.shape[2], pl.shape[3], pl.shape[1]
        d.w_smooth = tvw if k in SPATIAL else tsw
        d.w_l1 = l1w if k in TEMPORAL else 0.0
    return arr


class _Regulation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tsw, l1w, tvw, *planes):
        L = _lib.lib()
        if planes[0].device.type != "cuda":
            raise _lib.FdgsError("the regulariser kernel runs on the GPU only")
        pl = [_cl(p.detach()) for p in planes]
        loss = torch.zeros(1, device=pl[0].device, dtype=torch.float32)
        check(L.fdgs_plane_regulation(stream_ptr(), len(pl), _descriptors(pl, None, tsw, l1w, tvw), 1.0, None, ptr(loss)))
        ctx.w = (tsw, l1w, tvw)
        ctx.pl = pl
        ctx.shapes = [tuple(p.shape) for p in planes]
        ctx.needs = ctx.needs_input_grad[3:]
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        tsw, l1w, tvw = ctx.w
        pl = ctx.pl
        dev = pl[0].device
        sizes = [(int(p.numel()) + 63) // 64 * 64 for p in pl]
        arena = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
        grads, off = [], 0
        for p, n, need in zip(pl, sizes, ctx.needs):
            s = p.shape
            v = arena[off:off + p.numel()].view(1, s[2], s[3], s[1]).permute(0, 3, 1, 2)   # logical [1,C,H,W], channels_last
            grads.append(v if need else None)
            off += n
        gs = g.detach().float().reshape(1).contiguous()
        check(L.fdgs_plane_regulation(stream_ptr(), len(pl), _descriptors(pl, grads, tsw, l1w, tvw), 1.0, ptr(gs), None))
        return (None, None, None, *grads)
