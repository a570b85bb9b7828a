"""Densification bookkeeping of the Gaussian set: drop-ins for the GaussianModel methods the train loop calls on the
consumer side of render()'s `radii` / `viewspace_points.grad` (train.py:259-285):

    gaussians.add_densification_stats(grad, visibility_filter)   ->  fdgs.densify.add_densification_stats(gaussians, grad, visibility_filter, radii)
                                                                     (also does train.py:261's max_radii2D update; one launch, no host sync)
    gaussians.densify(max_grad, min_opacity, extent, size, ...)  ->  fdgs.densify.densify(gaussians, max_grad, min_opacity, extent, size, ...)
    gaussians.prune(max_grad, min_opacity, extent, size)         ->  fdgs.densify.prune(gaussians, ...)
    gaussians.prune_points(mask)                                 ->  fdgs.densify.prune_points(gaussians, mask)
    gaussians.reset_opacity()                                    ->  fdgs.densify.reset_opacity(gaussians)

(scene/gaussian_model.py:269-272, 316-456, 481-500, 516-518).  `gaussians` is the reference's GaussianModel (or anything
with the same attributes: _xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation, optimizer with the named
param groups, xyz_gradient_accum, denom, max_radii2D, _deformation_accum, _deformation_table, percent_dense).  Clone,
split and prune run as a plan (classify + scan, ONE readback of three counts) and one apply launch that writes all six
Parameters, both Adam moments and the side arrays (csrc/densify.hip); the optimizer-state surgery keeps the reference's
contract: new nn.Parameter objects in the same param_groups, `exp_avg` / `exp_avg_sq` rows of kept Gaussians preserved,
zeros for new rows, `step` untouched.  No CPU fallback.
"""
import ctypes
import os

import torch
from torch import nn

from . import _lib
from ._lib import GaussiansIn, GaussiansOut, check, ptr, stream_ptr

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
        "rotation": "_rotation"}
PLAN_DENSIFY, PLAN_PRUNE, PLAN_MASK = 0, 1, 2
# densify / prune / prune_points leave the set in the reference's row order (kept originals, clones, first children, second children)
# and then -- unless switched off -- put it back along the Hilbert curve (spatial_reorder): the order is semantically free, and the
# deformation kernels are measured (bench.py) in the order the train loop keeps.  FDGS_AUTO_REORDER=0 / `reorder=False` keeps the
# reference's row order (what the row-order tests compare).  Cost: bench.py's "reorder" figure, once per densification interval.
AUTO_REORDER = os.environ.get("FDGS_AUTO_REORDER", "1") != "0"


def _rows(n, tail, dev, dtype=torch.float32):
    """([n, *tail] uninitialised, its device address); backed by at least one row because torch reports a NULL data_ptr for
    empty tensors and the C-ABI treats NULL outputs as errors / "skip"."""
    base = torch.empty((max(n, 1),) + tuple(tail), device=dev, dtype=dtype)
    return base[:n], base.data_ptr()


def _f32(t):
    return t if t.dtype == torch.float32 and t.is_contiguous() else t.float().contiguous()


def add_densification_stats(pc, viewspace_point_tensor, update_filter, radii=None):
    """xyz_gradient_accum[f] += |grad[f, :2]|, denom[f] += 1 (scene/gaussian_model.py:516-518) and, when `radii` is given,
    max_radii2D[f] = max(max_radii2D[f], radii[f]) (train.py:261), in place, without the host syncs of boolean indexing."""
    g = viewspace_point_tensor          # (the reference passes the GRADIENT tensor under this name, train.py:262)
    if g.device.type != "cuda":
        raise _lib.FdgsError("the densification kernels run on the GPU only")
    g = _f32(g.detach())
    n = g.shape[0]
    vis = update_filter.contiguous().view(torch.uint8) if update_filter.dtype == torch.bool else (update_filter != 0).view(torch.uint8)
    r = None
    if radii is not None:
        r = radii if radii.dtype == torch.int32 and radii.is_contiguous() else radii.to(torch.int32).contiguous()
    for name in ("xyz_gradient_accum", "denom", "max_radii2D"):
        t = getattr(pc, name)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError(f"{name} must be a contiguous float32 tensor")
    check(_lib.lib().fdgs_densification_stats(stream_ptr(), n, ptr(r), ptr(vis), ptr(g), g.shape[1], ptr(pc.max_radii2D) if r is not None else None,
                                              ptr(pc.xyz_gradient_accum), ptr(pc.denom)))


def _groups(pc):
    by_name = {}
    for group in pc.optimizer.param_groups:
        if len(group["params"]) == 1 and group.get("name") in GROUPS:
            by_name[group["name"]] = group
    missing = [n for n in GROUPS if n not in by_name]
    if missing:
        raise ValueError(f"optimizer has no single-tensor param group named {missing}")
    return by_name


def _restructure(pc, mode, *, grad_threshold=0.0, dense_size=0.0, min_opacity=0.0, max_screen_size=0.0, max_world_size=0.0,
                 drop_mask=None, normals=None, carry_stats=True):
    L = _lib.lib()
    groups = _groups(pc)
    params = {n: groups[n]["params"][0] for n in GROUPS}
    dev = params["xyz"].device
    if dev.type != "cuda":
        raise _lib.FdgsError("the densification kernels run on the GPU only")
    N = params["xyz"].shape[0]
    src = {n: _f32(params[n].detach()) for n in GROUPS}
    nbytes = ctypes.c_size_t()
    check(L.fdgs_densify_scratch_bytes(N, ctypes.byref(nbytes)))
    scratch = torch.empty(max(nbytes.value, 16), dtype=torch.uint8, device=dev)
    counts = (ctypes.c_uint32 * 3)()
    accum, denom = _f32(pc.xyz_gradient_accum), _f32(pc.denom)
    radii2d = _f32(pc.max_radii2D)
    mask8 = None
    if drop_mask is not None:
        mask8 = (drop_mask if drop_mask.dtype == torch.bool else drop_mask != 0).contiguous().view(torch.uint8)
    check(L.fdgs_densify_plan(stream_ptr(), mode, N, ptr(accum), ptr(denom), ptr(src["scaling"]), ptr(src["opacity"]), ptr(radii2d),
                              ptr(mask8), grad_threshold, dense_size, min_opacity, max_screen_size, max_world_size, ptr(scratch), counts))
    kept, clones, splits = int(counts[0]), int(counts[1]), int(counts[2])
    n_out = kept + clones + 2 * splits
    # data-parallel replicas must restructure identically.  The plan (kept / clones / splits) is a function of statistics the caller has
    # already reduced over ranks (parallel.allreduce_densification_stats); it is CHECKED here on every call, before anything depends on it
    # (a forgotten reduction, a threshold comparison that fell the other way on one rank, or one rank arriving with its own `normals`
    # would otherwise leave the ranks that do reach the broadcast below blocked until the process-group timeout)
    import torch.distributed as dist
    from . import parallel as _parallel
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    _parallel.check_same_plan([kept, clones, splits, 1 if normals is None else 0], dev)
    if splits and normals is None:
        normals = torch.randn(2 * splits, 3, device=dev)
        # the children's positions are sampled: rank 0's samples are used everywhere instead of relying on lock-stepped per-rank RNG streams
        if multi:
            dist.broadcast(normals, src=0)
    if normals is not None:
        normals = _f32(normals)
        if normals.shape[0] < 2 * splits:
            raise ValueError("need 2 * splits rows of standard-normal samples")
        normals = normals[:2 * splits].contiguous()

    gin, gout = GaussiansIn(), GaussiansOut()
    gin.N = N
    new_param, new_m, new_v, states = {}, {}, {}, {}
    for i, n in enumerate(GROUPS):
        p = src[n]
        w = 1
        for d in p.shape[1:]:
            w *= int(d)
        gin.width[i] = w
        gin.param[i] = p.data_ptr()
        st = pc.optimizer.state.get(params[n], None)
        states[n] = st
        has = st is not None and "exp_avg" in st
        m = _f32(st["exp_avg"]) if has else None
        v = _f32(st["exp_avg_sq"]) if has else None
        gin.exp_avg[i] = m.data_ptr() if has else None
        gin.exp_avg_sq[i] = v.data_ptr() if has else None
        new_param[n], gout.param[i] = _rows(n_out, p.shape[1:], dev)
        if has:
            new_m[n], gout.exp_avg[i] = _rows(n_out, p.shape[1:], dev)
            new_v[n], gout.exp_avg_sq[i] = _rows(n_out, p.shape[1:], dev)
        states[n] = (st, m, v)                                  # keeps m / v alive until the launch is queued
    table = pc._deformation_table
    table8 = table.contiguous().view(torch.uint8) if table.dtype == torch.bool else (table != 0).view(torch.uint8)
    new_table, gout.deformation_table = _rows(n_out, (), dev, torch.uint8)
    gin.deformation_table = table8.data_ptr()
    dacc = _f32(pc._deformation_accum)
    if carry_stats:
        new_stats = {}
        new_stats["xyz_gradient_accum"], gout.xyz_gradient_accum = _rows(n_out, (1,), dev)
        new_stats["denom"], gout.denom = _rows(n_out, (1,), dev)
        new_stats["max_radii2D"], gout.max_radii2D = _rows(n_out, (), dev)
        new_stats["_deformation_accum"], gout.deformation_accum = _rows(n_out, (3,), dev)
        gin.xyz_gradient_accum, gin.denom, gin.max_radii2D, gin.deformation_accum = accum.data_ptr(), denom.data_ptr(), radii2d.data_ptr(), dacc.data_ptr()
    else:   # densification_postfix re-creates the four statistics as zeros of the new size
        new_stats = {"xyz_gradient_accum": torch.zeros(n_out, 1, device=dev), "denom": torch.zeros(n_out, 1, device=dev),
                     "max_radii2D": torch.zeros(n_out, device=dev), "_deformation_accum": torch.zeros(n_out, 3, device=dev)}
    check(L.fdgs_densify_apply(stream_ptr(), ctypes.byref(gin), ctypes.byref(gout), ptr(scratch), ptr(normals)))

    for n in GROUPS:
        st = states[n][0]
        new = nn.Parameter(new_param[n].requires_grad_(True))
        if st is not None:
            if n in new_m:
                st["exp_avg"], st["exp_avg_sq"] = new_m[n], new_v[n]
            del pc.optimizer.state[params[n]]
            pc.optimizer.state[new] = st
        groups[n]["params"][0] = new
        setattr(pc, ATTR[n], new)
    pc._deformation_table = new_table.view(torch.bool)
    for k, v in new_stats.items():
        setattr(pc, k, v)
    return kept, clones, splits


def densify(pc, max_grad, min_opacity, extent, max_screen_size, density_threshold=None, displacement_scale=None, model_path=None,
            iteration=None, stage=None, normals=None, reorder=None):
    """GaussianModel.densify (scene/gaussian_model.py:495-500): clone the small high-gradient Gaussians, split the large
    ones into two.  `normals` ([2*splits, 3] standard-normal samples) may be supplied for reproducibility; by default they
    come from torch.randn on the device.  `reorder` (default: AUTO_REORDER) re-orders the grown set along the Hilbert curve afterwards.
    Returns (kept, clones, splits)."""
    out = _restructure(pc, PLAN_DENSIFY, grad_threshold=float(max_grad), dense_size=float(pc.percent_dense) * float(extent),
                       normals=normals, carry_stats=False)
    _maybe_reorder(pc, reorder, changed=out[1] + out[2] > 0)
    return out


def _maybe_reorder(pc, reorder, changed):
    """Back onto the space-filling curve after the set was rebuilt (new rows were appended at the tail / rows were dropped)."""
    if ((AUTO_REORDER and changed) if reorder is None else reorder) and pc._xyz.shape[0] > 1:
        spatial_reorder(pc)


def prune(pc, max_grad, min_opacity, extent, max_screen_size, reorder=None):
    """GaussianModel.prune (scene/gaussian_model.py:481-494): drop transparent, screen-filling or oversized Gaussians."""
    out = _restructure(pc, PLAN_PRUNE, min_opacity=float(min_opacity), max_screen_size=float(max_screen_size or 0.0),
                       max_world_size=0.1 * float(extent))
    _maybe_reorder(pc, reorder, changed=False)     # (dropping rows of an ordered set leaves it ordered: only an explicit reorder=True acts)
    return out


def prune_points(pc, mask, reorder=None):
    """GaussianModel.prune_points (scene/gaussian_model.py:350-365): drop the Gaussians where `mask` is True."""
    out = _restructure(pc, PLAN_MASK, drop_mask=mask)
    _maybe_reorder(pc, reorder, changed=False)
    return out


def reset_opacity(pc):
    """GaussianModel.reset_opacity (scene/gaussian_model.py:269-272): opacity <- inverse_sigmoid(min(sigmoid(o), 0.01)), Adam
    moments of the opacity group zeroed.  Two elementwise device ops; no kernel of its own."""
    group = _groups(pc)["opacity"]
    old = group["params"][0]
    x = torch.clamp_max(torch.sigmoid(old.detach()), 0.01)
    new = nn.Parameter(torch.log(x / (1 - x)).requires_grad_(True))
    st = pc.optimizer.state.get(old, None)
    if st is not None:
        st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(new), torch.zeros_like(new)
        del pc.optimizer.state[old]
        pc.optimizer.state[new] = st
    group["params"][0] = new
    pc._opacity = new


# ---- spatial (Hilbert / Morton) ordering of the Gaussian set -----------------------------------------------------------
def _quantise(xyz, lo, hi, bits):
    p = xyz.detach().float()
    lo = p.min(0).values if lo is None else torch.as_tensor(lo, device=p.device, dtype=torch.float32)
    hi = p.max(0).values if hi is None else torch.as_tensor(hi, device=p.device, dtype=torch.float32)
    lo, hi = torch.minimum(lo, hi), torch.maximum(lo, hi)
    return ((p - lo) / (hi - lo).clamp_min(1e-20) * (2 ** bits)).clamp(0, 2 ** bits - 1).to(torch.int64)


def _spread3(v):          # 10 bits abc... -> a00b00c00...
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def morton_keys(xyz, lo=None, hi=None, bits=10):
    """30-bit Morton (Z-order) code of every position: each axis quantised to `bits` (<= 10) bits inside [lo, hi] (default:
    the bounding box of the points), bits interleaved x -> bit 0, y -> bit 1, z -> bit 2.  Elementwise torch integer ops."""
    q = _quantise(xyz, lo, hi, bits)
    return _spread3(q[:, 0]) | (_spread3(q[:, 1]) << 1) | (_spread3(q[:, 2]) << 2)


def hilbert_keys(xyz, lo=None, hi=None, bits=10):
    """Position along a 3-D Hilbert curve of 2^bits cells per axis (Skilling's transpose algorithm, vectorised over the
    points with elementwise torch integer ops).  Unlike the Z-curve the Hilbert curve never jumps: any run of consecutive
    points along it is a connected blob, so a chunk of G consecutive Gaussians has a tight bounding box -- which is what the
    plane-gradient kernel's texel windows need (at 300 k points, 128-point chunks fit a 16-texel window of the 128^2 planes
    for 96 % of the points in Hilbert order against 80 % in Morton order)."""
    q = _quantise(xyz, lo, hi, bits)
    X = [q[:, 0].clone(), q[:, 1].clone(), q[:, 2].clone()]
    Q = 1 << (bits - 1)
    while Q > 1:                                    # inverse undo of the excess work
        P = Q - 1
        for i in range(3):
            hit = (X[i] & Q) != 0
            t = (X[0] ^ X[i]) & P
            x0 = torch.where(hit, X[0] ^ P, X[0] ^ t)
            if i != 0:
                X[i] = torch.where(hit, X[i], X[i] ^ t)
            X[0] = x0
        Q >>= 1
    X[1] = X[1] ^ X[0]                              # Gray encode
    X[2] = X[2] ^ X[1]
    t = torch.zeros_like(X[0])
    Q = 1 << (bits - 1)
    while Q > 1:
        t = torch.where((X[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    X = [x ^ t for x in X]
    return (_spread3(X[0]) << 2) | (_spread3(X[1]) << 1) | _spread3(X[2])


def spatial_reorder(pc, perm=None, curve="hilbert"):
    """Re-order the Gaussian set along a Hilbert (default) or Morton curve of the canonical positions, or by the given
    permutation.

    The ORDER of the Gaussians carries no meaning anywhere in the reference (every per-Gaussian array is indexed in lock-step
    and the rasterizer sorts by depth; only exact depth ties are broken by index), but it decides how the HexPlane kernels
    touch memory: with neighbours in the array being neighbours in space, the lanes of a wave sample the same texel lines
    (deformation forward gather) and the plane-gradient kernel can sum a whole workgroup's contributions to a small texel
    window on the matrix cores before touching memory (csrc/deform.hip D4) instead of issuing one atomic line per
    (Gaussian, corner).  The train loop calls this where the set changes anyway -- after densify / prune (every 100
    iterations, scene/gaussian_model.py:409-500) -- so it costs nothing per frame.

    Permutes, consistently: the six Parameters (new nn.Parameter objects in the same optimizer param groups, as densify
    does), their Adam moments, xyz_gradient_accum / denom / max_radii2D / _deformation_accum / _deformation_table.  A model
    without an optimizer (evaluation, synthetic scenes) has its Parameters' storage replaced in place.  Returns the
    permutation (new row i = old row perm[i])."""
    xyz = pc._xyz
    if perm is None:
        aabb = None
        try:
            aabb = pc._deformation.deformation_net.grid.aabb
        except AttributeError:
            pass
        keyfn = hilbert_keys if curve == "hilbert" else morton_keys
        if aabb is not None and aabb.device == xyz.device:
            keys = keyfn(xyz, aabb[1], aabb[0])              # aabb[0] = max, aabb[1] = min (scene/hexplane.py:19-20)
        else:
            keys = keyfn(xyz)
        perm = torch.argsort(keys, stable=True)
    perm = perm.to(xyz.device)
    opt = getattr(pc, "optimizer", None)
    groups = _groups(pc) if opt is not None else None
    from . import deformation as _deformation
    for n in GROUPS:
        old = getattr(pc, ATTR[n])
        data = old.detach().index_select(0, perm).contiguous()
        if opt is None:
            with torch.no_grad():
                old.data = data
            old.grad = None
            _deformation.invalidate_caches(old)     # same Parameter object, new order: the cached order hint is stale
            continue
        new = nn.Parameter(data.requires_grad_(True))
        st = opt.state.get(old, None)
        if st is not None:
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st:
                    st[k] = st[k].index_select(0, perm).contiguous()
            del opt.state[old]
            opt.state[new] = st
        groups[n]["params"][0] = new
        setattr(pc, ATTR[n], new)
    for name in ("xyz_gradient_accum", "denom", "max_radii2D", "_deformation_accum", "_deformation_table"):
        t = getattr(pc, name, None)
        if isinstance(t, torch.Tensor) and t.dim() >= 1 and t.shape[0] == perm.shape[0]:
            setattr(pc, name, t.index_select(0, perm.to(t.device)).contiguous())
    return perm
