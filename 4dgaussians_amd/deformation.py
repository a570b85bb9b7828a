"""Drop-in `deform_network` whose forward/backward run as fused HIP kernels (csrc/deform.hip).

Mirrors the module tree and therefore the `state_dict()` keys of the reference:
  scene/deformation.py:161-216  deform_network  (timenet, deformation_net, *_poc buffers, get_mlp_parameters /
                                                 get_grid_parameters: optimizer groups are split on the substring "grid")
  scene/deformation.py:16-160   Deformation     (grid, feature_out, pos/scales/rotations/opacity/shs_deform)
  scene/hexplane.py:109-183     HexPlaneField   (aabb Parameter [[max],[min]], grids[level][plane] of shape [1,C,res_j,res_i])
so `deformation.pth` / checkpoints of the reference load unchanged and `GaussianModel.training_setup`'s parameter
grouping keeps working.  Only `forward` differs: one kernel launch instead of ~60.  Plane parameters are kept in
channels_last memory (same logical shape and values) because the kernels gather all C channels of a texel at once.

Unsupported reference options raise instead of silently running something else: no_grid, grid_pe, static_mlp,
empty_voxel, apply_rotation, defor_depth > 1 (none is enabled by any config under arguments/ of the reference).
"""
import itertools
import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import MAX_LEVELS, NUM_HEADS, DeformGrads, DeformOut, DeformParams, check, ptr, stream_ptr

# Training-time trade: when a backward will follow, the forward also stores the HexPlane features, relu(hidden) and
# relu(h1) of every active head (~3.2 KB per Gaussian at net_width 128 with five heads) so that the backward skips the
# gather and the recomputation of those layers.  False = recompute everything (no extra memory).
SAVE_ACTIVATIONS = os.environ.get("FDGS_SAVE_ACTIVATIONS", "1") != "0"

HEAD_NAMES = ("pos_deform", "scales_deform", "rotations_deform", "opacity_deform", "shs_deform")
HEAD_FLAGS = ("no_dx", "no_ds", "no_dr", "no_do", "no_dshs")
HEAD_K = (3, 3, 4, 1, 48)
PLANE_PAIRS = list(itertools.combinations(range(4), 2))


def _is_hip_device(dev):
    """(one predicate so that the host dry-run test can stand in for a device; there is no CPU path)"""
    return dev.type == "cuda"


class HexPlaneField(nn.Module):
    def __init__(self, bounds, planeconfig, multires):
        super().__init__()
        self.aabb = nn.Parameter(torch.tensor([[bounds] * 3, [-bounds] * 3], dtype=torch.float32), requires_grad=False)
        self.grid_config = [planeconfig]
        self.multiscale_res_multipliers = list(multires)
        self.concat_features = True
        assert planeconfig["grid_dimensions"] == 2 and planeconfig["input_coordinate_dim"] == 4
        C = planeconfig["output_coordinate_dim"]
        base = planeconfig["resolution"]
        self.grids = nn.ModuleList()
        self.feat_dim = 0
        for m in self.multiscale_res_multipliers:
            reso = [r * m for r in base[:3]] + list(base[3:])
            level = nn.ParameterList()
            for (i, j) in PLANE_PAIRS:
                t = torch.empty(1, C, reso[j], reso[i])
                if 3 in (i, j):
                    nn.init.ones_(t)          # time planes start as identity (scene/hexplane.py:64-65)
                else:
                    nn.init.uniform_(t, a=0.1, b=0.5)
                level.append(nn.Parameter(t.contiguous(memory_format=torch.channels_last)))
            self.grids.append(level)
            self.feat_dim += C

    @property
    def get_aabb(self):
        return self.aabb[0], self.aabb[1]

    def set_aabb(self, xyz_max, xyz_min):
        self.aabb = nn.Parameter(torch.tensor([list(xyz_max), list(xyz_min)], dtype=torch.float32,
                                              device=self.aabb.device), requires_grad=False)


def _head(W, k):
    return nn.Sequential(nn.ReLU(), nn.Linear(W, W), nn.ReLU(), nn.Linear(W, k))


class Deformation(nn.Module):
    def __init__(self, D=8, W=256, input_ch=27, input_ch_time=9, grid_pe=0, skips=(), args=None):
        super().__init__()
        self.D, self.W, self.grid_pe, self.args = D, W, grid_pe, args
        self.no_grid = args.no_grid
        for flag in ("no_grid", "empty_voxel", "static_mlp", "apply_rotation"):
            if getattr(args, flag, False):
                raise NotImplementedError(f"{flag}=True is not supported by the fused HIP deformation path")
        if grid_pe not in (0, 1):
            raise NotImplementedError("grid_pe > 1 is not supported by the fused HIP deformation path")
        if D > 1:
            raise NotImplementedError("defor_depth > 1 is not supported by the fused HIP deformation path")
        self.grid = HexPlaneField(args.bounds, args.kplanes_config, args.multires)
        self.ratio = 0
        self.feature_out = nn.Sequential(nn.Linear(self.grid.feat_dim, W))
        self.pos_deform = _head(W, 3)
        self.scales_deform = _head(W, 3)
        self.rotations_deform = _head(W, 4)
        self.opacity_deform = _head(W, 1)
        self.shs_deform = _head(W, 48)

    @property
    def get_aabb(self):
        return self.grid.get_aabb

    @property
    def get_empty_ratio(self):
        return self.ratio

    def set_aabb(self, xyz_max, xyz_min):
        self.grid.set_aabb(xyz_max, xyz_min)

    def get_mlp_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" not in n]

    def get_grid_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" in n]


def _init_linear(m):
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight, gain=1)  # biases keep the PyTorch default (scene/deformation.py:218-224)


class deform_network(nn.Module):
    def __init__(self, args):
        super().__init__()
        times_ch = 2 * args.timebase_pe + 1
        self.timenet = nn.Sequential(nn.Linear(times_ch, args.timenet_width), nn.ReLU(),
                                     nn.Linear(args.timenet_width, args.timenet_output))
        self.deformation_net = Deformation(W=args.net_width, D=args.defor_depth, input_ch=3 + 3 * args.posebase_pe * 2,
                                           grid_pe=args.grid_pe, input_ch_time=args.timenet_output, args=args)
        self.register_buffer("time_poc", torch.FloatTensor([2 ** i for i in range(args.timebase_pe)]))
        self.register_buffer("pos_poc", torch.FloatTensor([2 ** i for i in range(args.posebase_pe)]))
        self.register_buffer("rotation_scaling_poc", torch.FloatTensor([2 ** i for i in range(args.scale_rotation_pe)]))
        self.register_buffer("opacity_poc", torch.FloatTensor([2 ** i for i in range(args.opacity_pe)]))
        self.args = args
        self.apply(_init_linear)

    @property
    def get_aabb(self):
        return self.deformation_net.get_aabb

    @property
    def get_empty_ratio(self):
        return self.deformation_net.get_empty_ratio

    def get_mlp_parameters(self):
        return self.deformation_net.get_mlp_parameters() + list(self.timenet.parameters())

    def get_grid_parameters(self):
        return self.deformation_net.get_grid_parameters()

    def forward(self, point, scales=None, rotations=None, opacity=None, shs=None, times_sel=None):
        return self.forward_dynamic(point, scales, rotations, opacity, shs, times_sel)

    def forward_dynamic(self, point, scales=None, rotations=None, opacity=None, shs=None, times_sel=None):
        """Module-level API of the reference: raw (pre-activation) outputs (scene/deformation.py:198-212)."""
        out = deform(self, point, scales, rotations, opacity, shs=shs, time=times_sel, activate=False)
        return out[0], out[1], out[2], out[3], out[4].reshape(shs.shape)


def _head_on(args):
    return [0 if getattr(args, f) else 1 for f in HEAD_FLAGS]


def _collect(net):
    """Flat, ordered tensor list handed to the autograd Function (must match _DeformFunction.backward).
    The walk through the module tree (~90 nn.Module.__getattr__ / container lookups, 0.1 ms per frame on the host) is done once per module
    and kept ON the module (it dies with it; nothing else holds the Parameters).  Every call re-validates EVERY entry with plain dict
    look-ups: each cached (container, key) slot must still hold the cached object -- the 40 Parameters in their owners' `_parameters`, and
    the modules on the way down in their parents' `_modules` -- so a replaced Parameter (`load_state_dict(assign=True)`, `lin.weight = ...`),
    a swapped head Sequential or a re-created grid level all rebuild the list."""
    e = net.__dict__.get("_fdgs_collect")
    if e is not None:
        for owner_dict, key, obj in e[0]:
            if owner_dict.get(key) is not obj:
                break
        else:
            return e[1], e[2]
    dn = net.deformation_net
    slots = [(net._modules, "deformation_net", dn), (dn._modules, "grid", dn.grid), (dn.grid._modules, "grids", dn.grid.grids),
             (dn._modules, "feature_out", dn.feature_out), (dn.feature_out._modules, "0", dn.feature_out[0])]
    planes = []
    for l in range(len(dn.grid.grids)):
        level = dn.grid.grids[l]
        slots.append((dn.grid.grids._modules, str(l), level))
        for k in range(6):
            planes.append(level[k])
            slots.append((level._parameters, str(k), level[k]))
    lin0 = dn.feature_out[0]
    mlp = [lin0.weight, lin0.bias]
    slots += [(lin0._parameters, "weight", lin0.weight), (lin0._parameters, "bias", lin0.bias)]
    for name in HEAD_NAMES:
        seq = getattr(dn, name)
        slots.append((dn._modules, name, seq))
        for idx in ("1", "3"):
            lin = seq._modules[idx]
            slots += [(seq._modules, idx, lin), (lin._parameters, "weight", lin.weight), (lin._parameters, "bias", lin.bias)]
            mlp += [lin.weight, lin.bias]
    net.__dict__["_fdgs_collect"] = (slots, planes, mlp)
    return planes, mlp


def deform(net, xyz, scales, rotations, opacity, shs=None, shs_dc=None, shs_rest=None, time=None, activate=False, ordered=None):
    """Runs the fused deformation.  Either `shs` ([N,16,3]) or the pair (shs_dc [N,1,3], shs_rest [N,15,3]) is given
    (the pair skips the torch.cat of GaussianModel.get_features, scene/gaussian_model.py:121-124).  `time` is a python
    float (one frame time for all Gaussians, as render() uses it) or a [N,1] tensor.
    Returns (means3D, scales, rotations, opacity, shs [N,16,3]); with activate=True scales/rotations/opacity have had
    exp / normalize / sigmoid applied (gaussian_renderer/__init__.py:97-99).  `ordered`: the rows are spatial neighbours (None: measured
    once per position tensor, see spatial_order_hint) -- selects the plane-gradient kernel, never the result."""
    planes, mlp = _collect(net)
    dn = net.deformation_net
    cfg = dict(C=dn.grid.grid_config[0]["output_coordinate_dim"], L=len(dn.grid.grids), W=dn.W,
               head_on=_head_on(dn.args), activate=bool(activate),
               # grad mode is read HERE: inside autograd.Function.forward it is always off, and needs_input_grad stays True
               # under torch.no_grad() -- evaluation frames (render.py, training_report) must not allocate ~3.2 KB per
               # Gaussian of saved activations nor take the slower saving forward
               save=bool(SAVE_ACTIVATIONS and torch.is_grad_enabled()),
               ordered=spatial_order_hint(xyz) if ordered is None else bool(ordered))
    if isinstance(time, torch.Tensor):
        t_tensor, t_scalar = time, 0.0
    else:
        t_tensor, t_scalar = None, float(time)
    if shs is not None:
        a, b = shs, None
    else:
        a, b = shs_dc, shs_rest
    return _DeformFunction.apply(cfg, t_scalar, xyz, scales, rotations, opacity, a, b, t_tensor, dn.grid.aabb, *planes, *mlp)


def _cl(p):
    """channels_last-contiguous view/copy of a plane [1,C,H,W] -> memory [H][W][C]."""
    return p if p.is_contiguous(memory_format=torch.channels_last) and p.dtype == torch.float32 else \
        p.float().contiguous(memory_format=torch.channels_last)


def _fill_params(cfg, t_scalar, xyz, scales, rotations, opacity, sh_a, sh_b, t_tensor, aabb_host, planes, mlp, keep):
    p = DeformParams()
    N = xyz.shape[0]
    p.N, p.C, p.L, p.W, p.activate = N, cfg["C"], cfg["L"], cfg["W"], int(cfg["activate"])
    for i in range(NUM_HEADS):
        p.head_on[i] = cfg["head_on"][i]
    for l in range(cfg["L"]):
        for k, (i, j) in enumerate(PLANE_PAIRS):
            pl = planes[l * 6 + k]
            p.planes[l][k] = pl.data_ptr()
        # axis resolutions of this level from the plane shapes: plane (0,1) is [1,C,res_y,res_x], (2,3) is [1,C,res_t,res_z]
        p.res[l][0], p.res[l][1] = planes[l * 6 + 0].shape[3], planes[l * 6 + 0].shape[2]
        p.res[l][2], p.res[l][3] = planes[l * 6 + 5].shape[3], planes[l * 6 + 5].shape[2]
    for i in range(6):
        p.aabb[i] = aabb_host[i]
    p.w0, p.b0 = mlp[0].data_ptr(), mlp[1].data_ptr()
    for h in range(NUM_HEADS):
        p.w1[h], p.b1[h] = mlp[2 + 4 * h].data_ptr(), mlp[3 + 4 * h].data_ptr()
        p.w2[h], p.b2[h] = mlp[4 + 4 * h].data_ptr(), mlp[5 + 4 * h].data_ptr()
    p.xyz, p.scales, p.rotations, p.opacity = ptr(xyz), ptr(scales), ptr(rotations), ptr(opacity)
    if sh_b is None:  # one combined [N,16,3] tensor
        p.shs_dc, p.shs_rest = sh_a.data_ptr(), sh_a.data_ptr() + 12
        p.shs_dc_stride = p.shs_rest_stride = 48
    else:
        p.shs_dc, p.shs_rest, p.shs_dc_stride, p.shs_rest_stride = sh_a.data_ptr(), sh_b.data_ptr(), 3, 45
    p.time = ptr(t_tensor)
    p.time_scalar = t_scalar
    keep.append((xyz, scales, rotations, opacity, sh_a, sh_b, t_tensor, planes, mlp))
    return p


# Host-side caches keyed by tensor OBJECT (weak references: an entry dies with its tensor; keying on data_ptr() is wrong -- the caching
# allocator hands a freed tensor's address to the next model's).  Small dicts, not single entries: code that alternates between
# models (merge_many_4dgs.py:85-135 renders three per frame) must not pay a host sync per call.
_CACHE_MAX = 16
_aabb_cache = {}     # id(aabb) -> (weakref, (version, data_ptr), host floats)
_order_cache = {}    # id(xyz)  -> (weakref, (shape, data_ptr), ordered?)
_fresh_streak = {}   # shape -> (measurements made on non-Parameter tensors of this shape, last answer)


def _cache_get(cache, t, key):
    e = cache.get(id(t))
    if e is not None and e[0]() is t and e[1] == key:
        return e
    return None


def _cache_put(cache, t, key, val):
    import weakref
    if len(cache) >= _CACHE_MAX:
        for k in [k for k, e in cache.items() if e[0]() is None]:
            del cache[k]
        while len(cache) >= _CACHE_MAX:
            del cache[next(iter(cache))]
    cache[id(t)] = (weakref.ref(t), key, val)


def invalidate_caches(*tensors):
    """Forget what is cached about these tensors (all tensors when called without arguments).  fdgs.densify calls it for every position
    tensor it replaces or permutes IN PLACE (a model without an optimizer keeps its Parameter object through spatial_reorder);
    code that writes `xyz.data = ...` itself must do the same -- a stale order hint only ever selects the slower of two equivalent
    plane-gradient kernels, a stale aabb would be wrong, which is why that one is also keyed on the tensor's version counter."""
    if not tensors:
        _aabb_cache.clear()
        _order_cache.clear()
        _perm_cache.clear()
        _fresh_streak.clear()
        _net_cache.clear()
        return
    for t in tensors:
        _aabb_cache.pop(id(t), None)
        _order_cache.pop(id(t), None)
        _perm_cache.pop(id(t), None)


def _aabb_to_host(aabb):
    """The 6 aabb floats are needed in the kernel-argument struct; the host copy is cached per tensor object, version and address."""
    key = (aabb._version, aabb.data_ptr())
    e = _cache_get(_aabb_cache, aabb, key)
    if e is not None:
        return e[2]
    vals = [float(x) for x in aabb.detach().reshape(-1).cpu().tolist()]
    _cache_put(_aabb_cache, aabb, key, vals)
    return vals


# ---- the spatial order kept INSIDE the library (round 6) ---------------------------------------------------------------------------------
# The order of the Gaussians is free in the reference, and the deformation kernels are ~2x faster on a set whose neighbours in memory are
# neighbours in space.  fdgs.densify.* keeps the model itself in that order -- but a train loop that keeps the reference's own
# GaussianModel.densify / prune (train.py:259-285) appends clones and children at the tail and its set is in no order at all.  For such a
# model render() reads the parameters THROUGH a Hilbert permutation of the positions, cached per position-Parameter OBJECT (the reference's
# densify / prune always create new Parameters, scene/gaussian_model.py:331-389; optimizer steps move the points in place and leave the
# order as good as it was), and hands radii / visibility / every per-Gaussian gradient back through the inverse: one gather and one
# scatter launch per frame (fdgs_permute_rows, 2 x 236 bytes per Gaussian each).
_perm_cache = {}     # id(xyz) -> (weakref, (shape, data_ptr), int32 permutation)


def implicit_permutation(xyz, aabb=None):
    """The cached Hilbert permutation of `xyz` (new row i = old row perm[i]), computed once per tensor object and storage."""
    key = (tuple(xyz.shape), xyz.data_ptr())
    e = _cache_get(_perm_cache, xyz, key)
    if e is not None:
        return e[2]
    from . import densify as _densify
    with torch.no_grad():
        if aabb is not None and aabb.device == xyz.device:
            keys = _densify.hilbert_keys(xyz, aabb[1], aabb[0])          # aabb[0] = max, aabb[1] = min (scene/hexplane.py:19-20)
        else:
            keys = _densify.hilbert_keys(xyz)
        perm = torch.argsort(keys, stable=True).to(torch.int32).contiguous()
    _cache_put(_perm_cache, xyz, key, perm)
    return perm


def permute_rows(perm, tensors, scatter=False):
    """[out[i] = t[perm[i]] for t in tensors] (scatter: out[perm[i]] = t[i]) for per-Gaussian arrays of 4-byte elements, one launch per
    eight arrays (fdgs_permute_rows).  None entries stay None."""
    L = _lib.lib()
    outs = [None] * len(tensors)
    todo = [(i, t if t.is_contiguous() else t.contiguous()) for i, t in enumerate(tensors) if t is not None]
    n = perm.shape[0]
    for c0 in range(0, len(todo), _lib.MAX_ROW_ARRAYS):
        chunk = todo[c0:c0 + _lib.MAX_ROW_ARRAYS]
        arr = (_lib.RowArray * len(chunk))()
        for k, (i, t) in enumerate(chunk):
            if t.element_size() != 4 or t.shape[0] != n:
                raise ValueError("permute_rows: arrays of 4-byte elements with one row per Gaussian")
            outs[i] = torch.empty_like(t)
            arr[k].src, arr[k].dst, arr[k].width = t.data_ptr(), outs[i].data_ptr(), (t.numel() // n if n else 1)
        check(L.fdgs_permute_rows(stream_ptr(), n, ptr(perm), len(chunk), arr, 1 if scatter else 0))
    return outs


class PermuteRows(torch.autograd.Function):
    """The model's per-Gaussian arrays read through a permutation; the gradients go back through its inverse."""

    @staticmethod
    def forward(ctx, perm, *tensors):
        ctx.perm = perm
        ctx.set_materialize_grads(False)
        return tuple(permute_rows(perm, [t.detach() for t in tensors]))

    @staticmethod
    def backward(ctx, *grads):
        return (None, *permute_rows(ctx.perm, list(grads), scatter=True))


def spatial_order_hint(xyz):
    """True when consecutive rows of `xyz` are spatial neighbours (the set was ordered along a space-filling curve, e.g. by
    fdgs.densify.spatial_reorder).  Measured, not declared: mean distance between consecutive positions against the bounding-box
    diagonal (random order: ~0.38; Hilbert order at 10^4..10^6 points: < 0.05), once per tensor OBJECT and storage -- the train loop only
    creates a new position Parameter when the set is rebuilt (densify / prune / reorder), optimizer steps move it in place and do not
    change the order.  Decides only which of two equivalent plane-gradient kernels runs (fdgs_deform_grads::spatially_ordered).
    A tensor that is not a leaf Parameter (a clone / view made per call) is not measured at all: three reductions and a blocking
    `.item()` per frame would cost more than the slower kernel; pass `ordered=` to deform() to declare its order."""
    key = (tuple(xyz.shape), xyz.data_ptr())
    e = _cache_get(_order_cache, xyz, key)
    if e is not None:
        return e[2]
    if not isinstance(xyz, torch.nn.Parameter):
        # a caller that hands in a FRESH tensor on every call (clone, detached copy) would pay the measurement per frame: after a few
        # consecutive misses on non-Parameter tensors of one shape the last answer for that shape is reused instead
        streak = _fresh_streak.get(key[0], (0, False))
        if not xyz.is_leaf:
            return streak[1]
        if streak[0] >= 4:
            return streak[1]
    val = False
    if xyz.shape[0] >= 256:
        with torch.no_grad():
            x = xyz.detach()
            step = (x[1:] - x[:-1]).norm(dim=1).mean()
            diag = (x.max(0).values - x.min(0).values).norm().clamp_min(1e-20)
            val = bool((step / diag).item() < 0.1)
    _cache_put(_order_cache, xyz, key, val)
    if not isinstance(xyz, torch.nn.Parameter):
        _fresh_streak[key[0]] = (_fresh_streak.get(key[0], (0, False))[0] + 1, val)
        if len(_fresh_streak) > 64:
            _fresh_streak.clear()
    return val


class _FwdState:
    """What one deformation forward leaves behind for its backward (shared by _DeformFunction and the fused render Function)."""
    __slots__ = ("cfg", "p", "keep", "saved_act", "o_xyz", "o_sc", "o_rot", "o_op", "o_sh", "o_norm", "shapes", "plane_shapes")


_pack_scratch = {}
_net_cache = {}       # (ids + storages of a network's 40 tensors) -> (planes, mlp): normalised tensor lists (forward_impl); a few networks (code
                      # that alternates between models -- A/B legs, merge_many_4dgs.py -- must not thrash); the entries are detached views, i.e.
                      # they keep the storages of at most _NET_CACHE_MAX networks alive (invalidate_caches() drops them)
_NET_CACHE_MAX = 4


def forward_impl(cfg, t_scalar, xyz, scales, rotations, opacity, sh_a, sh_b, t_tensor, aabb, rest, want_backward):
    """Runs fdgs_deform_fwd.  `want_backward`: a backward will follow (grad mode on and some input requires grad): the forward
    then also leaves the activations behind (SAVE_ACTIVATIONS)."""
    L = _lib.lib()
    dev = xyz.device
    if not _is_hip_device(dev):
        raise _lib.FdgsError("the deformation kernels run on the GPU only")
    nplanes = cfg["L"] * 6
    planes_in, mlp_in = rest[:nplanes], rest[nplanes:]
    c = lambda t: None if t is None else t.detach().float().contiguous()
    xyz_, scales_, rot_, op_ = c(xyz), c(scales), c(rotations), c(opacity)
    sh_a_, sh_b_ = c(sh_a), c(sh_b)
    t_ = None if t_tensor is None else c(t_tensor).reshape(-1)
    # the network's 40 tensors: normalised (float32, planes channels-last) once per set of (tensor objects, storages) -- the optimizer
    # steps them in place, so on every frame but the first this is 80 id() / data_ptr() calls instead of ~120 tensor ops
    nkey = tuple(map(id, rest)) + tuple(t.data_ptr() for t in rest)
    e = _net_cache.get(nkey)
    if e is not None:
        planes, mlp = e
    else:
        planes = [_cl(p.detach()) for p in planes_in]
        mlp = [c(m) for m in mlp_in]
        # (only cached when nothing had to be converted: a converted copy would go stale under an in-place optimizer step)
        if all(a.data_ptr() == b.data_ptr() for a, b in zip(planes + mlp, rest)):
            while len(_net_cache) >= _NET_CACHE_MAX:
                del _net_cache[next(iter(_net_cache))]
            _net_cache[nkey] = (planes, mlp)
    N = xyz_.shape[0]
    st = _FwdState()
    st.keep = []
    p = _fill_params(cfg, t_scalar, xyz_, scales_, rot_, op_, sh_a_, sh_b_, t_, _aabb_to_host(aabb), planes, mlp, st.keep)
    out = DeformOut()
    st.o_xyz, st.o_sc = torch.empty(N, 3, device=dev), torch.empty(N, 3, device=dev)
    st.o_rot, st.o_op = torch.empty(N, 4, device=dev), torch.empty(N, 1, device=dev)
    st.o_sh = torch.empty(N, 16, 3, device=dev)
    st.o_norm = torch.empty(N, device=dev) if cfg["activate"] else None
    out.xyz, out.scales, out.rotations, out.opacity, out.shs, out.rot_norm = ptr(st.o_xyz), ptr(st.o_sc), ptr(st.o_rot), ptr(st.o_op), ptr(st.o_sh), ptr(st.o_norm)
    saved = None
    if cfg["save"] and want_backward:
        nbytes = _lib.c_size_t()
        check(L.fdgs_deform_saved_bytes(p, nbytes))
        saved = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    out.saved = ptr(saved)
    st.saved_act = saved
    # scratch of the forward kernel: the gathered HexPlane features of the weight-stationary form when nothing is saved for a backward (the
    # default form), or the operand-stream copy of W0 / W1 of the 16-Gaussians-per-wave form (d1_form = 16; without a scratch the library
    # falls back to the 32-Gaussian form).  One persistent buffer per (device, stream, size): forwards on one stream are ordered, forwards on
    # different streams must not share it.
    nbytes = _lib.c_size_t()
    check(L.fdgs_deform_pack_bytes(p, nbytes))
    key = (dev, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0, nbytes.value)
    packed = _pack_scratch.get(key)
    if packed is None:
        if len(_pack_scratch) > 8:
            _pack_scratch.clear()
        packed = _pack_scratch[key] = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    out.packed = ptr(packed)
    check(L.fdgs_deform_fwd(stream_ptr(), p, out))
    st.cfg, st.p = cfg, p
    st.shapes = dict(scales=scales.shape, rot=rotations.shape, op=opacity.shape, sh_a=sh_a.shape, sh_b=None if sh_b is None else sh_b.shape)
    st.plane_shapes = [tuple(q.shape) for q in planes_in]
    return st


class _BwdBuffers:
    __slots__ = ("g", "arena", "d_xyz", "d_sc", "d_rot", "d_op", "d_sha", "d_shb", "d_planes", "d_mlp", "scratch", "keep", "zero_range")


_arena_layouts = {}


def _arena_layout(N, sh_combined, plane_shapes, mlp_shapes):
    """(total floats, floats of the per-Gaussian head, number of per-Gaussian views, [(shape, strides, offset)]) of the gradient arena:
    per-Gaussian arrays, planes (logical [1,C,H,W] on channels-last memory [H][W][C]), MLP tensors; every slice starts 256-byte aligned."""
    key = (N, sh_combined, plane_shapes, mlp_shapes)
    lay = _arena_layouts.get(key)
    if lay is not None:
        return lay

    def contiguous(shape):
        st_, acc = [], 1
        for d_ in reversed(shape):
            st_.append(acc)
            acc *= d_
        return tuple(reversed(st_)), acc

    specs, off = [], 0
    fixed = [(N, 3), (N, 3), (N, 4), (N, 1)] + ([(N, 16, 3)] if sh_combined else [(N, 1, 3), (N, 15, 3)])
    for shp in fixed:
        st_, n_ = contiguous(shp)
        specs.append((shp, st_, off))
        off += (n_ + 63) // 64 * 64
    head = off
    for (_, C_, H_, W_) in plane_shapes:
        specs.append(((1, C_, H_, W_), (H_ * W_ * C_, 1, W_ * C_, C_), off))
        off += (C_ * H_ * W_ + 63) // 64 * 64
    for shp in mlp_shapes:
        st_, n_ = contiguous(shp)
        specs.append((shp, st_, off))
        off += (n_ + 63) // 64 * 64
    if len(_arena_layouts) >= 32:
        _arena_layouts.clear()
    lay = _arena_layouts[key] = (off, head, len(fixed), specs)
    return lay


def backward_prepare(st, o_sc, o_rot, o_op, identity_assigned=False, zero_by_epilogue=False):
    """Allocates the gradient arena and the scratch of fdgs_deform_bwd and fills fdgs_deform_grads (without the upstream gradients).
    The arena is zero-filled (every kernel accumulates); with `identity_assigned` the six per-Gaussian arrays at its head are left
    uninitialised -- the rasterizer backward's epilogue ASSIGNS them (fdgs_raster_deform_epilogue::assign) before anything adds to them;
    with `zero_by_epilogue` (needs `identity_assigned`) the rest (planes, MLP) is not filled here either: `b.zero_range` = (pointer, floats)
    goes into fdgs_raster_deform_epilogue::zero_fill and that kernel clears it on the way (one fill launch per frame less).
    Returns the buffers; `backward_run` launches."""
    L = _lib.lib()
    cfg, p = st.cfg, st.p
    xyz, scales, rotations, opacity, sh_a, sh_b, t_tensor, planes, mlp = st.keep[0]
    dev, N = xyz.device, xyz.shape[0]
    b = _BwdBuffers()
    g = DeformGrads()
    g.out_scales, g.out_rotations, g.out_opacity, g.rot_norm = ptr(o_sc), ptr(o_rot), ptr(o_op), ptr(st.o_norm)
    # every outgoing gradient is accumulated into (+=) by the kernels: ONE zero-filled arena, carved into views
    # (40 separate torch.zeros launches cost more than the fill itself at 150 frames/s)
    # (the layout -- per view: logical shape, strides, offset in floats -- depends on N, the SH form and the parameter shapes only: computed
    # once and kept; per frame it is ONE allocation and one as_strided per view, with the planes' channels-last strides folded in, instead of
    # slice + view (+ permute) per tensor: ~110 tensor ops per frame less on the host)
    lay = _arena_layout(N, sh_b is None, tuple(st.plane_shapes), tuple(tuple(m.shape) for m in mlp))
    total, head, n_fixed, specs = lay
    b.zero_range = None
    if identity_assigned:
        arena = torch.empty(total, device=dev, dtype=torch.float32)
        if zero_by_epilogue:
            b.zero_range = (arena.data_ptr() + 4 * head, total - head)
        else:
            arena[head:].zero_()
    else:
        arena = torch.zeros(total, device=dev, dtype=torch.float32)
    views = [arena.as_strided(sh_, st_, off_) for sh_, st_, off_ in specs]
    base = arena.data_ptr()
    b.arena = arena
    b.d_xyz, b.d_sc, b.d_rot, b.d_op = views[:4]
    if sh_b is None:
        b.d_sha, b.d_shb = views[4], None
        g.d_shs_dc, g.d_shs_rest = base + 4 * specs[4][2], base + 4 * specs[4][2] + 12
    else:
        b.d_sha, b.d_shb = views[4], views[5]
        g.d_shs_dc, g.d_shs_rest = base + 4 * specs[4][2], base + 4 * specs[5][2]
    g.d_xyz, g.d_scales, g.d_rotations, g.d_opacity = base + 4 * specs[0][2], base + 4 * specs[1][2], base + 4 * specs[2][2], base + 4 * specs[3][2]
    nplanes = len(st.plane_shapes)
    b.d_planes = views[n_fixed:n_fixed + nplanes]          # logical [1,C,H,W] with channels_last strides
    for l in range(cfg["L"]):
        for k in range(6):
            g.d_planes[l][k] = base + 4 * specs[n_fixed + l * 6 + k][2]
    b.d_mlp = views[n_fixed + nplanes:]
    mo = [base + 4 * sp[2] for sp in specs[n_fixed + nplanes:]]
    g.d_w0, g.d_b0 = mo[0], mo[1]
    for h in range(NUM_HEADS):
        g.d_w1[h], g.d_b1[h] = mo[2 + 4 * h], mo[3 + 4 * h]
        g.d_w2[h], g.d_b2[h] = mo[4 + 4 * h], mo[5 + 4 * h]
    nbytes = _lib.c_size_t()
    check(L.fdgs_deform_bwd_scratch_bytes(p, nbytes))
    b.scratch = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    g.scratch = ptr(b.scratch)
    g.saved = ptr(st.saved_act)
    g.spatially_ordered = 1 if cfg.get("ordered") else 0
    b.g = g
    return b


# Diagnostics (bench.py's FLOP accounting, tests): when set, every deformation backward reads back how many 32-row tiles it actually
# processed -- (live tiles, tiles, live plane-gradient chunks, chunks) -> `last_live_tiles`.  One blocking read-back per backward: off
# by default.
COUNT_LIVE_TILES = False
last_live_tiles = None


def note_live_tiles(st, b):
    global last_live_tiles
    if COUNT_LIVE_TILES:
        out = (_lib.c_uint32 * 4)()
        check(_lib.lib().fdgs_deform_bwd_live_tiles(stream_ptr(), st.p, b.g.scratch, out))
        last_live_tiles = tuple(int(x) for x in out)


def backward_run(st, b):
    """Launches fdgs_deform_bwd on prepared buffers and returns the gradients in _DeformFunction's input order (after cfg, t_scalar)."""
    check(_lib.lib().fdgs_deform_bwd(stream_ptr(), st.p, b.g))
    note_live_tiles(st, b)
    cfg, sh = st.cfg, st.shapes
    d_mlp = b.d_mlp
    for h in range(NUM_HEADS):  # parameters of a disabled head receive no gradient (as under autograd)
        if not cfg["head_on"][h]:
            d_mlp[2 + 4 * h:6 + 4 * h] = [None] * 4
    return (b.d_xyz, b.d_sc.reshape(sh["scales"]), b.d_rot.reshape(sh["rot"]), b.d_op.reshape(sh["op"]),
            b.d_sha.reshape(sh["sh_a"]), None if b.d_shb is None else b.d_shb.reshape(sh["sh_b"]), None, None,
            *b.d_planes, *d_mlp)


class _DeformFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, t_scalar, xyz, scales, rotations, opacity, sh_a, sh_b, t_tensor, aabb, *rest):
        st = forward_impl(cfg, t_scalar, xyz, scales, rotations, opacity, sh_a, sh_b, t_tensor, aabb, rest, any(ctx.needs_input_grad))
        ctx.st = st
        ctx.saved_act = st.saved_act
        # outputs the backward needs go through save_for_backward: keeping them as plain attributes makes a reference
        # cycle (output -> grad_fn -> ctx -> output) that only the cyclic GC breaks -- with ~1 GB of per-frame buffers
        # hanging on it (saved activations) every frame then costs a fresh 1-GB hipMalloc (30 ms)
        ctx.save_for_backward(st.o_sc, st.o_rot, st.o_op)
        outs = (st.o_xyz, st.o_sc, st.o_rot, st.o_op, st.o_sh)
        st.o_xyz = st.o_sc = st.o_rot = st.o_op = st.o_sh = None
        ctx.set_materialize_grads(False)   # unused outputs arrive as None -> NULL pointers, which the kernels skip
        return outs

    @staticmethod
    def backward(ctx, g_xyz, g_sc, g_rot, g_op, g_sh):
        st = ctx.st
        o_sc, o_rot, o_op = ctx.saved_tensors
        b = backward_prepare(st, o_sc, o_rot, o_op)
        c = lambda t: None if t is None else t.float().contiguous()
        gx, gs, gr, go, gsh = c(g_xyz), c(g_sc), c(g_rot), c(g_op), c(g_sh)
        g = b.g
        g.g_xyz, g.g_scales, g.g_rotations, g.g_opacity, g.g_shs = ptr(gx), ptr(gs), ptr(gr), ptr(go), ptr(gsh)
        return (None, None) + backward_run(st, b)
