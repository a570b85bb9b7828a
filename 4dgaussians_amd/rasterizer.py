"""Drop-in for the python package `diff_gaussian_rasterization` of the reference's rasterizer submodule.

Mirrors what the reference imports and calls at gaussian_renderer/__init__.py:14,38-58,120-128,
merge_many_4dgs.py:33,85-135 and scene/dataset_readers.py:485-508:

    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix,
                                  projmatrix, sh_degree, campos, prefiltered, debug)      # plain picklable NamedTuple
    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                                        rotations=None, cov3D_precomp=None) -> (color [3,H,W], radii [P] int32,
                                                                                depth [1,H,W])
    GaussianRasterizer.markVisible(positions) -> bool [P]

All arithmetic runs in libfdgs.so (hand-written HIP for gfx950) through the C-ABI of include/fdgs.h; PyTorch only
supplies device memory, the current HIP stream and autograd bookkeeping.  There is no CPU fallback.
"""
import ctypes
import os
import threading
import warnings
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import RasterGrads, RasterParams, check, ptr, stream_ptr


def _is_hip_device(dev):
    """(one predicate so that the host dry-run test can stand in for a device; there is no CPU path)"""
    return dev.type == "cuda"


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ---- pair-count bookkeeping of the binning stage -------------------------------------------------------------------------------------
# The reference's rasterizer reads the number of (tile, Gaussian) pairs back to the host in the middle of every forward (to size its
# binning buffer), with nothing queued behind the read: host and device wait for each other once per frame.  Here, once a pair count has
# been seen for (image size, number of Gaussians), the binning buffer is sized for a CAPACITY predicted from the largest count seen so far
# (x CAPACITY_SLACK) and the whole forward is queued by ONE C call (fdgs_raster_fwd_capacity).  BINNING selects what happens next:
#
#   "auto" (default)   VERIFIED: the host waits for the frame's pair count -- it reaches pinned host memory with the projection kernel, while
#                      the depth sort, the tile sort and the blending kernel are still queued, so the device does not drain -- and, if the
#                      count exceeds the capacity, finishes the frame EXACTLY (fdgs_bin_sort + fdgs_render_fwd on a buffer of the true size;
#                      stages 1-2 are complete in `geom`) before it returns.  No image that differs from the exact path ever leaves
#                      rasterize_forward (the reference never truncates a list: SURVEY Appendix B.2); `capacity_reruns` counts such frames.
#   "exact"            the reference's shape: four C calls, the blocking read-back between stage 2 and 3 (FDGS_BINNING=exact).  Also the
#                      first frame of every (size, count) pair in the other modes.
#   "capacity"         OPT-IN, never waits: the count is looked at when a later frame starts; a frame whose count exceeded its capacity has
#                      dropped its FARTHEST pairs -- forward and backward alike, so its gradient is the exact gradient of the image it
#                      returned -- raises a RuntimeWarning and is counted in `capacity_overflows`.  For loops that prefer a host that runs
#                      a full frame ahead over the guarantee (the frame is then NOT the reference's).
# State is per host thread (threading.local): one thread drives a stream, as include/fdgs.h requires.
BINNING = os.environ.get("FDGS_BINNING", "auto")
CAPACITY_SLACK = 1.5
capacity_overflows = 0        # "capacity" mode: frames that dropped pairs
capacity_reruns = 0           # "auto" mode: frames finished exactly after their speculative capacity proved too small
_SENTINEL = 0xFFFFFFFF
_RING = 64
_tls = threading.local()
_seen = {}            # (device index, W, H, P) -> [largest pair count seen, P it was seen at]
_size_cache = {}      # ("geom", P) / ("img", W, H) / ("bin", R) -> bytes


class PairCount:
    """The pair count of one forward: known at once on the exact and verified paths, later in "capacity" mode (`value()` waits for it)."""
    __slots__ = ("addr", "capacity", "key", "P", "R", "stream", "buf")

    def poll(self):
        if self.R is None:
            v = ctypes.c_uint32.from_address(self.addr).value
            if v != _SENTINEL:
                self.R = v
                _note_count(self)
        return self.R

    def value(self):
        if self.poll() is None:
            self.stream.synchronize()
            self.poll()
        return self.R


def _note_count(c):
    global capacity_overflows
    e = _seen.get(c.key)
    if e is None:
        if len(_seen) > 64:
            _seen.clear()
        _seen[c.key] = [c.R, c.P]
    else:
        e[0] = max(e[0], c.R)
    if c.capacity is not None and c.R > c.capacity:
        capacity_overflows += 1
        warnings.warn(f"fdgs rasterizer (BINNING='capacity'): a frame listed {c.R} (tile, Gaussian) pairs but its binning buffer was sized for "
                      f"{c.capacity}: the farthest {c.R - c.capacity} pairs of that frame were dropped; the capacity grows from the next frame on "
                      "(the default BINNING='auto' finishes such a frame exactly instead)", RuntimeWarning, stacklevel=3)


def _thread_state(dev):
    st = getattr(_tls, "st", None)
    if st is None:
        st = _tls.st = {"ring": {}, "pending": [], "last": None}
    ring = st["ring"].get(dev.index)
    if ring is None:
        ring = st["ring"][dev.index] = [_pinned_words(_RING), 0]
    return st, ring


def _pinned_words(n):
    return torch.zeros(n, dtype=torch.int32).pin_memory()


def _current_stream(dev):
    return torch.cuda.current_stream(dev)


def _bytes(L, kind, *args):
    key = (kind,) + args
    v = _size_cache.get(key)
    if v is None:
        nbytes = _lib.c_size_t()
        if kind == "geom":
            check(L.fdgs_geom_bytes(args[0], nbytes))
        elif kind == "img":
            check(L.fdgs_img_bytes(args[0], args[1], nbytes))
        else:
            check(L.fdgs_binning_bytes(args[0], args[1], args[2], nbytes))
        if len(_size_cache) > 256:
            _size_cache.clear()
        v = _size_cache[key] = nbytes.value
    return v


def last_num_rendered():
    """Pair count of this thread's most recent forward (diagnostics; waits for the frame if its count has not arrived yet)."""
    st = getattr(_tls, "st", None)
    return 0 if st is None or st["last"] is None else int(st["last"].value())


def _f32(t, device):
    if t is None:
        return None
    if t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _none_if_empty(t):
    return None if (t is None or t.numel() == 0) else t


class RasterState:
    """Buffers one forward pass leaves behind for its backward (the reference's geom/binning/img buffers).  `capacity` = the pair count
    the binning buffer is laid out for (what the C calls take as num_rendered); `num_rendered` = the true count (waits for it if needed)."""
    __slots__ = ("params", "geom", "binning", "img", "capacity", "count", "keep", "visibility", "acc")

    def take_accumulator(self):
        """(buffer, zeroed) for fdgs_raster_grads::scratch_acc: the [P,16] accumulator the forward zero-filled on the way (training frames), once;
        a second backward of the same state gets a fresh buffer that fdgs_raster_bwd fills itself."""
        acc, self.acc = self.acc, None
        if acc is not None:
            return acc, 1
        return torch.empty(self.params.P, 16, device=self.geom.device), 0

    @property
    def num_rendered(self):
        return int(self.count.value())


def rasterize_forward(settings, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, out=None, expect_backward=False):
    """Runs stages 1-4 of include/fdgs.h. Returns (color, radii, depth, state).  `out` = (color [3,H,W], radii [P] int32, depth [1,H,W])
    buffers to write into (contiguous float32 / int32 on the device), e.g. slices of a batch tensor.  `expect_backward`: a training frame
    (the blending forward zero-fills the backward's accumulator on the way).  Which binning path runs: BINNING above -- every mode but the
    opt-in "capacity" returns the exact path's image."""
    L = _lib.lib()
    dev = means3D.device
    if not _is_hip_device(dev):
        raise _lib.FdgsError("the rasterizer runs on the GPU only (tensors must live on a HIP device)")
    means3D = _f32(means3D, dev)
    P = means3D.shape[0]
    H, W = int(settings.image_height), int(settings.image_width)
    shs, colors_precomp = _f32(_none_if_empty(shs), dev), _f32(_none_if_empty(colors_precomp), dev)
    scales, rotations = _f32(_none_if_empty(scales), dev), _f32(_none_if_empty(rotations), dev)
    cov3D_precomp, opacities = _f32(_none_if_empty(cov3D_precomp), dev), _f32(opacities, dev)
    bg, view = _f32(settings.bg, dev), _f32(settings.viewmatrix, dev)
    proj, campos = _f32(settings.projmatrix, dev), _f32(settings.campos, dev)
    if (shs is None) == (colors_precomp is None) and P > 0:
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        if P > 0:
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
    p = RasterParams()
    p.P, p.sh_degree, p.W, p.H = P, int(settings.sh_degree), W, H
    p.sh_coeffs = 0 if shs is None else int(shs.numel() // max(P, 1) // 3)
    p.tanfovx, p.tanfovy, p.scale_modifier = float(settings.tanfovx), float(settings.tanfovy), float(settings.scale_modifier)
    p.prefiltered, p.debug = int(bool(settings.prefiltered)), int(bool(settings.debug))
    p.bg, p.viewmatrix, p.projmatrix, p.campos = ptr(bg), ptr(view), ptr(proj), ptr(campos)
    p.means3D, p.shs, p.colors_precomp, p.opacities = ptr(means3D), ptr(shs), ptr(colors_precomp), ptr(opacities)
    p.scales, p.rotations, p.cov3D_precomp = ptr(scales), ptr(rotations), ptr(cov3D_precomp)
    geom = torch.empty(_bytes(L, "geom", P), dtype=torch.uint8, device=dev)
    img = torch.empty(_bytes(L, "img", W, H), dtype=torch.uint8, device=dev)
    if out is not None:
        color, radii, depth = out
    else:
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    vis = torch.empty(P, dtype=torch.bool, device=dev)       # radii > 0, written by the projection kernel (no elementwise launch)
    p.visibility = ptr(vis)
    # a training frame: the blending forward zero-fills the backward's per-Gaussian accumulator on the way (no fill launch before the backward)
    acc = torch.empty(P, 16, device=dev) if expect_backward and P > 0 else None
    p.acc_zero = ptr(acc)
    st = stream_ptr()
    tstate, ring = _thread_state(dev)
    pending = tstate["pending"]
    if pending:           # counts of earlier frames that have arrived meanwhile: feed the predictor (never waits)
        tstate["pending"] = pending = [c for c in pending if c.poll() is None]
    key = (dev.index, W, H, P)          # (another model / a densified set: another key, i.e. one exact frame first)
    seen = _seen.get(key)
    cnt = PairCount()
    cnt.key, cnt.P, cnt.R, cnt.stream = key, P, None, None
    cnt.buf = ring[0]                    # (keeps the pinned ring alive for as long as anything may read or write this word)
    slot = ring[1] = (ring[1] + 1) % _RING
    cnt.addr = ring[0].data_ptr() + 4 * slot
    for c in pending:          # (a count still in flight in this slot -- 64 forwards old: cannot happen on a live device -- is waited for, never overwritten)
        if c.addr == cnt.addr:
            c.value()
    speculative = BINNING == "auto" or (BINNING == "capacity" and len(pending) < _RING - 2)
    if speculative and seen is not None and P > 0:
        global capacity_reruns
        cap = (int(seen[0] * CAPACITY_SLACK) + 8192) // 4096 * 4096
        ctypes.c_uint32.from_address(cnt.addr).value = _SENTINEL
        binning = torch.empty(_bytes(L, "bin", cap, W, H), dtype=torch.uint8, device=dev)
        check(L.fdgs_raster_fwd_capacity(st, p, ptr(geom), ptr(binning), ptr(img), cap, _lib.c_void_p(cnt.addr), ptr(radii), ptr(color), ptr(depth)))
        if BINNING == "capacity":      # opt-in: never waits; the count is looked at when a later frame starts
            cnt.capacity = cap
            cnt.stream = _current_stream(dev)
            pending.append(cnt)
        else:                          # verified: the image this call returns is the exact path's
            true_n = _lib.c_uint32()
            check(L.fdgs_pair_count_wait(st, _lib.c_void_p(cnt.addr), true_n))
            cnt.capacity = None
            cnt.R = true_n.value
            _note_count(cnt)
            if cnt.R > cap:            # the speculative frame dropped pairs: finish it exactly (stages 1-2 are complete in `geom`)
                capacity_reruns += 1
                cap = cnt.R
                binning = torch.empty(_bytes(L, "bin", cap, W, H), dtype=torch.uint8, device=dev)
                check(L.fdgs_bin_sort(st, p, ptr(geom), ptr(binning), ptr(img), cap))
                check(L.fdgs_render_fwd(st, p, ptr(geom), ptr(binning), ptr(img), cap, ptr(color), ptr(depth)))
    else:
        check(L.fdgs_preprocess_fwd(st, p, ptr(geom), ptr(radii)))
        check(L.fdgs_bin_prepare(st, p, ptr(geom), _lib.c_void_p(cnt.addr)))          # (blocks until the count is in host memory)
        cap = ctypes.c_uint32.from_address(cnt.addr).value
        cnt.capacity = None
        cnt.R = cap
        _note_count(cnt)
        binning = torch.empty(_bytes(L, "bin", cap, W, H), dtype=torch.uint8, device=dev)
        check(L.fdgs_bin_sort(st, p, ptr(geom), ptr(binning), ptr(img), cap))
        check(L.fdgs_render_fwd(st, p, ptr(geom), ptr(binning), ptr(img), cap, ptr(color), ptr(depth)))
    tstate["last"] = cnt
    state = RasterState()
    state.params, state.geom, state.binning, state.img, state.capacity, state.count = p, geom, binning, img, cap, cnt
    state.visibility = vis
    state.acc = acc
    state.keep = (means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, bg, view, proj, campos)
    return color, radii, depth, state


def rasterize_backward(state, grad_color, grad_depth=None):
    """Backward of rasterize_forward. Returns dict of gradients (None where the input was absent)."""
    L = _lib.lib()
    p = state.params
    means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp = state.keep[:7]
    dev, P = means3D.device, p.P
    g = RasterGrads()
    grad_color = _f32(grad_color, dev)
    grad_depth = _f32(grad_depth, dev)
    out = dict(means2D=torch.empty(P, 3, device=dev), means3D=torch.empty(P, 3, device=dev),
               opacities=torch.empty(P, 1, device=dev), colors=torch.empty(P, 3, device=dev),
               cov3D=torch.empty(P, 6, device=dev),
               shs=None if shs is None else torch.empty(P, p.sh_coeffs, 3, device=dev),
               scales=None if scales is None else torch.empty(P, 3, device=dev),
               rotations=None if rotations is None else torch.empty(P, 4, device=dev))
    scratch, g.scratch_acc_zeroed = state.take_accumulator()
    g.dL_dcolor, g.dL_ddepth = ptr(grad_color), ptr(grad_depth)
    g.dL_dmeans2D, g.dL_dmeans3D, g.dL_dopacity = ptr(out["means2D"]), ptr(out["means3D"]), ptr(out["opacities"])
    g.dL_dcolors, g.dL_dsh, g.dL_dscales = ptr(out["colors"]), ptr(out["shs"]), ptr(out["scales"])
    g.dL_drotations, g.dL_dcov3D, g.scratch_acc = ptr(out["rotations"]), ptr(out["cov3D"]), ptr(scratch)
    check(L.fdgs_raster_bwd(stream_ptr(), p, ptr(state.geom), ptr(state.binning), ptr(state.img), state.capacity, g))
    return out


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, grad_mode=True):
        color, radii, depth, state = rasterize_forward(raster_settings, means3D, sh, colors_precomp, opacities, scales,
                                                       rotations, cov3Ds_precomp, expect_backward=bool(grad_mode) and any(ctx.needs_input_grad))
        ctx.state = state
        ctx.had = (sh is not None and sh.numel() > 0, colors_precomp is not None and colors_precomp.numel() > 0,
                   scales is not None and scales.numel() > 0, cov3Ds_precomp is not None and cov3Ds_precomp.numel() > 0)
        ctx.shapes = (opacities.shape, None if sh is None else sh.shape)
        ctx.mark_non_differentiable(radii)
        # an output the loss never used must arrive in backward as None, not as a materialised zero image: the
        # reference's training loss ignores `depth` (train.py:201-214), and a NULL dL_ddepth selects the blending
        # backward without the depth terms (render_bwd_kernel<false>, csrc/render.hip)
        ctx.set_materialize_grads(False)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth):
        if grad_color is None:
            if grad_depth is None:      # neither image reached the loss
                return (None,) * 10
            grad_color = torch.zeros(3, ctx.state.params.H, ctx.state.params.W, device=grad_depth.device)
        g = rasterize_backward(ctx.state, grad_color, grad_depth)
        had_sh, had_col, had_scale, had_cov = ctx.had
        op_shape, sh_shape = ctx.shapes
        return (g["means3D"], g["means2D"], g["shs"].reshape(sh_shape) if had_sh else None, g["colors"] if had_col else None,
                g["opacities"].reshape(op_shape), g["scales"] if had_scale else None, g["rotations"] if had_scale else None,
                g["cov3D"] if had_cov else None, None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    # (grad mode is read HERE: inside autograd.Function.forward it is always off and needs_input_grad stays True under torch.no_grad())
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings, torch.is_grad_enabled())


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        L = _lib.lib()
        rs = self.raster_settings
        with torch.no_grad():
            pos = _f32(positions, positions.device)
            out = torch.empty(pos.shape[0], dtype=torch.uint8, device=pos.device)
            check(L.fdgs_mark_visible(stream_ptr(), pos.shape[0], ptr(pos), ptr(_f32(rs.viewmatrix, pos.device)),
                                      ptr(_f32(rs.projmatrix, pos.device)), ptr(out)))
        return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)
