"""Drop-in for `gaussian_renderer.render` of the reference (gaussian_renderer/__init__.py:18-138).

Same signature, same result dict {"render","viewspace_points","visibility_filter","radii","depth"}; honours
pipe.convert_SHs_python / pipe.compute_cov3D_python / pipe.debug and cam_type == "PanopticSports" exactly where the
reference does.  When `pc._deformation` is this package's `deform_network`, the fine stage runs as two fused HIP
stages: deform(+activations, + the cat of features_dc/features_rest) -> rasterize; no `time.repeat(N,1)` tensor, no
positional-embedding scratch.  With a foreign deformation module (e.g. the reference's own) the module is called as
the reference calls it and only the rasterizer is replaced.
"""
import math
import os

import torch

from . import _lib
from . import deformation as _deformation
from . import rasterizer as _rasterizer
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

# Fused fine stage: deformation and rasterizer behind ONE autograd node, so that the rasterizer backward's per-Gaussian chain rule
# writes straight into the deformation backward's buffers (fdgs_raster_deform_epilogue) instead of handing five gradient tensors
# through autograd to a separate packing kernel.  FDGS_FUSED_BACKWARD=0 keeps the two-node form (A/B, and what the tests compare with).
FUSED_BACKWARD = os.environ.get("FDGS_FUSED_BACKWARD", "1") != "0"
EPILOGUE_ASSIGN = True     # False: the epilogue accumulates (+=) into a zero-filled arena (the C-ABI's other mode; tests compare both)
# fdgs_raster_deform_epilogue::tile_flags of the fused backward.  None = 2 (dead tiles' rows unwritten, always skipped) unless the library's
# tuning knob skip_dead = 0 asks for every tile to be walked (then 1: flags written, every row written).  0 / 1 / 2 force a mode (tests).
# The deformation backward is told which mode produced its rows (packed_rows_ready = tile_flags + 1), never the environment alone.
EPILOGUE_TILE_FLAGS = None


# The spatial order kept inside the library: a model whose rows are NOT spatial neighbours (the reference's own densify / prune appends at
# the tail) is read through a cached Hilbert permutation of its positions (deformation.implicit_permutation); radii, visibility and the
# per-Gaussian gradients come back in the model's own order.  FDGS_IMPLICIT_ORDER=0 switches it off (the A/B figure `random_order` of bench.py).
IMPLICIT_ORDER = os.environ.get("FDGS_IMPLICIT_ORDER", "1") != "0"
IMPLICIT_ORDER_MIN_N = 8192


def _implicit_perm(pc, cfg, dn):
    if IMPLICIT_ORDER and not cfg["ordered"] and pc._xyz.shape[0] >= IMPLICIT_ORDER_MIN_N and pc._xyz.dtype == torch.float32:
        return _deformation.implicit_permutation(pc._xyz, dn.grid.aabb)
    return None


def _tile_flags():
    if EPILOGUE_TILE_FLAGS is not None:
        return int(EPILOGUE_TILE_FLAGS)
    return 2 if _lib.tuning_get("skip_dead") != 0 else 1


class _FusedRenderFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, t_scalar, raster_settings, means2D, xyz, scales, rotations, opacity, sh_a, sh_b, aabb, *rest):
        st = _deformation.forward_impl(cfg, t_scalar, xyz, scales, rotations, opacity, sh_a, sh_b, None, aabb, rest,
                                       any(ctx.needs_input_grad))
        color, radii, depth, rstate = _rasterizer.rasterize_forward(raster_settings, st.o_xyz, st.o_sh, None, st.o_op, st.o_sc, st.o_rot, None,
                                                                    expect_backward=bool(cfg.get("grad")) and any(ctx.needs_input_grad))
        ctx.st, ctx.rstate = st, rstate
        ctx.save_for_backward(st.o_sc, st.o_rot, st.o_op)       # (see _DeformFunction.forward: no reference cycle through ctx)
        st.o_xyz = st.o_sc = st.o_rot = st.o_op = st.o_sh = None
        vis = rstate.visibility                                 # radii > 0, written by the projection kernel itself
        ctx.mark_non_differentiable(radii, vis)
        ctx.set_materialize_grads(False)
        return color, radii, depth, vis

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_vis=None):
        st, rstate = ctx.st, ctx.rstate
        n_in = 11 + len(st.plane_shapes) + len(st.keep[0][8])
        if grad_color is None and grad_depth is None:
            return (None,) * n_in
        L = _lib.lib()
        o_sc, o_rot, o_op = ctx.saved_tensors
        b = _deformation.backward_prepare(st, o_sc, o_rot, o_op, identity_assigned=EPILOGUE_ASSIGN, zero_by_epilogue=EPILOGUE_ASSIGN)
        p = rstate.params
        dev, P = b.d_xyz.device, p.P
        if grad_color is None:
            grad_color = torch.zeros(3, p.H, p.W, device=dev)
        f = lambda t: None if t is None else t.float().contiguous()
        grad_color, grad_depth = f(grad_color), f(grad_depth)
        g = _lib.RasterGrads()
        g_means2D = torch.empty(P, 3, device=dev)
        acc, g.scratch_acc_zeroed = rstate.take_accumulator()       # (zero-filled by the blending forward of this frame: no fill launch here)
        g.dL_dcolor, g.dL_ddepth, g.dL_dmeans2D, g.scratch_acc = _lib.ptr(grad_color), _lib.ptr(grad_depth), _lib.ptr(g_means2D), _lib.ptr(acc)
        epi = _lib.RasterDeformEpilogue()
        epi.activate, epi.Npad = 1, (P + 127) // 128 * 128
        epi.rot_norm, epi.G = _lib.ptr(st.o_norm), _lib.ptr(b.scratch)
        epi.d_xyz, epi.d_scales, epi.d_rotations, epi.d_opacity = b.g.d_xyz, b.g.d_scales, b.g.d_rotations, b.g.d_opacity
        epi.d_shs_dc, epi.d_shs_rest = b.g.d_shs_dc, b.g.d_shs_rest
        epi.shs_dc_stride, epi.shs_rest_stride = st.p.shs_dc_stride, st.p.shs_rest_stride
        epi.assign = 1 if EPILOGUE_ASSIGN else 0
        epi.tile_flags = _tile_flags()      # per-tile non-zero flags behind the packed rows; 2: rows of dead tiles stay unwritten
        if b.zero_range is not None:        # the accumulate-into part of the gradient arena is cleared by the epilogue kernel on the way
            epi.zero_fill, epi.zero_floats = b.zero_range
        g.deform_epilogue = _lib.ctypes.pointer(epi)
        _lib.check(L.fdgs_raster_bwd(_lib.stream_ptr(), p, _lib.ptr(rstate.geom), _lib.ptr(rstate.binning), _lib.ptr(rstate.img),
                                     rstate.capacity, g))
        b.g.packed_rows_ready = epi.tile_flags + 1
        grads = _deformation.backward_run(st, b)       # (d_xyz, d_sc, d_rot, d_op, d_sha, d_shb, None [time], None [aabb], planes..., mlp...)
        return (None, None, None, g_means2D) + grads[:6] + grads[7:]


class _FusedRenderViewsFunction(torch.autograd.Function):
    """The `batch_size` views of one optimizer step (train.py:180-201) behind ONE autograd node: every view's backward accumulates into
    the SAME gradient arena (first view: the epilogue assigns the per-Gaussian identity paths, later views accumulate), so a step pays
    one arena allocation / zero fill and no per-parameter `grad += grad` kernels (the per-view graph costs 40 of them per extra view)."""

    @staticmethod
    def forward(ctx, cfg, times, settings_list, nviews, *tensors):
        means2D = tensors[:nviews]
        xyz, scales, rotations, opacity, sh_a, sh_b, aabb = tensors[nviews:nviews + 7]
        rest = tensors[nviews + 7:]
        want = any(ctx.needs_input_grad)
        dev, P = xyz.device, xyz.shape[0]
        H, W = int(settings_list[0].image_height), int(settings_list[0].image_width)
        colors = torch.empty(nviews, 3, H, W, device=dev)          # every view renders straight into its slice
        radii_all = torch.empty(nviews, P, dtype=torch.int32, device=dev)
        depths = torch.empty(nviews, 1, H, W, device=dev)
        states = []
        for v in range(nviews):
            st = _deformation.forward_impl(cfg, times[v], xyz, scales, rotations, opacity, sh_a, sh_b, None, aabb, rest, want)
            _, _, _, rstate = _rasterizer.rasterize_forward(settings_list[v], st.o_xyz, st.o_sh, None, st.o_op, st.o_sc, st.o_rot, None,
                                                            out=(colors[v], radii_all[v], depths[v]), expect_backward=bool(cfg.get("grad")) and want)
            st.o_xyz = st.o_sh = None
            states.append((st, rstate))
        ctx.states, ctx.nviews = states, nviews
        # the activated outputs each view's backward needs: through save_for_backward (no reference cycle through ctx)
        ctx.save_for_backward(*[t for st, _ in states for t in (st.o_sc, st.o_rot, st.o_op)])
        for st, _ in states:
            st.o_sc = st.o_rot = st.o_op = None
        ctx.mark_non_differentiable(radii_all)
        ctx.set_materialize_grads(False)
        return colors, radii_all, depths

    @staticmethod
    def backward(ctx, grad_colors, grad_radii, grad_depths):
        states, nviews = ctx.states, ctx.nviews
        st0 = states[0][0]
        n_in = 4 + nviews + 7 + len(st0.plane_shapes) + len(st0.keep[0][8])
        if grad_colors is None and grad_depths is None:
            return (None,) * n_in
        L = _lib.lib()
        saved = ctx.saved_tensors
        b = _deformation.backward_prepare(st0, saved[0], saved[1], saved[2], identity_assigned=EPILOGUE_ASSIGN)
        dev = b.d_xyz.device
        f = lambda t: None if t is None else t.float().contiguous()
        g_means2D = [None] * nviews
        keep = []
        # last view first: its saved activations, lists and records are the ones still (partly) resident in the 256 MB Infinity Cache
        for k_, v in enumerate(reversed(range(nviews))):
            st, rstate = states[v]
            p = rstate.params
            P = p.P
            gc = f(grad_colors[v]) if grad_colors is not None else torch.zeros(3, p.H, p.W, device=dev)
            gd = f(grad_depths[v]) if grad_depths is not None else None
            g = _lib.RasterGrads()
            gm = torch.empty(P, 3, device=dev)
            acc, g.scratch_acc_zeroed = rstate.take_accumulator()
            g.dL_dcolor, g.dL_ddepth, g.dL_dmeans2D, g.scratch_acc = _lib.ptr(gc), _lib.ptr(gd), _lib.ptr(gm), _lib.ptr(acc)
            epi = _lib.RasterDeformEpilogue()
            epi.activate, epi.Npad = 1, (P + 127) // 128 * 128
            epi.rot_norm, epi.G = _lib.ptr(st.o_norm), _lib.ptr(b.scratch)
            epi.d_xyz, epi.d_scales, epi.d_rotations, epi.d_opacity = b.g.d_xyz, b.g.d_scales, b.g.d_rotations, b.g.d_opacity
            epi.d_shs_dc, epi.d_shs_rest = b.g.d_shs_dc, b.g.d_shs_rest
            epi.shs_dc_stride, epi.shs_rest_stride = st.p.shs_dc_stride, st.p.shs_rest_stride
            epi.assign = 1 if (EPILOGUE_ASSIGN and k_ == 0) else 0       # later views accumulate into what the first processed one assigned
            epi.tile_flags = _tile_flags()
            g.deform_epilogue = _lib.ctypes.pointer(epi)
            _lib.check(L.fdgs_raster_bwd(_lib.stream_ptr(), p, _lib.ptr(rstate.geom), _lib.ptr(rstate.binning), _lib.ptr(rstate.img),
                                         rstate.capacity, g))
            # this view's deformation backward: same output pointers and scratch (stream-ordered reuse), its own saved activations
            b.g.out_scales, b.g.out_rotations, b.g.out_opacity = _lib.ptr(saved[3 * v]), _lib.ptr(saved[3 * v + 1]), _lib.ptr(saved[3 * v + 2])
            b.g.rot_norm, b.g.saved = _lib.ptr(st.o_norm), _lib.ptr(st.saved_act)
            b.g.packed_rows_ready = epi.tile_flags + 1
            _lib.check(L.fdgs_deform_bwd(_lib.stream_ptr(), st.p, b.g))
            _deformation.note_live_tiles(st, b)
            g_means2D[v] = gm
            keep.append((gc, gd, acc, epi))
        cfg, sh = st0.cfg, st0.shapes
        d_mlp = b.d_mlp
        for h in range(_lib.NUM_HEADS):
            if not cfg["head_on"][h]:
                d_mlp[2 + 4 * h:6 + 4 * h] = [None] * 4
        return (None, None, None, None, *g_means2D, b.d_xyz, b.d_sc.reshape(sh["scales"]), b.d_rot.reshape(sh["rot"]), b.d_op.reshape(sh["op"]),
                b.d_sha.reshape(sh["sh_a"]), None if b.d_shb is None else b.d_shb.reshape(sh["sh_b"]), None, *b.d_planes, *d_mlp)


def render_views(viewpoint_cameras, pc, pipe, bg_color, scaling_modifier=1.0, stage="fine", cam_type=None):
    """The views of one optimizer step rendered behind one autograd node: `[render(cam, ...) for cam in viewpoint_cameras]` of the
    reference's batch loop (train.py:180-198) with the same per-view result dicts, but one gradient arena for the whole step (see
    _FusedRenderViewsFunction).  Falls back to per-view render() wherever the fused fine stage does not apply."""
    cams = list(viewpoint_cameras)
    fusable = (FUSED_BACKWARD and "fine" in stage and isinstance(pc._deformation, _deformation.deform_network)
               and not pipe.compute_cov3D_python and not pipe.convert_SHs_python and len(cams) > 0)
    if not fusable:
        return [render(c, pc, pipe, bg_color, scaling_modifier, None, stage, cam_type) for c in cams]
    means3D = pc.get_xyz
    device = means3D.device
    settings, times = [], []
    for cam in cams:
        if cam_type != "PanopticSports":
            settings.append(GaussianRasterizationSettings(
                image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
                tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color, scale_modifier=scaling_modifier,
                viewmatrix=_dev(cam.world_view_transform, device), projmatrix=_dev(cam.full_proj_transform, device),
                sh_degree=pc.active_sh_degree, campos=_dev(cam.camera_center, device), prefiltered=False, debug=pipe.debug))
            times.append(float(cam.time))
        else:
            settings.append(cam["camera"])
            times.append(float(cam["time"]))
    if len({(s_.image_height, s_.image_width) for s_ in settings}) != 1:
        return [render(c, pc, pipe, bg_color, scaling_modifier, None, stage, cam_type) for c in cams]
    sinks = [_zero_leaf(means3D) for _ in cams]
    net = pc._deformation
    planes, mlp = _deformation._collect(net)
    dn = net.deformation_net
    cfg = dict(C=dn.grid.grid_config[0]["output_coordinate_dim"], L=len(dn.grid.grids), W=dn.W, head_on=_deformation._head_on(dn.args),
               activate=True, save=bool(_deformation.SAVE_ACTIVATIONS and torch.is_grad_enabled()), grad=torch.is_grad_enabled(),
               ordered=_deformation.spatial_order_hint(pc._xyz))
    perm = _implicit_perm(pc, cfg, dn)
    ins = (*sinks, means3D, pc._scaling, pc._rotation, pc._opacity, pc._features_dc, pc._features_rest)
    if perm is not None:       # the set in Hilbert order for this step (gradients come back through the inverse)
        cfg["ordered"] = True
        ins = _deformation.PermuteRows.apply(perm, *ins) if cfg["grad"] else tuple(_deformation.permute_rows(perm, [t.detach() for t in ins]))
    colors, radii, depths = _FusedRenderViewsFunction.apply(cfg, times, settings, len(cams), *ins, dn.grid.aabb, *planes, *mlp)
    if perm is not None:
        radii = torch.stack(_deformation.permute_rows(perm, [radii[v] for v in range(len(cams))], scatter=True))
    return [{"render": colors[v], "viewspace_points": sinks[v], "visibility_filter": radii[v] > 0, "radii": radii[v], "depth": depths[v]}
            for v in range(len(cams))]


_zero_pool = {}


def _zero_leaf(like):
    """A fresh LEAF of zeros shaped like `like` without a fill launch per frame: every frame's leaf is a detached alias of one cached,
    never-written zero buffer per (shape, dtype, device) -- the values are only ever read (the rasterizer ignores them, as the reference's
    does), the gradient arrives in the leaf's own `.grad`.  READ-ONLY by contract: `viewspace_points` of every frame shares this storage, so
    an in-place write through it (`.data`, an op under no_grad) would show up in every later frame's tensor (the reference's is a fresh
    `zeros_like` per frame; nothing in the reference writes to it, train.py:223-225 only reads `.grad`)."""
    key = (tuple(like.shape), like.dtype, like.device)
    z = _zero_pool.get(key)
    if z is None:
        if len(_zero_pool) > 8:
            _zero_pool.clear()
        # (a normal tensor whatever mode the first caller is in: under torch.inference_mode() it would be an inference tensor and the
        # requires_grad_ of a later training frame would raise)
        with torch.inference_mode(False), torch.no_grad():
            z = _zero_pool[key] = torch.zeros(like.shape, dtype=like.dtype, device=like.device)
    if torch.is_inference_mode_enabled():
        return z.detach()                       # (no autograd inside inference mode: nothing will ask for a gradient)
    return z.detach().requires_grad_(True)


def _dev(t, device):
    return t if t.device == device else t.to(device, non_blocking=True)


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, stage="fine", cam_type=None):
    means3D = pc.get_xyz
    device = means3D.device
    # gradient sink for the screen-space means (train.py:223-225 reads .grad of this tensor).  A LEAF of zeros: the reference's
    # `zeros_like(...) + 0` with retain_grad() costs an extra add kernel per frame plus a clone of the gradient in the retain hook;
    # a leaf's .grad is filled by AccumulateGrad directly (same values, same attribute, read the same way by the train loop)
    screenspace_points = _zero_leaf(means3D)
    if cam_type != "PanopticSports":
        tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
        tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
        raster_settings = GaussianRasterizationSettings(
            image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
            tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
            viewmatrix=_dev(viewpoint_camera.world_view_transform, device),
            projmatrix=_dev(viewpoint_camera.full_proj_transform, device), sh_degree=pc.active_sh_degree,
            campos=_dev(viewpoint_camera.camera_center, device), prefiltered=False, debug=pipe.debug)
        frame_time = float(viewpoint_camera.time)
    else:
        raster_settings = viewpoint_camera["camera"]
        frame_time = float(viewpoint_camera["time"])
    means2D = screenspace_points
    opacity = pc._opacity
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc._scaling, pc._rotation

    fused = isinstance(pc._deformation, _deformation.deform_network) and not pipe.compute_cov3D_python
    if "coarse" in stage:
        means3D_final, shs_final = means3D, pc.get_features
        scales_final = None if scales is None else pc.scaling_activation(scales)
        rotations_final = None if rotations is None else pc.rotation_activation(rotations)
        opacity_final = pc.opacity_activation(opacity)
    elif "fine" in stage:
        if fused and FUSED_BACKWARD and override_color is None and not pipe.convert_SHs_python:
            # deformation + rasterizer as one autograd node (see _FusedRenderFunction)
            net = pc._deformation
            planes, mlp = _deformation._collect(net)
            dn = net.deformation_net
            cfg = dict(C=dn.grid.grid_config[0]["output_coordinate_dim"], L=len(dn.grid.grids), W=dn.W,
                       head_on=_deformation._head_on(dn.args), activate=True,
                       save=bool(_deformation.SAVE_ACTIVATIONS and torch.is_grad_enabled()), grad=torch.is_grad_enabled(),
                       ordered=_deformation.spatial_order_hint(pc._xyz))
            perm = _implicit_perm(pc, cfg, dn)
            f_dc, f_rest = pc._features_dc, pc._features_rest
            if perm is not None:       # the set in Hilbert order for this frame (gradients come back through the inverse)
                cfg["ordered"] = True
                if cfg["grad"]:
                    means2D, means3D, scales, rotations, opacity, f_dc, f_rest = _deformation.PermuteRows.apply(
                        perm, means2D, means3D, scales, rotations, opacity, f_dc, f_rest)
                else:
                    means3D, scales, rotations, opacity, f_dc, f_rest = _deformation.permute_rows(
                        perm, [t.detach() for t in (means3D, scales, rotations, opacity, f_dc, f_rest)])
            if not cfg["grad"]:
                # no graph will be recorded (render.py:57-70, evaluation): the two stages called directly -- autograd.Function.apply costs
                # 0.05 ms per frame for its 46 inputs even when it has nothing to record
                st = _deformation.forward_impl(cfg, frame_time, means3D, scales, rotations, opacity, f_dc, f_rest,
                                               None, dn.grid.aabb, (*planes, *mlp), False)
                rendered_image, radii, depth, rstate = _rasterizer.rasterize_forward(raster_settings, st.o_xyz, st.o_sh, None, st.o_op, st.o_sc,
                                                                                     st.o_rot, None, expect_backward=False)
                vis = rstate.visibility
                if perm is not None:
                    radii, = _deformation.permute_rows(perm, [radii], scatter=True)
                    vis = radii > 0
                return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": vis,
                        "radii": radii, "depth": depth}
            rendered_image, radii, depth, vis = _FusedRenderFunction.apply(
                cfg, frame_time, raster_settings, means2D, means3D, scales, rotations, opacity, f_dc, f_rest,
                dn.grid.aabb, *planes, *mlp)
            if perm is not None:
                radii, = _deformation.permute_rows(perm, [radii], scatter=True)
                vis = radii > 0
            return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": vis,
                    "radii": radii, "depth": depth}
        elif fused:
            means3D_final, scales_final, rotations_final, opacity_final, shs_final = _deformation.deform(
                pc._deformation, means3D, scales, rotations, opacity, shs_dc=pc._features_dc, shs_rest=pc._features_rest,
                time=frame_time, activate=True)
        else:
            time = torch.tensor(frame_time).to(device).repeat(means3D.shape[0], 1)
            means3D_final, scales_final, rotations_final, opacity_final, shs_final = pc._deformation(
                means3D, scales, rotations, opacity, pc.get_features, time)
            scales_final = None if scales_final is None else pc.scaling_activation(scales_final)
            rotations_final = None if rotations_final is None else pc.rotation_activation(rotations_final)
            opacity_final = pc.opacity_activation(opacity_final)
    else:
        raise NotImplementedError

    colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            # python SH path of the reference (gaussian_renderer/__init__.py:106-111): basis polynomial of
            # utils/sh_utils.py:57-112 evaluated with torch ops, colours handed to the rasterizer precomputed
            from .sh import eval_sh
            feats = pc.get_features
            shs_view = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - _dev(viewpoint_camera.camera_center, device).repeat(feats.shape[0], 1)
            dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
            shs_final = None
    else:
        colors_precomp = override_color
        shs_final = None

    rasterizer = GaussianRasterizer(raster_settings=raster_settings)       # (an nn.Module: only built on the paths that call it)
    rendered_image, radii, depth = rasterizer(
        means3D=means3D_final, means2D=means2D, shs=shs_final, colors_precomp=colors_precomp, opacities=opacity_final,
        scales=scales_final, rotations=rotations_final, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth": depth}
