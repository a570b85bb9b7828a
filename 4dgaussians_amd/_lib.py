"""ctypes binding of libfdgs.so (the C-ABI declared in include/fdgs.h).

There is NO CPU or PyTorch fallback: if the HIP library is missing or cannot be loaded every product entry point
raises.  (The oracle under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint32, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FDGS_LIB") or os.path.join(HERE, "libfdgs.so")   # FDGS_LIB: A/B a development build
MAX_LEVELS, NUM_HEADS = 4, 5


class RasterParams(Structure):
    _fields_ = [("P", c_int), ("sh_degree", c_int), ("sh_coeffs", c_int), ("W", c_int), ("H", c_int),
                ("tanfovx", c_float), ("tanfovy", c_float), ("scale_modifier", c_float), ("prefiltered", c_int),
                ("debug", c_int), ("bg", c_void_p), ("viewmatrix", c_void_p), ("projmatrix", c_void_p),
                ("campos", c_void_p), ("means3D", c_void_p), ("shs", c_void_p), ("colors_precomp", c_void_p),
                ("opacities", c_void_p), ("scales", c_void_p), ("rotations", c_void_p), ("cov3D_precomp", c_void_p),
                ("visibility", c_void_p), ("acc_zero", c_void_p)]


class RasterDeformEpilogue(Structure):
    _fields_ = [("activate", c_int), ("Npad", c_int), ("rot_norm", c_void_p), ("G", c_void_p), ("d_xyz", c_void_p),
                ("d_scales", c_void_p), ("d_rotations", c_void_p), ("d_opacity", c_void_p), ("d_shs_dc", c_void_p),
                ("d_shs_rest", c_void_p), ("shs_dc_stride", c_int), ("shs_rest_stride", c_int), ("assign", c_int), ("tile_flags", c_int),
                ("zero_fill", c_void_p), ("zero_floats", c_size_t)]


class RasterGrads(Structure):
    _fields_ = [("dL_dcolor", c_void_p), ("dL_ddepth", c_void_p), ("dL_dmeans2D", c_void_p), ("dL_dmeans3D", c_void_p),
                ("dL_dopacity", c_void_p), ("dL_dcolors", c_void_p), ("dL_dsh", c_void_p), ("dL_dscales", c_void_p),
                ("dL_drotations", c_void_p), ("dL_dcov3D", c_void_p), ("scratch_acc", c_void_p),
                ("deform_epilogue", POINTER(RasterDeformEpilogue)), ("scratch_acc_zeroed", c_int)]


class DeformParams(Structure):
    _fields_ = [("N", c_int), ("C", c_int), ("L", c_int), ("W", c_int), ("head_on", c_int * NUM_HEADS),
                ("activate", c_int), ("res", (c_int * 4) * MAX_LEVELS), ("planes", (c_void_p * 6) * MAX_LEVELS),
                ("aabb", c_float * 6), ("w0", c_void_p), ("b0", c_void_p), ("w1", c_void_p * NUM_HEADS),
                ("b1", c_void_p * NUM_HEADS), ("w2", c_void_p * NUM_HEADS), ("b2", c_void_p * NUM_HEADS),
                ("xyz", c_void_p), ("scales", c_void_p), ("rotations", c_void_p), ("opacity", c_void_p),
                ("shs_dc", c_void_p), ("shs_rest", c_void_p), ("shs_dc_stride", c_int), ("shs_rest_stride", c_int),
                ("time", c_void_p), ("time_scalar", c_float)]


class DeformOut(Structure):
    _fields_ = [("xyz", c_void_p), ("scales", c_void_p), ("rotations", c_void_p), ("opacity", c_void_p), ("shs", c_void_p),
                ("rot_norm", c_void_p), ("saved", c_void_p), ("packed", c_void_p)]


class DeformGrads(Structure):
    _fields_ = [("g_xyz", c_void_p), ("g_scales", c_void_p), ("g_rotations", c_void_p), ("g_opacity", c_void_p),
                ("g_shs", c_void_p), ("out_scales", c_void_p), ("out_rotations", c_void_p), ("out_opacity", c_void_p),
                ("rot_norm", c_void_p), ("d_xyz", c_void_p), ("d_scales", c_void_p), ("d_rotations", c_void_p), ("d_opacity", c_void_p),
                ("d_shs_dc", c_void_p), ("d_shs_rest", c_void_p), ("d_planes", (c_void_p * 6) * MAX_LEVELS),
                ("d_w0", c_void_p), ("d_b0", c_void_p), ("d_w1", c_void_p * NUM_HEADS), ("d_b1", c_void_p * NUM_HEADS),
                ("d_w2", c_void_p * NUM_HEADS), ("d_b2", c_void_p * NUM_HEADS), ("scratch", c_void_p), ("saved", c_void_p),
                ("packed_rows_ready", c_int), ("spatially_ordered", c_int)]


class RegPlane(Structure):
    _fields_ = [("plane", c_void_p), ("grad_opt", c_void_p), ("H", c_int), ("W", c_int), ("C", c_int),
                ("w_smooth", c_float), ("w_l1", c_float)]


class AdamTensor(Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("n", c_size_t),
                ("lr", c_float), ("step", c_int)]


NGROUPS = 6


class GaussiansIn(Structure):
    _fields_ = [("N", c_int), ("width", c_int * NGROUPS), ("param", c_void_p * NGROUPS), ("exp_avg", c_void_p * NGROUPS),
                ("exp_avg_sq", c_void_p * NGROUPS), ("deformation_table", c_void_p), ("xyz_gradient_accum", c_void_p),
                ("denom", c_void_p), ("max_radii2D", c_void_p), ("deformation_accum", c_void_p)]


class GaussiansOut(Structure):
    _fields_ = [("param", c_void_p * NGROUPS), ("exp_avg", c_void_p * NGROUPS), ("exp_avg_sq", c_void_p * NGROUPS),
                ("deformation_table", c_void_p), ("xyz_gradient_accum", c_void_p), ("denom", c_void_p),
                ("max_radii2D", c_void_p), ("deformation_accum", c_void_p)]


class RowArray(Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("width", c_int)]


MAX_ROW_ARRAYS = 8

# every symbol include/fdgs.h declares: (restype, argtypes)
SYMBOLS = {
    "fdgs_last_error": (c_char_p, []),
    "fdgs_abi_version": (c_int, []),
    "fdgs_device_arch": (c_int, [c_int, c_char_p, c_size_t]),
    "fdgs_tuning_set": (c_int, [c_char_p, c_int]),
    "fdgs_tuning_get": (c_int, [c_char_p, POINTER(c_int)]),
    "fdgs_tuning_reset": (c_int, []),
    "fdgs_timing_enable": (c_int, [c_int]),
    "fdgs_timing_report": (c_int, [c_char_p, c_size_t, c_int]),
    "fdgs_geom_bytes": (c_int, [c_int, POINTER(c_size_t)]),
    "fdgs_img_bytes": (c_int, [c_int, c_int, POINTER(c_size_t)]),
    "fdgs_binning_bytes": (c_int, [c_uint32, c_int, c_int, POINTER(c_size_t)]),
    "fdgs_preprocess_fwd": (c_int, [c_void_p, POINTER(RasterParams), c_void_p, c_void_p]),
    "fdgs_bin_prepare": (c_int, [c_void_p, POINTER(RasterParams), c_void_p, c_void_p]),
    "fdgs_bin_sort": (c_int, [c_void_p, POINTER(RasterParams), c_void_p, c_void_p, c_void_p, c_uint32]),
    "fdgs_render_fwd": (c_int, [c_void_p, POINTER(RasterParams), c_void_p, c_void_p, c_void_p, c_uint32, c_void_p, c_void_p]),
    "fdgs_raster_fwd_capacity": (c_int, [c_void_p, POINTER(RasterParams), c_void_p, c_void_p, c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdgs_pair_count_wait": (c_int, [c_void_p, c_void_p, POINTER(c_uint32)]),
    "fdgs_raster_bwd": (c_int, [c_void_p, POINTER(RasterParams), c_void_p, c_void_p, c_void_p, c_uint32, POINTER(RasterGrads)]),
    "fdgs_mark_visible": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdgs_geom_field": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "fdgs_binning_field": (c_int, [c_void_p, c_uint32, c_int, c_int, c_int, POINTER(c_void_p)]),
    "fdgs_img_field": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_void_p)]),
    "fdgs_deform_saved_bytes": (c_int, [POINTER(DeformParams), POINTER(c_size_t)]),
    "fdgs_deform_pack_bytes": (c_int, [POINTER(DeformParams), POINTER(c_size_t)]),
    "fdgs_deform_fwd": (c_int, [c_void_p, POINTER(DeformParams), POINTER(DeformOut)]),
    "fdgs_deform_bwd_scratch_bytes": (c_int, [POINTER(DeformParams), POINTER(c_size_t)]),
    "fdgs_deform_bwd": (c_int, [c_void_p, POINTER(DeformParams), POINTER(DeformGrads)]),
    "fdgs_deform_bwd_live_tiles": (c_int, [c_void_p, POINTER(DeformParams), c_void_p, POINTER(c_uint32)]),
    "fdgs_l1_stats": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "fdgs_l1_stats_scratch_bytes": (c_int, [POINTER(c_size_t)]),
    "fdgs_l1_stats_assign": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "fdgs_image_loss_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdgs_image_loss_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p]),
    "fdgs_plane_regulation": (c_int, [c_void_p, c_int, POINTER(RegPlane), c_float, c_void_p, c_void_p]),
    "fdgs_adam_step": (c_int, [c_void_p, c_int, POINTER(AdamTensor), c_double, c_double, c_double]),
    "fdgs_knn3_mean_dist2": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "fdgs_densification_stats": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "fdgs_densify_scratch_bytes": (c_int, [c_int, POINTER(c_size_t)]),
    "fdgs_densify_plan": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_float, c_float, c_float, c_float, c_float, c_void_p, POINTER(c_uint32)]),
    "fdgs_densify_apply": (c_int, [c_void_p, POINTER(GaussiansIn), POINTER(GaussiansOut), c_void_p, c_void_p]),
    "fdgs_permute_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, POINTER(RowArray), c_int]),
}

ABI_VERSION = 6       # what this Python host was written against (include/fdgs.h); checked when the library is loaded

_lib = None


class FdgsError(RuntimeError):
    pass


def lib():
    """Load libfdgs.so once; fail loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FdgsError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the render path.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        have = l.fdgs_abi_version()
        if have != ABI_VERSION:     # structs grow at their end between versions: a stale .so would read past what this host fills in
            raise FdgsError(f"{LIB_PATH} has ABI {have}, this Python host needs ABI {ABI_VERSION}: rebuild it "
                            "(python -c 'import __graft_entry__ as g; g.build()')")
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise FdgsError(f"libfdgs error {rc}: {lib().fdgs_last_error().decode(errors='replace')}")


def tuning_set(name, value):
    """Set one of the library's development knobs (include/fdgs.h: fdgs_tuning_set).  Process-global; set between frames."""
    check(lib().fdgs_tuning_set(name.encode(), int(value)))


def tuning_get(name):
    v = c_int()
    check(lib().fdgs_tuning_get(name.encode(), v))
    return v.value


class tuning:
    """`with _lib.tuning(skip_dead=0, d1_form=32): ...` -- knobs set for the block, previous values restored after it (tests, bench A/B legs)."""

    def __init__(self, **knobs):
        self.knobs, self.old = knobs, {}

    def __enter__(self):
        for k, v in self.knobs.items():
            self.old[k] = tuning_get(k)
            tuning_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            tuning_set(k, v)
        return False


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())
