"""ctypes front-end of oracle/raster_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It restates the un-vendored `diff_gaussian_rasterization` CUDA extension the reference calls at
gaussian_renderer/__init__.py:120-128; parity is UNPINNED by reference golden vectors (see the header
of raster_oracle.c and DESIGN.md).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIBS = {}

FIELDS = {
    "depth": (0, "real", 1), "xy": (1, "real", 2), "conic_opacity": (2, "real", 4), "rgb": (3, "real", 3),
    "cov3D": (4, "real", 6), "radii": (5, np.int32, 1), "tiles_touched": (6, np.int32, 1),
    "clamped": (7, np.uint8, 3), "rect": (13, np.uint32, 4),
}


def build(force=False):
    """Compile the f32 and f64 variants of the C restatement (gcc only)."""
    os.makedirs(_BUILD, exist_ok=True)
    src = os.path.join(_HERE, "raster_oracle.c")
    for name, flags in (("f32", []), ("f64", ["-DORACLE_DOUBLE"])):
        out = os.path.join(_BUILD, f"libraster_oracle_{name}.so")
        if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", *flags, src,
                                   "-o", out, "-lm"])


def _lib(dtype):
    name = "f64" if np.dtype(dtype) == np.float64 else "f32"
    if name not in _LIBS:
        build()
        lib = ctypes.CDLL(os.path.join(_BUILD, f"libraster_oracle_{name}.so"))
        lib.oracle_raster_forward.restype = ctypes.c_void_p
        lib.oracle_raster_field.restype = ctypes.c_void_p
        lib.oracle_raster_field.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.oracle_raster_num_rendered.argtypes = [ctypes.c_void_p]
        lib.oracle_raster_free.argtypes = [ctypes.c_void_p]
        lib.oracle_raster_set_threads.argtypes = [ctypes.c_int]
        lib.oracle_raster_set_threads.restype = ctypes.c_int
        _LIBS[name] = lib
    return _LIBS[name]


def set_threads(n, dtype=np.float32):
    """OpenMP thread count of the C restatement; returns the count in effect."""
    return int(_lib(dtype).oracle_raster_set_threads(int(n)))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class RasterOracle:
    """One forward pass; keeps the state needed by backward()."""

    def __init__(self, *, means3D, opacities, viewmatrix, projmatrix, campos, bg, image_height, image_width, tanfovx,
                 tanfovy, sh_degree=0, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                 scale_modifier=1.0, dtype=np.float32):
        self.dtype = dt = np.dtype(dtype)
        self.real = ctypes.c_double if dt == np.float64 else ctypes.c_float
        c = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=dt))
        self.means3D = c(means3D).reshape(-1, 3)
        self.P = P = self.means3D.shape[0]
        self.shs = c(shs)
        self.M = 0 if shs is None else self.shs.reshape(P, -1, 3).shape[1]
        self.colors_precomp, self.opacities = c(colors_precomp), c(opacities).reshape(-1)
        self.scales, self.rotations, self.cov3D_precomp = c(scales), c(rotations), c(cov3D_precomp)
        self.view, self.proj = c(viewmatrix).reshape(16), c(projmatrix).reshape(16)
        self.campos, self.bg = c(campos).reshape(3), c(bg).reshape(3)
        self.H, self.W, self.D = int(image_height), int(image_width), int(sh_degree)
        assert (self.shs is None) != (self.colors_precomp is None)
        assert (self.scales is None) == (self.rotations is None)
        assert (self.scales is None) != (self.cov3D_precomp is None)
        self.lib = _lib(dt)
        self.color = np.zeros((3, self.H, self.W), dt)
        self.depth = np.zeros((1, self.H, self.W), dt)
        self.radii = np.zeros(P, np.int32)
        r = self.real
        self.handle = self.lib.oracle_raster_forward(
            ctypes.c_int(P), ctypes.c_int(self.D), ctypes.c_int(self.M), _ptr(self.bg), ctypes.c_int(self.W),
            ctypes.c_int(self.H), _ptr(self.means3D), _ptr(self.shs), _ptr(self.colors_precomp), _ptr(self.opacities),
            _ptr(self.scales), r(scale_modifier), _ptr(self.rotations), _ptr(self.cov3D_precomp), _ptr(self.view),
            _ptr(self.proj), _ptr(self.campos), r(tanfovx), r(tanfovy), ctypes.c_int(0), _ptr(self.color),
            _ptr(self.depth), _ptr(self.radii))
        self.handle = ctypes.c_void_p(self.handle)
        self.num_rendered = self.lib.oracle_raster_num_rendered(self.handle)

    def field(self, name):
        idx, ty, width = FIELDS[name]
        ty = self.dtype if ty == "real" else np.dtype(ty)
        p = self.lib.oracle_raster_field(self.handle, idx)
        n = self.P * width
        buf = (ctypes.c_char * (n * ty.itemsize)).from_address(p)
        a = np.frombuffer(buf, dtype=ty).copy()
        return a.reshape(self.P, width) if width > 1 else a

    def pairs(self):
        R = self.num_rendered
        out = []
        for idx in (8, 9):
            p = self.lib.oracle_raster_field(self.handle, idx)
            buf = (ctypes.c_char * (R * 4)).from_address(p) if R else b""
            out.append(np.frombuffer(buf, dtype=np.uint32).copy())
        return out

    def image_state(self):
        n = self.H * self.W
        p = self.lib.oracle_raster_field(self.handle, 11)
        fT = np.frombuffer((ctypes.c_char * (n * self.dtype.itemsize)).from_address(p), dtype=self.dtype).copy()
        p = self.lib.oracle_raster_field(self.handle, 12)
        nc = np.frombuffer((ctypes.c_char * (n * 4)).from_address(p), dtype=np.uint32).copy()
        return fT.reshape(self.H, self.W), nc.reshape(self.H, self.W)

    def stage_tensors(self):
        """The forward's stage values under the names tools/dump_cuda_golden.py::decode_stage_tensors uses for the upstream
        extension's scratch buffers (tests/test_cuda_golden.py compares the two)."""
        fT, nc = self.image_state()
        return {"internal_radii": self.radii.copy(), "depths": self.field("depth"), "tiles_touched": self.field("tiles_touched").astype(np.uint32),
                "means2D": self.field("xy"), "conic_opacity": self.field("conic_opacity"), "rgb": self.field("rgb"), "cov3D": self.field("cov3D"),
                "num_rendered": int(self.num_rendered), "accum_alpha": fT.reshape(-1), "n_contrib": nc.reshape(-1)}

    def backward(self, dL_dcolor, dL_ddepth=None):
        dt, P = self.dtype, self.P
        dc = np.ascontiguousarray(np.asarray(dL_dcolor, dtype=dt)).reshape(3, self.H, self.W)
        dd = None if dL_ddepth is None else np.ascontiguousarray(np.asarray(dL_ddepth, dtype=dt)).reshape(self.H, self.W)
        g = dict(means2D=np.zeros((P, 3), dt), means3D=np.zeros((P, 3), dt), shs=np.zeros((P, max(self.M, 1), 3), dt),
                 colors=np.zeros((P, 3), dt), opacities=np.zeros(P, dt), scales=np.zeros((P, 3), dt),
                 rotations=np.zeros((P, 4), dt), cov3D=np.zeros((P, 6), dt))
        self.lib.oracle_raster_backward(self.handle, _ptr(dc), _ptr(dd), _ptr(g["means2D"]), _ptr(g["means3D"]),
                                        _ptr(g["shs"]), _ptr(g["colors"]), _ptr(g["opacities"]), _ptr(g["scales"]),
                                        _ptr(g["rotations"]), _ptr(g["cov3D"]))
        if self.M == 0:
            g["shs"] = None
        return g

    def close(self):
        if self.handle:
            self.lib.oracle_raster_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
