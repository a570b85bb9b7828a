"""CPU-side checks of the C-ABI: libfdgs.so loads (hipcc cross-compiled, no GPU needed) and exports every symbol
include/fdgs.h declares; host-only size queries work; struct mirrors match the header's field order."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    _lib = importlib.import_module("4dgaussians_amd._lib")
    if not os.path.exists(_lib.LIB_PATH):
        importlib.import_module("4dgaussians_amd.build").build()
    return _lib


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "fdgs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fdgs_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(L):
    lib = L.lib()
    names = _header_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
        assert n in L.SYMBOLS, f"{n} declared in fdgs.h but not bound in _lib.SYMBOLS"
    assert sorted(L.SYMBOLS) == names
    assert lib.fdgs_abi_version() == 6


def test_host_only_size_queries(L):
    lib = L.lib()
    n = ctypes.c_size_t()
    assert lib.fdgs_geom_bytes(300000, n) == 0 and n.value > 300000 * (4 + 48 + 24)
    assert lib.fdgs_img_bytes(1352, 1014, n) == 0 and n.value >= 1352 * 1014 * 8 + 85 * 64 * 8
    assert lib.fdgs_binning_bytes(3000000, 1352, 1014, n) == 0 and n.value >= 3000000 * 16
    assert lib.fdgs_geom_bytes(-1, n) != 0
    assert b"bad" in lib.fdgs_last_error()


def test_struct_field_order_matches_header(L):
    src = open(os.path.join(ROOT, "include", "fdgs.h")).read()

    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])*\s*$", part.strip())
                out.append(m.group(1))
        return out
    for cname, cls in (("fdgs_raster_params", L.RasterParams), ("fdgs_raster_grads", L.RasterGrads),
                       ("fdgs_deform_params", L.DeformParams), ("fdgs_deform_out", L.DeformOut),
                       ("fdgs_deform_grads", L.DeformGrads), ("fdgs_raster_deform_epilogue", L.RasterDeformEpilogue)):
        assert fields(cname) == [f[0] for f in cls._fields_], cname
