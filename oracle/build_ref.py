"""Recipe for oracle/_ref/ -- the REAL reference, byte-compiled -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference's render path is Python (gaussian_renderer/__init__.py:18-138 `render()`, scene/deformation.py:161-216 `deform_network`,
scene/hexplane.py, scene/grid.py, utils/graphics_utils.py, utils/sh_utils.py) around an un-vendored CUDA rasterizer.  `/root/reference` does
not exist on the GPU box and its sources are never copied into this repository; what CAN travel is a build product, like a `.so` compiled
from C sources where they lie: this script byte-compiles those reference files from `/root/reference` into `oracle/_ref/**/*.pyc`
(git-ignored, shipped with the gpurun snapshot next to libfdgs.so).  The GPU tests import them sourceless (oracle/ref_modules.py) and run
the reference's OWN render() and the reference's OWN deform_network over this repository's `diff_gaussian_rasterization` shim on the MI355X.

Also extracted: `setup_camera` of scene/dataset_readers.py:485-508 (the PanopticSports construction of the rasterizer settings) -- that one
function only, compiled from the module's AST (the module itself imports PIL / plyfile / the dataset stack).

    python -m oracle.build_ref            # needs /root/reference; __graft_entry__.build() calls it when the tree is present
"""
import ast
import importlib.util
import marshal
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("FDGS_REFERENCE_ROOT", "/root/reference")

FILES = [
    "gaussian_renderer/__init__.py",      # render()
    "scene/deformation.py",               # deform_network / Deformation
    "scene/hexplane.py",
    "scene/grid.py",
    "utils/graphics_utils.py",
    "utils/sh_utils.py",
]


def have_reference():
    return os.path.isfile(os.path.join(REF, "gaussian_renderer", "__init__.py"))


def build(force=False):
    """Byte-compile the reference's render-path modules into oracle/_ref/.  Returns the list of files written (empty when the reference
    tree is absent: the GPU box uses what the build container shipped)."""
    if not have_reference():
        return []
    written = []
    for rel in FILES:
        src = os.path.join(REF, rel)
        dst = os.path.join(OUT, rel[:-3] + ".pyc")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        if force or not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            # dfile: what tracebacks show -- the reference path, so a failure inside points at the reference line
            py_compile.compile(src, cfile=dst, dfile="reference:" + rel, doraise=True,
                               invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        written.append(dst)
    # setup_camera only (scene/dataset_readers.py:485-508)
    src = os.path.join(REF, "scene", "dataset_readers.py")
    dst = os.path.join(OUT, "scene", "panoptic_setup_camera.pyc")
    if force or not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
        tree = ast.parse(open(src).read(), filename="reference:scene/dataset_readers.py")
        fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "setup_camera"]
        assert len(fn) == 1, "scene/dataset_readers.py: setup_camera not found"
        mod = ast.Module(body=[ast.Import(names=[ast.alias(name="torch")]), fn[0]], type_ignores=[])
        ast.fix_missing_locations(mod)
        code = compile(mod, "reference:scene/dataset_readers.py", "exec")
        with open(dst, "wb") as f:                     # an unchecked hash-based pyc: magic, flags = 1, 8 zero bytes, marshalled code
            f.write(importlib.util.MAGIC_NUMBER + (1).to_bytes(4, "little") + b"\0" * 8 + marshal.dumps(code))
    written.append(dst)
    return written


if __name__ == "__main__":
    w = build(force="--force" in sys.argv)
    print("\n".join(w) if w else f"no reference tree at {REF}: nothing built")
