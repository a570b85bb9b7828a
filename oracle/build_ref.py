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
    # the train step (round 5): GaussianModel (training_setup, update_learning_rate, add_densification_stats, densify, prune, reset_opacity,
    # compute_regulation), the regulariser it calls, the helpers it imports, the loss / metric functions of train.py:201-205
    "scene/gaussian_model.py",
    "scene/regulation.py",
    "utils/general_utils.py",
    "utils/system_utils.py",
    "utils/loss_utils.py",
    "utils/image_utils.py",
]
# sha256 of the reference sources this recipe was written against (`scene/dataset_readers.py`: only `setup_camera` is taken from it).  A tree
# whose files differ is NOT compiled (the byte code is executed in-process by the tests): set FDGS_REFERENCE_UNPINNED=1 to override knowingly.
PINNED_SHA256 = {
    "gaussian_renderer/__init__.py": "74a53789013150843efb0fde2a71b4af36ac44bfaa2cb9e5ee807a5adb277ff9",
    "scene/deformation.py": "40908bf3746f1a8257afbd253ebc044dc0b3b7de6936093af97d6668ef3bf668",
    "scene/hexplane.py": "aed9fddd27e2a0b328def4e2958536c166d0c046ce5f1dd0d084548d7c8651d7",
    "scene/grid.py": "ff8c169238b4d47ab506b88cb3da897efda471e8f72919e7b2531b9df916ae0e",
    "utils/graphics_utils.py": "264edb7a8a7ff2ee13b12139f081af4c408417e33cbb16b123b5b118a3884f25",
    "utils/sh_utils.py": "7d1ff267546390635e6d1f68c4f88c4a8b052482d5c9be1d32f06bc69e9a96e7",
    "scene/dataset_readers.py": "e90cee1484df2dc456d21f18173d8b34fde87b1da749a809132c0ca5990aec06",
    "scene/gaussian_model.py": "73aba8fa6d8d0fc523b38eba780b6b35d225074ac00f02ce1caa2444a403258d",
    "scene/regulation.py": "990293a020cf43a1d7ef9df66941dc8535ad8d2f59c91374673102b4be8c9bf1",
    "utils/general_utils.py": "a2136f28ccd481e25da1f4454f6f3a8b633be2a51786ec24c30d9dd052adbeec",
    "utils/system_utils.py": "be01c02d3118c5d53808bc04efccd00c35a705a43a87c3583ea2f4752bc48553",
    "utils/loss_utils.py": "980f64ffa391d9b673a9412d0f4b38ff6b0aac3618e237b079f8a5107c33810f",
    "utils/image_utils.py": "a159f2e2767c96f1b4e408dd1f49b7c2c3195cfa0a839a567206cdc8c60bd4ba",
}


def _check_pin(rel):
    import hashlib
    h = hashlib.sha256(open(os.path.join(REF, rel), "rb").read()).hexdigest()
    if h != PINNED_SHA256[rel] and os.environ.get("FDGS_REFERENCE_UNPINNED") != "1":
        raise RuntimeError(f"oracle/build_ref.py: {REF}/{rel} has sha256 {h}, the recipe is pinned to {PINNED_SHA256[rel]}: refusing to byte-compile "
                           "(and later execute) an unknown file; FDGS_REFERENCE_UNPINNED=1 overrides")


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
            _check_pin(rel)
            # dfile: what tracebacks show -- the reference path, so a failure inside points at the reference line
            py_compile.compile(src, cfile=dst, dfile="reference:" + rel, doraise=True,
                               invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        written.append(dst)
    # setup_camera only (scene/dataset_readers.py:485-508)
    src = os.path.join(REF, "scene", "dataset_readers.py")
    dst = os.path.join(OUT, "scene", "panoptic_setup_camera.pyc")
    if force or not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
        _check_pin("scene/dataset_readers.py")
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
