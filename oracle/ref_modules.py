"""Sourceless import of the byte-compiled reference modules in oracle/_ref/ (built by oracle/build_ref.py from /root/reference) -- TEST
INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

`load()` returns a namespace with the reference's OWN objects:
    render          gaussian_renderer/__init__.py:18   (its `diff_gaussian_rasterization` import resolves to THIS repository's shim package)
    deform_network  scene/deformation.py:161
    eval_sh         utils/sh_utils.py:57
    setup_camera    scene/dataset_readers.py:485       (PanopticSports rasterizer settings)
The reference imports `scene.gaussian_model.GaussianModel` (an annotation only: the class pulls open3d / plyfile / simple_knn) and
`tkinter.W` (unused): both are stubbed.
"""
import importlib
import importlib.machinery
import importlib.util
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
ROOT = os.path.dirname(HERE)


def available():
    from . import build_ref
    if build_ref.have_reference():
        build_ref.build()
    return os.path.isfile(os.path.join(OUT, "gaussian_renderer", "__init__.pyc"))


def _pyc(name, rel, package_path=None):
    if name in sys.modules and getattr(sys.modules[name], "__fdgs_ref__", False):
        return sys.modules[name]
    path = os.path.join(OUT, rel)
    loader = importlib.machinery.SourcelessFileLoader(name, path)
    spec = importlib.util.spec_from_loader(name, loader, is_package=package_path is not None)
    mod = importlib.util.module_from_spec(spec)
    mod.__fdgs_ref__ = True
    if package_path is not None:
        mod.__path__ = [package_path]
    sys.modules[name] = mod
    loader.exec_module(mod)
    return mod


def _pkg(name, path):
    m = sys.modules.get(name)
    if m is None or not hasattr(m, "__path__"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    if path not in list(m.__path__):
        m.__path__ = list(m.__path__) + [path]
    return m


def load():
    if not available():
        raise FileNotFoundError("oracle/_ref/ is not built (python -m oracle.build_ref, where /root/reference exists)")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)                   # `diff_gaussian_rasterization` = the shim package at the repository root
    shim = importlib.import_module("diff_gaussian_rasterization")
    assert os.path.dirname(os.path.dirname(os.path.abspath(shim.__file__))) == ROOT, "diff_gaussian_rasterization does not resolve to the shim"
    if "tkinter" not in sys.modules:
        try:
            import tkinter  # noqa: F401
        except Exception:
            tk = types.ModuleType("tkinter")
            tk.W = "w"
            sys.modules["tkinter"] = tk
    _pkg("scene", os.path.join(OUT, "scene"))
    _pkg("utils", os.path.join(OUT, "utils"))
    if "scene.gaussian_model" not in sys.modules:
        gm = types.ModuleType("scene.gaussian_model")
        gm.GaussianModel = type("GaussianModel", (), {})       # annotation only (gaussian_renderer/__init__.py:18)
        gm.__fdgs_stub__ = True
        sys.modules["scene.gaussian_model"] = gm
    ns = types.SimpleNamespace()
    # modules other tests may already have imported from the reference SOURCES (same code) are reused as they are
    gu = sys.modules.get("utils.graphics_utils") or _pyc("utils.graphics_utils", "utils/graphics_utils.pyc")
    su = sys.modules.get("utils.sh_utils") or _pyc("utils.sh_utils", "utils/sh_utils.pyc")
    for name in ("hexplane", "grid", "deformation"):
        if f"scene.{name}" not in sys.modules:
            _pyc(f"scene.{name}", f"scene/{name}.pyc")
    gr = _pyc("_fdgs_ref_gaussian_renderer", "gaussian_renderer/__init__.pyc", package_path=os.path.join(OUT, "gaussian_renderer"))
    sc = _pyc("_fdgs_ref_panoptic_setup_camera", "scene/panoptic_setup_camera.pyc")
    ns.render, ns.deform_network = gr.render, sys.modules["scene.deformation"].deform_network
    ns.eval_sh, ns.setup_camera, ns.graphics_utils = su.eval_sh, sc.setup_camera, gu
    ns.render_module = gr
    return ns


def load_train():
    """`load()` plus the reference's train-step objects (round 5):
        GaussianModel        scene/gaussian_model.py:27   (training_setup :165, update_learning_rate :197, add_densification_stats :516, densify :495,
                                                           prune :481, reset_opacity :269, compute_regulation :576)
        l1_loss, ssim        utils/loss_utils.py:20,40    psnr  utils/image_utils.py:17    get_expon_lr_func  utils/general_utils.py:37
    Its imports of open3d / plyfile / lpips (point-cloud files, the perceptual loss: not on this path) are stubbed; `simple_knn._C` resolves
    to this repository's shim package (csrc/knn.hip)."""
    ns = load()
    for name, attrs in (("open3d", {}), ("plyfile", {"PlyData": object, "PlyElement": object}), ("lpips", {})):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                m = types.ModuleType(name)
                for k, v in attrs.items():
                    setattr(m, k, v)
                sys.modules[name] = m
    knn = importlib.import_module("simple_knn._C")
    assert os.path.dirname(os.path.dirname(os.path.abspath(knn.__file__))) == ROOT, "simple_knn does not resolve to the shim"
    for name in ("general_utils", "system_utils", "loss_utils", "image_utils"):
        if f"utils.{name}" not in sys.modules:
            _pyc(f"utils.{name}", f"utils/{name}.pyc")
    if "scene.regulation" not in sys.modules:
        _pyc("scene.regulation", "scene/regulation.pyc")
    gm = sys.modules.get("scene.gaussian_model")
    if gm is None or getattr(gm, "__fdgs_stub__", False):
        sys.modules.pop("scene.gaussian_model", None)
        gm = _pyc("scene.gaussian_model", "scene/gaussian_model.pyc")
    ns.GaussianModel = gm.GaussianModel
    ns.l1_loss, ns.ssim = sys.modules["utils.loss_utils"].l1_loss, sys.modules["utils.loss_utils"].ssim
    ns.psnr = sys.modules["utils.image_utils"].psnr
    ns.get_expon_lr_func = sys.modules["utils.general_utils"].get_expon_lr_func
    return ns
