"""The reference's OWN `render()` (gaussian_renderer/__init__.py:18-138) and the reference's OWN `deform_network`
(scene/deformation.py:161-216) executed over this repository's `diff_gaussian_rasterization` shim.

The reference sources cannot travel to the GPU box and are never copied into the repository; oracle/build_ref.py byte-compiles them from
/root/reference into oracle/_ref/ (a build product, shipped like libfdgs.so) and oracle/ref_modules.py imports them sourceless with
`diff_gaussian_rasterization` resolving to the shim package at the repository root.

  * CPU leg (`-m "not gpu"`): the reference's render() on CPU tensors, the shim's two device calls routed to the C rasterizer oracle
    (tests/cpu_raster_standin.py) -- pins tests/scenes.py::oracle_render_chain, THE checker of every full-size GPU parity test and of
    bench.py's parity block, to the reference's own render() source (activation order, time tensor, feature cat, result dict, gradient sink).
  * GPU leg (`-m gpu`): the same function on the MI355X over the HIP rasterizer -- every stage / pipe branch / camera type -- against
    `fdgs.render` on the same model: the drop-in claim, executed rather than read.
"""
import importlib
import math

import numpy as np
import pytest
import torch

from oracle import ref_modules
from scenes import oracle_render_chain, rel_l2

synthetic = importlib.import_module("4dgaussians_amd.synthetic")
needs_ref = pytest.mark.skipif(not ref_modules.available(), reason="oracle/_ref not built (python -m oracle.build_ref where /root/reference exists)")
GAUSS = ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")


class _Pipe:
    def __init__(self, sh=False, cov=False, debug=False):
        self.convert_SHs_python, self.compute_cov3D_python, self.debug = sh, cov, debug


def _with_reference_network(pc, ns, device="cpu"):
    """A model with the same parameters whose `_deformation` is the REFERENCE's deform_network (state_dict copied)."""
    cfgname = pc._cfgname
    net = ns.deform_network(synthetic.deform_args(cfgname))
    net.load_state_dict(pc._deformation.state_dict(), strict=True)
    twin = synthetic.SynthModel(pc._xyz.shape[0], cfgname, seed=pc._seed, deformation=net)
    with torch.no_grad():
        for k in GAUSS:
            getattr(twin, k).copy_(getattr(pc, k))
    return twin.to(device)


def _model(n, cfg="dynerf_default", seed=17, boost=1.0):
    pc = synthetic.SynthModel(n, cfg, seed=seed)
    pc._cfgname, pc._seed = cfg, seed
    with torch.no_grad():
        pc._scaling.add_(boost)
    return pc


def _grads(pc):
    return {k: p.grad.detach().cpu().numpy().copy() for k, p in pc.named_parameters() if p.grad is not None}


@needs_ref
@pytest.mark.parametrize("stage", ["coarse", "fine"])
def test_reference_render_source_on_cpu_equals_the_oracle_chain(stage, monkeypatch):
    import cpu_raster_standin
    ns = ref_modules.load()
    pc = _model(2500, "dynerf_default", seed=3)
    cam = synthetic.make_camera(200, 152, theta_deg=40.0, time=0.43)
    o, dc, dd, gref = oracle_render_chain(pc, cam, stage, target_seed=1)
    twin = _with_reference_network(pc, ns)
    with cpu_raster_standin.installed(monkeypatch):
        res = ns.render(cam, twin, _Pipe(), torch.zeros(3), stage=stage)
        assert set(res) == {"render", "viewspace_points", "visibility_filter", "radii", "depth"}
        (res["render"] * torch.tensor(dc)).sum().backward()
    img = res["render"].detach().numpy()
    assert (o.radii > 0).sum() > 500
    assert np.abs(img - o.color).max() < 2e-5 and np.abs(res["depth"].detach().numpy() - o.depth).max() < 2e-4
    assert (res["radii"].numpy() != o.radii).mean() < 1e-3 and bool((res["visibility_filter"].numpy() == (res["radii"].numpy() > 0)).all())
    g = _grads(twin)
    for k, v in gref.items():
        if k.startswith("__") or v is None or float(np.abs(v).max()) == 0:
            continue
        assert rel_l2(g[k], v) < 2e-4, (k, rel_l2(g[k], v))
    assert rel_l2(res["viewspace_points"].grad.numpy(), gref["__means2D"]) < 1e-5


# --------------------------------------------------------------------------------------------------------------------------- GPU leg
def _run(render_fn, cam, pc, pipe, bg, w, **kw):
    for p in pc.parameters():
        p.grad = None
    res = render_fn(cam, pc, pipe, bg, **kw)
    (res["render"] * w).sum().backward()
    torch.cuda.synchronize()
    return res, _grads(pc), res["viewspace_points"].grad.detach().cpu().numpy().copy()


def _compare(tag, a, b, img_tol, grad_tol, keys=None):
    (ra, ga, va), (rb, gb, vb) = a, b
    d = (ra["render"] - rb["render"]).abs()
    mism = float((ra["radii"] != rb["radii"]).float().mean())
    print(f"[{tag}] visible {int((ra['radii'] > 0).sum())}  image max diff {float(d.max()):.2e} mean {float(d.mean()):.2e}  radii mismatch {mism:.2e}")
    assert int((ra["radii"] > 0).sum()) > 500
    assert float(d.mean()) <= img_tol and mism <= 1e-3, (tag, float(d.mean()), mism)
    assert float((ra["depth"] - rb["depth"]).abs().mean()) <= 10 * img_tol
    assert torch.equal(ra["visibility_filter"], ra["radii"] > 0)
    errs = {}
    for k in (keys or ga):
        assert k in gb, (tag, k)
        if float(np.abs(ga[k]).max()) > 0:
            errs[k] = rel_l2(gb[k], ga[k])
    errs["viewspace"] = rel_l2(vb, va)
    print("   gradient rel-L2: " + ", ".join(f"{k.replace('_deformation.deformation_net.', '')}={v:.1e}" for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:8]))
    for k, e in errs.items():
        assert e <= grad_tol, (tag, k, e)


@needs_ref
@pytest.mark.gpu
def test_reference_render_coarse_over_the_shim_equals_fdgs_render():
    fd = importlib.import_module("4dgaussians_amd")
    ns = ref_modules.load()
    dev = torch.device("cuda:0")
    pc = _model(6000).to(dev)
    cam = synthetic.make_camera(240, 180, theta_deg=55.0, time=0.4).to(dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    w = torch.randn(3, 180, 240, generator=torch.Generator().manual_seed(2)).to(dev)
    a = _run(ns.render, cam, pc, _Pipe(), bg, w, stage="coarse")
    b = _run(fd.render, cam, pc, _Pipe(), bg, w, stage="coarse")
    _compare("coarse", a, b, 1e-7, 1e-5, keys=GAUSS)


@needs_ref
@pytest.mark.gpu
def test_the_reference_render_python_branches_are_dead_code_in_the_reference_itself():
    """Executing the reference's render() shows that its two python branches cannot run in the reference either:
    `pipe.convert_SHs_python` leaves `shs_final` set next to `colors_precomp` (gaussian_renderer/__init__.py:104-123: the `shs = None` is
    commented out), so the rasterizer's own argument check -- the upstream extension raises the same message -- rejects the call;
    `pipe.compute_cov3D_python` leaves `scales` None and then applies the scaling activation to it (:74-76, :97).
    The shim reproduces the first failure verbatim; fdgs.render implements what the branches INTEND (tests/test_gpu_render_branches.py)."""
    fd = importlib.import_module("4dgaussians_amd")
    ns = ref_modules.load()
    dev = torch.device("cuda:0")
    pc = _model(3000).to(dev)
    cam = synthetic.make_camera(160, 120, theta_deg=55.0, time=0.4).to(dev)
    bg = torch.zeros(3, device=dev)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        ns.render(cam, pc, _Pipe(sh=True), bg, stage="coarse")
    with pytest.raises(TypeError):
        ns.render(cam, pc, _Pipe(cov=True), bg, stage="coarse")
    for pipe in (_Pipe(sh=True), _Pipe(cov=True)):
        res = fd.render(cam, pc, pipe, bg, stage="coarse")
        assert torch.isfinite(res["render"]).all() and int((res["radii"] > 0).sum()) > 300


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["dynerf_default", "dnerf_bouncingballs", "hypernerf_default"])
def test_reference_render_fine_with_the_reference_deform_network_equals_fdgs_render(cfg):
    """Reference render() + reference deform_network (torch ops on the device) + HIP rasterizer  vs  fdgs.render (fused HIP path)."""
    fd = importlib.import_module("4dgaussians_amd")
    ns = ref_modules.load()
    dev = torch.device("cuda:0")
    pc = _model(6000, cfg)
    twin = _with_reference_network(pc, ns, dev)
    pc = pc.to(dev)
    cam = synthetic.make_camera(240, 180, theta_deg=-35.0, time=0.61).to(dev)
    bg = torch.zeros(3, device=dev)
    w = torch.randn(3, 180, 240, generator=torch.Generator().manual_seed(4)).to(dev)
    a = _run(ns.render, cam, twin, _Pipe(), bg, w, stage="fine")
    assert type(twin._deformation).__module__ == "scene.deformation"
    b = _run(fd.render, cam, pc, _Pipe(), bg, w, stage="fine")
    # the deformation differs in summation order (torch GEMMs vs MFMA tiles): images to 1e-6 in the mean, gradients to 1e-3
    # (typ. 1e-5; a ReLU within rounding of zero on one Gaussian is what the bound leaves room for at this size)
    _compare(f"fine/{cfg} reference network", a, b, 2e-6, 1e-3)
    # ... and fdgs.render with the foreign (reference) module: only the rasterizer replaced -- the same graph as the reference's render()
    c = _run(fd.render, cam, twin, _Pipe(), bg, w, stage="fine")
    _compare(f"fine/{cfg} fdgs.render over the reference network", a, c, 1e-7, 1e-5)


@needs_ref
@pytest.mark.gpu
def test_reference_render_fine_with_our_deform_network_as_the_module():
    """The reference's render() calling OUR deform_network through the reference's module call (six tensors incl. the [N,1] time tensor)."""
    fd = importlib.import_module("4dgaussians_amd")
    ns = ref_modules.load()
    dev = torch.device("cuda:0")
    pc = _model(6000).to(dev)
    cam = synthetic.make_camera(240, 180, theta_deg=10.0, time=0.25).to(dev)
    bg = torch.zeros(3, device=dev)
    w = torch.randn(3, 180, 240, generator=torch.Generator().manual_seed(5)).to(dev)
    a = _run(ns.render, cam, pc, _Pipe(), bg, w, stage="fine")
    b = _run(fd.render, cam, pc, _Pipe(), bg, w, stage="fine")
    _compare("fine, our module inside the reference's render()", a, b, 1e-7, 2e-5)


@needs_ref
@pytest.mark.gpu
def test_reference_panoptic_sports_settings_through_both_renders():
    """scene/dataset_readers.py:485-508 builds the settings itself ([1,4,4] matrices, sh_degree 0, debug=True) and render() takes them as they
    are (cam_type == "PanopticSports", gaussian_renderer/__init__.py:53-55)."""
    fd = importlib.import_module("4dgaussians_amd")
    ns = ref_modules.load()
    dev = torch.device("cuda:0")
    pc = _model(6000).to(dev)
    W, H = 256, 192
    sc = synthetic.make_camera(W, H, theta_deg=70.0, time=0.5)
    fx, fy = W / (2 * math.tan(sc.FoVx * 0.5)), H / (2 * math.tan(sc.FoVy * 0.5))
    k = [[fx, 0.0, W / 2.0], [0.0, fy, H / 2.0], [0.0, 0.0, 1.0]]
    w2c = sc.world_view_transform.t().contiguous().numpy()
    settings = ns.setup_camera(W, H, k, w2c, near=0.01, far=100)
    assert type(settings).__name__ == "GaussianRasterizationSettings" and tuple(settings.viewmatrix.shape) == (1, 4, 4) and settings.debug is True
    view = {"camera": settings, "time": 0.5}
    wimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(6)).to(dev)
    bg = torch.zeros(3, device=dev)
    a = _run(ns.render, view, pc, _Pipe(), bg, wimg, stage="fine", cam_type="PanopticSports")
    b = _run(fd.render, view, pc, _Pipe(), bg, wimg, stage="fine", cam_type="PanopticSports")
    _compare("PanopticSports", a, b, 1e-7, 2e-5)
