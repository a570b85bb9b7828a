"""Parity at the sizes the headline is quoted on (BASELINE.json configs 2, 3, 4, 5): render() forward + backward through the HIP
path against the oracle chain on the SAME frame -- image, depth, radii, every Gaussian-parameter gradient, every HexPlane and
MLP gradient, the view-space gradient.  The small-size tests exercise one tile round per CU; these run the persistent tile
loops (9 rounds per CU), the cost-model work split of the weight-gradient kernel, the LDS-privatised time planes at full
occupancy, the multi-chunk scans and the multi-million-pair sort -- with the numbers looked at.

Tolerances (north_star): image PSNR vs oracle >= 80 dB (our reading of "within 1e-4 PSNR": mean squared error <= 1e-8), gradients <= 1e-3
rel-L2 per parameter group against ONE reference -- the float64 evaluation of the pinned oracle -- with the rows on which the float32
implementation took a near-zero ReLU / texel-cell decision the other way proven, named and attributed (oracle/parity.py)."""
import importlib
import math

import numpy as np
import pytest
import torch

from oracle import parity as P
from scenes import oracle_render_chain, rel_l2

pytestmark = pytest.mark.gpu
synthetic = importlib.import_module("4dgaussians_amd.synthetic")

CONFIGS = {
    "cfg2_dnerf_100k_800x800": (100_000, 800, 800, "dnerf_bouncingballs"),
    "cfg3_hypernerf_300k_536x960": (300_000, 536, 960, "hypernerf_default"),
    "cfg4_dynerf_300k_1352x1014": (300_000, 1352, 1014, "dynerf_default"),
    # BASELINE.json configs[4]: the 2 M / 2048^2 stress case -- multi-chunk scans, a 23 M-pair sort, 16 384 tiles, 62 500 backward tiles,
    # > 4 GB of saved activations; the oracle chain takes ~1 minute of host time
    "cfg5_stress_2M_2048x2048": (2_000_000, 2048, 2048, "dynerf_default"),
}


def _groups(grads):
    g = {"xyz": ["_xyz"], "scaling": ["_scaling"], "rotation": ["_rotation"], "opacity": ["_opacity"], "f_dc": ["_features_dc"],
         "f_rest": ["_features_rest"],
         "planes": [k for k in grads if "grids" in k and grads[k] is not None],
         "mlp": [k for k in grads if k.startswith("_deformation.") and "grids" not in k and grads[k] is not None]}
    return g


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("with_depth", [False])
def test_render_fwd_bwd_parity_at_baseline_size(name, with_depth, order="hilbert", scene="cube"):
    fd = importlib.import_module("4dgaussians_amd")
    dev = torch.device("cuda:0")
    N, W, H, dcfg = CONFIGS[name]
    pc = synthetic.SynthModel(N, dcfg, seed=6666, scene=scene)     # the bench scene ...
    if order != "random":
        fd.densify.spatial_reorder(pc, curve=order)                # ... in the order bench.py runs it (CPU tensors: torch ops)
    cam = synthetic.orbit_cameras(W, H, n=160)[8]
    # Parity is FACTORED (oracle/chain.py: `deformed`): (A) the HIP deformation's outputs against the deformation oracle; (B) render() -- the
    # same deformation kernel, then the HIP rasterizer -- against the C rasterizer oracle blending THOSE deformed Gaussians, gradients
    # chained through the oracle's deformation (float64).  A rasterizer fed with inputs that differ by 1e-6 may order two near-equal-depth
    # Gaussians the other way round; that is no property of either stage.
    import copy
    pcg = copy.deepcopy(pc).to(dev)
    with torch.no_grad():
        hip_def = fd.deformation.deform(pcg._deformation, pcg._xyz, pcg._scaling, pcg._rotation, pcg._opacity, shs_dc=pcg._features_dc,
                                        shs_rest=pcg._features_rest, time=cam.time, activate=True)
    hip_def = [t.cpu() for t in hip_def]
    # gradient reference: the oracle's deformation backward evaluated in float64 on the live rows (scenes.oracle_render_chain)
    o, dc, dd, gref = oracle_render_chain(pc, cam, "fine", target_seed=3, with_depth_grad=with_depth, grad_dtype=torch.float64, deformed=hip_def)
    ctx = gref.pop("__ctx")
    dmax = {k: float((a.reshape(N, -1) - b.reshape(N, -1)).abs().max()) for k, a, b in zip(("xyz", "scales", "rotations", "opacity", "shs"), hip_def, gref.pop("__deformed"))}
    print(f"[{name} scene={scene} order={order}] (A) HIP deformation vs deformation oracle, max abs: " + ", ".join(f"{k} {v:.1e}" for k, v in dmax.items()))
    assert max(dmax.values()) < 2e-5, dmax        # (measured: positions 8e-7, rotations 2e-6, SH 9e-7, scales 2e-8, opacity 1e-7)
    pc = pcg
    res = fd.render(cam.to(dev), pc, synthetic.PipelineParams(), torch.zeros(3, device=dev), stage="fine")
    img = res["render"]
    loss = (img * torch.tensor(dc, device=dev)).sum()
    if with_depth:
        loss = loss + (res["depth"] * torch.tensor(dd, device=dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    im = img.detach().cpu().numpy()
    d = np.abs(im - o.color)
    psnr = 10 * math.log10(1.0 / max(float(((im.astype(np.float64) - o.color) ** 2).mean()), 1e-20))
    dmean = float(np.abs(res["depth"].detach().cpu().numpy() - o.depth).mean())
    radii = res["radii"].cpu().numpy()
    mism = float((radii != o.radii).mean())
    # isolated pixels where a 1/255 / T < 1e-4 decision falls the other way in float rounding: counted and bounded, not hidden in the mean
    n_flip = int((d.max(axis=0) > 1e-4).sum())
    print(f"[{name} scene={scene} order={order} depth_grad={with_depth}] visible {(o.radii > 0).sum()}  image psnr {psnr:.1f} dB  max|dC| {d.max():.2e}  mean {d.mean():.2e}  pixels over 1e-4: {n_flip} of {H * W}  "
          f"depth mean abs {dmean:.2e}  radii mismatch {mism:.2e}")
    assert psnr >= 80.0 and d.mean() < 2e-6
    # measured: 18 .. 958 such pixels (0.003 % .. 0.07 %); each is ONE alpha >= 1/255 decision taken the other way for an entry
    # whose alpha sits within rounding of the threshold, which moves a pixel channel by at most alpha * T * colour <= colour / 255 -- the SH colour
    # max(0, sh + 0.5) has no upper clamp (utils/sh_utils.py:113), so the bound is the largest colour of a visible Gaussian, not 1
    cmax = max(1.0, float(o.field("rgb")[o.radii > 0].max()))
    assert n_flip <= int(2e-3 * H * W) and d.max() <= 1.1 * cmax / 255.0, (n_flip, float(d.max()), cmax)
    assert dmean < 2e-5 and mism < 2e-4
    named = dict(pc.named_parameters())
    means2D_ref = gref.pop("__means2D")
    impl = {k: (named[k].grad.cpu().numpy() if named[k].grad is not None else np.zeros_like(v)) for k, v in gref.items() if v is not None}
    rep = P.attribute(*ctx, impl, gref)
    vs = rel_l2(res["viewspace_points"].grad.cpu().numpy(), means2D_ref)
    per_tensor = {k: rel_l2(impl[k], v) for k, v in gref.items() if v is not None and float(np.abs(v).max()) > 0}
    worst = sorted(per_tensor.items(), key=lambda kv: -kv[1])[:4]
    print("   group rel-L2 vs float64 oracle (raw): " + ", ".join(f"{k}={v:.2e}" for k, v in rep["grad_rel_l2_vs_float64_raw"].items()) + f", viewspace={vs:.2e}")
    print("   with kink rows attributed:            " + ", ".join(f"{k}={v:.2e}" for k, v in rep["grad_rel_l2_vs_float64_kink_rows_attributed"].items()))
    print(f"   kink rows ({rep['n_kink_rows']} of {N}, allowed {rep['max_kink_rows']}; each with the margin that admitted it): {rep['kink_rows']}  "
          f"heavy rows: {rep['heavy_rows_within_tol_rowwise']}  unexplained rows (counted in the figures above): {rep['unexplained_rows']}")
    print(f"   windows: {rep['windows']}")
    print("   worst tensors (raw): " + ", ".join(f"{k}={v:.2e}" for k, v in worst))
    assert rep["ok"], rep["failures"]
    assert vs <= 1e-3, vs
    # ---- the apples-to-apples figure for north_star's 1e-3 (round 6): the same HIP gradients against the REFERENCE'S OWN float32 modules --
    # its render() + deform_network as torch ops on this device, over the same HIP rasterizer -- RAW, no attribution (oracle/ref_f32.py).
    # Two float32 evaluations of the same function take near-zero ReLU / texel-cell decisions apart from each other just as a float32 and a
    # float64 one do, so a scene with such rows (shell: five of 300 000) shows them here too; the bound asserted is what the reference's own
    # float32 modules are measured to differ from the float64 oracle by on the same scenes (tools/parity_windows.py), 5e-3, and the figure is
    # printed for the record (bench.py carries it in its parity block: `vs_reference_f32_modules_raw`).
    from oracle import ref_f32
    if ref_f32.available() and not with_depth:
        dcol = torch.tensor(dc, device=dev)
        img32, radii32, g32, vs32 = ref_f32.reference_f32_frame(pc, pc._deformation.args, cam.to(dev), synthetic.PipelineParams(), torch.zeros(3, device=dev), dcol)
        raw = ref_f32.compare_raw(pc, g32, img32, im)
        vs_raw = rel_l2(res["viewspace_points"].grad.cpu().numpy(), vs32)
        print(f"   RAW vs the reference's own float32 modules on this device: " + ", ".join(f"{k}={v:.2e}" for k, v in raw["groups"].items())
              + f", viewspace={vs_raw:.2e}; image mean abs {raw['image_mean_abs']:.2e}; worst tensor {raw['worst_single_tensor']}")
        assert raw["image_mean_abs"] < 2e-6 and float((radii != radii32).mean()) < 2e-4
        assert max(raw["groups"].values()) <= 5e-3 and vs_raw <= 1e-3, raw
        assert sum(v <= 1e-3 for v in raw["groups"].values()) >= len(raw["groups"]) - 2, raw      # (north_star's bound itself: on all but the kink-carrying groups)
    # parameters of disabled heads / the unused time net get no gradient on either side
    for k, v in gref.items():
        if not k.startswith("__") and (v is None or float(np.abs(v).max()) == 0.0):
            a = named[k].grad
            assert a is None or float(a.abs().max()) < 1e-12, k


def test_render_parity_with_depth_gradient_at_config4():
    """Same as above with a gradient on the depth output too (the DEPTH instantiation of the blending backward), on the
    generator's (random) order of the Gaussians: the plane-gradient kernel's fallback path at full size."""
    test_render_fwd_bwd_parity_at_baseline_size("cfg4_dynerf_300k_1352x1014", True, order="random")


@pytest.mark.parametrize("order", ["hilbert", "random"])
def test_render_parity_on_the_shell_scene_at_config4(order):
    """The regime a trained model is in (bench.py --scene shell: surface-like, translucent -- ~98 % of the visible Gaussians receive a
    gradient, ~87 % of the backward tiles are live, long un-terminated per-pixel lists), at the size the headline is quoted on, in the
    train loop's (Hilbert) order and in the generator's (random) order: same checks, same tolerances as the cube scene above."""
    test_render_fwd_bwd_parity_at_baseline_size("cfg4_dynerf_300k_1352x1014", False, order=order, scene="shell")


def test_render_parity_with_depth_gradient_at_config3():
    """The DEPTH instantiation of the blending backward on the HyperNeRF deformation config (three levels, net_width 128, three heads)."""
    test_render_fwd_bwd_parity_at_baseline_size("cfg3_hypernerf_300k_536x960", True)


def test_spatial_reorder_leaves_the_frame_unchanged_and_permutes_the_gradients():
    """Hilbert-ordering the model (what the train loop does after every densification) is semantically free: same image to
    1e-6, same radii and per-Gaussian gradients up to the permutation, same plane / MLP gradients."""
    fd = importlib.import_module("4dgaussians_amd")
    dev = torch.device("cuda:0")
    N, W, H = 60_000, 640, 480
    cam = synthetic.orbit_cameras(W, H, n=160)[21].to(dev)
    wimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(8)).to(dev)
    outs = []
    for reorder in (False, True):
        pc = synthetic.SynthModel(N, "dynerf_default", seed=77).to(dev)
        with torch.no_grad():
            pc._scaling.add_(0.5)
        perm = fd.densify.spatial_reorder(pc) if reorder else torch.arange(N, device=dev)
        res = fd.render(cam, pc, synthetic.PipelineParams(), torch.zeros(3, device=dev), stage="fine")
        (res["render"] * wimg).sum().backward()
        outs.append((res, {k: v.grad for k, v in pc.named_parameters() if v.grad is not None}, perm))
    (ra, ga, _), (rb, gb, perm) = outs
    d = (ra["render"] - rb["render"]).abs()
    print(f"image max diff {float(d.max()):.2e}, mean {float(d.mean()):.2e}")
    assert float(d.mean()) < 1e-7 and float(torch.quantile(d.flatten()[::7], 0.9999)) < 1e-5     # (equal-depth ties may reorder)
    assert torch.equal(ra["radii"][perm], rb["radii"])
    for k in ga:
        a, b = ga[k], gb[k]
        if k in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest"):
            a = a[perm]
        assert rel_l2(b.cpu().numpy(), a.cpu().numpy()) < 2e-4, k


@pytest.mark.parametrize("views", [0, 2])
def test_implicit_permutation_of_an_unordered_model_changes_nothing_the_caller_sees(views, monkeypatch):
    """An UNMODIFIED train loop (the reference's own densify / prune) keeps its Gaussians in no order at all.  render() then reads the model
    through a cached Hilbert permutation of its positions (renderer.IMPLICIT_ORDER, the default) -- and everything the caller sees is in
    the MODEL'S order: same image, same radii / visibility, same per-Gaussian gradients and viewspace gradient row for row, same plane /
    MLP gradients as with the permutation switched off; also through render_views and under torch.no_grad()."""
    fd = importlib.import_module("4dgaussians_amd")
    dev = torch.device("cuda:0")
    N, W, H = 60_000, 640, 480
    cams = [c.to(dev) for c in synthetic.orbit_cameras(W, H, n=160)[21:23]]
    wimg = torch.randn(3, H, W, generator=torch.Generator().manual_seed(8)).to(dev)
    outs = {}
    for implicit in (False, True):
        monkeypatch.setattr(fd.renderer, "IMPLICIT_ORDER", implicit)
        pc = synthetic.SynthModel(N, "dynerf_default", seed=77).to(dev)
        with torch.no_grad():
            pc._scaling.add_(0.5)
        assert fd.deformation.spatial_order_hint(pc._xyz) is False           # the generator's order
        pipe, bg = synthetic.PipelineParams(), torch.zeros(3, device=dev)
        if views:
            res = fd.render_views(cams[:views], pc, pipe, bg, stage="fine")
            sum((r["render"] * wimg).sum() for r in res).backward()
        else:
            res = [fd.render(cams[0], pc, pipe, bg, stage="fine")]
            (res[0]["render"] * wimg).sum().backward()
        with torch.no_grad():
            plain = fd.render(cams[0], pc, pipe, bg, stage="fine")
        outs[implicit] = (res, {k: v.grad for k, v in pc.named_parameters() if v.grad is not None}, plain)
        # (a LIVE entry of this very object: id() values are reused, so a dead model's entry may sit under the same key -- gpu_full_3.log)
        e = fd.deformation._perm_cache.get(id(pc._xyz))
        assert (e is not None and e[0]() is pc._xyz) == implicit
    (ra, ga, pa), (rb, gb, pb) = outs[False], outs[True]
    for a, b in zip(ra + [pa], rb + [pb]):
        d = (a["render"] - b["render"]).abs()
        assert float(d.mean()) < 1e-7 and float(torch.quantile(d.flatten()[::7], 0.9999)) < 1e-5     # (equal-depth ties may reorder)
        assert torch.equal(a["radii"], b["radii"]) and torch.equal(a["visibility_filter"], b["visibility_filter"])
        assert a["radii"].dtype == b["radii"].dtype and a["visibility_filter"].dtype == torch.bool
    for a, b in zip(ra, rb):
        ga2, gb2 = a["viewspace_points"].grad, b["viewspace_points"].grad
        assert rel_l2(gb2.cpu().numpy(), ga2.cpu().numpy()) < 2e-4
    assert set(ga) == set(gb)
    for k in ga:
        assert ga[k].shape == gb[k].shape and rel_l2(gb[k].cpu().numpy(), ga[k].cpu().numpy()) < 2e-4, k
