"""The reference's TRAIN STEP executed over the shims, against the same step through this repository's drop-ins (GPU; test infrastructure).

Two legs from identical state, both holding their Gaussians in the REFERENCE'S OWN `GaussianModel` object (byte-compiled from
/root/reference into oracle/_ref, imported sourceless -- oracle/ref_modules.load_train()):

  leg A "reference over the shims"   the reference's `render()` (gaussian_renderer/__init__.py:18) with the reference's `deform_network`,
        its `l1_loss` (utils/loss_utils.py:20), `GaussianModel.training_setup` (:165, torch.optim.Adam eps 1e-15), `update_learning_rate` (:197),
        `compute_regulation` (:576), `add_densification_stats` (:516), `densify` (:495), `prune` (:481), `reset_opacity` (:269) -- only
        `diff_gaussian_rasterization` and `simple_knn` resolve to this repository's shims.
  leg B "drop-ins"   the SAME model class as the holder of the state, with `fdgs.deform_network`, `fdgs.render`, `fdgs.losses.l1_loss`,
        `fdgs.compute_regulation`, `fdgs.densify.{add_densification_stats, densify, prune, reset_opacity}` and `fdgs.FusedAdam` in the places of
        the calls above -- what a user of the reference does when switching.

The loop body is train.py:180-292 for batch size 1 in the fine stage (per-view render, L1 + plane regulariser, backward, max_radii2D update
:259-261, statistics :262, densify :273, prune :277 [its N > 200 000 guard dropped so that the method runs], opacity reset :283, optimizer
step :291), with the schedule shortened: densification every `interval` iterations, opacity reset once.  `torch.normal` (the split children,
scene/gaussian_model.py:424) is fed the SAME pre-drawn samples in both legs.  After every densification event leg B continues from leg A's
state (`resync_every_event`): an optimisation trajectory amplifies float rounding (Adam divides by sqrt(v) + 1e-15), so only a segment that
STARTS from identical state can be compared to rounding; run with `resync_every_event=False` the two legs stayed on the same N through three
densifications + a prune and differed by ONE Gaussian at the fourth -- three candidates within 0.5 % of the threshold
(profiles/r05_reference_train_step_first_run.json).
"""
import importlib
import types

import numpy as np
import torch

import fit_proxy

NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def train_opt(iters):
    """arguments/__init__.py:109-149 (OptimizationParams defaults), schedule lengths scaled to a run of `iters` iterations."""
    return types.SimpleNamespace(
        position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01, position_lr_max_steps=iters,
        deformation_lr_init=0.00016, deformation_lr_final=0.000016, deformation_lr_delay_mult=0.01, grid_lr_init=0.0016, grid_lr_final=0.00016,
        feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001, percent_dense=0.01,
        opacity_threshold_fine_init=0.005, opacity_threshold_fine_after=0.005, densify_until_iter=iters + 1)


def _new_model(ns, student, dev, net):
    gm = ns.GaussianModel(3, student._deformation.args)
    net.load_state_dict(student._deformation.state_dict(), strict=True)
    gm._deformation = net.to(dev)
    for k in NAMES:
        setattr(gm, k, torch.nn.Parameter(getattr(student, k).detach().clone().to(dev).requires_grad_(True)))
    n = gm._xyz.shape[0]
    gm.active_sh_degree = 3
    gm.max_radii2D = torch.zeros(n, device=dev)
    gm._deformation_table = torch.ones(n, dtype=torch.bool, device=dev)
    gm.spatial_lr_scale = 1.0
    return gm


def _copy_state(fdgs, src, dst):
    """dst (leg B) <- src (leg A): parameters, network, Adam moments and step counts, the densification statistics."""
    dev = src._xyz.device
    dst._deformation.load_state_dict(src._deformation.state_dict(), strict=True)
    for k in NAMES:
        setattr(dst, k, torch.nn.Parameter(getattr(src, k).detach().clone().requires_grad_(True)))
    groups = [{"params": [dst._xyz], "name": "xyz"}, {"params": list(dst._deformation.get_mlp_parameters()), "name": "deformation"},
              {"params": list(dst._deformation.get_grid_parameters()), "name": "grid"}, {"params": [dst._features_dc], "name": "f_dc"},
              {"params": [dst._features_rest], "name": "f_rest"}, {"params": [dst._opacity], "name": "opacity"},
              {"params": [dst._scaling], "name": "scaling"}, {"params": [dst._rotation], "name": "rotation"}]
    for g, ga in zip(groups, src.optimizer.param_groups):
        assert g["name"] == ga["name"]
        g["lr"] = ga["lr"]
    dst.optimizer = fdgs.FusedAdam(groups, lr=0.0, eps=1e-15)
    for g, ga in zip(dst.optimizer.param_groups, src.optimizer.param_groups):
        for p, pa in zip(g["params"], ga["params"]):
            sa = src.optimizer.state.get(pa)
            if sa:
                dst.optimizer.state[p] = {"step": torch.tensor(float(sa["step"])), "exp_avg": sa["exp_avg"].detach().clone(),
                                          "exp_avg_sq": sa["exp_avg_sq"].detach().clone()}
    for k in ("xyz_gradient_accum", "denom", "max_radii2D", "_deformation_accum", "_deformation_table"):
        setattr(dst, k, getattr(src, k).detach().clone())
    assert dst._xyz.device == dev


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    d = float(b.norm())
    return float((a - b).norm()) / d if d > 0 else float(a.norm())


def run(iters=200, interval=50, n=3000, W=128, H=96, prune_at=100, reset_at=120, extent=None, device="cuda:0", grad_quantile=0.8, only_leg_a=False,
        resync_every_event=True):
    """Runs both legs in lock step.  Returns a report dict (see the test for what is asserted)."""
    from oracle import ref_modules
    fdgs = importlib.import_module("4dgaussians_amd")
    ns = ref_modules.load_train()
    dev = torch.device(device)
    student, cams, targets = fit_proxy.make_problem(n=n, W=W, H=H)
    hyper = student._deformation.args
    opt = train_opt(iters)
    if extent is None:      # scene extent such that percent_dense * extent sits at the median splat size: clones AND splits happen
        extent = float(torch.exp(student._scaling).max(1).values.median()) / opt.percent_dense
    A = _new_model(ns, student, dev, ns.deform_network(hyper))
    if only_leg_a:        # (CPU dry run of the reference side of this harness: tools only)
        return _run_leg_a_only(ns, A, opt, student, cams, targets, iters, interval, prune_at, reset_at, extent, grad_quantile, dev)
    B = _new_model(ns, student, dev, fdgs.deform_network(hyper))
    for m in (A, B):
        m.training_setup(opt)                                     # the reference's own method on both holders ...
    _copy_state(fdgs, A, B)                                       # ... then leg B's optimizer becomes FusedAdam over the same groups / state
    tg = [torch.tensor(t, device=dev) for t in targets]
    cg = [c.to(dev) for c in cams]
    pipe, bg = fit_proxy.synthetic.PipelineParams(), torch.zeros(3, device=dev)
    tsw, l1w, tvw = 0.01, 0.0001, 0.0001                          # arguments/__init__.py:85-87 (ModelHiddenParams defaults)
    bank = torch.randn(400_000, 3, generator=torch.Generator().manual_seed(99)).to(dev)
    real_normal = torch.normal

    def fed_normal(mean=None, std=None, **k):                     # torch.normal(mean=0, std=stds) = stds * standard-normal samples
        return mean + bank[:std.shape[0]] * std

    rep = {"events": [], "psnr_A": [], "psnr_B": [], "resyncs": 0, "threshold": None}
    thr = None
    for it in range(1, iters + 1):
        v = (it - 1) % len(cg)
        for m in (A, B):
            m.update_learning_rate(it)                            # train.py:172
        # ---- leg A: the reference's calls
        pa = ns.render(cg[v], A, pipe, bg, stage="fine", cam_type=None)
        la = ns.l1_loss(pa["render"], tg[v]) + A.compute_regulation(tsw, l1w, tvw)
        la.backward()
        # ---- leg B: the drop-ins
        pb = fdgs.render(cg[v], B, pipe, bg, stage="fine", cam_type=None)
        lb = fdgs.losses.l1_loss(pb["render"], tg[v]) + fdgs.compute_regulation(B, tsw, l1w, tvw)
        lb.backward()
        with torch.no_grad():
            rep["psnr_A"].append(float(ns.psnr(pa["render"][None], tg[v][None]).mean()))
            rep["psnr_B"].append(float(ns.psnr(pb["render"][None], tg[v][None]).mean()))
            # train.py:259-262
            va, ra = pa["visibility_filter"], pa["radii"]
            A.max_radii2D[va] = torch.max(A.max_radii2D[va], ra[va])
            A.add_densification_stats(pa["viewspace_points"].grad, va)
            fdgs.densify.add_densification_stats(B, pb["viewspace_points"].grad, pb["visibility_filter"], pb["radii"])
            if it % interval == 0:
                ev = {"iteration": it, "N_before": (int(A._xyz.shape[0]), int(B._xyz.shape[0]))}
                same_n = A._xyz.shape[0] == B._xyz.shape[0]
                if same_n:
                    ev["accum_rel_l2"] = _rel(B.xyz_gradient_accum, A.xyz_gradient_accum)
                    ev["denom_mismatch_frac"] = float((B.denom != A.denom).float().mean())
                    ev["max_radii2D_mismatch_frac"] = float((B.max_radii2D != A.max_radii2D).float().mean())
                    ev["xyz_rel_l2"] = _rel(B._xyz, A._xyz)
                ga = A.xyz_gradient_accum / A.denom
                ga[ga.isnan()] = 0.0
                if thr is None:     # one threshold for the run, placed where this scene densifies ~20 % of its Gaussians per event
                    thr = rep["threshold"] = float(torch.quantile(ga[ga > 0].flatten(), grad_quantile))
                gb = B.xyz_gradient_accum / B.denom
                gb[gb.isnan()] = 0.0
                opacity_thr = opt.opacity_threshold_fine_init
                torch.normal = fed_normal
                try:
                    A.densify(thr, opacity_thr, extent, None, 5, 5, None, it, "fine")            # train.py:273
                finally:
                    torch.normal = real_normal
                kept, clones, splits = fdgs.densify.densify(B, thr, opacity_thr, extent, None, 5, 5, None, it, "fine", normals=bank, reorder=False)
                ev["plan_B"] = (kept, clones, splits)
                if it == prune_at:                                                                  # train.py:277 (guard dropped)
                    A.prune(thr, opacity_thr, extent, 20)
                    fdgs.densify.prune(B, thr, opacity_thr, extent, 20, reorder=False)
                ev["N_after"] = (int(A._xyz.shape[0]), int(B._xyz.shape[0]))
                if same_n and ev["N_after"][0] != ev["N_after"][1]:
                    # which Gaussians were classified differently, and how close to the threshold were they in leg A?
                    sel_a, sel_b = (ga.flatten() >= thr), (gb.flatten() >= thr)
                    diff = torch.nonzero(sel_a != sel_b).flatten()
                    ev["differently_selected"] = [(int(i), float(ga.flatten()[i] / thr - 1.0)) for i in diff[:16]]
                    ev["n_differently_selected"] = int(diff.numel())
                if ev["N_after"][0] == ev["N_after"][1]:
                    ev["xyz_rel_l2_after"] = _rel(B._xyz, A._xyz)
                    ev["table_equal"] = bool(torch.equal(A._deformation_table, B._deformation_table))
                rep["events"].append(ev)
            if it == reset_at:                                                                      # train.py:283
                A.reset_opacity()
                fdgs.densify.reset_opacity(B)
            A.optimizer.step()                                                                      # train.py:291-292
            A.optimizer.zero_grad(set_to_none=True)
            B.optimizer.step()
            B.optimizer.zero_grad(set_to_none=True)
            if it % interval == 0 and (resync_every_event or A._xyz.shape[0] != B._xyz.shape[0]):
                # every segment starts from identical state (so that every event's statistics are comparable to rounding); without
                # `resync_every_event` only when the legs disagree on N (a Gaussian on the threshold: named above)
                rep["free_running_N_equal"] = rep.get("free_running_N_equal", True) and A._xyz.shape[0] == B._xyz.shape[0]
                _copy_state(fdgs, A, B)
                rep["resyncs"] += 1
    with torch.no_grad():
        rep["final_psnr_A"] = [float(ns.psnr(ns.render(c, A, pipe, bg, stage="fine")["render"][None], t[None]).mean()) for c, t in zip(cg, tg)]
        rep["final_psnr_B"] = [float(ns.psnr(fdgs.render(c, B, pipe, bg, stage="fine")["render"][None], t[None]).mean()) for c, t in zip(cg, tg)]
    rep["N_final"] = (int(A._xyz.shape[0]), int(B._xyz.shape[0]))
    rep["optimizer_B"] = type(B.optimizer).__name__
    rep["drift"] = float(np.abs(np.array(rep["psnr_A"]) - np.array(rep["psnr_B"])).max())
    return rep


def _run_leg_a_only(ns, A, opt, student, cams, targets, iters, interval, prune_at, reset_at, extent, grad_quantile, dev):
    A.training_setup(opt)
    tg = [torch.tensor(t, device=dev) for t in targets]
    cg = [c.to(dev) for c in cams]
    pipe, bg = fit_proxy.synthetic.PipelineParams(), torch.zeros(3, device=dev)
    bank = torch.randn(400_000, 3, generator=torch.Generator().manual_seed(99)).to(dev)
    real_normal = torch.normal
    thr, ns_ = None, []
    for it in range(1, iters + 1):
        v = (it - 1) % len(cg)
        A.update_learning_rate(it)
        pa = ns.render(cg[v], A, pipe, bg, stage="fine", cam_type=None)
        (ns.l1_loss(pa["render"], tg[v]) + A.compute_regulation(0.01, 0.0001, 0.0001)).backward()
        with torch.no_grad():
            va, ra = pa["visibility_filter"], pa["radii"]
            A.max_radii2D[va] = torch.max(A.max_radii2D[va], ra[va].float())
            A.add_densification_stats(pa["viewspace_points"].grad, va)
            if it % interval == 0:
                ga = A.xyz_gradient_accum / A.denom
                ga[ga.isnan()] = 0.0
                if thr is None:
                    thr = float(torch.quantile(ga[ga > 0].flatten(), grad_quantile))
                torch.normal = lambda mean=None, std=None, **k: mean + bank[:std.shape[0]] * std
                try:
                    A.densify(thr, 0.005, extent, None, 5, 5, None, it, "fine")
                finally:
                    torch.normal = real_normal
                if it == prune_at:
                    A.prune(thr, 0.005, extent, 20)
                ns_.append(int(A._xyz.shape[0]))
            if it == reset_at:
                A.reset_opacity()
            A.optimizer.step()
            A.optimizer.zero_grad(set_to_none=True)
    return {"N": ns_, "threshold": thr}
