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
        (`leg_b="reference"`: a SECOND reference leg instead -- the noise floor of this comparison: both legs run the reference's code, and
        differ only by the summation order of the blending backward's float atomics, amplified by Adam.)

The loop body is train.py:180-292 for batch size 1 in the fine stage (per-view render, L1 + plane regulariser, backward, max_radii2D update
:259-261, statistics :262, densify :273, prune :277 [its N > 200 000 guard dropped so that the method runs], opacity reset :283, optimizer
step :291), with the schedule shortened: densification every `interval` iterations, one prune at `prune_at` (an iteration of its own: right
after a densification `max_radii2D` is all zeros), opacity reset once.  `torch.normal` (the split children, scene/gaussian_model.py:424) is
fed the SAME pre-drawn samples in both legs.  After every densification event leg B continues from leg A's state (`resync_every_event`):
an optimisation trajectory amplifies float rounding (Adam divides by sqrt(v) + 1e-15), so only a segment that STARTS from identical state
can be compared to rounding; `resync_every_event=False` lets the legs run free (resynchronised only when their N differs).

What an event records (round 6; `_compare_events`): BOTH legs' plans -- the reference leg's masks are RECORDED from the reference's own calls
(the split mask is the one densify_and_split hands to prune_points, the clone mask is confirmed by the rows densify_and_clone appends, the
prune mask is the one prune hands to prune_points), a drop-in leg's are the same expressions on its state, checked against the counts
fdgs_densify_plan returned --, whether the rows came out in the same ORDER, and when they did not, the comparison of the two results AS SETS
(rows matched by parent index + slot) plus every differently classified Gaussian with its margin to both thresholds
(`grad / threshold - 1`, `max scale / (percent_dense * extent) - 1`) in both legs.  A clone in one leg that is a split in the other adds
one row in both (N stays equal) and shifts every later kept row by one: such an event is a threshold tie when the margins say so, and a
defect otherwise -- the harness tells which.
"""
import importlib
import types

import numpy as np
import torch

import fit_proxy

NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
STATS = ("xyz_gradient_accum", "denom", "max_radii2D", "_deformation_accum", "_deformation_table")
TSW, L1W, TVW = 0.01, 0.0001, 0.0001                      # arguments/__init__.py:85-87 (ModelHiddenParams defaults)


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


def _copy_state(make_optimizer, src, dst):
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
    dst.optimizer = make_optimizer(groups)
    for g, ga in zip(dst.optimizer.param_groups, src.optimizer.param_groups):
        for p, pa in zip(g["params"], ga["params"]):
            sa = src.optimizer.state.get(pa)
            if sa:
                dst.optimizer.state[p] = {"step": torch.tensor(float(sa["step"])), "exp_avg": sa["exp_avg"].detach().clone(),
                                          "exp_avg_sq": sa["exp_avg_sq"].detach().clone()}
    for k in STATS:
        setattr(dst, k, getattr(src, k).detach().clone())
    assert dst._xyz.device == dev


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    d = float(b.norm())
    return float((a - b).norm()) / d if d > 0 else float(a.norm())


def _margins(m, thr, extent):
    """What GaussianModel.densify decides on (scene/gaussian_model.py:399-406,415-419,495-497), evaluated on the holder's state with the
    reference's own expressions: per Gaussian the averaged gradient, the largest activated scale, and the two masks."""
    g = m.xyz_gradient_accum / m.denom
    g[g.isnan()] = 0.0
    g = g.flatten()
    mx = torch.max(m.get_scaling, dim=1).values
    dense = m.percent_dense * extent
    clone = torch.logical_and(torch.where(torch.norm(g[:, None], dim=-1) >= thr, True, False), mx <= dense)
    split = torch.logical_and(torch.where(g >= thr, True, False), mx > dense)
    # the two DECISIONS themselves (the reported margins are float64 quotients of the same operands: a margin one ulp from zero may carry
    # the other sign than the float32 comparison the reference makes -- what "flipped" means is read from these, never from a margin's sign)
    return g, mx, dense, clone, split, (g >= thr), (mx > dense)


def _prune_terms(m, min_opacity, extent, max_screen_size):
    """What GaussianModel.prune decides on (scene/gaussian_model.py:481-490) with the reference's own expressions."""
    op = m.get_opacity.squeeze(-1)
    mx = m.get_scaling.max(dim=1).values
    mask = (op < min_opacity)
    if max_screen_size:
        mask = torch.logical_or(torch.logical_or(mask, m.max_radii2D > max_screen_size), mx > 0.1 * extent)
    return op, mx, m.max_radii2D.clone(), mask


def _row_keys(clone, split):
    """Identity of every row the clone-then-split-then-prune sequence leaves behind, in its order: 4 * parent + slot (0 = the original,
    1 = its clone, 2 / 3 = first / second child): [originals not split | clones | first children | second children]."""
    idx = torch.arange(clone.shape[0], device=clone.device)
    return torch.cat([idx[~split] * 4, idx[clone] * 4 + 1, idx[split] * 4 + 2, idx[split] * 4 + 3])


class ReferenceLeg:
    """The reference's own calls on the reference's own GaussianModel (only the rasterizer / knn packages are this repository's shims)."""
    name = "reference"

    def __init__(self, ns, model):
        self.ns, self.m = ns, model

    def make_optimizer(self, groups):
        return torch.optim.Adam(groups, lr=0.0, eps=1e-15)                     # scene/gaussian_model.py:184

    def frame(self, cam, target, pipe, bg):
        m, ns = self.m, self.ns
        p = ns.render(cam, m, pipe, bg, stage="fine", cam_type=None)
        (ns.l1_loss(p["render"], target) + m.compute_regulation(TSW, L1W, TVW)).backward()
        with torch.no_grad():
            psnr = float(ns.psnr(p["render"][None], target[None]).mean())
            v, r = p["visibility_filter"], p["radii"]
            m.max_radii2D[v] = torch.max(m.max_radii2D[v], r[v])                 # train.py:259-262
            m.add_densification_stats(p["viewspace_points"].grad, v)
        return psnr

    def densify(self, thr, opacity_thr, extent, it, fed_normal):
        """m.densify (train.py:273) with the reference's own masks RECORDED: the split mask is what densify_and_split hands to prune_points
        (:413-414), the clone mask is confirmed by the rows densify_and_clone hands to densification_postfix (:421-429)."""
        A = self.m
        n0 = int(A._xyz.shape[0])
        g, mx, dense, clone, split, sel, big = _margins(A, thr, extent)
        xyz0 = A._xyz.detach().clone()
        seen = {"postfix": [], "split_mask": torch.zeros(n0, dtype=torch.bool, device=xyz0.device)}
        postfix, prune_points = A.densification_postfix, A.prune_points

        def rec_postfix(new_xyz, *a, **k):
            seen["postfix"].append(new_xyz.detach().clone())
            return postfix(new_xyz, *a, **k)

        def rec_prune_points(mask):
            seen["split_mask"] = mask[:n0].clone()
            return prune_points(mask)

        A.densification_postfix, A.prune_points = rec_postfix, rec_prune_points
        real_normal, torch.normal = torch.normal, fed_normal
        try:
            A.densify(thr, opacity_thr, extent, None, 5, 5, None, it, "fine")
        finally:
            torch.normal = real_normal
            del A.densification_postfix, A.prune_points
        assert torch.equal(seen["postfix"][0], xyz0[clone]), "the recorded clone rows are not the rows of the restated clone mask"
        assert torch.equal(seen["split_mask"], split), "the recorded split mask is not the restated split mask"
        assert len(seen["postfix"]) == (2 if bool(split.any()) else 1)
        return g, mx, dense, clone, split, sel, big

    def prune(self, thr, opacity_thr, extent, size):
        A = self.m
        terms = _prune_terms(A, opacity_thr, extent, size)
        seen = {}
        prune_points = A.prune_points

        def rec_prune_points(mask):
            seen["mask"] = mask.clone()
            return prune_points(mask)

        A.prune_points = rec_prune_points
        try:
            A.prune(thr, opacity_thr, extent, size)                               # train.py:277 (guard dropped)
        finally:
            del A.prune_points
        assert torch.equal(seen["mask"], terms[3]), "the recorded prune mask is not the restated prune mask"
        return terms

    def reset_opacity(self):
        self.m.reset_opacity()

    def final_psnr(self, cams, targets, pipe, bg):
        return [float(self.ns.psnr(self.ns.render(c, self.m, pipe, bg, stage="fine")["render"][None], t[None]).mean()) for c, t in zip(cams, targets)]


class DropInLeg(ReferenceLeg):
    """The same holder class, every call replaced by this repository's drop-in."""
    name = "drop-ins"

    def __init__(self, ns, model, fdgs, bank):
        super().__init__(ns, model)
        self.fdgs, self.bank = fdgs, bank

    def make_optimizer(self, groups):
        return self.fdgs.FusedAdam(groups, lr=0.0, eps=1e-15)

    def frame(self, cam, target, pipe, bg):
        m, fdgs = self.m, self.fdgs
        p = fdgs.render(cam, m, pipe, bg, stage="fine", cam_type=None)
        (fdgs.losses.l1_loss(p["render"], target) + fdgs.compute_regulation(m, TSW, L1W, TVW)).backward()
        with torch.no_grad():
            psnr = float(self.ns.psnr(p["render"][None], target[None]).mean())
            fdgs.densify.add_densification_stats(m, p["viewspace_points"].grad, p["visibility_filter"], p["radii"])
        return psnr

    def densify(self, thr, opacity_thr, extent, it, fed_normal):
        d = _margins(self.m, thr, extent)                     # the same expressions on this leg's state, checked against the plan's counts
        plan = self.fdgs.densify.densify(self.m, thr, opacity_thr, extent, None, 5, 5, None, it, "fine", normals=self.bank, reorder=False)
        assert plan == (int((~d[4]).sum()), int(d[3].sum()), int(d[4].sum())), (plan, "the drop-in's plan is not its restated masks")
        return d

    def prune(self, thr, opacity_thr, extent, size):
        terms = _prune_terms(self.m, opacity_thr, extent, size)
        kept, _, _ = self.fdgs.densify.prune(self.m, thr, opacity_thr, extent, size, reorder=False)
        assert kept == int((~terms[3]).sum()), (kept, "the drop-in's prune plan is not its restated mask")
        return terms

    def reset_opacity(self):
        self.fdgs.densify.reset_opacity(self.m)

    def final_psnr(self, cams, targets, pipe, bg):
        return [float(self.ns.psnr(self.fdgs.render(c, self.m, pipe, bg, stage="fine")["render"][None], t[None]).mean()) for c, t in zip(cams, targets)]


def _compare_events(ev, A, B, da, db, thr):
    """After one densification in both legs: same rows in the same order?  If not: the same SET of rows (matched by parent + slot)?  And
    which Gaussians were classified differently, with their margins to BOTH thresholds in BOTH legs."""
    ga, mxa, dense, clone_a, split_a, sel_a, big_a = da
    gb, mxb, _, clone_b, split_b, sel_b, big_b = db
    ka, kb = _row_keys(clone_a, split_a), _row_keys(clone_b, split_b)
    ev["plan_A"] = (int((~split_a).sum()), int(clone_a.sum()), int(split_a.sum()))
    ev["plan_B"] = (int((~split_b).sum()), int(clone_b.sum()), int(split_b.sum()))
    ev["order_equal"] = bool(ka.shape == kb.shape and torch.equal(ka, kb))
    diff = torch.nonzero((clone_a != clone_b) | (split_a != split_b)).flatten()
    ev["n_differently_classified"] = int(diff.numel())
    name = lambda c, s: "split" if bool(s) else ("clone" if bool(c) else "keep")
    ev["differently_classified"] = [
        {"index": int(i), "A": name(clone_a[i], split_a[i]), "B": name(clone_b[i], split_b[i]),
         "grad_decision_A": bool(sel_a[i]), "grad_decision_B": bool(sel_b[i]), "size_decision_A": bool(big_a[i]), "size_decision_B": bool(big_b[i]),
         "grad_margin_A": float(ga[i].double() / thr - 1.0), "grad_margin_B": float(gb[i].double() / thr - 1.0),
         "size_margin_A": float(mxa[i].double() / dense - 1.0), "size_margin_B": float(mxb[i].double() / dense - 1.0)} for i in diff[:16]]
    # rows matched by identity (parent, slot): every row both legs hold
    sa, sb = torch.argsort(ka), torch.argsort(kb)
    ra, rb = sa[torch.isin(ka[sa], kb)], sb[torch.isin(kb[sb], ka)]
    assert torch.equal(ka[ra], kb[rb])
    ev["rows_only_in_one_leg"] = int(ka.numel() - ra.numel()) + int(kb.numel() - rb.numel())
    ev["matched_rows_rel_l2"] = {n: _rel(getattr(B, n).detach()[rb], getattr(A, n).detach()[ra]) for n in NAMES}
    if not torch.equal(split_a, split_b):
        # the children's positions are drawn from the fed samples by (child, rank among the split Gaussians): another split set hands the
        # SAME children other samples -- their positions are comparable only when the split sets agree; the copies (slots 0 / 1) always are
        copies = (ka[ra] % 4) < 2
        ev["matched_rows_rel_l2"]["_xyz"] = _rel(B._xyz.detach()[rb][copies], A._xyz.detach()[ra][copies])
    ev["xyz_rel_l2_after"] = ev["matched_rows_rel_l2"]["_xyz"]
    ev["table_equal"] = bool(torch.equal(A._deformation_table[ra], B._deformation_table[rb]))


def _compare_prunes(ev, ta, tb, opacity_thr, extent, size):
    opa, mxa, ra_, mask_a = ta
    opb, mxb, rb_, mask_b = tb
    ev["pruned"] = (int(mask_a.sum()), int(mask_b.sum()))
    diff = torch.nonzero(mask_a != mask_b).flatten()
    ev["n_differently_pruned"] = int(diff.numel())
    ev["differently_pruned"] = [
        {"index": int(i), "A": bool(mask_a[i]), "B": bool(mask_b[i]),
         "opacity_margin_A": float(opa[i] / opacity_thr - 1.0), "opacity_margin_B": float(opb[i] / opacity_thr - 1.0),
         "screen_margin_A": float(ra_[i] / size - 1.0), "screen_margin_B": float(rb_[i] / size - 1.0),
         "world_margin_A": float(mxa[i] / (0.1 * extent) - 1.0), "world_margin_B": float(mxb[i] / (0.1 * extent) - 1.0)} for i in diff[:16]]


def run(iters=200, interval=50, n=3000, W=128, H=96, prune_at=130, reset_at=120, extent=None, device="cuda:0", grad_quantile=0.8, only_leg_a=False,
        resync_every_event=True, size_quantile=0.35, leg_b="drop-ins", prune_screen_size=None):
    """Runs both legs in lock step.  Returns a report dict (see the test for what is asserted)."""
    from oracle import ref_modules
    fdgs = importlib.import_module("4dgaussians_amd")
    ns = ref_modules.load_train()
    dev = torch.device(device)
    student, cams, targets = fit_proxy.make_problem(n=n, W=W, H=H)
    hyper = student._deformation.args
    opt = train_opt(iters)
    if extent is None:      # scene extent such that percent_dense * extent sits inside the splat-size distribution: clones AND splits happen
        # (round 6: at its 35th percentile instead of the median -- off the densest part of the distribution, so that fewer Gaussians sit
        # within rounding of the clone / split boundary; the ones that do are named with their margins by _compare_events)
        extent = float(torch.quantile(torch.exp(student._scaling).max(1).values, size_quantile)) / opt.percent_dense
    A = _new_model(ns, student, dev, ns.deform_network(hyper))
    if only_leg_a:        # (CPU dry run of the reference side of this harness: tools only)
        return _run_leg_a_only(ns, A, opt, student, cams, targets, iters, interval, prune_at, reset_at, extent, grad_quantile, dev)
    bank = torch.randn(400_000, 3, generator=torch.Generator().manual_seed(99)).to(dev)
    legA = ReferenceLeg(ns, A)
    if leg_b == "reference":
        legB = ReferenceLeg(ns, _new_model(ns, student, dev, ns.deform_network(hyper)))
    else:
        legB = DropInLeg(ns, _new_model(ns, student, dev, fdgs.deform_network(hyper)), fdgs, bank)
    B = legB.m
    for m in (A, B):
        m.training_setup(opt)                                     # the reference's own method on both holders ...
    _copy_state(legB.make_optimizer, A, B)                        # ... then leg B's optimizer becomes FusedAdam over the same groups / state
    tg = [torch.tensor(t, device=dev) for t in targets]
    cg = [c.to(dev) for c in cams]
    pipe, bg = fit_proxy.synthetic.PipelineParams(), torch.zeros(3, device=dev)

    def fed_normal(mean=None, std=None, **k):                     # torch.normal(mean=0, std=stds) = stds * standard-normal samples
        return mean + bank[:std.shape[0]] * std

    rep = {"events": [], "psnr_A": [], "psnr_B": [], "resyncs": 0, "threshold": None, "extent": extent, "leg_b": legB.name}
    thr = None
    opacity_thr = opt.opacity_threshold_fine_init
    for it in range(1, iters + 1):
        v = (it - 1) % len(cg)
        for m in (A, B):
            m.update_learning_rate(it)                            # train.py:172
        rep["psnr_A"].append(legA.frame(cg[v], tg[v], pipe, bg))
        rep["psnr_B"].append(legB.frame(cg[v], tg[v], pipe, bg))
        with torch.no_grad():
            ev = None
            if it % interval == 0 or it == prune_at:
                ev = {"iteration": it, "N_before": (int(A._xyz.shape[0]), int(B._xyz.shape[0]))}
                assert ev["N_before"][0] == ev["N_before"][1]     # (every mismatch is resynchronised below)
                ev["accum_rel_l2"] = _rel(B.xyz_gradient_accum, A.xyz_gradient_accum)
                ev["denom_mismatch_frac"] = float((B.denom != A.denom).float().mean())
                ev["max_radii2D_mismatch_frac"] = float((B.max_radii2D != A.max_radii2D).float().mean())
                ev["xyz_rel_l2"] = _rel(B._xyz, A._xyz)
                ev["opacity_rel_l2"] = _rel(B._opacity, A._opacity)
                rep["events"].append(ev)
            if it % interval == 0:
                if thr is None:     # one threshold for the run, placed where this scene densifies ~20 % of its Gaussians per event
                    ga = A.xyz_gradient_accum / A.denom
                    ga[ga.isnan()] = 0.0
                    thr = rep["threshold"] = float(torch.quantile(ga[ga > 0].flatten(), grad_quantile))
                ev["kind"] = "densify"
                da = legA.densify(thr, opacity_thr, extent, it, fed_normal)                      # train.py:273
                db = legB.densify(thr, opacity_thr, extent, it, fed_normal)
                _compare_events(ev, A, B, da, db, thr)
            if it == prune_at:                                                                      # train.py:277 (guard dropped)
                ev["kind"] = ev.get("kind", "") + "+prune" if "kind" in ev else "prune"
                size = prune_screen_size
                if size is None:    # the reference's 20 px would drop nearly every splat of this small image: the screen-size limit sits
                    # where it drops the largest ~3 % (of leg A's radii; integers -- no ties), next to the transparent ones
                    size = float(torch.quantile(A.max_radii2D[A.max_radii2D > 0], 0.97))
                ev["prune_screen_size"] = size
                ta = legA.prune(thr, opacity_thr, extent, size)
                tb = legB.prune(thr, opacity_thr, extent, size)
                _compare_prunes(ev, ta, tb, opacity_thr, extent, size)
                if ev["n_differently_pruned"] == 0:
                    ev["xyz_rel_l2_after_prune"] = _rel(B._xyz, A._xyz)
            if ev is not None:
                ev["N_after"] = (int(A._xyz.shape[0]), int(B._xyz.shape[0]))
            if it == reset_at:                                                                      # train.py:283
                legA.reset_opacity()
                legB.reset_opacity()
            A.optimizer.step()                                                                      # train.py:291-292
            A.optimizer.zero_grad(set_to_none=True)
            B.optimizer.step()
            B.optimizer.zero_grad(set_to_none=True)
            differs = ev is not None and (ev["N_after"][0] != ev["N_after"][1] or ev.get("n_differently_classified", 0) > 0
                                          or ev.get("n_differently_pruned", 0) > 0)
            if ev is not None and ((resync_every_event and it % interval == 0) or differs):
                # every segment starts from identical state (so that every event's statistics are comparable to rounding); without
                # `resync_every_event` only when the legs hold different sets (a Gaussian on a threshold: named above)
                _copy_state(legB.make_optimizer, A, B)
                rep["resyncs"] += 1
                ev["resynced"] = True
    with torch.no_grad():
        rep["final_psnr_A"] = legA.final_psnr(cg, tg, pipe, bg)
        rep["final_psnr_B"] = legB.final_psnr(cg, tg, pipe, bg)
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
    leg = ReferenceLeg(ns, A)
    thr, ns_ = None, []
    for it in range(1, iters + 1):
        v = (it - 1) % len(cg)
        A.update_learning_rate(it)
        leg.frame(cg[v], tg[v], pipe, bg)
        with torch.no_grad():
            if it % interval == 0:
                ga = A.xyz_gradient_accum / A.denom
                ga[ga.isnan()] = 0.0
                if thr is None:
                    thr = float(torch.quantile(ga[ga > 0].flatten(), grad_quantile))
                leg.densify(thr, 0.005, extent, it, lambda mean=None, std=None, **k: mean + bank[:std.shape[0]] * std)
                ns_.append(int(A._xyz.shape[0]))
            if it == prune_at:
                leg.prune(thr, 0.005, extent, 20)
                ns_.append(int(A._xyz.shape[0]))
            if it == reset_at:
                A.reset_opacity()
            A.optimizer.step()
            A.optimizer.zero_grad(set_to_none=True)
    return {"N": ns_, "threshold": thr}
