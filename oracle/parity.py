"""Gradient parity with per-row kink attribution -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

north_star: "grads within 1e-3 rel-L2".  The deformation field (scene/hexplane.py:21-46 bilinear + border clamp, scene/deformation.py:45-65
ReLU MLP) has derivative discontinuities, the per-Gaussian gradient magnitudes of a rendered frame are heavy-tailed, and ONE Gaussian whose
pre-activation sits within float32 rounding of zero moves a whole tensor's rel-L2 by ~1e-3 when an implementation rounds it to the other side
(tools/oracle_f32_vs_f64.py).  Round 3 tolerated that by accepting "the closer of the float32 and float64 evaluations of the oracle".  This
module replaces that with an attribution that names the rows:

  reference  = float64 evaluation of the pinned oracle (deform_oracle.backward_float64) -- ONE reference, always.
  K          = rows whose per-Gaussian gradient differs from the reference by more than `row_tol` x the tensor's norm.
  a row of K is ATTRIBUTED only when it is a PROVEN kink row: in the float64 forward of that row some ReLU pre-activation lies within the
               float32 forward error of zero (or a plane coordinate within rounding of a texel boundary / the border), and the
               implementation's row equals -- row-wise, to `variant_tol` = 1e-3 -- the float64 evaluation of that SAME row with a subset of
               exactly those decisions taken the other way (deform_oracle.KinkDecisions).  Every attributed row REPORTS THE MARGIN that
               admitted it: |pre-activation| in units of u32 * (sum |w_i x_i| + |b|) for a ReLU, the distance to the texel boundary in
               texels and in float32 ulps of the axis extent for a cell / border decision.
  windows    = what a float32 evaluation of the REFERENCE'S OWN arithmetic is measured to differ from its float64 evaluation by
               (tools/parity_windows.py, profiles/r05_parity_windows_float32_vs_float64.txt; asserted by
               tests/test_oracle_parity.py::test_windows_are_what_float32_rounding_justifies): trunk pre-activations by up to 119 .. 348
               u32-units over 20 k .. 100 k Gaussians and three deformation configs (the inputs of the trunk are products of six bilinear
               samples whose weights move with the coordinate's rounding, so the error is far above the dot product's own ~sqrt(K) u; it
               is a tail, the maximum grows slowly with the sample), head pre-activations by up to 39 .. 101, coordinates by 2.2 ulp32 of
               the axis extent = 8.3e-6 (64 texels) .. 4.1e-5 texels (256 texels).  Windows: RELU_WINDOW = 512 (trunk) / 160 (heads)
               u32-units, CELL_EPS = 1e-4 texels -- 1.5x .. 2.5x the measured maxima.  (Round 4: 256 / 256 u32-units -- too tight for
               the trunk by this measurement, 1.6x too wide for the heads --, 2e-3 texels -- 50x too wide --, row-wise tolerance 2e-3.)  Any other row of K is left as it is (`unexplained_rows`: in a
               rendered frame, Gaussians whose upstream gradient moved with one of the rasterizer's alpha >= 1/255 decisions), counts fully in
               the group figures, and fails on its own above `unexplained_tol` = 5e-4 of a tensor's norm.
  groups     = every parameter group is then compared with the float64 reference in which the ATTRIBUTED rows (their rows of the per-Gaussian
               tensors, their contributions to the plane / MLP sums) are replaced by the matched variant: <= 1e-3 rel-L2; the number of
               attributed rows is bounded and every row is printed.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / parity leg may import this module.
"""
import itertools
import math

import numpy as np
import torch

from . import deform_oracle as DO

PER_GAUSSIAN = ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")
U32 = 2.0 ** -24           # float32 unit round-off


def _rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / d) if d > 0 else float(np.linalg.norm(a))


def group_keys(grads):
    """Parameter groups as the reference's optimizer has them (scene/gaussian_model.py:165-183), planes and MLP pooled."""
    return {"xyz": ["_xyz"], "scaling": ["_scaling"], "rotation": ["_rotation"], "opacity": ["_opacity"], "f_dc": ["_features_dc"],
            "f_rest": ["_features_rest"],
            "planes": [k for k in grads if "grids" in k and not k.startswith("__") and grads[k] is not None],
            "mlp": [k for k in grads if k.startswith("_deformation.") and "grids" not in k and grads[k] is not None]}


def _row_eval(sd64, flags, leaves, row, t_value, gouts, dec):
    """float64 gradients of ONE Gaussian's deformation (+ activations) for its upstream gradients, under decision overrides `dec`
    (None = as evaluated).  Returns {name: numpy} -- rows [1, ...] for the per-Gaussian tensors, full-size arrays for planes / MLP."""
    dt = torch.float64
    sub = {k: v.detach()[row:row + 1].to(dt).requires_grad_(True) for k, v in leaves.items()}
    shs = torch.cat([sub["_features_dc"], sub["_features_rest"]], 1)
    outs = DO.deform_forward(sd64, flags, sub["_xyz"], sub["_scaling"], sub["_rotation"], sub["_opacity"], shs,
                             torch.full((1, 1), float(t_value), dtype=dt), activate=True, decisions=dec)
    g = [x.detach()[row:row + 1].to(dt).reshape(o.shape) for x, o in zip(gouts, outs)]
    wanted = list(sub.values()) + [v for v in sd64.values() if v.dtype.is_floating_point and v.requires_grad]
    names = list(sub.keys()) + ["_deformation." + k for k, v in sd64.items() if v.dtype.is_floating_point and v.requires_grad]
    grads = torch.autograd.grad(list(outs), wanted, grad_outputs=g, allow_unused=True)
    return {k: (None if x is None else x.numpy()) for k, x in zip(names, grads)}


RELU_WINDOW = {"trunk": 512.0, "head": 160.0}     # u32-units of sum |w x| + |b| (module docstring: measured up to 348 / 101 for the reference itself)
CELL_EPS = 1e-4                                    # texels (measured 8.3e-6 .. 4.1e-5 for the reference itself on 64 .. 256-texel axes)


def _ulp32(x):
    x = abs(float(x))
    return 2.0 ** (math.floor(math.log2(x)) - 23) if x > 0 else 2.0 ** -149


def kink_items(sd64, flags, leaves, row, t_value, relu_window=None, cell_eps=None, max_items=7):
    """The decisions of row `row` a float32 evaluation may legitimately take the other way: ReLU inputs with
    |pre-activation| <= relu_window[layer] * u32 * (sum |w_i x_i| + |b|), spatial plane coordinates within `cell_eps` texels of a texel
    boundary / the border (windows: module docstring).
    Returns [(rel, kind, key, payload, margin)], nearest first (rel = margin / window), at most `max_items`; `margin` is what the report
    prints: {"u32_units": ...} for a ReLU, {"texels": ..., "ulp32_of_extent": ...} for a cell / border decision."""
    relu_window = RELU_WINDOW if relu_window is None else relu_window
    cell_eps = CELL_EPS if cell_eps is None else cell_eps
    dt = torch.float64
    dec = DO.KinkDecisions(1)
    with torch.no_grad():
        x = leaves["_xyz"].detach()[row:row + 1].to(dt)
        z = lambda k: leaves[k].detach()[row:row + 1].to(dt)
        DO.deform_forward(sd64, flags, x, z("_scaling"), z("_rotation"), z("_opacity"), torch.cat([z("_features_dc"), z("_features_rest")], 1),
                          torch.full((1, 1), float(t_value), dtype=dt), decisions=dec)
        items = []
        for layer, (pre, absdot) in dec.captured.items():
            win = relu_window["trunk" if layer == "trunk" else "head"]
            units = (pre.abs() / (absdot * U32).clamp_min(1e-300))[0]
            for j in torch.nonzero(units <= win).flatten().tolist():
                items.append((float(units[j]) / win, "relu", layer, j, {"u32_units": float(f"{float(units[j]):.3g}"), "window": win}))
        aabb = sd64["deformation_net.grid.aabb"]
        pts = ((x - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0)[0]
        for lvl in range(DO.count_levels(sd64)):
            for axis, plane in ((0, 0), (1, 0), (2, 1)):            # plane (0,1) carries x (width) and y (height); (0,2) carries z
                pl = sd64[f"deformation_net.grid.grids.{lvl}.{plane}"]
                size = pl.shape[3] if axis == 0 else pl.shape[2]
                p = float((pts[axis] + 1.0) / 2.0 * (size - 1))
                inside = 0 < p < size - 1
                mg = lambda d: {"texels": float(f"{d:.3g}"), "ulp32_of_extent": float(f"{d / _ulp32(size - 1):.3g}"), "window_texels": cell_eps}
                if min(abs(p), abs(p - (size - 1))) <= cell_eps:
                    d = min(abs(p), abs(p - (size - 1)))
                    items.append((d / cell_eps, "gate", (lvl, axis), None, mg(d)))
                if inside:
                    fl = math.floor(p)
                    frac = p - fl
                    if frac <= cell_eps and fl - 1 >= 0:
                        items.append((frac / cell_eps, "cell", (lvl, axis), -1, mg(frac)))
                    elif 1 - frac <= cell_eps and fl + 1 <= size - 2:
                        items.append(((1 - frac) / cell_eps, "cell", (lvl, axis), +1, mg(1 - frac)))
    items.sort(key=lambda it: it[0])
    return items[:max_items]


def _widest(kink_rows):
    """The largest margin that admitted a decision of an attributed row, per kind (None when no row was attributed)."""
    out = {"relu_u32_units": None, "cell_texels": None}
    for r in kink_rows:
        for m in r.get("margins", []):
            if "u32_units" in m:
                out["relu_u32_units"] = max(out["relu_u32_units"] or 0.0, m["u32_units"])
            else:
                out["cell_texels"] = max(out["cell_texels"] or 0.0, m["texels"])
    return out


def _decisions_for(subset, width_of):
    dec = DO.KinkDecisions(1)
    for (_, kind, key, payload, _m) in subset:
        if kind == "relu":
            f = dec.relu_flip.setdefault(key, torch.zeros(1, width_of[key], dtype=torch.bool))
            f[0, payload] = True
        elif kind == "cell":
            dec.cell_shift[key] = torch.tensor([payload], dtype=torch.int64)
        else:
            dec.gate_flip[key] = torch.tensor([True])
    return dec


def attribute(sd, flags, leaves, t_value, gouts, impl, ref64, row_tol=1e-4, variant_tol=1e-3, tol=1e-3, max_rows=None, unexplained_tol=5e-4):
    """Compare an implementation's gradients `impl` ({name: numpy}) with the float64 reference `ref64` (deform_oracle.backward_float64 of the
    same `sd`, `leaves`, `gouts`) under the rule in the module docstring.  Returns a report dict; report["ok"] says whether every assertion
    holds (callers assert on it and print report["failures"]).  `sd`: state_dict tensors of the oracle chain (requires_grad marks the
    differentiated ones); `leaves`: the six per-Gaussian tensors; `gouts`: float32 upstream gradients of the five deformation outputs."""
    n = leaves["_xyz"].shape[0]
    if max_rows is None:
        max_rows = max(4, int(math.ceil(2e-5 * n)))
    sd64 = {k: (v.detach().to(torch.float64).requires_grad_(bool(v.requires_grad)) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    groups = group_keys(ref64)
    raw = {g: _rel(np.concatenate([np.asarray(impl[k]).ravel() for k in ks]), np.concatenate([ref64[k].ravel() for k in ks]))
           for g, ks in groups.items() if ks}
    # ---- K: rows that differ by more than row_tol x the tensor norm (only _xyz passes through the field; the others are checked too)
    cand = {}
    for k in PER_GAUSSIAN:
        if ref64.get(k) is None:
            continue
        a, b = np.asarray(impl[k], np.float64).reshape(n, -1), ref64[k].reshape(n, -1)
        d = np.linalg.norm(a - b, axis=1)
        for r in np.nonzero(d > row_tol * max(np.linalg.norm(b), 1e-300))[0].tolist():
            cand[r] = max(cand.get(r, 0.0), float(d[r] / max(np.linalg.norm(b), 1e-300)))
    rows = sorted(cand, key=lambda r: -cand[r])
    failures, kink_rows, heavy_rows, unexplained = [], [], [], []
    width_of = {"trunk": sd64["deformation_net.feature_out.0.weight"].shape[0]}
    for (name, flag, k) in DO.HEADS:
        width_of[name] = sd64[f"deformation_net.{name}.1.weight"].shape[0] if f"deformation_net.{name}.1.weight" in sd64 else 0
    expected = {k: (None if v is None else np.array(v, np.float64, copy=True)) for k, v in ref64.items() if not k.startswith("__")}
    for r in rows[:max_rows * 8]:
        base = _row_eval(sd64, flags, leaves, r, t_value, gouts, None)
        row_impl = np.concatenate([np.asarray(impl[k], np.float64).reshape(n, -1)[r] for k in PER_GAUSSIAN if base.get(k) is not None])

        def err(var):
            row_var = np.concatenate([var[k].reshape(-1) for k in PER_GAUSSIAN if var.get(k) is not None])
            return float(np.linalg.norm(row_impl - row_var) / max(np.linalg.norm(row_var), 1e-300))

        e0 = err(base)
        if e0 <= tol:
            # a row that carries a large share of the tensor's norm and is itself accurate row-wise: ordinary rounding, nothing to attribute
            heavy_rows.append({"row": int(r), "diff_over_tensor_norm": float(f"{cand[r]:.3e}"), "row_rel_l2": float(f"{e0:.3e}")})
            continue
        items = kink_items(sd64, flags, leaves, r, t_value)
        best = (e0, (), base)
        for m in range(1, len(items) + 1):
            for subset in itertools.combinations(items, m):
                var = _row_eval(sd64, flags, leaves, r, t_value, gouts, _decisions_for(subset, width_of))
                e = err(var)
                if e < best[0]:
                    best = (e, subset, var)
            if best[0] <= variant_tol:
                break                       # smallest explaining subset
        e, subset, var = best
        kink_rows.append({"row": int(r), "diff_over_tensor_norm": float(f"{cand[r]:.3e}"), "near_kink_decisions": len(items),
                          "row_rel_l2_unflipped": float(f"{e0:.3e}"),
                          "matched": [f"{kind}:{key}:{payload}" for (_, kind, key, payload, _m) in subset],
                          "margins": [m_ for (_, _k, _key, _p, m_) in subset],       # what admitted each matched decision (units: kink_items)
                          "row_rel_l2_to_matched_variant": float(f"{e:.3e}")})
        # A row no kink variant explains differs for another reason -- in a rendered frame: the rasterizer upstream took an alpha >= 1/255 /
        # T < 1e-4 decision the other way on one of the Gaussian's pixels (the image comparison counts those pixels), which moves that
        # Gaussian's upstream gradient.  Such rows are REPORTED and count fully in the group figures below; one that carries more than
        # `unexplained_tol` of a tensor's norm on its own (half the tolerance) fails here and now.
        if (not items or e > variant_tol):
            unexplained.append(kink_rows.pop())
            if cand[r] > unexplained_tol:
                failures.append(f"row {r} differs from the float64 reference by {cand[r]:.2e} of the tensor norm ({e0:.2e} row-wise) and no kink "
                                f"variant explains it ({len(items)} near-kink decisions, best variant {e:.2e})")
        if subset and e <= variant_tol:
            for k, v in var.items():
                if v is None or expected.get(k) is None:
                    continue
                if k in PER_GAUSSIAN:
                    expected[k].reshape(n, -1)[r] = v.reshape(-1)
                else:
                    expected[k] += v - base[k]
    if len(kink_rows) > max_rows or len(rows) > max_rows * 8:
        failures.append(f"{len(kink_rows)} kink rows / {len(rows)} rows over {row_tol:g} of their tensor's norm (allowed: {max_rows} kink rows)")
    attributed = {g: _rel(np.concatenate([np.asarray(impl[k]).ravel() for k in ks]), np.concatenate([expected[k].ravel() for k in ks]))
                  for g, ks in groups.items() if ks}
    for g, v in attributed.items():
        if not v <= tol:
            failures.append(f"group {g}: rel-L2 {v:.2e} > {tol:g} against the float64 reference with the kink rows attributed")
    return {"ok": not failures, "failures": failures, "reference": "float64 evaluation of the pinned oracle (deform_oracle.backward_float64)",
            "grad_rel_l2_vs_float64_raw": {g: float(f"{v:.3e}") for g, v in raw.items()},
            "grad_rel_l2_vs_float64_kink_rows_attributed": {g: float(f"{v:.3e}") for g, v in attributed.items()},
            "kink_rows": kink_rows, "n_kink_rows": len(kink_rows), "max_kink_rows": max_rows, "n_gaussians": n,
            "heavy_rows_within_tol_rowwise": heavy_rows[:8], "unexplained_rows": unexplained[:8], "n_unexplained_rows": len(unexplained),
            "windows": {"relu_u32_units": dict(RELU_WINDOW), "cell_texels": CELL_EPS, "variant_tol": variant_tol,
                        "widest_margin_admitted": _widest(kink_rows)},
            "rule": f"a row differing by > {row_tol:g} of a tensor's norm is attributed only if it equals (row-wise, <= {variant_tol:g}) the float64 evaluation of "
                    f"the same Gaussian with near-zero ReLU (|pre| <= {RELU_WINDOW['trunk']:g} / {RELU_WINDOW['head']:g} u32 * sum|wx|, trunk / heads) or "
                    f"texel-boundary (<= {CELL_EPS:g} texel) decisions flipped -- each row prints its margin; other such rows are listed as unexplained, count in the group "
                    f"figures and fail above {unexplained_tol:g} on their own; groups <= {tol:g} with the attributed rows replaced by the matched variant"}
