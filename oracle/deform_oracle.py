"""CPU restatement of the reference's per-Gaussian deformation step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, in explicit index arithmetic (no F.grid_sample, no nn.Module), what the reference computes in
  scene/hexplane.py:19-20   normalize_aabb            (aabb[0]=max, aabb[1]=min: the axis is flipped)
  scene/hexplane.py:21-46   grid_sample_wrapper       (bilinear, align_corners=True, padding_mode='border')
  scene/hexplane.py:73-106  interpolate_ms_features   (product over the 6 planes, concat over levels)
  scene/deformation.py:67-83,97-148  query_time / forward_dynamic (trunk Linear, 5 heads, out = in + delta)
  gaussian_renderer/__init__.py:97-99  exp / normalize / sigmoid activations (optional here)
Parameters are consumed by their reference `state_dict()` names, so a reference
`deform_network.state_dict()` can be fed straight in.  PINNED: tests/test_oracle_deform.py checks this file
against the reference modules imported from /root/reference and against the golden
vectors in tests/golden/ generated from them by tests/golden/make_deform_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import itertools

import torch
import torch.nn.functional as F

PLANE_PAIRS = list(itertools.combinations(range(4), 2))  # (0,1),(0,2),(0,3),(1,2),(1,3),(2,3)
HEADS = [("pos_deform", "no_dx", 3), ("scales_deform", "no_ds", 3), ("rotations_deform", "no_dr", 4),
         ("opacity_deform", "no_do", 1), ("shs_deform", "no_dshs", 48)]


def _bilinear_border(plane, x, y, dec_x=None, dec_y=None):
    """plane [1,C,Hy,Wx]; x,y in normalised [-1,1] coords ([N]); returns [N,C].
    pixel = ((coord+1)/2)*(size-1), clamped to [0,size-1]; the clamp has zero gradient outside (0,size-1).
    `dec_x` / `dec_y` (oracle/parity.py only; None = the pinned arithmetic above, untouched): (cell_shift [N] int, gate_flip [N] bool) --
    evaluate the SAME continuous interpolant from the neighbouring cell (floor + shift: the value and the plane weights are continuous across a
    cell boundary, the coordinate derivative is not) and / or with the border clamp's derivative gate inverted."""
    _, C, Hy, Wx = plane.shape

    def unnorm(c, size, dec):
        p = ((c + 1.0) / 2.0) * (size - 1)
        inside = (p > 0) & (p < size - 1)
        if dec is None:
            return torch.where(inside, p, p.detach().clamp(0, size - 1))
        gate = (inside ^ dec[1]).to(p.dtype)
        return p.detach().clamp(0, size - 1) + (p - p.detach()) * gate       # value clamped as always; derivative gated

    ix, iy = unnorm(x, Wx, dec_x), unnorm(y, Hy, dec_y)
    x0, y0 = torch.floor(ix.detach()), torch.floor(iy.detach())
    if dec_x is not None:
        x0 = x0 + dec_x[0].to(x0.dtype)
    if dec_y is not None:
        y0 = y0 + dec_y[0].to(y0.dtype)
    wx1, wy1 = ix - x0, iy - y0
    wx0, wy0 = 1.0 - wx1, 1.0 - wy1
    x0i, y0i = x0.long(), y0.long()
    x1i, y1i = (x0i + 1).clamp(max=Wx - 1), (y0i + 1).clamp(max=Hy - 1)
    p = plane[0].permute(1, 2, 0)  # [Hy,Wx,C]
    v00, v01 = p[y0i, x0i], p[y0i, x1i]
    v10, v11 = p[y1i, x0i], p[y1i, x1i]
    return (v00 * (wx0 * wy0)[:, None] + v01 * (wx1 * wy0)[:, None] + v10 * (wx0 * wy1)[:, None] +
            v11 * (wx1 * wy1)[:, None])


def hexplane_features(sd, xyz, t, n_levels, prefix="deformation_net.grid.", decisions=None):
    aabb = sd[prefix + "aabb"]
    pts = (xyz - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0
    q = torch.cat([pts, t.reshape(-1, 1)], dim=-1)
    feats = []
    for lvl in range(n_levels):
        prod = None
        for k, (i, j) in enumerate(PLANE_PAIRS):
            dx = decisions.cell(lvl, i) if decisions is not None else None
            dy = decisions.cell(lvl, j) if decisions is not None else None
            v = _bilinear_border(sd[f"{prefix}grids.{lvl}.{k}"], q[:, i], q[:, j], dx, dy)
            prod = v if prod is None else prod * v
        feats.append(prod)
    return torch.cat(feats, dim=-1)


def count_levels(sd, prefix="deformation_net.grid."):
    n = 0
    while f"{prefix}grids.{n}.0" in sd:
        n += 1
    return n


class KinkDecisions:
    """Per-row overrides of the field's derivative discontinuities (ReLU masks, bilinear cell, border-clamp gate) -- used ONLY by
    oracle/parity.py to evaluate "the same Gaussian with one decision taken the other way"; `deform_forward(decisions=None)` is the
    pinned arithmetic."""

    def __init__(self, n):
        self.n = n
        self.relu_flip = {}      # layer ("trunk" or head name) -> bool [n, width]
        self.cell_shift = {}     # (level, axis) -> int64 [n]
        self.gate_flip = {}      # (level, axis) -> bool [n]
        self.captured = {}       # layer -> (pre-activation, sum |w x| + |b|) of the last forward

    def cell(self, lvl, axis):
        if (lvl, axis) not in self.cell_shift and (lvl, axis) not in self.gate_flip:
            return None
        z = torch.zeros(self.n, dtype=torch.int64)
        return self.cell_shift.get((lvl, axis), z), self.gate_flip.get((lvl, axis), z.bool())

    def relu(self, layer, x, inp=None, w=None, b=None):
        if inp is not None:
            with torch.no_grad():
                self.captured[layer] = (x.detach().clone(), inp.detach().abs() @ w.detach().abs().t() + b.detach().abs())
        f = self.relu_flip.get(layer)
        if f is None:
            return torch.relu(x)
        return x * ((x.detach() > 0) ^ f).to(x.dtype)


def deform_forward(sd, flags, xyz, scales, rotations, opacity, shs, t, activate=False, decisions=None):
    """sd: reference state_dict (tensors, may require grad); flags: object with no_dx/no_ds/no_dr/no_do/no_dshs.
    Returns (means3D, scales, rotations, opacity, shs) like deform_network.forward; with activate=True the
    scales/rotations/opacity are passed through exp / normalize(eps 1e-12) / sigmoid.
    `decisions` (a KinkDecisions; oracle/parity.py only) overrides individual ReLU / cell / clamp decisions per row."""
    feat = hexplane_features(sd, xyz, t, count_levels(sd), decisions=decisions)
    w0, b0 = sd["deformation_net.feature_out.0.weight"], sd["deformation_net.feature_out.0.bias"]
    hidden = F.linear(feat, w0, b0)
    if decisions is None:
        relu_hidden = lambda: torch.relu(hidden)
    else:
        rh = decisions.relu("trunk", hidden, feat, w0, b0)
        relu_hidden = lambda: rh
    ins = [xyz, scales, rotations, opacity, shs]
    outs = []
    for (name, flag, k), x in zip(HEADS, ins):
        if getattr(flags, flag):
            outs.append(x)
            continue
        w1, b1 = sd[f"deformation_net.{name}.1.weight"], sd[f"deformation_net.{name}.1.bias"]
        h = F.linear(relu_hidden(), w1, b1)
        rh1 = torch.relu(h) if decisions is None else decisions.relu(name, h, relu_hidden(), w1, b1)
        d = F.linear(rh1, sd[f"deformation_net.{name}.3.weight"], sd[f"deformation_net.{name}.3.bias"])
        outs.append(x + d.reshape(x.shape))
    pts, sc, rot, op, sh = outs
    if activate:
        sc = torch.exp(sc)
        rot = rot / rot.norm(dim=1, keepdim=True).clamp_min(1e-12)
        op = torch.sigmoid(op)
    return pts, sc, rot, op, sh


def discontinuity_margin(sd, flags, xyz, t):
    """Per Gaussian: distance to the nearest derivative discontinuity of the deformation field = min over
    (|pre-activation| of every ReLU input, distance of every spatial plane coordinate to a texel boundary in pixels).
    Tests use it to pick inputs on which float rounding cannot flip a ReLU or a bilinear cell."""
    with torch.no_grad():
        n_levels = count_levels(sd)
        feat = hexplane_features(sd, xyz, t, n_levels)
        hidden = F.linear(feat, sd["deformation_net.feature_out.0.weight"], sd["deformation_net.feature_out.0.bias"])
        margin = hidden.abs().min(dim=1).values
        for (name, flag, k) in HEADS:
            if getattr(flags, flag):
                continue
            h = F.linear(torch.relu(hidden), sd[f"deformation_net.{name}.1.weight"], sd[f"deformation_net.{name}.1.bias"])
            margin = torch.minimum(margin, h.abs().min(dim=1).values)
        aabb = sd["deformation_net.grid.aabb"]
        pts = (xyz - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0
        for lvl in range(n_levels):
            for axis, plane in ((0, 0), (1, 0), (2, 1)):  # plane (0,1) carries x (width) and y (height); (0,2) carries z
                pl = sd[f"deformation_net.grid.grids.{lvl}.{plane}"]
                size = pl.shape[3] if axis == 0 else pl.shape[2]
                p = ((pts[:, axis] + 1.0) / 2.0) * (size - 1)
                frac = (p - torch.floor(p)).clamp(0, 1)
                dist = torch.minimum(frac, 1 - frac)
                dist = torch.where((p < 0) | (p > size - 1), torch.minimum((p - 0).abs(), (p - (size - 1)).abs()), dist)
                margin = torch.minimum(margin, dist * 0.1)
    return margin


def backward_float64(sd, flags, leaves, t_value, gouts):
    """Gradients of deform_forward(activate=True) w.r.t. every parameter, evaluated in FLOAT64 for given (float32) upstream gradients
    `gouts` = [d/d means3D, d/d scales, d/d rotations, d/d opacity, d/d shs] -- the reference for gradient-parity checks where the
    float32 autograd of this same oracle is not accurate enough to be the judge: the field has kinks (ReLU, border clamp), the
    per-Gaussian gradient magnitudes of a rendered frame are heavy-tailed, and ONE Gaussian whose pre-activation rounds to the
    other side of zero in float32 moves the rel-L2 of a whole tensor by ~1e-3 (measured: 99.99 % of the squared float32-vs-float64
    difference of the 2 M-Gaussian frame sits in ten rows, tools/oracle_f32_vs_f64.py).  Rows whose upstream gradients are all zero
    (culled / occluded Gaussians) contribute exactly zero to every sum and are left out of the evaluation.
    `leaves`: dict of the six Gaussian tensors (_xyz, _scaling, _rotation, _opacity, _features_dc, _features_rest).
    Returns {name: float64 numpy array or None} with the state-dict names prefixed "_deformation."."""
    dt = torch.float64
    n = leaves["_xyz"].shape[0]
    live = torch.zeros(n, dtype=torch.bool)
    for g in gouts:
        live |= (g.reshape(n, -1) != 0).any(dim=1)
    idx = torch.nonzero(live).squeeze(1)
    sd64 = {k: (v.detach().to(dt).requires_grad_(bool(v.requires_grad)) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    sub = {k: v.detach()[idx].to(dt).requires_grad_(True) for k, v in leaves.items()}
    shs = torch.cat([sub["_features_dc"], sub["_features_rest"]], 1)
    outs = deform_forward(sd64, flags, sub["_xyz"], sub["_scaling"], sub["_rotation"], sub["_opacity"], shs,
                          torch.full((idx.numel(), 1), float(t_value), dtype=dt), activate=True)
    g64 = [g.detach()[idx].to(dt).reshape(o.shape) for g, o in zip(gouts, outs)]
    wanted = list(sub.values()) + [v for v in sd64.values() if v.dtype.is_floating_point and v.requires_grad]
    names = list(sub.keys()) + ["_deformation." + k for k, v in sd64.items() if v.dtype.is_floating_point and v.requires_grad]
    grads = torch.autograd.grad(list(outs), wanted, grad_outputs=g64, allow_unused=True)
    out = {}
    for k, g in zip(names, grads):
        if g is None:
            out[k] = None
        elif k in sub:
            full = torch.zeros((n,) + tuple(g.shape[1:]), dtype=dt)
            full[idx] = g
            out[k] = full.numpy()
        else:
            out[k] = g.numpy()
    return out


def import_reference_deform_network():
    """import the reference's own deform_network on CPU (only where /root/reference exists)."""
    import sys
    import types
    REF = "/root/reference"
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if "scene" not in sys.modules or not hasattr(sys.modules["scene"], "__path__") or \
            REF + "/scene" not in list(sys.modules["scene"].__path__):
        pkg = types.ModuleType("scene")
        pkg.__path__ = [REF + "/scene"]
        sys.modules["scene"] = pkg
    if "tkinter" not in sys.modules:
        tk = types.ModuleType("tkinter")
        tk.W = "w"
        sys.modules["tkinter"] = tk
    from scene.deformation import deform_network
    return deform_network
