"""Dense PyTorch (autograd) restatement of the rasterizer -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Independent second statement of SURVEY.md Appendix B, used ONLY to cross-check the analytic backward of
oracle/raster_oracle.c (float64, tiny scenes: memory is O(P * H * W)).  It is written from the maths, not from
the C file: preprocess is vectorised over Gaussians, blending is a masked cumulative product over the
depth-sorted list per pixel.  Gradient conventions that differ from naive autograd are made explicit:

  * alpha = min(0.99, o*G) passes gradient straight through the clamp (Appendix B.4: dL_dG = o*dL_dalpha);
  * a view-space x (y) that was clamped to +-1.3*tanfov gets no gradient, and the clamped value is treated as a
    constant with respect to z (Appendix B.5 (i));
  * conic = inverse(cov2D) uses k = 1/(denom^2 + 1e-7) in its backward (Appendix B.5 (i)).

Reference call site this stands in for: gaussian_renderer/__init__.py:120-128 (rasterizer(...) -> image, radii,
depth).  Only tests/ may import this module.
"""
import math

import torch

TILE = 16
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


class _ConicInverse(torch.autograd.Function):
    """(a,b,c) -> (c,-b,a)/det with the reference's eps-regularised backward and half-weight xy convention."""

    @staticmethod
    def forward(ctx, a, b, c):
        det = a * c - b * b
        ctx.save_for_backward(a, b, c)
        return c / det, -b / det, a / det

    @staticmethod
    def backward(ctx, gx, gy_full, gz):
        a, b, c = ctx.saved_tensors
        gy = 0.5 * gy_full  # the kernel accumulates half of d/d(conic_xy)
        denom = a * c - b * b
        k = 1.0 / (denom * denom + 1e-7)
        da = k * (-c * c * gx + 2 * b * c * gy + (denom - a * c) * gz)
        dc = k * (-a * a * gz + 2 * a * b * gy + (denom - a * c) * gx)
        db = k * 2 * (b * c * gx - (denom + 2 * b * b) * gy + a * b * gz)
        return da, db, dc


def quat_to_rot(q):
    r, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)
    return R.reshape(-1, 3, 3)


def sh_to_rgb(deg, sh, dirs):
    """sh [P,K,3], dirs [P,3] unit -> [P,3] (before +0.5 / clamp); polynomial of utils/sh_utils.py:74-100."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
               + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
               + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def rasterize(*, means3D, opacities, viewmatrix, projmatrix, campos, bg, image_height, image_width, tanfovx, tanfovy,
              sh_degree=0, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
              scale_modifier=1.0, means2D=None):
    """Returns (color [3,H,W], depth [1,H,W], radii [P] int32). `means2D` ([P,3], optional) is the reference's
    gradient sink: its x,y are added (as zeros) in NDC units so that means2D.grad is d L / d ndc."""
    dt, dev = means3D.dtype, means3D.device
    H, W = int(image_height), int(image_width)
    P = means3D.shape[0]
    V = viewmatrix.reshape(4, 4).to(dt)  # row-vector convention: p_view = [p,1] @ V
    PM = projmatrix.reshape(4, 4).to(dt)
    focal_x, focal_y = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    ones = torch.ones(P, 1, dtype=dt, device=dev)
    ph4 = torch.cat([means3D, ones], 1)
    pv = (ph4 @ V)[:, :3]
    phom = ph4 @ PM
    pw = 1.0 / (phom[:, 3] + 1e-7)
    ndc = phom[:, :2] * pw[:, None]
    if means2D is not None:
        ndc = ndc + means2D[:, :2]
    in_front = pv[:, 2] > 0.2
    # cov3D
    if cov3D_precomp is None:
        R = quat_to_rot(rotations)
        L = R * (scale_modifier * scales)[:, None, :]
        Sigma = L @ L.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], -1).reshape(-1, 3, 3)
    # EWA
    tz = pv[:, 2]
    tz_safe = torch.where(in_front, tz, torch.ones_like(tz))
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = pv[:, 0] / tz_safe, pv[:, 1] / tz_safe
    xcl = (txtz < -limx) | (txtz > limx)
    ycl = (tytz < -limy) | (tytz > limy)
    tx = torch.where(xcl, (txtz.clamp(-limx, limx) * tz_safe).detach(), pv[:, 0])
    ty = torch.where(ycl, (tytz.clamp(-limy, limy) * tz_safe).detach(), pv[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([focal_x / tz_safe, zero, -(focal_x * tx) / (tz_safe * tz_safe),
                     zero, focal_y / tz_safe, -(focal_y * ty) / (tz_safe * tz_safe)], -1).reshape(-1, 2, 3)
    Rv = V[:3, :3].t()  # p_view = Rv p + t
    Mm = J @ Rv
    cov2 = Mm @ Sigma @ Mm.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c_ = cov2[:, 1, 1] + 0.3
    det = a * c_ - b * b
    ok = in_front & (det != 0)
    det_s = torch.where(ok, det, torch.ones_like(det))
    a_s, b_s, c_s = torch.where(ok, a, torch.ones_like(a)), torch.where(ok, b, zero), torch.where(ok, c_, torch.ones_like(a))
    conx, cony, conz = _ConicInverse.apply(a_s, b_s, c_s)
    mid = 0.5 * (a + c_)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det_s, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    trunc = lambda v: torch.trunc(v).to(torch.int64)
    rx0 = trunc((px.detach() - radius) / TILE).clamp(0, gx)
    ry0 = trunc((py.detach() - radius) / TILE).clamp(0, gy)
    rx1 = trunc((px.detach() + radius + TILE - 1) / TILE).clamp(0, gx)
    ry1 = trunc((py.detach() + radius + TILE - 1) / TILE).clamp(0, gy)
    vis = ok & ((rx1 - rx0) * (ry1 - ry0) > 0)
    radii = torch.where(vis, radius, torch.zeros_like(radius)).to(torch.int32)
    # colour
    if colors_precomp is None:
        d = means3D - campos.to(dt)[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(sh_to_rgb(sh_degree, shs.reshape(P, -1, 3), d) + 0.5, 0.0)
    else:
        rgb = colors_precomp
    # depth order (stable; ties by index like the radix sort on (tile|depth) keys)
    idx = torch.nonzero(vis).squeeze(1)
    order = torch.sort(tz[idx].detach(), stable=True).indices
    idx = idx[order]
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    xs, ys = xs.reshape(-1), ys.reshape(-1)
    tilex, tiley = xs // TILE, ys // TILE
    dx = px[idx, None] - xs[None].to(dt)
    dy = py[idx, None] - ys[None].to(dt)
    power = -0.5 * (conx[idx, None] * dx * dx + conz[idx, None] * dy * dy) - cony[idx, None] * dx * dy
    Gv = torch.exp(torch.clamp(power, max=0.0))
    raw = opacities.reshape(-1)[idx, None] * Gv
    alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()  # straight-through clamp
    in_rect = ((tilex[None] >= rx0[idx, None]) & (tilex[None] < rx1[idx, None]) & (tiley[None] >= ry0[idx, None]) &
               (tiley[None] < ry1[idx, None]))
    m = in_rect & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
    a_eff = torch.where(m, alpha, torch.zeros_like(alpha))
    one_minus = 1.0 - a_eff
    Tincl = torch.cumprod(one_minus, dim=0)
    Texcl = torch.cat([torch.ones_like(Tincl[:1]), Tincl[:-1]], 0)
    stop = m & ((Texcl * one_minus).detach() < 1e-4)
    stopped = torch.cummax(stop.to(torch.int8), dim=0).values.bool()
    use = m & ~stopped
    w = torch.where(use, a_eff * Texcl, torch.zeros_like(a_eff))
    T_final = torch.where(use, one_minus, torch.ones_like(one_minus)).prod(dim=0)
    color = (w[:, None, :] * rgb[idx][:, :, None]).sum(0) + T_final[None] * bg.to(dt)[:, None]
    depth = (w * tz[idx, None]).sum(0)
    return color.reshape(3, H, W), depth.reshape(1, H, W), radii
