"""Real spherical-harmonics basis up to degree 3 as torch ops (the `pipe.convert_SHs_python` path of
gaussian_renderer/__init__.py:106-111; polynomial of utils/sh_utils.py:57-112 of the reference)."""
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def basis(deg, dirs):
    """dirs [...,3] unit -> list of (deg+1)^2 tensors [...,1]."""
    x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
    b = [C0 + 0 * x]
    if deg > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz = x * x, y * y, z * z
        b += [C2[0] * x * y, C2[1] * y * z, C2[2] * (2.0 * zz - xx - yy), C2[3] * x * z, C2[4] * (xx - yy)]
    if deg > 2:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * x * y * z, C3[2] * y * (4 * zz - xx - yy),
              C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
              C3[6] * x * (xx - 3 * yy)]
    return b


def eval_sh(deg, sh, dirs):
    """sh [..., C, K], dirs [..., 3] -> [..., C]"""
    assert 0 <= deg <= 3 and sh.shape[-1] >= (deg + 1) ** 2
    out = 0
    for k, bk in enumerate(basis(deg, dirs)):
        out = out + bk * sh[..., k]
    return out


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
