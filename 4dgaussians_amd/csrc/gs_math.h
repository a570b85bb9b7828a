// gs_math.h -- per-Gaussian projection arithmetic (forward and backward) of the HIP kernels in preprocess.hip (written host / device
// neutral: plain inline functions; checked stage by stage against the oracle through the kernels, tests/test_gpu_raster.py).
//
// Arithmetic follows SURVEY.md Appendix B.1 / B.5 (the un-vendored rasterizer the reference calls at
// gaussian_renderer/__init__.py:120-128); the in-tree formulas it must agree with are
// utils/general_utils.py:84-116 (quaternion -> R, L = R S, Sigma = L L^T) and utils/sh_utils.py:57-112 (SH basis).
// The backward is derived here in matrix form (M = J*Rv, cov2D = M Sigma M^T): see DESIGN.md section "K8".
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define FDGS_HD __host__ __device__ __forceinline__
#else
#define FDGS_HD inline
#endif

namespace fdgs {

// named constants of the reference arithmetic
#define FDGS_NEAR_CULL 0.2f
#define FDGS_W_EPS 0.0000001f
#define FDGS_FOV_CLAMP 1.3f
#define FDGS_DILATION 0.3f
#define FDGS_LAMBDA_FLOOR 0.1f
#define FDGS_ALPHA_MAX 0.99f
#define FDGS_ALPHA_MIN (1.0f / 255.0f)
#define FDGS_T_STOP 0.0001f
#define FDGS_DENOM_EPS 0.0000001f

#define FDGS_SH_C0 0.28209479177387814f
#define FDGS_SH_C1 0.4886025119029199f
#define FDGS_SH_C2_0 1.0925484305920792f
#define FDGS_SH_C2_1 -1.0925484305920792f
#define FDGS_SH_C2_2 0.31539156525252005f
#define FDGS_SH_C2_3 -1.0925484305920792f
#define FDGS_SH_C2_4 0.5462742152960396f
#define FDGS_SH_C3_0 -0.5900435899266435f
#define FDGS_SH_C3_1 2.890611442640554f
#define FDGS_SH_C3_2 -0.4570457994644658f
#define FDGS_SH_C3_3 0.3731763325901154f
#define FDGS_SH_C3_4 -0.4570457994644658f
#define FDGS_SH_C3_5 1.445305721320277f
#define FDGS_SH_C3_6 -0.5900435899266435f

struct CamConst {
    float view[16], proj[16], campos[3];
    float tanfovx, tanfovy, focal_x, focal_y, scale_mod;
    int W, H, gx, gy, D, M;
};

FDGS_HD void view_xform(const float* m, const float* p, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
FDGS_HD float proj_w(const float* m, const float* p) { return m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15]; }

FDGS_HD void quat_to_R(const float* q, float* R) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = (R S)(R S)^T, S = diag(mod*s); c6 = (00,01,02,11,12,22)
FDGS_HD void cov3d_from_scale_rot(const float* s, float mod, const float* q, float* c6) {
    float R[9], L[9];
    quat_to_R(q, R);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) L[3 * i + k] = R[3 * i + k] * (mod * s[k]);
    c6[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
    c6[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
    c6[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
    c6[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
    c6[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
    c6[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
}

// M = J * Rv (2x3) with the +-1.3*tanfov clamp; tc = clamped view-space mean
FDGS_HD void ewa_M(const CamConst& c, const float* pv, float* Mx, float* tc, bool* xcl, bool* ycl) {
    float limx = FDGS_FOV_CLAMP * c.tanfovx, limy = FDGS_FOV_CLAMP * c.tanfovy;
    float txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
    *xcl = (txtz < -limx || txtz > limx);
    *ycl = (tytz < -limy || tytz > limy);
    tc[0] = fminf(limx, fmaxf(-limx, txtz)) * pv[2];
    tc[1] = fminf(limy, fmaxf(-limy, tytz)) * pv[2];
    tc[2] = pv[2];
    float J00 = c.focal_x / tc[2], J02 = -(c.focal_x * tc[0]) / (tc[2] * tc[2]);
    float J11 = c.focal_y / tc[2], J12 = -(c.focal_y * tc[1]) / (tc[2] * tc[2]);
    const float* v = c.view;  // Rv[i][j] = v[4*j+i]
#pragma unroll
    for (int j = 0; j < 3; j++) {
        Mx[j] = J00 * v[4 * j + 0] + J02 * v[4 * j + 2];
        Mx[3 + j] = J11 * v[4 * j + 1] + J12 * v[4 * j + 2];
    }
}

// cov2D (with dilation) from M and Sigma; also returns MS = M*Sigma (2x3) for the backward
FDGS_HD void cov2d_from(const float* Mx, const float* c6, float* abc, float* MS) {
    const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) MS[3 * r + j] = Mx[3 * r] * S[j] + Mx[3 * r + 1] * S[3 + j] + Mx[3 * r + 2] * S[6 + j];
    abc[0] = MS[0] * Mx[0] + MS[1] * Mx[1] + MS[2] * Mx[2] + FDGS_DILATION;
    abc[1] = MS[0] * Mx[3] + MS[1] * Mx[4] + MS[2] * Mx[5];
    abc[2] = MS[3] * Mx[3] + MS[4] * Mx[4] + MS[5] * Mx[5] + FDGS_DILATION;
}

FDGS_HD float ndc2pix(float v, int S) { return ((v + 1.0f) * (float)S - 1.0f) * 0.5f; }

struct GeoOut {
    float depth, px, py, conic[3];
    int radius, tiles;
    uint32_t rect_min, rect_max;  // x | y<<16
};

// Appendix B.1 (everything except colour). Returns false when the Gaussian is culled.
FDGS_HD bool project_gaussian(const CamConst& c, const float* p, const float* c6, GeoOut* o) {
    float pv[3];
    view_xform(c.view, p, pv);
    if (pv[2] <= FDGS_NEAR_CULL) return false;
    float ph[3];
    view_xform(c.proj, p, ph);
    float pw = 1.0f / (proj_w(c.proj, p) + FDGS_W_EPS);
    float Mx[6], tc[3], abc[3], MS[6];
    bool xc, yc;
    ewa_M(c, pv, Mx, tc, &xc, &yc);
    cov2d_from(Mx, c6, abc, MS);
    float det = abc[0] * abc[2] - abc[1] * abc[1];
    if (det == 0.0f) return false;
    float det_inv = 1.0f / det;
    o->conic[0] = abc[2] * det_inv; o->conic[1] = -abc[1] * det_inv; o->conic[2] = abc[0] * det_inv;
    float mid = 0.5f * (abc[0] + abc[2]);
    float sq = sqrtf(fmaxf(FDGS_LAMBDA_FLOOR, mid * mid - det));
    float rad = ceilf(3.0f * sqrtf(fmaxf(mid + sq, mid - sq)));
    float px = ndc2pix(ph[0] * pw, c.W), py = ndc2pix(ph[1] * pw, c.H);
    int rx0 = (int)((px - rad) / (float)FDGS_TILE), ry0 = (int)((py - rad) / (float)FDGS_TILE);
    int rx1 = (int)((px + rad + (float)(FDGS_TILE - 1)) / (float)FDGS_TILE);
    int ry1 = (int)((py + rad + (float)(FDGS_TILE - 1)) / (float)FDGS_TILE);
    rx0 = rx0 < 0 ? 0 : (rx0 > c.gx ? c.gx : rx0); ry0 = ry0 < 0 ? 0 : (ry0 > c.gy ? c.gy : ry0);
    rx1 = rx1 < 0 ? 0 : (rx1 > c.gx ? c.gx : rx1); ry1 = ry1 < 0 ? 0 : (ry1 > c.gy ? c.gy : ry1);
    int area = (rx1 - rx0) * (ry1 - ry0);
    if (area == 0) return false;
    o->depth = pv[2]; o->px = px; o->py = py; o->radius = (int)rad; o->tiles = area;
    o->rect_min = (uint32_t)rx0 | ((uint32_t)ry0 << 16);
    o->rect_max = (uint32_t)rx1 | ((uint32_t)ry1 << 16);
    return true;
}

// ---- exact tile culling ------------------------------------------------------------------------------------------
// The reference lists a Gaussian for every tile of the bounding square of its 3-sigma circle (Appendix B.1/B.2), but a
// pixel only blends it when alpha = min(0.99, opacity * exp(-q)) >= 1/255, q(d) = 0.5 (a dx^2 + c dy^2) + b dx dy, i.e.
// inside the ellipse q <= ln(255 * opacity).  A (Gaussian, tile) pair whose tile rectangle misses that ellipse changes
// neither T nor any accumulator of any pixel, forward or backward: dropping it leaves every output bit-identical and
// removes it from the sort and from both blending kernels.  The test is conservative: the ellipse is grown by a margin
// far above the rounding of the per-pixel evaluation, the tile is treated as a continuous rectangle, and anything
// numerically odd (non-positive-definite conic, NaN) keeps the reference's full list.
struct CullEllipse {
    float a, b, c, t2a, det, X, Y;   // q-ellipse, 2*a*t, a*c - b^2, half extents along x and y
    int mode;                        // 0: no pixel can pass, 1: test rows, 2: keep every tile of the square
};
FDGS_HD CullEllipse cull_setup(const float* conic, float opacity) {
    CullEllipse E;
    E.a = conic[0]; E.b = conic[1]; E.c = conic[2];
    E.det = E.a * E.c - E.b * E.b;
    E.mode = 2; E.t2a = 0.f; E.X = 0.f; E.Y = 0.f;
    if (!(opacity > 0.0f)) { E.mode = 0; return E; }          // alpha <= 0 < 1/255 everywhere
    if (!(E.a > 0.0f && E.c > 0.0f && E.det > 0.0f)) return E;
    float t = logf(255.0f * opacity);
    t = t + 1e-3f + 1e-3f * fabsf(t);                          // alpha threshold lowered by > 0.1 %
    if (!(t == t)) return E;
    if (t < 0.0f) { E.mode = 0; return E; }
    E.t2a = 2.0f * E.a * t;
    E.X = sqrtf(2.0f * t * E.c / E.det);
    E.Y = sqrtf(2.0f * t * E.a / E.det);
    if (!(E.X == E.X && E.Y == E.Y) || E.X > 1e8f || E.Y > 1e8f) return E;
    E.mode = 1;
    return E;
}
// Tiles [*txlo, *txhi] (inclusive, not clamped to the image) of tile row ty that the ellipse may reach; false when none.
// (px, py) = Gaussian centre in pixel coordinates; pixel centres sit at integer coordinates.
FDGS_HD bool cull_row(const CullEllipse& E, float px, float py, int ty, int* txlo, int* txhi) {
    const float y0 = (float)(ty * FDGS_TILE) - py, y1 = y0 + (float)(FDGS_TILE - 1);
    const float eps = 0.02f + 1e-5f * (E.X + E.Y);
    const float yl = fmaxf(y0, -E.Y - eps), yh = fminf(y1, E.Y + eps);
    if (yl > yh) return false;
    // x-range of the ellipse over the band: extreme points (+-X at dy = -+b X / c) when they lie in the band, else the ends
    const float dyR = -E.b * E.X / E.c;
    const float sl = sqrtf(fmaxf(0.0f, E.t2a - E.det * yl * yl)), sh = sqrtf(fmaxf(0.0f, E.t2a - E.det * yh * yh));
    const float inv_a = 1.0f / E.a;
    float hi = fmaxf((-E.b * yl + sl) * inv_a, (-E.b * yh + sh) * inv_a);
    float lo = fminf((-E.b * yl - sl) * inv_a, (-E.b * yh - sh) * inv_a);
    if (dyR >= yl && dyR <= yh) hi = E.X;
    if (-dyR >= yl && -dyR <= yh) lo = -E.X;
    lo -= eps; hi += eps;
    if (!(lo <= hi)) { *txlo = -(1 << 20); *txhi = 1 << 20; return true; }   // NaN: keep the row
    // smallest tx with 16 tx + 15 - px >= lo  <=>  tx >= (lo + px - 15) / 16
    *txlo = (int)ceilf((lo + px - (float)(FDGS_TILE - 1)) / (float)FDGS_TILE);
    // largest tx with 16 tx - px <= hi
    *txhi = (int)floorf((hi + px) / (float)FDGS_TILE);
    return *txlo <= *txhi;
}

FDGS_HD void sh_basis(int deg, float x, float y, float z, float* b) {
    b[0] = FDGS_SH_C0;
    if (deg > 0) {
        b[1] = -FDGS_SH_C1 * y; b[2] = FDGS_SH_C1 * z; b[3] = -FDGS_SH_C1 * x;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = FDGS_SH_C2_0 * xy; b[5] = FDGS_SH_C2_1 * yz; b[6] = FDGS_SH_C2_2 * (2.0f * zz - xx - yy);
            b[7] = FDGS_SH_C2_3 * xz; b[8] = FDGS_SH_C2_4 * (xx - yy);
            if (deg > 2) {
                b[9] = FDGS_SH_C3_0 * y * (3.0f * xx - yy);
                b[10] = FDGS_SH_C3_1 * xy * z;
                b[11] = FDGS_SH_C3_2 * y * (4.0f * zz - xx - yy);
                b[12] = FDGS_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                b[13] = FDGS_SH_C3_4 * x * (4.0f * zz - xx - yy);
                b[14] = FDGS_SH_C3_5 * z * (xx - yy);
                b[15] = FDGS_SH_C3_6 * x * (xx - 3.0f * yy);
            }
        }
    }
}

// rgb = max(sum_k basis_k * sh[k] + 0.5, 0); returns clamp bitmask (bit ch set when the channel was clamped)
FDGS_HD uint32_t sh_to_rgb(int deg, const float* sh, const float* p, const float* campos, float* rgb) {
    float dx = p[0] - campos[0], dy = p[1] - campos[1], dz = p[2] - campos[2];
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    dx /= len; dy /= len; dz /= len;
    float b[16];
    sh_basis(deg, dx, dy, dz, b);
    int nc = (deg + 1) * (deg + 1);
    float r[3] = {0.f, 0.f, 0.f};
    // fixed trip count + predicate: every array index is a compile-time constant (a runtime bound puts b[] / sh[] in scratch)
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < nc) { r[0] += b[k] * sh[3 * k + 0]; r[1] += b[k] * sh[3 * k + 1]; r[2] += b[k] * sh[3 * k + 2]; }
    }
    uint32_t cl = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float v = r[ch] + 0.5f;
        if (v < 0.0f) cl |= 1u << ch;
        rgb[ch] = fmaxf(v, 0.0f);
    }
    return cl;
}

// ---------------- backward ----------------

// d basis_k / d(x,y,z) contracted with v_k = sum_ch sh[k][ch]*g[ch]  ->  ddir
FDGS_HD void sh_bwd(int deg, const float* sh, const float* p, const float* campos, uint32_t clamped, const float* dL_drgb,
                    float* dL_dsh /* [nc*3] written for k<nc */, float* dmean /* += */) {
    float ex = p[0] - campos[0], ey = p[1] - campos[1], ez = p[2] - campos[2];
    float len = sqrtf(ex * ex + ey * ey + ez * ez);
    float x = ex / len, y = ey / len, z = ez / len;
    float g[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) g[ch] = ((clamped >> ch) & 1u) ? 0.0f : dL_drgb[ch];
    float b[16];
    sh_basis(deg, x, y, z, b);
    int nc = (deg + 1) * (deg + 1);
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < nc) {
            dL_dsh[3 * k + 0] = b[k] * g[0]; dL_dsh[3 * k + 1] = b[k] * g[1]; dL_dsh[3 * k + 2] = b[k] * g[2];
            v[k] = sh[3 * k] * g[0] + sh[3 * k + 1] * g[1] + sh[3 * k + 2] * g[2];
        } else {
            v[k] = 0.f;
        }
    }
    float ddx = 0.f, ddy = 0.f, ddz = 0.f;
    if (deg > 0) {
        ddy += -FDGS_SH_C1 * v[1]; ddz += FDGS_SH_C1 * v[2]; ddx += -FDGS_SH_C1 * v[3];
        if (deg > 1) {
            ddx += FDGS_SH_C2_0 * y * v[4]; ddy += FDGS_SH_C2_0 * x * v[4];
            ddy += FDGS_SH_C2_1 * z * v[5]; ddz += FDGS_SH_C2_1 * y * v[5];
            ddx += FDGS_SH_C2_2 * (-2.f * x) * v[6]; ddy += FDGS_SH_C2_2 * (-2.f * y) * v[6]; ddz += FDGS_SH_C2_2 * (4.f * z) * v[6];
            ddx += FDGS_SH_C2_3 * z * v[7]; ddz += FDGS_SH_C2_3 * x * v[7];
            ddx += FDGS_SH_C2_4 * (2.f * x) * v[8]; ddy += FDGS_SH_C2_4 * (-2.f * y) * v[8];
            if (deg > 2) {
                float xx = x * x, yy = y * y, zz = z * z;
                ddx += FDGS_SH_C3_0 * y * 6.f * x * v[9]; ddy += FDGS_SH_C3_0 * (3.f * xx - 3.f * yy) * v[9];
                ddx += FDGS_SH_C3_1 * y * z * v[10]; ddy += FDGS_SH_C3_1 * x * z * v[10]; ddz += FDGS_SH_C3_1 * x * y * v[10];
                ddx += FDGS_SH_C3_2 * y * (-2.f * x) * v[11]; ddy += FDGS_SH_C3_2 * (4.f * zz - xx - 3.f * yy) * v[11];
                ddz += FDGS_SH_C3_2 * y * 8.f * z * v[11];
                ddx += FDGS_SH_C3_3 * z * (-6.f * x) * v[12]; ddy += FDGS_SH_C3_3 * z * (-6.f * y) * v[12];
                ddz += FDGS_SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy) * v[12];
                ddx += FDGS_SH_C3_4 * (4.f * zz - 3.f * xx - yy) * v[13]; ddy += FDGS_SH_C3_4 * x * (-2.f * y) * v[13];
                ddz += FDGS_SH_C3_4 * x * 8.f * z * v[13];
                ddx += FDGS_SH_C3_5 * z * 2.f * x * v[14]; ddy += FDGS_SH_C3_5 * z * (-2.f * y) * v[14];
                ddz += FDGS_SH_C3_5 * (xx - yy) * v[14];
                ddx += FDGS_SH_C3_6 * (3.f * xx - 3.f * yy) * v[15]; ddy += FDGS_SH_C3_6 * x * (-6.f * y) * v[15];
            }
        }
    }
    float dot = x * ddx + y * ddy + z * ddz;
    dmean[0] += (ddx - x * dot) / len; dmean[1] += (ddy - y * dot) / len; dmean[2] += (ddz - z * dot) / len;
}

// conic / depth / mean2D gradients -> dmean (+=), dcov6 (written)
FDGS_HD void project_bwd(const CamConst& c, const float* p, const float* c6, const float* dconic /*xx, xy(half), yy*/,
                         float ddepth, float g_ndc_x, float g_ndc_y, float* dmean, float* dcov6) {
    float pv[3];
    view_xform(c.view, p, pv);
    float Mx[6], tc[3], abc[3], MS[6];
    bool xc, yc;
    ewa_M(c, pv, Mx, tc, &xc, &yc);
    cov2d_from(Mx, c6, abc, MS);
    float a = abc[0], b = abc[1], cc = abc[2];
    float denom = a * cc - b * b;
    float d2i = 1.0f / (denom * denom + FDGS_DENOM_EPS);
    float kx = dconic[0], ky = dconic[1], kz = dconic[2];
    float da = 0.f, db = 0.f, dc = 0.f;
    bool live = (d2i != 0.0f);
    if (live) {
        da = d2i * (-cc * cc * kx + 2.0f * b * cc * ky + (denom - a * cc) * kz);
        dc = d2i * (-a * a * kz + 2.0f * a * b * ky + (denom - a * cc) * kx);
        db = d2i * 2.0f * (b * cc * kx - (denom + 2.0f * b * b) * ky + a * b * kz);
    }
    // G2 = [[da, db/2],[db/2, dc]];  GM = G2*M (2x3);  dSigma_full = M^T GM;  dM = 2*GM*Sigma
    float hb = 0.5f * db;
    float GM[6];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        GM[j] = da * Mx[j] + hb * Mx[3 + j];
        GM[3 + j] = hb * Mx[j] + dc * Mx[3 + j];
    }
    dcov6[0] = Mx[0] * GM[0] + Mx[3] * GM[3];
    dcov6[1] = 2.0f * (Mx[0] * GM[1] + Mx[3] * GM[4]);
    dcov6[2] = 2.0f * (Mx[0] * GM[2] + Mx[3] * GM[5]);
    dcov6[3] = Mx[1] * GM[1] + Mx[4] * GM[4];
    dcov6[4] = 2.0f * (Mx[1] * GM[2] + Mx[4] * GM[5]);
    dcov6[5] = Mx[2] * GM[2] + Mx[5] * GM[5];
    const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    float dM[6];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) dM[3 * r + j] = 2.0f * (GM[3 * r] * S[j] + GM[3 * r + 1] * S[3 + j] + GM[3 * r + 2] * S[6 + j]);
    const float* v = c.view;
    float dJ00 = dM[0] * v[0] + dM[1] * v[4] + dM[2] * v[8];
    float dJ02 = dM[0] * v[2] + dM[1] * v[6] + dM[2] * v[10];
    float dJ11 = dM[3] * v[1] + dM[4] * v[5] + dM[5] * v[9];
    float dJ12 = dM[3] * v[2] + dM[4] * v[6] + dM[5] * v[10];
    float tz = 1.0f / tc[2], tz2 = tz * tz, tz3 = tz2 * tz;
    float dtx = (xc ? 0.0f : 1.0f) * -c.focal_x * tz2 * dJ02;
    float dty = (yc ? 0.0f : 1.0f) * -c.focal_y * tz2 * dJ12;
    float dtz = -c.focal_x * tz2 * dJ00 - c.focal_y * tz2 * dJ11 + (2.0f * c.focal_x * tc[0]) * tz3 * dJ02 +
                (2.0f * c.focal_y * tc[1]) * tz3 * dJ12;
    dtz += ddepth;  // depth_i = p_view.z
    dmean[0] += v[0] * dtx + v[1] * dty + v[2] * dtz;
    dmean[1] += v[4] * dtx + v[5] * dty + v[6] * dtz;
    dmean[2] += v[8] * dtx + v[9] * dty + v[10] * dtz;
    // mean2D (NDC) -> mean3D through the full projection
    const float* pr = c.proj;
    float mh0 = pr[0] * p[0] + pr[4] * p[1] + pr[8] * p[2] + pr[12];
    float mh1 = pr[1] * p[0] + pr[5] * p[1] + pr[9] * p[2] + pr[13];
    float mw = 1.0f / (proj_w(pr, p) + FDGS_W_EPS);
    float mul1 = mh0 * mw * mw, mul2 = mh1 * mw * mw;
    dmean[0] += (pr[0] * mw - pr[3] * mul1) * g_ndc_x + (pr[1] * mw - pr[3] * mul2) * g_ndc_y;
    dmean[1] += (pr[4] * mw - pr[7] * mul1) * g_ndc_x + (pr[5] * mw - pr[7] * mul2) * g_ndc_y;
    dmean[2] += (pr[8] * mw - pr[11] * mul1) * g_ndc_x + (pr[9] * mw - pr[11] * mul2) * g_ndc_y;
}

// dL/dSigma (6 unique entries, off-diagonals carry both symmetric halves) -> dL/dscale, dL/dq (q as given)
FDGS_HD void cov3d_bwd(const float* s, float mod, const float* q, const float* dcov6, float* dscale, float* dq) {
    float Rm[9];
    quat_to_R(q, Rm);
    const float Gs[9] = {dcov6[0], 0.5f * dcov6[1], 0.5f * dcov6[2], 0.5f * dcov6[1], dcov6[3],
                         0.5f * dcov6[4], 0.5f * dcov6[2], 0.5f * dcov6[4], dcov6[5]};
    float L[9], dL[9], dR[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 3; k++) L[3 * r + k] = Rm[3 * r + k] * (mod * s[k]);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 3; k++) dL[3 * r + k] = 2.0f * (Gs[3 * r] * L[k] + Gs[3 * r + 1] * L[3 + k] + Gs[3 * r + 2] * L[6 + k]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            acc += dL[3 * r + k] * Rm[3 * r + k];
            dR[3 * r + k] = dL[3 * r + k] * (mod * s[k]);
        }
        dscale[k] = mod * acc;
    }
    float r = q[0], x = q[1], y = q[2], z = q[3];
    dq[0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
    dq[1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
    dq[2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
    dq[3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
}

}  // namespace fdgs
