/*
 * oracle/raster_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the tile-based differentiable Gaussian rasterizer the reference
 * calls through `diff_gaussian_rasterization` (call sites: gaussian_renderer/__init__.py:14,38-58,
 * 120-128 of the reference).  The rasterizer itself is an UN-VENDORED third-party submodule
 * (github.com/ingra14m/depth-diff-gaussian-rasterization, fork of
 * graphdeco-inria/diff-gaussian-rasterization; pinned commit not recoverable from the mount), so
 * this file restates its published algorithm and PARITY IS UNPINNED by any
 * reference golden vector: trust is earned by (i) the float64 autograd restatement in
 * oracle/raster_torch.py agreeing with the analytic backward here, (ii) closed-form single-Gaussian
 * cases, (iii) the in-tree python SH / cov3D formulas (utils/sh_utils.py:57-112,
 * utils/general_utils.py:84-116 of the reference).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Every constant of the arithmetic is a named macro below so that it is greppable.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC [-DORACLE_DOUBLE] raster_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORACLE_DOUBLE
typedef double REAL;
#define R_EXP exp
#define R_SQRT sqrt
#define R_CEIL ceil
#define R_FMAX fmax
#define R_FMIN fmin
#else
typedef float REAL;
#define R_EXP expf
#define R_SQRT sqrtf
#define R_CEIL ceilf
#define R_FMAX fmaxf
#define R_FMIN fminf
#endif

/* ---- named assumptions -------------------------------------------- */
#define TILE 16                      /* BLOCK_X = BLOCK_Y */
#define NEAR_CULL ((REAL)0.2)        /* p_view.z <= 0.2 -> culled */
#define W_EPS ((REAL)0.0000001)      /* 1/(p_hom.w + 1e-7) */
#define FOV_CLAMP ((REAL)1.3)        /* clamp t.x/t.z to +-1.3*tanfov */
#define DILATION ((REAL)0.3)         /* cov2D diagonal += 0.3 */
#define LAMBDA_FLOOR ((REAL)0.1)     /* sqrt(max(0.1, mid^2-det)) */
#define RADIUS_SIGMA ((REAL)3.0)     /* radius = ceil(3*sqrt(lambda_max)) */
#define ALPHA_MAX ((REAL)0.99)
#define ALPHA_MIN ((REAL)(1.0 / 255.0))
#define T_STOP ((REAL)0.0001)
#define DENOM_EPS ((REAL)0.0000001)  /* 1/(denom^2 + 1e-7) in the conic backward */

static const REAL SH_C0 = (REAL)0.28209479177387814;
static const REAL SH_C1 = (REAL)0.4886025119029199;
static const REAL SH_C2[5] = {(REAL)1.0925484305920792, (REAL)-1.0925484305920792, (REAL)0.31539156525252005,
                              (REAL)-1.0925484305920792, (REAL)0.5462742152960396};
static const REAL SH_C3[7] = {(REAL)-0.5900435899266435, (REAL)2.890611442640554, (REAL)-0.4570457994644658,
                              (REAL)0.3731763325901154, (REAL)-0.4570457994644658, (REAL)1.445305721320277,
                              (REAL)-0.5900435899266435};

typedef struct {
    int P, D, M, W, H, gx, gy, R;
    REAL tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    REAL bg[3], view[16], proj[16], campos[3];
    /* borrowed input pointers (must outlive the handle) */
    const REAL *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    /* per-Gaussian state */
    REAL *depth, *xy, *conic_opacity, *rgb, *cov3D;
    int *radii, *tiles_touched;
    uint8_t *clamped;
    uint32_t *rect; /* xmin ymin xmax ymax */
    /* binning */
    uint32_t *pair_tile, *pair_gid;
    uint32_t *ranges; /* 2 per tile */
    /* image state */
    REAL *final_T;
    uint32_t *n_contrib;
} Oracle;

static inline void xf4x3(const REAL *m, const REAL *p, REAL *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xf4x4(const REAL *m, const REAL *p, REAL *o) {
    xf4x3(m, p, o);
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}
static inline REAL ndc2pix(REAL v, int S) { return ((v + (REAL)1.0) * (REAL)S - (REAL)1.0) * (REAL)0.5; }

/* Rotation from the quaternion AS GIVEN (r,x,y,z), no normalisation; same entries as the reference's
 * python path utils/general_utils.py:96-104. Row-major R[3*i+j]. */
static inline void quat_to_R(const REAL *q, REAL *R) {
    REAL r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - r * z);     R[2] = 2 * (x * z + r * y);
    R[3] = 2 * (x * y + r * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - r * x);
    R[6] = 2 * (x * z - r * y);     R[7] = 2 * (y * z + r * x);     R[8] = 1 - 2 * (x * x + y * y);
}

/* Sigma = (R S)(R S)^T, S = diag(mod*s); stored (00,01,02,11,12,22). */
static void cov3d_from_scale_rot(const REAL *s, REAL mod, const REAL *q, REAL *c6) {
    REAL R[9], L[9];
    quat_to_R(q, R);
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) L[3 * i + k] = R[3 * i + k] * (mod * s[k]);
    REAL S[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            REAL a = 0;
            for (int k = 0; k < 3; k++) a += L[3 * i + k] * L[3 * j + k];
            S[3 * i + j] = a;
        }
    c6[0] = S[0]; c6[1] = S[1]; c6[2] = S[2]; c6[3] = S[4]; c6[4] = S[5]; c6[5] = S[8];
}

/* M = J * Rv (2x3), with the +-1.3 tanfov clamp on t.x/t.z, t.y/t.z. Returns clamp flags. */
static void ewa_M(const Oracle *o, const REAL *t_in, REAL *Mx, REAL *t_out, int *xclamp, int *yclamp) {
    REAL t[3] = {t_in[0], t_in[1], t_in[2]};
    REAL limx = FOV_CLAMP * o->tanfovx, limy = FOV_CLAMP * o->tanfovy;
    REAL txtz = t[0] / t[2], tytz = t[1] / t[2];
    *xclamp = (txtz < -limx || txtz > limx);
    *yclamp = (tytz < -limy || tytz > limy);
    t[0] = R_FMIN(limx, R_FMAX(-limx, txtz)) * t[2];
    t[1] = R_FMIN(limy, R_FMAX(-limy, tytz)) * t[2];
    REAL J00 = o->focal_x / t[2], J02 = -(o->focal_x * t[0]) / (t[2] * t[2]);
    REAL J11 = o->focal_y / t[2], J12 = -(o->focal_y * t[1]) / (t[2] * t[2]);
    const REAL *v = o->view; /* Rv[i][j] = v[4*j+i] */
    for (int j = 0; j < 3; j++) {
        Mx[j] = J00 * v[4 * j + 0] + J02 * v[4 * j + 2];
        Mx[3 + j] = J11 * v[4 * j + 1] + J12 * v[4 * j + 2];
    }
    t_out[0] = t[0]; t_out[1] = t[1]; t_out[2] = t[2];
}

static void sym6_to_full(const REAL *c6, REAL *S) {
    S[0] = c6[0]; S[1] = c6[1]; S[2] = c6[2];
    S[3] = c6[1]; S[4] = c6[3]; S[5] = c6[4];
    S[6] = c6[2]; S[7] = c6[4]; S[8] = c6[5];
}

/* SH basis (degree <= 3) at unit direction d; utils/sh_utils.py:74-100 of the reference. */
static void sh_basis(int deg, const REAL *d, REAL *b) {
    REAL x = d[0], y = d[1], z = d[2];
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
        if (deg > 1) {
            REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * ((REAL)2.0 * zz - xx - yy);
            b[7] = SH_C2[3] * xz; b[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = SH_C3[0] * y * ((REAL)3.0 * xx - yy);
                b[10] = SH_C3[1] * xy * z;
                b[11] = SH_C3[2] * y * ((REAL)4.0 * zz - xx - yy);
                b[12] = SH_C3[3] * z * ((REAL)2.0 * zz - (REAL)3.0 * xx - (REAL)3.0 * yy);
                b[13] = SH_C3[4] * x * ((REAL)4.0 * zz - xx - yy);
                b[14] = SH_C3[5] * z * (xx - yy);
                b[15] = SH_C3[6] * x * (xx - (REAL)3.0 * yy);
            }
        }
    }
}
/* d basis_k / d (x,y,z) */
static void sh_basis_grad(int deg, const REAL *d, REAL *gx, REAL *gy, REAL *gz) {
    REAL x = d[0], y = d[1], z = d[2];
    for (int k = 0; k < 16; k++) gx[k] = gy[k] = gz[k] = 0;
    if (deg > 0) {
        gy[1] = -SH_C1; gz[2] = SH_C1; gx[3] = -SH_C1;
        if (deg > 1) {
            gx[4] = SH_C2[0] * y; gy[4] = SH_C2[0] * x;
            gy[5] = SH_C2[1] * z; gz[5] = SH_C2[1] * y;
            gx[6] = SH_C2[2] * (-2 * x); gy[6] = SH_C2[2] * (-2 * y); gz[6] = SH_C2[2] * (4 * z);
            gx[7] = SH_C2[3] * z; gz[7] = SH_C2[3] * x;
            gx[8] = SH_C2[4] * (2 * x); gy[8] = SH_C2[4] * (-2 * y);
            if (deg > 2) {
                REAL xx = x * x, yy = y * y, zz = z * z;
                gx[9] = SH_C3[0] * y * 6 * x; gy[9] = SH_C3[0] * (3 * xx - 3 * yy);
                gx[10] = SH_C3[1] * y * z; gy[10] = SH_C3[1] * x * z; gz[10] = SH_C3[1] * x * y;
                gx[11] = SH_C3[2] * y * (-2 * x); gy[11] = SH_C3[2] * (4 * zz - xx - 3 * yy); gz[11] = SH_C3[2] * y * 8 * z;
                gx[12] = SH_C3[3] * z * (-6 * x); gy[12] = SH_C3[3] * z * (-6 * y); gz[12] = SH_C3[3] * (6 * zz - 3 * xx - 3 * yy);
                gx[13] = SH_C3[4] * (4 * zz - 3 * xx - yy); gy[13] = SH_C3[4] * x * (-2 * y); gz[13] = SH_C3[4] * x * 8 * z;
                gx[14] = SH_C3[5] * z * 2 * x; gy[14] = SH_C3[5] * z * (-2 * y); gz[14] = SH_C3[5] * (xx - yy);
                gx[15] = SH_C3[6] * (3 * xx - 3 * yy); gy[15] = SH_C3[6] * x * (-6 * y);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* stable merge sort of pair indices by (tile, depth, original order) */
typedef struct { uint32_t tile; REAL depth; uint32_t gid; } PairKey;
static void *xcalloc(size_t n, size_t s) { return calloc(n ? n : 1, s); }

void oracle_raster_free(void *h) {
    Oracle *o = (Oracle *)h;
    if (!o) return;
    free(o->depth); free(o->xy); free(o->conic_opacity); free(o->rgb); free(o->cov3D);
    free(o->radii); free(o->tiles_touched); free(o->clamped); free(o->rect);
    free(o->pair_tile); free(o->pair_gid); free(o->ranges); free(o->final_T); free(o->n_contrib);
    free(o);
}

/*
 * Forward. All arrays are REAL (float, or double with -DORACLE_DOUBLE) unless noted.
 * out_color [3,H,W], out_depth [H,W], out_radii int[P]. Returns an opaque handle for backward.
 */
void *oracle_raster_forward(int P, int D, int M, const REAL *bg, int W, int H, const REAL *means3D, const REAL *shs,
                            const REAL *colors_precomp, const REAL *opacities, const REAL *scales, REAL scale_modifier,
                            const REAL *rotations, const REAL *cov3D_precomp, const REAL *viewmatrix,
                            const REAL *projmatrix, const REAL *campos, REAL tanfovx, REAL tanfovy, int prefiltered,
                            REAL *out_color, REAL *out_depth, int *out_radii) {
    (void)prefiltered;
    Oracle *o = (Oracle *)calloc(1, sizeof(Oracle));
    o->P = P; o->D = D; o->M = M; o->W = W; o->H = H;
    o->gx = (W + TILE - 1) / TILE; o->gy = (H + TILE - 1) / TILE;
    o->tanfovx = tanfovx; o->tanfovy = tanfovy; o->scale_modifier = scale_modifier;
    o->focal_x = (REAL)W / ((REAL)2.0 * tanfovx); o->focal_y = (REAL)H / ((REAL)2.0 * tanfovy);
    memcpy(o->bg, bg, 3 * sizeof(REAL)); memcpy(o->view, viewmatrix, 16 * sizeof(REAL));
    memcpy(o->proj, projmatrix, 16 * sizeof(REAL)); memcpy(o->campos, campos, 3 * sizeof(REAL));
    o->means3D = means3D; o->shs = shs; o->colors_precomp = colors_precomp; o->opacities = opacities;
    o->scales = scales; o->rotations = rotations; o->cov3D_precomp = cov3D_precomp;
    o->depth = xcalloc(P, sizeof(REAL)); o->xy = xcalloc(2 * (size_t)P, sizeof(REAL));
    o->conic_opacity = xcalloc(4 * (size_t)P, sizeof(REAL)); o->rgb = xcalloc(3 * (size_t)P, sizeof(REAL));
    o->cov3D = xcalloc(6 * (size_t)P, sizeof(REAL)); o->radii = xcalloc(P, sizeof(int));
    o->tiles_touched = xcalloc(P, sizeof(int)); o->clamped = xcalloc(3 * (size_t)P, 1);
    o->rect = xcalloc(4 * (size_t)P, sizeof(uint32_t));

    /* ---- B.1 preprocess ---- */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        const REAL *p = means3D + 3 * (size_t)i;
        REAL pv[3], ph[4];
        o->radii[i] = 0; o->tiles_touched[i] = 0;
        xf4x3(o->view, p, pv);
        if (pv[2] <= NEAR_CULL) continue;
        xf4x4(o->proj, p, ph);
        REAL pw = (REAL)1.0 / (ph[3] + W_EPS);
        REAL pp[3] = {ph[0] * pw, ph[1] * pw, ph[2] * pw};
        REAL *c6 = o->cov3D + 6 * (size_t)i;
        if (cov3D_precomp) memcpy(c6, cov3D_precomp + 6 * (size_t)i, 6 * sizeof(REAL));
        else cov3d_from_scale_rot(scales + 3 * (size_t)i, scale_modifier, rotations + 4 * (size_t)i, c6);
        REAL Mx[6], tc[3]; int xc, yc;
        ewa_M(o, pv, Mx, tc, &xc, &yc);
        REAL S[9]; sym6_to_full(c6, S);
        REAL MS[6];
        for (int r = 0; r < 2; r++)
            for (int j = 0; j < 3; j++) MS[3 * r + j] = Mx[3 * r] * S[j] + Mx[3 * r + 1] * S[3 + j] + Mx[3 * r + 2] * S[6 + j];
        REAL a = MS[0] * Mx[0] + MS[1] * Mx[1] + MS[2] * Mx[2] + DILATION;
        REAL b = MS[0] * Mx[3] + MS[1] * Mx[4] + MS[2] * Mx[5];
        REAL c = MS[3] * Mx[3] + MS[4] * Mx[4] + MS[5] * Mx[5] + DILATION;
        REAL det = a * c - b * b;
        if (det == (REAL)0.0) continue;
        REAL det_inv = (REAL)1.0 / det;
        REAL con[3] = {c * det_inv, -b * det_inv, a * det_inv};
        REAL mid = (REAL)0.5 * (a + c);
        REAL sq = R_SQRT(R_FMAX(LAMBDA_FLOOR, mid * mid - det));
        REAL l1 = mid + sq, l2 = mid - sq;
        REAL my_radius = R_CEIL(RADIUS_SIGMA * R_SQRT(R_FMAX(l1, l2)));
        REAL px = ndc2pix(pp[0], W), py = ndc2pix(pp[1], H);
        int rx0 = (int)((px - my_radius) / (REAL)TILE), ry0 = (int)((py - my_radius) / (REAL)TILE);
        int rx1 = (int)((px + my_radius + (REAL)(TILE - 1)) / (REAL)TILE), ry1 = (int)((py + my_radius + (REAL)(TILE - 1)) / (REAL)TILE);
        rx0 = rx0 < 0 ? 0 : (rx0 > o->gx ? o->gx : rx0); ry0 = ry0 < 0 ? 0 : (ry0 > o->gy ? o->gy : ry0);
        rx1 = rx1 < 0 ? 0 : (rx1 > o->gx ? o->gx : rx1); ry1 = ry1 < 0 ? 0 : (ry1 > o->gy ? o->gy : ry1);
        if ((rx1 - rx0) * (ry1 - ry0) == 0) continue;
        REAL *rgb = o->rgb + 3 * (size_t)i;
        if (colors_precomp) {
            for (int ch = 0; ch < 3; ch++) rgb[ch] = colors_precomp[3 * (size_t)i + ch];
        } else {
            REAL dir[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
            REAL len = R_SQRT(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            dir[0] /= len; dir[1] /= len; dir[2] /= len;
            REAL bs[16]; sh_basis(D, dir, bs);
            int nc = (D + 1) * (D + 1);
            const REAL *sh = shs + (size_t)i * M * 3;
            for (int ch = 0; ch < 3; ch++) {
                REAL r = 0;
                for (int k = 0; k < nc; k++) r += bs[k] * sh[3 * k + ch];
                r += (REAL)0.5;
                o->clamped[3 * (size_t)i + ch] = (r < 0);
                rgb[ch] = R_FMAX(r, (REAL)0.0);
            }
        }
        o->depth[i] = pv[2]; o->radii[i] = (int)my_radius;
        o->xy[2 * (size_t)i] = px; o->xy[2 * (size_t)i + 1] = py;
        REAL *co = o->conic_opacity + 4 * (size_t)i;
        co[0] = con[0]; co[1] = con[1]; co[2] = con[2]; co[3] = opacities[i];
        o->tiles_touched[i] = (rx1 - rx0) * (ry1 - ry0);
        uint32_t *rc = o->rect + 4 * (size_t)i; rc[0] = rx0; rc[1] = ry0; rc[2] = rx1; rc[3] = ry1;
    }
    for (int i = 0; i < P; i++) out_radii[i] = o->radii[i];

    /* ---- B.2 binning ---- */
    size_t R = 0;
    for (int i = 0; i < P; i++) R += (size_t)o->tiles_touched[i];
    o->R = (int)R;
    PairKey *pk = (PairKey *)xcalloc(R, sizeof(PairKey)), *tmp = (PairKey *)xcalloc(R, sizeof(PairKey));
    size_t off = 0;
    for (int i = 0; i < P; i++) {
        if (o->radii[i] <= 0) continue;
        uint32_t *rc = o->rect + 4 * (size_t)i;
        for (uint32_t y = rc[1]; y < rc[3]; y++)
            for (uint32_t x = rc[0]; x < rc[2]; x++) {
                pk[off].tile = y * (uint32_t)o->gx + x;
                pk[off].depth = o->depth[i]; /* in float builds the reference orders by the raw fp32 bits: same order for z>0 */
                pk[off].gid = (uint32_t)i; off++;
            }
    }
    /* bottom-up stable merge sort; track which buffer holds the result */
    {
        PairKey *a = pk, *b = tmp;
        for (size_t w = 1; w < R; w *= 2) {
            for (size_t lo = 0; lo < R; lo += 2 * w) {
                size_t mid = lo + w < R ? lo + w : R, hi = lo + 2 * w < R ? lo + 2 * w : R;
                size_t i = lo, j = mid, k = lo;
                while (i < mid && j < hi) {
                    int take_right = (a[j].tile < a[i].tile) || (a[j].tile == a[i].tile && a[j].depth < a[i].depth);
                    b[k++] = take_right ? a[j++] : a[i++];
                }
                while (i < mid) b[k++] = a[i++];
                while (j < hi) b[k++] = a[j++];
            }
            PairKey *s = a; a = b; b = s;
        }
        o->pair_tile = xcalloc(R, sizeof(uint32_t)); o->pair_gid = xcalloc(R, sizeof(uint32_t));
        for (size_t i = 0; i < R; i++) { o->pair_tile[i] = a[i].tile; o->pair_gid[i] = a[i].gid; }
    }
    free(pk); free(tmp);
    int ntiles = o->gx * o->gy;
    o->ranges = xcalloc(2 * (size_t)ntiles, sizeof(uint32_t));
    for (size_t i = 0; i < R; i++) {
        uint32_t t = o->pair_tile[i];
        if (i == 0 || o->pair_tile[i - 1] != t) o->ranges[2 * t] = (uint32_t)i;
        if (i == R - 1 || o->pair_tile[i + 1] != t) o->ranges[2 * t + 1] = (uint32_t)(i + 1);
    }

    /* ---- B.3 render forward ---- */
    o->final_T = xcalloc((size_t)W * H, sizeof(REAL)); o->n_contrib = xcalloc((size_t)W * H, sizeof(uint32_t));
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < ntiles; tile++) {
        int tx = tile % o->gx, ty = tile / o->gx;
        uint32_t r0 = o->ranges[2 * tile], r1 = o->ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                int x = tx * TILE + lx, y = ty * TILE + ly;
                if (x >= W || y >= H) continue;
                REAL T = 1, C[3] = {0, 0, 0}, Dp = 0;
                uint32_t contributor = 0, last = 0;
                for (uint32_t q = r0; q < r1; q++) {
                    uint32_t g = o->pair_gid[q];
                    contributor++;
                    REAL dx = o->xy[2 * (size_t)g] - (REAL)x, dy = o->xy[2 * (size_t)g + 1] - (REAL)y;
                    const REAL *co = o->conic_opacity + 4 * (size_t)g;
                    REAL power = (REAL)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > (REAL)0.0) continue;
                    REAL alpha = R_FMIN(ALPHA_MAX, co[3] * R_EXP(power));
                    if (alpha < ALPHA_MIN) continue;
                    REAL test_T = T * ((REAL)1.0 - alpha);
                    if (test_T < T_STOP) break;
                    REAL w = alpha * T;
                    for (int ch = 0; ch < 3; ch++) C[ch] += o->rgb[3 * (size_t)g + ch] * w;
                    Dp += o->depth[g] * w;
                    T = test_T; last = contributor;
                }
                size_t pix = (size_t)y * W + x;
                o->final_T[pix] = T; o->n_contrib[pix] = last;
                for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix] = C[ch] + T * bg[ch];
                out_depth[pix] = Dp;
            }
    }
    return o;
}

/* accessors for stage-wise tests */
int oracle_raster_num_rendered(void *h) { return ((Oracle *)h)->R; }
const void *oracle_raster_field(void *h, int which) {
    Oracle *o = (Oracle *)h;
    switch (which) {
        case 0: return o->depth; case 1: return o->xy; case 2: return o->conic_opacity; case 3: return o->rgb;
        case 4: return o->cov3D; case 5: return o->radii; case 6: return o->tiles_touched; case 7: return o->clamped;
        case 8: return o->pair_tile; case 9: return o->pair_gid; case 10: return o->ranges; case 11: return o->final_T;
        case 12: return o->n_contrib; case 13: return o->rect;
    }
    return 0;
}

/*
 * Backward (B.4 + B.5). dL_dcolor [3,H,W]; dL_ddepth [H,W] or NULL.
 * Outputs (all zero-filled here): dL_dmeans2D [P,3] (x,y in NDC units, z=0), dL_dmeans3D [P,3], dL_dsh [P,M,3],
 * dL_dcolors [P,3] (gradient w.r.t. the per-Gaussian rgb actually blended: equals dL/dcolors_precomp when that
 * input is used), dL_dopacity [P], dL_dscales [P,3], dL_drot [P,4], dL_dcov3D [P,6].
 */
void oracle_raster_backward(void *h, const REAL *dL_dcolor, const REAL *dL_ddepth_img, REAL *dL_dmeans2D, REAL *dL_dmeans3D,
                            REAL *dL_dsh, REAL *dL_dcolors, REAL *dL_dopacity, REAL *dL_dscales, REAL *dL_drot,
                            REAL *dL_dcov3D) {
    Oracle *o = (Oracle *)h;
    int P = o->P, W = o->W, H = o->H, M = o->M;
    size_t R = (size_t)o->R;
    memset(dL_dmeans2D, 0, 3 * (size_t)P * sizeof(REAL)); memset(dL_dmeans3D, 0, 3 * (size_t)P * sizeof(REAL));
    memset(dL_dsh, 0, (size_t)P * M * 3 * sizeof(REAL)); memset(dL_dcolors, 0, 3 * (size_t)P * sizeof(REAL));
    memset(dL_dopacity, 0, (size_t)P * sizeof(REAL)); memset(dL_dscales, 0, 3 * (size_t)P * sizeof(REAL));
    memset(dL_drot, 0, 4 * (size_t)P * sizeof(REAL)); memset(dL_dcov3D, 0, 6 * (size_t)P * sizeof(REAL));
    /* per-pair records: mean2D(2, pixel units) conic(3) opacity(1) color(3) depth(1) */
    enum { NREC = 10 };
    REAL *rec = (REAL *)xcalloc(R * NREC, sizeof(REAL));
    int ntiles = o->gx * o->gy;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < ntiles; tile++) {
        int tx = tile % o->gx, ty = tile / o->gx;
        uint32_t r0 = o->ranges[2 * tile], r1 = o->ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                int x = tx * TILE + lx, y = ty * TILE + ly;
                if (x >= W || y >= H) continue;
                size_t pix = (size_t)y * W + x;
                REAL T_final = o->final_T[pix], T = T_final;
                uint32_t last = o->n_contrib[pix];
                REAL dpix[3] = {dL_dcolor[pix], dL_dcolor[(size_t)H * W + pix], dL_dcolor[2 * (size_t)H * W + pix]};
                REAL ddep = dL_ddepth_img ? dL_ddepth_img[pix] : (REAL)0.0;
                REAL accum[3] = {0, 0, 0}, last_c[3] = {0, 0, 0}, accum_d = 0, last_d = 0, last_alpha = 0;
                REAL bgdot = o->bg[0] * dpix[0] + o->bg[1] * dpix[1] + o->bg[2] * dpix[2];
                for (uint32_t k = last; k-- > 0;) {
                    size_t q = (size_t)r0 + k;
                    (void)r1;
                    uint32_t g = o->pair_gid[q];
                    REAL dx = o->xy[2 * (size_t)g] - (REAL)x, dy = o->xy[2 * (size_t)g + 1] - (REAL)y;
                    const REAL *co = o->conic_opacity + 4 * (size_t)g;
                    REAL power = (REAL)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > (REAL)0.0) continue;
                    REAL G = R_EXP(power);
                    REAL alpha = R_FMIN(ALPHA_MAX, co[3] * G);
                    if (alpha < ALPHA_MIN) continue;
                    T = T / ((REAL)1.0 - alpha);
                    REAL w = alpha * T;
                    REAL *rc = rec + q * NREC;
                    REAL dL_dalpha = 0;
                    for (int ch = 0; ch < 3; ch++) {
                        REAL c = o->rgb[3 * (size_t)g + ch];
                        accum[ch] = last_alpha * last_c[ch] + ((REAL)1.0 - last_alpha) * accum[ch];
                        last_c[ch] = c;
                        dL_dalpha += (c - accum[ch]) * dpix[ch];
                        rc[6 + ch] += w * dpix[ch];
                    }
                    {
                        REAL cd = o->depth[g];
                        accum_d = last_alpha * last_d + ((REAL)1.0 - last_alpha) * accum_d;
                        last_d = cd;
                        dL_dalpha += (cd - accum_d) * ddep;
                        rc[9] += w * ddep;
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / ((REAL)1.0 - alpha)) * bgdot;
                    REAL dL_dG = co[3] * dL_dalpha;
                    REAL gdx = G * dx, gdy = G * dy;
                    REAL dG_ddelx = -gdx * co[0] - gdy * co[1];
                    REAL dG_ddely = -gdy * co[2] - gdx * co[1];
                    rc[0] += dL_dG * dG_ddelx;
                    rc[1] += dL_dG * dG_ddely;
                    rc[2] += (REAL)-0.5 * gdx * dx * dL_dG;
                    rc[3] += (REAL)-0.5 * gdx * dy * dL_dG;
                    rc[4] += (REAL)-0.5 * gdy * dy * dL_dG;
                    rc[5] += G * dL_dalpha;
                }
            }
    }
    /* deterministic reduction of pair records into per-Gaussian gradients (sorted-list order) */
    REAL *g_mean2D = (REAL *)xcalloc(2 * (size_t)P, sizeof(REAL)), *g_conic = (REAL *)xcalloc(3 * (size_t)P, sizeof(REAL));
    REAL *g_depth = (REAL *)xcalloc(P, sizeof(REAL));
    for (size_t q = 0; q < R; q++) {
        uint32_t g = o->pair_gid[q];
        const REAL *rc = rec + q * NREC;
        g_mean2D[2 * (size_t)g] += rc[0]; g_mean2D[2 * (size_t)g + 1] += rc[1];
        g_conic[3 * (size_t)g] += rc[2]; g_conic[3 * (size_t)g + 1] += rc[3]; g_conic[3 * (size_t)g + 2] += rc[4];
        dL_dopacity[g] += rc[5];
        for (int ch = 0; ch < 3; ch++) dL_dcolors[3 * (size_t)g + ch] += rc[6 + ch];
        g_depth[g] += rc[9];
    }
    free(rec);

    /* ---- B.5 per-Gaussian backward ---- */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (!(o->radii[i] > 0)) continue;
        const REAL *p = o->means3D + 3 * (size_t)i;
        /* mean2D in NDC units (what the reference hands back through `means2D.grad`) */
        REAL gx = g_mean2D[2 * (size_t)i] * (REAL)0.5 * (REAL)W, gy = g_mean2D[2 * (size_t)i + 1] * (REAL)0.5 * (REAL)H;
        dL_dmeans2D[3 * (size_t)i] = gx; dL_dmeans2D[3 * (size_t)i + 1] = gy;
        REAL dmean[3] = {0, 0, 0};
        /* (i) conic -> cov2D -> cov3D, t */
        REAL pv[3]; xf4x3(o->view, p, pv);
        REAL Mx[6], tc[3]; int xc, yc;
        ewa_M(o, pv, Mx, tc, &xc, &yc);
        const REAL *c6 = o->cov3D + 6 * (size_t)i;
        REAL S[9]; sym6_to_full(c6, S);
        REAL MS[6];
        for (int r = 0; r < 2; r++)
            for (int j = 0; j < 3; j++) MS[3 * r + j] = Mx[3 * r] * S[j] + Mx[3 * r + 1] * S[3 + j] + Mx[3 * r + 2] * S[6 + j];
        REAL a = MS[0] * Mx[0] + MS[1] * Mx[1] + MS[2] * Mx[2] + DILATION;
        REAL b = MS[0] * Mx[3] + MS[1] * Mx[4] + MS[2] * Mx[5];
        REAL c = MS[3] * Mx[3] + MS[4] * Mx[4] + MS[5] * Mx[5] + DILATION;
        REAL denom = a * c - b * b;
        REAL d2i = (REAL)1.0 / (denom * denom + DENOM_EPS);
        REAL kx = g_conic[3 * (size_t)i], ky = g_conic[3 * (size_t)i + 1], kz = g_conic[3 * (size_t)i + 2];
        REAL dL_da = 0, dL_db = 0, dL_dc = 0;
        if (d2i != (REAL)0.0) {
            dL_da = d2i * (-c * c * kx + (REAL)2.0 * b * c * ky + (denom - a * c) * kz);
            dL_dc = d2i * (-a * a * kz + (REAL)2.0 * a * b * ky + (denom - a * c) * kx);
            dL_db = d2i * (REAL)2.0 * (b * c * kx - (denom + (REAL)2.0 * b * b) * ky + a * b * kz);
        }
        /* G2 = [[da, db/2],[db/2, dc]]; dSigma_full = M^T G2 M ; dM = 2 G2 M Sigma */
        REAL G2[4] = {dL_da, (REAL)0.5 * dL_db, (REAL)0.5 * dL_db, dL_dc};
        REAL GM[6];
        for (int j = 0; j < 3; j++) {
            GM[j] = G2[0] * Mx[j] + G2[1] * Mx[3 + j];
            GM[3 + j] = G2[2] * Mx[j] + G2[3] * Mx[3 + j];
        }
        REAL dS[9];
        for (int r = 0; r < 3; r++)
            for (int j = 0; j < 3; j++) dS[3 * r + j] = Mx[r] * GM[j] + Mx[3 + r] * GM[3 + j];
        REAL dcov6[6] = {dS[0], (REAL)2.0 * dS[1], (REAL)2.0 * dS[2], dS[4], (REAL)2.0 * dS[5], dS[8]};
        if (d2i == (REAL)0.0) for (int k = 0; k < 6; k++) dcov6[k] = 0;
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)i + k] = dcov6[k];
        REAL dM[6];
        for (int r = 0; r < 2; r++)
            for (int j = 0; j < 3; j++)
                dM[3 * r + j] = (REAL)2.0 * (GM[3 * r] * S[j] + GM[3 * r + 1] * S[3 + j] + GM[3 * r + 2] * S[6 + j]);
        /* dJ = dM Rv^T ; Rv[i][j] = view[4*j+i] */
        const REAL *v = o->view;
        REAL dJ00 = dM[0] * v[0] + dM[1] * v[4] + dM[2] * v[8];
        REAL dJ02 = dM[0] * v[2] + dM[1] * v[6] + dM[2] * v[10];
        REAL dJ11 = dM[3] * v[1] + dM[4] * v[5] + dM[5] * v[9];
        REAL dJ12 = dM[3] * v[2] + dM[4] * v[6] + dM[5] * v[10];
        REAL tz = (REAL)1.0 / tc[2], tz2 = tz * tz, tz3 = tz2 * tz;
        REAL dtx = (xc ? (REAL)0.0 : (REAL)1.0) * -o->focal_x * tz2 * dJ02;
        REAL dty = (yc ? (REAL)0.0 : (REAL)1.0) * -o->focal_y * tz2 * dJ12;
        REAL dtz = -o->focal_x * tz2 * dJ00 - o->focal_y * tz2 * dJ11 + ((REAL)2.0 * o->focal_x * tc[0]) * tz3 * dJ02 +
                   ((REAL)2.0 * o->focal_y * tc[1]) * tz3 * dJ12;
        /* depth output gradient: depth_i = p_view.z */
        dtz += g_depth[i];
        dmean[0] += v[0] * dtx + v[1] * dty + v[2] * dtz;
        dmean[1] += v[4] * dtx + v[5] * dty + v[6] * dtz;
        dmean[2] += v[8] * dtx + v[9] * dty + v[10] * dtz;
        /* (ii) mean2D -> mean3D through the full projection */
        {
            const REAL *pr = o->proj;
            REAL mh[4]; xf4x4(pr, p, mh);
            REAL mw = (REAL)1.0 / (mh[3] + W_EPS);
            REAL mul1 = mh[0] * mw * mw, mul2 = mh[1] * mw * mw;
            dmean[0] += (pr[0] * mw - pr[3] * mul1) * gx + (pr[1] * mw - pr[3] * mul2) * gy;
            dmean[1] += (pr[4] * mw - pr[7] * mul1) * gx + (pr[5] * mw - pr[7] * mul2) * gy;
            dmean[2] += (pr[8] * mw - pr[11] * mul1) * gx + (pr[9] * mw - pr[11] * mul2) * gy;
        }
        /* (iii) SH */
        if (!o->colors_precomp) {
            REAL dirr[3] = {p[0] - o->campos[0], p[1] - o->campos[1], p[2] - o->campos[2]};
            REAL len = R_SQRT(dirr[0] * dirr[0] + dirr[1] * dirr[1] + dirr[2] * dirr[2]);
            REAL d[3] = {dirr[0] / len, dirr[1] / len, dirr[2] / len};
            REAL bs[16], bgx[16], bgy[16], bgz[16];
            sh_basis(o->D, d, bs); sh_basis_grad(o->D, d, bgx, bgy, bgz);
            int nc = (o->D + 1) * (o->D + 1);
            const REAL *sh = o->shs + (size_t)i * M * 3;
            REAL ddir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ch++) {
                REAL gcol = o->clamped[3 * (size_t)i + ch] ? (REAL)0.0 : dL_dcolors[3 * (size_t)i + ch];
                for (int k = 0; k < nc; k++) {
                    dL_dsh[((size_t)i * M + k) * 3 + ch] = bs[k] * gcol;
                    ddir[0] += bgx[k] * sh[3 * k + ch] * gcol;
                    ddir[1] += bgy[k] * sh[3 * k + ch] * gcol;
                    ddir[2] += bgz[k] * sh[3 * k + ch] * gcol;
                }
            }
            /* d normalize: (I - d d^T)/len */
            REAL dot = d[0] * ddir[0] + d[1] * ddir[1] + d[2] * ddir[2];
            for (int k = 0; k < 3; k++) dmean[k] += (ddir[k] - d[k] * dot) / len;
        }
        for (int k = 0; k < 3; k++) dL_dmeans3D[3 * (size_t)i + k] = dmean[k];
        /* (iv) cov3D -> scale, rotation */
        if (!o->cov3D_precomp) {
            const REAL *s = o->scales + 3 * (size_t)i, *q = o->rotations + 4 * (size_t)i;
            REAL mod = o->scale_modifier;
            REAL Rm[9]; quat_to_R(q, Rm);
            REAL Gs[9] = {dcov6[0], (REAL)0.5 * dcov6[1], (REAL)0.5 * dcov6[2], (REAL)0.5 * dcov6[1], dcov6[3],
                          (REAL)0.5 * dcov6[4], (REAL)0.5 * dcov6[2], (REAL)0.5 * dcov6[4], dcov6[5]};
            REAL L[9], dLm[9];
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++) L[3 * r + k] = Rm[3 * r + k] * (mod * s[k]);
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++)
                    dLm[3 * r + k] = (REAL)2.0 * (Gs[3 * r] * L[k] + Gs[3 * r + 1] * L[3 + k] + Gs[3 * r + 2] * L[6 + k]);
            REAL dR[9];
            for (int k = 0; k < 3; k++) {
                REAL acc = 0;
                for (int r = 0; r < 3; r++) { acc += dLm[3 * r + k] * Rm[3 * r + k]; dR[3 * r + k] = dLm[3 * r + k] * (mod * s[k]); }
                dL_dscales[3 * (size_t)i + k] = mod * acc;
            }
            REAL r = q[0], x = q[1], y = q[2], z = q[3];
            REAL dq[4];
            dq[0] = 2 * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
            dq[1] = 2 * (y * dR[1] + z * dR[2] + y * dR[3] - 2 * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2 * x * dR[8]);
            dq[2] = 2 * (-2 * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2 * y * dR[8]);
            dq[3] = 2 * (-2 * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2 * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
            for (int k = 0; k < 4; k++) dL_drot[4 * (size_t)i + k] = dq[k];
        }
    }
    free(g_mean2D); free(g_conic); free(g_depth);
}

int oracle_real_size(void) { return (int)sizeof(REAL); }

/* Number of OpenMP threads used by the parallel loops above (bench.py's cpu_baseline states the count it used). */
int oracle_raster_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
