// deform.hip -- D1..D4: fused HexPlane + deformation-MLP forward and backward for gfx950.
//
// Replaces, per frame, the ~60 unfused PyTorch launches of scene/hexplane.py:73-106 (6*L grid_sample + product +
// concat), scene/deformation.py:67-83,97-148 (trunk Linear + five 2-layer heads, out = in + delta) and the
// activations of gaussian_renderer/__init__.py:97-99.
//
// Layout of the MLP on the matrix cores (exact-f32 v_mfma_f32_32x32x2_f32, so results equal an fmaf chain):
//   * one wave64 owns 32 Gaussians; lane = (g = lane&31, h = lane>>5);
//   * every layer is computed TRANSPOSED, D[feature][gaussian] = W[feature][k] * X[k][gaussian]: the weight matrix is
//     the A operand (row-major [out][in] as torch stores it -> contiguous float4 loads along k, no repacking) and the
//     activations are the B operand;
//   * the MFMA C/D layout puts row (reg&3)+8*(reg>>2)+4*h of column g into lane (g,h) register reg -- which is
//     exactly the B-operand layout of the NEXT layer if its k-steps are walked in that row order.  So activations
//     never leave registers between layers: no LDS, no barriers in the forward kernel.
//   D1 forward: gather features (channel-last planes, one float4 = 4 channels per corner), trunk, heads, epilogue.
//   D2 backward-data: recompute trunk + head hidden layers, back-propagate through the heads with the transposed
//      weight walk, write dH1 / dHidden / relu(hidden) / features for the weight-gradient GEMM, dW2 in-kernel (the
//      only place that needs a transpose, done through a padded per-wave LDS tile).
//   D3 weight gradients: dW = dY^T X with K = #Gaussians; both MFMA operands are read straight from HBM with 128-B
//      coalesced rows, split-K over workgroups, one coalesced atomic flush per workgroup.
//   D4 plane gradients: lanes <-> (x-corner, channel) so that every float atomic instruction covers whole 128-B lines
//      (scattered float atomics run at only ~20 G line-ops/s on MI355X: profiles/r01_atomic_microbench.txt).
#include "common.h"

namespace fdgs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.f;
    return z;
}
// feature row held by (tile t, register r, half h) of the MFMA C/D layout
__device__ __forceinline__ int frow(int t, int r, int h) { return t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h; }

// outputs per head (pos, scale, rot, opacity, shs) and the column of the head's outputs in the packed [N,64] gradient rows
__host__ __device__ __forceinline__ int head_k(int hd) { return hd == 0 ? 3 : hd == 1 ? 3 : hd == 2 ? 4 : hd == 3 ? 1 : 48; }
__host__ __device__ __forceinline__ int head_off(int hd) { return hd == 0 ? 0 : hd == 1 ? 3 : hd == 2 ? 6 : hd == 3 ? 10 : 16; }
constexpr int GCOLS = 64;

// ------------------------------------------------------------------------------------------------ HexPlane gather
struct AxisSample {
    int i0, i1;
    float w0, w1, dscale;  // dscale = d pixel / d coord, 0 where the border clamp is active
};
// grid_sample(align_corners=True, padding_mode='border') un-normalisation (scene/hexplane.py:39-43)
__device__ __forceinline__ AxisSample axis_sample(float coord, int size) {
    AxisSample s;
    const float hi = (float)(size - 1);
    float p = ((coord + 1.f) * 0.5f) * hi;
    s.dscale = (p > 0.f && p < hi) ? 0.5f * hi : 0.f;
    p = fminf(fmaxf(p, 0.f), hi);
    const float f = floorf(p);
    s.i0 = (int)f;
    s.i1 = s.i0 + 1 < size ? s.i0 + 1 : size - 1;
    s.w1 = p - f;
    s.w0 = 1.f - s.w1;
    return s;
}
__device__ __forceinline__ void plane_axes(int k, int& a, int& b) {
    // pairs (0,1),(0,2),(0,3),(1,2),(1,3),(2,3): a indexes the plane's width, b its height
    a = k < 3 ? 0 : (k < 5 ? 1 : 2);
    b = k < 3 ? k + 1 : (k < 5 ? k - 1 : 3);
}

struct DeformDev {
    fdgs_deform_params p;
    fdgs_deform_out out;
    int F;
};

// 4 consecutive features f0..f0+3 (all inside one level because C % 8 == 0) of one Gaussian
__device__ __forceinline__ float4 gather_chunk(const fdgs_deform_params& p, int f0, const float* q) {
    const int lvl = f0 / p.C, c0 = f0 - lvl * p.C;
    float4 prod = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int k = 0; k < 6; k++) {
        int a, b;
        plane_axes(k, a, b);
        const int Wd = p.res[lvl][a], Hd = p.res[lvl][b];
        const AxisSample sx = axis_sample(q[a], Wd), sy = axis_sample(q[b], Hd);
        const float* P = p.planes[lvl][k];
        const float4 v00 = *reinterpret_cast<const float4*>(P + ((size_t)(sy.i0 * Wd + sx.i0) * p.C + c0));
        const float4 v01 = *reinterpret_cast<const float4*>(P + ((size_t)(sy.i0 * Wd + sx.i1) * p.C + c0));
        const float4 v10 = *reinterpret_cast<const float4*>(P + ((size_t)(sy.i1 * Wd + sx.i0) * p.C + c0));
        const float4 v11 = *reinterpret_cast<const float4*>(P + ((size_t)(sy.i1 * Wd + sx.i1) * p.C + c0));
        const float w00 = sx.w0 * sy.w0, w01 = sx.w1 * sy.w0, w10 = sx.w0 * sy.w1, w11 = sx.w1 * sy.w1;
        prod.x *= v00.x * w00 + v01.x * w01 + v10.x * w10 + v11.x * w11;
        prod.y *= v00.y * w00 + v01.y * w01 + v10.y * w10 + v11.y * w11;
        prod.z *= v00.z * w00 + v01.z * w01 + v10.z * w10 + v11.z * w11;
        prod.w *= v00.w * w00 + v01.w * w01 + v10.w * w10 + v11.w * w11;
    }
    return prod;
}

__device__ __forceinline__ void load_query(const fdgs_deform_params& p, int n, float* q, float* xyz) {
    xyz[0] = p.xyz[3 * (size_t)n]; xyz[1] = p.xyz[3 * (size_t)n + 1]; xyz[2] = p.xyz[3 * (size_t)n + 2];
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = (xyz[i] - p.aabb[i]) * (2.0f / (p.aabb[3 + i] - p.aabb[i])) - 1.0f;
    q[3] = p.time ? p.time[n] : p.time_scalar;
}

// features of lane (g,h): chunk j holds features 8j+4h .. +3 = registers 4(j%4)..+3 of tile j/4
template <int FCH>
__device__ __forceinline__ void gather_features(const fdgs_deform_params& p, const float* q, int h, f32x16* feat) {
#pragma unroll
    for (int j = 0; j < FCH; j++) {
        const float4 v = gather_chunk(p, 8 * j + 4 * h, q);
        feat[j / 4][4 * (j % 4) + 0] = v.x; feat[j / 4][4 * (j % 4) + 1] = v.y;
        feat[j / 4][4 * (j % 4) + 2] = v.z; feat[j / 4][4 * (j % 4) + 3] = v.w;
    }
}

// ------------------------------------------------------------------------------------------------ MFMA layers
// out[ot] = bias + W[ot*32.., :] * in     (W row-major [out_dim][in_dim], in_dim = 8*KCH, rows >= out_dim read as 0)
template <int KCH, int OT>
__device__ __forceinline__ void dense(const float* __restrict__ Wm, const float* __restrict__ bias, int in_dim, int out_dim,
                                      const f32x16* in, f32x16* out, int g, int h) {
#pragma unroll
    for (int ot = 0; ot < OT; ot++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = frow(ot, r, h);
            out[ot][r] = row < out_dim ? bias[row] : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < KCH; j++) {
        float4 a4[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            const int row = ot * 32 + g;
            a4[ot] = row < out_dim ? *reinterpret_cast<const float4*>(Wm + (size_t)row * in_dim + 8 * j + 4 * h)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            out[ot] = mfma32(a4[ot].x, in[j / 4][4 * (j % 4) + 0], out[ot]);
            out[ot] = mfma32(a4[ot].y, in[j / 4][4 * (j % 4) + 1], out[ot]);
            out[ot] = mfma32(a4[ot].z, in[j / 4][4 * (j % 4) + 2], out[ot]);
            out[ot] = mfma32(a4[ot].w, in[j / 4][4 * (j % 4) + 3], out[ot]);
        }
    }
}

// dX[xt] += W^T dY : dX row (xt*32+g) valid below in_valid; W row-major [32*YT][ld]
template <int YT, int XT>
__device__ __forceinline__ void dense_bwd_data(const float* __restrict__ Wm, int ld, int in_valid, const f32x16* dY, f32x16* dX,
                                               int g, int h) {
#pragma unroll
    for (int it = 0; it < YT; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int f = frow(it, r, h);
#pragma unroll
            for (int xt = 0; xt < XT; xt++) {
                const int col = xt * 32 + g;
                const float a = col < in_valid ? Wm[(size_t)f * ld + col] : 0.f;
                dX[xt] = mfma32(a, dY[it][r], dX[xt]);
            }
        }
    }
}

template <int T>
__device__ __forceinline__ void relu_inplace(f32x16* x) {
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) x[t][r] = fmaxf(x[t][r], 0.f);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ------------------------------------------------------------------------------------------------ D1 forward
template <int WT, int FCH>
__global__ void __launch_bounds__(256, 2) deform_fwd_kernel(DeformDev d) {
    const fdgs_deform_params& p = d.p;
    constexpr int FT = (FCH + 3) / 4;
    const int lane = threadIdx.x & 63, g = lane & 31, h = lane >> 5;
    const int n_raw = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + g;
    const bool live = n_raw < p.N;
    const int n = live ? n_raw : p.N - 1;
    const int W = WT * 32;
    float q[4], xyz[3];
    load_query(p, n, q, xyz);
    f32x16 feat[FT];
#pragma unroll
    for (int t = 0; t < FT; t++) feat[t] = zero16();
    gather_features<FCH>(p, q, h, feat);
    f32x16 hid[WT];
    dense<FCH, WT>(p.w0, p.b0, d.F, W, feat, hid, g, h);
    relu_inplace<WT>(hid);  // every consumer of the trunk output starts with ReLU (scene/deformation.py:61-65)

    const bool writer = live && h == 0;
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) {
        const int k = head_k(hd);
        f32x16 o0 = zero16(), o1 = zero16();
        if (p.head_on[hd]) {
            f32x16 h1[WT];
            dense<WT * 4, WT>(p.w1[hd], p.b1[hd], W, W, hid, h1, g, h);
            relu_inplace<WT>(h1);
            dense<WT * 4, 1>(p.w2[hd], p.b2[hd], W, k, h1, &o0, g, h);
            if (k > 32) dense<WT * 4, 1>(p.w2[hd] + (size_t)32 * W, p.b2[hd] + 32, W, k - 32, h1, &o1, g, h);
        }
        if (hd == FDGS_HEAD_POS) {
            if (writer) {
                d.out.xyz[3 * (size_t)n] = xyz[0] + o0[0]; d.out.xyz[3 * (size_t)n + 1] = xyz[1] + o0[1];
                d.out.xyz[3 * (size_t)n + 2] = xyz[2] + o0[2];
            }
        } else if (hd == FDGS_HEAD_SCALE) {
            if (writer) {
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    float v = p.scales[3 * (size_t)n + i] + o0[i];
                    d.out.scales[3 * (size_t)n + i] = p.activate ? __expf(v) : v;
                }
            }
        } else if (hd == FDGS_HEAD_ROT) {
            if (writer) {
                const float4 r = reinterpret_cast<const float4*>(p.rotations)[n];
                float v0 = r.x + o0[0], v1 = r.y + o0[1], v2 = r.z + o0[2], v3 = r.w + o0[3];
                if (p.activate) {
                    const float nrm = sqrtf(v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3);
                    const float inv = 1.0f / fmaxf(nrm, 1e-12f);  // F.normalize eps (scene/gaussian_model.py:44)
                    v0 *= inv; v1 *= inv; v2 *= inv; v3 *= inv;
                    if (d.out.rot_norm) d.out.rot_norm[n] = nrm;
                }
                reinterpret_cast<float4*>(d.out.rotations)[n] = make_float4(v0, v1, v2, v3);
            }
        } else if (hd == FDGS_HEAD_OPACITY) {
            if (writer) {
                const float v = p.opacity[n] + o0[0];
                d.out.opacity[n] = p.activate ? sigmoidf_(v) : v;
            }
        } else {
            // shs [N,16,3] = cat(features_dc, features_rest) (+ delta): rows 8u+4h..+3 of tile 0 (u<4) and tile 1 (u<2)
            if (live) {
#pragma unroll
                for (int u = 0; u < 6; u++) {
                    const int row0 = (u < 4 ? 0 : 32) + 8 * (u & 3) + 4 * h;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int m = row0 + i;
                        const float base = m < 3 ? p.shs_dc[(size_t)p.shs_dc_stride * n + m] : p.shs_rest[(size_t)p.shs_rest_stride * n + (m - 3)];
                        v[i] = base + (u < 4 ? o0[4 * (u & 3) + i] : o1[4 * (u & 3) + i]);
                    }
                    *reinterpret_cast<float4*>(d.out.shs + 48 * (size_t)n + row0) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: prep
// Per Gaussian: activation Jacobians -> packed pre-activation output gradients G[n][64]; direct (identity) paths.
struct PrepArgs {
    int N, Npad, activate, dc_stride, rest_stride;
    const float *g_xyz, *g_scales, *g_rot, *g_opacity, *g_shs, *out_scales, *out_rot, *out_opacity, *rot_norm;
    float *d_xyz, *d_scales, *d_rot, *d_opacity, *d_shs_dc, *d_shs_rest;
    float* G;
};
__global__ void __launch_bounds__(256) deform_bwd_prep_kernel(PrepArgs a) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= a.Npad) return;
    float row[16];
#pragma unroll
    for (int i = 0; i < 16; i++) row[i] = 0.f;
    float4* G4 = reinterpret_cast<float4*>(a.G + (size_t)n * GCOLS);
    if (n < a.N) {
        if (a.g_xyz) { row[0] = a.g_xyz[3 * (size_t)n]; row[1] = a.g_xyz[3 * (size_t)n + 1]; row[2] = a.g_xyz[3 * (size_t)n + 2]; }
        if (a.g_scales) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const float gs = a.g_scales[3 * (size_t)n + i];
                row[3 + i] = a.activate ? gs * a.out_scales[3 * (size_t)n + i] : gs;  // d exp
            }
        }
        if (a.g_rot) {
            const float4 gr = reinterpret_cast<const float4*>(a.g_rot)[n];
            if (a.activate) {
                const float4 o = reinterpret_cast<const float4*>(a.out_rot)[n];
                const float nrm = a.rot_norm[n];
                if (nrm > 1e-12f) {
                    const float dot = o.x * gr.x + o.y * gr.y + o.z * gr.z + o.w * gr.w;
                    const float inv = 1.0f / nrm;
                    row[6] = (gr.x - o.x * dot) * inv; row[7] = (gr.y - o.y * dot) * inv;
                    row[8] = (gr.z - o.z * dot) * inv; row[9] = (gr.w - o.w * dot) * inv;
                } else {  // below the F.normalize eps the division is by the constant 1e-12
                    row[6] = gr.x * 1e12f; row[7] = gr.y * 1e12f; row[8] = gr.z * 1e12f; row[9] = gr.w * 1e12f;
                }
            } else { row[6] = gr.x; row[7] = gr.y; row[8] = gr.z; row[9] = gr.w; }
        }
        if (a.g_opacity) {
            const float go = a.g_opacity[n];
            const float o = a.activate ? a.out_opacity[n] : 0.f;
            row[10] = a.activate ? go * o * (1.f - o) : go;
        }
        if (a.d_xyz) { a.d_xyz[3 * (size_t)n] += row[0]; a.d_xyz[3 * (size_t)n + 1] += row[1]; a.d_xyz[3 * (size_t)n + 2] += row[2]; }
        if (a.d_scales) { a.d_scales[3 * (size_t)n] += row[3]; a.d_scales[3 * (size_t)n + 1] += row[4]; a.d_scales[3 * (size_t)n + 2] += row[5]; }
        if (a.d_rot) {
            float4 v = reinterpret_cast<float4*>(a.d_rot)[n];
            v.x += row[6]; v.y += row[7]; v.z += row[8]; v.w += row[9];
            reinterpret_cast<float4*>(a.d_rot)[n] = v;
        }
        if (a.d_opacity) a.d_opacity[n] += row[10];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) G4[i] = make_float4(row[4 * i], row[4 * i + 1], row[4 * i + 2], row[4 * i + 3]);
    for (int i = 0; i < 12; i++) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < a.N && a.g_shs) {
            v = *reinterpret_cast<const float4*>(a.g_shs + 48 * (size_t)n + 4 * i);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int m = 4 * i + c;
                if (m < 3) { if (a.d_shs_dc) a.d_shs_dc[(size_t)a.dc_stride * n + m] += vv[c]; }
                else if (a.d_shs_rest) a.d_shs_rest[(size_t)a.rest_stride * n + (m - 3)] += vv[c];
            }
        }
        G4[4 + i] = v;
    }
}

// ------------------------------------------------------------------------------------------------ D2 backward-data
struct BwdScratch {
    float *G, *DH1, *DHID, *RH, *FEAT, *DFEAT;
    int Npad;
};
struct BwdDev {
    fdgs_deform_params p;
    BwdScratch s;
    float* d_w2[FDGS_NUM_HEADS];
    float* d_b2[FDGS_NUM_HEADS];
    int F;
    int head_slot[FDGS_NUM_HEADS];  // index of the head's dH1 slab
};

template <int WT>
struct LdsT {
    static constexpr int STRIDE = WT * 32 + 4;  // padded row: conflict-free ds_write_b128 and ds_read_b32
};

template <int WT, int FCH>
__global__ void __launch_bounds__(256, 2) deform_bwd_data_kernel(BwdDev d) {
    const fdgs_deform_params& p = d.p;
    constexpr int FT = (FCH + 3) / 4;
    constexpr int STRIDE = LdsT<WT>::STRIDE;
    __shared__ float lds_all[4 * 32 * STRIDE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane & 31, h = lane >> 5;
    float* lds = lds_all + wave * 32 * STRIDE;
    const int n0 = (blockIdx.x * 4 + wave) * 32;  // first Gaussian of this wave (rows < Npad always exist in scratch)
    const int n_row = n0 + g;
    const int n = n_row < p.N ? n_row : p.N - 1;
    const int W = WT * 32, F = d.F;
    float q[4], xyz[3];
    load_query(p, n, q, xyz);
    f32x16 feat[FT];
#pragma unroll
    for (int t = 0; t < FT; t++) feat[t] = zero16();
    gather_features<FCH>(p, q, h, feat);
#pragma unroll
    for (int j = 0; j < FCH; j++)
        *reinterpret_cast<float4*>(d.s.FEAT + (size_t)n_row * F + 8 * j + 4 * h) =
            make_float4(feat[j / 4][4 * (j % 4)], feat[j / 4][4 * (j % 4) + 1], feat[j / 4][4 * (j % 4) + 2], feat[j / 4][4 * (j % 4) + 3]);
    f32x16 hid[WT], dhid[WT];
    dense<FCH, WT>(p.w0, p.b0, F, W, feat, hid, g, h);
    relu_inplace<WT>(hid);
#pragma unroll
    for (int t = 0; t < WT; t++) {
        dhid[t] = zero16();
#pragma unroll
        for (int u = 0; u < 4; u++)
            *reinterpret_cast<float4*>(d.s.RH + (size_t)n_row * W + t * 32 + 8 * u + 4 * h) =
                make_float4(hid[t][4 * u], hid[t][4 * u + 1], hid[t][4 * u + 2], hid[t][4 * u + 3]);
    }
    const float* Grow = d.s.G + (size_t)n_row * GCOLS;

    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) {
        if (!p.head_on[hd]) continue;
        const int k = head_k(hd), off = head_off(hd);
        uint32_t mask[WT];  // bit r of mask[t]: h1[t][r] > 0
        {
            f32x16 h1[WT];
            dense<WT * 4, WT>(p.w1[hd], p.b1[hd], W, W, hid, h1, g, h);
#pragma unroll
            for (int t = 0; t < WT; t++) {
                mask[t] = 0;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    h1[t][r] = fmaxf(h1[t][r], 0.f);
                    mask[t] |= (h1[t][r] > 0.f ? 1u : 0u) << r;
                }
                // transposed copy relu(h1)[gaussian][feature] for the dW2 product
#pragma unroll
                for (int u = 0; u < 4; u++)
                    *reinterpret_cast<float4*>(lds + g * STRIDE + t * 32 + 8 * u + 4 * h) =
                        make_float4(h1[t][4 * u], h1[t][4 * u + 1], h1[t][4 * u + 2], h1[t][4 * u + 3]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // ---- dW2[o][in] += sum_g G[g][o] * relu(h1)[g][in]; db2 through a column of ones
        // (two feature tiles of relu(h1) at a time: keeps the accumulator footprint at 48 VGPRs)
        for (int ot2 = 0; ot2 * 32 < k; ot2++) {
            const int o = ot2 * 32 + g;
#pragma unroll
            for (int tb = 0; tb < WT; tb += 2) {
                f32x16 acc0 = zero16(), acc1 = zero16(), accb = zero16();
#pragma unroll 4
                for (int s = 0; s < 16; s++) {
                    const int gs = 2 * s + h;
                    const float a = o < k ? d.s.G[(size_t)(n0 + gs) * GCOLS + off + o] : 0.f;
                    acc0 = mfma32(a, lds[gs * STRIDE + tb * 32 + g], acc0);
                    acc1 = mfma32(a, lds[gs * STRIDE + (tb + 1) * 32 + g], acc1);
                    if (tb == 0) accb = mfma32(a, 1.0f, accb);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int orow = ot2 * 32 + frow(0, r, h);
                    if (orow < k) {
                        atomicAdd(&d.d_w2[hd][(size_t)orow * W + tb * 32 + g], acc0[r]);
                        atomicAdd(&d.d_w2[hd][(size_t)orow * W + (tb + 1) * 32 + g], acc1[r]);
                        if (tb == 0 && g == 0) atomicAdd(&d.d_b2[hd][orow], accb[r]);
                    }
                }
            }
        }
        // ---- dh1 = W2^T G_head, masked by relu'(h1)
        f32x16 dh1[WT];
#pragma unroll
        for (int t = 0; t < WT; t++) dh1[t] = zero16();
        for (int s = 0; 2 * s < k; s++) {
            const int o = 2 * s + h;
            const float b = o < k ? Grow[off + o] : 0.f;
#pragma unroll
            for (int t = 0; t < WT; t++) {
                const float a = o < k ? p.w2[hd][(size_t)o * W + t * 32 + g] : 0.f;
                dh1[t] = mfma32(a, b, dh1[t]);
            }
        }
        float* slab = d.s.DH1 + (size_t)d.head_slot[hd] * d.s.Npad * W;
#pragma unroll
        for (int t = 0; t < WT; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) dh1[t][r] = ((mask[t] >> r) & 1u) ? dh1[t][r] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; u++)
                *reinterpret_cast<float4*>(slab + (size_t)n_row * W + t * 32 + 8 * u + 4 * h) =
                    make_float4(dh1[t][4 * u], dh1[t][4 * u + 1], dh1[t][4 * u + 2], dh1[t][4 * u + 3]);
        }
        // ---- dhid += W1^T dh1
        dense_bwd_data<WT, WT>(p.w1[hd], W, W, dh1, dhid, g, h);
        __builtin_amdgcn_wave_barrier();
    }
    // relu'(hidden), store for the trunk weight gradient, then dfeat = W0^T dhid
#pragma unroll
    for (int t = 0; t < WT; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++) dhid[t][r] = hid[t][r] > 0.f ? dhid[t][r] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; u++)
            *reinterpret_cast<float4*>(d.s.DHID + (size_t)n_row * W + t * 32 + 8 * u + 4 * h) =
                make_float4(dhid[t][4 * u], dhid[t][4 * u + 1], dhid[t][4 * u + 2], dhid[t][4 * u + 3]);
    }
    f32x16 dfeat[FT];
#pragma unroll
    for (int t = 0; t < FT; t++) dfeat[t] = zero16();
    dense_bwd_data<WT, FT>(p.w0, F, F, dhid, dfeat, g, h);
#pragma unroll
    for (int j = 0; j < FCH; j++)
        *reinterpret_cast<float4*>(d.s.DFEAT + (size_t)n_row * F + 8 * j + 4 * h) =
            make_float4(dfeat[j / 4][4 * (j % 4)], dfeat[j / 4][4 * (j % 4) + 1], dfeat[j / 4][4 * (j % 4) + 2],
                        dfeat[j / 4][4 * (j % 4) + 3]);
}

// ------------------------------------------------------------------------------------------------ D3 weight grads
// dW[m][c] += sum_n DY[n][m] * X[n][c]   (m < 32*MT rows of DY, c < ncols of X), db[m] += sum_n DY[n][m]
struct WgradJob {
    const float* DY; const float* X; float* dW; float* db;
    int ldx, ncols, ldw;
};
struct WgradArgs {
    WgradJob job[FDGS_NUM_HEADS + 1];
    int njobs, Npad, W, chunk;
};
template <int WT>
__global__ void __launch_bounds__(256, 2) deform_wgrad_kernel(WgradArgs a) {
    const WgradJob J = a.job[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane & 31, h = lane >> 5;
    constexpr int WAVES_PER_MT = 4 / WT;  // WT=4: one wave per row tile; WT=2: two waves split K
    const int mt = wave % WT, ksub = wave / WT;
    int n_begin = blockIdx.x * a.chunk, n_end = n_begin + a.chunk;
    if (n_end > a.Npad) n_end = a.Npad;
    const int span = (n_end - n_begin) / WAVES_PER_MT;  // chunk is a multiple of 4
    n_begin += ksub * span;
    n_end = n_begin + span;
    const int CT = (J.ncols + 31) / 32;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = zero16();
    float asum = 0.f;
    const int W = a.W;
    const bool c0 = g < J.ncols, c1 = 32 + g < J.ncols, c2 = 64 + g < J.ncols, c3 = 96 + g < J.ncols;
#pragma unroll 8
    for (int n = n_begin + h; n < n_end; n += 2) {
        const float av = J.DY[(size_t)n * W + mt * 32 + g];
        const float* xr = J.X + (size_t)n * J.ldx + g;
        const float b0 = c0 ? xr[0] : 0.f;
        const float b1 = c1 ? xr[32] : 0.f;
        const float b2 = c2 ? xr[64] : 0.f;
        const float b3 = c3 ? xr[96] : 0.f;
        asum += av;
        acc[0] = mfma32(av, b0, acc[0]);
        if (CT > 1) acc[1] = mfma32(av, b1, acc[1]);
        if (CT > 2) acc[2] = mfma32(av, b2, acc[2]);
        if (CT > 3) acc[3] = mfma32(av, b3, acc[3]);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (t < CT) {
            const int col = t * 32 + g;
            if (col < J.ncols) {
#pragma unroll
                for (int r = 0; r < 16; r++) atomicAdd(&J.dW[(size_t)frow(mt, r, h) * J.ldw + col], acc[t][r]);
            }
        }
    }
    asum += __shfl_xor(asum, 32, 64);
    if (h == 0) atomicAdd(&J.db[mt * 32 + g], asum);
}

// ------------------------------------------------------------------------------------------------ D4 plane grads
struct PlaneGradArgs {
    fdgs_deform_params p;
    const float* DFEAT;
    float* d_planes[FDGS_MAX_LEVELS][6];
    float* d_xyz;
    int F;
};
template <int C>
__global__ void __launch_bounds__(256) deform_plane_grad_kernel(PlaneGradArgs a) {
    const fdgs_deform_params& p = a.p;
    constexpr int LPG = 2 * C, GPW = 64 / LPG;
    const int lane = threadIdx.x & 63;
    const int ch = lane % C, xc = (lane / C) & 1, gsub = lane / LPG;
    const int n_raw = (blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + gsub;
    const bool live = n_raw < p.N;
    const int n = live ? n_raw : p.N - 1;
    float q[4], xyz[3];
    load_query(p, n, q, xyz);
    float dq[3] = {0.f, 0.f, 0.f};
    for (int lvl = 0; lvl < p.L; lvl++) {
        const float df = live ? a.DFEAT[(size_t)n * a.F + lvl * C + ch] : 0.f;
        float vk[6], sk[6], tk[6], wA[6], wB[6], dsx[6], dsy[6];
        size_t oA[6], oB[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            int ax, bx;
            plane_axes(k, ax, bx);
            const int Wd = p.res[lvl][ax], Hd = p.res[lvl][bx];
            const AxisSample sx = axis_sample(q[ax], Wd), sy = axis_sample(q[bx], Hd);
            const int xi = xc ? sx.i1 : sx.i0;
            const float wx = xc ? sx.w1 : sx.w0;
            oA[k] = (size_t)(sy.i0 * Wd + xi) * C + ch;
            oB[k] = (size_t)(sy.i1 * Wd + xi) * C + ch;
            const float v0 = p.planes[lvl][k][oA[k]], v1 = p.planes[lvl][k][oB[k]];
            sk[k] = sy.w0 * v0 + sy.w1 * v1;             // d/d(ix) carries sign(xc)
            tk[k] = wx * (v1 - v0);                      // d/d(iy)
            float part = wx * sk[k];
            if (C == 16) {
                auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(part), __float_as_uint(part), false, false);
                part = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            } else {
                auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
                part = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
            vk[k] = part;
            wA[k] = wx * sy.w0; wB[k] = wx * sy.w1;
            dsx[k] = sx.dscale; dsy[k] = sy.dscale;
        }
        float pre[6], suf[6];
        pre[0] = 1.f; suf[5] = 1.f;
#pragma unroll
        for (int k = 1; k < 6; k++) pre[k] = pre[k - 1] * vk[k - 1];
#pragma unroll
        for (int k = 4; k >= 0; k--) suf[k] = suf[k + 1] * vk[k + 1];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            int ax, bx;
            plane_axes(k, ax, bx);
            const float dv = df * pre[k] * suf[k];
            float* dP = a.d_planes[lvl][k];
            if (dP && live) {
                atomicAdd(&dP[oA[k]], dv * wA[k]);
                atomicAdd(&dP[oB[k]], dv * wB[k]);
            }
            const float gx = dv * (xc ? sk[k] : -sk[k]) * dsx[k];
            const float gy = dv * tk[k] * dsy[k];
            if (ax < 3) dq[ax] += gx;   // ax in {0,1,2}
            if (bx < 3) dq[bx] += gy;   // bx == 3 is time: no gradient
        }
    }
    if (a.d_xyz) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            float v = dq[i];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            if (LPG == 64) v += __shfl_xor(v, 32, 64);
            dq[i] = v;
        }
        if (live && (lane % LPG) == 0) {
#pragma unroll
            for (int i = 0; i < 3; i++) a.d_xyz[3 * (size_t)n + i] += dq[i] * (2.0f / (p.aabb[3 + i] - p.aabb[i]));
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int validate_deform(const fdgs_deform_params* p) {
    FDGS_REQUIRE(p != nullptr, "params is NULL");
    FDGS_REQUIRE(p->N >= 0, "N < 0");
    FDGS_REQUIRE(p->C == 16 || p->C == 32, "output_coordinate_dim must be 16 or 32");
    FDGS_REQUIRE(p->L >= 1 && p->L <= FDGS_MAX_LEVELS, "1 <= len(multires) <= 4");
    FDGS_REQUIRE(p->W == 64 || p->W == 128, "net_width must be 64 or 128");
    const int F = p->C * p->L;
    FDGS_REQUIRE(F <= 128, "C*L must be <= 128");
    for (int l = 0; l < p->L; l++) {
        for (int k = 0; k < 6; k++) FDGS_REQUIRE(p->planes[l][k] != nullptr, "plane pointer is NULL");
        for (int i = 0; i < 4; i++) FDGS_REQUIRE(p->res[l][i] >= 2, "plane resolution must be >= 2");
    }
    FDGS_REQUIRE(p->w0 && p->b0, "trunk weights missing");
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++)
        if (p->head_on[hd]) FDGS_REQUIRE(p->w1[hd] && p->b1[hd] && p->w2[hd] && p->b2[hd], "head weights missing");
    if (p->N > 0) FDGS_REQUIRE(p->xyz && p->scales && p->rotations && p->opacity && p->shs_dc && p->shs_rest, "input pointer missing");
    for (int i = 0; i < 3; i++) FDGS_REQUIRE(p->aabb[3 + i] != p->aabb[i], "degenerate aabb");
    FDGS_REQUIRE(p->shs_dc_stride >= 3 && p->shs_rest_stride >= 45, "shs strides too small");
    return FDGS_OK;
}

template <template <int, int> class Launcher, typename Arg>
static int dispatch_wf(int W, int F, hipStream_t stream, int blocks, const Arg& arg) {
#define FDGS_CASE(WT_, FCH_) \
    if (W == WT_ * 32 && F == FCH_ * 8) { Launcher<WT_, FCH_>::go(stream, blocks, arg); return FDGS_OK; }
    FDGS_CASE(2, 4) FDGS_CASE(2, 6) FDGS_CASE(2, 8) FDGS_CASE(2, 12) FDGS_CASE(2, 16)
    FDGS_CASE(4, 4) FDGS_CASE(4, 6) FDGS_CASE(4, 8) FDGS_CASE(4, 12) FDGS_CASE(4, 16)
#undef FDGS_CASE
    return fail(FDGS_E_INVALID, "%s", "unsupported (net_width, C*L) combination");
}
template <int WT, int FCH>
struct FwdLauncher {
    static void go(hipStream_t s, int blocks, const DeformDev& d) {
        hipLaunchKernelGGL((deform_fwd_kernel<WT, FCH>), dim3(blocks), dim3(256), 0, s, d);
    }
};
template <int WT, int FCH>
struct BwdLauncher {
    static void go(hipStream_t s, int blocks, const BwdDev& d) {
        hipLaunchKernelGGL((deform_bwd_data_kernel<WT, FCH>), dim3(blocks), dim3(256), 0, s, d);
    }
};

static size_t npad_of(int N) { return ((size_t)(N > 0 ? N : 1) + 127) / 128 * 128; }
static int active_heads(const fdgs_deform_params* p) {
    int c = 0;
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) c += p->head_on[hd] ? 1 : 0;
    return c;
}

}  // namespace fdgs

using namespace fdgs;

extern "C" int fdgs_deform_fwd(void* stream_, const fdgs_deform_params* p, const fdgs_deform_out* out) {
    int rc = validate_deform(p);
    if (rc) return rc;
    FDGS_REQUIRE(out && (p->N == 0 || (out->xyz && out->scales && out->rotations && out->opacity && out->shs)), "output pointer missing");
    if (p->N == 0) return FDGS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    DeformDev d;
    d.p = *p; d.out = *out; d.F = p->C * p->L;
    {
        FDGS_TIMED("deform_fwd", stream);
        rc = dispatch_wf<FwdLauncher>(p->W, d.F, stream, cdiv(p->N, 128), d);
    }
    if (rc) return rc;
    FDGS_LAUNCH_CHECK("deform_fwd", 0, stream);
    return FDGS_OK;
}

extern "C" int fdgs_deform_bwd_scratch_bytes(const fdgs_deform_params* p, size_t* bytes) {
    int rc = validate_deform(p);
    if (rc) return rc;
    FDGS_REQUIRE(bytes, "bytes is NULL");
    const size_t Np = npad_of(p->N), F = (size_t)p->C * p->L, W = p->W;
    *bytes = Np * (GCOLS + (size_t)active_heads(p) * W + 2 * W + 2 * F) * sizeof(float) + 1024;
    return FDGS_OK;
}

extern "C" int fdgs_deform_bwd(void* stream_, const fdgs_deform_params* p, const fdgs_deform_grads* g) {
    int rc = validate_deform(p);
    if (rc) return rc;
    FDGS_REQUIRE(g && g->scratch, "grads/scratch is NULL");
    if (p->N == 0) return FDGS_OK;
    if (p->activate) {
        FDGS_REQUIRE(!g->g_scales || g->out_scales, "out_scales needed with activate=1");
        FDGS_REQUIRE(!g->g_rotations || (g->out_rotations && g->rot_norm), "out_rotations/rot_norm needed with activate=1");
        FDGS_REQUIRE(!g->g_opacity || g->out_opacity, "out_opacity needed with activate=1");
    }
    hipStream_t stream = (hipStream_t)stream_;
    const size_t Np = npad_of(p->N), F = (size_t)p->C * p->L, W = p->W;
    const int nh = active_heads(p);
    BwdScratch s;
    float* base = reinterpret_cast<float*>(g->scratch);
    s.Npad = (int)Np;
    s.G = base; base += Np * GCOLS;
    s.DH1 = base; base += Np * W * nh;
    s.DHID = base; base += Np * W;
    s.RH = base; base += Np * W;
    s.FEAT = base; base += Np * F;
    s.DFEAT = base;
    // prep: activation Jacobians, identity paths, packed gradient rows
    PrepArgs pa{};
    pa.N = p->N; pa.Npad = (int)Np; pa.activate = p->activate; pa.dc_stride = p->shs_dc_stride; pa.rest_stride = p->shs_rest_stride;
    pa.g_xyz = g->g_xyz; pa.g_scales = g->g_scales; pa.g_rot = g->g_rotations; pa.g_opacity = g->g_opacity; pa.g_shs = g->g_shs;
    pa.out_scales = g->out_scales; pa.out_rot = g->out_rotations; pa.out_opacity = g->out_opacity; pa.rot_norm = g->rot_norm;
    pa.d_xyz = g->d_xyz; pa.d_scales = g->d_scales; pa.d_rot = g->d_rotations; pa.d_opacity = g->d_opacity;
    pa.d_shs_dc = g->d_shs_dc; pa.d_shs_rest = g->d_shs_rest; pa.G = s.G;
    { FDGS_TIMED("deform_bwd_prep", stream); hipLaunchKernelGGL(deform_bwd_prep_kernel, dim3(cdiv((long long)Np, 256)), dim3(256), 0, stream, pa); }
    FDGS_LAUNCH_CHECK("deform_bwd_prep", 0, stream);
    if (nh == 0) return FDGS_OK;  // no head active: the deformation is the identity
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++)
        if (p->head_on[hd]) FDGS_REQUIRE(g->d_w1[hd] && g->d_b1[hd] && g->d_w2[hd] && g->d_b2[hd], "head gradient buffer missing");
    FDGS_REQUIRE(g->d_w0 && g->d_b0, "trunk gradient buffer missing");
    BwdDev bd;
    bd.p = *p; bd.s = s; bd.F = (int)F;
    int slot = 0;
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) {
        bd.d_w2[hd] = g->d_w2[hd]; bd.d_b2[hd] = g->d_b2[hd];
        bd.head_slot[hd] = p->head_on[hd] ? slot++ : 0;
    }
    {
        FDGS_TIMED("deform_bwd_data", stream);
        rc = dispatch_wf<BwdLauncher>(p->W, (int)F, stream, (int)(Np / 128), bd);
    }
    if (rc) return rc;
    FDGS_LAUNCH_CHECK("deform_bwd_data", 0, stream);
    // weight gradients: one job per active head (dW1, db1) + the trunk (dW0, db0)
    WgradArgs wa{};
    wa.Npad = (int)Np; wa.W = (int)W;
    int nj = 0;
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) {
        if (!p->head_on[hd]) continue;
        WgradJob& J = wa.job[nj++];
        J.DY = s.DH1 + (size_t)bd.head_slot[hd] * Np * W; J.X = s.RH; J.dW = g->d_w1[hd]; J.db = g->d_b1[hd];
        J.ldx = (int)W; J.ncols = (int)W; J.ldw = (int)W;
    }
    {
        WgradJob& J = wa.job[nj++];
        J.DY = s.DHID; J.X = s.FEAT; J.dW = g->d_w0; J.db = g->d_b0; J.ldx = (int)F; J.ncols = (int)F; J.ldw = (int)F;
    }
    wa.njobs = nj;
    int ksplit = 1024 / nj;
    if (ksplit < 1) ksplit = 1;
    int chunk = (int)((Np + ksplit - 1) / ksplit);
    chunk = (chunk + 3) / 4 * 4;
    if (chunk < 64) chunk = 64;
    wa.chunk = chunk;
    const int kblocks = (int)((Np + chunk - 1) / chunk);
    if (W == 128) { FDGS_TIMED("deform_wgrad", stream); hipLaunchKernelGGL((deform_wgrad_kernel<4>), dim3(kblocks, nj), dim3(256), 0, stream, wa); }
    else { FDGS_TIMED("deform_wgrad", stream); hipLaunchKernelGGL((deform_wgrad_kernel<2>), dim3(kblocks, nj), dim3(256), 0, stream, wa); }
    FDGS_LAUNCH_CHECK("deform_wgrad", 0, stream);
    // plane + coordinate gradients
    bool any_plane = g->d_xyz != nullptr;
    PlaneGradArgs ga{};
    ga.p = *p; ga.DFEAT = s.DFEAT; ga.d_xyz = g->d_xyz; ga.F = (int)F;
    for (int l = 0; l < p->L; l++)
        for (int k = 0; k < 6; k++) { ga.d_planes[l][k] = g->d_planes[l][k]; any_plane = any_plane || g->d_planes[l][k]; }
    if (any_plane) {
        if (p->C == 16) { FDGS_TIMED("deform_plane_grad", stream); hipLaunchKernelGGL((deform_plane_grad_kernel<16>), dim3(cdiv(p->N, 8)), dim3(256), 0, stream, ga); }
        else { FDGS_TIMED("deform_plane_grad", stream); hipLaunchKernelGGL((deform_plane_grad_kernel<32>), dim3(cdiv(p->N, 4)), dim3(256), 0, stream, ga); }
        FDGS_LAUNCH_CHECK("deform_plane_grad", 0, stream);
    }
    return FDGS_OK;
}
