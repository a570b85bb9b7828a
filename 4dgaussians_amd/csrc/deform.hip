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
//     never leave registers between layers (hidden layers use the "interleaved" tile layout described at the MFMA
//     layer helpers below, which makes W X and W^T dY both read the weights with 16-byte loads).
//   D1 forward: gather features (channel-last planes, one float4 = 4 channels per corner), trunk, heads (k <= 4 outputs on
//      the 4x4x1 MFMA), epilogue; optionally parks features / relu(hidden) / relu(h1) for the backward ("saved").
//   D2 backward-data (persistent, one workgroup per CU): per 32-Gaussian tile and head -- the relu(h1) tile (copied from
//      the saved activations, or recomputed), dW2/db2 (register sums for the k <= 4 heads, column ownership over the
//      workgroup's four tiles for the 48-row SH head), dh1 = W2^T G, dhid += W1^T dh1; then dfeat = W0^T dhid.  Writes
//      dH1 / dHidden (/ relu(hidden) / features when recomputing) for the weight-gradient GEMM.
//   D3 weight gradients: dW = dY^T X with K = #Gaussians; one wave owns a whole [W x W] product for its slice, one 16-byte
//      load per operand per 16 MFMAs, LDS reduction over the workgroup, one coalesced atomic flush per workgroup.
//   D4 plane gradients: lanes <-> (x-corner, channel) so that every float atomic instruction covers whole texel lines
//      (scattered float atomics run at only ~20 G line-ops/s on MI355X: profiles/r01_atomic_microbench.txt); the three
//      time planes are privatised in LDS.
//   Round 3: the backward kernels walk only the 32-Gaussian tiles that carry a non-zero gradient row (tile_compact_kernel turns the
//      rasterizer backward's per-tile flags into lists: culled / occluded Gaussians add exactly zero to every sum); D1 deals the tiles left
//      over after the last full round of its persistent loop out by head (FDGS_D1_SPLIT).
// Environment knobs (development / A-B only, defaults are the tuned values): FDGS_SMALL_HEADS, FDGS_USE_SAVED, FDGS_SKIP_DEAD, FDGS_D1_SPLIT,
// FDGS_D1_WGS (0 = one workgroup per four tiles instead of the persistent tile loop), FDGS_D2_WGS, FDGS_WGRAD_WGS,
// FDGS_WGRAD_TRUNK, FDGS_PG_LDS, FDGS_PG_LDS_KB, FDGS_PG_WGS; -DFDGS_PROFILE_D1 / -DFDGS_PROFILE_D2 add an in-kernel
// s_memtime phase profile of D1 / D2 (printed once to stderr); -DFDGS_DEV_ONLY_44 builds only the (128, 32) instance.
#include "common.h"

#include <vector>

namespace fdgs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// 16 independent 4x4x1 outer products: lane 4b+i holds A_b[i], lane 4b+j holds B_b[j], lane 4b+j register i gets D_b[i][j]
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}
// same with block ABID of the A operand broadcast to all 16 blocks (CBSZ = 4)
template <int ABID>
__device__ __forceinline__ f32x4 mfma4_bcast(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);
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
// first row of a head's k output rows in arrays that stack the five heads (3 + 3 + 4 + 1 + 48 = 59 rows)
__host__ __device__ __forceinline__ int head_row0(int hd) { return hd == 0 ? 0 : hd == 1 ? 3 : hd == 2 ? 6 : hd == 3 ? 10 : 11; }

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

// inv2[i] = 2 / (aabb[3+i] - aabb[i]), the same float division the reference performs, done once on the host (three
// full-precision divisions per lane are ~36 VALU instructions)
struct AabbScale { float inv2[3]; };
static AabbScale aabb_scale(const fdgs_deform_params* p) {
    AabbScale s;
    for (int i = 0; i < 3; i++) s.inv2[i] = 2.0f / (p->aabb[3 + i] - p->aabb[i]);
    return s;
}
struct DeformDev {
    fdgs_deform_params p;
    fdgs_deform_out out;
    AabbScale sc;
    int F;
    int small_heads;   // 1: k <= 4 heads on the 4x4x1 MFMA (default), 0: padded 32x32x2 tiles (A/B switch, FDGS_SMALL_HEADS)
    int split_tail;    // 1: the tiles left over after the last full round of the persistent loop are split by head over the waves (FDGS_D1_SPLIT)
    // optional saved activations for the backward (rows < Npad): features [Np][F], relu(hidden) [Np][W], relu(h1) [slot][Np][W]
    float *sv_feat, *sv_rh, *sv_h1;
    int ntiles;                     // 32-Gaussian tiles (a multiple of 4)
    unsigned long long* prof;       // development builds (-DFDGS_PROFILE_D1): cycle sums per phase
    uint32_t* sv_hmask;             // [Npad/32][64 lanes][4]: bit r of word t = relu(hidden) tile t register r > 0 (what D2's lane needs)
    int Npad;
    int head_slot[FDGS_NUM_HEADS];
    const float* packed;            // form 16: W0 / W1 as operand streams (pack_weights16_kernel)
    const float* feat;              // weight-stationary form: the HexPlane features [Npad][F] (deform_gather_kernel)
    unsigned head_mask;             // weight-stationary form: bit hd = head hd is on (a scalar the lanes can test with their own head index)
    int skew;                       // form 16: start delay (s_memtime ticks) of the second half of the grid (FDGS_D16_SKEW)
};

// 4 consecutive features f0..f0+3 (all inside one level because C % 8 == 0) of one Gaussian.
// Texel addresses are 32-bit byte offsets from the (wave-uniform) plane pointer: one VALU op per address and the
// SGPR-base + VGPR-offset load form, instead of 64-bit multiply-adds per corner (planes are < 2^32 bytes by validation).
// The level index is the same for both lane halves (C % 8 == 0): computed from the wave-uniform chunk index and pinned
// to an SGPR, so that the resolutions and plane pointers are scalar (kernarg) loads.  (Derived from the per-lane f0 they
// were two dependent VECTOR loads per chunk, each a full memory round trip in front of the 24 texel requests.)
__device__ __forceinline__ float4 gather_chunk(const fdgs_deform_params& p, int j, int h, const float* q) {
    const int lvl = __builtin_amdgcn_readfirstlane((8 * j) / p.C);
    const int c0 = 8 * j + 4 * h - lvl * p.C;
    float4 prod = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int k = 0; k < 6; k++) {
        int a, b;
        plane_axes(k, a, b);
        const int Wd = p.res[lvl][a], Hd = p.res[lvl][b];
        const AxisSample sx = axis_sample(q[a], Wd), sy = axis_sample(q[b], Hd);
        const char* P = reinterpret_cast<const char*>(p.planes[lvl][k]);
        const uint32_t texel = (uint32_t)p.C * 4u, cb = (uint32_t)c0 * 4u;
        const uint32_t r0 = (uint32_t)(sy.i0 * Wd) * texel + cb, r1 = (uint32_t)(sy.i1 * Wd) * texel + cb;
        const uint32_t x0 = (uint32_t)sx.i0 * texel, x1 = (uint32_t)sx.i1 * texel;
        const float4 v00 = *reinterpret_cast<const float4*>(P + (r0 + x0));
        const float4 v01 = *reinterpret_cast<const float4*>(P + (r0 + x1));
        const float4 v10 = *reinterpret_cast<const float4*>(P + (r1 + x0));
        const float4 v11 = *reinterpret_cast<const float4*>(P + (r1 + x1));
        const float w00 = sx.w0 * sy.w0, w01 = sx.w1 * sy.w0, w10 = sx.w0 * sy.w1, w11 = sx.w1 * sy.w1;
        prod.x *= v00.x * w00 + v01.x * w01 + v10.x * w10 + v11.x * w11;
        prod.y *= v00.y * w00 + v01.y * w01 + v10.y * w10 + v11.y * w11;
        prod.z *= v00.z * w00 + v01.z * w01 + v10.z * w10 + v11.z * w11;
        prod.w *= v00.w * w00 + v01.w * w01 + v10.w * w10 + v11.w * w11;
    }
    return prod;
}

__device__ __forceinline__ void load_query(const fdgs_deform_params& p, const AabbScale& sc, int n, float* q, float* xyz) {
    xyz[0] = p.xyz[3 * (size_t)n]; xyz[1] = p.xyz[3 * (size_t)n + 1]; xyz[2] = p.xyz[3 * (size_t)n + 2];
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = (xyz[i] - p.aabb[i]) * sc.inv2[i] - 1.0f;
    q[3] = p.time ? p.time[n] : p.time_scalar;
}

// Two adjacent chunks j0, j0+1 of ONE level (C >= 16: their channels are the two halves of the same 32 texel bytes): the
// four axis samples and the texel offsets are computed once and all 48 texel requests are issued before the first one is
// consumed -- one memory round trip for the pair.  (Chunk by chunk the four round trips of a 32-feature gather were 12 %
// of D1 in its in-kernel cycle profile.)  Same arithmetic per chunk as gather_chunk.
__device__ __forceinline__ void gather_chunk_pair(const fdgs_deform_params& p, int j0, int h, const float* q, float4& out0, float4& out1) {
    const int lvl = __builtin_amdgcn_readfirstlane((8 * j0) / p.C);
    const int c0 = 8 * j0 + 4 * h - lvl * p.C;
    AxisSample S[4];
#pragma unroll
    for (int ax = 0; ax < 4; ax++) S[ax] = axis_sample(q[ax], p.res[lvl][ax]);
    float4 v[6][4], u[6][4];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        int a, b;
        plane_axes(k, a, b);
        const int Wd = p.res[lvl][a];
        const AxisSample sx = S[a], sy = S[b];
        const char* P = reinterpret_cast<const char*>(p.planes[lvl][k]);
        const uint32_t texel = (uint32_t)p.C * 4u, cb = (uint32_t)c0 * 4u;
        const uint32_t r0 = (uint32_t)(sy.i0 * Wd) * texel + cb, r1 = (uint32_t)(sy.i1 * Wd) * texel + cb;
        const uint32_t x0 = (uint32_t)sx.i0 * texel, x1 = (uint32_t)sx.i1 * texel;
        v[k][0] = *reinterpret_cast<const float4*>(P + (r0 + x0)); u[k][0] = *reinterpret_cast<const float4*>(P + (r0 + x0 + 32u));
        v[k][1] = *reinterpret_cast<const float4*>(P + (r0 + x1)); u[k][1] = *reinterpret_cast<const float4*>(P + (r0 + x1 + 32u));
        v[k][2] = *reinterpret_cast<const float4*>(P + (r1 + x0)); u[k][2] = *reinterpret_cast<const float4*>(P + (r1 + x0 + 32u));
        v[k][3] = *reinterpret_cast<const float4*>(P + (r1 + x1)); u[k][3] = *reinterpret_cast<const float4*>(P + (r1 + x1 + 32u));
    }
    __builtin_amdgcn_sched_barrier(0);
    out0 = make_float4(1.f, 1.f, 1.f, 1.f); out1 = out0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        int a, b;
        plane_axes(k, a, b);
        const AxisSample sx = S[a], sy = S[b];
        const float w00 = sx.w0 * sy.w0, w01 = sx.w1 * sy.w0, w10 = sx.w0 * sy.w1, w11 = sx.w1 * sy.w1;
        out0.x *= v[k][0].x * w00 + v[k][1].x * w01 + v[k][2].x * w10 + v[k][3].x * w11;
        out0.y *= v[k][0].y * w00 + v[k][1].y * w01 + v[k][2].y * w10 + v[k][3].y * w11;
        out0.z *= v[k][0].z * w00 + v[k][1].z * w01 + v[k][2].z * w10 + v[k][3].z * w11;
        out0.w *= v[k][0].w * w00 + v[k][1].w * w01 + v[k][2].w * w10 + v[k][3].w * w11;
        out1.x *= u[k][0].x * w00 + u[k][1].x * w01 + u[k][2].x * w10 + u[k][3].x * w11;
        out1.y *= u[k][0].y * w00 + u[k][1].y * w01 + u[k][2].y * w10 + u[k][3].y * w11;
        out1.z *= u[k][0].z * w00 + u[k][1].z * w01 + u[k][2].z * w10 + u[k][3].z * w11;
        out1.w *= u[k][0].w * w00 + u[k][1].w * w01 + u[k][2].w * w10 + u[k][3].w * w11;
    }
}

// features of lane (g,h): chunk j holds features 8j+4h .. +3 = registers 4(j%4)..+3 of tile j/4
template <int FCH>
__device__ __forceinline__ void gather_features(const fdgs_deform_params& p, const float* q, int h, f32x16* feat) {
    auto put = [&](int j, const float4& v) {
        feat[j / 4][4 * (j % 4) + 0] = v.x; feat[j / 4][4 * (j % 4) + 1] = v.y;
        feat[j / 4][4 * (j % 4) + 2] = v.z; feat[j / 4][4 * (j % 4) + 3] = v.w;
    };
#pragma unroll
    for (int j = 0; j < FCH; j += 2) {
        if (j + 1 < FCH && (8 * j) / p.C == (8 * (j + 1)) / p.C) {   // (wave-uniform) both chunks in one level
            float4 v0, v1;
            gather_chunk_pair(p, j, h, q, v0, v1);
            put(j, v0); put(j + 1, v1);
        } else {
            put(j, gather_chunk(p, j, h, q));
            if (j + 1 < FCH) put(j + 1, gather_chunk(p, j + 1, h, q));
        }
    }
}

// ------------------------------------------------------------------------------------------------ MFMA layers
// Register layouts.  rho(r,h) = row of the 32x32 MFMA C/D tile that register r holds in lane half h.
//   * "chunk" layout (HexPlane features, F = 8*FCH): tile j/4, registers 4(j%4)..+3 of lane (g,h) hold features
//     8j+4h..+3 of Gaussian g  (== feature 32*tile + rho(r,h));
//   * "interleaved" layout (hidden activations, T = W/32 tiles): tile t, register r of lane (g,h) holds feature
//     T*rho(r,h) + t.  With it BOTH products read the torch-layout weights with one 16-byte load per lane:
//       Y = W X   : A-lane (row, h) needs W[row][T*rho(r,h) + t], t = 0..T-1  -> one vector load per r feeds T MFMAs;
//       dX = W^T dY: A-lane (i, h) needs W[f][T*i + xt],       xt = 0..T-1 -> one vector load per k-step feeds T MFMAs
//     (the transposed product with the naive 32t+row layout needs T separate dword loads per k-step).
// Every A operand is software-prefetched PD steps ahead (the compiler serialises load -> wait -> 4 MFMA otherwise:
// round-1 profile, 52 % / 31 % MFMA utilisation in D1 / D2).
__device__ __forceinline__ constexpr int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int VW>
struct AVec { float v[VW]; };
template <int VW>
__device__ __forceinline__ AVec<VW> ldv(const float* __restrict__ p) {
    AVec<VW> a;
    if constexpr (VW == 4) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        a.v[0] = q.x; a.v[1] = q.y; a.v[2] = q.z; a.v[3] = q.w;
    } else if constexpr (VW == 2) {
        const float2 q = *reinterpret_cast<const float2*>(p);
        a.v[0] = q.x; a.v[1] = q.y;
    } else {
        a.v[0] = *p;
    }
    return a;
}

// Y[ot] = bias + W X, X in interleaved layout (KT tiles, K = 32*KT), W row-major [out_dim][ld].
// Output rows: ROW_IL ? interleaved (row = OT*i + ot) : standard (row = 32*ot + i), rows >= out_dim are duplicates
// of the last valid row (never read back).  Usage: setup(); preload(); ...; run().
template <int KT, int OT, bool ROW_IL, int PD, bool CLAMP = true>
struct DenseIL {
    const float* rp[OT];
    const float* bp[OT];
    float bv[OT];
    AVec<KT> buf[PD][OT];
    // The bias enters as one extra MFMA k-step (A = bias[row] in the k = 0 half, 0 in the k = 1 half; B = 1): its four
    // dword loads ride with the weight prefetch instead of stalling the first MFMA of the layer on 64 bias loads.
    __device__ __forceinline__ void setup(const float* __restrict__ Wm, const float* __restrict__ bias, int ld, int out_dim, int g, int h) {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            int row = ROW_IL ? OT * g + ot : 32 * ot + g;
            if (CLAMP) row = row < out_dim ? row : out_dim - 1;
            rp[ot] = Wm + (size_t)row * ld + KT * 4 * h;
            bp[ot] = bias + row;
        }
    }
    __device__ __forceinline__ void fetch(int s, AVec<KT>* dst) const {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) dst[ot] = ldv<KT>(rp[ot] + KT * rho(s, 0));
    }
    __device__ __forceinline__ void preload() {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) bv[ot] = *bp[ot];
#pragma unroll
        for (int s = 0; s < PD; s++) fetch(s, buf[s]);
    }
    // ---- k <= 4 output rows (position / scale / rotation / opacity heads): instead of padding the 3..4 rows to a 32-row
    // MFMA tile (64 MFMAs of 64 cycles at 9-12 % use) the product runs on v_mfma_f32_4x4x1_16b: block b = lane/4 holds
    // four Gaussians (B = the interleaved activation register as it is), A-lane 4b+i = W[i][feature of this half];
    // 64 instructions of 8 cycles.  The two lane halves hold partial sums over their halves of the features.
    __device__ __forceinline__ void setup4(const float* __restrict__ Wm, const float* __restrict__ bias, int ld, int out_dim, int g, int h) {
        static_assert(OT == 1, "small-output form has one output tile");
        int row = g & 3;
        row = row < out_dim ? row : out_dim - 1;
        rp[0] = Wm + (size_t)row * ld + KT * 4 * h;
        bp[0] = bias + row;
    }
    // returns out[i] (i < 4) of the lane's Gaussian in .x .y .z .w, valid in every lane
    __device__ __forceinline__ f32x4 run4(const f32x16* X) {
        f32x4 acc[KT];
#pragma unroll
        for (int t = 0; t < KT; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const AVec<KT> cur = buf[s % PD][0];
            if (s + PD < 16) fetch(s + PD, buf[s % PD]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < KT; t++) acc[t] = mfma4(cur.v[t], X[t][s], acc[t]);
        }
        f32x4 sum = acc[0];
#pragma unroll
        for (int t = 1; t < KT; t++) sum += acc[t];
        // lane 4b+j register i = partial out[i] of Gaussian (4b+j)&31 over this half's features; bias of row i sits in the
        // lanes with (lane & 3) == i
        f32x4 out;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float tot = sum[i] + __shfl_xor(sum[i], 32, 64);
            out[i] = tot + __shfl(bv[0], i, 4);
        }
        return out;
    }
    struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
    __device__ __forceinline__ void run(const f32x16* X, f32x16* Y, int h) { run(X, Y, h, NoHook()); }
    // `hook(s)` is issued once per k-walk step, in the shadow of that step's MFMAs (used to drain a staged tile)
    template <class Hook>
    __device__ __forceinline__ void run(const f32x16* X, f32x16* Y, int h, Hook hook) {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) Y[ot] = mfma32(h == 0 ? bv[ot] : 0.f, 1.0f, zero16());
#pragma unroll
        for (int s = 0; s < 16; s++) {
            AVec<KT> cur[OT];
#pragma unroll
            for (int ot = 0; ot < OT; ot++) cur[ot] = buf[s % PD][ot];
            if (s + PD < 16) fetch(s + PD, buf[s % PD]);
            hook(s);
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch PD steps ahead (the scheduler sinks it to its use otherwise)
#pragma unroll
            for (int t = 0; t < KT; t++)
#pragma unroll
                for (int ot = 0; ot < OT; ot++) Y[ot] = mfma32(cur[ot].v[t], X[t][s], Y[ot]);
        }
    }
};

// hid[ot] = b0 + W0 feat: feat in chunk layout (FCH chunks of 8 features), output rows interleaved (row = OT*i + ot)
template <int FCH, int OT, int PD>
struct DenseTrunk {
    const float* rp[OT];
    const float* bp;
    float bv[OT];
    float4 buf[PD][OT];
    __device__ __forceinline__ void setup(const float* __restrict__ Wm, const float* __restrict__ bias, int ld, int g, int h) {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) rp[ot] = Wm + (size_t)(OT * g + ot) * ld + 4 * h;
        bp = bias + OT * g;
    }
    __device__ __forceinline__ void fetch(int j, float4* dst) const {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) dst[ot] = *reinterpret_cast<const float4*>(rp[ot] + 8 * j);
    }
    __device__ __forceinline__ void preload() {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) bv[ot] = bp[ot];
#pragma unroll
        for (int s = 0; s < PD; s++) if (s < FCH) fetch(s, buf[s]);
    }
    __device__ __forceinline__ void run(const f32x16* feat, f32x16* Y, int h) {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) Y[ot] = mfma32(h == 0 ? bv[ot] : 0.f, 1.0f, zero16());
#pragma unroll
        for (int j = 0; j < FCH; j++) {
            float4 cur[OT];
#pragma unroll
            for (int ot = 0; ot < OT; ot++) cur[ot] = buf[j % PD][ot];
            if (j + PD < FCH) fetch(j + PD, buf[j % PD]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ot = 0; ot < OT; ot++) Y[ot] = mfma32(cur[ot].x, feat[j / 4][4 * (j % 4) + 0], Y[ot]);
#pragma unroll
            for (int ot = 0; ot < OT; ot++) Y[ot] = mfma32(cur[ot].y, feat[j / 4][4 * (j % 4) + 1], Y[ot]);
#pragma unroll
            for (int ot = 0; ot < OT; ot++) Y[ot] = mfma32(cur[ot].z, feat[j / 4][4 * (j % 4) + 2], Y[ot]);
#pragma unroll
            for (int ot = 0; ot < OT; ot++) Y[ot] = mfma32(cur[ot].w, feat[j / 4][4 * (j % 4) + 3], Y[ot]);
        }
    }
};

// dX[xt] += W^T dY: dY interleaved (YT tiles, feature f = YT*rho(r,h) + t), W row-major [32*YT][ld].
// COL_IL: dX interleaved with XT tiles (column XT*i + xt, one vector load per k-step);
// else:   dX in tile layout (column 32*xt + i, clamped to in_valid-1; rows beyond are never read back).
template <int YT, int XT, bool COL_IL, int PD>
struct DenseT {
    static constexpr int VW = COL_IL ? XT : 1;
    static constexpr int NL = COL_IL ? 1 : XT;   // loads per k-step
    const float* cp[NL];
    int ld;
    AVec<VW> buf[PD][NL];
    __device__ __forceinline__ void setup(const float* __restrict__ Wm, int ld_, int in_valid, int g, int h) {
        ld = ld_;
#pragma unroll
        for (int x = 0; x < NL; x++) {
            int col = COL_IL ? XT * g : 32 * x + g;
            col = col < in_valid ? col : in_valid - 1;
            cp[x] = Wm + (size_t)(YT * 4 * h) * ld_ + col;
        }
    }
    // k-step s = (r, t): r = s / YT, t = s % YT  ->  weight row YT*rho(r,0) + t (+ YT*4*h folded into cp)
    __device__ __forceinline__ void fetch(int s, AVec<VW>* dst) const {
        const int r = s / YT, t = s % YT;
#pragma unroll
        for (int x = 0; x < NL; x++) dst[x] = ldv<VW>(cp[x] + (size_t)(YT * rho(r, 0) + t) * ld);
    }
    __device__ __forceinline__ void preload() {
#pragma unroll
        for (int s = 0; s < PD; s++) fetch(s, buf[s]);
    }
    __device__ __forceinline__ void run(const f32x16* dY, f32x16* dX) {
#pragma unroll
        for (int s = 0; s < 16 * YT; s++) {
            AVec<VW> cur[NL];
#pragma unroll
            for (int x = 0; x < NL; x++) cur[x] = buf[s % PD][x];
            if (s + PD < 16 * YT) fetch(s + PD, buf[s % PD]);
            __builtin_amdgcn_sched_barrier(0);
            const float b = dY[s % YT][s / YT];
#pragma unroll
            for (int xt = 0; xt < XT; xt++) dX[xt] = mfma32(COL_IL ? cur[0].v[xt] : cur[xt].v[0], b, dX[xt]);
        }
    }
};

template <int T>
__device__ __forceinline__ void relu_inplace(f32x16* x) {
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) x[t][r] = fmaxf(x[t][r], 0.f);
}

// interleaved activations -> row-major [n][W] global rows: register r of the T tiles = T consecutive features
template <int T>
__device__ __forceinline__ void store_il(float* __restrict__ rowp, const f32x16* x, int h) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if constexpr (T == 4) *reinterpret_cast<float4*>(rowp + 4 * rho(r, h)) = make_float4(x[0][r], x[1][r], x[2][r], x[3][r]);
        else *reinterpret_cast<float2*>(rowp + 2 * rho(r, h)) = make_float2(x[0][r], x[1][r]);
    }
}

// [32 gaussians][W] tile of interleaved activations -> 32 contiguous global rows, through a padded per-wave LDS tile:
// the lanes park "their" Gaussian's row (store_il layout), then the wave copies the 32*W floats out lane-consecutively
// (1 KB per store instruction).  Direct store_il to global writes 16-byte pieces at 512-byte strides: measured +0.19 ms on
// the forward for the 960 MB of saved activations.
template <int T>
__device__ __forceinline__ void store_tile_coalesced(float* lds_tile, float* __restrict__ gdst, const f32x16* x, int g, int h, int lane) {
    constexpr int W = 32 * T, STRIDE = W + 4;
    __builtin_amdgcn_wave_barrier();
    store_il<T>(lds_tile + g * STRIDE, x, h);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < T * 4; j++) {
        const int e4 = j * 64 + lane, row = e4 / (W / 4), c4 = e4 - row * (W / 4);
        reinterpret_cast<float4*>(gdst)[e4] = *reinterpret_cast<const float4*>(lds_tile + row * STRIDE + 4 * c4);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ int next_head(const int* head_on, int hd) {
    hd++;
    while (hd < FDGS_NUM_HEADS && !head_on[hd]) hd++;
    return hd;
}

__device__ __forceinline__ int next_head_m(unsigned mask, int hd) {
    hd++;
    while (hd < FDGS_NUM_HEADS && !((mask >> hd) & 1u)) hd++;
    return hd;
}

// ------------------------------------------------------------------------------------------------ D1 forward
#ifndef FDGS_D1_PD1
#define FDGS_D1_PD1 2
#endif
template <int WT>
struct FwdPD { static constexpr int L1 = WT == 4 ? 2 : 4, L2 = 8; };

template <int WT, int FCH>
__global__ void __launch_bounds__(256, 1) deform_fwd_kernel(DeformDev d) {
    constexpr int PD1 = WT == 4 ? FDGS_D1_PD1 : FwdPD<WT>::L1, PD2 = FwdPD<WT>::L2;
    const fdgs_deform_params& p = d.p;
    const bool tunable_small = d.small_heads != 0;
    // LDS: the four waves' staging tiles of the saved activations + the second-layer weights of all heads (59 rows, padded row
    // stride: rows i = 0..3 of a 4x4x1 product and the two lane halves fall into distinct banks).  The second layers are short
    // products (64 MFMAs of 8 cycles for a k <= 4 head) whose operand ring cannot cover an L2 round trip: read from LDS they lose
    // the ~2 k cycles per head that the in-kernel cycle profile charged to "L2" beyond its MFMA time.
    constexpr int LDW = WT * 32 + 4;
    __shared__ __attribute__((aligned(16))) float fwd_lds[4 * 32 * LDW + 59 * LDW];
    float* my_tile = fwd_lds + (threadIdx.x >> 6) * 32 * LDW;
    float* w2lds = fwd_lds + 4 * 32 * LDW;
    for (int hd_ = 0; hd_ < FDGS_NUM_HEADS; hd_++) {
        if (!p.head_on[hd_]) continue;
        const int k_ = head_k(hd_), r0_ = head_row0(hd_);
        for (int i = threadIdx.x; i < k_ * (WT * 8); i += 256) {
            const int r = i / (WT * 8), c4 = i - r * (WT * 8);
            *reinterpret_cast<float4*>(w2lds + (r0_ + r) * LDW + 4 * c4) = reinterpret_cast<const float4*>(p.w2[hd_])[i];
        }
    }
    __syncthreads();
    constexpr int FT = (FCH + 3) / 4;
    const int lane = threadIdx.x & 63, g0 = lane & 31, h0 = lane >> 5;
    // Every wave walks its own tiles (32 Gaussians each): nothing in the body synchronises the workgroup, so with
    // gridDim.x = #CUs the kernel is persistent -- no workgroup relaunch between tiles and no SIMD waiting for the slowest
    // of the four waves of its workgroup; with gridDim.x = ntiles / 4 the loop runs once (FDGS_D1_WGS selects).
#ifdef FDGS_PROFILE_D1
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pt = __builtin_amdgcn_s_memtime();
#define D1_TICK(ph) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[ph] += t_ - pt; pt = t_; } while (0)
#else
#define D1_TICK(ph) do { } while (0)
#endif
    // Persistent loop: wave w takes tiles w, w + #waves, ...  The tiles left over after the last FULL round would keep a few waves busy
    // for a whole tile while the others idle (300 k Gaussians: 9 376 tiles on 1 024 waves = 9 full rounds + 160 tiles, 8.4 % of the
    // kernel).  Where they fit, those tiles are split BY HEAD instead: wave u takes head u % nh of tile u / nh -- every such wave repeats
    // the gather and the trunk (cheap) and evaluates one head, so the last round lasts about a third of a tile.  The first wave of a tile
    // ("primary") also writes what is per tile rather than per head: saved features / trunk activations, outputs of switched-off heads.
    unsigned all_heads = 0u;
    int nh = 0;
#pragma unroll
    for (int i = 0; i < FDGS_NUM_HEADS; i++) if (p.head_on[i]) { all_heads |= 1u << i; nh++; }
    const int nwaves = (int)gridDim.x * 4, wave_id = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    const int full_rounds = d.ntiles / nwaves, rem = d.ntiles - full_rounds * nwaves;
    const bool split = d.split_tail != 0 && nh > 1 && rem > 0 && rem * nh <= nwaves;
    for (int it = 0; it <= full_rounds; it++) {
    int tile = it * nwaves + wave_id;
    unsigned head_mask = all_heads;
    bool primary = true;
    if (it == full_rounds) {
        if (split) {
            if (wave_id >= rem * nh) break;
            tile = full_rounds * nwaves + wave_id / nh;
            int ord = wave_id % nh, hsel = -1;
            for (int i = 0; i < FDGS_NUM_HEADS; i++) if (p.head_on[i] && ord-- == 0) hsel = i;
            head_mask = 1u << hsel;
            primary = wave_id % nh == 0;
        } else if (tile >= d.ntiles) {
            break;
        }
    }
    int g = g0, h = h0;
    asm volatile("" : "+v"(g), "+v"(h));   // keeps the per-layer weight addresses from being hoisted out of the tile loop
    const size_t tile_n0 = (size_t)tile * 32;       // first Gaussian slot of this wave's tile
    const int n_raw = tile * 32 + g;
    const bool live = n_raw < p.N;
    const int n = live ? n_raw : p.N - 1;
    const int W = WT * 32;
    DenseTrunk<FCH, WT, 2> T0;
    T0.setup(p.w0, p.b0, d.F, g, h);
    T0.preload();
    int hd = next_head_m(head_mask, -1);
    DenseIL<WT, WT, true, PD1, false> L1;
    if (hd < FDGS_NUM_HEADS) { L1.setup(p.w1[hd], p.b1[hd], W, W, g, h); L1.preload(); }
    float q[4], xyz[3];
    load_query(p, d.sc, n, q, xyz);
    // every per-Gaussian input of the epilogues is fetched now (one HBM round trip under the gather) instead of once per
    // head behind its last MFMA
    float in_sc[3], in_op, in_sh[24];
#pragma unroll
    for (int i = 0; i < 3; i++) in_sc[i] = p.scales[3 * (size_t)n + i];
    const float4 in_rot = reinterpret_cast<const float4*>(p.rotations)[n];
    in_op = p.opacity[n];
#pragma unroll
    for (int u = 0; u < 6; u++) {
        const int row0 = (u < 4 ? 0 : 32) + 8 * (u & 3) + 4 * h;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = row0 + i;
            in_sh[4 * u + i] = m < 3 ? p.shs_dc[(size_t)p.shs_dc_stride * n + m] : p.shs_rest[(size_t)p.shs_rest_stride * n + (m - 3)];
        }
    }
    f32x16 feat[FT];
#pragma unroll
    for (int t = 0; t < FT; t++) feat[t] = zero16();
    D1_TICK(0);
    gather_features<FCH>(p, q, h, feat);
    D1_TICK(1);
    const size_t n_row = (size_t)n_raw;   // saved rows are indexed by the un-clamped Gaussian slot (< Npad)
    if (d.sv_feat && primary) {
#pragma unroll
        for (int j = 0; j < FCH; j++)
            *reinterpret_cast<float4*>(d.sv_feat + n_row * d.F + 8 * j + 4 * h) =
                make_float4(feat[j / 4][4 * (j % 4)], feat[j / 4][4 * (j % 4) + 1], feat[j / 4][4 * (j % 4) + 2], feat[j / 4][4 * (j % 4) + 3]);
    }
    f32x16 hid[WT];
    T0.run(feat, hid, h);
    relu_inplace<WT>(hid);  // every consumer of the trunk output starts with ReLU (scene/deformation.py:61-65)
    // saved activations leave through the LDS tile: parked right after they are computed, copied out (lane-consecutive,
    // 1 KB per store) one 1-KB piece per k-walk step of the NEXT hidden layer, i.e. in the shadow of its MFMAs
    float* pending_dst = nullptr;
    constexpr int TSTRIDE = WT * 32 + 4;
    auto park = [&](const f32x16* x, float* dst) {
        store_il<WT>(my_tile + g * TSTRIDE, x, h);
        pending_dst = dst;
    };
    auto drain_piece = [&](int j) {     // pieces j = 0 .. 4*WT-1 of 64 float4 each
        if (pending_dst && j < WT * 4) {
            const int e4 = j * 64 + lane, row = e4 / (W / 4), c4 = e4 - row * (W / 4);
            reinterpret_cast<float4*>(pending_dst)[e4] = *reinterpret_cast<const float4*>(my_tile + row * TSTRIDE + 4 * c4);
        }
    };
    if (d.sv_rh && primary) park(hid, d.sv_rh + tile_n0 * W);
    if (d.sv_hmask && primary) {   // the backward's ReLU mask of the trunk output, in its own lane layout: one 16-byte load there
        uint32_t m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < WT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) m[t] |= (hid[t][r] > 0.f ? 1u : 0u) << r;
        reinterpret_cast<uint4*>(d.sv_hmask)[(size_t)(tile_n0 / 32) * 64 + lane] = make_uint4(m[0], m[1], m[2], m[3]);
    }

    const bool writer = live && h == 0;
    // epilogue of head hd applied to the head's output delta (zero for a switched-off head: it returns its input
    // unchanged, scene/deformation.py:106-146)
    auto epilogue = [&](int hd_, const f32x16& o0, const f32x16& o1) {
        if (hd_ == FDGS_HEAD_POS) {
            if (writer) {
                d.out.xyz[3 * (size_t)n] = xyz[0] + o0[0]; d.out.xyz[3 * (size_t)n + 1] = xyz[1] + o0[1];
                d.out.xyz[3 * (size_t)n + 2] = xyz[2] + o0[2];
            }
        } else if (hd_ == FDGS_HEAD_SCALE) {
            if (writer) {
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const float v = in_sc[i] + o0[i];
                    d.out.scales[3 * (size_t)n + i] = p.activate ? __expf(v) : v;
                }
            }
        } else if (hd_ == FDGS_HEAD_ROT) {
            if (writer) {
                float v0 = in_rot.x + o0[0], v1 = in_rot.y + o0[1], v2 = in_rot.z + o0[2], v3 = in_rot.w + o0[3];
                if (p.activate) {
                    const float nrm = sqrtf(v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3);
                    const float inv = 1.0f / fmaxf(nrm, 1e-12f);  // F.normalize eps (scene/gaussian_model.py:44)
                    v0 *= inv; v1 *= inv; v2 *= inv; v3 *= inv;
                    if (d.out.rot_norm) d.out.rot_norm[n] = nrm;
                }
                reinterpret_cast<float4*>(d.out.rotations)[n] = make_float4(v0, v1, v2, v3);
            }
        } else if (hd_ == FDGS_HEAD_OPACITY) {
            if (writer) {
                const float v = in_op + o0[0];
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
                    for (int i = 0; i < 4; i++) v[i] = in_sh[4 * u + i] + (u < 4 ? o0[4 * (u & 3) + i] : o1[4 * (u & 3) + i]);
                    *reinterpret_cast<float4*>(d.out.shs + 48 * (size_t)n + row0) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    };
    {
        const f32x16 z = zero16();
        for (int h0 = 0; h0 < FDGS_NUM_HEADS; h0++)
            if (!p.head_on[h0] && primary) epilogue(h0, z, z);
    }

    D1_TICK(2);
    while (hd < FDGS_NUM_HEADS) {
        const int k = head_k(hd);
        DenseIL<WT, 1, false, PD2> L2, L2b;
        const bool small = k <= 4 && tunable_small;
        const float* w2h = w2lds + head_row0(hd) * LDW;
        if (small) L2.setup4(w2h, p.b2[hd], LDW, k, g, h);
        else L2.setup(w2h, p.b2[hd], LDW, k < 32 ? k : 32, g, h);
        L2.preload();
        f32x16 h1[WT];
        L1.run(hid, h1, h, drain_piece);
        D1_TICK(3);
        relu_inplace<WT>(h1);
        if (d.sv_h1) park(h1, d.sv_h1 + ((size_t)d.head_slot[hd] * d.Npad + tile_n0) * W);
        if (k > 32) { L2b.setup(w2h + 32 * LDW, p.b2[hd] + 32, LDW, k - 32, g, h); L2b.preload(); }
        const int nxt = next_head_m(head_mask, hd);
        if (nxt < FDGS_NUM_HEADS) { L1.setup(p.w1[nxt], p.b1[nxt], W, W, g, h); L1.preload(); }
        f32x16 o0 = zero16(), o1 = zero16();
        D1_TICK(4);
        if (small) {
            const f32x4 o4 = L2.run4(h1);
            o0[0] = o4[0]; o0[1] = o4[1]; o0[2] = o4[2]; o0[3] = o4[3];
        } else {
            L2.run(h1, &o0, h);
        }
        if (k > 32) L2b.run(h1, &o1, h);
        D1_TICK(5);
        epilogue(hd, o0, o1);
        D1_TICK(6);
        hd = nxt;
    }
    // the last parked tile has no following layer to hide under
#pragma unroll
    for (int j = 0; j < WT * 4; j++) drain_piece(j);
    D1_TICK(7);
    }   // tile loop
#ifdef FDGS_PROFILE_D1
    if (d.prof && lane == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(&d.prof[i], pacc[i]);
        atomicAdd(&d.prof[8], 1ull);
    }
#endif
}

#include "deform_fwd16.h"
#include "deform_fwd_ws.h"

// ------------------------------------------------------------------------------------------------ backward: prep
// Per Gaussian: activation Jacobians -> packed pre-activation output gradients G[n][64]; direct (identity) paths.
struct PrepArgs {
    int N, Npad, activate, dc_stride, rest_stride;
    const float *g_xyz, *g_scales, *g_rot, *g_opacity, *g_shs, *out_scales, *out_rot, *out_opacity, *rot_norm;
    float *d_xyz, *d_scales, *d_rot, *d_opacity, *d_shs_dc, *d_shs_rest;
    float* G;
    uint32_t* tile_live;   // [Npad/32]: bit r set when packed row r of the 32-row tile is non-zero
};
// One wave per 64 consecutive Gaussians.  Every array is written as one contiguous block per wave (G: 16 KB, d_shs:
// 12 KB, d_xyz: 768 B ...) by staging the per-Gaussian rows in LDS and walking the block linearly, lane-consecutive:
// the round-1 kernel wrote 4..16-byte pieces at 12..256-byte strides and rocprofv3 showed 2.1x write and 2.4x fetch
// amplification on it (profiles/r01c_pmc_*).
__global__ void __launch_bounds__(256) deform_bwd_prep_kernel(PrepArgs a) {
    __shared__ __attribute__((aligned(16))) float lds_all[4 * 64 * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* small = lds_all + wave * 64 * 64;   // [64][16]: the 11 pre-activation gradients of the k<=4 heads (+ padding)
    float* sh = small + 64 * 16;               // [64][48]: g_shs rows
    const int n0 = (blockIdx.x * 4 + wave) * 64;
    if (n0 >= a.Npad) return;
    const int n = n0 + lane;
    const int nvalid = a.N - n0 < 64 ? (a.N - n0 > 0 ? a.N - n0 : 0) : 64;   // Gaussians of this wave that exist
    float row[16];
#pragma unroll
    for (int i = 0; i < 16; i++) row[i] = 0.f;
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
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
        reinterpret_cast<float4*>(small)[lane * 4 + i] = make_float4(row[4 * i], row[4 * i + 1], row[4 * i + 2], row[4 * i + 3]);
    // g_shs block of this wave: [64][48] floats = 768 float4, contiguous in memory
    const float4* gsh4 = a.g_shs ? reinterpret_cast<const float4*>(a.g_shs + (size_t)n0 * 48) : nullptr;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const int v = j * 64 + lane;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gsh4 && v < nvalid * 12) x = gsh4[v];
        reinterpret_cast<float4*>(sh)[v] = x;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- per 32-row tile: does any row carry a gradient?  (the backward kernels skip tiles of all-zero rows)
    {
        bool nz = false;
#pragma unroll
        for (int i = 0; i < 11; i++) nz = nz || (row[i] != 0.f);
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const float4 x = reinterpret_cast<const float4*>(sh)[lane * 12 + j];
            nz = nz || x.x != 0.f || x.y != 0.f || x.z != 0.f || x.w != 0.f;
        }
        const unsigned long long m = __ballot(nz);
        if (lane == 0) {
            a.tile_live[n0 >> 5] = (uint32_t)(m & 0xffffffffull);       // (bit r = row r of the tile is non-zero: the row lists are built from these)
            a.tile_live[(n0 >> 5) + 1] = (uint32_t)(m >> 32);
        }
    }
    // ---- packed gradient rows G[n][64] = [small 16 | shs 48] (padded rows n >= N are zero)
    {
        float4* G4 = reinterpret_cast<float4*>(a.G + (size_t)n0 * GCOLS);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int v = j * 64 + lane, r = v >> 4, c4 = v & 15;
            G4[v] = c4 < 4 ? reinterpret_cast<const float4*>(small)[r * 4 + c4] : reinterpret_cast<const float4*>(sh)[r * 12 + (c4 - 4)];
        }
    }
    // ---- identity paths (out = in + delta): accumulate into the parameter gradients, block-linear
    if (a.d_shs_dc && a.d_shs_rest && a.dc_stride == 48 && a.rest_stride == 48 && a.d_shs_rest == a.d_shs_dc + 3) {
        float4* d4 = reinterpret_cast<float4*>(a.d_shs_dc + (size_t)n0 * 48);   // one combined [N,16,3] tensor
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const int v = j * 64 + lane;
            if (v < nvalid * 12) {
                float4 x = d4[v];
                const float4 y = reinterpret_cast<const float4*>(sh)[v];
                x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
                d4[v] = x;
            }
        }
    } else {
        if (a.d_shs_dc) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int idx = j * 64 + lane, r = idx / 3, c = idx - 3 * r;
                if (r < nvalid) a.d_shs_dc[(size_t)(n0 + r) * a.dc_stride + c] += sh[r * 48 + c];
            }
        }
        if (a.d_shs_rest) {
            for (int j = 0; j < 45; j++) {
                const int idx = j * 64 + lane, r = idx / 45, c = idx - 45 * r;
                if (r < nvalid) a.d_shs_rest[(size_t)(n0 + r) * a.rest_stride + c] += sh[r * 48 + 3 + c];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int idx = j * 64 + lane, r = idx / 3, c = idx - 3 * r;
        if (r < nvalid) {
            if (a.d_xyz) a.d_xyz[(size_t)n0 * 3 + idx] += small[r * 16 + c];
            if (a.d_scales) a.d_scales[(size_t)n0 * 3 + idx] += small[r * 16 + 3 + c];
        }
    }
    if (a.d_rot) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int idx = j * 64 + lane, r = idx >> 2, c = idx & 3;
            if (r < nvalid) a.d_rot[(size_t)n0 * 4 + idx] += small[r * 16 + 6 + c];
        }
    }
    if (a.d_opacity && lane < nvalid) a.d_opacity[n] += small[lane * 16 + 10];
}

// ------------------------------------------------------------------------------------------------ live-tile lists
// Lists written by an EARLIER kernel are read through the constant address space: a wave-uniform index then gives a scalar
// (s_load) read.  Through a plain global pointer the compiler must assume the kernel's own stores may alias the list and falls back
// to a vector load + readfirstlane -- which joins the in-order vmcnt queue behind the prefetched activation rows and drags the
// derived addresses into vector registers.
typedef const uint32_t __attribute__((address_space(4))) * const_u32p;
__device__ __forceinline__ const_u32p as_const(const uint32_t* p) { return (const_u32p)(unsigned long long)p; }

// The rasterizer hands zero gradient rows to every Gaussian that is culled, off-screen or fully occluded (on the bench scene:
// 88 % of them, profiles/r03a_zero_gradient_rows.jsonl), and a zero row adds exactly zero to every sum the backward forms.
// With the set in spatial order such Gaussians are contiguous, so whole 32-row tiles are zero: the packing stage
// (deform_bwd_prep_kernel, or fdgs_raster_bwd's epilogue) leaves one flag per tile behind, this kernel turns the flags into
//   live[]   : ascending indices of the tiles with a non-zero row, padded to a multiple of 4 with a zero tile (D2's workgroups take
//              four tiles at a time and meet at barriers),
//   chunks[] : ascending indices of the plane-gradient chunks (tpc tiles each) that contain a live tile,
//   counters : { live tiles, live tiles padded, live chunks, tiles },
// and D2 / D3 / D4 walk the lists instead of 0 .. Npad/32.  One workgroup; ~5 us.  skip = 0 lists every tile (A/B, FDGS_SKIP_DEAD=0).
struct CompactArgs {
    uint32_t* flags; uint32_t* live; uint32_t* chunks; uint32_t* counters;    // (skip = 0: flags are WRITTEN here, all ones)
    int ntiles, tpc, skip;
    float* G;      // packed rows: the dead tile used as padding of live[] gets zero rows here (its producer may have left them unwritten)
    // ROW lists (fdgs_tuning "row_compact", saved activations + spatially ordered input): rowbase[t] = live rows in front of tile t;
    // row_gather_kernel then lists the live rows (ascending) in rows[] and copies their packed gradient rows, in that order, to Gc;
    // this kernel pads both to a multiple of row_pad rows (pad entries: ROW_PAD | 0, zero gradient rows).  NULL: tile lists only.
    uint32_t* rowbase; uint32_t* rows; float* Gc;
    int row_pad;
};
constexpr uint32_t ROW_PAD = 0x80000000u;       // rows[] entry: padding (index bits = a valid row to read activations from, here 0)
__global__ void __launch_bounds__(1024) tile_compact_kernel(CompactArgs a) {
    __shared__ uint32_t wl[16], wc[16], wr[16];
    __shared__ uint32_t first_dead;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) first_dead = 0xffffffffu;
    __syncthreads();
    int span = (a.ntiles + 1023) / 1024;
    span = (span + 3) & ~3;                              // (ntiles and span are multiples of 4: aligned uint4 reads, whole chunks)
    const int b = t * span, e = b + span < a.ntiles ? b + span : a.ntiles;
    uint32_t nl = 0, nc = 0, nr = 0, fd = 0xffffffffu;
    for (int i = b; i < e; i += 4) {
        uint4 f = make_uint4(1u, 1u, 1u, 1u);
        if (a.skip) f = reinterpret_cast<const uint4*>(a.flags)[i >> 2];
        // every tile is walked: D2 writes every tile's DFEAT rows, and the plane-gradient kernels (which mask DFEAT rows with these flags)
        // must see them all -- the flags may never have been written (packed_rows_ready = 1) or mark zero rows (harmless either way)
        else reinterpret_cast<uint4*>(a.flags)[i >> 2] = f;
        const uint32_t fv[4] = {f.x != 0u, f.y != 0u, f.z != 0u, f.w != 0u};
        nr += (uint32_t)(__popc(f.x) + __popc(f.y) + __popc(f.z) + __popc(f.w));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            nl += fv[j];
            if (!fv[j] && fd == 0xffffffffu) fd = (uint32_t)(i + j);
        }
        if (a.tpc == 4) nc += (fv[0] | fv[1] | fv[2] | fv[3]);
        else if (a.tpc == 2) nc += (fv[0] | fv[1]) + (fv[2] | fv[3]);
        else nc += fv[0] + fv[1] + fv[2] + fv[3];
    }
    if (fd != 0xffffffffu) atomicMin(&first_dead, fd);
    uint32_t il = nl, ic = nc, ir = nr;       // inclusive scans inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t ul = __shfl_up(il, o, 64), uc = __shfl_up(ic, o, 64), ur = __shfl_up(ir, o, 64);
        if (lane >= o) { il += ul; ic += uc; ir += ur; }
    }
    if (lane == 63) { wl[wv] = il; wc[wv] = ic; wr[wv] = ir; }
    __syncthreads();
    uint32_t pl = il - nl, pc = ic - nc, pr = ir - nr, tl = 0, tc = 0, tr = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        if (w < wv) { pl += wl[w]; pc += wc[w]; pr += wr[w]; }
        tl += wl[w]; tc += wc[w]; tr += wr[w];
    }
    for (int i = b; i < e; i += 4) {
        uint4 f = make_uint4(1u, 1u, 1u, 1u);
        if (a.skip) f = reinterpret_cast<const uint4*>(a.flags)[i >> 2];
        const uint32_t fv[4] = {f.x != 0u, f.y != 0u, f.z != 0u, f.w != 0u};
        if (a.rowbase) {
            const uint32_t c0 = (uint32_t)__popc(f.x), c1 = (uint32_t)__popc(f.y), c2 = (uint32_t)__popc(f.z);
            reinterpret_cast<uint4*>(a.rowbase)[i >> 2] = make_uint4(pr, pr + c0, pr + c0 + c1, pr + c0 + c1 + c2);
            pr += c0 + c1 + c2 + (uint32_t)__popc(f.w);
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (fv[j]) a.live[pl++] = (uint32_t)(i + j);
        if (a.tpc == 4) { if (fv[0] | fv[1] | fv[2] | fv[3]) a.chunks[pc++] = (uint32_t)(i >> 2); }
        else if (a.tpc == 2) { if (fv[0] | fv[1]) a.chunks[pc++] = (uint32_t)(i >> 1); if (fv[2] | fv[3]) a.chunks[pc++] = (uint32_t)((i >> 1) + 1); }
        else {
#pragma unroll
            for (int j = 0; j < 4; j++) if (fv[j]) a.chunks[pc++] = (uint32_t)(i + j);
        }
    }
    if (a.skip && a.G && (tl & 3u) != 0u && first_dead != 0xffffffffu) {
        float* rows = a.G + (size_t)first_dead * 32 * GCOLS;
        for (int k = t; k < 32 * GCOLS; k += 1024) rows[k] = 0.f;
    }
    // row lists: pad to whole units of row_pad rows (>= 128: D2's workgroups take four 32-row tiles at a time; D4 takes whole chunks)
    const uint32_t rp = a.rowbase ? (tr + (uint32_t)a.row_pad - 1u) / (uint32_t)a.row_pad * (uint32_t)a.row_pad : 0u;
    if (a.rowbase) {
        for (uint32_t k = tr + t; k < rp; k += 1024) a.rows[k] = ROW_PAD;
        float4* gz = reinterpret_cast<float4*>(a.Gc + (size_t)tr * GCOLS);
        for (uint32_t k = t; k < (rp - tr) * (GCOLS / 4); k += 1024) gz[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (t == 0) {
        const uint32_t n4 = (tl + 3u) & ~3u;
        // (tl % 4 != 0 implies a dead tile exists, because ntiles % 4 == 0)
        for (uint32_t k = tl; k < n4; k++) a.live[k] = first_dead;
        a.counters[0] = tl; a.counters[1] = n4; a.counters[2] = tc; a.counters[3] = (uint32_t)a.ntiles;
        // [4] live rows, [5] 32-row tiles of the padded row list, [6] plane-gradient chunks of it
        a.counters[4] = tr; a.counters[5] = rp / 32u; a.counters[6] = a.rowbase ? rp / (32u * (uint32_t)a.tpc) : 0u;
    }
}

// rows[] and the compact copy of the packed gradient rows.  One wave per 64 rows (two tiles); a wave without a live row returns at once.
struct RowGatherArgs { const uint32_t* flags; const uint32_t* rowbase; const float* G; uint32_t* rows; float* Gc; int ntiles; };
__global__ void __launch_bounds__(256) row_gather_kernel(RowGatherArgs a) {
    __shared__ uint8_t lst_all[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t0 = (blockIdx.x * 4 + wv) * 2;           // first tile of this wave (ntiles is a multiple of 4)
    if (t0 >= a.ntiles) return;
    const uint32_t m0 = a.flags[t0], m1 = a.flags[t0 + 1];
    if ((m0 | m1) == 0u) return;
    const uint32_t b0 = a.rowbase[t0];                  // (rowbase[t0 + 1] = b0 + popc(m0): the wave's live rows are one run of the list)
    const unsigned long long m = (unsigned long long)m0 | ((unsigned long long)m1 << 32);
    const bool on = (m >> lane) & 1ull;
    const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    uint8_t* lst = lst_all[wv];
    if (on) { a.rows[b0 + rank] = (uint32_t)(t0 * 32 + lane); lst[rank] = (uint8_t)lane; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int cnt = __popcll(m), sub = lane >> 4, c4 = lane & 15;    // four rows per pass: 16 lanes x 16 bytes each
    const float4* G4 = reinterpret_cast<const float4*>(a.G + (size_t)t0 * 32 * GCOLS);
    float4* O4 = reinterpret_cast<float4*>(a.Gc + (size_t)b0 * GCOLS);
    for (int e0 = 0; e0 < cnt; e0 += 4) {
        const int e = e0 + sub;
        if (e < cnt) O4[e * 16 + c4] = G4[(int)lst[e] * 16 + c4];
    }
}

// The same in ONE launch (no rowbase array, no tile_compact_kernel) for sets of up to ROW_LIST_MAX_TILES tiles: a workgroup owns 8 tiles
// (256 rows) and counts the live rows in front of them itself -- a sum over the row masks of the earlier tiles, at most 64 KB of
// coalesced reads, skipped by the workgroups without a live row (80 % on the bench scene) -- then lists and copies like row_gather_kernel.
// The workgroup of the last tiles also leaves the counters and the padding (what tile_compact_kernel does in the two-launch form).
constexpr int ROW_LIST_MAX_TILES = 16384;
struct RowListArgs { const uint32_t* flags; const float* G; uint32_t* rows; float* Gc; uint32_t* counters; int ntiles, tpc, row_pad; };
__global__ void __launch_bounds__(256) row_list_kernel(RowListArgs a) {
    __shared__ uint8_t lst_all[4][64];
    __shared__ uint32_t s_part[4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int tb = blockIdx.x * 8;                      // first tile of this workgroup (ntiles is a multiple of 4)
    const bool last = tb + 8 >= a.ntiles;
    const int t0 = tb + 2 * wv;                         // first tile of this wave
    const uint32_t m0 = t0 < a.ntiles ? a.flags[t0] : 0u, m1 = t0 + 1 < a.ntiles ? a.flags[t0 + 1] : 0u;
    const int wcnt = __popc(m0) + __popc(m1);
    if (!last && __syncthreads_or(wcnt) == 0) return;   // (uniform: a workgroup of dead tiles has nothing to list)
    uint32_t part = 0;
    for (int i = t; i < tb; i += 256) part += (uint32_t)__popc(a.flags[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) s_part[wv] = part;
    __syncthreads();
    const uint32_t base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    __syncthreads();
    if (lane == 0) s_part[wv] = (uint32_t)wcnt;
    __syncthreads();
    uint32_t b0 = base;
    for (int w = 0; w < wv; w++) b0 += s_part[w];
    if (wcnt) {
        const unsigned long long m = (unsigned long long)m0 | ((unsigned long long)m1 << 32);
        const bool on = (m >> lane) & 1ull;
        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        uint8_t* lst = lst_all[wv];
        if (on) { a.rows[b0 + rank] = (uint32_t)(t0 * 32 + lane); lst[rank] = (uint8_t)lane; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int sub = lane >> 4, c4 = lane & 15;      // four rows per pass: 16 lanes x 16 bytes each
        const float4* G4 = reinterpret_cast<const float4*>(a.G + (size_t)t0 * 32 * GCOLS);
        float4* O4 = reinterpret_cast<float4*>(a.Gc + (size_t)b0 * GCOLS);
        for (int e0 = 0; e0 < wcnt; e0 += 4) {
            const int e = e0 + sub;
            if (e < wcnt) O4[e * 16 + c4] = G4[(int)lst[e] * 16 + c4];
        }
    }
    if (last) {      // totals, padding to whole units of row_pad rows, counters (the tile lists are not built: nobody reads them in this form)
        const uint32_t tr = base + s_part[0] + s_part[1] + s_part[2] + s_part[3];
        const uint32_t rp = (tr + (uint32_t)a.row_pad - 1u) / (uint32_t)a.row_pad * (uint32_t)a.row_pad;
        for (uint32_t k = tr + t; k < rp; k += 256) a.rows[k] = ROW_PAD;
        float4* gz = reinterpret_cast<float4*>(a.Gc + (size_t)tr * GCOLS);
        for (uint32_t k = t; k < (rp - tr) * (GCOLS / 4); k += 256) gz[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t == 0) {
            a.counters[0] = 0u; a.counters[1] = 0u; a.counters[2] = 0u; a.counters[3] = (uint32_t)a.ntiles;
            a.counters[4] = tr; a.counters[5] = rp / 32u; a.counters[6] = rp / (32u * (uint32_t)a.tpc);
        }
    }
}

// ------------------------------------------------------------------------------------------------ D2 backward-data
struct BwdScratch {
    float *G, *DH1, *DHID, *RH, *FEAT, *DFEAT;
    uint32_t *tile_live, *live, *chunks, *counters;   // per-tile non-zero flags and the lists tile_compact_kernel builds from them
    const uint32_t* rows;                             // ROWS kernels: the live rows (ascending, padded; row_gather_kernel); G is then the compact copy
    int Npad;
};
struct BwdDev {
    fdgs_deform_params p;
    AabbScale sc;
    BwdScratch s;
    float* d_w2[FDGS_NUM_HEADS];
    float* d_b2[FDGS_NUM_HEADS];
    int F;
    int head_slot[FDGS_NUM_HEADS];  // index of the head's dH1 slab
    int ntiles;                     // 32-Gaussian tiles (Npad / 32)
    const uint32_t* sv_hmask;       // SAVED kernels: the forward's per-lane ReLU bits of the trunk output
    const float *sv_rh, *sv_h1;     // SAVED kernels: relu(hidden) [Np][W], relu(h1) [slot][Np][W] written by the forward
    int small_heads;                // 1: dW2 of the k<=4 heads on the 4x4x1 MFMA with register-resident sums (FDGS_SMALL_HEADS)
    unsigned long long* prof;       // development builds (-DFDGS_PROFILE_D2): per-wave cycle sums per phase
};
#ifdef FDGS_PROFILE_D2
#define D2_TICK(ph) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof_acc[ph] += t_ - prof_t; prof_t = t_; } while (0)
#else
#define D2_TICK(ph) do { } while (0)
#endif

// LDS of the backward kernel: per-wave transposed relu(h1) tile [32 gaussians][W + 4] (padded: conflict-free
// ds_write_b128 / ds_read_b32) + workgroup accumulators of dW2 / db2 for all five heads (59 rows), flushed to global
// memory once per (persistent) workgroup instead of once per 32 Gaussians.
template <int WT>
struct BwdLds {
    static constexpr int W = WT * 32;
    static constexpr int STRIDE = W + 4;
    static constexpr int TILE_FLOATS = 32 * STRIDE;
    static constexpr int KSUM = 59;                    // 3 + 3 + 4 + 1 + 48 output rows over the five heads
    static constexpr int ACC_W = KSUM * W;
    static constexpr int TOTAL = 4 * TILE_FLOATS + ACC_W + 64;
};
template <int NCH, int GQ>
__device__ __forceinline__ void small_dw2_steps(f32x4* acc, float sa0, float sa1, const float* lds, int stride, int lane) {
    if constexpr (GQ < 32) {
#pragma unroll
        for (int u = 0; u < NCH; u++)
            acc[u] = mfma4_bcast<(GQ & 15)>(GQ < 16 ? sa0 : sa1, lds[GQ * stride + 64 * u + lane], acc[u]);
        small_dw2_steps<NCH, GQ + 1>(acc, sa0, sa1, lds, stride, lane);
    }
}

// SAVED: the forward left features / relu(hidden) / relu(h1) behind (fdgs_deform_out::saved): no gather, no trunk, no
// recomputation of the heads' hidden layers -- the h1 tile is copied straight from memory into the (already transposed)
// LDS tile, the ReLU masks are read back from it, and `hid` never occupies registers.
// ROWS (with SAVED): the unit of work is a tile of 32 entries of the ROW LIST -- the Gaussians whose gradient row is non-zero, in ascending
// order -- instead of 32 consecutive Gaussians: G, DH1, DHID and DFEAT are indexed by list position (compact), the saved activations and
// the ReLU bits of a row are fetched through the list.  On the bench scene 12 % of the rows but 17.5 % of the 32-row tiles are live.
template <int WT, int FCH, bool SAVED, bool ROWS = false>
__global__ void __launch_bounds__(256, 1) deform_bwd_data_kernel(BwdDev d) {
    static_assert(SAVED || !ROWS, "the row-list form reads the saved activations");
    const fdgs_deform_params& p = d.p;
    constexpr int FT = (FCH + 3) / 4;
    using LD = BwdLds<WT>;
    constexpr int STRIDE = LD::STRIDE, W = WT * 32;
    __shared__ __attribute__((aligned(16))) float lds_all[LD::TOTAL];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g0 = lane & 31, h0 = lane >> 5;
    float* lds = lds_all + wave * LD::TILE_FLOATS;
    float* accW2 = lds_all + 4 * LD::TILE_FLOATS;
    float* accB2 = accW2 + LD::ACC_W;
    for (int i = threadIdx.x; i < LD::ACC_W + 64; i += 256) accW2[i] = 0.f;
    __syncthreads();
    const int F = d.F;

    // dW2 / db2 of the four k<=4 heads live in registers for the whole (persistent) kernel: 4x4x1 MFMA form, lane l register
    // i = dW2[i][64u + l] (u-th 64-feature chunk); db2 partials per lane (block b = lane/4 holds Gaussians b and 16+b)
    constexpr int NCH = W / 64;
    f32x4 sw[4][NCH];
    float sb[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        sb[q] = 0.f;
#pragma unroll
        for (int u = 0; u < NCH; u++) sw[q][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool small_on = d.small_heads != 0;
    // dW2 of the 48-row SH head: the four waves of the workgroup pool their transposed relu(h1) tiles (128 Gaussians) and
    // every wave OWNS one 32-column block of dW2 for both 32-row tiles, so its sums go into the LDS accumulator with plain
    // read-add-write instead of atomics.  (The per-wave form needed ~96 ds_add_f32 per tile to merge the waves' partial
    // sums: ~30 k cycles per tile, 9 % of the kernel, in-kernel cycle profile of round 1.)  Two workgroup barriers per tile.
    constexpr int NU = WT == 4 ? 2 : 1;            // (ot2, tb) units per wave: WT=4: (0,w),(1,w); WT=2: (w>>1, w&1)
    bool tiles_shared = false;   // another wave may still be reading this wave's tile: barrier before overwriting it
#ifdef FDGS_PROFILE_D2
    unsigned long long prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long prof_t = __builtin_amdgcn_s_memtime();
    const unsigned long long prof_t0 = prof_t;
#endif
#define FDGS_TV_LIST(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define FDGS_TV_DECL(j) float4 tv##j = make_float4(0.f, 0.f, 0.f, 0.f);
    FDGS_TV_LIST(FDGS_TV_DECL)
    bool tv_loaded = false;   // SAVED: the first head's relu(h1) tile of this tile was requested during the previous tile
    // the tiles to process: the live list (tiles with a non-zero gradient row, padded to whole groups of four)
    const const_u32p live_list = as_const(d.s.live);
    const int nlive4 = (int)as_const(d.s.counters)[ROWS ? 5 : 1];
    const int it_stride = gridDim.x * 4;
    // ROWS: list entry of lane g (both halves) for this wave's current / next tile, fetched a whole tile ahead
    uint32_t ridx_cur = 0u, ridx_nxt = 0u;
    if constexpr (ROWS) {
        const int it0 = blockIdx.x * 4 + wave;
        if (it0 < nlive4) ridx_nxt = d.s.rows[(size_t)it0 * 32 + (lane & 31)];
    }
    // ROWS: the LEFTOVER round split by head.  A tile keeps a wave busy for ~100 us and a launch has 4 x CUs waves: with 1 126 tiles on
    // 1 024 waves (the bench scene) the kernel takes two tile times although the second round holds 102 tiles.  When the tiles behind the
    // last full round number at most one per workgroup, workgroup b takes tile nfull + b with its FOUR waves: each wave runs a subset of
    // the heads (the 48-row SH head alone, the fifth head with the second wave), the partial dhid meet in LDS and wave 0 finishes the tile.
    unsigned all_heads = 0u, my_heads = 0u;
    int n_on = 0;
    {
        const int ord[FDGS_NUM_HEADS] = {FDGS_HEAD_SHS, FDGS_HEAD_POS, FDGS_HEAD_SCALE, FDGS_HEAD_ROT, FDGS_HEAD_OPACITY};
#pragma unroll
        for (int i = 0; i < FDGS_NUM_HEADS; i++)
            if (p.head_on[ord[i]]) {
                all_heads |= 1u << ord[i];
                if ((n_on < 4 ? n_on : 1) == wave) my_heads |= 1u << ord[i];
                n_on++;
            }
    }
    const int nfull = nlive4 / it_stride * it_stride, nleft = nlive4 - nfull;
    const bool split = ROWS && n_on > 1 && nleft > 0 && nleft <= (int)gridDim.x;
    const int n_normal = split ? nfull : nlive4;
    bool sp_pending = split && (int)blockIdx.x < nleft;
    // (list indices are made wave-uniform BEFORE they address the list: scalar loads.  As vector loads they would join the in-order
    // vmcnt queue behind the prefetched activation rows and every read of the list would wait for those.)
    for (int it = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave); ROWS || it < nlive4; it += it_stride) {
        bool sp = false;       // this iteration is the workgroup's tile of the split round
        if constexpr (ROWS) {
            if (it >= n_normal) {
                if (!sp_pending) break;
                sp = true; sp_pending = false;
            }
        }
        const unsigned heads_it = sp ? my_heads : all_heads;
        // (the tile-list kernels keep walking p.head_on: their register allocation is at the edge, 512 registers and 92 bytes of spills)
        auto nexth = [&](unsigned m, int cur) { if constexpr (ROWS) return next_head_m(m, cur); else return next_head(p.head_on, cur); };
        if (ROWS && sp && tiles_shared) { __syncthreads(); tiles_shared = false; }     // (a wave without a head in the split round would miss the barrier at the head top)
        // opaque per-iteration copies of the lane coordinates: keeps the (hundreds of) loop-invariant weight addresses
        // from being hoisted out of the tile loop and held in registers across it
        int g = g0, h = h0;
        asm volatile("" : "+v"(g), "+v"(h));
        const int tile = sp ? nfull + (int)blockIdx.x : (ROWS ? it : (int)live_list[it]);
        const int tile_next = !sp && it + it_stride < n_normal ? (ROWS ? it + it_stride : (int)live_list[it + it_stride]) : -1;
        if constexpr (ROWS) {
            if (sp) ridx_nxt = d.s.rows[(size_t)tile * 32 + (lane & 31)];      // (not requested ahead: one exposed round trip per launch)
            ridx_cur = ridx_nxt & ~ROW_PAD;
            if (tile_next >= 0) ridx_nxt = d.s.rows[(size_t)tile_next * 32 + (lane & 31)];
        }
        const int n0 = tile * 32;  // first Gaussian (ROWS: first list position) of this wave's tile (rows < Npad always exist in scratch)
        const int n_row = n0 + g;
        const int n = n_row < p.N ? n_row : p.N - 1;
        int hd = nexth(heads_it, -1);
        DenseIL<WT, WT, true, FwdPD<WT>::L1, false> L1;
        f32x16 hid[SAVED ? 1 : WT], dhid[WT];
        uint32_t hidmask[WT];   // SAVED: bit r of hidmask[t] = relu(hidden)[t][r] > 0
        if constexpr (!SAVED) {
            DenseTrunk<FCH, WT, 2> T0;
            T0.setup(p.w0, p.b0, F, g, h);
            T0.preload();
            L1.setup(p.w1[hd], p.b1[hd], W, W, g, h);   // at least one head is active (checked on the host)
            L1.preload();
            float q[4], xyz[3];
            load_query(p, d.sc, n, q, xyz);
            f32x16 feat[FT];
#pragma unroll
            for (int t = 0; t < FT; t++) feat[t] = zero16();
            gather_features<FCH>(p, q, h, feat);
#pragma unroll
            for (int j = 0; j < FCH; j++)
                *reinterpret_cast<float4*>(d.s.FEAT + (size_t)n_row * F + 8 * j + 4 * h) =
                    make_float4(feat[j / 4][4 * (j % 4)], feat[j / 4][4 * (j % 4) + 1], feat[j / 4][4 * (j % 4) + 2], feat[j / 4][4 * (j % 4) + 3]);
            D2_TICK(0);
            T0.run(feat, hid, h);
            relu_inplace<WT>(hid);
            store_il<WT>(d.s.RH + (size_t)n_row * W, hid, h);
            D2_TICK(1);
        } else {
            (void)n;
            // (sixteen 16-byte loads of the saved relu(hidden) row used to be spilled one by one here: sixteen serialised
            // HBM round trips per tile; the forward now leaves the bits behind in this lane layout)
            // (ROWS: the bits of list entry g sit in the word quadruple of its own Gaussian: tile r / 32, lane (r % 32, h))
            const uint4 hm = reinterpret_cast<const uint4*>(d.sv_hmask)[ROWS ? (size_t)(ridx_cur >> 5) * 64 + 32 * h + (ridx_cur & 31u) : (size_t)tile * 64 + lane];
            const uint32_t hmw[4] = {hm.x, hm.y, hm.z, hm.w};
#pragma unroll
            for (int t = 0; t < WT; t++) hidmask[t] = hmw[t];
        }
        // SAVED: the relu(h1) tile of the NEXT head to process is fetched one head ahead (64 registers), under the long
        // transposed product of the current one -- the kernel is otherwise HBM-latency bound on these 16-KB tiles
        // (sixteen named registers quadruples, not an array: an array carried across the head loop is "promoted" to LDS by
        // the compiler's alloca pass instead of being scalarised)
#if defined(FDGS_NT_LOAD) && FDGS_NT_LOAD      // (development variant: the saved rows are read once)
#define FDGS_TV_LOAD(j) if (j < WT * 4) { typedef float v4nt_ __attribute__((ext_vector_type(4))); \
        const v4nt_ t_ = __builtin_nontemporal_load(reinterpret_cast<const v4nt_*>(tsrc + (j * 64 + lane))); tv##j = make_float4(t_.x, t_.y, t_.z, t_.w); }
#else
#define FDGS_TV_LOAD(j) if (j < WT * 4) tv##j = tsrc[j * 64 + lane];
#endif
        // ROWS: float4 j * 64 + lane of the tile is columns 4 lc4 .. of list entry j * RPL + lrow: that entry's row of the head's slab (tslab).
        // (lrow / lc4 come from the per-head opaque copies of the lane coordinates: derived from the plain lane id they are loop
        // invariants, and the compiler keeps -- and spills -- one select index and one column offset per request)
#define FDGS_TV_LOAD_ROWS(j) if (j < WT * 4) { \
        const uint32_t r_ = (uint32_t)__shfl((int)ridx_sel, j * (256 / W) + lrow_, 64) & ~ROW_PAD; \
        tv##j = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tslab) + (r_ * (uint32_t)(W * 4) + coff_)); }
#define FDGS_TV_ROWS_COORDS const int l64_ = 32 * h + g, lrow_ = l64_ / (W / 4); const uint32_t coff_ = (uint32_t)(l64_ % (W / 4)) * 16u;
#define FDGS_TV_STORE(j) if (j < WT * 4) { const int e4 = j * 64 + lane, row = e4 / (W / 4), c4 = e4 - row * (W / 4); \
                                          *reinterpret_cast<float4*>(lds + row * STRIDE + 4 * c4) = tv##j; }
        if constexpr (SAVED) {
            if (!tv_loaded && (!ROWS || hd < FDGS_NUM_HEADS)) {
                if constexpr (ROWS) {
                    const float* tslab = d.sv_h1 + (size_t)d.head_slot[hd] * d.s.Npad * W;
                    const uint32_t ridx_sel = ridx_cur;
                    FDGS_TV_ROWS_COORDS
                    FDGS_TV_LIST(FDGS_TV_LOAD_ROWS)
                } else {
                    const float4* tsrc = reinterpret_cast<const float4*>(d.sv_h1 + ((size_t)d.head_slot[hd] * d.s.Npad + n0) * W);
                    FDGS_TV_LIST(FDGS_TV_LOAD)
                }
            }
        }
        // SAVED: request the relu(h1) rows that are needed NEXT (next head of this tile, or the first head of this wave's
        // next tile) as soon as the 64 staging registers are free, i.e. right after they were copied to LDS -- a whole head
        // iteration ahead.  (Requested just before B1.run they sat in front of B1's operand ring in the in-order load
        // queue and every head paid their HBM latency at its first MFMA: 21 k instead of 18 k cycles per head.)
        auto request_next_rows = [&](int cur_hd) {
            if constexpr (SAVED) {
                int nx = nexth(heads_it, cur_hd);
                int nn0 = n0;
                const bool wrap = nx >= FDGS_NUM_HEADS;
                if (wrap) { nx = nexth(all_heads, -1); nn0 = tile_next * 32; }
                const bool have = !wrap || tile_next >= 0;
                tv_loaded = have && wrap;   // "this wave's next tile finds its first rows already requested"
                if (have) {
                    if constexpr (ROWS) {
                        const float* tslab = d.sv_h1 + (size_t)d.head_slot[nx] * d.s.Npad * W;
                        const uint32_t ridx_sel = wrap ? ridx_nxt : ridx_cur;
                        FDGS_TV_ROWS_COORDS
                        FDGS_TV_LIST(FDGS_TV_LOAD_ROWS)
                    } else {
                        const float4* tsrc = reinterpret_cast<const float4*>(d.sv_h1 + ((size_t)d.head_slot[nx] * d.s.Npad + nn0) * W);
                        FDGS_TV_LIST(FDGS_TV_LOAD)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma unroll
        for (int t = 0; t < WT; t++) dhid[t] = zero16();
        const float* Grow = d.s.G + (size_t)n_row * GCOLS;

        while (hd < FDGS_NUM_HEADS) {
            asm volatile("" : "+v"(g), "+v"(h));   // no hoisting of per-layer address arithmetic out of the head loop
            const int k = head_k(hd), off = head_off(hd), row0 = head_row0(hd);
            uint32_t mask[WT];  // bit r of mask[t]: h1[t][r] > 0
            const int nt2 = k > 32 ? 2 : 1;
            // operands that depend on nothing computed in this head are requested first: the head's packed output-gradient
            // rows for the dW2 product (A-lane: output o = 32*ot2 + g, gaussian 2s+h) ...
            // (loads are unconditional -- columns past the head's k outputs lie inside the scratch buffer -- and zeroed by
            // a select: a conditional load becomes a branch that the compiler sinks to the use, exposing its latency)
            const bool small = small_on && k <= 4;
            const bool coop_e = !small && k > 32 && !(ROWS && sp);     // (split round: the four waves hold the SAME tile -- the per-wave form below)
            float ga[16];
            float sa0 = 0.f, sa1 = 0.f;   // small path: A-lane 4b+i = G[gaussian b (+16)][output i]
            if (small) {
                const float* gp = d.s.G + (size_t)(n0 + (lane >> 2)) * GCOLS + off + (lane & 3);
                sa0 = gp[0];
                sa1 = gp[(size_t)16 * GCOLS];
                sa0 = (lane & 3) < k ? sa0 : 0.f;
                sa1 = (lane & 3) < k ? sa1 : 0.f;
            } else if (ROWS ? !coop_e : k <= 32) {   // (the 48-row head takes the cooperative path and loads its rows there)
                const float* gp = d.s.G + (size_t)(n0 + h) * GCOLS + off + g;
#pragma unroll
                for (int s = 0; s < 16; s++) ga[s] = gp[(size_t)2 * s * GCOLS];
#pragma unroll
                for (int s = 0; s < 16; s++) ga[s] = g < k ? ga[s] : 0.f;
            } else {
#pragma unroll
                for (int s = 0; s < 16; s++) ga[s] = 0.f;
            }
            if constexpr (!SAVED) {
                f32x16 h1[WT];
                L1.run(hid, h1, h);
                D2_TICK(2);
#pragma unroll
                for (int t = 0; t < WT; t++) {
                    mask[t] = 0;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        h1[t][r] = fmaxf(h1[t][r], 0.f);
                        mask[t] |= (h1[t][r] > 0.f ? 1u : 0u) << r;
                    }
                }
                // transposed copy relu(h1)[gaussian][feature] for the dW2 product
                if (tiles_shared) { __syncthreads(); tiles_shared = false; }
                store_il<WT>(lds + g * STRIDE, h1, h);
            } else {
                // the saved relu(h1) rows of this tile are 32 x W contiguous floats: copy them, lane-consecutive, into the
                // padded LDS tile [gaussian][W + 4]; then every lane reads its own Gaussian's row back for the ReLU mask
                if (tiles_shared) { __syncthreads(); tiles_shared = false; }
                FDGS_TV_LIST(FDGS_TV_STORE)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                D2_TICK(0);
#pragma unroll
                for (int t = 0; t < WT; t++) mask[t] = 0;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const AVec<WT> v = ldv<WT>(lds + g * STRIDE + WT * rho(r, h));
#pragma unroll
                    for (int t = 0; t < WT; t++) mask[t] |= (v.v[t] > 0.f ? 1u : 0u) << r;
                }
                if (ROWS ? !coop_e : !(!small && k > 32)) request_next_rows(hd);   // (the cooperative SH block needs the registers first)
                D2_TICK(2);
            }
            DenseT<WT, WT, true, 4> B1;
            const float* w2p = p.w2[hd] + WT * g;
            const int nsteps = (k + 1) >> 1;
            auto ldA = [&](int s) { int o = 2 * s + h; o = o < k ? o : k - 1; return ldv<WT>(w2p + (size_t)o * W); };
            // (raw loads: the o < k select is applied where the value is consumed -- a select next to the request would wait
            // for it, and with the next tile rows queued in front of it that wait is an HBM round trip)
            auto ldB = [&](int s) { return Grow[off + 2 * s + h]; };
            auto selB = [&](float v, int s) { return 2 * s + h < k ? v : 0.f; };
            AVec<WT> a0, a1, a2;
            float b0, b1, b2;
            auto early_requests = [&]() {
                B1.setup(p.w1[hd], W, W, g, h);
                B1.preload();
                a0 = ldA(0); a1 = ldA(1 < nsteps ? 1 : 0); a2 = ldA(2 < nsteps ? 2 : 0);
                b0 = ldB(0); b1 = ldB(1 < nsteps ? 1 : 0); b2 = ldB(2 < nsteps ? 2 : 0);   // steps >= nsteps are never consumed
            };
            const bool coop = ROWS ? coop_e : (!small && k > 32);
            if (!coop) early_requests();   // (the cooperative SH block needs the registers: requests follow it)
            __builtin_amdgcn_sched_barrier(0);   // keep these requests ahead of the dW2 block
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            D2_TICK(3);
            // ---- dW2[o][in] += sum_g G[g][o] * relu(h1)[g][in];  db2[o] += sum_g G[g][o]
            if (small) {
                // one 4x4x1 MFMA per (Gaussian, 64-feature chunk): B-lane l = relu(h1)[gaussian][64u + l] straight from
                // the transposed tile, A = the Gaussian's k gradient values broadcast from block gq%16 of sa0/sa1
                auto small_dw2 = [&](f32x4* acc, float& bsum) {
                    bsum += sa0 + sa1;
                    small_dw2_steps<NCH, 0>(acc, sa0, sa1, lds, STRIDE, lane);
                };
                if (hd == FDGS_HEAD_POS) small_dw2(sw[0], sb[0]);
                else if (hd == FDGS_HEAD_SCALE) small_dw2(sw[1], sb[1]);
                else if (hd == FDGS_HEAD_ROT) small_dw2(sw[2], sb[2]);
                else small_dw2(sw[3], sb[3]);
            }
            if (coop) {
                D2_TICK(4);
                __syncthreads();                                   // all four tiles of the workgroup are written
                D2_TICK(1);
                // first Gaussians of the workgroup's four tiles (entries it - wave .. + 3 of the live list)
                int wgn0[4];
#pragma unroll
                for (int c = 0; c < 4; c++) wgn0[c] = (ROWS ? (it & ~3) + c : (int)live_list[(it & ~3) + c]) * 32;
#pragma unroll
                for (int j = 0; j < NU; j++) {
                    const int ot2 = WT == 4 ? j : (wave >> 1), tb = WT == 4 ? wave : (wave & 1);
                    const int o = ot2 * 32 + g;
                    // (a uniform 64-bit base per tile + ONE 32-bit lane offset for all tiles and steps: SGPR-base addressing.  Written as
                    // `gp[(wgn0[c] + 2 s) * GCOLS]` the 64 requests took a 64-bit vector address each: +300 bytes of spills per lane,
                    // and the spill reloads wait in the in-order vmcnt queue behind the prefetched activation rows)
                    const uint32_t gvo = (uint32_t)((h * GCOLS + off + o) * 4);
                    auto gld = [&](int c, int s) {
                        const char* base = reinterpret_cast<const char*>(d.s.G) + (size_t)wgn0[c] * (GCOLS * 4);
                        return *reinterpret_cast<const float*>(base + (gvo + (uint32_t)(2 * s * GCOLS * 4)));
                    };
                    const float* bp = lds_all + tb * 32 + g;
                    float gq[2][16];
                    f32x16 accS = zero16();
                    float asumS = 0.f;
#pragma unroll
                    for (int s = 0; s < 16; s++) gq[0][s] = gld(0, s);
#pragma unroll
                    for (int c = 0; c < 4; c++) {                  // 4 chunks of 16 k-steps = the 4 tiles (32 Gaussians each)
                        if (c + 1 < 4) {
#pragma unroll
                            for (int s = 0; s < 16; s++) gq[(c + 1) & 1][s] = gld(c + 1, s);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int s = 0; s < 16; s++) {
                            const float a = o < k ? gq[c & 1][s] : 0.f;
                            asumS += a;
                            accS = mfma32(a, bp[c * LD::TILE_FLOATS + (2 * s + h) * STRIDE], accS);
                        }
                    }
                    // this wave is the only writer of these cells: plain LDS read-add-write
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int orow = ot2 * 32 + rho(r, h);
                        if (orow < k) accW2[(row0 + orow) * W + tb * 32 + g] += accS[r];
                    }
                    if (tb == 0) {   // db2: one wave per row tile
                        asumS += __shfl_xor(asumS, 32, 64);
                        if (h == 0 && o < k) accB2[row0 + o] += asumS;
                    }
                }
                tiles_shared = true;
                D2_TICK(10);
                request_next_rows(hd);
                early_requests();
                D2_TICK(11);
            }
            for (int ot2 = 0; ot2 < ((small || coop) ? 0 : nt2); ot2++) {
                const int o = ot2 * 32 + g;
                if (ot2 > 0) {
                    const float* gp = d.s.G + (size_t)(n0 + h) * GCOLS + off + o;
#pragma unroll
                    for (int s = 0; s < 16; s++) ga[s] = gp[(size_t)2 * s * GCOLS];
#pragma unroll
                    for (int s = 0; s < 16; s++) ga[s] = o < k ? ga[s] : 0.f;
                }
                float asum = 0.f;
#pragma unroll
                for (int s = 0; s < 16; s++) asum += ga[s];
                asum += __shfl_xor(asum, 32, 64);
                const int kk = k - ot2 * 32;  // valid rows of this 32-row output tile
                if (h == 0 && g < kk) atomicAdd(&accB2[row0 + ot2 * 32 + g], asum);
                D2_TICK(10);
#pragma unroll
                for (int tb = 0; tb < WT; tb += 2) {   // two feature tiles at a time: 32 accumulator registers
                    f32x16 acc0 = zero16(), acc1 = zero16();
#pragma unroll
                    for (int s = 0; s < 16; s++) {
                        acc0 = mfma32(ga[s], lds[(2 * s + h) * STRIDE + tb * 32 + g], acc0);
                        acc1 = mfma32(ga[s], lds[(2 * s + h) * STRIDE + (tb + 1) * 32 + g], acc1);
                    }
                    D2_TICK(11);
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int orow = rho(r, h);
                        if (orow < kk) {
                            atomicAdd(&accW2[(row0 + ot2 * 32 + orow) * W + tb * 32 + g], acc0[r]);
                            atomicAdd(&accW2[(row0 + ot2 * 32 + orow) * W + (tb + 1) * 32 + g], acc1[r]);
                        }
                    }
                }
            }
            D2_TICK(4);
            // ---- dh1 = W2^T G_head, masked by relu'(h1)
            f32x16 dh1[WT];
#pragma unroll
            for (int t = 0; t < WT; t++) dh1[t] = zero16();
            if (k > 32) {
                // the 48-output head: 24 k-steps, fully unrolled with a 6-deep operand ring (a 3-deep rotating ring left the
                // MFMAs waiting on L2 for most steps: 12 % of the kernel in the cycle profile)
                constexpr int PDH = 6, NSH = 24;
                AVec<WT> ra[PDH];
                float rb[PDH];
                ra[0] = a0; ra[1] = a1; ra[2] = a2; rb[0] = b0; rb[1] = b1; rb[2] = b2;
#pragma unroll
                for (int s = 3; s < PDH; s++) { ra[s] = ldA(s); rb[s] = ldB(s); }
#pragma unroll
                for (int s = 0; s < NSH; s++) {
                    const AVec<WT> a = ra[s % PDH];
                    const float b = selB(rb[s % PDH], s);
                    if (s + PDH < NSH) { ra[s % PDH] = ldA(s + PDH); rb[s % PDH] = ldB(s + PDH); }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < WT; t++) dh1[t] = mfma32(a.v[t], b, dh1[t]);
                }
            } else {
                for (int s = 0; s < nsteps; s++) {
                    const AVec<WT> a = a0;
                    const float b = selB(b0, s);
                    a0 = a1; b0 = b1; a1 = a2; b1 = b2;
                    if (s + 3 < nsteps) { a2 = ldA(s + 3); b2 = ldB(s + 3); }
#pragma unroll
                    for (int t = 0; t < WT; t++) dh1[t] = mfma32(a.v[t], b, dh1[t]);
                }
            }
            D2_TICK(5);
            float* slab = d.s.DH1 + (size_t)d.head_slot[hd] * d.s.Npad * W;
#pragma unroll
            for (int t = 0; t < WT; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) dh1[t][r] = ((mask[t] >> r) & 1u) ? dh1[t][r] : 0.f;
            // through the (now idle) LDS tile when no other wave can still be reading it: 32 contiguous rows, 1 KB per store
            if (!tiles_shared) store_tile_coalesced<WT>(lds, slab + (size_t)n0 * W, dh1, g, h, lane);
            else store_il<WT>(slab + (size_t)n_row * W, dh1, h);
            D2_TICK(6);
            // ---- dhid += W1^T dh1
            B1.run(dh1, dhid);
            __builtin_amdgcn_wave_barrier();
            D2_TICK(7);
            hd = nexth(heads_it, hd);
            if constexpr (!SAVED) {
                if (hd < FDGS_NUM_HEADS) { L1.setup(p.w1[hd], p.b1[hd], W, W, g, h); L1.preload(); }
            }
        }
        if constexpr (ROWS) {
            if (sp) {      // the waves' partial dhid meet in their (idle) LDS tiles; wave 0 sums them and finishes the tile
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < WT; t++)
#pragma unroll
                    for (int r = 0; r < 16; r++) lds[(t * 16 + r) * 64 + lane] = dhid[t][r];
                __syncthreads();
                if (wave != 0) continue;      // (the last iteration of this workgroup: everybody meets again at the flush below)
                const int nsets = n_on < 4 ? n_on : 4;
                for (int w = 1; w < nsets; w++) {
#pragma unroll
                    for (int t = 0; t < WT; t++)
#pragma unroll
                        for (int r = 0; r < 16; r++) dhid[t][r] += lds_all[w * LD::TILE_FLOATS + (t * 16 + r) * 64 + lane];
                }
            }
        }
        // relu'(hidden), store for the trunk weight gradient, then dfeat = W0^T dhid
        DenseT<WT, FT, false, 16> B0;   // one dword per k-step and only FT MFMAs behind it: a deep ring hides the L2 latency
        B0.setup(p.w0, F, F, g, h);
        B0.preload();
#pragma unroll
        for (int t = 0; t < WT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                bool pos;
                if constexpr (SAVED) pos = (hidmask[t] >> r) & 1u; else pos = hid[t][r] > 0.f;
                dhid[t][r] = pos ? dhid[t][r] : 0.f;
            }
        store_il<WT>(d.s.DHID + (size_t)n_row * W, dhid, h);
        f32x16 dfeat[FT];
#pragma unroll
        for (int t = 0; t < FT; t++) dfeat[t] = zero16();
        B0.run(dhid, dfeat);
#pragma unroll
        for (int j = 0; j < FCH; j++)
            *reinterpret_cast<float4*>(d.s.DFEAT + (size_t)n_row * F + 8 * j + 4 * h) =
                make_float4(dfeat[j / 4][4 * (j % 4)], dfeat[j / 4][4 * (j % 4) + 1], dfeat[j / 4][4 * (j % 4) + 2],
                            dfeat[j / 4][4 * (j % 4) + 3]);
        D2_TICK(8);
    }
#ifdef FDGS_PROFILE_D2
    if (d.prof && lane == 0) {
        unsigned long long* out = d.prof + (size_t)(blockIdx.x * 4 + wave) * 12;
        for (int i = 0; i < 9; i++) out[i] = prof_acc[i];
        out[9] = __builtin_amdgcn_s_memtime() - prof_t0;
        out[10] = prof_acc[10]; out[11] = prof_acc[11];
    }
#endif
    // the register-resident sums of the k<=4 heads join the LDS accumulators
    if (small_on) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int kq = head_k(q), r0 = head_row0(q);
            if (!p.head_on[q]) continue;
#pragma unroll
            for (int u = 0; u < NCH; u++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (i < kq) atomicAdd(&accW2[(r0 + i) * W + 64 * u + lane], sw[q][u][i]);
            float v = sb[q];   // lanes with equal (lane & 3): sum over the 16 blocks
            v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (lane < kq) atomicAdd(&accB2[r0 + lane], v);
        }
    }
    // flush the workgroup's dW2 / db2 sums
    __syncthreads();
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) {
        if (!p.head_on[hd]) continue;
        const int k = head_k(hd), row0 = head_row0(hd);
        for (int i = threadIdx.x; i < k * W; i += 256) {
            const float v = accW2[row0 * W + i];
            if (v != 0.f) atomicAdd(&d.d_w2[hd][i], v);
        }
        if ((int)threadIdx.x < k && accB2[row0 + threadIdx.x] != 0.f) atomicAdd(&d.d_b2[hd][threadIdx.x], accB2[row0 + threadIdx.x]);
    }
}

#include "deform_bwd_ws.h"

// ------------------------------------------------------------------------------------------------ D3 weight grads
// dW[m][c] += sum_n DY[n][m] * X[n][c]   (m < W rows of DY, c < ncols of X), db[m] += sum_n DY[n][m];  K = #Gaussians.
// One wave owns the WHOLE [W x ncols] product for its slice of Gaussians (16 accumulator tiles = 256 AGPRs at W = 128,
// one wave per SIMD): with the interleaved tile mapping (row m = WT*i + a, column c = WT*j + b) a k-step of two Gaussians
// needs exactly ONE 16-byte load of DY[n][WT*g ..] and ONE of X[n][WT*g ..] per lane for its 16 MFMAs -- both
// 512-byte coalesced rows, prefetched WG_PD steps ahead.  The four waves of a workgroup split the workgroup's Gaussians,
// meet in an LDS accumulator (ds_add_f32) and flush each weight once per workgroup with coalesced global atomics.
// (Round-1 kernel: one dword load per MFMA operand, 4 MFMAs per vmcnt(0) -> 36 % MFMA utilisation.)
struct WgradJob {
    const float* DY; const float* X; float* dW; float* db;
    int ldx, ncols, ldw;
    int first_block, nblocks;   // workgroups [first_block, first_block + nblocks) share the live tiles of this job evenly
};
struct WgradArgs {
    WgradJob job[FDGS_NUM_HEADS + 1];
    int njobs, Npad, W;
    const uint32_t* live;       // live-tile list (tile_compact_kernel)
    const uint32_t* counters;   // [1] = entries of the list
    const uint32_t* rows;       // ROWS kernel: the live-row list; DY is indexed by list position, X by the listed row; counters[5] = tiles of the list
};

// COLS_IL: X has exactly W columns, column mapping interleaved (vector loads); else tile mapping c = 32*b + j with
// CT = ceil(ncols/32) dword loads per k-step (the small trunk product, ncols = C*L).
// PD (8 or 16) = depth of the operand ring in k-steps: the narrow trunk products run only CT MFMAs per step, so they need a deeper
// ring than the square head products to cover the same memory latency.
// The wave walks the tiles [t_begin, t_end) of the LIVE list: a tile is 32 consecutive Gaussians = 16 k-steps, tiles need not be
// adjacent in memory (tiles whose gradient rows are all zero were dropped from the list: their products are exactly zero).
// ROWS: the tiles are 32 consecutive entries of the live-ROW list (`live` = that list): DY rows are list positions, X rows the listed Gaussians.
template <int WT, int CT, bool COLS_IL, int PD, bool ROWS>
__device__ __forceinline__ void wgrad_wave(const WgradJob& J, int W, const_u32p live, int t_begin, int t_end, float* ldsW, float* ldsB,
                                           int g, int h, int wave) {
    static_assert(16 % PD == 0, "the ring must divide a tile's 16 k-steps");
    constexpr int BV = COLS_IL ? WT : 1, NB = COLS_IL ? 1 : CT;
    f32x16 acc[WT][CT];
#pragma unroll
    for (int a = 0; a < WT; a++)
#pragma unroll
        for (int b = 0; b < CT; b++) acc[a][b] = zero16();
    float asum[WT];
#pragma unroll
    for (int a = 0; a < WT; a++) asum[a] = 0.f;
    const float* ap = J.DY + (size_t)h * W + WT * g;          // + row * W, row = first Gaussian of the k-step (even)
    const float* bp[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        int col = COLS_IL ? WT * g : 32 * b + g;
        col = col < J.ncols ? col : J.ncols - 1;
        bp[b] = J.X + (ROWS ? (size_t)0 : (size_t)h * J.ldx) + col;
    }
    AVec<WT> abuf[PD];
    AVec<BV> bbuf[PD][NB];
    // slot u is consumed, THEN refilled in place with the step PD ahead; no control flow inside a tile.  (A first form copied the
    // slot, refilled it and then ran the MFMAs under `if (s < nsteps)`; the register copies at the loop back-edge and the per-step
    // branches made the wait-count pass put `vmcnt(0)` in front of the last MFMAs of EVERY step: 57 % MFMA utilisation, rocprofv3 r01h.)
    auto consume = [&](int u) {
#pragma unroll
        for (int a = 0; a < WT; a++) asum[a] += abuf[u].v[a];
#pragma unroll
        for (int a = 0; a < WT; a++)
#pragma unroll
            for (int b = 0; b < CT; b++)
                acc[a][b] = mfma32(abuf[u].v[a], COLS_IL ? bbuf[u][0].v[b] : bbuf[u][b].v[0], acc[a][b]);
    };
    auto fill = [&](int u, int row) {
        abuf[u] = ldv<WT>(ap + (size_t)row * W);
        if constexpr (ROWS) {      // (two scalar reads of the list, one select: lane half h takes entry row + h)
            const uint32_t ra = live[row] & ~ROW_PAD, rb = live[row + 1] & ~ROW_PAD;
            const uint32_t xr = h ? rb : ra;
#pragma unroll
            for (int b = 0; b < NB; b++) bbuf[u][b] = ldv<BV>(bp[b] + (size_t)xr * J.ldx);
        } else {
#pragma unroll
            for (int b = 0; b < NB; b++) bbuf[u][b] = ldv<BV>(bp[b] + (size_t)row * J.ldx);
        }
    };
    if (t_end > t_begin) {
        auto tile_row = [&](int ti) { const int tc = ti < t_end ? ti : t_end - 1; return ROWS ? tc * 32 : (int)live[tc] * 32; };      // (uniform index: scalar load)
        int cur = tile_row(t_begin), nxt = tile_row(t_begin + 1);
#pragma unroll
        for (int u = 0; u < PD; u++) fill(u, cur + 2 * u);
        for (int ti = t_begin; ti < t_end; ti++) {
            const int nxt2 = tile_row(ti + 2);      // (scalar load, consumed one tile later)
#pragma unroll
            for (int gi = 0; gi < 16 / PD; gi++) {
#pragma unroll
                for (int u = 0; u < PD; u++) {
                    consume(u);
                    __builtin_amdgcn_sched_barrier(0);
                    const int sn = gi * PD + u + PD;              // the step this slot holds next: same tile, or the next one
                    fill(u, (sn < 16 ? cur : nxt) + 2 * (sn & 15));   // (past the last tile: harmless re-load, never consumed)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            cur = nxt; nxt = nxt2;
        }
    }
    // workgroup reduction in LDS, ldsW[m * ldl + c] with ldl = 32*CT: the four waves take turns (barrier between turns) and
    // use plain stores / read-add-writes -- ds_add_f32 runs at 0.33 lanes/clk/CU on MI355X (tools/lds_atomic_bench.hip:
    // 37x slower than ds_add_u32, ~200x slower than plain LDS traffic) and 4 x 16 k float atomics cost ~15 % of this kernel
    constexpr int LDL = 32 * CT;
    float bsum[WT];
#pragma unroll
    for (int a = 0; a < WT; a++) bsum[a] = asum[a] + __shfl_xor(asum[a], 32, 64);
    // (interleaved columns with WT = 4: the four column tiles of a lane are 16 contiguous bytes -- one ds_write_b128 / ds_read_b128 instead of
    // four 4-byte accesses at a 16-byte lane stride, which run four-way bank-conflicted)
    constexpr bool V4 = COLS_IL && CT == 4;
    if (wave == 0) {
        if constexpr (V4) {
#pragma unroll
            for (int a = 0; a < WT; a++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    *reinterpret_cast<float4*>(&ldsW[(WT * rho(r, h) + a) * LDL + WT * g]) = make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
        } else {
#pragma unroll
        for (int a = 0; a < WT; a++)
#pragma unroll
            for (int b = 0; b < CT; b++)
#pragma unroll
                for (int r = 0; r < 16; r++) ldsW[(WT * rho(r, h) + a) * LDL + (COLS_IL ? WT * g + b : 32 * b + g)] = acc[a][b][r];
        }
        if (h == 0) {
#pragma unroll
            for (int a = 0; a < WT; a++) ldsB[WT * g + a] = bsum[a];
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int turn = 1; turn < 4; turn++) {
        if (wave == turn) {
            if constexpr (V4) {
#pragma unroll
                for (int a = 0; a < WT; a++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        float4* q = reinterpret_cast<float4*>(&ldsW[(WT * rho(r, h) + a) * LDL + WT * g]);
                        float4 v = *q;
                        v.x += acc[a][0][r]; v.y += acc[a][1][r]; v.z += acc[a][2][r]; v.w += acc[a][3][r];
                        *q = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);   // 16 read-add-writes at a time (hoisted ds_reads would spill)
                }
            } else
#pragma unroll
            for (int a = 0; a < WT; a++)
#pragma unroll
                for (int b = 0; b < CT; b++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) ldsW[(WT * rho(r, h) + a) * LDL + (COLS_IL ? WT * g + b : 32 * b + g)] += acc[a][b][r];
                    __builtin_amdgcn_sched_barrier(0);   // 16 read-add-writes at a time (256 hoisted ds_reads would spill)
                }
            if (h == 0) {
#pragma unroll
                for (int a = 0; a < WT; a++) ldsB[WT * g + a] += bsum[a];
            }
        }
        __syncthreads();
    }
}

template <int WT, bool ROWS>
__global__ void __launch_bounds__(256, 1) deform_wgrad_kernel(WgradArgs a) {
    constexpr int W = WT * 32;
    __shared__ __attribute__((aligned(16))) float lds[W * W + W];
    // job of this workgroup
    int j = 0;
#pragma unroll
    for (int q = 1; q < FDGS_NUM_HEADS + 1; q++)
        if (q < a.njobs && (int)blockIdx.x >= a.job[q].first_block) j = q;
    const WgradJob J = a.job[j];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane & 31, h = lane >> 5;
    const int blk = (int)blockIdx.x - J.first_block;
    // this workgroup's share of the live tiles, split over its four waves
    const long long nlive = (long long)(int)as_const(a.counters)[ROWS ? 5 : 1];
    const const_u32p live_list = as_const(ROWS ? a.rows : a.live);
    const int t0 = (int)(nlive * blk / J.nblocks), t1 = (int)(nlive * (blk + 1) / J.nblocks);
    if (t1 <= t0) return;           // (uniform: nothing to add to the weight gradients)
    const int per = (t1 - t0 + 3) / 4;
    int wb = t0 + __builtin_amdgcn_readfirstlane(wave) * per, we = wb + per;      // (wave-uniform: the list is read with scalar loads)
    if (wb > t1) wb = t1;
    if (we > t1) we = t1;
    float* ldsW = lds;
    float* ldsB = lds + W * W;
    const int CTn = (J.ncols + 31) / 32;
    // (every wave joins, also one whose slice is empty: the reduction inside is a workgroup-wide protocol)
    if (J.ncols == W) wgrad_wave<WT, WT, true, 8, ROWS>(J, W, live_list, wb, we, ldsW, ldsB, g, h, wave);
    else if (CTn == 1) wgrad_wave<WT, 1, false, 16, ROWS>(J, W, live_list, wb, we, ldsW, ldsB, g, h, wave);
    else if (CTn == 2) wgrad_wave<WT, 2, false, 16, ROWS>(J, W, live_list, wb, we, ldsW, ldsB, g, h, wave);
    else if (CTn == 3) wgrad_wave<WT, 3, false, 8, ROWS>(J, W, live_list, wb, we, ldsW, ldsB, g, h, wave);
    else if constexpr (WT != 4) wgrad_wave<WT, 4, false, 8, ROWS>(J, W, live_list, wb, we, ldsW, ldsB, g, h, wave);   // (W = 128, 128 columns) is the interleaved case
    const int ldl = J.ncols == W ? W : 32 * CTn;
    for (int i = threadIdx.x; i < W * ldl; i += 256) {
        const int m = i / ldl, c = i - m * ldl;
        const float v = ldsW[i];
        if (c < J.ncols && v != 0.f) atomicAdd(&J.dW[(size_t)m * J.ldw + c], v);
    }
    if ((int)threadIdx.x < W && ldsB[threadIdx.x] != 0.f) atomicAdd(&J.db[threadIdx.x], ldsB[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------ D4 plane grads
// lanes <-> (x-corner, channel) of one Gaussian, so every atomic instruction covers whole 64/128-B texel lines.
// When every Gaussian shares one frame time (render()), all of them hit the SAME two rows of the three time planes
// (x,t),(y,t),(z,t): 50 % of the plane-gradient atomics land on ~128 hot lines (5 G float-atomics/s measured vs 20 G/s
// scattered).  Those planes are therefore privatised per workgroup in LDS: the two time rows receive the same sum
// scaled by the two (uniform) time weights, so ONE LDS tile [res_a][C] per plane accumulates sum(dv * wx) with
// ds_add_f32 and is flushed once per workgroup with coalesced global atomics (x w_t0 and x w_t1).
struct PlaneGradArgs {
    fdgs_deform_params p;
    AabbScale sc;
    const float* DFEAT;
    float* d_planes[FDGS_MAX_LEVELS][6];
    float* d_xyz;
    int F;
    int lds_off[FDGS_MAX_LEVELS][3];  // float offset of the LDS tile of time plane k = 2,4,5 (axis a = 0,1,2); -1: global atomics
    int lds_floats;
    int per_block;                    // Gaussians per workgroup
    const uint32_t* tile_live;        // [Npad/32] 1 = the tile's DFEAT rows were written by D2 (a dead tile's rows are garbage and count as zero)
};
constexpr int PG_THREADS = 512;
__host__ __device__ __forceinline__ int time_plane_slot(int k) { return k == 2 ? 0 : (k == 4 ? 1 : (k == 5 ? 2 : -1)); }

template <int C>
__global__ void __launch_bounds__(PG_THREADS) deform_plane_grad_kernel(PlaneGradArgs a) {
    const fdgs_deform_params& p = a.p;
    extern __shared__ float4 pg_lds4[];
    float* lds = reinterpret_cast<float*>(pg_lds4);
    constexpr int LPG = 2 * C, GPW = 64 / LPG, GPB = (PG_THREADS / 64) * GPW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ch = lane % C, xc = (lane / C) & 1, gsub = lane / LPG;
    for (int i = threadIdx.x; i < a.lds_floats; i += PG_THREADS) lds[i] = 0.f;
    __syncthreads();
    const int n_begin = blockIdx.x * a.per_block;
    const int n_end = n_begin + a.per_block < p.N ? n_begin + a.per_block : p.N;
    // software pipeline over the workgroup's batches: the coordinates (and frame time) of the NEXT batch are requested
    // before the current one is processed -- the kernel is latency bound (xyz -> texel addresses -> texels -> atomics)
    auto fetch_xyz = [&](int nbq, float* x4) {
        const int nr = nbq + wave * GPW + gsub;
        const int nn = nr < n_end ? nr : n_end - 1;
        x4[0] = p.xyz[3 * (size_t)nn]; x4[1] = p.xyz[3 * (size_t)nn + 1]; x4[2] = p.xyz[3 * (size_t)nn + 2];
        x4[3] = p.time ? p.time[nn] : p.time_scalar;
    };
    float xn[4];
    if (n_begin < n_end) fetch_xyz(n_begin, xn);
    for (int nb = n_begin; nb < n_end; nb += GPB) {
        const int n_raw = nb + wave * GPW + gsub;
        const bool live = n_raw < n_end && a.tile_live[(n_raw < n_end ? n_raw : n_end - 1) >> 5] != 0u;
        const int n = n_raw < n_end ? n_raw : n_end - 1;
        float q[4];
#pragma unroll
        for (int i = 0; i < 3; i++) q[i] = (xn[i] - p.aabb[i]) * a.sc.inv2[i] - 1.0f;
        q[3] = xn[3];
        if (nb + GPB < n_end) fetch_xyz(nb + GPB, xn);
        float dq[3] = {0.f, 0.f, 0.f};
        for (int lvl = 0; lvl < p.L; lvl++) {
            const float df = live ? a.DFEAT[(size_t)n * a.F + lvl * C + ch] : 0.f;
            float vk[6], sk[6], tk[6], wA[6], wB[6], wX[6], dsx[6], dsy[6];
            uint32_t oA[6], oB[6], oX[6];   // unsigned element offsets: SGPR base + VGPR offset addressing for loads and atomics
            // one sample per axis (x, y, z, t), shared by the planes that contain the axis
            AxisSample S[4];
#pragma unroll
            for (int ax4 = 0; ax4 < 4; ax4++) S[ax4] = axis_sample(q[ax4], p.res[lvl][ax4]);
#pragma unroll
            for (int k = 0; k < 6; k++) {
                int ax, bx;
                plane_axes(k, ax, bx);
                const int Wd = p.res[lvl][ax];
                const AxisSample sx = S[ax], sy = S[bx];
                const int xi = xc ? sx.i1 : sx.i0;
                const float wx = xc ? sx.w1 : sx.w0;
                oX[k] = (uint32_t)(xi * C + ch);
                oA[k] = (uint32_t)((sy.i0 * Wd + xi) * C + ch);
                oB[k] = (uint32_t)((sy.i1 * Wd + xi) * C + ch);
                const char* Pb = reinterpret_cast<const char*>(p.planes[lvl][k]);   // SGPR base + 32-bit byte offset
                const float v0 = *reinterpret_cast<const float*>(Pb + oA[k] * 4u);
                const float v1 = *reinterpret_cast<const float*>(Pb + oB[k] * 4u);
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
                wX[k] = wx; wA[k] = wx * sy.w0; wB[k] = wx * sy.w1;
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
                const int slot = time_plane_slot(k);
                const int loff = slot >= 0 ? a.lds_off[lvl][slot] : -1;
                if (dP && live) {
                    if (loff >= 0) {
                        atomicAdd(&lds[loff + oX[k]], dv * wX[k]);   // ds_add_f32
                    } else {
                        atomicAdd(&dP[oA[k]], dv * wA[k]);
                        atomicAdd(&dP[oB[k]], dv * wB[k]);
                    }
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
                for (int i = 0; i < 3; i++) a.d_xyz[3 * (size_t)n + i] += dq[i] * a.sc.inv2[i];
            }
        }
    }
    if (a.lds_floats == 0) return;
    __syncthreads();
    // flush the privatised time planes: rows t0, t1 of plane (a, t) get the tile scaled by the two time weights
    for (int lvl = 0; lvl < p.L; lvl++) {
        const AxisSample st = axis_sample(p.time_scalar, p.res[lvl][3]);
#pragma unroll
        for (int slot = 0; slot < 3; slot++) {
            const int loff = a.lds_off[lvl][slot];
            if (loff < 0) continue;
            const int k = slot == 0 ? 2 : (slot == 1 ? 4 : 5);
            const int Wd = p.res[lvl][slot];
            float* dP = a.d_planes[lvl][k];
            float* r0 = dP + (size_t)st.i0 * Wd * C;
            float* r1 = dP + (size_t)st.i1 * Wd * C;
            for (int i = threadIdx.x; i < Wd * C; i += PG_THREADS) {
                const float v = lds[loff + i];
                if (v != 0.f) {
                    atomicAdd(&r0[i], v * st.w0);
                    atomicAdd(&r1[i], v * st.w1);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ D4, matrix-core splat
// The plane gradient is a SPLAT: dP[texel][c] = sum_n w_n(texel) * dv_n[c] -- per plane a (sparse) [texels x Gaussians]
// weight matrix times the dense [Gaussians x channels] matrix dv.  When the Gaussian set is kept in spatial (Hilbert)
// order (fdgs.densify.spatial_reorder), the G consecutive Gaussians a workgroup takes at a time fall into a window of a
// few texels per axis, and the product over that window is a small DENSE GEMM: it runs on v_mfma_f32_16x16x4_f32
// (M = 16 texels of one window row, N = 16 channels, K = 4 Gaussians; A = wx(texel) * wy(row) built on the fly from three
// numbers per Gaussian and axis), and memory sees ONE atomic line per touched texel of the window instead of one per
// (Gaussian, corner): 24 -> ~6 line-ops per Gaussian at BASELINE config 4.  (Float atomics cost per 64-B line-op,
// ~20 G/s on MI355X whatever the lane count; and with neighbours in the array being neighbours in space the per-corner
// atomics of the kernel above collide on the same lines and get SLOWER, 0.42 -> 0.69 ms.)
//   phase S0  one thread per Gaussian: normalised coordinates -> LDS, window origin per (level, axis) by LDS atomicMin/Max
//   phase S   (per level) lanes <-> (x-corner, channel) as above: sample the six planes, dv_k[c] = dfeat[c] * prod_{k'!=k} v_k'[c]
//             -> LDS, coordinate gradient -> LDS; a Gaussian outside a plane's 16 x 16 window (unsorted input, sparse
//             levels) takes the direct atomics of the kernel above for that plane
//   phase M   (per level) wave k < 6 owns plane k: spatial planes accumulate the window in 16 x 4 accumulator registers
//             (rows no Gaussian of the k-step touches are skipped, wave-uniform) and flush it with one atomic per touched
//             texel line; the time planes (one frame time for all Gaussians: 1-D rows) accumulate 16-texel tiles and add
//             them to the workgroup's private LDS row with plain read-add-writes (one owner wave: no LDS float atomics,
//             which run at 0.33 lanes/clk/CU), flushed once per workgroup with the two time weights.
// Every path adds the same products; only the summation order differs from the per-corner atomics.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

#ifdef FDGS_PROFILE_D4
#define D4_TICK(ph) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof_acc[ph] += t_ - prof_t; prof_t = t_; } while (0)
#else
#define D4_TICK(ph) do { } while (0)
#endif
struct PlaneGradMArgs {
    PlaneGradArgs g;
    unsigned long long* prof;
    const uint32_t* chunks;      // ascending indices of the chunks that contain a live tile (tile_compact_kernel)
    const uint32_t* counters;    // [2] = entries of `chunks`
    const uint32_t* rows;        // row-list form (non-NULL): chunk ci = entries ci * G .. of the live-row list, counters[6] chunks; DFEAT is
                                 // indexed by list position, coordinates and d_xyz by the listed Gaussian; pad entries (ROW_PAD) count as dead
    int off_dv, off_q, off_desc, off_dq, off_org, off_row;   // float offsets into the dynamic LDS
};

__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float f4dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// NWV = 8: one 512-thread workgroup per CU, 2048 / C Gaussians per chunk, two waves per spatial plane (halves of the window rows),
//          the window flush deferred by one phase (see `acc`).
// NWV = 4: 256-thread workgroups with half the chunk, TWO per CU when the LDS allows (one's sampling phase overlaps the other's
//          matrix-core phase, which a single workgroup can only do by idling six of its waves at the barrier); one wave per plane,
//          windows flushed at once (the other workgroup covers the wait).
template <int C, int NWV>
__global__ void __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) deform_plane_grad_mfma_kernel(PlaneGradMArgs ma) {
    constexpr int PGM_T = 64 * NWV;
    const PlaneGradArgs& a = ma.g;
    const fdgs_deform_params& p = a.p;
    extern __shared__ float4 pgm_lds4[];
    float* lds = reinterpret_cast<float*>(pgm_lds4);
    constexpr int G = (NWV == 8 ? 2048 : 1024) / C;   // Gaussians per chunk
    constexpr int NW = NWV;
    constexpr bool DEFER = NWV == 8;
    constexpr int LISTS = NWV == 8 ? 8 : 15, NACC = LISTS + 1;   // row lists per spatial wave / window rows it accumulates
    constexpr int CG = C / 4, LPG = 2 * CG, GPW = 64 / LPG;   // sampling pass: lane = (Gaussian, x-corner, group of 4 channels)
    constexpr int NB = G / (GPW * NW);                        // sampling passes per chunk
    static_assert(NB * GPW * NW == G, "whole sampling passes");
    constexpr int LPGO = 2 * C, GPWO = 64 / LPGO;     // miss pass (per-corner atomics): lane = (Gaussian, x-corner, channel)
    constexpr int NH = C / 16;                        // channel halves of 16
    float* s_dv = lds + ma.off_dv;                    // [6][G][C]
    float* s_q = lds + ma.off_q;                      // [3][G] normalised coordinates
    float4* s_ax = reinterpret_cast<float4*>(lds + ma.off_desc);    // [3][G] {i0 (int bits), weight at i0, weight at i0 + 1, -} of this level
    uint32_t* s_in = reinterpret_cast<uint32_t*>(lds + ma.off_desc + 12 * G);   // [G] bit k: plane k of this level goes through its window
    float* s_dq = lds + ma.off_dq;                    // [G][3]
    float* s_part = lds + ma.off_org;                 // [G / 64][3] per-wave minima of the coordinates
    // Gaussians of the chunk binned by window row, once per row axis (y for plane (x,y); z for planes (x,z), (y,z)): the
    // matrix-core loop walks one row's list at a time, so its accumulators are static registers and a step is two MFMAs
    int* s_cnt_all = reinterpret_cast<int*>(lds + ma.off_org + 16);  // [2 (level parity)][40]: [2][16] row counts + miss count; zeroed one level ahead
    uint32_t* s_miss = reinterpret_cast<uint32_t*>(lds + ma.off_org + 96);     // [G] Gaussian | planes that take the per-corner atomics << 8
    uint8_t* s_list = reinterpret_cast<uint8_t*>(lds + ma.off_org + 96 + G);   // [2][16][G]
    uint32_t* s_row = reinterpret_cast<uint32_t*>(lds + ma.off_row);           // [G] row-list form: the chunk's list entries
    const bool by_rows = ma.rows != nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = lane % CG, xc = (lane / CG) & 1, gl_s = wave * GPW + lane / LPG;
    // window of the previous phase, kept in registers: its atomics are issued at the START of the next matrix-core loop and drain
    // while that loop runs (loads, stores and no-return atomics share one in-order counter: a sampling load issued right after a
    // flush could only be waited for together with the whole flush)
    f32x4v acc[NACC];
    uint32_t pend_rows = 0;
    float* pend_dP = nullptr;
    int pend_ox = 0, pend_oy = 0, pend_Wd = 0, pend_Hd = 0, pend_hf = 0;
    for (int i = tid; i < a.lds_floats; i += PGM_T) lds[i] = 0.f;     // private time rows
#ifdef FDGS_PROFILE_D4
    unsigned long long prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long prof_t = __builtin_amdgcn_s_memtime();
    const unsigned long long prof_t0 = prof_t;
#endif
    const int nchunks = (int)as_const(ma.counters)[2];
    int c_begin = (int)((long long)blockIdx.x * nchunks / gridDim.x);         // contiguous runs of the chunk list: spatial locality
    int c_end = (int)((long long)(blockIdx.x + 1) * nchunks / gridDim.x);
    // row-list form: every workgroup takes an EQUAL share of the list's entries (rounded to 8), cut into chunks of at most G -- not whole
    // chunks: 283 chunks on 256 workgroups cost two chunk times, 128 + 16 entries cost about 1.4 (a short chunk runs one sampling pass
    // and short matrix-core loops)
    int e0 = 0, e1 = 0;
    if (by_rows) {
        const int tot = (int)as_const(ma.counters)[5] * 32;
        int per = (tot + (int)gridDim.x - 1) / (int)gridDim.x;
        per = (per + 7) & ~7;
        e0 = (int)blockIdx.x * per; e0 = e0 < tot ? e0 : tot;
        e1 = e0 + per < tot ? e0 + per : tot;
        c_begin = 0; c_end = (e1 - e0 + G - 1) / G;
    }
    const float tq = p.time_scalar;
    for (int ci = c_begin; ci < c_end; ci++) {
        const int chunk = by_rows ? ci : (int)as_const(ma.chunks)[ci];
        const int n0 = by_rows ? e0 + ci * G : chunk * G;        // (row-list form: first list position)
        const int cnt = by_rows ? (e1 - n0 < G ? e1 - n0 : G) : G;      // entries of this chunk
        const int nb_dyn = by_rows ? (cnt + GPW * NW - 1) / (GPW * NW) : NB;      // sampling passes that hold an entry
        // ---- S0: coordinates -> LDS, per-axis minimum over the chunk (the texel index is monotonic in the coordinate, so the
        // window origin of every level follows from the three minima)
        if (tid >= PGM_T - 40) s_cnt_all[tid - (PGM_T - 40)] = 0;
        if (tid < G) {
            const int n = n0 + tid;
            const uint32_t ent = by_rows ? (tid < cnt ? ma.rows[n] : ROW_PAD) : 0u;       // (behind the chunk's end: counts as padding everywhere below)
            if (by_rows) s_row[tid] = ent;
            const int nn = by_rows ? (int)(ent & ~ROW_PAD) : (n < p.N ? n : p.N - 1);
            const bool live = by_rows ? !(ent & ROW_PAD) : (n < p.N && a.tile_live[nn >> 5] != 0u);   // (rows of dead tiles do not stretch the window)
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const float q = (p.xyz[3 * (size_t)nn + i] - p.aabb[i]) * a.sc.inv2[i] - 1.0f;
                s_q[i * G + tid] = q;
                s_dq[tid * 3 + i] = 0.f;
                float m = live ? q : 3.0e38f;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
                if (lane == 0) s_part[wave * 3 + i] = m;
            }
        }
        __syncthreads();
        float qmin[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            float m = s_part[i];
#pragma unroll
            for (int w = 1; w < G / 64; w++) m = fminf(m, s_part[w * 3 + i]);
            qmin[i] = m;
        }
        D4_TICK(0);
        for (int lvl = 0; lvl < p.L; lvl++) {
            const int org0 = axis_sample(qmin[0], p.res[lvl][0]).i0, org1 = axis_sample(qmin[1], p.res[lvl][1]).i0;
            const int org2 = axis_sample(qmin[2], p.res[lvl][2]).i0;
            const int lo0 = a.lds_off[lvl][0], lo1 = a.lds_off[lvl][1], lo2 = a.lds_off[lvl][2];
            int* s_cnt = s_cnt_all + (lvl & 1) * 40;      // (zeroed during the previous level's M phase / in S0)
            // ---- S: lane = (Gaussian, x-corner, 4 channels): 16-byte texel loads, the per-Gaussian index / weight arithmetic is
            // shared by four channels.  dv -> LDS, coordinate gradient -> LDS, window bookkeeping.
#pragma nounroll
            for (int b = 0; b < nb_dyn; b++) {
                const int gl = b * (GPW * NW) + gl_s;
                const int n = n0 + gl;
                const bool live = by_rows ? !(s_row[gl] & ROW_PAD) : (n < p.N && a.tile_live[n >> 5] != 0u);     // (a dead tile's DFEAT rows were never written)
                float q[4];
                q[0] = s_q[gl]; q[1] = s_q[G + gl]; q[2] = s_q[2 * G + gl]; q[3] = tq;
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 df = *reinterpret_cast<const float4*>(a.DFEAT + (size_t)(by_rows || n < p.N ? n : p.N - 1) * a.F + lvl * C + 4 * cg);
                if (!live) df = z4;
                float4 vk[6], sk[6], tk[6];
                float dsx[6], dsy[6];
                AxisSample S[4];
#pragma unroll
                for (int ax4 = 0; ax4 < 4; ax4++) S[ax4] = axis_sample(q[ax4], p.res[lvl][ax4]);
#pragma unroll
                for (int grp = 0; grp < 2; grp++) {       // three planes (six 16-byte texel requests) at a time: bounded registers
#pragma unroll
                    for (int kk = 0; kk < 3; kk++) {
                        const int k = 3 * grp + kk;
                        int ax, bx;
                        plane_axes(k, ax, bx);
                        const int Wd = p.res[lvl][ax];
                        const AxisSample sx = S[ax], sy = S[bx];
                        const int xi = xc ? sx.i1 : sx.i0;
                        const float wx = xc ? sx.w1 : sx.w0;
                        const uint32_t oA = (uint32_t)((sy.i0 * Wd + xi) * C + 4 * cg), oB = (uint32_t)((sy.i1 * Wd + xi) * C + 4 * cg);
                        const char* Pb = reinterpret_cast<const char*>(p.planes[lvl][k]);
                        const float4 v0 = *reinterpret_cast<const float4*>(Pb + oA * 4u);
                        const float4 v1 = *reinterpret_cast<const float4*>(Pb + oB * 4u);
                        sk[k] = make_float4(sy.w0 * v0.x + sy.w1 * v1.x, sy.w0 * v0.y + sy.w1 * v1.y, sy.w0 * v0.z + sy.w1 * v1.z, sy.w0 * v0.w + sy.w1 * v1.w);
                        tk[k] = make_float4(wx * (v1.x - v0.x), wx * (v1.y - v0.y), wx * (v1.z - v0.z), wx * (v1.w - v0.w));
                        const float4 part = f4scale(sk[k], wx);
                        vk[k] = make_float4(part.x + __shfl_xor(part.x, CG, 64), part.y + __shfl_xor(part.y, CG, 64),
                                            part.z + __shfl_xor(part.z, CG, 64), part.w + __shfl_xor(part.w, CG, 64));
                        dsx[k] = sx.dscale; dsy[k] = sy.dscale;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                float4 suf[6];        // suf[k] = prod_{k' > k} v_k'; the prefix product runs along with the plane loop below
                suf[5] = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
                for (int k = 4; k >= 0; k--) suf[k] = f4mul(suf[k + 1], vk[k + 1]);
                float4 pre = suf[5];
                // which planes of this Gaussian go through a window: spatial planes when both axes lie within 15 texels of the
                // chunk's minimum; time planes (private rows) when the texel pair lies in the two 16-texel tiles that start at
                // the tile of the chunk's minimum.  Anything else takes the per-corner atomics in the miss pass below.
                const int d0 = S[0].i0 - org0, d1 = S[1].i0 - org1, d2 = S[2].i0 - org2;
                const bool in0 = d0 <= 14, in1 = d1 <= 14, in2 = d2 <= 14;
                const bool t0 = lo0 >= 0 && S[0].i0 - (org0 & ~15) <= 30, t1 = lo1 >= 0 && S[1].i0 - (org1 & ~15) <= 30;
                const bool t2 = lo2 >= 0 && S[2].i0 - (org2 & ~15) <= 30;
                uint32_t inw = 0, want = 0;
                if (live) {
                    inw = (in0 && in1 ? 1u : 0u) | (in0 && in2 ? 2u : 0u) | (t0 ? 4u : 0u) | (in1 && in2 ? 8u : 0u) | (t1 ? 16u : 0u) | (t2 ? 32u : 0u);
#pragma unroll
                    for (int k = 0; k < 6; k++) want |= a.d_planes[lvl][k] ? (1u << k) : 0u;
                }
                inw &= want;
                const uint32_t miss = want & ~inw;
                float dq[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    int ax, bx;
                    plane_axes(k, ax, bx);
                    const float4 dv = f4mul(df, f4mul(pre, suf[k]));
                    pre = f4mul(pre, vk[k]);
                    const bool wk = (inw >> k) & 1u;      // (element-wise selects: a float4 ?: goes through scratch memory)
                    if (xc == 0)
                        *reinterpret_cast<float4*>(s_dv + (k * G + gl) * C + 4 * cg) = make_float4(wk ? dv.x : 0.f, wk ? dv.y : 0.f, wk ? dv.z : 0.f, wk ? dv.w : 0.f);
                    const float gx = (xc ? 1.f : -1.f) * f4dot(dv, sk[k]) * dsx[k];
                    const float gy = f4dot(dv, tk[k]) * dsy[k];
                    if (ax < 3) dq[ax] += gx;
                    if (bx < 3) dq[bx] += gy;
                }
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    float v = dq[i];
#pragma unroll
                    for (int o = 1; o < LPG; o <<= 1) v += __shfl_xor(v, o, 64);
                    dq[i] = v;
                }
                if ((lane % LPG) == 0) {
                    s_in[gl] = inw;
                    if (inw & 1u) s_list[(0 * 16 + d1) * G + atomicAdd(&s_cnt[d1], 1)] = (uint8_t)gl;            // rows of plane (x,y): y
                    if (inw & 10u) s_list[(1 * 16 + d2) * G + atomicAdd(&s_cnt[16 + d2], 1)] = (uint8_t)gl;      // rows of (x,z), (y,z): z
                    if (miss) s_miss[atomicAdd(&s_cnt[32], 1)] = (uint32_t)gl | (miss << 8);
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const bool edge = S[i].i1 == S[i].i0;      // clamped at the last texel: both corners are the same texel
                        s_ax[i * G + gl] = make_float4(__int_as_float(S[i].i0), edge ? S[i].w0 + S[i].w1 : S[i].w0, edge ? 0.f : S[i].w1, 0.f);
                        s_dq[gl * 3 + i] += dq[i];
                    }
                }
            }
            D4_TICK(1);
            __syncthreads();
            D4_TICK(2);
            if (wave == NW - 1 && lane < 40) s_cnt_all[((lvl + 1) & 1) * 40 + lane] = 0;
            // ---- M: waves 0..5 = (spatial plane, half of the window rows), waves 6, 7 = time planes
            const int gk = lane >> 4, il = lane & 15;         // Gaussian of the k-step / texel of the tile (A), channel (B)
            if (wave < (NWV == 8 ? 6 : 3)) {
                const int pi = NWV == 8 ? wave >> 1 : wave, hh = NWV == 8 ? (wave & 1) : 0;
                const int k = pi == 2 ? 3 : pi;
                float* dP = a.d_planes[lvl][k];
                if (dP) {
                    int ax, bx;
                    plane_axes(k, ax, bx);
                    const int ox = ax == 0 ? org0 : org1, oy = bx == 1 ? org1 : org2;
                    const int bin = k == 0 ? 0 : 1;
                    const uint32_t kbit = 1u << k;
                    // the two waves of a plane split the row lists where the Gaussian count is halved (the rows fill up from the
                    // window origin: a fixed split at row 8 left one wave with 80 % of the work): lists [0, rs) and [rs, 15), each at
                    // most LISTS long -- rs in [last - LISTS + 1, LISTS], `last` = last non-empty list
                    int base = 0, lend = 15;
                    if (NWV == 8) {
                        const int c = lane < 15 ? s_cnt[bin * 16 + lane] : 0;
                        int pre = c;
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) { const int u = __shfl_up(pre, o, 64); if ((lane & 15) >= o) pre += u; }
                        const int total = __shfl(pre, 15, 64);
                        const uint64_t half = __ballot(lane < 16 && 2 * pre >= total);          // first list whose prefix reaches half
                        const uint64_t nonz = __ballot(lane < 16 && c > 0);
                        const int last = nonz ? 63 - __builtin_clzll(nonz) : 0;
                        int rs = half ? __builtin_ctzll(half) + 1 : LISTS;
                        const int lo = last - LISTS + 1;
                        rs = rs < lo ? lo : rs;
                        rs = rs > LISTS ? LISTS : rs;
                        rs = rs < 1 ? 1 : rs;
                        base = hh ? rs : 0;
                        lend = hh ? 15 : rs;
                    }
#pragma nounroll
                    for (int hf = 0; hf < NH; hf++) {
                        // the previous window of this wave: accumulator register j of lane (gk, il) is texel x = ox + 4 gk + j of window
                        // row base + rr, channel il
                        if (pend_rows) {
#pragma unroll
                            for (int rr = 0; rr < NACC; rr++) {
                                if ((pend_rows >> rr) & 1u) {
                                    const int y = pend_oy + rr;
#pragma unroll
                                    for (int j = 0; j < 4; j++) {
                                        const int x = pend_ox + 4 * gk + j;
                                        const float v = acc[rr][j];
                                        if (v != 0.f && x < pend_Wd && y < pend_Hd) atomicAdd(&pend_dP[((size_t)y * pend_Wd + x) * C + pend_hf * 16 + il], v);
                                    }
                                }
                            }
                        }
                        D4_TICK(4);
#pragma unroll
                        for (int r = 0; r < NACC; r++) acc[r] = f32x4v{0.f, 0.f, 0.f, 0.f};
                        uint32_t rows_any = 0;
#pragma unroll
                        for (int rr = 0; rr < LISTS; rr++) {
                            const int r = base + rr;                   // list of window row r feeds rows r and r + 1 (r <= 14)
                            const int nr = r < lend ? s_cnt[bin * 16 + r] : 0;
                            const uint8_t* lst = s_list + (bin * 16 + r) * G;
                            // two steps (8 Gaussians) per iteration; the list bytes of the next iteration are requested before this
                            // iteration's operands, and all operand requests before the first MFMA
                            int ga = gk < nr ? (int)lst[gk] : 0, gb = 4 + gk < nr ? (int)lst[4 + gk] : 0;
                            for (int j0 = 0; j0 < nr; j0 += 8) {
                                const bool ha = j0 + gk < nr, hb = j0 + 4 + gk < nr;
                                const int gla = ga, glb = gb;
                                ga = j0 + 8 + gk < nr ? (int)lst[j0 + 8 + gk] : 0;
                                gb = j0 + 12 + gk < nr ? (int)lst[j0 + 12 + gk] : 0;
                                const float4 cxa = s_ax[ax * G + gla], cya = s_ax[bx * G + gla];
                                const float4 cxb = s_ax[ax * G + glb], cyb = s_ax[bx * G + glb];
                                const uint32_t cia = s_in[gla], cib = s_in[glb];
                                const float bra = s_dv[(k * G + gla) * C + hf * 16 + il], brb = s_dv[(k * G + glb) * C + hf * 16 + il];
                                __builtin_amdgcn_sched_barrier(0);
                                // (planes (x,z) and (y,z) share the z lists: an entry counts for this plane only if its own window test passed)
                                const float bva = (ha && (cia & kbit)) ? bra : 0.f, bvb = (hb && (cib & kbit)) ? brb : 0.f;
                                const int dxa = __float_as_int(cxa.x) - ox, dxb = __float_as_int(cxb.x) - ox;
                                const float wxa = il == dxa ? cxa.y : (il == dxa + 1 ? cxa.z : 0.f);
                                const float wxb = il == dxb ? cxb.y : (il == dxb + 1 ? cxb.z : 0.f);
                                acc[rr] = mfma16(wxa * cya.y, bva, acc[rr]);
                                acc[rr + 1] = mfma16(wxa * cya.z, bva, acc[rr + 1]);
                                acc[rr] = mfma16(wxb * cyb.y, bvb, acc[rr]);
                                acc[rr + 1] = mfma16(wxb * cyb.z, bvb, acc[rr + 1]);
                            }
                            if (nr > 0) rows_any |= 3u << rr;
                        }
                        pend_rows = rows_any; pend_dP = dP; pend_ox = ox; pend_oy = oy + base; pend_Wd = p.res[lvl][ax]; pend_Hd = p.res[lvl][bx];
                        pend_hf = hf;
                        D4_TICK(3);
                        if (!DEFER && pend_rows) {
#pragma unroll
                            for (int rr = 0; rr < NACC; rr++) {
                                if ((pend_rows >> rr) & 1u) {
                                    const int y = pend_oy + rr;
#pragma unroll
                                    for (int j = 0; j < 4; j++) {
                                        const int x = pend_ox + 4 * gk + j;
                                        const float v = acc[rr][j];
                                        if (v != 0.f && x < pend_Wd && y < pend_Hd) atomicAdd(&pend_dP[((size_t)y * pend_Wd + x) * C + pend_hf * 16 + il], v);
                                    }
                                }
                            }
                            pend_rows = 0;
                        }
                    }
                }
            } else if (NWV == 8 || wave == 3) {
                for (int slot = NWV == 4 ? 0 : (wave == 6 ? 0 : 2); slot < (NWV == 4 ? 3 : (wave == 6 ? 2 : 3)); slot++) {
                    const int k = slot == 0 ? 2 : (slot == 1 ? 4 : 5);
                    const int loff = slot == 0 ? lo0 : (slot == 1 ? lo1 : lo2);
                    if (loff < 0) continue;
                    const int ax = slot;
                    const int Wd = p.res[lvl][ax];
                    float* row = lds + loff;                              // [Wd][C] private to the workgroup
                    const int x0 = (slot == 0 ? org0 : (slot == 1 ? org1 : org2)) & ~15;   // two 16-texel tiles from the tile of the chunk's minimum
#pragma nounroll
                    for (int hf = 0; hf < NH; hf++) {
                        f32x4v acc0 = f32x4v{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4v{0.f, 0.f, 0.f, 0.f};
                        float4 nxa = s_ax[ax * G + gk], nxb = s_ax[ax * G + 4 + gk];
                        float nba = s_dv[(k * G + gk) * C + hf * 16 + il], nbb = s_dv[(k * G + 4 + gk) * C + hf * 16 + il];
                        const int ksn = nb_dyn * (GPW * NW) / 4;         // (entries of passes that did not run hold the previous chunk's values)
                        for (int ks = 0; ks < ksn; ks += 2) {
                            const float4 cxa = nxa, cxb = nxb;
                            const float bva = nba, bvb = nbb;
                            const int gn = 4 * (ks + 2 < ksn ? ks + 2 : ks) + gk;
                            nxa = s_ax[ax * G + gn]; nxb = s_ax[ax * G + gn + 4];
                            nba = s_dv[(k * G + gn) * C + hf * 16 + il]; nbb = s_dv[(k * G + gn + 4) * C + hf * 16 + il];
                            __builtin_amdgcn_sched_barrier(0);
                            const int ra = __float_as_int(cxa.x) - x0, rb = __float_as_int(cxb.x) - x0;    // (Gaussians outside the two tiles have dv = 0)
                            // a tile that none of the iteration's eight Gaussians touches is skipped (wave-uniform): a compact chunk
                            // usually sits inside one of the two
                            if (__ballot(ra <= 15 || rb <= 15)) {
                                acc0 = mfma16(il == ra ? cxa.y : (il == ra + 1 ? cxa.z : 0.f), bva, acc0);
                                acc0 = mfma16(il == rb ? cxb.y : (il == rb + 1 ? cxb.z : 0.f), bvb, acc0);
                            }
                            if (__ballot((ra >= 15 && ra <= 31) || (rb >= 15 && rb <= 31))) {
                                acc1 = mfma16(il + 16 == ra ? cxa.y : (il + 16 == ra + 1 ? cxa.z : 0.f), bva, acc1);
                                acc1 = mfma16(il + 16 == rb ? cxb.y : (il + 16 == rb + 1 ? cxb.z : 0.f), bvb, acc1);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int xa = x0 + 4 * gk + j, xb = xa + 16;
                            if (xa < Wd && acc0[j] != 0.f) row[xa * C + hf * 16 + il] += acc0[j];
                            if (xb < Wd && acc1[j] != 0.f) row[xb * C + hf * 16 + il] += acc1[j];
                        }
                    }
                }
                D4_TICK(5);
            }
            // ---- miss pass: the (Gaussian, plane) pairs outside their window take the per-corner atomics, lane = (Gaussian, x-corner,
            // channel) so that every atomic instruction covers whole texel lines.  Rare on spatially ordered input; on unordered input
            // it is the whole plane gradient (the windows only catch what happens to lie near the chunk's minimum).  The time rows it
            // touches are LDS atomics on rows that only their owner wave (above) writes with plain adds: this pass therefore runs
            // after a barrier when a time plane is among the misses.
            const int nmiss = s_cnt[32];
            if (nmiss > 0) {
                __syncthreads();       // (uniform: nmiss is the same for every thread)
                const int cho = lane % C, xco = (lane / C) & 1, gso = lane / LPGO;
                for (int e0 = 0; e0 < nmiss; e0 += NW * GPWO) {
                    const int e = e0 + wave * GPWO + gso;
                    if (e < nmiss) {
                        const uint32_t ent = s_miss[e];
                        const int gl = (int)(ent & 0xFFu);
                        const uint32_t mm = ent >> 8;
                        const int n = n0 + gl;
                        float q[4];
                        q[0] = s_q[gl]; q[1] = s_q[G + gl]; q[2] = s_q[2 * G + gl]; q[3] = tq;
                        const float dfo = a.DFEAT[(size_t)n * a.F + lvl * C + cho];
                        float vko[6], wA[6], wB[6], wX[6];
                        uint32_t oA[6], oB[6], oX[6];
                        AxisSample S[4];
#pragma unroll
                        for (int ax4 = 0; ax4 < 4; ax4++) S[ax4] = axis_sample(q[ax4], p.res[lvl][ax4]);
#pragma unroll
                        for (int k = 0; k < 6; k++) {
                            int ax, bx;
                            plane_axes(k, ax, bx);
                            const int Wd = p.res[lvl][ax];
                            const AxisSample sx = S[ax], sy = S[bx];
                            const int xi = xco ? sx.i1 : sx.i0;
                            const float wx = xco ? sx.w1 : sx.w0;
                            oX[k] = (uint32_t)(xi * C + cho);
                            oA[k] = (uint32_t)((sy.i0 * Wd + xi) * C + cho);
                            oB[k] = (uint32_t)((sy.i1 * Wd + xi) * C + cho);
                            const char* Pb = reinterpret_cast<const char*>(p.planes[lvl][k]);
                            const float v0 = *reinterpret_cast<const float*>(Pb + oA[k] * 4u);
                            const float v1 = *reinterpret_cast<const float*>(Pb + oB[k] * 4u);
                            float part = wx * (sy.w0 * v0 + sy.w1 * v1);
                            if (C == 16) {
                                auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(part), __float_as_uint(part), false, false);
                                part = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                            } else {
                                auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
                                part = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                            }
                            vko[k] = part;
                            wX[k] = wx; wA[k] = wx * sy.w0; wB[k] = wx * sy.w1;
                        }
                        float preo[6], sufo[6];
                        preo[0] = 1.f; sufo[5] = 1.f;
#pragma unroll
                        for (int k = 1; k < 6; k++) preo[k] = preo[k - 1] * vko[k - 1];
#pragma unroll
                        for (int k = 4; k >= 0; k--) sufo[k] = sufo[k + 1] * vko[k + 1];
#pragma unroll
                        for (int k = 0; k < 6; k++) {
                            if ((mm >> k) & 1u) {
                                const float dv = dfo * preo[k] * sufo[k];
                                const int slot = time_plane_slot(k);
                                const int loff = slot == 0 ? lo0 : (slot == 1 ? lo1 : (slot == 2 ? lo2 : -1));
                                if (loff >= 0) {
                                    atomicAdd(&lds[loff + oX[k]], dv * wX[k]);      // private time row (ds_add_f32)
                                } else {
                                    float* dP = a.d_planes[lvl][k];
                                    atomicAdd(&dP[oA[k]], dv * wA[k]);
                                    atomicAdd(&dP[oB[k]], dv * wB[k]);
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();
            D4_TICK(6);
        }
        if (by_rows) {
            if (a.d_xyz && tid < G && !(s_row[tid] & ROW_PAD)) {
                const size_t nr = (size_t)s_row[tid];
#pragma unroll
                for (int i = 0; i < 3; i++) a.d_xyz[3 * nr + i] += s_dq[tid * 3 + i] * a.sc.inv2[i];
            }
        } else if (a.d_xyz && tid < G && n0 + tid < p.N) {
#pragma unroll
            for (int i = 0; i < 3; i++) a.d_xyz[3 * (size_t)(n0 + tid) + i] += s_dq[tid * 3 + i] * a.sc.inv2[i];
        }
        __syncthreads();
        D4_TICK(7);
    }
    if (pend_rows) {          // the last window of this wave (deferred flush)
        const int gk = lane >> 4, il = lane & 15;
#pragma unroll
        for (int rr = 0; rr < NACC; rr++) {
            if ((pend_rows >> rr) & 1u) {
                const int y = pend_oy + rr;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int x = pend_ox + 4 * gk + j;
                    const float v = acc[rr][j];
                    if (v != 0.f && x < pend_Wd && y < pend_Hd) atomicAdd(&pend_dP[((size_t)y * pend_Wd + x) * C + pend_hf * 16 + il], v);
                }
            }
        }
    }
#ifdef FDGS_PROFILE_D4
    if (ma.prof && lane == 0 && blockIdx.x < 64) {
        unsigned long long* out = ma.prof + (size_t)(blockIdx.x * 8 + wave) * 10;
        for (int i = 0; i < 8; i++) out[i] = prof_acc[i];
        out[9] = __builtin_amdgcn_s_memtime() - prof_t0;
    }
#endif
    if (a.lds_floats == 0) return;
    // flush the private time rows: rows t0, t1 of plane (axis, t) get the row scaled by the two time weights
    for (int lvl = 0; lvl < p.L; lvl++) {
        const AxisSample st = axis_sample(p.time_scalar, p.res[lvl][3]);
#pragma unroll
        for (int slot = 0; slot < 3; slot++) {
            const int loff = a.lds_off[lvl][slot];
            if (loff < 0) continue;
            const int k = slot == 0 ? 2 : (slot == 1 ? 4 : 5);
            const int Wd = p.res[lvl][slot];
            float* dP = a.d_planes[lvl][k];
            float* r0 = dP + (size_t)st.i0 * Wd * C;
            float* r1 = dP + (size_t)st.i1 * Wd * C;
            for (int i = tid; i < Wd * C; i += PGM_T) {
                const float v = lds[loff + i];
                if (v != 0.f) {
                    atomicAdd(&r0[i], v * st.w0);
                    atomicAdd(&r1[i], v * st.w1);
                }
            }
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
        for (int i = 0; i < 4; i++) FDGS_REQUIRE(p->res[l][i] >= 2 && p->res[l][i] <= 4096, "plane resolution must be in [2, 4096]");
        for (int k = 0; k < 6; k++) {
            const int a = k < 3 ? 0 : (k < 5 ? 1 : 2), b = k < 3 ? k + 1 : (k < 5 ? k - 1 : 3);
            FDGS_REQUIRE((long long)p->res[l][a] * p->res[l][b] * p->C * 4 < (1ll << 31), "plane larger than 2 GiB");
        }
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
#ifdef FDGS_DEV_ONLY_44   // development builds: only the (net_width 128, C*L 32) instance, for quick ISA inspection
    FDGS_CASE(4, 4)
#else
    FDGS_CASE(2, 4) FDGS_CASE(2, 6) FDGS_CASE(2, 8) FDGS_CASE(2, 12) FDGS_CASE(2, 16)
    FDGS_CASE(4, 4) FDGS_CASE(4, 6) FDGS_CASE(4, 8) FDGS_CASE(4, 12) FDGS_CASE(4, 16)
#endif
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
struct Fwd16Launcher {      // (the caller only selects this form when FCH is even: C*L % 16 == 0)
    static void go(hipStream_t s, int blocks, const DeformDev& d) {
        if constexpr ((FCH % 2) == 0) hipLaunchKernelGGL((deform_fwd16_kernel<2 * WT, FCH / 2>), dim3(blocks), dim3(256), 0, s, d);
        else hipLaunchKernelGGL((deform_fwd_kernel<WT, FCH>), dim3(blocks), dim3(256), 0, s, d);
    }
};
template <int WT, int FCH, bool SAVED, bool ROWS = false>
static void launch_bwd_data(hipStream_t s, int max_blocks, const BwdDev& d) {
    // persistent: as many workgroups as are co-resident (each keeps dW2/db2 sums in LDS), tiles handed out round-robin
    static int resident = 0;
    if (resident == 0) {
        int dev = 0, cus = 256, per_cu = 1;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, deform_bwd_data_kernel<WT, FCH, SAVED, ROWS>, 256, 0) != hipSuccess || per_cu < 1)
            per_cu = 1;
        resident = cus * per_cu;
    }
    int blocks = resident;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((deform_bwd_data_kernel<WT, FCH, SAVED, ROWS>), dim3(blocks), dim3(256), 0, s, d);
}
template <int WT, int FCH>
struct BwdLauncher {
    static void go(hipStream_t s, int max_blocks, const BwdDev& d) {
        // the weight-stationary form (deform_bwd_ws.h) where it applies: row lists, net_width 128, C*L in {32, 48}, all five heads (dynerf) or the
        // three of position / scale / rotation (hypernerf, dnerf)
        if constexpr (WT == 4 && (FCH % 2) == 0 && FCH <= 6) {
            int mask = 0;
            for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) mask |= d.p.head_on[hd] ? 1 << hd : 0;
            if (d.sv_h1 && d.s.rows && (mask == 31 || mask == 7) && g_tune.d2_form != 32) {
                static int cus = 0;
                if (cus == 0) {
                    int dev = 0;
                    (void)hipGetDevice(&dev);
                    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
                }
                int blocks = cus;
                if (blocks > max_blocks * 8) blocks = max_blocks * 8;      // (never more workgroups than 16-row tiles)
                if (mask == 31) hipLaunchKernelGGL((deform_bwd_data_ws_kernel<FCH / 2, 31>), dim3(blocks), dim3(256), 0, s, d);
                else hipLaunchKernelGGL((deform_bwd_data_ws_kernel<FCH / 2, 7>), dim3(blocks), dim3(256), 0, s, d);
                return;
            }
        }
        if (d.sv_h1 && d.s.rows) launch_bwd_data<WT, FCH, true, true>(s, max_blocks, d);
        else if (d.sv_h1) launch_bwd_data<WT, FCH, true>(s, max_blocks, d);
        else launch_bwd_data<WT, FCH, false>(s, max_blocks, d);
    }
};

static size_t npad_of(int N) { return ((size_t)(N > 0 ? N : 1) + 127) / 128 * 128; }
static int active_heads(const fdgs_deform_params* p) {
    int c = 0;
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) c += p->head_on[hd] ? 1 : 0;
    return c;
}
// activations the forward can leave behind for the backward (float offsets into the `saved` buffer)
struct SavedLayout { size_t Np, feat, rh, h1, hmask, floats; };
static SavedLayout saved_layout(const fdgs_deform_params* p) {
    SavedLayout s;
    s.Np = npad_of(p->N);
    const size_t F = (size_t)p->C * p->L, W = p->W;
    s.feat = 0; s.rh = s.feat + s.Np * F; s.h1 = s.rh + s.Np * W;
    s.hmask = s.h1 + s.Np * W * active_heads(p);
    s.floats = s.hmask + s.Np * 8;   // 64 lanes x 4 words per 32 Gaussians
    return s;
}

// scratch of fdgs_deform_bwd (float offsets): the packed gradient rows, DIRECTLY followed by the per-tile flags (the layout
// include/fdgs.h promises to fdgs_raster_bwd's epilogue), the lists built from them, then the inter-kernel arrays
struct BwdLayout { size_t G, flags, live, chunks, counters, DH1, DHID, RH, FEAT, DFEAT, rowbase, rows, Gc, floats; };
static BwdLayout bwd_layout(const fdgs_deform_params* p) {
    BwdLayout b;
    const size_t Np = npad_of(p->N), F = (size_t)p->C * p->L, W = p->W, nt = Np / 32;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t r = o; o = (o + n + 63) / 64 * 64; return r; };
    b.G = take(Np * GCOLS); b.flags = take(nt); b.live = take(nt + 4); b.chunks = take(nt); b.counters = take(64);
    b.DH1 = take(Np * W * (size_t)active_heads(p)); b.DHID = take(Np * W); b.RH = take(Np * W); b.FEAT = take(Np * F); b.DFEAT = take(Np * F);
    // row-list form (behind everything the older layout promised): live rows in front of each tile, the list, the compact copy of G
    b.rowbase = take(nt); b.rows = take(Np + 256); b.Gc = take((Np + 256) * GCOLS);
    b.floats = o;
    return b;
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
    d.p = *p; d.out = *out; d.F = p->C * p->L; d.small_heads = 1; d.split_tail = g_tune.d1_split; d.sc = aabb_scale(p);
    {
        const SavedLayout sl = saved_layout(p);
        float* sv = reinterpret_cast<float*>(out->saved);
        d.sv_feat = sv ? sv + sl.feat : nullptr; d.sv_rh = sv ? sv + sl.rh : nullptr; d.sv_h1 = sv ? sv + sl.h1 : nullptr;
        d.sv_hmask = sv ? reinterpret_cast<uint32_t*>(sv + sl.hmask) : nullptr;
        d.Npad = (int)sl.Np;
        int slot = 0;
        for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) d.head_slot[hd] = p->head_on[hd] ? slot++ : 0;
    }
    {
        d.ntiles = 4 * cdiv(p->N, 128);
        static int cus = 0;
        if (!cus) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256; }
        // 16-Gaussian form (deform_fwd16.h): two workgroups per CU, needs whole float4 texel quarters per lane group (C % 16 == 0)
        // which form of the forward kernel: 16 (default where it applies: two waves per SIMD; in the frame, where the gather starts on cold
        // caches behind the previous frame's backward, it measured 5 - 7 % faster than the 32-form on every workload, profiles/r04_d1_forms.txt)
        // | 32 (also what runs when C*L is not a multiple of 16 or the caller hands over no pack scratch)
        // which form of the forward kernel: 0 (default) = by shape -- form 8 at net_width 128 with up to two HexPlane levels, form 16 otherwise (config 2's 0.8-ms frame is paced by the host: one more launch costs more than the kernel gains): the gather of form 8 is a
        // kernel of its own whose time grows with the levels while form 16 hides it under its products (measured, profiles/r06_d1_forms_by_config.txt:
        // config 3, three levels: 0.432 + 0.062 ms against 0.462 + 0.006; configs 4 / 5, two levels: 0.570 + 0.039 against 0.633, 3.70 + 0.21 against 4.11);
        // 8 (forced where it applies: the weight-stationary form of deform_fwd_ws.h -- net_width 64 / 128,
        // C*L a multiple of 16; the gather runs as a kernel of its own in front of it) | 16 (two waves per SIMD on packed operand streams)
        // | 32 (also what runs when C*L is not a multiple of 16 or the caller hands over no scratch)
        const bool formws = (g_tune.d1_form == 8 || (g_tune.d1_form == 0 && p->L <= 2 && p->W == 128)) && (p->W == 64 || (p->W == 128 && d.F <= 48)) && p->C % 4 == 0 && d.F % 16 == 0 && (out->saved || out->packed);   // (net_width 128 with C*L > 48: W0 no longer fits the registers next to the five W1 -- and this kernel must not spill)
        const bool form16 = !formws && g_tune.d1_form != 32 &&   /* (d1_form 0 = by shape: form 8 up to two HexPlane levels, form 16 above) */ p->C % 16 == 0 && d.F % 16 == 0 && out->packed != nullptr;
        d.packed = reinterpret_cast<const float*>(out->packed);
        d.feat = nullptr;
        d.head_mask = 0u;
        for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) d.head_mask |= p->head_on[hd] ? 1u << hd : 0u;
        d.skew = 600;       // (s_memtime ticks; sweep 0 .. 24 k in profiles/r04_d1_forms.txt)
        if (formws) {
            // features: into the saved activations when the backward will want them, into the scratch otherwise
            float* feat = d.sv_feat ? d.sv_feat : reinterpret_cast<float*>(out->packed);
            GatherArgs ga{};
            ga.p = *p; ga.sc = d.sc; ga.F = d.F; ga.Npad = d.Npad; ga.feat = feat;
            d.feat = feat;
            FDGS_TIMED("deform_gather", stream);
            hipLaunchKernelGGL(deform_gather_kernel, dim3(cdiv((long long)d.Npad * (p->C / 4), 256), p->L), dim3(256), 0, stream, ga);
        }
        if (form16) {
            FDGS_TIMED("pack_weights", stream);
            PackArgs pa{};
            pa.w0 = p->w0; pa.W = p->W; pa.F = d.F; pa.out = reinterpret_cast<float*>(out->packed);
            for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) { pa.w1[hd] = p->w1[hd]; pa.head_on[hd] = p->head_on[hd]; }
            const int n4 = (d.F * p->W + FDGS_NUM_HEADS * p->W * p->W) / 4;
            hipLaunchKernelGGL(pack_weights16_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, stream, pa);
        }
        int wgs;
        if (formws) {
            const int per_cu = p->W == 128 ? 1 : 2, nt16 = d.Npad / 16;
            const int want = g_tune.d1_wgs > 0 ? g_tune.d1_wgs : per_cu * cus;
            wgs = want < nt16 ? want : nt16;
        } else {
            const int want = g_tune.d1_wgs >= 0 ? g_tune.d1_wgs : (form16 ? 2 * cus : cus);     // 0: one workgroup per four tiles (not persistent)
            const int wg_tiles = form16 ? d.ntiles / 2 : d.ntiles / 4;           // (form 16: four 16-Gaussian tiles per workgroup)
            wgs = want > 0 && want < wg_tiles ? want : wg_tiles;
        }
        d.prof = nullptr;
#if defined(FDGS_PROFILE_D1) || defined(FDGS_PROFILE_WS)
        static unsigned long long* prof_dev = nullptr;
        if (!prof_dev) { (void)hipMalloc(&prof_dev, 16 * sizeof(unsigned long long)); }
        (void)hipMemsetAsync(prof_dev, 0, 16 * sizeof(unsigned long long), stream);
        d.prof = prof_dev;
#endif
        {
            FDGS_TIMED("deform_fwd", stream);       // (the forward kernel alone; the operand-stream copy above is timed as "pack_weights")
            rc = formws ? dispatch_wf<FwdWsLauncher>(p->W, d.F, stream, wgs, d)
               : form16 ? dispatch_wf<Fwd16Launcher>(p->W, d.F, stream, wgs, d) : dispatch_wf<FwdLauncher>(p->W, d.F, stream, wgs, d);
        }
#ifdef FDGS_PROFILE_WS
        {
            static int reports = 0;
            if (reports++ == 5) {
                unsigned long long hb[16];
                (void)hipStreamSynchronize(stream);
                (void)hipMemcpy(hb, prof_dev, sizeof(hb), hipMemcpyDeviceToHost);
                const char* names[7] = {"(loop top)", "feature park + request", "barrier", "copy-out + hmask", "heads", "epilogue", "trunk of the next tile"};
                double tot = 0;
                for (int i = 0; i < 7; i++) tot += (double)hb[i];
                fprintf(stderr, "[D1-ws profile] %llu waves, s_memtime ticks per wave:\n", hb[8]);
                for (int i = 0; i < 7; i++) fprintf(stderr, "  %-28s %12.0f  (%.1f %%)\n", names[i], (double)hb[i] / (double)hb[8], 100.0 * hb[i] / tot);
            }
        }
#endif
#ifdef FDGS_PROFILE_D1
        {
            static int reports = 0;
            if (reports++ == 5) {
                unsigned long long hb[16];
                (void)hipStreamSynchronize(stream);
                (void)hipMemcpy(hb, prof_dev, sizeof(hb), hipMemcpyDeviceToHost);
                const char* names[8] = {"prologue (query, inputs, preloads)", "gather", "feat store + trunk + relu + park + hmask", "L1.run (+drains)",
                                        "relu + park + next preloads", "L2 (+L2b)", "epilogue", "final drain"};
                double tot = 0;
                for (int i = 0; i < 8; i++) tot += (double)hb[i];
                fprintf(stderr, "[D1 profile] %llu waves, cycles per wave:\n", hb[8]);
                for (int i = 0; i < 8; i++) fprintf(stderr, "  %-44s %12.0f  (%.1f %%)\n", names[i], (double)hb[i] / (double)hb[8], 100.0 * hb[i] / tot);
                if (hb[9] || hb[10]) fprintf(stderr, "  of which: ring commit vmcnt(0) %.0f, s_barrier %.0f\n", (double)hb[9] / (double)hb[8], (double)hb[10] / (double)hb[8]);
            }
        }
#endif
    }
    if (rc) return rc;
    FDGS_LAUNCH_CHECK("deform_fwd", 0, stream);
    return FDGS_OK;
}

extern "C" int fdgs_deform_pack_bytes(const fdgs_deform_params* p, size_t* bytes) {
    int rc = validate_deform(p);
    if (rc) return rc;
    FDGS_REQUIRE(bytes, "bytes is NULL");
    // the operand streams of the 16-Gaussian form, or -- weight-stationary form without saved activations -- the gathered features [Npad][C*L]
    const size_t streams = ((size_t)p->C * p->L * p->W + (size_t)FDGS_NUM_HEADS * p->W * p->W) * sizeof(float);
    const size_t feats = npad_of(p->N) * (size_t)p->C * p->L * sizeof(float);
    *bytes = streams > feats ? streams : feats;
    return FDGS_OK;
}

extern "C" int fdgs_deform_saved_bytes(const fdgs_deform_params* p, size_t* bytes) {
    int rc = validate_deform(p);
    if (rc) return rc;
    FDGS_REQUIRE(bytes, "bytes is NULL");
    *bytes = saved_layout(p).floats * sizeof(float) + 256;
    return FDGS_OK;
}

extern "C" int fdgs_deform_bwd_scratch_bytes(const fdgs_deform_params* p, size_t* bytes) {
    int rc = validate_deform(p);
    if (rc) return rc;
    FDGS_REQUIRE(bytes, "bytes is NULL");
    *bytes = bwd_layout(p).floats * sizeof(float) + 1024;
    return FDGS_OK;
}

extern "C" int fdgs_deform_bwd_live_tiles(void* stream_, const fdgs_deform_params* p, const void* scratch, uint32_t* out_host) {
    int rc = validate_deform(p);
    if (rc) return rc;
    FDGS_REQUIRE(scratch && out_host, "NULL pointer");
    const BwdLayout bl = bwd_layout(p);
    uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    FDGS_HIP_CHECK(hipMemcpyAsync(c, reinterpret_cast<const float*>(scratch) + bl.counters, sizeof(c), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    FDGS_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
    // (row-list form: the 32-row units / chunks of the padded ROW list are what the kernels walked)
    out_host[0] = c[5] ? c[5] : c[1]; out_host[1] = c[3]; out_host[2] = c[5] ? c[6] : c[2];
    out_host[3] = (c[3] + 3) / 4;     // (reported in 128-Gaussian units)
    return FDGS_OK;
}

extern "C" int fdgs_deform_bwd(void* stream_, const fdgs_deform_params* p, const fdgs_deform_grads* g) {
    int rc = validate_deform(p);
    if (rc) return rc;
    FDGS_REQUIRE(g && g->scratch, "grads/scratch is NULL");
    if (p->N == 0) return FDGS_OK;
    if (p->activate && !g->packed_rows_ready) {
        FDGS_REQUIRE(!g->g_scales || g->out_scales, "out_scales needed with activate=1");
        FDGS_REQUIRE(!g->g_rotations || (g->out_rotations && g->rot_norm), "out_rotations/rot_norm needed with activate=1");
        FDGS_REQUIRE(!g->g_opacity || g->out_opacity, "out_opacity needed with activate=1");
    }
    hipStream_t stream = (hipStream_t)stream_;
    const size_t Np = npad_of(p->N), F = (size_t)p->C * p->L, W = p->W;
    const int nh = active_heads(p);
    BwdScratch s;
    float* base = reinterpret_cast<float*>(g->scratch);
    const BwdLayout bl = bwd_layout(p);
    s.Npad = (int)Np;
    s.G = base + bl.G; s.DH1 = base + bl.DH1; s.DHID = base + bl.DHID; s.RH = base + bl.RH; s.FEAT = base + bl.FEAT; s.DFEAT = base + bl.DFEAT;
    s.tile_live = reinterpret_cast<uint32_t*>(base + bl.flags); s.live = reinterpret_cast<uint32_t*>(base + bl.live);
    s.chunks = reinterpret_cast<uint32_t*>(base + bl.chunks); s.counters = reinterpret_cast<uint32_t*>(base + bl.counters);
    s.rows = nullptr;
    // prep: activation Jacobians, identity paths, packed gradient rows
    PrepArgs pa{};
    pa.N = p->N; pa.Npad = (int)Np; pa.activate = p->activate; pa.dc_stride = p->shs_dc_stride; pa.rest_stride = p->shs_rest_stride;
    pa.g_xyz = g->g_xyz; pa.g_scales = g->g_scales; pa.g_rot = g->g_rotations; pa.g_opacity = g->g_opacity; pa.g_shs = g->g_shs;
    pa.out_scales = g->out_scales; pa.out_rot = g->out_rotations; pa.out_opacity = g->out_opacity; pa.rot_norm = g->rot_norm;
    pa.d_xyz = g->d_xyz; pa.d_scales = g->d_scales; pa.d_rot = g->d_rotations; pa.d_opacity = g->d_opacity;
    pa.d_shs_dc = g->d_shs_dc; pa.d_shs_rest = g->d_shs_rest; pa.G = s.G; pa.tile_live = s.tile_live;
    if (!g->packed_rows_ready) {    // (1: fdgs_raster_bwd's deformation epilogue already wrote G and the identity paths)
        { FDGS_TIMED("deform_bwd_prep", stream); hipLaunchKernelGGL(deform_bwd_prep_kernel, dim3(cdiv((long long)Np, 256)), dim3(256), 0, stream, pa); }
        FDGS_LAUNCH_CHECK("deform_bwd_prep", 0, stream);
    }
    if (nh == 0) return FDGS_OK;  // no head active: the deformation is the identity
    // plane-gradient kernel choice and its chunk size (the chunk list is built for it)
    const int d4_env = g_tune.d4_mfma;
    const bool use_mfma = !p->time && (d4_env >= 0 ? d4_env != 0 : g->spatially_ordered != 0);
    const int Gc = 2048 / p->C;      // Gaussians per chunk of the splat
    {
        // tiles with a non-zero gradient row -> lists.  packed_rows_ready = 1: rows without flags -- every tile is live and the kernel writes
        // the flags (all ones); 3: the rows of dead tiles were never written -- skipping is not a choice then, whatever the A/B knob says
        CompactArgs ca{};
        ca.flags = s.tile_live; ca.live = s.live; ca.chunks = s.chunks; ca.counters = s.counters; ca.G = s.G;
        ca.ntiles = (int)(Np / 32); ca.tpc = Gc / 32;
        ca.skip = g->packed_rows_ready == 3 ? 1 : ((g_tune.skip_dead != 0 && g->packed_rows_ready != 1) ? 1 : 0);
        // ROW lists instead of tile lists: D2 / D3 / D4 walk only the rows that carry a gradient (on the bench scene 12 % of the rows, but
        // 17.5 % of the 32-row tiles and 21 % of the 128-row chunks, are live).  Needs the saved activations (fetched row by row) and the
        // splat form of D4 (which takes its chunks from any list); the dead tiles must be skippable at all (flags present).
        const bool by_rows = g_tune.row_compact != 0 && ca.skip && g->saved && use_mfma;
        if (by_rows) {
            ca.rowbase = reinterpret_cast<uint32_t*>(base + bl.rowbase); ca.rows = reinterpret_cast<uint32_t*>(base + bl.rows); ca.Gc = base + bl.Gc;
            ca.row_pad = Gc > 128 ? Gc : 128;
        }
        const bool one_launch = by_rows && ca.ntiles <= ROW_LIST_MAX_TILES && g_tune.row_compact != 2;     // (knob value 2: the two-launch form at any size, tests)
        if (one_launch) {
            RowListArgs ra{};
            ra.flags = s.tile_live; ra.G = s.G; ra.rows = ca.rows; ra.Gc = ca.Gc; ra.counters = s.counters; ra.ntiles = ca.ntiles; ra.tpc = ca.tpc; ra.row_pad = ca.row_pad;
            { FDGS_TIMED("row_list", stream); hipLaunchKernelGGL(row_list_kernel, dim3(cdiv(ca.ntiles, 8)), dim3(256), 0, stream, ra); }
            FDGS_LAUNCH_CHECK("row_list", 0, stream);
            s.rows = ca.rows; s.G = ca.Gc;       // (from here on "G" is the compact copy)
        } else {
        { FDGS_TIMED("tile_compact", stream); hipLaunchKernelGGL(tile_compact_kernel, dim3(1), dim3(1024), 0, stream, ca); }
        FDGS_LAUNCH_CHECK("tile_compact", 0, stream);
        }
        if (by_rows && !one_launch) {
            RowGatherArgs ra{};
            ra.flags = s.tile_live; ra.rowbase = ca.rowbase; ra.G = s.G; ra.rows = ca.rows; ra.Gc = ca.Gc; ra.ntiles = ca.ntiles;
            { FDGS_TIMED("row_gather", stream); hipLaunchKernelGGL(row_gather_kernel, dim3(cdiv(ca.ntiles, 8)), dim3(256), 0, stream, ra); }
            FDGS_LAUNCH_CHECK("row_gather", 0, stream);
            s.rows = ca.rows; s.G = ca.Gc;       // (from here on "G" is the compact copy)
        }
    }
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++)
        if (p->head_on[hd]) FDGS_REQUIRE(g->d_w1[hd] && g->d_b1[hd] && g->d_w2[hd] && g->d_b2[hd], "head gradient buffer missing");
    FDGS_REQUIRE(g->d_w0 && g->d_b0, "trunk gradient buffer missing");
    BwdDev bd;
    bd.p = *p; bd.sc = aabb_scale(p); bd.s = s; bd.F = (int)F; bd.ntiles = (int)(Np / 32); bd.small_heads = 1;
    const float* X_rh = s.RH;      // operands of the weight-gradient GEMMs: recomputed into scratch, or saved by the forward
    const float* X_feat = s.FEAT;
    bd.sv_rh = bd.sv_h1 = nullptr; bd.sv_hmask = nullptr;
    if (g->saved) {
        const SavedLayout sl = saved_layout(p);
        const float* sv = reinterpret_cast<const float*>(g->saved);
        bd.sv_rh = sv + sl.rh; bd.sv_h1 = sv + sl.h1; bd.sv_hmask = reinterpret_cast<const uint32_t*>(sv + sl.hmask);
        X_rh = sv + sl.rh; X_feat = sv + sl.feat;
    }
    int slot = 0;
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) {
        bd.d_w2[hd] = g->d_w2[hd]; bd.d_b2[hd] = g->d_b2[hd];
        bd.head_slot[hd] = p->head_on[hd] ? slot++ : 0;
    }
    bd.prof = nullptr;
#ifdef FDGS_PROFILE_D2
    static unsigned long long* prof_dev = nullptr;
    const size_t prof_n = 4096 * 12;
    if (!prof_dev) (void)hipMalloc(&prof_dev, prof_n * sizeof(unsigned long long));
    (void)hipMemsetAsync(prof_dev, 0, prof_n * sizeof(unsigned long long), stream);
    bd.prof = prof_dev;
#endif
#ifdef FDGS_PROFILE_D2WS
    static unsigned long long* prof_ws = nullptr;
    if (!prof_ws) (void)hipMalloc(&prof_ws, 16 * sizeof(unsigned long long));
    (void)hipMemsetAsync(prof_ws, 0, 16 * sizeof(unsigned long long), stream);
    bd.prof = prof_ws;
#endif
    {
        FDGS_TIMED("deform_bwd_data", stream);
        rc = dispatch_wf<BwdLauncher>(p->W, (int)F, stream, (int)(Np / 128), bd);
    }
#ifdef FDGS_PROFILE_D2WS
    {
        static int reports = 0;
        if (reports++ == 5) {
            unsigned long long hb[16];
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpy(hb, prof_ws, sizeof(hb), hipMemcpyDeviceToHost);
            const char* names[8] = {"loop edge", "wait + barrier 1", "ReLU bits, DMA issue, first head's slots", "product 4 (no riders)", "xp write",
                                    "product 0 (+ head 1, exchange A)", "product 1 (+ head 2, exchange B, barrier 2)", "products 2, 3 (+ head 3, exchange C; + SH head)"};
            double tot = 0;
            for (int i = 0; i < 8; i++) tot += (double)hb[i];
            fprintf(stderr, "[D2-ws profile] %llu waves, s_memtime ticks per wave (100 MHz):\n", hb[8]);
            for (int i = 0; i < 8; i++) fprintf(stderr, "  %-44s %12.0f  (%.1f %%)\n", names[i], (double)hb[i] / (double)(hb[8] ? hb[8] : 1), 100.0 * hb[i] / tot);
        }
    }
#endif
#ifdef FDGS_PROFILE_D2
    {
        static int reports = 0;
        if (reports++ == 5) {   // one report, after warm-up
            std::vector<unsigned long long> hbuf(prof_n);
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpy(hbuf.data(), prof_dev, prof_n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double sum[12] = {0}; int waves = 0;
            for (size_t w = 0; w < 4096; w++) {
                if (hbuf[w * 12 + 9] == 0) continue;
                waves++;
                for (int i = 0; i < 12; i++) sum[i] += (double)hbuf[w * 12 + i];
            }
            const char* names[12] = {"saved: head top .. rows in LDS", "saved: SH barrier wait", "recompute: L1.run | saved: masks + next-row requests",
                                     "operand preloads", "dW2 small heads (+ pre-barrier)", "dh1", "mask+DH1 store", "B1.run", "tail(DHID,B0,DFEAT)",
                                     "wave total", "saved: SH cooperative block", "saved: SH requests after the block"};
            fprintf(stderr, "[D2 profile] %d waves, s_memtime ticks (100 MHz) per wave:\n", waves);
            for (int i = 0; i < 12; i++) fprintf(stderr, "  %-26s %12.0f  (%.1f %%)\n", names[i], sum[i] / waves, 100.0 * sum[i] / sum[9]);
        }
    }
#endif
    if (rc) return rc;
    FDGS_LAUNCH_CHECK("deform_bwd_data", 0, stream);
    // weight gradients: one job per active head (dW1, db1) + the trunk (dW0, db0)
    WgradArgs wa{};
    wa.Npad = (int)Np; wa.W = (int)W; wa.live = s.live; wa.counters = s.counters; wa.rows = s.rows;
    int nj = 0;
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) {
        if (!p->head_on[hd]) continue;
        WgradJob& J = wa.job[nj++];
        J.DY = s.DH1 + (size_t)bd.head_slot[hd] * Np * W; J.X = X_rh; J.dW = g->d_w1[hd]; J.db = g->d_b1[hd];
        J.ldx = (int)W; J.ncols = (int)W; J.ldw = (int)W;
    }
    {
        WgradJob& J = wa.job[nj++];
        J.DY = s.DHID; J.X = X_feat; J.dW = g->d_w0; J.db = g->d_b0; J.ldx = (int)F; J.ncols = (int)F; J.ldw = (int)F;
    }
    wa.njobs = nj;
    {
        // one workgroup per CU; workgroups are shared out in proportion to the MFMA work of each job
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int total_wgs = cus;
        int work[FDGS_NUM_HEADS + 1], total_work = 0;
        // cost model: MFMAs per k-step; the narrow trunk product (dword loads, only CT MFMAs per load pair, deep ring) is
        // memory-latency rather than MFMA bound: it costs about twice its MFMA count (sweep: 0.53 / 0.44 / 0.46 / 0.48 ms at
        // factor 1 / 2 / 3 / 4)
        const int trunk_factor = 2;
        for (int j = 0; j < nj; j++) {
            work[j] = (wa.job[j].ncols + 31) / 32;
            if (wa.job[j].ncols != (int)W) work[j] *= trunk_factor;
            total_work += work[j];
        }
        // floor shares: never more workgroups than CUs (a single straggler would double the kernel time)
        int nbs[FDGS_NUM_HEADS + 1], used = 0;
        for (int j = 0; j < nj; j++) { nbs[j] = (int)((long long)total_wgs * work[j] / total_work); if (nbs[j] < 1) nbs[j] = 1; used += nbs[j]; }
        for (int j = 0; used < total_wgs; j = (j + 1) % nj) { nbs[j]++; used++; }   // leftovers round-robin, heads first
        int first = 0;
        const int ntl = (int)(Np / 32);
        for (int j = 0; j < nj; j++) {
            WgradJob& J = wa.job[j];
            int nb = nbs[j];
            if (nb > ntl) nb = ntl;             // (never more workgroups than tiles)
            J.first_block = first; J.nblocks = nb;
            first += nb;
        }
        FDGS_TIMED("deform_wgrad", stream);
        if (W == 128 && wa.rows) hipLaunchKernelGGL((deform_wgrad_kernel<4, true>), dim3(first), dim3(256), 0, stream, wa);
        else if (W == 128) hipLaunchKernelGGL((deform_wgrad_kernel<4, false>), dim3(first), dim3(256), 0, stream, wa);
        else if (wa.rows) hipLaunchKernelGGL((deform_wgrad_kernel<2, true>), dim3(first), dim3(256), 0, stream, wa);
        else hipLaunchKernelGGL((deform_wgrad_kernel<2, false>), dim3(first), dim3(256), 0, stream, wa);
    }
    FDGS_LAUNCH_CHECK("deform_wgrad", 0, stream);
    // plane + coordinate gradients
    bool any_plane = g->d_xyz != nullptr;
    PlaneGradArgs ga{};
    ga.p = *p; ga.sc = aabb_scale(p); ga.DFEAT = s.DFEAT; ga.d_xyz = g->d_xyz; ga.F = (int)F; ga.tile_live = s.tile_live;
    for (int l = 0; l < p->L; l++)
        for (int k = 0; k < 6; k++) { ga.d_planes[l][k] = g->d_planes[l][k]; any_plane = any_plane || g->d_planes[l][k]; }
    if (any_plane) {
        // matrix-core splat (default whenever one frame time is shared by all Gaussians, i.e. on the render() path): the
        // fixed LDS part is the dv tile, coordinates, descriptors; the time rows get what is left of the 160 KB
        // FDGS_D4_MFMA = 1 / 0 forces a kernel (A/B, tests); otherwise the caller's order hint decides
        // workgroup shape of the splat: 8 waves, one workgroup per CU (4-wave workgroups with half the chunk, two per CU, measured equal on
        // BASELINE config 4 -- 0.331 vs 0.322 ms -- and were dropped)
        const int fixed_floats = 6 * Gc * p->C + 3 * Gc + 13 * Gc + 3 * Gc + (96 + 9 * Gc) + Gc + 64;
        // LDS privatisation of the time planes (one frame time for all Gaussians): greedy by level while the tiles fit
        // bytes per workgroup: up to 128 KB (one 512-thread workgroup per CU then; the un-privatised alternative, float
        // atomics on ~128 hot lines, is 4x slower than scattered atomics)
        int lds_budget = use_mfma ? 160 * 1024 - fixed_floats * 4
                                  : 128 * 1024;
        if (use_mfma && g_tune.d4_rows_kb >= 0 && g_tune.d4_rows_kb * 1024 < lds_budget)
            lds_budget = g_tune.d4_rows_kb * 1024;     // (tests: time rows that do not fit take the global-atomic path)
        int used = 0;
        for (int l = 0; l < FDGS_MAX_LEVELS; l++)
            for (int sl = 0; sl < 3; sl++) ga.lds_off[l][sl] = -1;
        if (!p->time) {
            for (int l = 0; l < p->L; l++) {
                int need = 0;
                for (int sl = 0; sl < 3; sl++) need += p->res[l][sl] * p->C;
                const int kk[3] = {2, 4, 5};
                bool wanted = true;
                for (int sl = 0; sl < 3; sl++) wanted = wanted && g->d_planes[l][kk[sl]];
                if (!wanted || (size_t)(used + need) * 4 > (size_t)lds_budget) continue;
                for (int sl = 0; sl < 3; sl++) { ga.lds_off[l][sl] = used; used += p->res[l][sl] * p->C; }
            }
        }
        ga.lds_floats = used;
        if (use_mfma) {
            PlaneGradMArgs ma{};
            ma.g = ga;
            ma.prof = nullptr;
#ifdef FDGS_PROFILE_D4
            static unsigned long long* prof_dev = nullptr;
            const size_t prof_n = 64 * 16 * 10;   // (8 waves per workgroup are used)
            if (!prof_dev) (void)hipMalloc(&prof_dev, prof_n * sizeof(unsigned long long));
            (void)hipMemsetAsync(prof_dev, 0, prof_n * sizeof(unsigned long long), stream);
            ma.prof = prof_dev;
#endif
            ma.chunks = s.chunks; ma.counters = s.counters; ma.rows = s.rows;
            const int nchunks_max = cdiv(p->N, Gc);
            int o = (used + 63) / 64 * 64;
            ma.off_dv = o; o += 6 * Gc * p->C;
            ma.off_q = o; o += 3 * Gc;
            ma.off_desc = o; o += 13 * Gc;   // [3][G] float4 axis descriptors + [G] window masks
            ma.off_dq = o; o += 3 * Gc;
            ma.off_org = o; o += 96 + 9 * Gc;
            ma.off_row = o; o += Gc;
            const size_t lds_bytes = (size_t)o * 4;
#ifndef FDGS_D4_WGS
#define FDGS_D4_WGS 256
#endif
            int blocks = FDGS_D4_WGS;
            if (blocks > nchunks_max) blocks = nchunks_max;
            const void* fn = p->C == 16 ? reinterpret_cast<const void*>(&deform_plane_grad_mfma_kernel<16, 8>)
                                        : reinterpret_cast<const void*>(&deform_plane_grad_mfma_kernel<32, 8>);
            static bool raised[FDGS_MAX_DEVICES][2] = {};      // (a function attribute is set per device)
            bool& r_ = raised[current_device_slot()][p->C == 16 ? 0 : 1];
            if (!r_) {
                FDGS_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                r_ = true;
            }
            {
                FDGS_TIMED("deform_plane_grad", stream);
                if (p->C == 16) hipLaunchKernelGGL((deform_plane_grad_mfma_kernel<16, 8>), dim3(blocks), dim3(512), lds_bytes, stream, ma);
                else hipLaunchKernelGGL((deform_plane_grad_mfma_kernel<32, 8>), dim3(blocks), dim3(512), lds_bytes, stream, ma);
            }
            FDGS_LAUNCH_CHECK("deform_plane_grad", 0, stream);
#ifdef FDGS_PROFILE_D4
            {
                static int reports = 0;
                if (reports++ == 5) {
                    std::vector<unsigned long long> hbuf(prof_n);
                    (void)hipStreamSynchronize(stream);
                    (void)hipMemcpy(hbuf.data(), prof_dev, prof_n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                    const char* nm[8] = {"S0+bar", "S", "bar(S)", "M.sp.loop", "M.sp.flush", "M.time", "bar(M)", "dxyz+bar"};
                    for (int wv : {0, 1, 2, 3, 5, 6, 7}) {
                        double sum[10] = {0}; int cnt = 0;
                        for (int b = 0; b < 64; b++) { const unsigned long long* r = &hbuf[(size_t)(b * 8 + wv) * 10]; if (!r[9]) continue; cnt++; for (int i = 0; i < 10; i++) sum[i] += (double)r[i]; }
                        if (!cnt) continue;
                        fprintf(stderr, "[D4 profile] wave %2d (total %.0f cyc, s_memtime units):", wv, sum[9] / cnt);
                        for (int i = 0; i < 8; i++) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * sum[i] / sum[9]);
                        fprintf(stderr, "\n");
                    }
                }
            }
#endif
            return FDGS_OK;
        }
        const int gpb = (PG_THREADS / 64) * (64 / (2 * p->C));       // Gaussians per workgroup iteration
        int nwg = 512;                                                // ~2 workgroups per CU
        int per_block = cdiv(p->N, nwg);
        per_block = cdiv(per_block, gpb) * gpb;
        if (per_block < gpb) per_block = gpb;
        ga.per_block = per_block;
        const int blocks = cdiv(p->N, per_block);
        const size_t lds_bytes = (size_t)used * 4;
        if (lds_bytes > 64 * 1024) {   // above the default dynamic-LDS limit: opt in once per kernel
            static bool raised_pc[FDGS_MAX_DEVICES][2] = {};
            bool& raised = raised_pc[current_device_slot()][p->C == 16 ? 0 : 1];
            if (!raised) {
                const void* fn = p->C == 16 ? reinterpret_cast<const void*>(&deform_plane_grad_kernel<16>)
                                            : reinterpret_cast<const void*>(&deform_plane_grad_kernel<32>);
                FDGS_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                raised = true;
            }
        }
        if (p->C == 16) { FDGS_TIMED("deform_plane_grad", stream); hipLaunchKernelGGL((deform_plane_grad_kernel<16>), dim3(blocks), dim3(PG_THREADS), lds_bytes, stream, ga); }
        else { FDGS_TIMED("deform_plane_grad", stream); hipLaunchKernelGGL((deform_plane_grad_kernel<32>), dim3(blocks), dim3(PG_THREADS), lds_bytes, stream, ga); }
        FDGS_LAUNCH_CHECK("deform_plane_grad", 0, stream);
    }
    return FDGS_OK;
}
