// deform_fwd16.h -- D1 (HexPlane gather -> trunk -> five heads -> activations) in its 16-GAUSSIAN form.  Included by deform.hip
// (inside namespace fdgs, after the 32-Gaussian kernel whose helpers -- axis_sample, plane_axes, load_query, DeformDev, head_k,
// store / epilogue conventions -- it shares).
//
// Why a second form (round 4).  deform_fwd_kernel gives one wave 32 Gaussians on v_mfma_f32_32x32x2_f32: 64 + 64 accumulator registers
// for the trunk output and a head's hidden layer, 494 registers in all, ONE wave per SIMD -- and nothing runs on that SIMD while the wave
// gathers texels, applies ReLU, parks a tile or waits at the top of a layer: SQ_VALU_MFMA_BUSY 58 %, the in-kernel cycle profile has 44 %
// of a wave's cycles in phases no MFMA overlaps (profiles/r03y_d1_d2_cycle_profiles.txt); two attempts to interleave those phases with the
// long products inside the one wave were defeated by the register allocator.  Here a wave owns 16 Gaussians on
// v_mfma_f32_16x16x4_f32 (same 64 FLOP / clk / SIMD): the activations of a layer are 32 registers instead of 64, the kernel fits 256
// registers, TWO workgroups are resident per CU = two waves per SIMD -- and the hardware overlaps one wave's gather / ReLU / epilogue
// with the other's MFMAs, which the compiler could not be made to do inside one instruction stream.  Price: every weight element now
// feeds 16 Gaussians instead of 32 (twice the operand traffic from L2 / L1: 17.6 B/clk/CU of 64), 4 lanes instead of 2 repeat a
// Gaussian's address arithmetic in the gather.  Same memory formats in and out (saved activations, ReLU bit masks in D2's lane layout),
// so D2 / D3 / D4 do not know which form ran.
// MEASURED (round 4, profiles/r04_d1_forms.txt): row-major operands 1.55 ms (the texture path: 64 cache lines per request, two waves per
// SIMD no faster than one) -> packed operand streams 0.73 ms -> group-wise hidden layers, biases from LDS, per-head input requests 0.714 ms
// (forward-only 0.561 ms) against 0.720 / 0.587 ms of the 32-Gaussian form on the same box: a draw.  What the in-kernel cycle profile says is
// left: a hidden-layer product takes 11.4 k cycles alone even with every operand request an L1 hit (8.2 k of MFMA issue: ~200 cycles per
// 16-MFMA stage that are not the stream's latency), 13.3 k with the real stream at 2 stages of requests in flight, and a third / fourth
// stage in flight does not fit 256 registers without spills (each A register feeds ONE 32-cycle MFMA here, against one 64-cycle MFMA in the
// 32-form: twice the registers for the same latency cover).  MEASURED IN THE FRAME (where the kernel starts on cold planes behind the previous
// frame's backward) this form is 5 - 7 % faster on every workload and is the DEFAULT; the 32-form (deform_fwd_kernel) runs when
// C*L is not a multiple of 16 or the caller hands over no pack scratch (fdgs_tuning "d1_form" = 32 forces it).  Two further forms
// built on it (operand streams through an LDS ring; a group-wise 32-Gaussian form) measured no faster and live as a patch under
// tools/_experiments/deform_d1_forms_17_33.diff.txt.
//
// Lane = (n, q): n = lane & 15 the Gaussian, q = lane >> 4.  MFMA 16x16x4: A[i][k] in lane i + 16k, B[k][n] in lane n + 16k,
// D[row][n] in lane n + 16q register r with row = 4q + r.
// "IL16" activation layout (T = W / 16 tiles): tile t, register r of lane (n, q) = feature T * (4q + r) + t of Gaussian n.  With it a
// product Y = W X reads the torch-layout weights with 16-byte loads: k-step (r, t) has lane group q supply feature T (4q + r) + t, so
// A-lane (i, q) needs W[row][T (4q + r) + t], t = 0 .. T-1 contiguous.  Output rows interleaved the same way (row = OT i + ot) so that
// the output is again IL16.  HexPlane features: register c of group u of lane (n, q) = feature 16u + 4q + c (one float4 per texel
// corner), k-step (u, c).  tools/mfma_layout_model.py (check16) replays all of it lane by lane against plain matrix algebra
// (tests/test_mfma_layouts.py).

#ifndef FDGS_D16_PD1
#define FDGS_D16_PD1 3        // operand-request stages in flight per hidden-layer product: 3 fit 256 registers since the requests of a packed
                              // stream are addressed by ONE running lane offset + immediates (fetch below; one offset register per request
                              // before: 64 registers, and 3 stages spilled); 4 spill.  D1 -1 % (profiles/r05_d1_running_offset_ab.txt)
#endif
#ifndef FDGS_NT_SAVE
#define FDGS_NT_SAVE 1        // the saved activations (written once, read once by the backward ~1 ms later) leave with non-temporal stores:
#endif                         // +0.2 .. 0.6 % frames/s in the frame, cube and shell scene (profiles/r05_variants_ab_nt_ppl_form.txt; 0 = plain stores)
#ifndef FDGS_D16_PRIO
#define FDGS_D16_PRIO 2       // wave priority by phase: 2 = raised OUTSIDE the matrix-core stretches (gather, ReLU / parking, epilogues: the wave
#endif                         // whose next memory request is on its critical path goes first; the other wave's MFMA chain tolerates the delay):
                               // D1 -0.5 .. -1.2 %, +0.3 .. 0.8 % frames/s in the frame, four pairs of runs; 1 = raised INSIDE them: slower;
                               // also raising it around the operand requests of the products: no better (profiles/r05_d1_wave_priority_ab.txt); 0 = off
#define FDGS_PRIO_MFMA(on) do { if (FDGS_D16_PRIO == 1) __builtin_amdgcn_s_setprio((on) ? 2 : 0); else if (FDGS_D16_PRIO == 2) __builtin_amdgcn_s_setprio((on) ? 0 : 2); } while (0)
typedef float nt4f_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_saved16(float* dst, const float4& v) {
#if FDGS_NT_SAVE
    __builtin_nontemporal_store(nt4f_{v.x, v.y, v.z, v.w}, reinterpret_cast<nt4f_*>(dst));
#else
    *reinterpret_cast<float4*>(dst) = v;
#endif
}
__device__ __forceinline__ f32x4 mm16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ float f4c(const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

// Y[ot] = bias + W X.  X in IL16 (KT tiles).  ROW_IL: output rows interleaved (row = OT n + ot, Y is IL16 again), else standard
// (row = 16 ot + n, clamped to out_dim - 1: rows beyond are duplicates nobody reads).  W row-major [rows][ld], global or LDS.
// Stage s = (oh, r, hf): one float4 (features KT (4q + r) + 4 hf .. + 3) for each of the OG <= 4 output tiles of GROUP oh feeds 4 OG
// MFMAs; PD buffers: a stage's buffer is refilled right behind its MFMAs, i.e. requests run PD - 1 stages (512 MFMA cycles each at
// OG = 4) ahead.  The groups are walked one after the other (run_group): the caller finishes a group -- bias, ReLU, parking, its share of
// the second layer -- before the next one starts, so only OG (not OT) output tiles are ever held in accumulators; the request ring runs
// on across the group boundary.
template <int KT, int OT, bool ROW_IL, int PD, bool PACKED = false>
struct Dense16 {
    static constexpr int HV = KT / 4, OG = OT < 4 ? OT : 4, OH = (OT + OG - 1) / OG, NS = 4 * HV * OH;
    static_assert(OT % OG == 0, "output tiles come in whole groups");
    int bq;
    uint32_t ro[PACKED ? 1 : OT];   // byte offsets of the lane's rows from the (wave-uniform) matrix base: SGPR base + VGPR offset loads
    const char* base;
    float bv[OT];
    float4 buf[PD][OG];
    // row-major weights [rows][ld] (global or LDS)
    __device__ __forceinline__ void setup(const float* __restrict__ Wm, int ld, int out_dim, int n, int q) {
        static_assert(!PACKED, "packed operand streams use setup_packed");
        base = reinterpret_cast<const char*>(Wm);
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            int row = ROW_IL ? OT * n + ot : 16 * ot + n;
            row = row < out_dim ? row : out_dim - 1;
            ro[ot] = (uint32_t)(row * ld + KT * 4 * q) * 4u;
        }
    }
    // PACKED: the matrix re-ordered into the operand stream of this very loop (pack_weights16_kernel): float4 index (s * OG + o) * 64 + lane.
    // Row-major, a lane's float4 sits in its own 128-byte line (row stride 4 W bytes, lane groups 128 bytes apart): 64 lines per load
    // instruction at one line per clock and CU -- four waves' operand requests alone occupied the texture path for longer than their MFMAs
    // ran (measured: two waves per SIMD no faster than one).  Packed, a load instruction is 1 KB contiguous = 8 lines.
    __device__ __forceinline__ void setup_packed(const float* __restrict__ P, int lane) {
        base = reinterpret_cast<const char*>(P);
        ro[0] = (uint32_t)lane * 16u;
    }
    __device__ __forceinline__ void load_bias(const float* __restrict__ bias, int out_dim, int n, int q) {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            int row = ROW_IL ? OT * n + ot : 16 * ot + n;
            row = row < out_dim ? row : out_dim - 1;
            bv[ot] = bias[row];             // (raw: the q == 0 select sits at the use, so that nothing waits for this load here)
        }
        bq = q;
    }
    __device__ __forceinline__ void fetch(int s, float4* dst) {
        if constexpr (PACKED) {
            // the stages of a packed stream are requested strictly in order (preload: 0 .. PD - 1, then s + PD behind stage s): a running
            // lane offset, the OG pieces of a stage as immediates.  Written as `ro[0] + (s * OG + o) * 1024` the compiler materialises one
            // 32-bit offset register PER REQUEST of a head (it may not fold a constant into a 32-bit lane offset) and keeps all 64 alive.
            (void)s;
#pragma unroll
            for (int o = 0; o < OG; o++) dst[o] = *reinterpret_cast<const float4*>(base + (size_t)ro[0] + (size_t)(o * 1024));    // (64-bit sum: the piece folds into the instruction)
            ro[0] += (uint32_t)(OG * 1024);
            asm volatile("" : "+v"(ro[0]));
        } else {
            const int hf = s % HV, r = (s / HV) % 4, oh = s / (4 * HV);
#pragma unroll
            for (int o = 0; o < OG; o++) dst[o] = *reinterpret_cast<const float4*>(base + (ro[OG * oh + o] + (uint32_t)(KT * r + 4 * hf) * 4u));
        }
    }
    __device__ __forceinline__ void preload() {
#pragma unroll
        for (int s = 0; s < PD; s++) if (s < NS) fetch(s, buf[s]);
    }
    struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
    __device__ __forceinline__ void run(const f32x4* X, f32x4* Y) { run(X, Y, NoHook()); }
    template <class Hook>
    __device__ __forceinline__ void run(const f32x4* X, f32x4* Y, Hook hook) {
        // the bias rides in as k = 0 of one extra k-step (A = bias in lane group 0, B = 1)
#pragma unroll
        for (int ot = 0; ot < OT; ot++) Y[ot] = mm16(bq == 0 ? bv[ot] : 0.f, 1.0f, zero4());
#pragma unroll
        for (int oh = 0; oh < OH; oh++) run_group(oh, X, Y + OG * oh, hook, false);
    }
    // the stages of output group `oh` (compile-time constant at every call site): Yg[o] (+)= W[rows of group oh] X; `hook(j)` once per stage,
    // j = 0 .. NS / OH - 1
    static constexpr int NSG = 4 * HV;
    template <class Hook>
    __device__ __forceinline__ void run_group(int oh, const f32x4* X, f32x4* Yg, Hook hook, bool zero_init = true) {
        if (zero_init) {
#pragma unroll
            for (int o = 0; o < OG; o++) Yg[o] = zero4();
        }
#pragma unroll
        for (int j = 0; j < NSG; j++) {
            const int s = oh * NSG + j, hf = j % HV, r = j / HV;
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int o = 0; o < OG; o++) Yg[o] = mm16(f4c(buf[s % PD][o], c), X[4 * hf + c][r], Yg[o]);
            __builtin_amdgcn_sched_barrier(0);      // the refill stays behind the MFMAs that read the buffer, and ahead of the next stage's
            if (s + PD < NS) fetch(s + PD, buf[s % PD]);
            hook(j);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// hid[ot] = b0 + W0 feat: feat[u] = float4 of features 16u + 4q .. + 3, output IL16 (row = OT n + ot).  Stage = (u, oh).
template <int FU, int OT, int PD>
struct Trunk16 {
    static constexpr int OG = OT < 4 ? OT : 4, OH = OT / OG, NS = FU * OH;
    uint32_t lo;
    int bq;
    const char* base;
    float bv[OT];
    float4 buf[PD][OG];
    // P: W0 as this loop's operand stream (pack_weights16_kernel): float4 index (s * OG + o) * 64 + lane, s = u * OH + oh
    __device__ __forceinline__ void setup(const float* __restrict__ P, const float* __restrict__ bias, int n, int q, int lane) {
        base = reinterpret_cast<const char*>(P);
        lo = (uint32_t)lane * 16u;
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            bv[ot] = bias[OT * n + ot];
        }
        bq = q;
    }
    __device__ __forceinline__ void fetch(int s, float4* dst) const {
#pragma unroll
        for (int o = 0; o < OG; o++) dst[o] = *reinterpret_cast<const float4*>(base + (lo + (uint32_t)((s * OG + o) * 1024)));
    }
    __device__ __forceinline__ void preload() {
#pragma unroll
        for (int s = 0; s < PD; s++) if (s < NS) fetch(s, buf[s]);
    }
    __device__ __forceinline__ void run(const float4* feat, f32x4* Y) {
#pragma unroll
        for (int ot = 0; ot < OT; ot++) Y[ot] = mm16(bq == 0 ? bv[ot] : 0.f, 1.0f, zero4());
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int oh = s % OH, u = s / OH;
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int o = 0; o < OG; o++) Y[OG * oh + o] = mm16(f4c(buf[s % PD][o], c), f4c(feat[u], c), Y[OG * oh + o]);
            __builtin_amdgcn_sched_barrier(0);
            if (s + PD < NS) fetch(s + PD, buf[s % PD]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// W0 and the heads' W1 re-ordered into the operand streams of Trunk16 / Dense16 (one float4 per thread; 366 KB at net_width 128: a few
// microseconds, run by fdgs_deform_fwd in front of the forward kernel -- the weights change every optimizer step).
// Packed buffer: [W0: F * W floats][head 0 W1: W * W floats] ... [head 4].
struct PackArgs { const float* w0; const float* w1[FDGS_NUM_HEADS]; int head_on[FDGS_NUM_HEADS]; int W, F; float* out; };
__global__ void __launch_bounds__(256) pack_weights16_kernel(PackArgs a) {
    const int W = a.W, F = a.F, OT = W / 16, OG = 4, OH = OT / OG, KT = W / 16, HV = KT / 4;
    const int n4_trunk = F * W / 4, n4_head = W * W / 4;
    int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n4_trunk + FDGS_NUM_HEADS * n4_head) return;
    const float* src;
    int row, col, ld;
    if (e < n4_trunk) {
        const int lane = e & 63, o = (e >> 6) % OG, s = e / (64 * OG);
        const int oh = s % OH, u = s / OH, n = lane & 15, q = lane >> 4;
        row = OT * n + OG * oh + o; col = 16 * u + 4 * q; ld = F; src = a.w0;
    } else {
        const int e1 = e - n4_trunk, hd = e1 / n4_head, k = e1 - hd * n4_head;
        if (!a.head_on[hd]) return;
        const int lane = k & 63, o = (k >> 6) % OG, s = k / (64 * OG);
        const int hf = s % HV, r = (s / HV) % 4, oh = s / (4 * HV), n = lane & 15, q = lane >> 4;      // stage order of Dense16: (oh, r, hf)
        row = OT * n + OG * oh + o; col = KT * (4 * q + r) + 4 * hf; ld = W; src = a.w1[hd];
    }
    reinterpret_cast<float4*>(a.out)[e] = *reinterpret_cast<const float4*>(src + (size_t)row * ld + col);
}

// Features 16u + 4q .. + 3 of the lane's Gaussian: product over the six planes of the bilinear samples, the arithmetic of gather_chunk
// (same order of operations: the two forms of the kernel produce the same feature bits).  All 24 texel requests are issued before the
// first is consumed.
__device__ __forceinline__ float4 gather_group16(const fdgs_deform_params& p, int u, int q, const float* qc) {
    const int lvl = __builtin_amdgcn_readfirstlane((16 * u) / p.C);
    const int c0 = 16 * u - lvl * p.C + 4 * q;
    AxisSample S[4];
#pragma unroll
    for (int ax = 0; ax < 4; ax++) S[ax] = axis_sample(qc[ax], p.res[lvl][ax]);
    float4 prod = make_float4(1.f, 1.f, 1.f, 1.f);
    // two batches of three planes (12 requests = 48 registers in flight each): the kernel lives on 256 registers so that a second wave
    // shares the SIMD, and that wave -- not a deeper request queue -- covers the extra round trip
#pragma unroll
    for (int kb = 0; kb < 6; kb += 3) {
        float4 v[3][4];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            const int k = kb + kk;
            int a, b;
            plane_axes(k, a, b);
            const int Wd = p.res[lvl][a];
            const AxisSample sx = S[a], sy = S[b];
            const char* P = reinterpret_cast<const char*>(p.planes[lvl][k]);
            const uint32_t texel = (uint32_t)p.C * 4u, cb = (uint32_t)c0 * 4u;
            const uint32_t r0 = (uint32_t)(sy.i0 * Wd) * texel + cb, r1 = (uint32_t)(sy.i1 * Wd) * texel + cb;
            const uint32_t x0 = (uint32_t)sx.i0 * texel, x1 = (uint32_t)sx.i1 * texel;
            v[kk][0] = *reinterpret_cast<const float4*>(P + (r0 + x0));
            v[kk][1] = *reinterpret_cast<const float4*>(P + (r0 + x1));
            v[kk][2] = *reinterpret_cast<const float4*>(P + (r1 + x0));
            v[kk][3] = *reinterpret_cast<const float4*>(P + (r1 + x1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            const int k = kb + kk;
            int a, b;
            plane_axes(k, a, b);
            const AxisSample sx = S[a], sy = S[b];
            const float w00 = sx.w0 * sy.w0, w01 = sx.w1 * sy.w0, w10 = sx.w0 * sy.w1, w11 = sx.w1 * sy.w1;
            prod.x *= v[kk][0].x * w00 + v[kk][1].x * w01 + v[kk][2].x * w10 + v[kk][3].x * w11;
            prod.y *= v[kk][0].y * w00 + v[kk][1].y * w01 + v[kk][2].y * w10 + v[kk][3].y * w11;
            prod.z *= v[kk][0].z * w00 + v[kk][1].z * w01 + v[kk][2].z * w10 + v[kk][3].z * w11;
            prod.w *= v[kk][0].w * w00 + v[kk][1].w * w01 + v[kk][2].w * w10 + v[kk][3].w * w11;
        }
        // (the next batch's addresses depend on this batch's result: sched_barrier does not order loads at the IR level)
        if (kb == 0) asm volatile("" : "+v"(S[0].i0), "+v"(S[1].i0), "+v"(S[2].i0) : "v"(prod.x));
    }
    return prod;
}

// IL16 activations -> row n of a per-wave LDS tile [16][STRIDE] in feature order (the memory layout of the saved activations)
template <int T>
__device__ __forceinline__ void park16(float* tile, int stride, const f32x4* x, int n, int q) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int hf = 0; hf < T / 4; hf++)
            *reinterpret_cast<float4*>(tile + n * stride + T * (4 * q + r) + 4 * hf) =
                make_float4(x[4 * hf][r], x[4 * hf + 1][r], x[4 * hf + 2][r], x[4 * hf + 3][r]);
}

template <int WT16, int FU>
__global__ void __launch_bounds__(256, 2) deform_fwd16_kernel(DeformDev d) {
    constexpr int W = 16 * WT16, KT = WT16, OT = WT16;
    constexpr int PD1 = FDGS_D16_PD1;
    using L1T = Dense16<KT, OT, true, PD1, true>;
    constexpr int NS = L1T::NS, OG = L1T::OG, OH = L1T::OH, NSG = L1T::NSG;
    constexpr int LDW = W + 4;            // LDS row stride of the staged second-layer weights and of the parking tiles (bank spread)
    constexpr int NPIECE = W / 16;        // 1-KB pieces of a parked [16][W] tile (64 float4 each)
    static_assert(NPIECE <= NSG, "a parked tile is drained within the first output group of the next layer (whose results then park over it)");
    const fdgs_deform_params& p = d.p;
    constexpr int W2ROWS = 59;                      // second-layer rows of the five heads (3 + 3 + 4 + 1 + 48)
    __shared__ __attribute__((aligned(16))) float lds[4 * 16 * LDW + W2ROWS * LDW + FDGS_NUM_HEADS * W];
    float* my_tile = lds + (threadIdx.x >> 6) * 16 * LDW;
    float* w2lds = lds + 4 * 16 * LDW;
    float* b1lds = w2lds + W2ROWS * LDW;        // the heads' first-layer biases (added with the ReLU, in IL16 order: two ds_read_b128 per register row)
    for (int i = threadIdx.x; i < FDGS_NUM_HEADS * W; i += 256) b1lds[i] = p.head_on[i / W] ? p.b1[i / W][i % W] : 0.f;
    for (int hd_ = 0; hd_ < FDGS_NUM_HEADS; hd_++) {
        if (!p.head_on[hd_]) continue;
        const int k_ = head_k(hd_), r0_ = head_row0(hd_);
        for (int i = threadIdx.x; i < k_ * (W / 4); i += 256) {
            const int r = i / (W / 4), c4 = i - r * (W / 4);
            *reinterpret_cast<float4*>(w2lds + (r0_ + r) * LDW + 4 * c4) = reinterpret_cast<const float4*>(p.w2[hd_])[i];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, n0 = lane & 15, q0 = lane >> 4;
    // The two workgroups of a CU start together and do the same work per tile: left alone, the two waves of a SIMD stay IN PHASE -- both in
    // their matrix-core stretches (sharing the pipe), then both in their gather / ReLU / epilogue stretches (pipe idle): measured, the
    // kernel ran at 2 m + v per tile (m = MFMA time, v = the rest) instead of max(2 m, m + v).  The second half of the grid therefore
    // starts `skew` cycles late (about half a head iteration), once.
    if (d.skew > 0 && blockIdx.x >= (gridDim.x >> 1)) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)d.skew) __builtin_amdgcn_s_sleep(32);
    }
#ifdef FDGS_PROFILE_D1
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pt = __builtin_amdgcn_s_memtime();
#endif
    unsigned all_heads = 0u;
    int nh = 0;
#pragma unroll
    for (int i = 0; i < FDGS_NUM_HEADS; i++) if (p.head_on[i]) { all_heads |= 1u << i; nh++; }
    // persistent loop over 16-Gaussian tiles; the tiles left over after the last full round are dealt out by head (as in the 32-form)
    const int ntiles = d.ntiles * 2;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nunits = ntiles;
    const int nwaves = (int)gridDim.x * 4, wave_id = (int)blockIdx.x * 4 + wv;
    const int full_rounds = nunits / nwaves, rem = nunits - full_rounds * nwaves;
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
        } else if (tile >= nunits) {
            break;
        }
    }
    int n = n0, q = q0;
    asm volatile("" : "+v"(n), "+v"(q));   // keeps the per-layer weight addresses from being hoisted out of the tile loop
    const size_t tile_n0 = (size_t)tile * 16;
    const int g_raw = tile * 16 + n;
    const bool live = g_raw < p.N;
    const int g = live ? g_raw : p.N - 1;
    float qc[4], xyz[3];
    load_query(p, d.sc, g, qc, xyz);
    // (the per-Gaussian inputs of the epilogues are loaded where they are used: their latency is the other wave's time, their registers
    // would be this wave's for the whole tile)
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
    Trunk16<FU, OT, 2> T0;
    T0.setup(d.packed, p.b0, n, q, lane_);
    T0.preload();                    // (in flight under the gather)
    D1_TICK(0);
    float4 feat[FU];
#pragma unroll
    for (int u = 0; u < FU; u++) {
        feat[u] = gather_group16(p, u, q, qc);
        // one group (24 texel requests = 96 registers) in flight at a time: the next group's addresses are made to depend on this group's
        // result (sched_barrier does not order loads at the IR level, and two groups in flight spill)
        // ... and the blend is pinned HERE (an opaque use of its result): instruction sinking otherwise moves it down to the trunk product,
        // and the 48 texel registers stay live across the trunk's and the first head's operand requests
        asm volatile("" : "+v"(feat[u].x), "+v"(feat[u].y), "+v"(feat[u].z), "+v"(feat[u].w));
        if (u + 1 < FU) asm volatile("" : "+v"(qc[0]), "+v"(qc[1]), "+v"(qc[2]) : "v"(feat[u].x));
    }
    D1_TICK(1);
    // the first head's operand requests stay behind the gather's blend (registers, see above): a compiler-level fence for the IR passes
    // (sched_barrier only binds the machine scheduler, and by then the loads have been hoisted); the trunk's 32 registers of operands
    // were requested in front of the gather on purpose
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    int hd = __builtin_amdgcn_readfirstlane(next_head_m(head_mask, -1));     // (wave-uniform: scalar base addresses for the streams)
    Dense16<KT, OT, true, PD1, true> L1;
    const float* packed_w1 = d.packed + (size_t)d.F * W;
    if (hd < FDGS_NUM_HEADS) { L1.setup_packed(packed_w1 + (size_t)hd * W * W, lane_); L1.preload(); }
    const size_t g_row = (size_t)g_raw;     // saved rows are indexed by the un-clamped Gaussian slot (< Npad)
    if (d.sv_feat && primary) {
#pragma unroll
        for (int u = 0; u < FU; u++) store_saved16(d.sv_feat + g_row * d.F + 16 * u + 4 * q, feat[u]);
    }
    f32x4 hid[WT16];
    FDGS_PRIO_MFMA(true);
    T0.run(feat, hid);
    FDGS_PRIO_MFMA(false);
#pragma unroll
    for (int t = 0; t < WT16; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) hid[t][r] = fmaxf(hid[t][r], 0.f);   // every consumer of the trunk output starts with ReLU
    // saved activations leave through the wave's LDS tile: parked in feature order, copied out lane-consecutively (1 KB per store), one
    // piece per stage of the NEXT hidden layer's product
    float* pending_dst = nullptr;
    auto park = [&](const f32x4* x, float* dst) {
        park16<WT16>(my_tile, LDW, x, n, q);
        pending_dst = dst;
    };
    auto drain_piece = [&](int j) {
        if (pending_dst && j < NPIECE) {
            const int e4 = j * 64 + lane, row = e4 / (W / 4), c4 = e4 - row * (W / 4);
            store_saved16(pending_dst + 4 * (size_t)e4, *reinterpret_cast<const float4*>(my_tile + row * LDW + 4 * c4));
        }
    };
    if (d.sv_rh && primary) {
        park(hid, d.sv_rh + tile_n0 * W);
        if (d.sv_hmask) {
            // the backward's ReLU bits of the trunk output in ITS lane layout (32-Gaussian tiles, lane (g32, h), word t bit r = feature
            // T32 * rho(r, h) + t): read back from the parked rows by lane groups q = 0 (h = 0) and q = 1 (h = 1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (q < 2) {
                constexpr int T32 = W / 32;
                uint32_t m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float* src = my_tile + n * LDW + T32 * rho(r, q);
#pragma unroll
                    for (int t = 0; t < T32; t++) m[t] |= (src[t] > 0.f ? 1u : 0u) << r;
                }
                const size_t t32 = (size_t)(tile >> 1);
                reinterpret_cast<uint4*>(d.sv_hmask)[t32 * 64 + 32 * q + 16 * (tile & 1) + n] = make_uint4(m[0], m[1], m[2], m[3]);
            }
        }
    }
    const bool writer = live && q == 0;
    // the per-Gaussian inputs of a head's epilogue are requested in front of that head's hidden-layer product (12 registers for its
    // duration) -- not at the top of the tile (registers for the whole tile), not in the epilogue (a memory round trip in the open)
    float ein[12];
    auto request_inputs = [&](int hd_) {
        if (hd_ == FDGS_HEAD_SCALE) {
#pragma unroll
            for (int i = 0; i < 3; i++) ein[i] = p.scales[3 * (size_t)g + i];
        } else if (hd_ == FDGS_HEAD_ROT) {
            const float4 r4 = reinterpret_cast<const float4*>(p.rotations)[g];
            ein[0] = r4.x; ein[1] = r4.y; ein[2] = r4.z; ein[3] = r4.w;
        } else if (hd_ == FDGS_HEAD_OPACITY) {
            ein[0] = p.opacity[g];
        } else if (hd_ == FDGS_HEAD_SHS) {      // rows 16 ot + 4q + i of the [16,3] SH block of this Gaussian
#pragma unroll
            for (int ot = 0; ot < 3; ot++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int m = 16 * ot + 4 * q + i;
                    ein[4 * ot + i] = m < 3 ? p.shs_dc[(size_t)p.shs_dc_stride * g + m] : p.shs_rest[(size_t)p.shs_rest_stride * g + (m - 3)];
                }
        }
    };
    auto epilogue_small = [&](int hd_, const f32x4& o) {
        if (!writer) return;
        const float* in_sc = ein;
        const float4 in_rot = make_float4(ein[0], ein[1], ein[2], ein[3]);
        const float in_op = ein[0];
        if (hd_ == FDGS_HEAD_POS) {
            d.out.xyz[3 * (size_t)g] = xyz[0] + o[0]; d.out.xyz[3 * (size_t)g + 1] = xyz[1] + o[1]; d.out.xyz[3 * (size_t)g + 2] = xyz[2] + o[2];
        } else if (hd_ == FDGS_HEAD_SCALE) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const float v = in_sc[i] + o[i];
                d.out.scales[3 * (size_t)g + i] = p.activate ? __expf(v) : v;
            }
        } else if (hd_ == FDGS_HEAD_ROT) {
            float v0 = in_rot.x + o[0], v1 = in_rot.y + o[1], v2 = in_rot.z + o[2], v3 = in_rot.w + o[3];
            if (p.activate) {
                const float nrm = sqrtf(v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3);
                const float inv = 1.0f / fmaxf(nrm, 1e-12f);  // F.normalize eps (scene/gaussian_model.py:44)
                v0 *= inv; v1 *= inv; v2 *= inv; v3 *= inv;
                if (d.out.rot_norm) d.out.rot_norm[g] = nrm;
            }
            reinterpret_cast<float4*>(d.out.rotations)[g] = make_float4(v0, v1, v2, v3);
        } else {
            const float v = in_op + o[0];
            d.out.opacity[g] = p.activate ? sigmoidf_(v) : v;
        }
    };
    // shs [N,16,3] = cat(features_dc, features_rest) (+ delta): lane (n, q) holds rows 16 ot + 4q .. + 3 of tile ot
    auto epilogue_sh = [&](const f32x4* o) {
        if (!live) return;
        const float* in_sh = ein;
#pragma unroll
        for (int ot = 0; ot < 3; ot++)
            *reinterpret_cast<float4*>(d.out.shs + 48 * (size_t)g + 16 * ot + 4 * q) =
                make_float4(in_sh[4 * ot] + o[ot][0], in_sh[4 * ot + 1] + o[ot][1], in_sh[4 * ot + 2] + o[ot][2], in_sh[4 * ot + 3] + o[ot][3]);
    };
    if (primary) {      // a switched-off head returns its input unchanged (scene/deformation.py:106-146)
        const f32x4 z = zero4();
        const f32x4 z3[3] = {z, z, z};
        for (int h0 = 0; h0 < FDGS_NUM_HEADS; h0++) {
            if (p.head_on[h0]) continue;
            request_inputs(h0);
            if (h0 == FDGS_HEAD_SHS) epilogue_sh(z3); else epilogue_small(h0, z);
        }
    }
    D1_TICK(2);
    while (hd < FDGS_NUM_HEADS) {
        const int k = head_k(hd);
        const float* w2h = w2lds + head_row0(hd) * LDW;
        request_inputs(hd);
        const int row2 = (lane & 3) < k ? (lane & 3) : k - 1;
        const float bias2 = k <= 4 ? p.b2[hd][row2] : 0.f;
        const float* wr = w2h + row2 * LDW + KT * 4 * q;                 // k <= 4: A-lane 4b + i = W2[i][features of lane group q]
        int rsh[3];                                                      // SH head: rows 16 ot + n of its 48 (standard row order)
#pragma unroll
        for (int ot = 0; ot < 3; ot++) rsh[ot] = ((16 * ot + n) < k ? (16 * ot + n) : k - 1) * LDW + KT * 4 * q;
        const float* b1h = b1lds + hd * W + KT * 4 * q;
        float* h1_dst = d.sv_h1 ? d.sv_h1 + ((size_t)d.head_slot[hd] * d.Npad + tile_n0) * W : nullptr;
        // second-layer accumulators, carried over the output groups of the hidden layer (each group = 4 of its KT k-tiles)
        f32x4 acc4[4] = {zero4(), zero4(), zero4(), zero4()};
        f32x4 osh[3] = {zero4(), zero4(), zero4()};
        const int nxt_r = __builtin_amdgcn_readfirstlane(next_head_m(head_mask, hd));
#pragma unroll
        for (int oh = 0; oh < OH; oh++) {
            f32x4 y[OG];
            FDGS_PRIO_MFMA(true);
            if (oh == 0) L1.run_group(0, hid, y, [&](int j) { drain_piece(j); });      // (the previous layer's parked tile leaves under group 0)
            else L1.run_group(oh, hid, y, [&](int) {});
            FDGS_PRIO_MFMA(false);
            D1_TICK(3);
            // bias + ReLU: register r of k-tile t = 4 oh + c holds feature KT (4q + r) + 4 oh + c
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float4 b = *reinterpret_cast<const float4*>(b1h + KT * r + 4 * oh);
#pragma unroll
                for (int c = 0; c < 4; c++) y[c][r] = fmaxf(y[c][r] + f4c(b, c), 0.f);
            }
            if (h1_dst) {      // parked in feature order (the memory layout of the saved activations)
                if (oh == 0) pending_dst = nullptr;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    *reinterpret_cast<float4*>(my_tile + n * LDW + KT * (4 * q + r) + 4 * oh) = make_float4(y[0][r], y[1][r], y[2][r], y[3][r]);
                if (oh == OH - 1) pending_dst = h1_dst;
            }
            // this group's share of the second layer
            if (k <= 4) {
                // k <= 4 output rows on v_mfma_f32_4x4x1_16b: block b = lane / 4 holds four Gaussians of lane group q = b / 4 (B = the IL16
                // register as it is), A-lane 4b + i = W2[i][feature of this group]; the four lane groups hold partial sums over their features
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float4 a = *reinterpret_cast<const float4*>(wr + KT * r + 4 * oh);
#pragma unroll
                    for (int c = 0; c < 4; c++) acc4[c] = mfma4(f4c(a, c), y[c][r], acc4[c]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float4 a[3];
#pragma unroll
                    for (int ot = 0; ot < 3; ot++) a[ot] = *reinterpret_cast<const float4*>(w2h + rsh[ot] + KT * r + 4 * oh);
#pragma unroll
                    for (int c = 0; c < 4; c++)
#pragma unroll
                        for (int ot = 0; ot < 3; ot++) osh[ot] = mm16(f4c(a[ot], c), y[c][r], osh[ot]);
                }
            }
            D1_TICK(4);
        }
        const int nxt = nxt_r;
        if (nxt < FDGS_NUM_HEADS) { L1.setup_packed(packed_w1 + (size_t)nxt * W * W, lane_); L1.preload(); }
        if (k <= 4) {
            const f32x4 sum = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float tot = sum[i] + __shfl_xor(sum[i], 16, 64);
                tot += __shfl_xor(tot, 32, 64);
                o[i] = tot + __shfl(bias2, i, 4);      // bias of row i sits in the lanes with (lane & 3) == i
            }
            D1_TICK(5);
            epilogue_small(hd, o);
        } else {
            // lane (n, q) holds rows 16 ot + 4q + r of the SH delta: their biases
#pragma unroll
            for (int ot = 0; ot < 3; ot++)
#pragma unroll
                for (int r = 0; r < 4; r++) { const int m = 16 * ot + 4 * q + r; osh[ot][r] += p.b2[hd][m < k ? m : k - 1]; }
            D1_TICK(5);
            epilogue_sh(osh);
        }
        D1_TICK(6);
        hd = nxt;
    }
    // the last parked tile has no following layer to hide under
#pragma unroll
    for (int j = 0; j < NPIECE; j++) drain_piece(j);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    D1_TICK(7);
    }   // tile loop
#ifdef FDGS_PROFILE_D1
    if (d.prof && lane == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(&d.prof[i], pacc[i]);
        atomicAdd(&d.prof[8], 1ull);
    }
#endif
}
