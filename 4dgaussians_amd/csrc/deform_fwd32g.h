// deform_fwd32g.h -- D1 in its GROUP-WISE 32-Gaussian form (FDGS_D1_FORM=33).  Included by deform.hip inside namespace fdgs, after the
// 32-Gaussian kernel and deform_fwd16.h, whose helpers (gather arithmetic, DenseTrunk, epilogue conventions, saved-activation formats) it shares.
//
// Why a third form (late round 4).  What the first two taught:
//   * in the frame the forward starts on cold planes and operand streams, and two waves per SIMD (the 16-form) beat one (the 32-form) by 5 - 7 %;
//   * a global_load costs an MFMA stream its ISSUE (tools/mfma_chain_probe.hip: ~26 pipe cycles per request alone, ~90 with a second wave on
//     the SIMD), and the 16-form issues one request per FOUR 32-cycle MFMAs -- twice the request density of the 32-form (one per four 64-cycle MFMAs).
// This form keeps the 32-form's arithmetic intensity per request (one wave = 32 Gaussians on v_mfma_f32_32x32x2_f32: an operand register feeds
// a 64-cycle MFMA) AND the 16-form's two waves per SIMD, by evaluating a head's hidden layer ONE 32-row output tile at a time ("group-wise",
// as the 16-form does): 16 accumulator registers instead of 64, the tile's share of the second layer taken right behind its ReLU, the tile
// parked and drained under the next tile's product -- 256 registers, two 256-thread workgroups per CU.
// Layouts: trunk output in the interleaved layout of deform.hip (tile t, register r of lane (g, h) = feature T rho(r, h) + t), so that a first-layer
// row is read with one 16-byte load per k-walk step feeding T MFMAs; the hidden layer's OUTPUT tile ot holds rows 32 ot + rho(r, h) (standard
// order): its 32 features are contiguous in the saved-activation row (128-byte pieces) and the second-layer weights of the group are
// four ds_read_b128 per output row tile.  Same memory formats in and out as the other forms.
#ifndef FDGS_D32G_PD
#define FDGS_D32G_PD 4            // requests in flight (k-walk steps of 4 x 64 MFMA cycles); must divide 16 WT: the ring runs on across heads
#endif

// the HexPlane features of lane (g, h), chunk pair (j0, j0 + 1) of one level, in two batches of three planes (24 requests = 96 registers in
// flight): the arithmetic of gather_chunk_pair (same order of operations, same feature bits)
__device__ __forceinline__ void gather_pair_lean(const fdgs_deform_params& p, int j0, int h, const float* q, float4& out0, float4& out1) {
    const int lvl = __builtin_amdgcn_readfirstlane((8 * j0) / p.C);
    const int c0 = 8 * j0 + 4 * h - lvl * p.C;
    AxisSample S[4];
#pragma unroll
    for (int ax = 0; ax < 4; ax++) S[ax] = axis_sample(q[ax], p.res[lvl][ax]);
    out0 = make_float4(1.f, 1.f, 1.f, 1.f); out1 = out0;
#pragma unroll
    for (int kb = 0; kb < 6; kb += 3) {
        float4 v[3][4], u[3][4];
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            int a, b;
            plane_axes(kb + kk, a, b);
            const int Wd = p.res[lvl][a];
            const AxisSample sx = S[a], sy = S[b];
            const char* P = reinterpret_cast<const char*>(p.planes[lvl][kb + kk]);
            const uint32_t texel = (uint32_t)p.C * 4u, cb = (uint32_t)c0 * 4u;
            const uint32_t r0 = (uint32_t)(sy.i0 * Wd) * texel + cb, r1 = (uint32_t)(sy.i1 * Wd) * texel + cb;
            const uint32_t x0 = (uint32_t)sx.i0 * texel, x1 = (uint32_t)sx.i1 * texel;
            v[kk][0] = *reinterpret_cast<const float4*>(P + (r0 + x0)); u[kk][0] = *reinterpret_cast<const float4*>(P + (r0 + x0 + 32u));
            v[kk][1] = *reinterpret_cast<const float4*>(P + (r0 + x1)); u[kk][1] = *reinterpret_cast<const float4*>(P + (r0 + x1 + 32u));
            v[kk][2] = *reinterpret_cast<const float4*>(P + (r1 + x0)); u[kk][2] = *reinterpret_cast<const float4*>(P + (r1 + x0 + 32u));
            v[kk][3] = *reinterpret_cast<const float4*>(P + (r1 + x1)); u[kk][3] = *reinterpret_cast<const float4*>(P + (r1 + x1 + 32u));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            int a, b;
            plane_axes(kb + kk, a, b);
            const AxisSample sx = S[a], sy = S[b];
            const float w00 = sx.w0 * sy.w0, w01 = sx.w1 * sy.w0, w10 = sx.w0 * sy.w1, w11 = sx.w1 * sy.w1;
            out0.x *= v[kk][0].x * w00 + v[kk][1].x * w01 + v[kk][2].x * w10 + v[kk][3].x * w11;
            out0.y *= v[kk][0].y * w00 + v[kk][1].y * w01 + v[kk][2].y * w10 + v[kk][3].y * w11;
            out0.z *= v[kk][0].z * w00 + v[kk][1].z * w01 + v[kk][2].z * w10 + v[kk][3].z * w11;
            out0.w *= v[kk][0].w * w00 + v[kk][1].w * w01 + v[kk][2].w * w10 + v[kk][3].w * w11;
            out1.x *= u[kk][0].x * w00 + u[kk][1].x * w01 + u[kk][2].x * w10 + u[kk][3].x * w11;
            out1.y *= u[kk][0].y * w00 + u[kk][1].y * w01 + u[kk][2].y * w10 + u[kk][3].y * w11;
            out1.z *= u[kk][0].z * w00 + u[kk][1].z * w01 + u[kk][2].z * w10 + u[kk][3].z * w11;
            out1.w *= u[kk][0].w * w00 + u[kk][1].w * w01 + u[kk][2].w * w10 + u[kk][3].w * w11;
        }
        // (the second batch's addresses are made to depend on the first batch's result: sched_barrier does not order loads at the IR level)
        if (kb == 0) asm volatile("" : "+v"(S[0].i0), "+v"(S[1].i0), "+v"(S[2].i0) : "v"(out0.x), "v"(out1.x));
    }
}

// The heads' W1 as the operand stream of this form: float4 index ((hd * 16 WT + idx) * 64 + lane), idx = 16 ot + s, lane (g, h) =
// W1[hd][32 ot + g][WT (4h + rho(s, 0)) .. + 3]  (WT = 4; one contiguous 1-KB request per k-walk step instead of 32 cache lines)
struct Pack32gArgs { const float* w1[FDGS_NUM_HEADS]; int head_on[FDGS_NUM_HEADS]; int W; float* out; };
__global__ void __launch_bounds__(256) pack_weights32g_kernel(Pack32gArgs a) {
    const int W = a.W, WT = W / 32, NI = 16 * WT;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= FDGS_NUM_HEADS * NI * 64) return;
    const int lane = e & 63, idx = (e >> 6) % NI, hd = e / (64 * NI);
    if (!a.head_on[hd]) return;
    const int g = lane & 31, h = lane >> 5, ot = idx >> 4, s = idx & 15;
    reinterpret_cast<float4*>(a.out)[e] = *reinterpret_cast<const float4*>(a.w1[hd] + (size_t)(32 * ot + g) * W + WT * (4 * h + rho(s, 0)));
}

template <int WT, int FCH, bool PACKED = false>
__global__ void __launch_bounds__(256, 2) deform_fwd32g_kernel(DeformDev d) {
    constexpr int W = 32 * WT, FT = (FCH + 3) / 4, PD = FDGS_D32G_PD;
    static_assert((16 * WT) % PD == 0, "the request ring keeps its phase from one head to the next");
    constexpr int LDW = W + 4;        // LDS row stride of the staged second-layer weights (rows of a 4x4x1 product in distinct banks)
    constexpr int PTS = 36;           // row stride of a parked [32 Gaussians][32 features] tile
    constexpr int RPP = 16 / WT;      // trunk-output registers per 32-feature block of the saved relu(hidden) row
    const fdgs_deform_params& p = d.p;
    __shared__ __attribute__((aligned(16))) float lds[4 * 32 * PTS + 59 * LDW + FDGS_NUM_HEADS * W];
    float* my_tile = lds + (threadIdx.x >> 6) * 32 * PTS;
    float* w2lds = lds + 4 * 32 * PTS;
    float* b1lds = w2lds + 59 * LDW;
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
    const int lane = threadIdx.x & 63, g0 = lane & 31, h0 = lane >> 5;
    unsigned all_heads = 0u;
    int nh = 0;
#pragma unroll
    for (int i = 0; i < FDGS_NUM_HEADS; i++) if (p.head_on[i]) { all_heads |= 1u << i; nh++; }
    // persistent loop over 32-Gaussian tiles, leftover tiles dealt out by head (as in deform_fwd_kernel)
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
    const size_t tile_n0 = (size_t)tile * 32;
    const int n_raw = tile * 32 + g;
    const bool live = n_raw < p.N;
    const int n = live ? n_raw : p.N - 1;
    DenseTrunk<FCH, WT, 2> T0;
    T0.setup(p.w0, p.b0, d.F, g, h);
    T0.preload();
    float q[4], xyz[3];
    load_query(p, d.sc, n, q, xyz);
    f32x16 feat[FT];
#pragma unroll
    for (int t = 0; t < FT; t++) feat[t] = zero16();
    {
        auto put = [&](int j, const float4& v) {
            feat[j / 4][4 * (j % 4) + 0] = v.x; feat[j / 4][4 * (j % 4) + 1] = v.y;
            feat[j / 4][4 * (j % 4) + 2] = v.z; feat[j / 4][4 * (j % 4) + 3] = v.w;
        };
#pragma unroll
        for (int j = 0; j < FCH; j += 2) {
            float4 v0, v1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j + 1 < FCH && (8 * j) / p.C == (8 * (j + 1)) / p.C) {   // (wave-uniform) both chunks in one level
                gather_pair_lean(p, j, h, q, v0, v1);
            } else {
                v0 = gather_chunk(p, j, h, q);
                if (j + 1 < FCH) { asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]) : "v"(v0.x)); v1 = gather_chunk(p, j + 1, h, q); }
            }
            put(j, v0);
            if (j + 1 < FCH) put(j + 1, v1);
            // one pair's requests in flight at a time: the next pair's addresses depend on this pair's result
            if (j + 2 < FCH) asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]) : "v"(v0.x), "v"(v1.x));
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const size_t n_row = (size_t)n_raw;
    if (d.sv_feat && primary) {
#pragma unroll
        for (int j = 0; j < FCH; j++)
            *reinterpret_cast<float4*>(d.sv_feat + n_row * d.F + 8 * j + 4 * h) =
                make_float4(feat[j / 4][4 * (j % 4)], feat[j / 4][4 * (j % 4) + 1], feat[j / 4][4 * (j % 4) + 2], feat[j / 4][4 * (j % 4) + 3]);
    }
    // ---- first-layer operand stream of the heads: row-major W1, lane (g, h) reads row 32 ot + g, floats WT (4h + rho(s, 0)) .. + WT - 1 at
    // k-walk step s; index = 16 ot + s runs on across the output tiles and into the next head (requests PD steps ahead)
    int hd = __builtin_amdgcn_readfirstlane(next_head_m(head_mask, -1));
    const int first_hd = hd;
    static_assert(!PACKED || WT == 4, "packed operand stream: float4 per lane and step");
    const size_t lane_w1 = PACKED ? (size_t)lane * 4 : (size_t)g * W + (size_t)(WT * 4) * h;
    auto head_base = [&](int hd_) -> const float* {
        return (PACKED ? d.packed + (size_t)hd_ * (16 * WT * 256) : p.w1[hd_]) + lane_w1;
    };
    const float* hbase = head_base(hd < FDGS_NUM_HEADS ? hd : 0);
    const float* nbase = hbase;
    AVec<WT> ring[PD];
    auto fetch = [&](int idx, AVec<WT>& dst) {      // idx compile-time: < 16 WT this head, else the next head
        constexpr int NI = 16 * WT;
        const float* b = idx < NI ? hbase : nbase;
        const int i = idx < NI ? idx : idx - NI, ot = i >> 4, s = i & 15;
        if constexpr (PACKED) dst = ldv<WT>(b + (size_t)i * 256);
        else dst = ldv<WT>(b + (size_t)ot * 32 * W + WT * rho(s, 0));
    };
    if (hd < FDGS_NUM_HEADS) {
#pragma unroll
        for (int s = 0; s < PD; s++) fetch(s, ring[s]);
    }
    f32x16 hid[WT];
    T0.run(feat, hid, h);
    relu_inplace<WT>(hid);  // every consumer of the trunk output starts with ReLU (scene/deformation.py:61-65)
    if (d.sv_hmask && primary) {   // the backward's ReLU mask of the trunk output, in its own lane layout: one 16-byte load there
        uint32_t m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < WT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) m[t] |= (hid[t][r] > 0.f ? 1u : 0u) << r;
        reinterpret_cast<uint4*>(d.sv_hmask)[(size_t)(tile_n0 / 32) * 64 + lane] = make_uint4(m[0], m[1], m[2], m[3]);
    }
    // ---- parked [32][32] tiles: one at a time in the wave's LDS tile, copied out (128 contiguous bytes per Gaussian row, 1 KB per store) in
    // the hooks of the NEXT output tile's product (the hook of step s sits in front of that step's four 64-cycle MFMAs)
    float* pend = nullptr;           // destination of the parked tile's first row, column offset applied
    // the four 1-KB pieces of the parked tile leave in ONE burst (four LDS reads, one wait, four stores).  vmcnt counts loads and stores
    // in issue order: with one store per k-walk step (the first version) every operand wait of the following steps also waited for the
    // acknowledgement of a store issued a step or two earlier; after a burst the next PD steps consume requests that are OLDER than the
    // stores, and by the time a younger request is consumed the stores are a thousand cycles old
    auto drain_all = [&]() {
        if (pend) {
            // piece jp = float4 elements jp * 64 + lane of the [32][8] tile: row 8 jp + (lane >> 3), column lane & 7
            const float* src = my_tile + (lane >> 3) * PTS + 4 * (lane & 7);
            float* dst = pend + (size_t)(lane >> 3) * W + 4 * (lane & 7);
            const float4 t0 = *reinterpret_cast<const float4*>(src), t1 = *reinterpret_cast<const float4*>(src + 8 * PTS);
            const float4 t2 = *reinterpret_cast<const float4*>(src + 16 * PTS), t3 = *reinterpret_cast<const float4*>(src + 24 * PTS);
            *reinterpret_cast<float4*>(dst) = t0; *reinterpret_cast<float4*>(dst + 8 * W) = t1;
            *reinterpret_cast<float4*>(dst + 16 * W) = t2; *reinterpret_cast<float4*>(dst + 24 * W) = t3;
        }
    };
    auto park_tile = [&](const f32x16& y, float* dst) {      // standard tile: register r = feature rho(r, h) of the 32
#pragma unroll
        for (int j = 0; j < 4; j++)
            *reinterpret_cast<float4*>(my_tile + g * PTS + 8 * j + 4 * h) = make_float4(y[4 * j], y[4 * j + 1], y[4 * j + 2], y[4 * j + 3]);
        pend = dst;
    };
    auto park_rh = [&](int jb, float* dst) {                 // 32-feature block jb of relu(hidden): registers RPP jb .. + RPP - 1 of the WT tiles
#pragma unroll
        for (int x = 0; x < RPP; x++) {
            const int r = RPP * jb + x, off = WT * ((x & 3) + 8 * (x >> 2) + 4 * h);
            if constexpr (WT == 4) *reinterpret_cast<float4*>(my_tile + g * PTS + off) = make_float4(hid[0][r], hid[1][r], hid[2][r], hid[3][r]);
            else *reinterpret_cast<float2*>(my_tile + g * PTS + off) = make_float2(hid[0][r], hid[1][r]);
        }
        pend = dst;
    };
    const bool writer = live && h == 0;
    float ein[24];
    auto request_inputs = [&](int hd_) {
        if (hd_ == FDGS_HEAD_SCALE) {
#pragma unroll
            for (int i = 0; i < 3; i++) ein[i] = p.scales[3 * (size_t)n + i];
        } else if (hd_ == FDGS_HEAD_ROT) {
            const float4 r4 = reinterpret_cast<const float4*>(p.rotations)[n];
            ein[0] = r4.x; ein[1] = r4.y; ein[2] = r4.z; ein[3] = r4.w;
        } else if (hd_ == FDGS_HEAD_OPACITY) {
            ein[0] = p.opacity[n];
        } else if (hd_ == FDGS_HEAD_SHS) {
            // rows (u < 4 ? 0 : 32) + 8 (u & 3) + 4h .. + 3 of cat(features_dc [3], features_rest [45]); only u = 0 touches features_dc: the
            // other five groups are ONE lane pointer + immediate offsets (per-element pointer selects were hoisted out of the head loop as
            // 24 64-bit registers and spilled)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int m = 4 * h + i;
                ein[i] = m < 3 ? p.shs_dc[(size_t)p.shs_dc_stride * n + m] : p.shs_rest[(size_t)p.shs_rest_stride * n + (m - 3)];
            }
            const float* rest = p.shs_rest + ((size_t)p.shs_rest_stride * n + 4 * h - 3);
#pragma unroll
            for (int u = 1; u < 6; u++)
#pragma unroll
                for (int i = 0; i < 4; i++) ein[4 * u + i] = rest[(u < 4 ? 0 : 32) + 8 * (u & 3) + i];
        }
    };
    auto epilogue_small = [&](int hd_, const f32x4& o) {
        if (!writer) return;
        if (hd_ == FDGS_HEAD_POS) {
            d.out.xyz[3 * (size_t)n] = xyz[0] + o[0]; d.out.xyz[3 * (size_t)n + 1] = xyz[1] + o[1]; d.out.xyz[3 * (size_t)n + 2] = xyz[2] + o[2];
        } else if (hd_ == FDGS_HEAD_SCALE) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const float v = ein[i] + o[i];
                d.out.scales[3 * (size_t)n + i] = p.activate ? __expf(v) : v;
            }
        } else if (hd_ == FDGS_HEAD_ROT) {
            float v0 = ein[0] + o[0], v1 = ein[1] + o[1], v2 = ein[2] + o[2], v3 = ein[3] + o[3];
            if (p.activate) {
                const float nrm = sqrtf(v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3);
                const float inv = 1.0f / fmaxf(nrm, 1e-12f);  // F.normalize eps (scene/gaussian_model.py:44)
                v0 *= inv; v1 *= inv; v2 *= inv; v3 *= inv;
                if (d.out.rot_norm) d.out.rot_norm[n] = nrm;
            }
            reinterpret_cast<float4*>(d.out.rotations)[n] = make_float4(v0, v1, v2, v3);
        } else {
            const float v = ein[0] + o[0];
            d.out.opacity[n] = p.activate ? sigmoidf_(v) : v;
        }
    };
    // shs [N,16,3] = cat(features_dc, features_rest) (+ delta): rows 8u+4h..+3 of tile 0 (u<4) and tile 1 (u<2)
    auto epilogue_sh = [&](const f32x16& o0, const f32x16& o1) {
        if (!live) return;
#pragma unroll
        for (int u = 0; u < 6; u++) {
            const int row0 = (u < 4 ? 0 : 32) + 8 * (u & 3) + 4 * h;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = ein[4 * u + i] + (u < 4 ? o0[4 * (u & 3) + i] : o1[4 * (u & 3) + i]);
            *reinterpret_cast<float4*>(d.out.shs + 48 * (size_t)n + row0) = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    if (primary) {      // a switched-off head returns its input unchanged (scene/deformation.py:106-146)
        const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x16 z = zero16();
        for (int h0_ = 0; h0_ < FDGS_NUM_HEADS; h0_++) {
            if (p.head_on[h0_]) continue;
            request_inputs(h0_);
            if (h0_ == FDGS_HEAD_SHS) epilogue_sh(z, z); else epilogue_small(h0_, z4);
        }
    }

    while (hd < FDGS_NUM_HEADS) {
        const int k = head_k(hd);
        const float* w2h = w2lds + head_row0(hd) * LDW;
        const float* b1h = b1lds + hd * W;
        const int nxt = __builtin_amdgcn_readfirstlane(next_head_m(head_mask, hd));
        nbase = head_base(nxt < FDGS_NUM_HEADS ? nxt : hd);
        float* h1_dst = d.sv_h1 ? d.sv_h1 + ((size_t)d.head_slot[hd] * d.Npad + tile_n0) * W : nullptr;
        const bool do_rh = hd == first_hd && d.sv_rh != nullptr && primary;
        float* rh_dst = d.sv_rh ? d.sv_rh + tile_n0 * W : nullptr;
        request_inputs(hd);
        // one output tile of the hidden layer: Y = b1 + W1[rows 32 ot ..] hid, then ReLU
        auto hidden_tile = [&](int ot, f32x16& Y) {
            Y = mfma32(h == 0 ? b1h[32 * ot + g] : 0.f, 1.0f, zero16());
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const int idx = 16 * ot + s;
                const AVec<WT> cur = ring[idx % PD];
                fetch(idx + PD, ring[idx % PD]);
                // hooks: step 0 sends the parked tile out; the first head of a tile also parks (step 4) and sends (step 8) block `ot` of relu(hidden)
                if (s == 0) drain_all();
                if (s == 4 && do_rh) park_rh(ot, rh_dst + 32 * ot);
                if (s == 8 && do_rh) { drain_all(); pend = nullptr; }
                __builtin_amdgcn_sched_barrier(0);   // keep the requests PD steps ahead (the scheduler sinks them to their use otherwise)
#pragma unroll
                for (int t = 0; t < WT; t++) Y = mfma32(cur.v[t], hid[t][s], Y);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) Y[r] = fmaxf(Y[r], 0.f);
        };
        if (k <= 4) {
            // second layer of a k <= 4 head on v_mfma_f32_4x4x1_16b: block b = lane / 4 holds four Gaussians (B = the tile register as it
            // is), A-lane 4b + i = W2[i][32 ot + rho(r, h)]; the two lane halves hold partial sums over their halves of the features
            const int row2 = (lane & 3) < k ? (lane & 3) : k - 1;
            const float bias2 = p.b2[hd][row2];
            const float* wr = w2h + row2 * LDW + 4 * h;
            f32x4 acc4[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ot = 0; ot < WT; ot++) {
                f32x16 Y;
                hidden_tile(ot, Y);
                if (h1_dst) park_tile(Y, h1_dst + 32 * ot);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float4 a = *reinterpret_cast<const float4*>(wr + 32 * ot + 8 * j);
                    acc4[0] = mfma4(a.x, Y[4 * j + 0], acc4[0]); acc4[1] = mfma4(a.y, Y[4 * j + 1], acc4[1]);
                    acc4[2] = mfma4(a.z, Y[4 * j + 2], acc4[2]); acc4[3] = mfma4(a.w, Y[4 * j + 3], acc4[3]);
                }
            }
            const f32x4 sum = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float tot = sum[i] + __shfl_xor(sum[i], 32, 64);
                o[i] = tot + __shfl(bias2, i, 4);      // bias of row i sits in the lanes with (lane & 3) == i
            }
            epilogue_small(hd, o);
        } else {
            // the SH head's 48 rows as two 32-row tiles (rows beyond k are duplicates of the last row, never read back)
            const int r0 = g < k ? g : k - 1, r1 = (32 + g) < k ? (32 + g) : k - 1;
            f32x16 o0 = mfma32(h == 0 ? p.b2[hd][r0] : 0.f, 1.0f, zero16());
            f32x16 o1 = mfma32(h == 0 ? p.b2[hd][r1] : 0.f, 1.0f, zero16());
            const float* wa = w2h + r0 * LDW + 4 * h;
            const float* wb = w2h + r1 * LDW + 4 * h;
#pragma unroll
            for (int ot = 0; ot < WT; ot++) {
                f32x16 Y;
                hidden_tile(ot, Y);
                if (h1_dst) park_tile(Y, h1_dst + 32 * ot);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float4 a = *reinterpret_cast<const float4*>(wa + 32 * ot + 8 * j);
                    const float4 b = *reinterpret_cast<const float4*>(wb + 32 * ot + 8 * j);
                    o0 = mfma32(a.x, Y[4 * j + 0], o0); o1 = mfma32(b.x, Y[4 * j + 0], o1);
                    o0 = mfma32(a.y, Y[4 * j + 1], o0); o1 = mfma32(b.y, Y[4 * j + 1], o1);
                    o0 = mfma32(a.z, Y[4 * j + 2], o0); o1 = mfma32(b.z, Y[4 * j + 2], o1);
                    o0 = mfma32(a.w, Y[4 * j + 3], o0); o1 = mfma32(b.w, Y[4 * j + 3], o1);
                }
            }
            epilogue_sh(o0, o1);
        }
        // the stream runs on: what was requested beyond this head's last step belongs to the next head
        hbase = nbase;
        hd = nxt;
    }
    // the last parked tile has no following product to hide under
    drain_all();
    pend = nullptr;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }   // tile loop
}
