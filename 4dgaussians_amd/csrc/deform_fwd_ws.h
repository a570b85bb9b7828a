// deform_fwd_ws.h -- D1 in its WEIGHT-STATIONARY form (round 6).  Included by deform.hip (inside namespace fdgs, after deform_fwd16.h whose
// helpers -- mm16, zero4, store_saved16 -- and the shared ones -- axis_sample, plane_axes, load_query, DeformDev, head_k, rho -- it uses).
//
// Why a third form.  The 16- and 32-Gaussian forms give a wave its Gaussians and stream every weight of every layer past them: 374 KB of
// operands per 16 Gaussians, ~7 GB of L2 -> CU requests per launch, and every request costs the matrix-core pipe an issue slot
// (tools/mfma_chain_probe.hip: 13 - 29 % for global_load_dwordx4, 3 % for ds_read_b128) -- SQ_WAIT_INST_ANY 0.67, MFMA busy 0.61, and three
// rounds of tuning the stream (depth, packing, priorities, an LDS ring) moved it by +-1 %.  This form turns the loop inside out: the WEIGHTS
// stay and the Gaussians move.  The NW = W / 16 waves of a workgroup (W = 128: eight waves, two per SIMD, 256 registers each) hold all five
// heads' first-layer matrices in REGISTERS for the whole launch -- wave w owns rows 16 w .. 16 w + 15 of every W1 (5 x W / 4 registers) and
// of W0 -- and a 16-Gaussian tile visits them through LDS:
//     trunk     every wave computes ITS 16 rows of relu(W0 feat + b0) (A = its W0 rows, B = the tile's features, prefetched from memory one
//               tile ahead) and writes them into the tile's [16][W] LDS image;            ONE s_barrier per tile;
//     heads     every wave multiplies its W1 rows with the whole tile: W / 4 k-steps per head, A operands are registers that never change,
//               B operands come from the LDS image with ds_read_b128 (one per four MFMAs); two accumulators per head (even / odd k-steps:
//               a dependent v_mfma_f32_16x16x4_f32 issues every 40 cycles, an independent one every 32);
//     2nd layer every wave multiplies ITS 16 hidden features with the matching W2 columns (k <= 4 heads on v_mfma_f32_4x4x1_16b, the 48-row SH
//               head on three 16-row tiles; W2 from LDS) and leaves the partial sums in LDS; they are added in a fixed order (wave 0 .. NW-1:
//               deterministic) by the epilogue of the NEXT tile's iteration, behind the same barrier -- 256 work items (16 Gaussians x
//               16 four-float chunks) that add the input, apply exp / normalize / sigmoid and store coalesced.
// No operand request goes to L2 in the steady state.  tools/ws_probe.hip (the first-layer products alone, same structure): 0.84 of the f32
// MFMA peak without the saved activations, 0.68 - 0.71 with them (profiles/r06_ws_probe.txt).
// The HexPlane gather is a kernel of its own in this form (deform_gather_kernel: one thread per Gaussian and four channels, features to the
// `saved` buffer -- or to the pack scratch when nothing is saved): it is bound by L2 / texel traffic, not by the matrix cores, and inside a
// register-stationary kernel its 96 registers of texels in flight have no room.
// Same memory formats in and out as the other forms (saved activations, ReLU bit masks in D2's lane layout): D2 / D3 / D4 do not know which
// form ran.  Results differ from the other forms by summation order only (k is walked as (j, c, lane group) here).

struct GatherArgs { fdgs_deform_params p; AabbScale sc; int F, Npad; float* feat; };

// features 4 cq .. 4 cq + 3 of level blockIdx.y of one Gaussian per thread: the product over the six planes of the bilinear samples -- the
// arithmetic of gather_group16 / gather_chunk, operation by operation
__global__ void __launch_bounds__(256) deform_gather_kernel(GatherArgs a) {
    const fdgs_deform_params& p = a.p;
    const int lvl = blockIdx.y, cq_per = p.C >> 2;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long g_raw = id / cq_per;
    if (g_raw >= a.Npad) return;
    const int c0 = 4 * (int)(id - g_raw * cq_per);
    const int g = g_raw < p.N ? (int)g_raw : p.N - 1;
    float qc[4], xyz[3];
    load_query(p, a.sc, g, qc, xyz);
    AxisSample S[4];
#pragma unroll
    for (int ax = 0; ax < 4; ax++) S[ax] = axis_sample(qc[ax], p.res[lvl][ax]);
    float4 prod = make_float4(1.f, 1.f, 1.f, 1.f);
    float4 v[6][4];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        int ax, bx;
        plane_axes(k, ax, bx);
        const int Wd = p.res[lvl][ax];
        const AxisSample sx = S[ax], sy = S[bx];
        const char* P = reinterpret_cast<const char*>(p.planes[lvl][k]);
        const uint32_t texel = (uint32_t)p.C * 4u, cb = (uint32_t)c0 * 4u;
        const uint32_t r0 = (uint32_t)(sy.i0 * Wd) * texel + cb, r1 = (uint32_t)(sy.i1 * Wd) * texel + cb;
        const uint32_t x0 = (uint32_t)sx.i0 * texel, x1 = (uint32_t)sx.i1 * texel;
        v[k][0] = *reinterpret_cast<const float4*>(P + (r0 + x0));
        v[k][1] = *reinterpret_cast<const float4*>(P + (r0 + x1));
        v[k][2] = *reinterpret_cast<const float4*>(P + (r1 + x0));
        v[k][3] = *reinterpret_cast<const float4*>(P + (r1 + x1));
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        int ax, bx;
        plane_axes(k, ax, bx);
        const AxisSample sx = S[ax], sy = S[bx];
        const float w00 = sx.w0 * sy.w0, w01 = sx.w1 * sy.w0, w10 = sx.w0 * sy.w1, w11 = sx.w1 * sy.w1;
        prod.x *= v[k][0].x * w00 + v[k][1].x * w01 + v[k][2].x * w10 + v[k][3].x * w11;
        prod.y *= v[k][0].y * w00 + v[k][1].y * w01 + v[k][2].y * w10 + v[k][3].y * w11;
        prod.z *= v[k][0].z * w00 + v[k][1].z * w01 + v[k][2].z * w10 + v[k][3].z * w11;
        prod.w *= v[k][0].w * w00 + v[k][1].w * w01 + v[k][2].w * w10 + v[k][3].w * w11;
    }
    *reinterpret_cast<float4*>(a.feat + (size_t)g_raw * a.F + lvl * p.C + c0) = prod;
}

// partial-sum / bias rows of the second layers: 64 floats per Gaussian = 16 four-float chunks, the five heads at 16-byte aligned offsets
//   pos 0..2 | scale 4..6 | rot 8..11 | opacity 12 | SH 16..63      (chunk c: 0..3 = the k <= 4 heads, 4..15 = SH rows 4 (c - 4) .. + 3)
// In LDS chunk c of Gaussian n sits at chunk position c ^ n: the writers (a lane group = 16 Gaussians, one chunk) and the epilogue's readers
// (16 Gaussians, one chunk) both touch all 64 banks.

// x + (x of the lanes 16 / 32 / 48 away): the sum over the four lane groups, in every lane.  v_permlane16/32_swap are VALU moves -- a
// __shfl_xor across 16-lane rows is ds_bpermute, an LDS round trip in the middle of the wave's dependent chain.
__device__ __forceinline__ float sum_lane_groups(float x) {
    const unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);       // (r0, r0, r2, r2) , (r1, r1, r3, r3)
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned v = __float_as_uint(y);
    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);       // (lo, lo) , (hi, hi)
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// SAVE: the forward leaves relu(hidden), its ReLU bits and relu(h1) of every head behind for the backward.  ALLH: all five heads are on
// (the per-head tests vanish from the tile loop: straight-line code, exact wait counts).
// FOUR waves per workgroup, each owning RT row tiles of 16: W = 64 RT.  RT = 2 (net_width 128): one wave per SIMD with the whole 512-register
// file -- 320 registers of stationary W1, W0 and the tile's image in registers too, and room left to overlap the phases inside the one
// instruction stream (the eight-wave / 256-register layout of the first probe spilled in the full kernel, and a spill's reload is a vector
// memory wait for every store in flight: tools/ws_probe.hip has both layouts, 0.68 / 0.71 of peak with the saved activations).
template <int RT, int FU, bool SAVE, bool ALLH>
__global__ void __launch_bounds__(256, RT == 2 ? 1 : 2) deform_mlp_ws_kernel(DeformDev d) {
    constexpr int NW = 4, W = 64 * RT, KG = W / 16, LDW = W + 4, NT = 256, PLS = 64;
    constexpr int W2ROWS = 59;
    constexpr int F = 16 * FU, LDF = F + 4, FLD = (16 * F / 4 + NT - 1) / NT;  // feature tile: row stride, float4 per thread of the cooperative load
    const fdgs_deform_params& p = d.p;
    __shared__ __attribute__((aligned(16))) float xl[2][16 * LDW];          // relu(hidden) of the tile, [Gaussian][feature] (= the saved layout)
    __shared__ __attribute__((aligned(16))) float w2l[W2ROWS * LDW];        // the heads' second-layer matrices, row-major
    __shared__ __attribute__((aligned(16))) float b0l[W], b1l[FDGS_NUM_HEADS * W], b2l[64];
    __shared__ __attribute__((aligned(16))) float pl[2][NW][16][PLS];       // second-layer partial sums of the waves (chunk-swizzled rows)
    __shared__ __attribute__((aligned(16))) float fl[2][16 * LDF];          // the features of the next tile(s), [Gaussian][feature]
    __shared__ __attribute__((aligned(16))) float inl[2][256 * 4];          // the epilogue's inputs: one float4 per work item
    __shared__ uint32_t hml[SAVE ? 2 : 1][SAVE ? 16 * 2 * 4 : 1];           // ReLU bits of relu(hidden): [Gaussian][half h][word t] (D2's layout)
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, q = lane >> 4;
    const unsigned hmask = ALLH ? 31u : d.head_mask;
    for (int i = tid; i < W; i += NT) b0l[i] = p.b0[i];
    for (int i = tid; i < FDGS_NUM_HEADS * W; i += NT) b1l[i] = p.head_on[i / W] ? p.b1[i / W][i % W] : 0.f;
    if (tid < 64) {      // the second-layer biases in the layout of the partial-sum rows
        const int c = tid >> 2, i = tid & 3, hd = c < 4 ? c : FDGS_HEAD_SHS, o = c < 4 ? i : tid - 16;
        b2l[tid] = (p.head_on[hd] && o < head_k(hd)) ? p.b2[hd][o] : 0.f;
    }
    if (SAVE) for (int i = tid; i < 2 * 16 * 2 * 4; i += NT) (&hml[0][0])[i] = 0u;
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) {
        if (!p.head_on[hd]) continue;
        const int k_ = head_k(hd), r0_ = head_row0(hd);
        for (int i = tid; i < k_ * (W / 4); i += NT) {
            const int r = i / (W / 4), c4 = i - r * (W / 4);
            *reinterpret_cast<float4*>(w2l + (r0_ + r) * LDW + 4 * c4) = reinterpret_cast<const float4*>(p.w2[hd])[i];
        }
    }
    // the stationary operands: row tile t of this wave = rows 16 (RT w + t) .. + 15; A-lane (i = n, kk = q) of k-step (j, c) holds
    // W[16 (RT w + t) + i][16 j + 4 kk + c]
    float4 w0r[RT][FU], w1r[FDGS_NUM_HEADS][RT][KG];
#pragma unroll
    for (int t = 0; t < RT; t++)
#pragma unroll
        for (int u = 0; u < FU; u++) w0r[t][u] = *reinterpret_cast<const float4*>(p.w0 + (size_t)(16 * (RT * w + t) + n) * F + 16 * u + 4 * q);
#pragma unroll
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++)
#pragma unroll
        for (int t = 0; t < RT; t++)
#pragma unroll
            for (int j = 0; j < KG; j++)
                w1r[hd][t][j] = p.head_on[hd] ? *reinterpret_cast<const float4*>(p.w1[hd] + (size_t)(16 * (RT * w + t) + n) * W + 16 * j + 4 * q)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const int ntiles = d.Npad / 16;
    // ---- the only LOADS of the tile loop: a tile's features (cooperatively, one float4 per thread and piece) and the inputs its epilogue will
    // add the deltas to (work item = (chunk c, Gaussian nn), c-major: one float4 per thread), requested a whole iteration before they are
    // parked in LDS.  One wait point per iteration (loop top), nothing else in the loop ever waits for vector memory: the saved activations
    // leave through stores nobody waits for (a vmcnt wait for a load also waits for every older store -- and a spilled register's reload is
    // a load: this kernel must not spill).
    // WHERE the compiler's wait for these loads lands matters: vector memory completes in order, so "wait until at most K operations are in
    // flight" with K = the stores issued behind the loads (SAVE: >= 12 per iteration) does not wait for those stores, while vmcnt(0) does
    // (measured: 1.8 k of a tile's 20 k ticks).  hipcc derives K from the paths that reach the parking: the loop is entered with NOTHING in
    // flight (the prologue parks what it loads), so the only path with loads in flight is the back edge with its stores behind them.
    // Work item of the epilogue = (four-float chunk c, Gaussian nn), FIXED per thread: every wave takes one of the four k <= 4 heads (16
    // lanes) and three of the twelve SH row chunks (48 lanes) -- the same mix of stores for every wave (chunk-major with wave 0 taking all
    // four small heads had the other three waiting at the barrier).  The item's four inputs come from one of six arrays with its own row
    // stride: which array, which elements and the stride are worked out ONCE per thread; per tile an address is base + row * stride (this
    // wave is alone on its SIMD: every VALU instruction of the tile loop that does not sit in the shadow of an MFMA costs its four cycles in
    // the open, and the per-tile version of this address arithmetic was 1.5 k of a tile's 20 k ticks).
    const int my_c = ((tid >> 4) & 3) == 0 ? (tid >> 6) : 4 + 3 * (tid >> 6) + ((tid >> 4) & 3) - 1, my_nn = tid & 15;
    const float* in_base[4];
    int in_stride[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = 4 * (my_c - 4) + i, i3 = i < 3 ? i : 2;
        if (my_c >= 4) { in_base[i] = m < 3 ? p.shs_dc + m : p.shs_rest + (m - 3); in_stride[i] = m < 3 ? p.shs_dc_stride : p.shs_rest_stride; }
        else if (my_c == FDGS_HEAD_ROT) { in_base[i] = p.rotations + i; in_stride[i] = 4; }
        else if (my_c == FDGS_HEAD_OPACITY) { in_base[i] = p.opacity; in_stride[i] = 1; }
        else if (my_c == FDGS_HEAD_SCALE) { in_base[i] = p.scales + i3; in_stride[i] = 3; }
        else { in_base[i] = p.xyz + i3; in_stride[i] = 3; }
    }
    auto request = [&](int ftile, int itile, float4* fdst, float4& idst) {
#pragma unroll
        for (int e = 0; e < FLD; e++) {
            const int i4 = tid + e * NT;
            if (i4 < 16 * (F / 4) && ftile < ntiles) fdst[e] = reinterpret_cast<const float4*>(d.feat + (size_t)ftile * 16 * F)[i4];
        }
        // four unconditional dword loads straight into the destination -- loads inside an if / else chain are merged through copies, and a
        // copy of a loaded value is a wait where it stands (rows beyond N load duplicates nobody uses)
        int g = (itile < ntiles ? itile : ntiles - 1) * 16 + my_nn;
        g = g < p.N ? g : p.N - 1;
        idst.x = in_base[0][(size_t)g * in_stride[0]]; idst.y = in_base[1][(size_t)g * in_stride[1]];
        idst.z = in_base[2][(size_t)g * in_stride[2]]; idst.w = in_base[3][(size_t)g * in_stride[3]];
    };
    auto park = [&](const float4* fsrc, const float4& isrc, float* fimg, float* iimg) {      // (the inputs sit in the slot of their thread)
#pragma unroll
        for (int e = 0; e < FLD; e++) {
            const int i4 = tid + e * NT;
            if (i4 < 16 * (F / 4)) { const int r = i4 / (F / 4), c4 = i4 - r * (F / 4); *reinterpret_cast<float4*>(fimg + r * LDF + 4 * c4) = fsrc[e]; }
        }
        *reinterpret_cast<float4*>(iimg + 4 * tid) = isrc;
    };
    // the epilogue of one tile: work item (chunk c, Gaussian nn) adds the waves' partial sums in a fixed order, the bias and the input,
    // applies exp / normalize / sigmoid and stores (a wave holds four chunks of all 16 Gaussians: wave 0 the four k <= 4 heads, the other
    // three the SH rows)
    auto epilogue = [&](int tile, int buf, int item, const float* iimg) {
        const int c = item >> 4, nn = item & 15;
        const int hd = c < 4 ? c : FDGS_HEAD_SHS;
        const long long g_raw = (long long)tile * 16 + nn;
        if (g_raw >= p.N) return;
        const size_t g = (size_t)g_raw;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((hmask >> hd) & 1u) {
#pragma unroll
            for (int ww = 0; ww < NW; ww++) {
                const float4 t = *reinterpret_cast<const float4*>(&pl[buf][ww][nn][4 * ((c ^ nn) & 15)]);
                o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
            }
            const float4 b = *reinterpret_cast<const float4*>(&b2l[4 * c]);
            o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
        }
        const float4 in = *reinterpret_cast<const float4*>(iimg + 4 * tid);      // (requested and parked by this very thread)
        const float v0 = in.x + o.x, v1 = in.y + o.y, v2 = in.z + o.z, v3 = in.w + o.w;
        if (c >= 4) {
            *reinterpret_cast<float4*>(d.out.shs + 48 * g + 4 * (c - 4)) = make_float4(v0, v1, v2, v3);
        } else if (c == FDGS_HEAD_POS) {
            d.out.xyz[3 * g] = v0; d.out.xyz[3 * g + 1] = v1; d.out.xyz[3 * g + 2] = v2;
        } else if (c == FDGS_HEAD_SCALE) {
            d.out.scales[3 * g] = p.activate ? __expf(v0) : v0;
            d.out.scales[3 * g + 1] = p.activate ? __expf(v1) : v1;
            d.out.scales[3 * g + 2] = p.activate ? __expf(v2) : v2;
        } else if (c == FDGS_HEAD_ROT) {
            float r0 = v0, r1 = v1, r2 = v2, r3 = v3;
            if (p.activate) {
                const float nrm = sqrtf(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
                const float inv = 1.0f / fmaxf(nrm, 1e-12f);  // F.normalize eps (scene/gaussian_model.py:44)
                r0 *= inv; r1 *= inv; r2 *= inv; r3 *= inv;
                if (d.out.rot_norm) d.out.rot_norm[g] = nrm;
            }
            reinterpret_cast<float4*>(d.out.rotations)[g] = make_float4(r0, r1, r2, r3);
        } else {
            d.out.opacity[g] = p.activate ? sigmoidf_(v0) : v0;
        }
    };
    // this wave's RT x 16 rows of relu(W0 feat + b0) of the tile whose features sit in `ft`, into the image `xt_out`;
    // B-lane (n, kk = q) of k-step (u, c) = feature 16 u + 4 kk + c of Gaussian n
    auto trunk = [&](const float* ft, float* xt_out, uint32_t* hm_out) {
        float4 fb[FU];
#pragma unroll
        for (int u = 0; u < FU; u++) fb[u] = *reinterpret_cast<const float4*>(ft + n * LDF + 16 * u + 4 * q);
#pragma unroll
        for (int t = 0; t < RT; t++) {
            const int row0 = 16 * (RT * w + t) + 4 * q;                            // D-lane (n, q) register r = row row0 + r
            const float4 b = *reinterpret_cast<const float4*>(&b0l[row0]);
            f32x4 acc = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int u = 0; u < FU; u++) {
                acc = mm16(w0r[t][u].x, fb[u].x, acc); acc = mm16(w0r[t][u].y, fb[u].y, acc);
                acc = mm16(w0r[t][u].z, fb[u].z, acc); acc = mm16(w0r[t][u].w, fb[u].w, acc);
            }
            *reinterpret_cast<float4*>(xt_out + n * LDW + row0) = make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
            if constexpr (SAVE) {
                // the backward's ReLU bits (D2's lane layout: 32-Gaussian tiles, lane (g32, h), word t bit r = feature T32 * rho(r, h) + t): feature
                // f has t = f % T32, rho = f / T32, h = (rho >> 2) & 1, r = (rho & 3) + 4 (rho >> 3); OR-ed into the tile's words
                constexpr int T32 = W / 32;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int f = row0 + rr, tt = f % T32, rho_ = f / T32, h = (rho_ >> 2) & 1, r = (rho_ & 3) + 4 * (rho_ >> 3);
                    atomicOr(&hm_out[(n * 2 + h) * 4 + tt], acc[rr] > 0.f ? 1u << r : 0u);
                }
            }
        }
    };

#ifdef FDGS_PROFILE_WS
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pt = __builtin_amdgcn_s_memtime();
#define WS_TICK(ph) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[ph] += t_ - pt; pt = t_; } while (0)
#else
#define WS_TICK(ph) do { } while (0)
#endif
    int it = 0, prev = -1;
    const int G = (int)gridDim.x;
    float4 fq[FLD], iq;
    if ((int)blockIdx.x < ntiles) {      // prologue: the first tile's image; the second tile's features and the first tile's inputs parked
        request(blockIdx.x, ntiles, fq, iq);
        park(fq, iq, fl[0], inl[1]);
        __syncthreads();
        trunk(fl[0], xl[0], hml[0]);
        request(blockIdx.x + G, blockIdx.x, fq, iq);
        park(fq, iq, fl[1], inl[0]);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += G, it ^= 1) {
        WS_TICK(0);
        // what the previous iteration requested: the features of tile + G into fl[it ^ 1], the inputs of THIS tile (its epilogue runs in the
        // next iteration) into inl[it] (the first iteration finds both parked by the prologue); then the next requests
        if (prev >= 0) park(fq, iq, fl[it ^ 1], inl[it]);
        request(tile + 2 * G, tile + G, fq, iq);
        WS_TICK(1);
        __syncthreads();      // this tile's image is complete -- and so are the partial sums of the previous tile and the features of the next
        WS_TICK(2);
        const float* xt = xl[it];
        const size_t tile_n0 = (size_t)tile * 16;
        // the tile's image as B operands: lane (n, kk = q) of k-step (j, c) = feature 16 j + 4 kk + c of Gaussian n
        float4 xr[KG];
#pragma unroll
        for (int j = 0; j < KG; j++) xr[j] = *reinterpret_cast<const float4*>(xt + n * LDW + 16 * j + 4 * q);
        if constexpr (SAVE) {
            // saved relu(hidden): whole rows, W / 64 float4 per thread; its ReLU bits: 32 lanes (Gaussian, half) write their four words and clear them
#pragma unroll
            for (int e = 0; e < W / 64; e++) {
                const int i4 = tid + e * NT, row = i4 / (W / 4), c4 = i4 - row * (W / 4);
                store_saved16(d.sv_rh + (tile_n0 + row) * W + 4 * c4, *reinterpret_cast<const float4*>(xt + row * LDW + 4 * c4));
            }
            if (d.sv_hmask && tid < 32) {
                const int nn = tid & 15, h = tid >> 4;
                uint4* src = reinterpret_cast<uint4*>(&hml[it][(nn * 2 + h) * 4]);
                reinterpret_cast<uint4*>(d.sv_hmask)[(size_t)(tile >> 1) * 64 + 32 * h + 16 * (tile & 1) + nn] = *src;
                *src = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        WS_TICK(3);
        // ---- the heads, software-pipelined: ONE wave per SIMD means nothing else runs while a head's tail (sum, ReLU, stores, W2 from LDS, the
        // short second-layer products, the lane-group sums, the partial sums to LDS) works through its dependent chain -- so the tail of head
        // hd - 1 is cut into three pieces that ride inside the first-layer product of head hd (64 independent MFMAs = 2 048 cycles), and the
        // next tile's trunk rides inside the first head's.  sched_barriers keep the pieces where they are put.
        f32x4 acc0[RT], acc1[RT];                 // the head in flight
        f32x4 y[RT];                              // relu(h1) of the head whose tail is under way
        float4 w2a[3][RT];                        // its W2 operands
        f32x4 o2[3];                              // its second-layer accumulators
        auto tail1 = [&](int hd) {                // sum + ReLU, the saved rows, the W2 operand requests
#pragma unroll
            for (int t = 0; t < RT; t++) {
                y[t] = acc0[t] + acc1[t];
#pragma unroll
                for (int r = 0; r < 4; r++) y[t][r] = fmaxf(y[t][r], 0.f);
            }
            if constexpr (SAVE) {      // lane (n, q): features 16 (RT w + t) + 4 q .. + 3 of Gaussian n
#pragma unroll
                for (int t = 0; t < RT; t++)
                    store_saved16(d.sv_h1 + ((size_t)d.head_slot[hd] * d.Npad + tile_n0 + n) * W + 16 * (RT * w + t) + 4 * q,
                                  make_float4(y[t][0], y[t][1], y[t][2], y[t][3]));
            }
            if (hd != FDGS_HEAD_SHS) {
                const int k = head_k(hd), row2 = (lane & 3) < k ? (lane & 3) : k - 1;
#pragma unroll
                for (int t = 0; t < RT; t++) w2a[0][t] = *reinterpret_cast<const float4*>(w2l + (head_row0(hd) + row2) * LDW + 16 * (RT * w + t) + 4 * q);
            } else {
#pragma unroll
                for (int ot = 0; ot < 3; ot++)
#pragma unroll
                    for (int t = 0; t < RT; t++)
                        w2a[ot][t] = *reinterpret_cast<const float4*>(w2l + (head_row0(hd) + 16 * ot + n) * LDW + 16 * (RT * w + t) + 4 * q);
            }
        };
        auto tail2 = [&](int hd) {                // this wave's share of the second layer: k-step (t, r) = feature 16 (RT w + t) + 4 q + r
            if (hd != FDGS_HEAD_SHS) {
                // k <= 4 rows on v_mfma_f32_4x4x1_16b: block b = lane / 4 holds four Gaussians of lane group q, A-lane 4 b + i = W2[i][feature]
                o2[0] = zero4();
#pragma unroll
                for (int t = 0; t < RT; t++) {
                    o2[0] = mfma4(w2a[0][t].x, y[t][0], o2[0]); o2[0] = mfma4(w2a[0][t].y, y[t][1], o2[0]);
                    o2[0] = mfma4(w2a[0][t].z, y[t][2], o2[0]); o2[0] = mfma4(w2a[0][t].w, y[t][3], o2[0]);
                }
            } else {
#pragma unroll
                for (int ot = 0; ot < 3; ot++) {  // three 16-row tiles; D-lane (n, q) register r = SH row 16 ot + 4 q + r
                    o2[ot] = zero4();
#pragma unroll
                    for (int t = 0; t < RT; t++) {
                        o2[ot] = mm16(w2a[ot][t].x, y[t][0], o2[ot]); o2[ot] = mm16(w2a[ot][t].y, y[t][1], o2[ot]);
                        o2[ot] = mm16(w2a[ot][t].z, y[t][2], o2[ot]); o2[ot] = mm16(w2a[ot][t].w, y[t][3], o2[ot]);
                    }
                }
            }
        };
        auto tail3 = [&](int hd) {                // the partial sums to LDS
            if (hd != FDGS_HEAD_SHS) {
                // lane 4 b + j register i = row i of Gaussian 4 (b % 4) + j = n, summed over this lane group's features: add the groups
                const int k = head_k(hd);
                float4 os;
                os.x = sum_lane_groups(o2[0][0]); os.y = k > 1 ? sum_lane_groups(o2[0][1]) : 0.f;
                os.z = k > 2 ? sum_lane_groups(o2[0][2]) : 0.f; os.w = k > 3 ? sum_lane_groups(o2[0][3]) : 0.f;
                if (q == 0) *reinterpret_cast<float4*>(&pl[it][w][n][4 * ((hd ^ n) & 15)]) = os;
            } else {
#pragma unroll
                for (int ot = 0; ot < 3; ot++)
                    *reinterpret_cast<float4*>(&pl[it][w][n][4 * (((4 + 4 * ot + q) ^ n) & 15)]) = make_float4(o2[ot][0], o2[ot][1], o2[ot][2], o2[ot][3]);
            }
        };
        bool trunk_done = false;
#pragma unroll
        for (int hd = 0; hd < FDGS_NUM_HEADS; hd++) {
            if (!ALLH && !((hmask >> hd) & 1u)) continue;
            f32x4 n0[RT], n1[RT];
#pragma unroll
            for (int t = 0; t < RT; t++) {
                const float4 b = *reinterpret_cast<const float4*>(&b1l[hd * W + 16 * (RT * w + t) + 4 * q]);
                n0[t] = f32x4{b.x, b.y, b.z, b.w}; n1[t] = zero4();
            }
#pragma unroll
            for (int j = 0; j < KG; j++) {
#pragma unroll
                for (int t = 0; t < RT; t++) { n0[t] = mm16(w1r[hd][t][j].x, xr[j].x, n0[t]); n1[t] = mm16(w1r[hd][t][j].y, xr[j].y, n1[t]); }
#pragma unroll
                for (int t = 0; t < RT; t++) { n0[t] = mm16(w1r[hd][t][j].z, xr[j].z, n0[t]); n1[t] = mm16(w1r[hd][t][j].w, xr[j].w, n1[t]); }
                // the riders of this product (with a runtime head mask the tails run behind their own products instead)
                if constexpr (ALLH) {
                    if (hd > 0) {
                        if (j == 0) tail1(hd - 1);
                        if (j == KG / 2 - 1) tail2(hd - 1);
                        if (j == KG - 2) tail3(hd - 1);
                    } else if (j == 1) {
                        // the trunk of the NEXT tile (its features were complete at the barrier above; nobody reads that image before the next barrier)
                        if (tile + G < ntiles) trunk(fl[it ^ 1], xl[it ^ 1], hml[SAVE ? (it ^ 1) : 0]);
                        trunk_done = true;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int t = 0; t < RT; t++) { acc0[t] = n0[t]; acc1[t] = n1[t]; }
            if constexpr (!ALLH) { tail1(hd); tail2(hd); tail3(hd); }
        }
        if constexpr (ALLH) { tail1(FDGS_NUM_HEADS - 1); tail2(FDGS_NUM_HEADS - 1); tail3(FDGS_NUM_HEADS - 1); }
        if (!trunk_done && tile + G < ntiles) trunk(fl[it ^ 1], xl[it ^ 1], hml[SAVE ? (it ^ 1) : 0]);
        WS_TICK(4);
        // ---- epilogue of the PREVIOUS tile (its partial sums and inputs were complete at the barrier above)
        if (prev >= 0) epilogue(prev, it ^ 1, (my_c << 4) | my_nn, inl[it ^ 1]);
        prev = tile;
        WS_TICK(5);
    }
    __syncthreads();      // the last tile's epilogue (its inputs were parked at the top of its iteration)
    if (prev >= 0) epilogue(prev, it ^ 1, (my_c << 4) | my_nn, inl[it ^ 1]);
#ifdef FDGS_PROFILE_WS
    if (d.prof && lane == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(&d.prof[i], pacc[i]);
        atomicAdd(&d.prof[8], 1ull);
    }
#endif
}

template <int WT, int FCH>
struct FwdWsLauncher {      // (the caller only selects this form when C*L % 16 == 0 and W is 64, or 128 with C*L <= 48)
    static void go(hipStream_t s, int blocks, const DeformDev& d) {
        if constexpr ((FCH % 2) == 0 && (WT == 2 || (WT == 4 && FCH <= 6))) {
            constexpr int RT = WT / 2;
            const bool save = d.sv_h1 != nullptr, allh = d.head_mask == 31u;
            if (save && allh) hipLaunchKernelGGL((deform_mlp_ws_kernel<RT, FCH / 2, true, true>), dim3(blocks), dim3(256), 0, s, d);
            else if (save) hipLaunchKernelGGL((deform_mlp_ws_kernel<RT, FCH / 2, true, false>), dim3(blocks), dim3(256), 0, s, d);
            else if (allh) hipLaunchKernelGGL((deform_mlp_ws_kernel<RT, FCH / 2, false, true>), dim3(blocks), dim3(256), 0, s, d);
            else hipLaunchKernelGGL((deform_mlp_ws_kernel<RT, FCH / 2, false, false>), dim3(blocks), dim3(256), 0, s, d);
        } else {
            hipLaunchKernelGGL((deform_fwd_kernel<WT, FCH>), dim3(blocks), dim3(256), 0, s, d);
        }
    }
};
