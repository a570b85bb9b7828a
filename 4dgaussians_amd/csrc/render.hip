// render.hip -- K6 (front-to-back alpha blending) and K7 (its backward) for gfx950.
//
// Forward: one 256-thread workgroup (4 wave64) per 16x16-pixel tile, one pixel per lane; per round the workgroup stages 256 list
// entries (3 x float4 per Gaussian) in LDS and the inner loop reads them with wave-uniform (broadcast) ds_read_b128.
// Backward (round 3): ONE wave per tile, four pixels per lane as two packed-f32 pairs (render_bwd_strip_kernel): each lane recomputes
// alpha back-to-front, forms the per-Gaussian sums of its pixels as moments over dy, the wave adds them with a DPP / permlane
// reduce-scatter and issues ONE 64-byte atomic line-op per (tile, Gaussian) into a packed [P,16] gradient array.  The 256-thread
// backward of rounds 1-2 (four waves meeting in LDS slots) stays selectable (tuning knob rbwd_ppl = 0; 2 = two waves per tile with two
// pixels per lane): the three forms are compared with each other in tests/test_gpu_raster.py.  (A strip form of the FORWARD measured
// slower and was dropped in round 3.)
// Tile rows are dealt to the XCDs in groups (unit_of_block: workgroup b runs on XCD b % 8), so that neighbouring tiles -- which list
// the same Gaussians -- share an XCD's 4 MiB L2 and every XCD owns rows from the whole height of the image.
// Replaces renderCUDA forward/backward of the un-vendored rasterizer (Appendix B.3/B.4).
#include "common.h"
#include "gs_math.h"

namespace fdgs {

// Workgroup b runs on XCD b % 8 (round-robin dispatch).  The tile rows are dealt to the XCDs in groups of `bh` rows: group G (rows
// G*bh .. G*bh+bh-1) belongs to XCD G % 8, and an XCD walks its groups top to bottom.  Neighbouring tiles (which list the same Gaussians)
// share an XCD's L2 inside a group, and every XCD owns rows from the whole height of the image: with ONE contiguous band per XCD (rounds
// 1-2) the bands that only see the rim of the scene ran empty while the XCDs of the central bands carried the frame.
// `parts` work units per tile (strip kernels: waves per tile), units of a tile adjacent.  Returns the unit index or -1.
__device__ __forceinline__ int unit_of_block(int b, int gx, int gy, int parts, int bh) {
    const int xcd = b & 7, idx = b >> 3;
    const int per_group = bh * gx * parts;
    const int g = idx / per_group, rem = idx - g * per_group;
    const int row = (g * 8 + xcd) * bh + rem / (gx * parts);
    if (row >= gy) return -1;
    return row * gx * parts + rem % (gx * parts);
}
__host__ __device__ __forceinline__ int unit_grid(int gx, int gy, int parts, int bh) {
    const int groups = (gy + bh - 1) / bh;
    return 8 * ((groups + 7) / 8) * bh * gx * parts;
}
// Heaviest-first dispatch (round 5).  A blending kernel is ONE wave (backward) or one workgroup (forward) per tile, the tiles' list
// lengths differ by 5x between the rim and the centre of the image, and with 5 440 tiles on 1 024 SIMDs x 4 wave slots nearly every
// tile is resident from the start: nothing balances the load, a SIMD that drew four centre tiles finishes long after one that drew
// four rim tiles (a dispatch simulation on the bench frame's per-tile work puts the backward at 0.62 of perfect balance in image order
// and at 0.82 heaviest-first; tools/tile_balance_sim.py).  `order` (tile_order_kernel) lists each XCD's tiles by descending work: block b
// still runs on XCD b % 8 and still only takes tiles of that XCD's row groups (their Gaussians stay in its L2), but the heavy ones are
// dispatched first and the light ones fill in behind them.  order == nullptr: image order.
__device__ __forceinline__ int unit_lookup(const uint32_t* __restrict__ order, int per_xcd, int b, int gx, int gy, int parts, int bh) {
    if (!order) return unit_of_block(b, gx, gy, parts, bh);
    const int xcd = b & 7, idx = b >> 3;
    const uint32_t t = order[xcd * per_xcd + idx / parts];
    return t == 0xFFFFFFFFu ? -1 : (int)t * parts + idx % parts;
}
// One 1024-thread workgroup per XCD: counting sort of the XCD's tiles by descending key (256 levels of the XCD's largest key; LPT needs
// no finer order), key = list length (mode 0, forward) or the forward's per-tile `todo` (mode 1, backward).  Padding slots of the map
// (rows below the image) and tiles with key 0 go last.
__global__ void __launch_bounds__(1024) tile_order_kernel(int mode, int gx, int gy, int bh, int per_xcd, const uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ todo, uint32_t* __restrict__ order) {
    __shared__ uint32_t hist[256], base[256], smax;
    const int x = blockIdx.x, t = threadIdx.x;
    if (t < 256) hist[t] = 0;
    if (t == 0) smax = 0;
    __syncthreads();
    constexpr int PER = TILE_ORDER_MAX / 1024;
    int tile[PER];
    uint32_t key[PER];
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int idx = t + 1024 * i;
        tile[i] = idx < per_xcd ? unit_of_block(idx * 8 + x, gx, gy, 1, bh) : -2;      // -1: padding slot of the map, -2: beyond the map
        key[i] = 0;
        if (tile[i] >= 0) key[i] = mode == 0 ? ranges[tile[i]].y - ranges[tile[i]].x : todo[tile[i]];
        m = key[i] > m ? key[i] : m;
    }
    if (m) atomicMax(&smax, m);
    __syncthreads();
    const float scale = 255.0f / (float)(smax > 0 ? smax : 1u);
    uint32_t bucket[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const uint32_t lvl = (uint32_t)((float)key[i] * scale);         // (monotonic in the key: all the order needs)
        bucket[i] = 255u - (lvl < 255u ? lvl : 255u);
        if (tile[i] == -1) bucket[i] = 255u;
        if (tile[i] != -2) atomicAdd(&hist[bucket[i]], 1u);
    }
    __syncthreads();
    if (t < 64) {       // exclusive scan of the 256 counts by one wave: four per lane
        const uint32_t c0 = hist[4 * t], c1 = hist[4 * t + 1], c2 = hist[4 * t + 2], c3 = hist[4 * t + 3];
        uint32_t inc = c0 + c1 + c2 + c3;
        const uint32_t tot = inc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o, 64); if (t >= o) inc += u; }
        const uint32_t ex = inc - tot;
        base[4 * t] = ex; base[4 * t + 1] = ex + c0; base[4 * t + 2] = ex + c0 + c1; base[4 * t + 3] = ex + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; i++)
        if (tile[i] != -2) order[(size_t)x * per_xcd + atomicAdd(&base[bucket[i]], 1u)] = tile[i] >= 0 ? (uint32_t)tile[i] : 0xFFFFFFFFu;
}

// The exponent of a blended Gaussian, written so that EVERY kernel that evaluates it rounds it the same way (the forward decides
// alpha >= 1/255 and T < 1e-4 from it, the backward has to take the same decisions): -0.5 * (dx*(dx*cxx) + dy*(dy*cyy)) - dy*(dx*cxy),
// products and the sum rounded separately, one fused multiply-add at the end.
__device__ __forceinline__ float blend_power_xx(float cxx, float dx) {
#pragma clang fp contract(off)
    return dx * (dx * cxx);
}
__device__ __forceinline__ float blend_power_xy(float cxy, float dx) {
#pragma clang fp contract(off)
    return dx * cxy;
}
__device__ __forceinline__ float blend_power(float u1, float cx, float cyy, float dy) {
#pragma clang fp contract(off)
    const float u2 = dy * (dy * cyy);
    const float c2 = dy * cx;
    const float q = u1 + u2;
    return __builtin_fmaf(q, -0.5f, -c2);
}
typedef float v2f_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f_ blend_power2(float u1, float cx, float cyy, v2f_ dy) {
#pragma clang fp contract(off)
    const v2f_ u2 = dy * (dy * cyy);
    const v2f_ c2 = dy * cx;
    const v2f_ q = u1 + u2;
    return __builtin_elementwise_fma(q, v2f_{-0.5f, -0.5f}, -c2);
}

struct RenderArgs {
    int W, H, gx, gy, bh;
    const uint32_t* order; int per_xcd;      // heaviest-first tile order (nullptr: image order)
    uint32_t* tile_todo;                     // out, per tile: max n_contrib = the entries the backward will walk
    const uint2* ranges; const uint32_t* pair_gid;
    const float4 *recA, *recB, *recC;
    const float* bg;
    float* final_T; uint32_t* n_contrib;
    float *out_color, *out_depth;
    float4* zfill; uint32_t zfill_n4;        // opt: a range zero-filled on the way (fdgs_raster_params::acc_zero: the backward's accumulator)
};

__global__ void __launch_bounds__(256) render_fwd_kernel(RenderArgs a) {
    if (a.zfill)     // (the kernel is bound by its arithmetic: ~one 16-byte store per thread rides along for free, and a fill launch is gone)
        for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < a.zfill_n4; k += gridDim.x * 256u) a.zfill[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int tile = unit_lookup(a.order, a.per_xcd, blockIdx.x, a.gx, a.gy, 1, a.bh);
    if (tile < 0) return;
    __shared__ float4 sA[256], sB[256], sC[256];
    __shared__ uint32_t sTodo[4];
    const int t = threadIdx.x;
    const int x = (tile % a.gx) * TILE + (t & 15), y = (tile / a.gx) * TILE + (t >> 4);
    const bool inside = x < a.W && y < a.H;
    const float pxf = (float)x, pyf = (float)y;
    const uint2 range = a.ranges[tile];
    int toDo = (int)(range.y - range.x);
    const int rounds = (toDo + 255) / 256;
    bool done = !inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t contributor = 0, last = 0;
    // staging pipeline (round 6; the backward has had it since round 3): list ids two rounds ahead, records one round ahead (the records'
    // addresses depend on the ids) -- the two dependent memory round trips of a round no longer sit between its barrier and its arithmetic
    const int n_list = (int)(range.y - range.x);
    auto gid_of = [&](int r) -> uint32_t {
        const int e = r * 256 + t;
        return a.pair_gid[range.x + (e < n_list ? e : n_list - 1)];
    };
    uint32_t g_cur = 0u, g_nxt = 0u;
    float4 rA, rB, rC;
    if (rounds > 0) {
        g_cur = gid_of(0);
        if (rounds > 1) g_nxt = gid_of(1);
        rA = a.recA[g_cur]; rB = a.recB[g_cur]; rC = a.recC[g_cur];
    }
    for (int r = 0; r < rounds; r++, toDo -= 256) {
        if (__syncthreads_count(done) == 256) break;
        sA[t] = rA; sB[t] = rB; sC[t] = rC;      // (entries behind the end of the list are copies of its last entry: never read, j < lim)
        __syncthreads();
        if (r + 1 < rounds) { g_cur = g_nxt; rA = a.recA[g_cur]; rB = a.recB[g_cur]; rC = a.recC[g_cur]; }
        if (r + 2 < rounds) g_nxt = gid_of(r + 2);
        const int lim = toDo < 256 ? toDo : 256;
        for (int j = 0; !done && j < lim; j++) {
            contributor++;
            const float4 A = sA[j];
            const float4 B = sB[j];
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float power = blend_power(blend_power_xx(A.z, dx), blend_power_xy(A.w, dx), B.x, dy);
            if (power > 0.0f) continue;
            const float alpha = fminf(FDGS_ALPHA_MAX, B.y * __expf(power));
            if (alpha < FDGS_ALPHA_MIN) continue;
            const float test_T = T * (1.0f - alpha);
            if (test_T < FDGS_T_STOP) { done = true; continue; }
            const float4 Cc = sC[j];
            const float w = alpha * T;
            C0 += Cc.x * w; C1 += Cc.y * w; C2 += Cc.z * w; Dp += B.z * w;
            T = test_T;
            last = contributor;
        }
    }
    if (inside) {
        const size_t pix = (size_t)y * a.W + x, hw = (size_t)a.H * a.W;
        a.final_T[pix] = T; a.n_contrib[pix] = last;
        a.out_color[pix] = C0 + T * a.bg[0];
        a.out_color[hw + pix] = C1 + T * a.bg[1];
        a.out_color[2 * hw + pix] = C2 + T * a.bg[2];
        a.out_depth[pix] = Dp;
    }
    {   // the tile's walk length for the backward (its work: the key of the backward's heaviest-first order)
        uint32_t m = inside ? last : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t u = __shfl_xor(m, o, 64); m = u > m ? u : m; }
        __syncthreads();            // (the staging loop above may have been left at different points: sTodo is only touched here)
        if ((t & 63) == 0) sTodo[t >> 6] = m;
        __syncthreads();
        if (t == 0) {
            uint32_t q = sTodo[0]; q = sTodo[1] > q ? sTodo[1] : q; q = sTodo[2] > q ? sTodo[2] : q; q = sTodo[3] > q ? sTodo[3] : q;
            a.tile_todo[tile] = q;
        }
    }
}

// ---- wave64 all-reduce (sum): DPP inside rows, permlane swaps across rows; every lane ends with the total ----
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_allsum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]  : xor 1
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]  : xor 2
    v += dpp_f<0x141>(v);  // row_half_mirror      : other quad of the 8
    v += dpp_f<0x140>(v);  // row_mirror           : other half of the 16
    {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v;
}

// sum over the 16 lanes of a DPP row (= one 16-pixel line of the tile); every lane of the row ends with the row total
__device__ __forceinline__ float row_allsum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]  : xor 1
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]  : xor 2
    v += dpp_f<0x141>(v);  // row_half_mirror      : other quad of the 8
    v += dpp_f<0x140>(v);  // row_mirror           : other half of the 16
    return v;
}

// Sums of ten values over the 64 lanes of a wave as a reduce-scatter.  xor-1 and xor-2 butterflies inside the quads halve the values
// per lane twice (lane bits 0,1 select which value a lane keeps); row_ror 4 / 8 add the four quads of a 16-lane row (they preserve
// lane & 3); v_permlane16_swap / v_permlane32_swap on PAIRS of values add the four rows while halving once more.  On return
//   lanes  0..15 hold sum(a[lane & 3]), lanes 16..31 sum(a[4 + (lane & 3)]), lanes 32..63 sum(a[8 + (lane & 1)]).
template <bool TEN>
__device__ __forceinline__ float wave_reduce_scatter10(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7,
                                                       float a8, float a9, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    // (the DPP moves stay unconditional: a select between two DPP results is compiled into divergent branches)
    auto rs1 = [&](float lo, float hi) { const float keep = b0 ? hi : lo, send = b0 ? lo : hi; return keep + dpp_f<0xB1>(send); };   // lane ^ 1
    auto rs2 = [&](float lo, float hi) { const float keep = b1 ? hi : lo, send = b1 ? lo : hi; return keep + dpp_f<0x4E>(send); };   // lane ^ 2
    const float c0 = rs1(a0, a1), c1 = rs1(a2, a3), c2 = rs1(a4, a5), c3 = rs1(a6, a7);
    const float c4 = TEN ? rs1(a8, a9) : a8 + dpp_f<0xB1>(a8);
    float d0 = rs2(c0, c1), d1 = rs2(c2, c3), d2 = c4 + dpp_f<0x4E>(c4);
    d0 += dpp_f<0x124>(d0); d1 += dpp_f<0x124>(d1); d2 += dpp_f<0x124>(d2);     // row_ror:4
    d0 += dpp_f<0x128>(d0); d1 += dpp_f<0x128>(d1); d2 += dpp_f<0x128>(d2);     // row_ror:8
    // rows: swap16(x, y) leaves x' = [x.r0, y.r0, x.r2, y.r2], y' = [x.r1, y.r1, x.r3, y.r3]
    auto q01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d0), __float_as_uint(d1), false, false);
    const float e01 = __uint_as_float(q01[0]) + __uint_as_float(q01[1]);      // rows 0,2: d0 over a row pair; rows 1,3: d1
    auto q2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d2), __float_as_uint(d2), false, false);
    const float e2 = __uint_as_float(q2[0]) + __uint_as_float(q2[1]);
    // swap32(x, y) leaves x' = [x.lo, y.lo], y' = [x.hi, y.hi]
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(e01), __float_as_uint(e2), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);                      // lower half: e01 totals, upper half: e2 totals
}

struct RenderBwdArgs {
    int W, H, gx, gy, bh;
    const uint32_t* order; int per_xcd;      // heaviest-first tile order (nullptr: image order)
    const uint2* ranges; const uint32_t* pair_gid;
    const float4 *recA, *recB, *recC;
    const float* bg;
    const float* final_T; const uint32_t* n_contrib;
    const float *dL_dcolor, *dL_ddepth;
    float* gacc;  // [P,16] per-Gaussian gradient line: 0,1 mean2D (pixel units) | 2,3,4 conic xx,xy(half),yy | 5 opacity |
                  // 6,7,8 rgb | 9 depth | 10..15 unused.  One 64-B line per Gaussian => one atomic line-op per (tile,Gaussian)
};

// DEPTH: a gradient w.r.t. the depth image was handed in (dL_ddepth != NULL).  The reference's training loss never uses it
// (utils/scene_utils.py:29 reads depth under no_grad), so the common instantiation drops the depth accumulators, their
// row reduction and the tenth value of the per-Gaussian sums.
// ROUND = list entries staged per round (256 or 128).  The four waves of the tile meet in per-WAVE LDS slots
// sAcc[entry][wave][NV]: a wave that hits an entry finishes the sum over its 64 pixels in registers (DPP inside the 16-lane rows, then
// permlane16/32 swaps across the rows on the ONE value each lane carries) and lanes 0..NV-1 store it with a plain ds_write; the flush
// adds the four slots.  (Until round 3 the rows met in LDS float atomics on one shared slot -- ds_add_f32 sustains 0.33 lanes/clk/CU,
// profiles/r01_lds_atomic_microbench.txt, and rocprofv3 showed the waves waiting 39 % of their cycles.)
template <bool DEPTH, int ROUND>
__global__ void __launch_bounds__(256) render_bwd_kernel(RenderBwdArgs a) {
    const int tile = unit_lookup(a.order, a.per_xcd, blockIdx.x, a.gx, a.gy, 1, a.bh);
    if (tile < 0) return;
    constexpr int NV = DEPTH ? 10 : 9;
    __shared__ float4 sA[ROUND], sB[ROUND], sC[ROUND];
    __shared__ uint32_t sGid[ROUND];
    __shared__ float sAcc[ROUND * 4 * NV];
    __shared__ uint32_t sMax[4];
    const int t = threadIdx.x, lane = t & 63;
    const int x = (tile % a.gx) * TILE + (t & 15), y = (tile / a.gx) * TILE + (t >> 4);
    const bool inside = x < a.W && y < a.H;
    const float pxf = (float)x, pyf = (float)y;
    const size_t pix = (size_t)y * a.W + x, hw = (size_t)a.H * a.W;
    const uint2 range = a.ranges[tile];
    const float T_final = inside ? a.final_T[pix] : 0.f;
    const uint32_t last_contributor = inside ? a.n_contrib[pix] : 0u;
    // the tile only ever needs the first max(n_contrib) entries of its list -- and each WAVE (a 16x4 pixel strip) only the first
    // max over its own 64 pixels: walking back to front, the entries behind that are skipped without being evaluated
    int wave_todo;
    {
        uint32_t m = last_contributor;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { uint32_t u = __shfl_xor(m, o, 64); m = u > m ? u : m; }
        if (lane == 0) sMax[t >> 6] = m;
        wave_todo = (int)m;
    }
    __syncthreads();
    uint32_t mx = sMax[0]; mx = sMax[1] > mx ? sMax[1] : mx; mx = sMax[2] > mx ? sMax[2] : mx; mx = sMax[3] > mx ? sMax[3] : mx;
    const int toDo = (int)mx;
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, ddep = 0.f;
    if (inside) {
        dp0 = a.dL_dcolor[pix]; dp1 = a.dL_dcolor[hw + pix]; dp2 = a.dL_dcolor[2 * hw + pix];
        if (DEPTH) ddep = a.dL_ddepth[pix];
    }
    const float bgdot = a.bg[0] * dp0 + a.bg[1] * dp1 + a.bg[2] * dp2;
    float T = T_final, acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accd = 0.f;
    float lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, ld = 0.f, last_alpha = 0.f;
    const int rounds = (toDo + ROUND - 1) / ROUND;
    const int wv_ = t >> 6;
    for (int r = 0; r < rounds; r++) {
        const int e = toDo - 1 - (r * ROUND + t);  // list entry (0-based from the front) staged by this thread
        if (t < ROUND && e >= 0) {
            const uint32_t gid = a.pair_gid[range.x + e];
            sGid[t] = gid; sA[t] = a.recA[gid]; sB[t] = a.recB[gid]; sC[t] = a.recC[gid];
        }
        for (int k = t; k < ROUND * 4 * NV; k += 256) sAcc[k] = 0.f;      // (an entry a wave does not hit keeps a zero slot)
        __syncthreads();
        const int rem = toDo - r * ROUND;
        const int lim = rem < ROUND ? rem : ROUND;
        int j0 = toDo - wave_todo - r * ROUND;        // first staged entry of this round that one of the wave's pixels blended
        j0 = j0 < 0 ? 0 : j0;
        for (int j = j0; j < lim; j++) {
            const uint32_t ej = (uint32_t)(toDo - 1 - (r * ROUND + j));
            const float4 A = sA[j];
            const float4 B = sB[j];
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float power = blend_power(blend_power_xx(A.z, dx), blend_power_xy(A.w, dx), B.x, dy);
            const float G = __expf(power);
            const float alpha = fminf(FDGS_ALPHA_MAX, B.y * G);
            const bool valid = (ej < last_contributor) && !(power > 0.0f) && !(alpha < FDGS_ALPHA_MIN);
            if (!__any(valid)) continue;
            float g_mx = 0.f, g_my = 0.f, g_cxx = 0.f, g_cxy = 0.f, g_cyy = 0.f, g_op = 0.f, g_c0 = 0.f, g_c1 = 0.f, g_c2 = 0.f,
                  g_d = 0.f;
            if (valid) {
                const float4 Cc = sC[j];
                // 1/(1-alpha) once (v_rcp_f32 + one Newton step, <= 1 ulp) for both divisions of the reference formula
                const float om = 1.0f - alpha;
                float inv = __builtin_amdgcn_rcpf(om);
                inv = fmaf(fmaf(-om, inv, 1.0f), inv, inv);
                T = T * inv;
                const float w = alpha * T;
                float dL_dalpha;
                acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0; lc0 = Cc.x;
                acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1; lc1 = Cc.y;
                acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2; lc2 = Cc.z;
                dL_dalpha = (Cc.x - acc0) * dp0 + (Cc.y - acc1) * dp1 + (Cc.z - acc2) * dp2;
                if (DEPTH) {
                    accd = last_alpha * ld + (1.f - last_alpha) * accd; ld = B.z;
                    dL_dalpha += (B.z - accd) * ddep;
                    g_d = w * ddep;
                }
                g_c0 = w * dp0; g_c1 = w * dp1; g_c2 = w * dp2;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final * inv) * bgdot;
                const float dL_dG = B.y * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * A.z - gdy * A.w;
                const float dG_ddely = -gdy * B.x - gdx * A.w;
                g_mx = dL_dG * dG_ddelx; g_my = dL_dG * dG_ddely;
                g_cxx = -0.5f * gdx * dx * dL_dG; g_cxy = -0.5f * gdx * dy * dL_dG; g_cyy = -0.5f * gdy * dy * dL_dG;
                g_op = G * dL_dalpha;
            }
            // per-Gaussian sums over the wave's 64 pixels: DPP butterflies inside each 16-lane row (4 instructions per value); lane
            // `sub` of every row then picks value `sub` and the four rows are added with two permlane swaps on that ONE value
            g_mx = row_allsum(g_mx); g_my = row_allsum(g_my);
            g_cxx = row_allsum(g_cxx); g_cxy = row_allsum(g_cxy); g_cyy = row_allsum(g_cyy);
            g_op = row_allsum(g_op);
            g_c0 = row_allsum(g_c0); g_c1 = row_allsum(g_c1); g_c2 = row_allsum(g_c2);
            if (DEPTH) g_d = row_allsum(g_d);
            const int sub = lane & 15;
            float v = g_mx;
            v = sub == 1 ? g_my : v; v = sub == 2 ? g_cxx : v; v = sub == 3 ? g_cxy : v; v = sub == 4 ? g_cyy : v;
            v = sub == 5 ? g_op : v; v = sub == 6 ? g_c0 : v; v = sub == 7 ? g_c1 : v; v = sub == 8 ? g_c2 : v;
            if (DEPTH) v = sub == 9 ? g_d : v;
            {
                auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                v = __uint_as_float(q[0]) + __uint_as_float(q[1]);
            }
            {
                auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                v = __uint_as_float(q[0]) + __uint_as_float(q[1]);
            }
            if (lane < NV) sAcc[(j * 4 + wv_) * NV + lane] = v;      // this wave's slot of entry j: plain store, no other writer
        }
        __syncthreads();
        // flush: 16 lanes own the 16-float gradient line of one staged Gaussian, so every atomic instruction covers four
        // whole 64-byte lines (measured: scattered float atomics cost ~1 line-op each at ~20 G line-ops/s on MI355X)
        {
            const int sub = lane & 15;
#pragma unroll 4
            for (int i = 0; i < ROUND / 16; i++) {
                const int jj = wv_ * (ROUND / 4) + i * 4 + (lane >> 4);
                if (jj < lim) {
                    float v = 0.f;
                    if (sub < NV) {
                        const float* sl = &sAcc[jj * 4 * NV + sub];
                        v = (sl[0] + sl[NV]) + (sl[2 * NV] + sl[3 * NV]);
                    }
                    if (v != 0.f) atomicAdd(&a.gacc[(size_t)sGid[jj] * 16 + sub], v);
                }
            }
        }
        __syncthreads();
    }
}

// ---- K7, strip form (round 3): ONE wave per workgroup, PPL pixels per lane ------------------------------------------------------
// The 256-thread form above is bound by VALU issue (a wave64 instruction holds its SIMD for four cycles): ~125 instructions per
// (wave, entry), of which ~50 are the cross-lane reduction of the nine per-Gaussian sums and ~15 the per-entry overhead -- paid once
// per 64 pixels.  Here a wave owns a 16 x (4*PPL) pixel strip of the tile, lane = (x, y0) with PPL pixels y0, y0+4, ... below each other:
//   * the lane's pixels share dx, so the per-Gaussian sums are first formed PER LANE as moments over dy (V0 = sum v, V1 = sum v*dy,
//     V2 = sum v*dy^2 with v = G * dL_dalpha) and turned into the six geometric sums once per lane, not once per pixel;
//   * the cross-lane reduction and the atomic line-op are paid once per 64*PPL pixels;
//   * the blending recurrences run branch-free (an invalid pixel blends alpha = 0, which leaves T and the running colour exactly
//     unchanged), in the un-lagged form acc' = alpha*c + (1-alpha)*acc (the same expression the lagged reference form evaluates one
//     entry later);
//   * a wave is its own workgroup: it stages its own 64 entries per round (records prefetched one round ahead, list ids two), never
//     waits for another wave, and sends its sums straight to the gradient line (lanes 0..NV-1, one atomic instruction per entry).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat2(float x) { return v2f{x, x}; }

// PPL = 2 * NP pixels per lane, held as NP float2 pairs so that the per-pixel arithmetic issues as v_pk_{mul,add,fma}_f32
template <bool DEPTH, int NP>
__global__ void __launch_bounds__(64) render_bwd_strip_kernel(RenderBwdArgs a) {
    constexpr int PPL = 2 * NP;
    constexpr int PARTS = 4 / PPL;
    const int unit = unit_lookup(a.order, a.per_xcd, blockIdx.x, a.gx, a.gy, PARTS, a.bh);
    if (unit < 0) return;
    const int tile = unit / PARTS, part = unit - tile * PARTS;
    __shared__ float4 sA[64], sB[64], sC[64];
    __shared__ uint32_t sGid[64];
    const int lane = threadIdx.x;
    const int x = (tile % a.gx) * TILE + (lane & 15);
    const int y0 = (tile / a.gx) * TILE + part * (4 * PPL) + (lane >> 4);
    const float pxf = (float)x;
    const size_t hw = (size_t)a.H * a.W;
    // which of the per-Gaussian sums this lane holds after wave_reduce_scatter10, and whether it is the lane that sends it
    const int slot = lane < 16 ? (lane & 3) : lane < 32 ? 4 + (lane & 3) : 8 + (lane & 1);
    const bool slot_on = (lane & 15) < 4 && (lane < 32 || (lane < 48 && (lane & 15) < (DEPTH ? 2 : 1)));
    const uint2 range = a.ranges[tile];
    v2f py[NP], T[NP], tfb[NP], dp0[NP], dp1[NP], dp2[NP], ddep[NP], acc0[NP], acc1[NP], acc2[NP], accd[NP];
    uint32_t lastc[PPL];
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < PPL; i++) {
        const int y = y0 + 4 * i;
        const bool inside = x < a.W && y < a.H;
        const size_t pix = (size_t)y * a.W + x;
        const int p = i >> 1, c = i & 1;
        py[p][c] = (float)y;
        T[p][c] = inside ? a.final_T[pix] : 0.f;
        lastc[i] = inside ? a.n_contrib[pix] : 0u;
        dp0[p][c] = inside ? a.dL_dcolor[pix] : 0.f;
        dp1[p][c] = inside ? a.dL_dcolor[hw + pix] : 0.f;
        dp2[p][c] = inside ? a.dL_dcolor[2 * hw + pix] : 0.f;
        ddep[p][c] = (DEPTH && inside) ? a.dL_ddepth[pix] : 0.f;
        m = lastc[i] > m ? lastc[i] : m;
    }
#pragma unroll
    for (int p = 0; p < NP; p++) {
        tfb[p] = -T[p] * (a.bg[0] * dp0[p] + a.bg[1] * dp1[p] + a.bg[2] * dp2[p]);
        acc0[p] = acc1[p] = acc2[p] = accd[p] = splat2(0.f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t u = __shfl_xor(m, o, 64); m = u > m ? u : m; }
    const int todo = (int)m;             // the strip only ever needs the first max(n_contrib) entries of the tile's list
    if (todo == 0) return;
    const int rounds = (todo + 63) >> 6;
    // staging pipeline: list ids two rounds ahead, records one round ahead (the records' addresses depend on the ids)
    auto gid_of = [&](int r) -> uint32_t {
        const int e = todo - 1 - (r * 64 + lane);
        return a.pair_gid[range.x + (e >= 0 ? e : 0)];
    };
    // (staging the records with global_load_lds_dwordx4 into a second LDS buffer frees their 12 registers -- 90 VGPRs, five waves per
    // SIMD -- and measured SLOWER, 0.341 vs 0.313 ms: the round-start vmcnt(0) also waits for the round's 64 atomics, and with a
    // counted wait hipcc still puts a vmcnt(0) in front of every LDS read that may alias an outstanding DMA -- one per list entry;
    // profiles/r03j_bench_cfg4_rbwd_glds.json)
    uint32_t g_cur = gid_of(0), g_nxt = rounds > 1 ? gid_of(1) : 0u;
    float4 rA = a.recA[g_cur], rB = a.recB[g_cur], rC = a.recC[g_cur];
    for (int r = 0; r < rounds; r++) {
        __syncthreads();                 // (one wave: orders the previous round's LDS reads before these writes)
        sGid[lane] = g_cur; sA[lane] = rA; sB[lane] = rB; sC[lane] = rC;
        __syncthreads();
        if (r + 1 < rounds) { g_cur = g_nxt; rA = a.recA[g_cur]; rB = a.recB[g_cur]; rC = a.recC[g_cur]; }
        if (r + 2 < rounds) g_nxt = gid_of(r + 2);
        const int rem = todo - r * 64;
        const int lim = rem < 64 ? rem : 64;
        for (int j = 0; j < lim; j++) {
            const uint32_t ej = (uint32_t)(todo - 1 - (r * 64 + j));
            const float4 A = sA[j];
            const float4 B = sB[j];
            const float dx = A.x - pxf;
            v2f dy[NP], G[NP], alpha[NP];
            bool valid[PPL], any = false;
            {
                const float u1 = blend_power_xx(A.z, dx);
                const float cx = blend_power_xy(A.w, dx);
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    dy[p] = splat2(A.y) - py[p];
                    const v2f power = blend_power2(u1, cx, B.x, dy[p]);
                    G[p][0] = __expf(fminf(power[0], 0.0f));      // (power > 0 is invalid below; the clamp only keeps G finite)
                    G[p][1] = __expf(fminf(power[1], 0.0f));
                    alpha[p] = B.y * G[p];
                    alpha[p][0] = fminf(FDGS_ALPHA_MAX, alpha[p][0]);
                    alpha[p][1] = fminf(FDGS_ALPHA_MAX, alpha[p][1]);
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        valid[2 * p + c] = (ej < lastc[2 * p + c]) && !(power[c] > 0.0f) && !(alpha[p][c] < FDGS_ALPHA_MIN);
                        any = any || valid[2 * p + c];
                    }
                }
            }
            if (!__any(any)) continue;
            const float4 Cc = sC[j];
            v2f V0 = splat2(0.f), V1 = V0, V2 = V0, gc0 = V0, gc1 = V0, gc2 = V0, gd = V0;
#pragma unroll
            for (int p = 0; p < NP; p++) {
                v2f al;
                al[0] = valid[2 * p] ? alpha[p][0] : 0.f;
                al[1] = valid[2 * p + 1] ? alpha[p][1] : 0.f;
                // 1/(1-alpha) once (v_rcp_f32 + one Newton step, <= 1 ulp) for both divisions of the reference formula
                const v2f om = splat2(1.0f) - al;
                v2f inv;
                inv[0] = __builtin_amdgcn_rcpf(om[0]); inv[1] = __builtin_amdgcn_rcpf(om[1]);
                inv = fma2(fma2(-om, inv, splat2(1.0f)), inv, inv);
                T[p] = T[p] * inv;
                const v2f w = al * T[p];
                v2f dLda = fma2(splat2(Cc.z) - acc2[p], dp2[p], fma2(splat2(Cc.y) - acc1[p], dp1[p], (splat2(Cc.x) - acc0[p]) * dp0[p]));
                if (DEPTH) {
                    dLda = fma2(splat2(B.z) - accd[p], ddep[p], dLda);
                    accd[p] = fma2(al, splat2(B.z), om * accd[p]);
                    gd = fma2(w, ddep[p], gd);
                }
                acc0[p] = fma2(al, splat2(Cc.x), om * acc0[p]);
                acc1[p] = fma2(al, splat2(Cc.y), om * acc1[p]);
                acc2[p] = fma2(al, splat2(Cc.z), om * acc2[p]);
                gc0 = fma2(w, dp0[p], gc0); gc1 = fma2(w, dp1[p], gc1); gc2 = fma2(w, dp2[p], gc2);
                dLda = fma2(dLda, T[p], tfb[p] * inv);
                v2f v = G[p] * dLda;
                v[0] = valid[2 * p] ? v[0] : 0.f;
                v[1] = valid[2 * p + 1] ? v[1] : 0.f;
                const v2f vy = v * dy[p];
                V0 += v; V1 += vy; V2 = fma2(vy, dy[p], V2);
            }
            // the lane's six geometric sums from its moments over dy (dL_dG * G = opacity * v)
            const float u0 = B.y * (V0[0] + V0[1]), u1 = B.y * (V1[0] + V1[1]), u2 = B.y * (V2[0] + V2[1]);
            const float ux = u0 * dx;
            float g_mx = -(A.z * ux + A.w * u1), g_my = -(B.x * u1 + A.w * ux);
            float g_cxx = -0.5f * ux * dx, g_cxy = -0.5f * u1 * dx, g_cyy = -0.5f * u2;
            float g_op = V0[0] + V0[1];
            float g_c0 = gc0[0] + gc0[1], g_c1 = gc1[0] + gc1[1], g_c2 = gc2[0] + gc2[1], g_d = gd[0] + gd[1];
            // nine (ten) sums over the wave's 64 lanes as a reduce-scatter: each butterfly step halves the number of values a lane
            // carries (36 instructions instead of 50 for nine all-reduces + select); lane `slot_lane` ends with value `slot`
            const float v = wave_reduce_scatter10<DEPTH>(g_mx, g_my, g_cxx, g_cxy, g_cyy, g_op, g_c0, g_c1, g_c2, g_d, lane);
            if (slot_on && v != 0.f) atomicAdd(&a.gacc[(size_t)sGid[j] * 16 + slot], v);
        }
    }
}

int validate_raster_params(const fdgs_raster_params* p);

// tile rows per XCD group (unit_of_block): common.h XCD_ROWS (0 was one contiguous band per XCD)
static int tile_rows_per_xcd_group(int) { return XCD_ROWS; }
// launches tile_order_kernel when the knob is on and an XCD's tiles fit its LDS sort; returns the order to hand to the kernel (or nullptr)
static const uint32_t* make_tile_order(hipStream_t stream, int mode, const ImgLayout& il, const void* img, void* img_w) {
    if (!g_tune.tile_order || il.per_xcd > TILE_ORDER_MAX || il.per_xcd < 1) return nullptr;
    uint32_t* order = at<uint32_t>(img_w, mode == 0 ? il.order_f : il.order_b);
    FDGS_TIMED("tile_order", stream);
    hipLaunchKernelGGL(tile_order_kernel, dim3(8), dim3(1024), 0, stream, mode, il.gx, il.gy, XCD_ROWS, il.per_xcd, at<uint2>(img, il.ranges),
                       at<uint32_t>(img, il.todo), order);
    return order;
}

}  // namespace fdgs

int fdgs_launch_preprocess_bwd(hipStream_t stream, const fdgs_raster_params* p, const void* geom, const fdgs_raster_grads* g);

using namespace fdgs;

extern "C" int fdgs_render_fwd(void* stream_, const fdgs_raster_params* p, const void* geom, const void* binning, void* img,
                               uint32_t R, float* out_color, float* out_depth) {
    int rc = validate_raster_params(p);
    if (rc) return rc;
    FDGS_REQUIRE(geom && img && out_color && out_depth && (binning || R == 0), "NULL buffer");
    hipStream_t stream = (hipStream_t)stream_;
    GeomLayout gl = geom_layout(p->P);
    BinLayout bl = bin_layout(R);
    ImgLayout il = img_layout(p->W, p->H);
    RenderArgs a{};
    a.W = p->W; a.H = p->H; a.gx = il.gx; a.gy = il.gy; a.bh = tile_rows_per_xcd_group(il.gy);
    a.ranges = at<uint2>(img, il.ranges); a.pair_gid = binning ? at<uint32_t>(binning, bl.gid0) : nullptr;
    a.recA = at<float4>(geom, gl.recA); a.recB = at<float4>(geom, gl.recB); a.recC = at<float4>(geom, gl.recC);
    a.bg = p->bg; a.final_T = at<float>(img, il.final_T); a.n_contrib = at<uint32_t>(img, il.n_contrib);
    a.out_color = out_color; a.out_depth = out_depth;
    a.tile_todo = at<uint32_t>(img, il.todo); a.per_xcd = il.per_xcd;
    a.zfill = reinterpret_cast<float4*>(p->acc_zero); a.zfill_n4 = (uint32_t)p->P * 4u;
    a.order = R > 0 ? make_tile_order(stream, 0, il, img, img) : nullptr;
    {
        FDGS_TIMED("render_fwd", stream);
        hipLaunchKernelGGL(render_fwd_kernel, dim3(unit_grid(a.gx, a.gy, 1, a.bh)), dim3(256), 0, stream, a);
    }
    FDGS_LAUNCH_CHECK("render_fwd", p->debug, stream);
    return FDGS_OK;
}

extern "C" int fdgs_raster_bwd(void* stream_, const fdgs_raster_params* p, const void* geom, const void* binning,
                               const void* img, uint32_t R, const fdgs_raster_grads* g) {
    int rc = validate_raster_params(p);
    if (rc) return rc;
    FDGS_REQUIRE(g && geom && img && (binning || R == 0), "NULL buffer");
    if (g->deform_epilogue) {     // the per-Gaussian results go to the deformation backward's buffers instead (include/fdgs.h)
        FDGS_REQUIRE(g->dL_dcolor && g->dL_dmeans2D && g->scratch_acc, "required gradient buffer is NULL");
    } else {
        FDGS_REQUIRE(g->dL_dcolor && g->dL_dmeans2D && g->dL_dmeans3D && g->dL_dopacity && g->dL_dcolors && g->dL_dcov3D &&
                         g->scratch_acc, "required gradient buffer is NULL");
        FDGS_REQUIRE(!p->shs || g->dL_dsh, "dL_dsh required when shs is given");
        FDGS_REQUIRE(p->cov3D_precomp || (g->dL_dscales && g->dL_drotations), "dL_dscales/dL_drotations required");
    }
    hipStream_t stream = (hipStream_t)stream_;
    const size_t P = (size_t)p->P;
    if (P == 0) return FDGS_OK;
    // the per-Gaussian outputs are written for every Gaussian by preprocess_bwd; only the atomic accumulator needs zeros
    if (!g->scratch_acc_zeroed) FDGS_HIP_CHECK(hipMemsetAsync(g->scratch_acc, 0, P * 64, stream));      // (1: the forward filled it, fdgs_raster_params::acc_zero)
    if (p->cov3D_precomp) {   // scale/rotation gradients do not exist on this path: hand back zeros if buffers were given
        if (g->dL_dscales) FDGS_HIP_CHECK(hipMemsetAsync(g->dL_dscales, 0, P * 12, stream));
        if (g->dL_drotations) FDGS_HIP_CHECK(hipMemsetAsync(g->dL_drotations, 0, P * 16, stream));
    }
    GeomLayout gl = geom_layout(p->P);
    BinLayout bl = bin_layout(R);
    ImgLayout il = img_layout(p->W, p->H);
    if (R > 0) {
        RenderBwdArgs a{};
        a.W = p->W; a.H = p->H; a.gx = il.gx; a.gy = il.gy; a.bh = tile_rows_per_xcd_group(il.gy);
        a.ranges = at<uint2>(img, il.ranges); a.pair_gid = at<uint32_t>(binning, bl.gid0);
        a.recA = at<float4>(geom, gl.recA); a.recB = at<float4>(geom, gl.recB); a.recC = at<float4>(geom, gl.recC);
        a.bg = p->bg; a.final_T = at<float>(img, il.final_T); a.n_contrib = at<uint32_t>(img, il.n_contrib);
        a.dL_dcolor = g->dL_dcolor; a.dL_ddepth = g->dL_ddepth;
        a.gacc = g->scratch_acc;
        // (the order lists live in `img`, the forward's state, which this call otherwise only reads: the slice written here is scratch of the
        // backward that belongs to this frame -- one backward per forward state at a time, as for every other buffer of the state)
        a.per_xcd = il.per_xcd;
        a.order = make_tile_order(stream, 1, il, img, const_cast<void*>(img));
        {
            FDGS_TIMED("render_bwd", stream);
            // (256-thread form: 128 entries staged per round keeps six workgroups per CU at 25 KB of LDS each)
            const dim3 grid(unit_grid(a.gx, a.gy, 1, a.bh));
            // pixels per lane of the strip form (1 wave per workgroup); 0 = the 256-thread form
            // -1 (default): by image size -- one wave per tile leaves the chip underfilled when the image has fewer tiles than the chip has wave
            // slots (1 024 SIMDs x 4): two waves per tile then (config 3, 2 040 tiles: blending backward 0.274 -> 0.213 ms; config 2, 2 500
            // tiles: 0.197 -> 0.187; at 5 440 tiles four pixels per lane win by 10 %, profiles/r05_rbwd_ppl_small_images_ab.txt)
            const int ppl = g_tune.rbwd_ppl >= 0 ? g_tune.rbwd_ppl : (a.gx * a.gy <= 4096 ? 2 : 4);
            if (ppl == 2 || ppl == 4) {
                const dim3 sgrid(unit_grid(a.gx, a.gy, 4 / ppl, a.bh));
#define FDGS_STRIP(D_, P_) hipLaunchKernelGGL((render_bwd_strip_kernel<D_, P_>), sgrid, dim3(64), 0, stream, a)
                if (g->dL_ddepth) { if (ppl == 2) FDGS_STRIP(true, 1); else FDGS_STRIP(true, 2); }
                else if (ppl == 2) FDGS_STRIP(false, 1);
                else FDGS_STRIP(false, 2);
#undef FDGS_STRIP
            } else if (g->dL_ddepth) {
                hipLaunchKernelGGL((render_bwd_kernel<true, 128>), grid, dim3(256), 0, stream, a);
            } else {
                hipLaunchKernelGGL((render_bwd_kernel<false, 128>), grid, dim3(256), 0, stream, a);
            }
        }
        FDGS_LAUNCH_CHECK("render_bwd", p->debug, stream);
    }
    return fdgs_launch_preprocess_bwd(stream, p, geom, g);
}
