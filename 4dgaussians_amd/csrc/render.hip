// render.hip -- K6 (front-to-back alpha blending) and K7 (its backward) for gfx950.
//
// One 256-thread workgroup (4 wave64) per 16x16-pixel tile, one pixel per lane; a wave covers a 16x4 pixel strip.
// Tiles are handed to XCDs in contiguous bands (workgroup b lands on XCD b%8 -> band b%8) so that the per-Gaussian
// records a band keeps re-reading stay in that XCD's 4 MiB L2.
// Per round the workgroup stages 256 list entries (3 x float4 per Gaussian) in LDS; the inner loop reads them with
// wave-uniform (broadcast) ds_read_b128.
// Backward: each lane recomputes alpha back-to-front; the per-Gaussian sums over the 64 pixels of a wave are formed
// with DPP butterflies inside the 16-lane rows plus v_permlane16_swap / v_permlane32_swap across rows (no LDS
// traffic), skipped entirely when no lane of the wave is hit; the four waves meet in a 10-float LDS slot per staged
// Gaussian and the tile issues ONE 64-byte atomic line-op per (tile, Gaussian) into a packed [P,16] gradient array.
// Replaces renderCUDA forward/backward of the un-vendored rasterizer (Appendix B.3/B.4).
#include "common.h"
#include "gs_math.h"

namespace fdgs {

__device__ __forceinline__ int tile_of_block(int b, int ntiles) {
    const int chunk = (ntiles + 7) >> 3;
    const int tile = (b & 7) * chunk + (b >> 3);
    return ((b >> 3) < chunk && tile < ntiles) ? tile : -1;
}

struct RenderArgs {
    int W, H, gx, gy;
    const uint2* ranges; const uint32_t* pair_gid;
    const float4 *recA, *recB, *recC;
    const float* bg;
    float* final_T; uint32_t* n_contrib;
    float *out_color, *out_depth;
};

__global__ void __launch_bounds__(256) render_fwd_kernel(RenderArgs a) {
    const int tile = tile_of_block(blockIdx.x, a.gx * a.gy);
    if (tile < 0) return;
    __shared__ float4 sA[256], sB[256], sC[256];
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
    for (int r = 0; r < rounds; r++, toDo -= 256) {
        if (__syncthreads_count(done) == 256) break;
        const int e = r * 256 + t;
        if (range.x + e < range.y) {
            const uint32_t gid = a.pair_gid[range.x + e];
            sA[t] = a.recA[gid]; sB[t] = a.recB[gid]; sC[t] = a.recC[gid];
        }
        __syncthreads();
        const int lim = toDo < 256 ? toDo : 256;
        for (int j = 0; !done && j < lim; j++) {
            contributor++;
            const float4 A = sA[j];
            const float4 B = sB[j];
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
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

struct RenderBwdArgs {
    int W, H, gx, gy;
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
    const int tile = tile_of_block(blockIdx.x, a.gx * a.gy);
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
            const float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
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

int validate_raster_params(const fdgs_raster_params* p);

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
    a.W = p->W; a.H = p->H; a.gx = il.gx; a.gy = il.gy;
    a.ranges = at<uint2>(img, il.ranges); a.pair_gid = binning ? at<uint32_t>(binning, bl.gid0) : nullptr;
    a.recA = at<float4>(geom, gl.recA); a.recB = at<float4>(geom, gl.recB); a.recC = at<float4>(geom, gl.recC);
    a.bg = p->bg; a.final_T = at<float>(img, il.final_T); a.n_contrib = at<uint32_t>(img, il.n_contrib);
    a.out_color = out_color; a.out_depth = out_depth;
    const int ntiles = il.gx * il.gy;
    { FDGS_TIMED("render_fwd", stream); hipLaunchKernelGGL(render_fwd_kernel, dim3(8 * ((ntiles + 7) / 8)), dim3(256), 0, stream, a); }
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
    FDGS_HIP_CHECK(hipMemsetAsync(g->scratch_acc, 0, P * 64, stream));
    if (p->cov3D_precomp) {   // scale/rotation gradients do not exist on this path: hand back zeros if buffers were given
        if (g->dL_dscales) FDGS_HIP_CHECK(hipMemsetAsync(g->dL_dscales, 0, P * 12, stream));
        if (g->dL_drotations) FDGS_HIP_CHECK(hipMemsetAsync(g->dL_drotations, 0, P * 16, stream));
    }
    GeomLayout gl = geom_layout(p->P);
    BinLayout bl = bin_layout(R);
    ImgLayout il = img_layout(p->W, p->H);
    if (R > 0) {
        RenderBwdArgs a{};
        a.W = p->W; a.H = p->H; a.gx = il.gx; a.gy = il.gy;
        a.ranges = at<uint2>(img, il.ranges); a.pair_gid = at<uint32_t>(binning, bl.gid0);
        a.recA = at<float4>(geom, gl.recA); a.recB = at<float4>(geom, gl.recB); a.recC = at<float4>(geom, gl.recC);
        a.bg = p->bg; a.final_T = at<float>(img, il.final_T); a.n_contrib = at<uint32_t>(img, il.n_contrib);
        a.dL_dcolor = g->dL_dcolor; a.dL_ddepth = g->dL_ddepth;
        a.gacc = g->scratch_acc;
        const int ntiles = il.gx * il.gy;
        {
            FDGS_TIMED("render_bwd", stream);
            // entries staged per round: 128 keeps six workgroups per CU (25 KB of LDS each), 256 halves the barriers at three per CU
            const int round = tunable("FDGS_RBWD_ROUND", 128);
            const dim3 grid(8 * ((ntiles + 7) / 8));
            if (g->dL_ddepth) {
                if (round == 256) hipLaunchKernelGGL((render_bwd_kernel<true, 256>), grid, dim3(256), 0, stream, a);
                else hipLaunchKernelGGL((render_bwd_kernel<true, 128>), grid, dim3(256), 0, stream, a);
            } else {
                if (round == 256) hipLaunchKernelGGL((render_bwd_kernel<false, 256>), grid, dim3(256), 0, stream, a);
                else hipLaunchKernelGGL((render_bwd_kernel<false, 128>), grid, dim3(256), 0, stream, a);
            }
        }
        FDGS_LAUNCH_CHECK("render_bwd", p->debug, stream);
    }
    return fdgs_launch_preprocess_bwd(stream, p, geom, g);
}
