// preprocess.hip -- K1 (per-Gaussian projection forward) and K8 (its backward) for gfx950.
// One thread per Gaussian, 256-thread workgroups (4 wave64).  Camera constants are read from device memory with
// wave-uniform (scalar) loads; per-Gaussian state is written as three float4 records so that the blending
// kernels stage a Gaussian with three 16-byte loads.
// Replaces preprocessCUDA / computeCov2DCUDA of the un-vendored rasterizer.
#include "common.h"
#include "gs_math.h"

namespace fdgs {

__device__ __forceinline__ void load_cam(CamConst& c, const float* __restrict__ view, const float* __restrict__ proj,
                                         const float* __restrict__ campos, int W, int H, float tanfovx, float tanfovy,
                                         float scale_mod, int D, int M) {
#pragma unroll
    for (int i = 0; i < 16; i++) { c.view[i] = view[i]; c.proj[i] = proj[i]; }
    if (campos) { c.campos[0] = campos[0]; c.campos[1] = campos[1]; c.campos[2] = campos[2]; }
    else { c.campos[0] = c.campos[1] = c.campos[2] = 0.f; }
    c.tanfovx = tanfovx; c.tanfovy = tanfovy;
    c.focal_x = (float)W / (2.0f * tanfovx); c.focal_y = (float)H / (2.0f * tanfovy);
    c.scale_mod = scale_mod; c.W = W; c.H = H;
    c.gx = (W + TILE - 1) / TILE; c.gy = (H + TILE - 1) / TILE;
    c.D = D; c.M = M;
}

struct PreFwdArgs {
    int P, D, M, W, H;
    float tanfovx, tanfovy, scale_mod;
    const float *view, *proj, *campos, *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    float* depth; float4 *recA, *recB, *recC; float* cov3D;
    uint32_t *tiles, *clamped; uint2* rect; uint32_t *keys, *ids, *total, *host_count;
    int32_t* radii;
    uint8_t* visibility;         // opt: radii > 0
    int cull; uint4* cullmask;   // exact tile culling (gs_math.h): 256-bit tile mask per Gaussian, two uint4 each
};

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void __launch_bounds__(256) preprocess_fwd_kernel(PreFwdArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    CamConst c;
    load_cam(c, a.view, a.proj, a.campos, a.W, a.H, a.tanfovx, a.tanfovy, a.scale_mod, a.D, a.M);
    uint32_t my_tiles = 0;
    if (i < a.P) {
        float p[3] = {a.means3D[3 * (size_t)i], a.means3D[3 * (size_t)i + 1], a.means3D[3 * (size_t)i + 2]};
        float c6[6];
        if (a.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) c6[k] = a.cov3D_precomp[6 * (size_t)i + k];
        } else {
            float s[3] = {a.scales[3 * (size_t)i], a.scales[3 * (size_t)i + 1], a.scales[3 * (size_t)i + 2]};
            const float4 qv = reinterpret_cast<const float4*>(a.rotations)[i];
            float q[4] = {qv.x, qv.y, qv.z, qv.w};
            cov3d_from_scale_rot(s, c.scale_mod, q, c6);
        }
        GeoOut g;
        bool vis = project_gaussian(c, p, c6, &g);
        int32_t radius = 0;
        uint32_t key = 0xFFFFFFFFu;
        if (vis) {
            float rgb[3];
            uint32_t cl = 0;
            if (a.colors_precomp) {
                rgb[0] = a.colors_precomp[3 * (size_t)i]; rgb[1] = a.colors_precomp[3 * (size_t)i + 1];
                rgb[2] = a.colors_precomp[3 * (size_t)i + 2];
            } else {
                const float* sh = a.shs + (size_t)i * a.M * 3;
                if (a.M == 16) {
                    float shl[48];
                    const float4* s4 = reinterpret_cast<const float4*>(sh);
#pragma unroll
                    for (int k = 0; k < 12; k++) {
                        float4 v = s4[k];
                        shl[4 * k] = v.x; shl[4 * k + 1] = v.y; shl[4 * k + 2] = v.z; shl[4 * k + 3] = v.w;
                    }
                    cl = sh_to_rgb(c.D, shl, p, c.campos, rgb);
                } else {
                    cl = sh_to_rgb(c.D, sh, p, c.campos, rgb);
                }
            }
            a.depth[i] = g.depth;
            a.recA[i] = make_float4(g.px, g.py, g.conic[0], g.conic[1]);
            a.recB[i] = make_float4(g.conic[2], a.opacities[i], g.depth, 0.f);
            a.recC[i] = make_float4(rgb[0], rgb[1], rgb[2], 0.f);
#pragma unroll
            for (int k = 0; k < 6; k++) a.cov3D[6 * (size_t)i + k] = c6[k];
            a.clamped[i] = cl;
            a.rect[i] = make_uint2(g.rect_min, g.rect_max);
            radius = g.radius;
            my_tiles = (uint32_t)g.tiles;
            if (a.cull && g.tiles <= 256) {
                // which tiles of the 3-sigma square can receive a contribution at all: bit (ty - y0) * w + (tx - x0)
                const int rx0 = (int)(g.rect_min & 0xFFFFu), ry0 = (int)(g.rect_min >> 16);
                const int rx1 = (int)(g.rect_max & 0xFFFFu), ry1 = (int)(g.rect_max >> 16), w = rx1 - rx0;
                unsigned long long m[4] = {0ull, 0ull, 0ull, 0ull};
                const CullEllipse E = cull_setup(g.conic, a.opacities[i]);
                uint32_t cnt = 0;
                if (E.mode != 0) {
                    for (int ty = ry0; ty < ry1; ty++) {
                        int lo = rx0, hi = rx1 - 1;
                        if (E.mode == 1) {
                            int tl, th;
                            if (!cull_row(E, g.px, g.py, ty, &tl, &th)) continue;
                            lo = tl > lo ? tl : lo; hi = th < hi ? th : hi;
                            if (lo > hi) continue;
                        }
                        const int s0 = (ty - ry0) * w + (lo - rx0), s1 = s0 + (hi - lo);
                        cnt += (uint32_t)(hi - lo + 1);
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int s = s0 > 64 * q ? s0 : 64 * q, e = s1 < 64 * q + 63 ? s1 : 64 * q + 63;
                            if (s <= e) m[q] |= (~0ull >> (63 - (e - s))) << (s - 64 * q);
                        }
                    }
                }
                my_tiles = cnt;
                a.cullmask[2 * (size_t)i] = make_uint4((uint32_t)m[0], (uint32_t)(m[0] >> 32), (uint32_t)m[1], (uint32_t)(m[1] >> 32));
                a.cullmask[2 * (size_t)i + 1] = make_uint4((uint32_t)m[2], (uint32_t)(m[2] >> 32), (uint32_t)m[3], (uint32_t)(m[3] >> 32));
            }
            key = __float_as_uint(g.depth);  // depth > 0.2: the raw bits order like the value
        }
        a.tiles[i] = my_tiles;
        a.radii[i] = radius;
        if (a.visibility) a.visibility[i] = radius > 0 ? 1 : 0;
        a.keys[i] = key;
        a.ids[i] = (uint32_t)i;
    }
    // ONE atomic per workgroup on the pair total (4 688 per-wave atomics on one address were serialised at the L2: round 4)
    __shared__ uint32_t wsum[4];
    const uint32_t s = wave_sum_u32(my_tiles);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (!a.host_count) {
            if (t) atomicAdd(a.total, t);
        } else {
            // capacity mode: the pair count goes to the host the moment it exists.  total[0] (pairs) and total[1] (workgroups done) are ONE
            // 64-bit counter: the workgroup whose add completes it knows the final count and stores it into the pinned, device-visible
            // host word (system scope) -- the host can size / verify the binning buffer while the depth sort is still running
            const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(a.total), (1ull << 32) | (unsigned long long)t);
            if ((uint32_t)(old >> 32) == gridDim.x - 1)
                __hip_atomic_store(a.host_count, (uint32_t)old + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (i == 0) a.total[2] = a.cull ? 1u : 0u;   // the pair expansion reads which list semantics the counts have
}

struct PreBwdArgs {
    int P, D, M, W, H;
    float tanfovx, tanfovy, scale_mod;
    const float *view, *proj, *campos, *means3D, *shs, *scales, *rotations;
    int has_cov_precomp;
    const float* cov3D; const uint32_t *tiles, *clamped;
    const float* gacc;           // [P,16] packed sums written by render_bwd (layout in render.hip)
    float *dL_dmeans2D, *dL_dopacity, *dL_dcolors;
    float *dL_dmeans3D, *dL_dsh, *dL_dscales, *dL_drot, *dL_dcov3D;
    // EPI: the deformation epilogue (fdgs_raster_deform_epilogue) + the activated opacities the rasterizer received
    fdgs_raster_deform_epilogue epi;
    const float* opacities;
};

// Block-linear store of K floats per Gaussian for the 64 Gaussians of a wave: lanes park their K values in LDS as
// [gaussian][K] (the memory layout of the output block) and the wave then copies the block with lane-consecutive
// addresses.  Stores of 4..24 bytes at 12..192-byte strides cost 3.7x write amplification in the round-1 kernel
// (rocprofv3 WRITE_SIZE 314 MB for 85 MB of results, profiles/r01c_pmc_write_size.txt).
template <int K>
__device__ __forceinline__ void wave_store_rows(float* __restrict__ out_block, const float* vals, int nvalid, float* buf, int lane) {
    __builtin_amdgcn_wave_barrier();
    if constexpr (K % 4 == 0) {
#pragma unroll
        for (int i = 0; i < K / 4; i++)
            reinterpret_cast<float4*>(buf)[lane * (K / 4) + i] = make_float4(vals[4 * i], vals[4 * i + 1], vals[4 * i + 2], vals[4 * i + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < K / 4; j++) {
            const int v = j * 64 + lane;
            if (v < nvalid * (K / 4)) reinterpret_cast<float4*>(out_block)[v] = reinterpret_cast<const float4*>(buf)[v];
        }
    } else {
#pragma unroll
        for (int i = 0; i < K; i++) buf[lane * K + i] = vals[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < K; j++) {
            const int idx = j * 64 + lane;
            if (idx < nvalid * K) out_block[idx] = buf[idx];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// EPI: instead of dL_dmeans3D / dL_dscales / dL_drot / dL_dopacity / dL_dsh the kernel writes what fdgs_deform_bwd consumes: the packed
// pre-activation gradient rows G[Npad][64] (activation Jacobians applied) and the identity paths, block-linear like everything else
// (this is deform_bwd_prep_kernel's job done where the values are still in registers: one kernel and 2 x 236 B per Gaussian less).
template <bool EPI>
__global__ void __launch_bounds__(256) preprocess_bwd_kernel(PreBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float lds_all[4 * 64 * (EPI ? 64 : 48)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* buf = lds_all + wave * 64 * (EPI ? 64 : 48);
    const int n0 = (blockIdx.x * 4 + wave) * 64;
    if constexpr (EPI) {
        // zero fill of the caller's accumulate-into gradient range (planes, MLP) on the way: grid-stride float4 stores
        if (a.epi.zero_fill) {
            const size_t n4 = a.epi.zero_floats >> 2, stride = (size_t)gridDim.x * 256;
            for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += stride)
                reinterpret_cast<float4*>(a.epi.zero_fill)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (n0 >= (EPI ? a.epi.Npad : a.P)) return;
    const int i = n0 + lane;
    const int nvalid = a.P - n0 < 64 ? (a.P - n0 > 0 ? a.P - n0 : 0) : 64;
    const bool active = i < a.P && a.tiles[i] != 0;
    // EPI, accumulate mode with the reference's storage (separate contiguous [N,1,3] / [N,15,3] SH tensors): the old values of the
    // identity paths are requested NOW, block-linear, and added at the very end -- as a read-modify-write behind the chain rule
    // their latency doubled the kernel (0.157 vs 0.075 ms at 300 k)
    float old_rest[45], old_small[14];   // [0..2] dc, [3..5] xyz, [6..8] scales, [9..12] rotations, [13] opacity
    bool pre = false;
    if constexpr (EPI) {
        const fdgs_raster_deform_epilogue& e = a.epi;
        pre = !e.assign && e.d_shs_dc && e.d_shs_rest && e.shs_dc_stride == 3 && e.shs_rest_stride == 45 && e.d_xyz && e.d_scales &&
              e.d_rotations && e.d_opacity;
        if (pre) {
#pragma unroll
            for (int j = 0; j < 45; j++) { const int idx = j * 64 + lane; old_rest[j] = idx < nvalid * 45 ? e.d_shs_rest[(size_t)n0 * 45 + idx] : 0.f; }
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int idx = j * 64 + lane;
                const bool ok = idx < nvalid * 3;
                old_small[j] = ok ? e.d_shs_dc[(size_t)n0 * 3 + idx] : 0.f;
                old_small[3 + j] = ok ? e.d_xyz[(size_t)n0 * 3 + idx] : 0.f;
                old_small[6 + j] = ok ? e.d_scales[(size_t)n0 * 3 + idx] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) { const int idx = j * 64 + lane; old_small[9 + j] = idx < nvalid * 4 ? e.d_rotations[(size_t)n0 * 4 + idx] : 0.f; }
            old_small[13] = lane < nvalid ? e.d_opacity[i] : 0.f;
        }
    }
    // every output row is written (zeros for culled / zero-area Gaussians): the caller needs no memsets
    float m2d[3] = {0.f, 0.f, 0.f}, dop = 0.f, drgb[3] = {0.f, 0.f, 0.f}, dcov6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dmean[3] = {0.f, 0.f, 0.f}, ds[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
    float dsh[48];
#pragma unroll
    for (int k = 0; k < 48; k++) dsh[k] = 0.f;
    // The blending backward leaves an all-zero line for every Gaussian no pixel blended (behind saturated pixels: most of a dense scene).
    // Its chain rule is zero times everything: such a Gaussian skips the loads of its position / covariance / SH and the arithmetic and
    // writes its zero rows -- in spatial order whole waves do.  (Outputs identical: every term carries a factor from the line.)
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
    bool any_grad = false;
    if (active) {
        const float4* ga = reinterpret_cast<const float4*>(a.gacc + 16 * (size_t)i);
        g0 = ga[0]; g1 = ga[1]; g2 = ga[2];
        any_grad = g0.x != 0.f || g0.y != 0.f || g0.z != 0.f || g0.w != 0.f || g1.x != 0.f || g1.y != 0.f || g1.z != 0.f || g1.w != 0.f ||
                   g2.x != 0.f || g2.y != 0.f;
    }
    if (active && any_grad) {
        CamConst c;
        load_cam(c, a.view, a.proj, a.campos, a.W, a.H, a.tanfovx, a.tanfovy, a.scale_mod, a.D, a.M);
        float p[3] = {a.means3D[3 * (size_t)i], a.means3D[3 * (size_t)i + 1], a.means3D[3 * (size_t)i + 2]};
        float c6[6];
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = a.cov3D[6 * (size_t)i + k];
        // mean2D gradient is handed back in NDC units (d pix / d ndc = W/2, H/2), like the reference's viewspace grad
        m2d[0] = g0.x * 0.5f * (float)a.W; m2d[1] = g0.y * 0.5f * (float)a.H;
        dop = g1.y;
        drgb[0] = g1.z; drgb[1] = g1.w; drgb[2] = g2.x;
        float dconic[3] = {g0.z, g0.w, g1.x};
        project_bwd(c, p, c6, dconic, g2.y, m2d[0], m2d[1], dmean, dcov6);
        if (a.shs) {
            const float* sh = a.shs + (size_t)i * a.M * 3;
            sh_bwd(c.D, sh, p, c.campos, a.clamped[i], drgb, dsh, dmean);
        }
        if (!a.has_cov_precomp) {
            float s[3] = {a.scales[3 * (size_t)i], a.scales[3 * (size_t)i + 1], a.scales[3 * (size_t)i + 2]};
            const float4 qv = reinterpret_cast<const float4*>(a.rotations)[i];
            float q[4] = {qv.x, qv.y, qv.z, qv.w};
            cov3d_bwd(s, c.scale_mod, q, dcov6, ds, dq);
        }
    }
    if constexpr (EPI) {
        wave_store_rows<3>(a.dL_dmeans2D + (size_t)n0 * 3, m2d, nvalid, buf, lane);
        const fdgs_raster_deform_epilogue& e = a.epi;
        float row[16];
#pragma unroll
        for (int k = 0; k < 16; k++) row[k] = 0.f;
        if (i < a.P) {
            row[0] = dmean[0]; row[1] = dmean[1]; row[2] = dmean[2];
#pragma unroll
            for (int k = 0; k < 3; k++) row[3 + k] = e.activate ? ds[k] * a.scales[3 * (size_t)i + k] : ds[k];     // d exp
            if (e.activate) {
                const float4 o = reinterpret_cast<const float4*>(a.rotations)[i];
                const float nrm = e.rot_norm[i];
                if (nrm > 1e-12f) {
                    const float dot = o.x * dq[0] + o.y * dq[1] + o.z * dq[2] + o.w * dq[3];
                    const float inv = 1.0f / nrm;
                    row[6] = (dq[0] - o.x * dot) * inv; row[7] = (dq[1] - o.y * dot) * inv;
                    row[8] = (dq[2] - o.z * dot) * inv; row[9] = (dq[3] - o.w * dot) * inv;
                } else {  // below the F.normalize eps the division is by the constant 1e-12
                    row[6] = dq[0] * 1e12f; row[7] = dq[1] * 1e12f; row[8] = dq[2] * 1e12f; row[9] = dq[3] * 1e12f;
                }
                const float o1 = a.opacities[i];
                row[10] = dop * o1 * (1.f - o1);
            } else {
                row[6] = dq[0]; row[7] = dq[1]; row[8] = dq[2]; row[9] = dq[3];
                row[10] = dop;
            }
        }
        bool t0_live = true, t1_live = true;     // the wave's two 32-row tiles
        if (e.tile_flags) {
            // per 32-row tile of the deformation backward: does any row carry a gradient at all?  (culled / occluded / off-screen
            // Gaussians do not: fdgs_deform_bwd skips tiles made of them -- a zero row adds exactly zero to every sum)
            bool nz = false;
#pragma unroll
            for (int k = 0; k < 11; k++) nz = nz || (row[k] != 0.f);
#pragma unroll
            for (int k = 0; k < 48; k++) nz = nz || (dsh[k] != 0.f);
            const unsigned long long m = __ballot(nz);
            if (lane == 0) {
                uint32_t* tl = reinterpret_cast<uint32_t*>(e.G + (size_t)e.Npad * 64) + (n0 >> 5);
                tl[0] = (uint32_t)(m & 0xffffffffull);      // (bit r = row r of the tile is non-zero: fdgs_deform_bwd builds its row lists from these)
                tl[1] = (uint32_t)(m >> 32);
            }
            if (e.tile_flags == 2) { t0_live = (uint32_t)(m & 0xffffffffull) != 0u; t1_live = (uint32_t)(m >> 32) != 0u; }   // dead tiles' rows stay unwritten
        }
        float* small = buf;              // [64][16]
        float* sh = buf + 64 * 16;       // [64][48]
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++)
            reinterpret_cast<float4*>(small)[lane * 4 + k] = make_float4(row[4 * k], row[4 * k + 1], row[4 * k + 2], row[4 * k + 3]);
#pragma unroll
        for (int k = 0; k < 12; k++)
            reinterpret_cast<float4*>(sh)[lane * 12 + k] = make_float4(dsh[4 * k], dsh[4 * k + 1], dsh[4 * k + 2], dsh[4 * k + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        {
            float4* G4 = reinterpret_cast<float4*>(e.G + (size_t)n0 * 64);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if (!(j < 8 ? t0_live : t1_live)) continue;      // (rows 0..31 = pieces 0..7, rows 32..63 = pieces 8..15)
                const int v = j * 64 + lane, r = v >> 4, c4 = v & 15;
                G4[v] = c4 < 4 ? reinterpret_cast<const float4*>(small)[r * 4 + c4] : reinterpret_cast<const float4*>(sh)[r * 12 + (c4 - 4)];
            }
        }
        // identity paths (out = in + delta): accumulated into (or, e.assign, written as) the parameter gradients, block-linear
        const bool asg = e.assign != 0;
        if (pre) {      // accumulate mode, old values already in registers
#pragma unroll
            for (int j = 0; j < 45; j++) {
                const int idx = j * 64 + lane, r = idx / 45, c = idx - 45 * r;
                if (r < nvalid) e.d_shs_rest[(size_t)n0 * 45 + idx] = old_rest[j] + sh[r * 48 + 3 + c];
            }
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int idx = j * 64 + lane, r = idx / 3, c = idx - 3 * r;
                if (r < nvalid) {
                    e.d_shs_dc[(size_t)n0 * 3 + idx] = old_small[j] + sh[r * 48 + c];
                    e.d_xyz[(size_t)n0 * 3 + idx] = old_small[3 + j] + small[r * 16 + c];
                    e.d_scales[(size_t)n0 * 3 + idx] = old_small[6 + j] + small[r * 16 + 3 + c];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int idx = j * 64 + lane, r = idx >> 2, c = idx & 3;
                if (r < nvalid) e.d_rotations[(size_t)n0 * 4 + idx] = old_small[9 + j] + small[r * 16 + 6 + c];
            }
            if (lane < nvalid) e.d_opacity[i] = old_small[13] + small[lane * 16 + 10];
            return;
        }
        if (e.d_shs_dc && e.d_shs_rest && e.shs_dc_stride == 48 && e.shs_rest_stride == 48 && e.d_shs_rest == e.d_shs_dc + 3) {
            float4* d4 = reinterpret_cast<float4*>(e.d_shs_dc + (size_t)n0 * 48);   // one combined [N,16,3] tensor
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const int v = j * 64 + lane;
                if (v < nvalid * 12) {
                    float4 y = reinterpret_cast<const float4*>(sh)[v];
                    if (!asg) { const float4 x = d4[v]; y.x += x.x; y.y += x.y; y.z += x.z; y.w += x.w; }
                    d4[v] = y;
                }
            }
        } else {
            if (e.d_shs_dc) {
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int idx = j * 64 + lane, r = idx / 3, c = idx - 3 * r;
                    if (r < nvalid) {
                        float* q = e.d_shs_dc + (size_t)(n0 + r) * e.shs_dc_stride + c;
                        *q = (asg ? 0.f : *q) + sh[r * 48 + c];
                    }
                }
            }
            if (e.d_shs_rest) {
                if (asg && e.shs_rest_stride == 45) {      // a contiguous [N,15,3] tensor: the wave's block is 64 x 45 consecutive floats
                    for (int j = 0; j < 45; j++) {
                        const int idx = j * 64 + lane, r = idx / 45, c = idx - 45 * r;
                        if (r < nvalid) e.d_shs_rest[(size_t)n0 * 45 + idx] = sh[r * 48 + 3 + c];
                    }
                } else {
                    for (int j = 0; j < 45; j++) {
                        const int idx = j * 64 + lane, r = idx / 45, c = idx - 45 * r;
                        if (r < nvalid) {
                            float* q = e.d_shs_rest + (size_t)(n0 + r) * e.shs_rest_stride + c;
                            *q = (asg ? 0.f : *q) + sh[r * 48 + 3 + c];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int idx = j * 64 + lane, r = idx / 3, c = idx - 3 * r;
            if (r < nvalid) {
                if (e.d_xyz) e.d_xyz[(size_t)n0 * 3 + idx] = (asg ? 0.f : e.d_xyz[(size_t)n0 * 3 + idx]) + small[r * 16 + c];
                if (e.d_scales) e.d_scales[(size_t)n0 * 3 + idx] = (asg ? 0.f : e.d_scales[(size_t)n0 * 3 + idx]) + small[r * 16 + 3 + c];
            }
        }
        if (e.d_rotations) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int idx = j * 64 + lane, r = idx >> 2, c = idx & 3;
                if (r < nvalid) e.d_rotations[(size_t)n0 * 4 + idx] = (asg ? 0.f : e.d_rotations[(size_t)n0 * 4 + idx]) + small[r * 16 + 6 + c];
            }
        }
        if (e.d_opacity && lane < nvalid) e.d_opacity[i] = (asg ? 0.f : e.d_opacity[i]) + small[lane * 16 + 10];
        return;
    }
    wave_store_rows<3>(a.dL_dmeans2D + (size_t)n0 * 3, m2d, nvalid, buf, lane);
    wave_store_rows<3>(a.dL_dcolors + (size_t)n0 * 3, drgb, nvalid, buf, lane);
    wave_store_rows<3>(a.dL_dmeans3D + (size_t)n0 * 3, dmean, nvalid, buf, lane);
    wave_store_rows<6>(a.dL_dcov3D + (size_t)n0 * 6, dcov6, nvalid, buf, lane);
    if (i < a.P) a.dL_dopacity[i] = dop;
    if (a.shs) {
        if (a.M == 16) {
            wave_store_rows<48>(a.dL_dsh + (size_t)n0 * 48, dsh, nvalid, buf, lane);
        } else if (i < a.P) {
            float* out = a.dL_dsh + (size_t)i * a.M * 3;
#pragma unroll
            for (int k = 0; k < 48; k++) if (k < a.M * 3) out[k] = dsh[k];
        }
    }
    if (!a.has_cov_precomp) {
        wave_store_rows<3>(a.dL_dscales + (size_t)n0 * 3, ds, nvalid, buf, lane);
        if (i < a.P) reinterpret_cast<float4*>(a.dL_drot)[i] = make_float4(dq[0], dq[1], dq[2], dq[3]);
    }
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                           const float* __restrict__ view, uint8_t* present) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float z = view[2] * means3D[3 * (size_t)i] + view[6] * means3D[3 * (size_t)i + 1] + view[10] * means3D[3 * (size_t)i + 2] + view[14];
    present[i] = z > FDGS_NEAR_CULL ? 1 : 0;
}

int validate_raster_params(const fdgs_raster_params* p) {
    FDGS_REQUIRE(p != nullptr, "params is NULL");
    FDGS_REQUIRE(p->P >= 0 && p->W > 0 && p->H > 0, "bad P/W/H");
    FDGS_REQUIRE(p->sh_degree >= 0 && p->sh_degree <= 3, "sh_degree must be 0..3");
    FDGS_REQUIRE((p->shs != nullptr) != (p->colors_precomp != nullptr) || p->P == 0,
                 "Please provide excatly one of either SHs or precomputed colors!");
    FDGS_REQUIRE((((p->scales != nullptr) && (p->rotations != nullptr)) != (p->cov3D_precomp != nullptr)) || p->P == 0,
                 "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    FDGS_REQUIRE(((p->scales != nullptr) == (p->rotations != nullptr)) || p->P == 0, "scales and rotations come as a pair");
    if (p->shs) {
        FDGS_REQUIRE(p->sh_coeffs >= (p->sh_degree + 1) * (p->sh_degree + 1) && p->sh_coeffs <= FDGS_SH_COEFFS,
                     "sh_coeffs must cover the active degree and be <= 16");
    }
    FDGS_REQUIRE(p->bg && p->viewmatrix && p->projmatrix && (p->campos || !p->shs), "camera pointers missing");
    FDGS_REQUIRE((p->W + TILE - 1) / TILE < 65536 && (p->H + TILE - 1) / TILE < 65536, "image too large");
    return FDGS_OK;
}

}  // namespace fdgs

using namespace fdgs;

// host_count_dev (capacity mode): device address of the pinned host word that receives the pair count from the kernel's last workgroup
int fdgs_preprocess_fwd_impl(void* stream_, const fdgs_raster_params* p, void* geom, int32_t* radii, uint32_t* host_count_dev) {
    int rc = validate_raster_params(p);
    if (rc) return rc;
    FDGS_REQUIRE(geom && (radii || p->P == 0), "geom/radii is NULL");
    hipStream_t stream = (hipStream_t)stream_;
    GeomLayout gl = geom_layout(p->P);
    FDGS_HIP_CHECK(hipMemsetAsync(at<char>(geom, gl.total), 0, 256, stream));
    if (p->P == 0) return FDGS_OK;
    PreFwdArgs a{};
    a.P = p->P; a.D = p->sh_degree; a.M = p->sh_coeffs; a.W = p->W; a.H = p->H;
    a.tanfovx = p->tanfovx; a.tanfovy = p->tanfovy; a.scale_mod = p->scale_modifier;
    a.view = p->viewmatrix; a.proj = p->projmatrix; a.campos = p->campos; a.means3D = p->means3D; a.shs = p->shs;
    a.colors_precomp = p->colors_precomp; a.opacities = p->opacities; a.scales = p->scales; a.rotations = p->rotations;
    a.cov3D_precomp = p->cov3D_precomp;
    a.depth = at<float>(geom, gl.depth); a.recA = at<float4>(geom, gl.recA); a.recB = at<float4>(geom, gl.recB);
    a.recC = at<float4>(geom, gl.recC); a.cov3D = at<float>(geom, gl.cov3D); a.tiles = at<uint32_t>(geom, gl.tiles);
    a.clamped = at<uint32_t>(geom, gl.clamped); a.rect = at<uint2>(geom, gl.rect); a.keys = at<uint32_t>(geom, gl.keys0);
    a.ids = at<uint32_t>(geom, gl.ids0); a.total = at<uint32_t>(geom, gl.total); a.radii = radii; a.visibility = p->visibility;
    a.cull = g_tune.tile_cull != 0; a.cullmask = at<uint4>(geom, gl.cullmask);
    a.host_count = host_count_dev;
    { FDGS_TIMED("preprocess_fwd", stream); hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(cdiv(p->P, 256)), dim3(256), 0, stream, a); }
    FDGS_LAUNCH_CHECK("preprocess_fwd", p->debug, stream);
    return FDGS_OK;
}

extern "C" int fdgs_preprocess_fwd(void* stream_, const fdgs_raster_params* p, void* geom, int32_t* radii) {
    return fdgs_preprocess_fwd_impl(stream_, p, geom, radii, nullptr);
}

// called from fdgs_raster_bwd (render.hip)
int fdgs_launch_preprocess_bwd(hipStream_t stream, const fdgs_raster_params* p, const void* geom,
                               const fdgs_raster_grads* g) {
    if (p->P == 0) return FDGS_OK;
    GeomLayout gl = geom_layout(p->P);
    PreBwdArgs a{};
    a.P = p->P; a.D = p->sh_degree; a.M = p->sh_coeffs; a.W = p->W; a.H = p->H;
    a.tanfovx = p->tanfovx; a.tanfovy = p->tanfovy; a.scale_mod = p->scale_modifier;
    a.view = p->viewmatrix; a.proj = p->projmatrix; a.campos = p->campos; a.means3D = p->means3D; a.shs = p->shs;
    a.scales = p->scales; a.rotations = p->rotations; a.has_cov_precomp = p->cov3D_precomp != nullptr;
    a.cov3D = at<float>(geom, gl.cov3D); a.tiles = at<uint32_t>(geom, gl.tiles); a.clamped = at<uint32_t>(geom, gl.clamped);
    a.dL_dmeans2D = g->dL_dmeans2D; a.gacc = g->scratch_acc; a.dL_dopacity = g->dL_dopacity;
    a.dL_dcolors = g->dL_dcolors; a.dL_dmeans3D = g->dL_dmeans3D; a.dL_dsh = g->dL_dsh; a.dL_dscales = g->dL_dscales;
    a.dL_drot = g->dL_drotations; a.dL_dcov3D = g->dL_dcov3D;
    a.opacities = p->opacities;
    if (g->deform_epilogue) {
        const fdgs_raster_deform_epilogue* e = g->deform_epilogue;
        FDGS_REQUIRE(p->shs && p->sh_coeffs == 16 && p->scales && p->rotations && !p->cov3D_precomp,
                     "deform_epilogue needs SH (16 coefficients) and scale/rotation inputs");
        FDGS_REQUIRE(e->G && e->Npad >= p->P && e->Npad % 128 == 0 && (!e->activate || e->rot_norm), "bad deform_epilogue");
        FDGS_REQUIRE(!e->zero_fill || ((reinterpret_cast<uintptr_t>(e->zero_fill) & 15) == 0 && (e->zero_floats & 3) == 0),
                     "deform_epilogue.zero_fill must be 16-byte aligned with a multiple of 4 floats");
        a.epi = *e;
        { FDGS_TIMED("preprocess_bwd", stream); hipLaunchKernelGGL(preprocess_bwd_kernel<true>, dim3(cdiv(e->Npad, 256)), dim3(256), 0, stream, a); }
        FDGS_LAUNCH_CHECK("preprocess_bwd", p->debug, stream);
        return FDGS_OK;
    }
    { FDGS_TIMED("preprocess_bwd", stream); hipLaunchKernelGGL(preprocess_bwd_kernel<false>, dim3(cdiv(p->P, 256)), dim3(256), 0, stream, a); }
    FDGS_LAUNCH_CHECK("preprocess_bwd", p->debug, stream);
    return FDGS_OK;
}

extern "C" int fdgs_mark_visible(void* stream_, int P, const float* means3D, const float* viewmatrix,
                                 const float* projmatrix, uint8_t* present) {
    (void)projmatrix;
    FDGS_REQUIRE(P >= 0 && (P == 0 || (means3D && viewmatrix && present)), "bad arguments");
    if (P == 0) return FDGS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    { FDGS_TIMED("mark_visible", stream); hipLaunchKernelGGL(mark_visible_kernel, dim3(cdiv(P, 256)), dim3(256), 0, stream, P, means3D, viewmatrix, present); }
    FDGS_LAUNCH_CHECK("mark_visible", 0, stream);
    return FDGS_OK;
}
