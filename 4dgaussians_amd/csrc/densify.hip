// densify.hip -- densification bookkeeping of the Gaussian set: per-iteration statistics, clone / split / prune.
//
// Replaces, on the consumer side of render()'s `radii` and `viewspace_points.grad`:
//   every iteration   train.py:259-262 + GaussianModel.add_densification_stats (scene/gaussian_model.py:516-518):
//                     three boolean-mask read-modify-writes (each a nonzero() with a host sync + gather + scatter)
//                     -> fdgs_densification_stats, one launch, no sync
//   every 100 it.     GaussianModel.densify -> densify_and_clone + densify_and_split (scene/gaussian_model.py:409-456,
//                     495-500) and prune / prune_points (:350-365, 481-494), each rebuilding the six per-Gaussian
//                     Parameters and both Adam moments through cat_tensors_to_optimizer / _prune_optimizer
//                     (:331-348, 367-389): a few hundred small launches and several host syncs
//                     -> plan (classify + scan, ONE readback of the three counts) + apply (one launch that writes every
//                     output row of all 6 x 3 arrays and the per-Gaussian side arrays exactly once)
//
// Result order = what the reference's clone-then-split-then-prune sequence leaves behind:
//   [ originals that are not split, in order | clones, in order | first child of every split Gaussian | second child ]
// (the clone pass appends copies; the split pass sees zero gradient for those copies, appends N=2 children per selected
// original in repeat() order and then removes the selected originals).  Children: xyz' = R(q/|q|) (n * exp(s)) + xyz with
// n ~ N(0,1) supplied by the caller ([2*splits][3], row k*splits + r for child k of the r-th split Gaussian, the layout of
// torch.normal over `stds.repeat(2,1)`), s' = log(exp(s) / 1.6); all other rows are copies; Adam moments of new rows = 0.
// HBM-bound streaming copies: N x (59 floats x 3 arrays) read + written once.
#include "common.h"

namespace fdgs {

constexpr int DN_ITEMS = 4;                    // Gaussians per thread
constexpr int DN_BLOCK = 256 * DN_ITEMS;       // Gaussians per workgroup
enum : uint8_t { DN_KEEP = 1, DN_CLONE = 2, DN_SPLIT = 4 };

struct DensifyScratch {
    uint8_t* flags; uint32_t* blockcnt; uint32_t* blockoff; uint32_t* totals; size_t bytes; int nb;
};
inline DensifyScratch densify_scratch(void* base, int N) {
    DensifyScratch s{};
    s.nb = cdiv(N > 0 ? N : 1, DN_BLOCK);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes); return r; };
    const size_t f = take((size_t)s.nb * DN_BLOCK), c = take((size_t)s.nb * 3 * 4), b = take((size_t)s.nb * 3 * 4), t = take(16);
    s.bytes = o;
    char* p = (char*)base;
    s.flags = (uint8_t*)(p + f); s.blockcnt = (uint32_t*)(p + c); s.blockoff = (uint32_t*)(p + b); s.totals = (uint32_t*)(p + t);
    return s;
}

__global__ void __launch_bounds__(256) densification_stats_kernel(int N, const int* __restrict__ radii, const uint8_t* __restrict__ vis,
                                                                  const float* __restrict__ vg, int stride, float* __restrict__ max_radii,
                                                                  float* __restrict__ accum, float* __restrict__ denom) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const bool v = vis ? vis[i] != 0 : radii[i] > 0;
    if (!v) return;
    if (max_radii && radii) max_radii[i] = fmaxf(max_radii[i], (float)radii[i]);
    const float gx = vg[(size_t)i * stride], gy = vg[(size_t)i * stride + 1];
    accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.f;
}

struct PlanArgs {
    int mode, N;
    const float* accum; const float* denom; const float* scaling; const float* opacity; const float* max_radii; const uint8_t* drop_mask;
    float grad_threshold, dense_size, min_opacity, max_screen_size, max_world_size;
    uint8_t* flags; uint32_t* blockcnt;
};

__global__ void __launch_bounds__(256) densify_classify_kernel(PlanArgs a) {
    __shared__ uint32_t cnt[3];
    if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t nk = 0, nc = 0, ns = 0;
#pragma unroll
    for (int k = 0; k < DN_ITEMS; k++) {
        const int i = blockIdx.x * DN_BLOCK + threadIdx.x * DN_ITEMS + k;
        uint8_t f = 0;
        if (i < a.N) {
            float mx = 0.f;
            if (a.scaling) mx = fmaxf(fmaxf(expf(a.scaling[3 * i]), expf(a.scaling[3 * i + 1])), expf(a.scaling[3 * i + 2]));
            if (a.mode == FDGS_PLAN_DENSIFY) {
                float g = a.accum[i] / a.denom[i];
                if (g != g) g = 0.f;                                   // grads[grads.isnan()] = 0
                const bool big = mx > a.dense_size;
                const bool clone = fabsf(g) >= a.grad_threshold && !big;
                const bool split = g >= a.grad_threshold && big;
                f = (split ? DN_SPLIT : DN_KEEP) | (clone ? DN_CLONE : 0);
            } else if (a.mode == FDGS_PLAN_PRUNE) {
                const float op = 1.f / (1.f + expf(-a.opacity[i]));
                bool drop = op < a.min_opacity;
                if (a.max_screen_size > 0.f) drop = drop || a.max_radii[i] > a.max_screen_size || mx > a.max_world_size;
                f = drop ? 0 : DN_KEEP;
            } else {
                f = a.drop_mask[i] ? 0 : DN_KEEP;
            }
        }
        a.flags[(size_t)blockIdx.x * DN_BLOCK + threadIdx.x * DN_ITEMS + k] = f;
        nk += (f & DN_KEEP) ? 1 : 0; nc += (f & DN_CLONE) ? 1 : 0; ns += (f & DN_SPLIT) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nk += __shfl_xor(nk, o, 64); nc += __shfl_xor(nc, o, 64); ns += __shfl_xor(ns, o, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&cnt[0], nk); atomicAdd(&cnt[1], nc); atomicAdd(&cnt[2], ns); }
    __syncthreads();
    if (threadIdx.x < 3) a.blockcnt[blockIdx.x * 3 + threadIdx.x] = cnt[threadIdx.x];
}

// exclusive scan of the per-workgroup counts (one workgroup; nb <= a few thousand)
__global__ void __launch_bounds__(256) densify_scan_kernel(int nb, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ off,
                                                           uint32_t* __restrict__ totals) {
    __shared__ uint32_t wtmp[4];
    uint32_t carry[3] = {0, 0, 0};
    for (int base = 0; base < nb; base += 256) {
        const int b = base + threadIdx.x;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const uint32_t v = b < nb ? cnt[b * 3 + q] : 0u;
            uint32_t tot;
            const uint32_t e = block_excl_scan_256(v, wtmp, &tot);
            if (b < nb) off[b * 3 + q] = carry[q] + e;
            carry[q] += tot;
        }
    }
    if (threadIdx.x == 0) { totals[0] = carry[0]; totals[1] = carry[1]; totals[2] = carry[2]; totals[3] = 0; }
}

struct ApplyArgs {
    fdgs_gaussians_in in;
    fdgs_gaussians_out out;
    const uint8_t* flags; const uint32_t* blockoff; const uint32_t* totals;
    const float* samples;
};

__global__ void __launch_bounds__(256) densify_apply_kernel(ApplyArgs a) {
    __shared__ int dk[DN_BLOCK], dc[DN_BLOCK], ds[DN_BLOCK];
    __shared__ uint32_t wtmp[4];
    const int N = a.in.N, b = blockIdx.x, t = threadIdx.x;
    const uint32_t NK = a.totals[0], NC = a.totals[1], NS = a.totals[2];
    {   // destination rows of this workgroup's Gaussians
        uint8_t f[DN_ITEMS];
        uint32_t c[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < DN_ITEMS; k++) {
            f[k] = a.flags[(size_t)b * DN_BLOCK + t * DN_ITEMS + k];
            c[0] += (f[k] & DN_KEEP) ? 1 : 0; c[1] += (f[k] & DN_CLONE) ? 1 : 0; c[2] += (f[k] & DN_SPLIT) ? 1 : 0;
        }
        uint32_t run[3], tot;
#pragma unroll
        for (int q = 0; q < 3; q++) run[q] = a.blockoff[b * 3 + q] + block_excl_scan_256(c[q], wtmp, &tot);
#pragma unroll
        for (int k = 0; k < DN_ITEMS; k++) {
            const int il = t * DN_ITEMS + k;
            dk[il] = (f[k] & DN_KEEP) ? (int)(run[0]++) : -1;
            dc[il] = (f[k] & DN_CLONE) ? (int)(NK + run[1]++) : -1;
            ds[il] = (f[k] & DN_SPLIT) ? (int)(NK + NC + run[2]++) : -1;
        }
    }
    __syncthreads();
    const int i0 = b * DN_BLOCK;
    const int nloc = min(DN_BLOCK, N - i0);
    // the six parameter groups with their two Adam moments: copies (children of xyz / scaling are written further down)
#pragma unroll 1
    for (int g = 0; g < FDGS_NGROUPS; g++) {
        const int w = a.in.width[g];
        if (w <= 0) continue;
        const float* __restrict__ P = a.in.param[g];
        const float* __restrict__ M = a.in.exp_avg[g];
        const float* __restrict__ V = a.in.exp_avg_sq[g];
        float* __restrict__ Po = a.out.param[g];
        float* __restrict__ Mo = a.out.exp_avg[g];
        float* __restrict__ Vo = a.out.exp_avg_sq[g];
        const bool special = g == 0 || g == 4;
        for (int e = t; e < nloc * w; e += 256) {
            const int il = e / w, c = e - il * w;
            const size_t src = (size_t)(i0 + il) * w + c;
            const float p = P[src];
            const int k = dk[il], cl = dc[il], sp = ds[il];
            if (k >= 0) {
                const size_t d = (size_t)k * w + c;
                Po[d] = p;
                if (Mo) { Mo[d] = M ? M[src] : 0.f; Vo[d] = V ? V[src] : 0.f; }
            }
            if (cl >= 0) {
                const size_t d = (size_t)cl * w + c;
                Po[d] = p;
                if (Mo) { Mo[d] = 0.f; Vo[d] = 0.f; }
            }
            if (sp >= 0) {
                const size_t d0 = (size_t)sp * w + c, d1 = (size_t)(sp + NS) * w + c;
                if (!special) { Po[d0] = p; Po[d1] = p; }
                if (Mo) { Mo[d0] = 0.f; Vo[d0] = 0.f; Mo[d1] = 0.f; Vo[d1] = 0.f; }
            }
        }
    }
    // children of split Gaussians: position sampled inside the parent, scale shrunk by 0.8 * 2
    for (int e = t; e < nloc * 2; e += 256) {
        const int il = e >> 1, child = e & 1;
        const int sp = ds[il];
        if (sp < 0) continue;
        const size_t i = (size_t)(i0 + il);
        const float* __restrict__ X = a.in.param[0] + i * 3;
        const float* __restrict__ S = a.in.param[4] + i * 3;
        const float* __restrict__ Q = a.in.param[5] + i * 4;
        const uint32_t r = (uint32_t)sp - NK - NC;
        const float* __restrict__ nrm = a.samples ? a.samples + ((size_t)child * NS + r) * 3 : nullptr;   // NULL: children sit on the parent
        const float e0 = expf(S[0]), e1 = expf(S[1]), e2 = expf(S[2]);
        const float v0 = nrm ? nrm[0] * e0 : 0.f, v1 = nrm ? nrm[1] * e1 : 0.f, v2 = nrm ? nrm[2] * e2 : 0.f;
        const float qn = sqrtf(Q[0] * Q[0] + Q[1] * Q[1] + Q[2] * Q[2] + Q[3] * Q[3]);
        const float qr = Q[0] / qn, qx = Q[1] / qn, qy = Q[2] / qn, qz = Q[3] / qn;     // utils/general_utils.py:84-105
        const size_t d = (size_t)(sp + child * NS) * 3;
        float* __restrict__ Xo = a.out.param[0];
        float* __restrict__ So = a.out.param[4];
        Xo[d + 0] = (1.f - 2.f * (qy * qy + qz * qz)) * v0 + 2.f * (qx * qy - qr * qz) * v1 + 2.f * (qx * qz + qr * qy) * v2 + X[0];
        Xo[d + 1] = 2.f * (qx * qy + qr * qz) * v0 + (1.f - 2.f * (qx * qx + qz * qz)) * v1 + 2.f * (qy * qz - qr * qx) * v2 + X[1];
        Xo[d + 2] = 2.f * (qx * qz - qr * qy) * v0 + 2.f * (qy * qz + qr * qx) * v1 + (1.f - 2.f * (qx * qx + qy * qy)) * v2 + X[2];
        So[d + 0] = logf(e0 / 1.6f); So[d + 1] = logf(e1 / 1.6f); So[d + 2] = logf(e2 / 1.6f);
    }
    // per-Gaussian side arrays: the deformation table follows its Gaussian everywhere, statistics survive on kept rows only
    for (int il = t; il < nloc; il += 256) {
        const size_t i = (size_t)(i0 + il);
        const int k = dk[il], cl = dc[il], sp = ds[il];
        if (a.out.deformation_table) {
            const uint8_t v = a.in.deformation_table ? a.in.deformation_table[i] : 1;
            if (k >= 0) a.out.deformation_table[k] = v;
            if (cl >= 0) a.out.deformation_table[cl] = v;
            if (sp >= 0) { a.out.deformation_table[sp] = v; a.out.deformation_table[sp + NS] = v; }
        }
        auto carry = [&](const float* src, float* dst, int w) {
            if (!dst) return;
            for (int c = 0; c < w; c++) {
                if (k >= 0) dst[(size_t)k * w + c] = src ? src[i * w + c] : 0.f;
                if (cl >= 0) dst[(size_t)cl * w + c] = 0.f;
                if (sp >= 0) { dst[(size_t)sp * w + c] = 0.f; dst[(size_t)(sp + NS) * w + c] = 0.f; }
            }
        };
        carry(a.in.xyz_gradient_accum, a.out.xyz_gradient_accum, 1);
        carry(a.in.denom, a.out.denom, 1);
        carry(a.in.max_radii2D, a.out.max_radii2D, 1);
        carry(a.in.deformation_accum, a.out.deformation_accum, 3);
    }
}
}  // namespace fdgs

using namespace fdgs;

extern "C" int fdgs_densification_stats(void* stream_, int N, const int32_t* radii, const uint8_t* visibility_opt,
                                        const float* viewspace_grad, int grad_stride, float* max_radii2D_opt,
                                        float* xyz_gradient_accum, float* denom) {
    FDGS_REQUIRE(N >= 0 && grad_stride >= 2, "bad sizes");
    if (N == 0) return FDGS_OK;
    FDGS_REQUIRE(viewspace_grad && xyz_gradient_accum && denom && (radii || visibility_opt), "NULL pointer");
    FDGS_REQUIRE(!max_radii2D_opt || radii, "max_radii2D needs radii");
    hipStream_t stream = (hipStream_t)stream_;
    { FDGS_TIMED("densification_stats", stream);
      hipLaunchKernelGGL(densification_stats_kernel, dim3(cdiv(N, 256)), dim3(256), 0, stream, N, radii, visibility_opt,
                         viewspace_grad, grad_stride, max_radii2D_opt, xyz_gradient_accum, denom); }
    FDGS_LAUNCH_CHECK("densification_stats", 0, stream);
    return FDGS_OK;
}

extern "C" int fdgs_densify_scratch_bytes(int N, size_t* bytes) {
    FDGS_REQUIRE(N >= 0 && bytes, "bad arguments");
    *bytes = densify_scratch(nullptr, N).bytes;
    return FDGS_OK;
}

extern "C" int fdgs_densify_plan(void* stream_, int mode, int N, const float* xyz_gradient_accum, const float* denom,
                                 const float* scaling, const float* opacity, const float* max_radii2D, const uint8_t* drop_mask,
                                 float grad_threshold, float dense_size, float min_opacity, float max_screen_size,
                                 float max_world_size, void* scratch, uint32_t* counts_host) {
    FDGS_REQUIRE(N >= 0 && counts_host, "bad arguments");
    counts_host[0] = counts_host[1] = counts_host[2] = 0;
    if (N == 0) return FDGS_OK;
    FDGS_REQUIRE(scratch, "NULL scratch");
    if (mode == FDGS_PLAN_DENSIFY) FDGS_REQUIRE(xyz_gradient_accum && denom && scaling, "densify plan needs accum, denom, scaling");
    else if (mode == FDGS_PLAN_PRUNE) FDGS_REQUIRE(opacity && (max_screen_size <= 0.f || (max_radii2D && scaling)), "prune plan needs opacity (and max_radii2D, scaling with a screen-size limit)");
    else if (mode == FDGS_PLAN_MASK) FDGS_REQUIRE(drop_mask, "mask plan needs drop_mask");
    else return fail(FDGS_E_INVALID, "%s", "unknown plan mode");
    hipStream_t stream = (hipStream_t)stream_;
    const DensifyScratch s = densify_scratch(scratch, N);
    PlanArgs a{};
    a.mode = mode; a.N = N; a.accum = xyz_gradient_accum; a.denom = denom; a.scaling = scaling; a.opacity = opacity;
    a.max_radii = max_radii2D; a.drop_mask = drop_mask; a.grad_threshold = grad_threshold; a.dense_size = dense_size;
    a.min_opacity = min_opacity; a.max_screen_size = max_screen_size; a.max_world_size = max_world_size;
    a.flags = s.flags; a.blockcnt = s.blockcnt;
    { FDGS_TIMED("densify_plan", stream);
      hipLaunchKernelGGL(densify_classify_kernel, dim3(s.nb), dim3(256), 0, stream, a);
      hipLaunchKernelGGL(densify_scan_kernel, dim3(1), dim3(256), 0, stream, s.nb, s.blockcnt, s.blockoff, s.totals); }
    FDGS_LAUNCH_CHECK("densify_plan", 0, stream);
    uint32_t h[4];
    FDGS_HIP_CHECK(hipMemcpyAsync(h, s.totals, sizeof(h), hipMemcpyDeviceToHost, stream));   // the one readback
    FDGS_HIP_CHECK(hipStreamSynchronize(stream));
    counts_host[0] = h[0]; counts_host[1] = h[1]; counts_host[2] = h[2];
    return FDGS_OK;
}

extern "C" int fdgs_densify_apply(void* stream_, const fdgs_gaussians_in* in, const fdgs_gaussians_out* out, const void* scratch,
                                  const float* split_samples_opt) {
    FDGS_REQUIRE(in && out, "NULL pointer");
    if (in->N == 0) return FDGS_OK;
    FDGS_REQUIRE(in->N > 0 && scratch, "bad arguments");
    for (int g = 0; g < FDGS_NGROUPS; g++) {
        FDGS_REQUIRE(in->width[g] >= 0, "negative row width");
        if (in->width[g] == 0) continue;
        FDGS_REQUIRE(in->param[g] && out->param[g], "NULL parameter array");
        FDGS_REQUIRE((out->exp_avg[g] != nullptr) == (out->exp_avg_sq[g] != nullptr), "exp_avg / exp_avg_sq outputs come in pairs");
    }
    FDGS_REQUIRE(in->width[0] == 3 && in->width[4] == 3 && in->width[5] == 4, "xyz / scaling / rotation rows must be 3 / 3 / 4 floats");
    hipStream_t stream = (hipStream_t)stream_;
    const DensifyScratch s = densify_scratch(const_cast<void*>(scratch), in->N);
    ApplyArgs a{};
    a.in = *in; a.out = *out; a.flags = s.flags; a.blockoff = s.blockoff; a.totals = s.totals; a.samples = split_samples_opt;
    { FDGS_TIMED("densify_apply", stream); hipLaunchKernelGGL(densify_apply_kernel, dim3(s.nb), dim3(256), 0, stream, a); }
    FDGS_LAUNCH_CHECK("densify_apply", 0, stream);
    return FDGS_OK;
}

// ---- row permutation of per-Gaussian arrays (round 6: the library keeps the spatial order for itself) ------------------------------------
// out[i] = in[perm[i]] (gather) or out[perm[i]] = in[i] (scatter) for up to FDGS_MAX_ROW_ARRAYS arrays of 4-byte elements that share the
// row index, in ONE launch (62 floats per row for the six Gaussian parameters + the screen-space sink).  render() reads an unordered model's parameters through the cached Hilbert permutation of its positions and hands
// the per-Gaussian gradients back through the inverse -- 2 x 236 bytes per Gaussian and direction, HBM-bound.
namespace fdgs {
struct PermuteArgs {
    int N, narrays, total, scatter;
    const int32_t* perm;
    const uint32_t* src[FDGS_MAX_ROW_ARRAYS];
    uint32_t* dst[FDGS_MAX_ROW_ARRAYS];
    int width[FDGS_MAX_ROW_ARRAYS], first[FDGS_MAX_ROW_ARRAYS];
};
constexpr int PERM_ROWS = 64;      // rows per workgroup
// Per array, the workgroup's 64 rows are one contiguous block on the SEQUENTIAL side (element t of the block = row t / w, column t % w):
// that side is read / written fully coalesced, the permuted side in runs of one row's w elements.
__global__ void __launch_bounds__(256) permute_rows_kernel(PermuteArgs a) {
    __shared__ long long prm[PERM_ROWS];
    const long long row0 = (long long)blockIdx.x * PERM_ROWS;
    const int nrows = (int)(a.N - row0 < PERM_ROWS ? a.N - row0 : PERM_ROWS);
    if (threadIdx.x < nrows) prm[threadIdx.x] = a.perm[row0 + threadIdx.x];
    __syncthreads();
    for (int k = 0; k < a.narrays; k++) {
        const int w = a.width[k], total = nrows * w;
        const uint32_t* __restrict__ src = a.src[k];
        uint32_t* __restrict__ dst = a.dst[k];
        for (int t = threadIdx.x; t < total; t += 256) {
            const int r = t / w, c = t - r * w;
            const long long seq = row0 * w + t, rnd = prm[r] * w + c;
            if (a.scatter) dst[rnd] = src[seq]; else dst[seq] = src[rnd];
        }
    }
}
}  // namespace fdgs

extern "C" int fdgs_permute_rows(void* stream_, int N, const int32_t* perm, int narrays, const fdgs_row_array* arrays, int scatter) {
    FDGS_REQUIRE(N >= 0 && narrays >= 0 && narrays <= FDGS_MAX_ROW_ARRAYS, "bad sizes");
    if (N == 0 || narrays == 0) return FDGS_OK;
    FDGS_REQUIRE(perm && arrays, "NULL pointer");
    PermuteArgs a{};
    a.N = N; a.narrays = narrays; a.scatter = scatter ? 1 : 0; a.perm = perm;
    int total = 0;
    for (int k = 0; k < narrays; k++) {
        FDGS_REQUIRE(arrays[k].src && arrays[k].dst && arrays[k].width > 0, "row array needs src, dst and a positive width");
        a.src[k] = reinterpret_cast<const uint32_t*>(arrays[k].src); a.dst[k] = reinterpret_cast<uint32_t*>(arrays[k].dst);
        a.width[k] = arrays[k].width; a.first[k] = total;
        total += arrays[k].width;
    }
    a.total = total;
    hipStream_t stream = (hipStream_t)stream_;
    { FDGS_TIMED("permute_rows", stream); hipLaunchKernelGGL(permute_rows_kernel, dim3(cdiv(N, PERM_ROWS)), dim3(256), 0, stream, a); }
    FDGS_LAUNCH_CHECK("permute_rows", 0, stream);
    return FDGS_OK;
}
