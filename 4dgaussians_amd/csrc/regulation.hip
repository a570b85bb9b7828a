// regulation.hip -- fused HexPlane regulariser (forward value + gradient in one launch over all 6*L planes).
//
// Replaces GaussianModel.compute_regulation of the reference (scene/gaussian_model.py:538-577, scene/regulation.py:22-28),
// which train.py:208-211 evaluates every fine iteration as ~10 small PyTorch kernels per plane plus their autograd
// backward (6*L planes):
//     loss = plane_tv_weight       * sum_{k in (0,1,3)} smooth(plane_k)            (spatial planes)
//          + time_smoothness_weight* sum_{k in (2,4,5)} smooth(plane_k)            (planes with a time axis)
//          + l1_time_planes_weight * sum_{k in (2,4,5)} mean |1 - plane_k|
//     smooth(p) = mean over [C, H-2, W] of (p[h+2] - 2 p[h+1] + p[h])^2            (second difference along dim 2 = H)
// Planes live channels-last ([H][W][C], as the deformation kernels want them), so a row h is one contiguous run of W*C
// floats and the stencil along H is five fully coalesced float4 streams; the gradient
//     d smooth / d p[h] = 2/count * (sd[h-2] - 2 sd[h-1] + sd[h])
// comes from the same five rows.  HBM-bound: 2 x plane bytes (read + gradient RMW) per launch.
#include "common.h"

namespace fdgs {

constexpr int REG_MAX_PLANES = FDGS_MAX_LEVELS * 6;
struct RegPlane {
    const float* p; float* g;
    int H, RS4;            // rows, row stride in float4 (W*C/4)
    float ws, wl;          // w_smooth / count, w_l1 / numel
    int first_block;       // first workgroup of this plane
};
struct RegArgs {
    RegPlane pl[REG_MAX_PLANES];
    int nplanes;
    float grad_scale; const float* grad_scale_dev;   // gradient written = grad_scale * (*grad_scale_dev) * dloss/dp
    float* loss;
};

__device__ __forceinline__ float4 sd4(const float4& a, const float4& b, const float4& c) {   // a - 2b + c
    return make_float4(a.x - 2.f * b.x + c.x, a.y - 2.f * b.y + c.y, a.z - 2.f * b.z + c.z, a.w - 2.f * b.w + c.w);
}

__global__ void __launch_bounds__(256) plane_regulation_kernel(RegArgs a) {
    int j = 0;
#pragma unroll
    for (int q = 1; q < REG_MAX_PLANES; q++)
        if (q < a.nplanes && (int)blockIdx.x >= a.pl[q].first_block) j = q;
    const RegPlane P = a.pl[j];
    const int n4 = P.H * P.RS4;
    const int e = ((int)blockIdx.x - P.first_block) * 256 + threadIdx.x;   // float4 index inside the plane
    float part = 0.f;
    if (e < n4) {
        const int h = e / P.RS4;
        const float4* p4 = reinterpret_cast<const float4*>(P.p);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 c0 = p4[e];
        const float4 m1 = h >= 1 ? p4[e - P.RS4] : z, m2 = h >= 2 ? p4[e - 2 * P.RS4] : z;
        const float4 q1 = h + 1 < P.H ? p4[e + P.RS4] : z, q2 = h + 2 < P.H ? p4[e + 2 * P.RS4] : z;
        // second differences that involve row h: sd[h-2] (rows h-2..h), sd[h-1] (h-1..h+1), sd[h] (h..h+2)
        const bool v0 = h >= 2, v1 = h >= 1 && h + 1 < P.H, v2 = h + 2 < P.H;
        const float4 s0 = v0 ? sd4(m2, m1, c0) : z, s1 = v1 ? sd4(m1, c0, q1) : z, s2 = v2 ? sd4(c0, q1, q2) : z;
        // loss: every sd[h] is counted once, by the thread of its first row
        part = P.ws * (s2.x * s2.x + s2.y * s2.y + s2.z * s2.z + s2.w * s2.w);
        if (P.wl != 0.f) part += P.wl * (fabsf(1.f - c0.x) + fabsf(1.f - c0.y) + fabsf(1.f - c0.z) + fabsf(1.f - c0.w));
        if (P.g) {
            const float gs = a.grad_scale * (a.grad_scale_dev ? *a.grad_scale_dev : 1.f);
            const float k2 = 2.f * P.ws * gs, kl = P.wl * gs;
            auto sgn = [](float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); };   // d|1-p|/dp = -sign(1-p)
            float4* g4 = reinterpret_cast<float4*>(P.g);
            float4 g = g4[e];
            g.x += k2 * (s0.x - 2.f * s1.x + s2.x) - kl * sgn(1.f - c0.x);
            g.y += k2 * (s0.y - 2.f * s1.y + s2.y) - kl * sgn(1.f - c0.y);
            g.z += k2 * (s0.z - 2.f * s1.z + s2.z) - kl * sgn(1.f - c0.z);
            g.w += k2 * (s0.w - 2.f * s1.w + s2.w) - kl * sgn(1.f - c0.w);
            g4[e] = g;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0 && a.loss) atomicAdd(a.loss, ws[0] + ws[1] + ws[2] + ws[3]);
}

}  // namespace fdgs

using namespace fdgs;

extern "C" int fdgs_plane_regulation(void* stream_, int nplanes, const fdgs_reg_plane* planes, float grad_scale,
                                     const float* grad_scale_dev_opt, float* loss_acc_opt) {
    FDGS_REQUIRE(nplanes >= 0 && nplanes <= REG_MAX_PLANES && (planes || nplanes == 0), "bad plane list");
    if (nplanes == 0) return FDGS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    RegArgs a{};
    int blocks = 0;
    for (int i = 0; i < nplanes; i++) {
        const fdgs_reg_plane& s = planes[i];
        FDGS_REQUIRE(s.plane && s.H >= 1 && s.W >= 1 && s.C >= 4 && s.C % 4 == 0, "plane must be channels-last [H][W][C], C % 4 == 0");
        FDGS_REQUIRE((long long)s.H * s.W * s.C < (1ll << 31), "plane too large");
        FDGS_REQUIRE((((uintptr_t)s.plane | (uintptr_t)s.grad_opt) & 15) == 0, "plane pointers must be 16-byte aligned");
        RegPlane& P = a.pl[i];
        P.p = s.plane; P.g = s.grad_opt; P.H = s.H; P.RS4 = s.W * s.C / 4;
        // mean over [C, H-2, W]; torch's mean of an empty tensor is NaN (H < 3): same here
        const double cnt = (double)s.C * (s.H - 2) * s.W;
        P.ws = s.w_smooth != 0.f ? (float)(s.w_smooth / cnt) : 0.f;
        if (s.w_smooth != 0.f && s.H < 3) P.ws = __builtin_nanf("");
        P.wl = s.w_l1 != 0.f ? (float)(s.w_l1 / ((double)s.C * s.H * s.W)) : 0.f;
        P.first_block = blocks;
        blocks += cdiv((long long)s.H * P.RS4, 256);
    }
    a.nplanes = nplanes; a.grad_scale = grad_scale; a.grad_scale_dev = grad_scale_dev_opt; a.loss = loss_acc_opt;
    { FDGS_TIMED("plane_regulation", stream); hipLaunchKernelGGL(plane_regulation_kernel, dim3(blocks), dim3(256), 0, stream, a); }
    FDGS_LAUNCH_CHECK("plane_regulation", 0, stream);
    return FDGS_OK;
}
