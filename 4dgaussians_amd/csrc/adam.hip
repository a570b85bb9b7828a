// adam.hip -- multi-tensor Adam step (one launch for up to 64 parameter tensors).
//
// Replaces `gaussians.optimizer.step()` (train.py:291) = torch.optim.Adam(l, lr=0.0, eps=1e-15) over the eight per-Gaussian
// groups plus the deformation MLP / HexPlane groups (scene/gaussian_model.py:165-196): betas (0.9, 0.999), no weight decay,
// no amsgrad, per-group lr set every iteration by update_learning_rate (:198-212).  Arithmetic follows torch's
// single-tensor Adam:
//     m = m + (g - m) * (1 - b1)                   (lerp)
//     v = b2 * v + (1 - b2) * g * g
//     p = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// HBM-bound streaming kernel: 28 B per parameter element (p, m, v read+write, g read), float4 accesses, all tensors of a
// step in one launch (the per-tensor launch + ~5 elementwise passes of the foreach path cost more than the math at 200 fps).
#include "common.h"

#include <math.h>

namespace fdgs {

constexpr int ADAM_MAX_TENSORS = 64;
struct AdamT {
    float* p; const float* g; float* m; float* v;
    unsigned int n;
    float step_size;      // lr / (1 - b1^t)
    float sqrt_bc2;       // sqrt(1 - b2^t)
    int first_block;
};
struct AdamArgs {
    AdamT t[ADAM_MAX_TENSORS];
    int ntensors;
    float b1, b2, eps;
    float omb1, omb2;   // 1 - beta computed in double on the host like torch (1 - 0.999f in float is 2e-6 off)
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamT& T, const AdamArgs& a) {
    const float eps = a.eps;
    m = m + (g - m) * a.omb1;
    v = a.b2 * v + a.omb2 * g * g;
    const float denom = sqrtf(v) / T.sqrt_bc2 + eps;
    p = p - T.step_size * (m / denom);
}

__global__ void __launch_bounds__(256) adam_kernel(AdamArgs a) {
    int lo = 0, hi = a.ntensors - 1;   // last tensor whose first_block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.t[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AdamT T = a.t[lo];
    const unsigned int e4 = ((unsigned int)blockIdx.x - (unsigned int)T.first_block) * 256u + threadIdx.x;
    const unsigned int n4 = T.n >> 2;
    if (e4 < n4) {
        float4 p = reinterpret_cast<float4*>(T.p)[e4], m = reinterpret_cast<float4*>(T.m)[e4], v = reinterpret_cast<float4*>(T.v)[e4];
        const float4 g = reinterpret_cast<const float4*>(T.g)[e4];
        adam1(p.x, g.x, m.x, v.x, T, a); adam1(p.y, g.y, m.y, v.y, T, a);
        adam1(p.z, g.z, m.z, v.z, T, a); adam1(p.w, g.w, m.w, v.w, T, a);
        reinterpret_cast<float4*>(T.p)[e4] = p; reinterpret_cast<float4*>(T.m)[e4] = m; reinterpret_cast<float4*>(T.v)[e4] = v;
    } else if (e4 == n4) {   // scalar tail (n % 4 elements) by one thread
        for (unsigned int i = n4 * 4; i < T.n; i++) adam1(T.p[i], T.g[i], T.m[i], T.v[i], T, a);
    }
}

}  // namespace fdgs

using namespace fdgs;

extern "C" int fdgs_adam_step(void* stream_, int ntensors, const fdgs_adam_tensor* tensors, double beta1, double beta2, double eps) {
    FDGS_REQUIRE(ntensors >= 0 && (tensors || ntensors == 0), "bad tensor list");
    hipStream_t stream = (hipStream_t)stream_;
    int i = 0;   // next input tensor: empty tensors are skipped without using a slot, so the consumed index is tracked
    while (i < ntensors) {
        AdamArgs a{};
        int blocks = 0, cnt = 0;
        for (; i < ntensors && cnt < ADAM_MAX_TENSORS; i++) {
            const fdgs_adam_tensor& s = tensors[i];
            if (s.n == 0) continue;
            FDGS_REQUIRE(s.param && s.grad && s.exp_avg && s.exp_avg_sq, "adam: NULL tensor pointer");
            FDGS_REQUIRE(s.step >= 1, "adam: step counts from 1");
            FDGS_REQUIRE((((uintptr_t)s.param | (uintptr_t)s.grad | (uintptr_t)s.exp_avg | (uintptr_t)s.exp_avg_sq) & 15) == 0,
                         "adam: tensors must be 16-byte aligned");
            FDGS_REQUIRE(s.n < (1ull << 32), "adam: tensor too large");
            AdamT& T = a.t[cnt++];
            T.p = s.param; T.g = s.grad; T.m = s.exp_avg; T.v = s.exp_avg_sq; T.n = (unsigned int)s.n;
            const double bc1 = 1.0 - pow(beta1, (double)s.step), bc2 = 1.0 - pow(beta2, (double)s.step);
            T.step_size = (float)((double)s.lr / bc1);
            T.sqrt_bc2 = (float)sqrt(bc2);
            T.first_block = blocks;
            blocks += cdiv((long long)(s.n / 4) + 1, 256);
        }
        if (cnt == 0) continue;
        a.ntensors = cnt; a.b1 = (float)beta1; a.b2 = (float)beta2; a.eps = (float)eps;
        a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
        { FDGS_TIMED("adam_step", stream); hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, a); }
        FDGS_LAUNCH_CHECK("adam_step", 0, stream);
    }
    return FDGS_OK;
}
