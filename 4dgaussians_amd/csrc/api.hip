// api.hip -- library-wide entry points of libfdgs: error string, version, device query, L1 statistics kernel.
#include "common.h"

namespace fdgs {
thread_local char g_err[512] = {0};

// [sum |a-b|, sum (a-b)^2, n] with optional dL/da = sign(a-b)*scale (utils/loss_utils.py:20-21 of the reference)
__global__ void __launch_bounds__(256) l1_stats_kernel(size_t n, const float* __restrict__ a, const float* __restrict__ b,
                                                       float scale, float* __restrict__ grad, float* __restrict__ acc) {
    float s1 = 0.f, s2 = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = a[i] - b[i];
        s1 += fabsf(d); s2 += d * d;
        if (grad) grad[i] = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    __shared__ float w1[4], w2[4];
    if ((threadIdx.x & 63) == 0) { w1[threadIdx.x >> 6] = s1; w2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&acc[0], w1[0] + w1[1] + w1[2] + w1[3]);
        atomicAdd(&acc[1], w2[0] + w2[1] + w2[2] + w2[3]);
        if (blockIdx.x == 0) atomicAdd(&acc[2], (float)n);
    }
}
}  // namespace fdgs

using namespace fdgs;

extern "C" const char* fdgs_last_error(void) { return g_err; }
extern "C" int fdgs_abi_version(void) { return 1; }

extern "C" int fdgs_device_arch(int dev, char* buf, size_t buflen) {
    FDGS_REQUIRE(buf && buflen > 0, "bad arguments");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || dev < 0 || dev >= n) return fail(FDGS_E_NOGPU, "%s", "no such HIP device");
    hipDeviceProp_t prop;
    FDGS_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, buflen, "%s", prop.gcnArchName);
    return FDGS_OK;
}

extern "C" int fdgs_l1_stats(void* stream_, size_t n, const float* a, const float* b, float grad_scale, float* grad_out_opt,
                             float* acc) {
    FDGS_REQUIRE(a && b && acc, "NULL pointer");
    if (n == 0) return FDGS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(l1_stats_kernel, dim3(blocks), dim3(256), 0, stream, n, a, b, grad_scale, grad_out_opt, acc);
    FDGS_LAUNCH_CHECK("l1_stats", 0, stream);
    return FDGS_OK;
}
