// glds_probe.hip -- LDS-DMA (global_load_lds_dwordx4) as the weight-stationary D2 uses it: a wave copies a [16][32]-float slice of a row-major
// matrix through a LIST of row indices into its own LDS park, lane-linear destination, chunk swizzle on the SOURCE address; the wave then
// reads the park back in both layouts (natural float4, transposed dword) after a counted s_waitcnt and checks every element.
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/glds_probe.hip -o /tmp/glds_probe && /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ void __launch_bounds__(256, 1) probe(const float* X, const unsigned* rows, int W, int ntiles, unsigned* bad, float* sink) {
    __shared__ __attribute__((aligned(16))) float park[4][16 * 32];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, q = lane >> 4;
    const unsigned park_addr = (unsigned)(size_t)(&park[w][0]);      // LDS byte address (low 32 bits of the generic pointer)
    unsigned nbad = 0;
    float acc = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int slot = 64 * e + lane, g = slot >> 3, c8 = (slot & 7) ^ (g & 7);
            const unsigned r = rows[tile * 16 + g];
            glds16(X + (size_t)r * W + 32 * w + 4 * c8, park_addr + 1024u * e);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // natural: lane (n = g, q), tile t: features 16 t + 4 q .. + 3
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int c8 = 4 * t + q;
            const float4 v = *reinterpret_cast<const float4*>(&park[w][4 * (8 * n + (c8 ^ (n & 7)))]);
            const float* ref = X + (size_t)rows[tile * 16 + n] * W + 32 * w + 16 * t + 4 * q;
            nbad += (v.x != ref[0]) + (v.y != ref[1]) + (v.z != ref[2]) + (v.w != ref[3]);
            acc += v.x;
        }
        // transposed: lane (n, q), c: element (g = 4 q + c, f = 16 t + n)
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int g = 4 * q + c, c8 = 4 * t + (n >> 2);
                const float v = park[w][4 * (8 * g + (c8 ^ (g & 7))) + (n & 3)];
                nbad += v != X[(size_t)rows[tile * 16 + g] * W + 32 * w + 16 * t + n];
                acc += v;
            }
    }
    if (nbad) atomicAdd(bad, nbad);
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const int W = 128, N = 4096, ntiles = 1024;
    std::vector<float> hx((size_t)N * W);
    for (size_t i = 0; i < hx.size(); i++) hx[i] = (float)(i % 9973) * 0.25f;
    std::vector<unsigned> hr((size_t)ntiles * 16);
    for (size_t i = 0; i < hr.size(); i++) hr[i] = (unsigned)((i * 2654435761u) % N);
    float *dx, *sink; unsigned *dr, *bad;
    hipMalloc(&dx, hx.size() * 4); hipMalloc(&dr, hr.size() * 4); hipMalloc(&bad, 4); hipMalloc(&sink, 4);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(probe, dim3(64), dim3(256), 0, 0, dx, dr, W, ntiles, bad, sink);
    unsigned hb = 1;
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("glds probe: %u mismatching elements (%s), last error %s\n", hb, hb ? "FAIL" : "ok", hipGetErrorString(hipGetLastError()));
    return hb != 0;
}
