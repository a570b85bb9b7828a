// tools/atomic_bench.hip -- micro-benchmark of global float atomicAdd throughput on MI355X (design input for the
// HexPlane-gradient scatter and the per-(tile,Gaussian) gradient atomics). Build: hipcc --offload-arch=gfx950 -O3
// -munsafe-fp-atomics tools/atomic_bench.hip -o tools/atomic_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode 0: a[i] += 1 (distinct, coalesced). mode 1: random cell of `cells`, 16 consecutive floats (plane-like).
// mode 2: random single float among `cells` addresses (contended scalars). mode 3: like 1 but only 2 "rows" (time planes)
__global__ void k(float* a, uint32_t n, uint32_t cells, int mode, int reps) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int r = 0; r < reps; r++) {
        uint32_t h = hash(i * 9781u + r * 6271u + 17u);
        if (mode == 0) atomicAdd(&a[i], 1.0f);
        else if (mode == 1 || mode == 3) {
            uint32_t c = h % cells;
#pragma unroll
            for (int j = 0; j < 16; j++) atomicAdd(&a[(size_t)c * 16 + j], 1.0f);
        } else atomicAdd(&a[h % cells], 1.0f);
    }
}

int main() {
    const uint32_t n = 1u << 22;
    float* a; CK(hipMalloc(&a, (size_t)n * 16 * 4)); CK(hipMemset(a, 0, (size_t)n * 16 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { int mode; uint32_t cells; const char* name; int per; } cfg[] = {
        {0, 0, "distinct coalesced", 1}, {1, 4096, "16-float cell, 4096 cells (64x64 plane)", 16},
        {1, 16384, "16-float cell, 16384 cells (128x128 plane)", 16}, {1, 128, "16-float cell, 128 cells (time-plane rows)", 16},
        {2, 1u << 20, "random scalar, 1M addrs", 1}, {2, 300000 * 10, "random scalar, 3M addrs (per-Gaussian grads)", 1},
        {2, 2048, "random scalar, 2048 addrs", 1}, {2, 1, "single address", 1}};
    for (auto& c : cfg) {
        int reps = 4;
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, n, c.cells ? c.cells : 1, c.mode, 1);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, n, c.cells ? c.cells : 1, c.mode, reps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double atoms = (double)n * reps * c.per;
        printf("%-52s %8.3f ms  %8.2f G float-atomics/s\n", c.name, ms, atoms / ms * 1e-6);
    }
    return 0;
}
