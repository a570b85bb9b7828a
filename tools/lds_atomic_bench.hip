// tools/lds_atomic_bench.hip -- LDS atomic throughput on MI355X: ds_add_f32 vs ds_add_u32 vs ds_add_u64 vs plain
// read-add-write, 64 distinct consecutive addresses per wave instruction (the access shape of the plane-gradient and
// dW2 accumulators).  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o tools/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int reps) {
    __shared__ unsigned long long buf[4096];
    float* f = reinterpret_cast<float*>(buf);
    unsigned int* u = reinterpret_cast<unsigned int*>(buf);
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < reps; r++) {
        const int base = ((r * 37 + wave * 11) & 31) * 64;   // 64 consecutive slots, moving window
        if (MODE == 0) atomicAdd(&f[base + lane], 1.0f);
        else if (MODE == 1) atomicAdd(&u[base + lane], 1u);
        else if (MODE == 2) atomicAdd(&buf[base + lane], 1ull);
        else { f[4096 + wave * 512 + ((r * 37) & 7) * 64 + lane] += 1.0f; }   // plain RMW on a wave-private region
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = f[0] + f[5000];
}

int main() {
    float* out; CK(hipMalloc(&out, 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 4096, blocks = 1024;
    const char* names[4] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain ds_read+add+ds_write (wave-private)"};
    for (int m = 0; m < 4; m++) {
        for (int it = 0; it < 2; it++) {
            CK(hipEventRecord(e0));
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, reps);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, reps);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, reps);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, reps);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double lanes = (double)blocks * 256 * reps;
        printf("%-44s %8.3f ms  %8.1f G lane-ops/s  (%.2f lanes/clk/CU at 2.4 GHz, 256 CUs)\n", names[m], ms, lanes / ms * 1e-6,
               lanes / (ms * 1e-3) / 256 / 2.4e9);
    }
    return 0;
}
