// tools/mfma_chain_probe.hip -- issue rate of f32 MFMAs as a function of how many independent accumulators the stream alternates
// between (1 = every MFMA reads the accumulator the previous one wrote) and of the waves per SIMD, bare and with an operand stream
// (one 16-byte load per 4 MFMAs, requested PD steps ahead, L1/L2-resident) -- the shape of one output tile of a hidden layer
// evaluated on its own ("group-wise").  Development probe for the forms of deform_fwd.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain_probe tools/mfma_chain_probe.hip && /tmp/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void __launch_bounds__(512) bare32(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) acc[u % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % CH], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CH; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int CH>
__global__ void __launch_bounds__(512) bare16(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; c++) for (int r = 0; r < 4; r++) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) acc[u % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % CH], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CH; c++) for (int r = 0; r < 4; r++) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// One output tile of a 128 x 128 layer for 32 Gaussians: 16 k-walk steps, each ONE 16-byte operand load (row = lane & 31, 512-byte row
// stride, or a packed 1-KB stream) feeding four MFMAs that accumulate into CH accumulators; the load of step s + PD is requested at
// step s.  `tiles` output tiles back to back (the operand addresses walk a 64-KB matrix: L1/L2 resident).  The loads and their counted
// waits are inline assembly: hipcc regroups plain loads four at a time behind vmcnt(0) (the probe is about the hardware).
typedef float v4f __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void ld16(v4f& dst, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(dst) : "v"(p), "n"(OFF)); }
template <int N>
__device__ __forceinline__ void wait_vm(v4f& v) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N)); }
constexpr int koff(int s, bool packed) { return packed ? s * 1024 : 16 * ((s & 3) + 8 * (s >> 2)); }

template <int CH, int PD, bool PACKED>
__global__ void __launch_bounds__(512) stream32(const float* __restrict__ w, float* out, unsigned long long* cyc, int tiles) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    f32x16 x[4];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) x[t][r] = 1.0f + 1e-3f * (threadIdx.x + 7 * r + t);
    const int lane = threadIdx.x & 63, g = lane & 31, h = lane >> 5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int tl = 0; tl < tiles; tl++) {
        const int ot = tl & 3;
        const float* rp = PACKED ? w + (size_t)ot * 16 * 256 + lane * 4 : w + (size_t)(32 * ot + g) * 128 + 16 * h;
        v4f buf[PD];
#define LD(S) ld16<koff(S, PACKED) % 4096>(buf[(S) % PD], rp + (koff(S, PACKED) / 4096) * 1024)
        LD(0); if (PD > 1) LD(1); if (PD > 2) LD(2); if (PD > 3) LD(3);
#define STEP(S) { if ((S) + PD <= 16) wait_vm<PD - 1>(buf[(S) % PD]); else wait_vm<0>(buf[(S) % PD]); \
            const v4f cur = buf[(S) % PD]; \
            acc[(4 * (S) + 0) % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.x, x[0][S], acc[(4 * (S) + 0) % CH], 0, 0, 0); \
            acc[(4 * (S) + 1) % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.y, x[1][S], acc[(4 * (S) + 1) % CH], 0, 0, 0); \
            acc[(4 * (S) + 2) % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.z, x[2][S], acc[(4 * (S) + 2) % CH], 0, 0, 0); \
            acc[(4 * (S) + 3) % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.w, x[3][S], acc[(4 * (S) + 3) % CH], 0, 0, 0); \
            asm volatile("" :: "v"(cur)); \
            if ((S) + PD < 16) LD((S) + PD); }
        STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
#undef LD
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CH; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// the same loop with the operand stream (1) read from LDS with ds_read_b128 (lgkmcnt waits) or (2) requested from memory but never waited
// for (the MFMAs use fixed registers): separates "the vector-memory path returns late" from "issuing the requests costs MFMA issue slots"
template <int MODE, int PD>
__global__ void __launch_bounds__(512) stream32b(const float* __restrict__ w, float* out, unsigned long long* cyc, int tiles) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 16 * 256];
    for (int i = threadIdx.x; i < 4 * 16 * 256; i += blockDim.x) lds[i] = w[i];
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    f32x16 x[4];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) x[t][r] = 1.0f + 1e-3f * (threadIdx.x + 7 * r + t);
    const int lane = threadIdx.x & 63;
    v4f fixed = {1.f + lane, 2.f, 3.f, 4.f};
    for (int tl = 0; tl < tiles; tl++) {
        const int ot = tl & 3;
        const float* rp = w + (size_t)ot * 16 * 256 + lane * 4;
        const unsigned la = (unsigned)(size_t)(lds + ot * 16 * 256 + lane * 4);
        v4f buf[PD];
#define LDG(S) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(buf[(S) % PD]) : "v"(rp + ((S) / 4) * 1024), "n"(((S) % 4) * 1024))
#define LDL(S) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(buf[(S) % PD]) : "v"(la), "n"((S) * 1024))
#define LD(S) do { if (MODE == 1) LDL(S); else LDG(S); } while (0)
        LD(0); if (PD > 1) LD(1); if (PD > 2) LD(2); if (PD > 3) LD(3);
#define STEP(S) { if (MODE == 1) { if ((S) + PD <= 16) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(buf[(S) % PD]) : "n"(PD - 1)); else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(buf[(S) % PD])); } \
            else if (MODE == 0) { if ((S) + PD <= 16) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(buf[(S) % PD]) : "n"(PD - 1)); else asm volatile("s_waitcnt vmcnt(0)" : "+v"(buf[(S) % PD])); } \
            const v4f cur = MODE == 2 ? fixed : buf[(S) % PD]; \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.x, x[0][S], acc, 0, 0, 0); \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.y, x[1][S], acc, 0, 0, 0); \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.z, x[2][S], acc, 0, 0, 0); \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.w, x[3][S], acc, 0, 0, 0); \
            asm volatile("" :: "v"(cur)); \
            if ((S) + PD < 16) LD((S) + PD); }
        STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
        if (MODE == 2) { asm volatile("s_waitcnt vmcnt(0)"); for (int q = 0; q < PD; q++) asm volatile("" :: "v"(buf[q])); }
#undef STEP
#undef LD
    }
    float s = 0.f;
    for (int r = 0; r < 16; r++) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int iters = 2000;
    float* w; hipMalloc(&w, 128 * 128 * 4 + 4096); hipMemset(w, 0, 128 * 128 * 4 + 4096);
#define BARE(K, CH, TH, CYC) { float* out; unsigned long long* cyc; hipMalloc(&out, 256 * TH * 4); hipMalloc(&cyc, 256 * 8 * 8); \
        hipLaunchKernelGGL((K<CH>), dim3(256), dim3(TH), 0, 0, out, cyc, iters); hipDeviceSynchronize(); \
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); \
        hipLaunchKernelGGL((K<CH>), dim3(256), dim3(TH), 0, 0, out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1); \
        float ms; hipEventElapsedTime(&ms, e0, e1); std::vector<unsigned long long> hh(256 * TH / 64); hipMemcpy(hh.data(), cyc, hh.size() * 8, hipMemcpyDeviceToHost); \
        double sum = 0; for (auto v : hh) sum += (double)v; const double pw = sum / hh.size(), n = 16.0 * iters; \
        printf("%-10s accumulators %d  waves/SIMD %d  ticks/MFMA/wave %7.2f   kernel %.3f ms  wall cycles(2.4GHz)/MFMA/SIMD %6.2f (ideal %d)\n", #K, CH, TH / 256, pw / n, ms, ms * 1e-3 * 2.4e9 / (n * (TH / 256)), CYC); \
        hipFree(out); hipFree(cyc); }
    BARE(bare32, 1, 256, 64) BARE(bare32, 2, 256, 64) BARE(bare32, 4, 256, 64) BARE(bare32, 1, 512, 64) BARE(bare32, 2, 512, 64)
    BARE(bare16, 1, 256, 32) BARE(bare16, 2, 256, 32) BARE(bare16, 4, 256, 32) BARE(bare16, 1, 512, 32) BARE(bare16, 4, 512, 32)
    const int tiles = 800;
#define STREAM(CH, PD, PK, TH) { float* out; unsigned long long* cyc; hipMalloc(&out, 256 * TH * 4); hipMalloc(&cyc, 256 * 8 * 8); \
        hipLaunchKernelGGL((stream32<CH, PD, PK>), dim3(256), dim3(TH), 0, 0, w, out, cyc, tiles); hipDeviceSynchronize(); \
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); \
        hipLaunchKernelGGL((stream32<CH, PD, PK>), dim3(256), dim3(TH), 0, 0, w, out, cyc, tiles); hipEventRecord(e1); hipEventSynchronize(e1); \
        float ms; hipEventElapsedTime(&ms, e0, e1); const double n = 64.0 * tiles; \
        printf("stream32   accumulators %d  PD %d  packed %d  waves/SIMD %d   kernel %.3f ms  wall cycles(2.4GHz)/MFMA/SIMD %6.2f (ideal 64)  pipe use %5.3f\n", CH, PD, (int)PK, TH / 256, ms, \
               ms * 1e-3 * 2.4e9 / (n * (TH / 256)), 64.0 * n * (TH / 256) / (ms * 1e-3 * 2.4e9)); hipFree(out); hipFree(cyc); }
    STREAM(1, 2, false, 256) STREAM(2, 2, false, 256) STREAM(4, 2, false, 256) STREAM(1, 3, false, 256) STREAM(1, 4, false, 256) STREAM(2, 4, false, 256)
    STREAM(1, 2, true, 256) STREAM(1, 4, true, 256)
    STREAM(1, 2, false, 512) STREAM(2, 2, false, 512) STREAM(1, 3, false, 512) STREAM(1, 4, false, 512) STREAM(1, 2, true, 512) STREAM(1, 4, true, 512)
#define STREAMB(MODE, PD, TH) { float* out; unsigned long long* cyc; hipMalloc(&out, 256 * TH * 4); hipMalloc(&cyc, 256 * 8 * 8); \
        hipLaunchKernelGGL((stream32b<MODE, PD>), dim3(256), dim3(TH), 0, 0, w, out, cyc, tiles); hipDeviceSynchronize(); \
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); \
        hipLaunchKernelGGL((stream32b<MODE, PD>), dim3(256), dim3(TH), 0, 0, w, out, cyc, tiles); hipEventRecord(e1); hipEventSynchronize(e1); \
        float ms; hipEventElapsedTime(&ms, e0, e1); const double n = 64.0 * tiles; \
        printf("stream32b  mode %d (0 global packed, 1 LDS, 2 requests issued but not consumed)  PD %d  waves/SIMD %d   kernel %.3f ms  pipe use %5.3f\n", MODE, PD, TH / 256, ms, \
               64.0 * n * (TH / 256) / (ms * 1e-3 * 2.4e9)); hipFree(out); hipFree(cyc); }
    STREAMB(0, 2, 256) STREAMB(0, 2, 512) STREAMB(1, 2, 256) STREAMB(1, 2, 512) STREAMB(2, 2, 256) STREAMB(2, 2, 512) STREAMB(0, 4, 256) STREAMB(0, 4, 512) STREAMB(1, 4, 512)
    BARE(bare32, 1, 256, 64) BARE(bare32, 1, 512, 64)
    return 0;
}
