// tools/ws_probe.hip -- go / no-go probe for a WEIGHT-STATIONARY form of the deformation forward (round 6).
//
// Today's D1 gives a wave 16 Gaussians and streams every weight of every layer past it (374 KB of operands per 16 Gaussians, ~7 GB of
// L2 -> CU requests per launch; each request costs the MFMA pipe an issue slot).  The probe turns that inside out: the EIGHT waves of a
// workgroup (two per SIMD, 256 registers each) hold the five heads' first-layer weights in REGISTERS -- wave w owns rows 16 w .. 16 w + 15 of
// every W1 (5 x 32 registers) -- and the activations of a 16-Gaussian tile visit them through LDS: per tile each wave writes its 16 rows of
// relu(hidden) into an LDS tile, ONE s_barrier, then every wave multiplies its weight rows with the whole tile (32 k-steps per head, B
// operands read from LDS with ds_read_b128).  No operand requests to L2 at all.  Measures cycles per tile against the MFMA issue floor.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ws_probe tools/ws_probe.hip && /tmp/ws_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mm16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int W = 128, LDW = W + 4;

// XREG: keep the X tile in 32 registers (read once per tile) instead of re-reading it from LDS for every head
// bare MFMA loop (two accumulators, two waves per SIMD): what the matrix cores deliver on this box at its sustained clock
__global__ void __launch_bounds__(512) bare_kernel(float* out, int iters) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) { a0 = mm16(a, b, a0); a1 = mm16(b, a, a1); }
    }
    a0 += a1;
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a0[1] + a0[2] + a0[3];
}

// SAVE: 0 none, 1 non-temporal 64-byte pieces straight from the accumulators, 2 plain stores, 3 through an LDS tile as whole 512-byte rows
template <int NH, bool XREG, int SAVE>
__global__ void __launch_bounds__(512) ws_kernel(const float* __restrict__ W1, const float* __restrict__ Xg, float* __restrict__ out,
                                                  float* __restrict__ saved, int ntiles, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float xl[2][16 * LDW];
    __shared__ __attribute__((aligned(16))) float hl[SAVE == 3 ? 2 : 1][SAVE == 3 ? 16 * LDW : 4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    float4 wr[NH][8];
#pragma unroll
    for (int h = 0; h < NH; h++)
#pragma unroll
        for (int j = 0; j < 8; j++) wr[h][j] = *reinterpret_cast<const float4*>(W1 + ((size_t)h * W + 16 * w + n) * W + 16 * j + 4 * q);
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int it = 0, hb = 0;
    float4 vnext = *reinterpret_cast<const float4*>(Xg + ((size_t)blockIdx.x * 16 + n) * W + 16 * w + 4 * q);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, it ^= 1) {
        // "trunk": this wave's 16 features of the tile's 16 Gaussians (requested one tile ahead)
        float4 v = vnext;
        if (tile + (int)gridDim.x < ntiles) vnext = *reinterpret_cast<const float4*>(Xg + ((size_t)(tile + gridDim.x) * 16 + n) * W + 16 * w + 4 * q);
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        float* xt = xl[it];
        *reinterpret_cast<float4*>(xt + n * LDW + 16 * w + 4 * q) = v;
        __syncthreads();
        float4 xr[8];
        if (XREG) {
#pragma unroll
            for (int j = 0; j < 8; j++) xr[j] = *reinterpret_cast<const float4*>(xt + n * LDW + 16 * j + 4 * q);
        }
#pragma unroll
        for (int h = 0; h < NH; h++) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float4 x = XREG ? xr[j] : *reinterpret_cast<const float4*>(xt + n * LDW + 16 * j + 4 * q);
                a0 = mm16(wr[h][j].x, x.x, a0);
                a1 = mm16(wr[h][j].y, x.y, a1);
                a0 = mm16(wr[h][j].z, x.z, a0);
                a1 = mm16(wr[h][j].w, x.w, a1);
            }
            f32x4 y = a0 + a1;
#pragma unroll
            for (int r = 0; r < 4; r++) y[r] = fmaxf(y[r], 0.f);
            float* dst = saved + (((size_t)h * ntiles + tile) * 16) * W;
            if (SAVE == 1) __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(dst + n * W + 16 * w + 4 * q));
            if (SAVE == 2) *reinterpret_cast<f32x4*>(dst + n * W + 16 * w + 4 * q) = y;
            if (SAVE == 3) {
                // whole rows: every wave parks its 16 features of this head in an LDS tile; after the barrier the workgroup copies the tile out,
                // thread t one float4 of the [16][128] tile (512 consecutive bytes per 32 lanes)
                *reinterpret_cast<f32x4*>(&hl[hb][n * LDW + 16 * w + 4 * q]) = y;
                __syncthreads();
                const int row = threadIdx.x >> 5, c4 = threadIdx.x & 31;
                __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(&hl[hb][row * LDW + 4 * c4]), reinterpret_cast<f32x4*>(dst + row * W + 4 * c4));
                hb ^= 1;
            }
            keep += y;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    *reinterpret_cast<f32x4*>(out + ((size_t)blockIdx.x * 512 + threadIdx.x) * 4) = keep;
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}


// FOUR waves per workgroup, one per SIMD, 512 registers each: wave w owns rows 32 w .. 32 w + 31 (two 16-row tiles) of every W1 -- 320
// registers of stationary weights and 192 to work with; a B operand read from LDS feeds both row tiles.
template <int NH, int SAVE>
__global__ void __launch_bounds__(256, 1) ws4_kernel(const float* __restrict__ W1, const float* __restrict__ Xg, float* __restrict__ out,
                                                      float* __restrict__ saved, int ntiles, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float xl[2][16 * LDW];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    float4 wr[NH][2][8];
#pragma unroll
    for (int h = 0; h < NH; h++)
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int j = 0; j < 8; j++) wr[h][t][j] = *reinterpret_cast<const float4*>(W1 + ((size_t)h * W + 32 * w + 16 * t + n) * W + 16 * j + 4 * q);
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int it = 0;
    float4 vnext[2];
#pragma unroll
    for (int t = 0; t < 2; t++) vnext[t] = *reinterpret_cast<const float4*>(Xg + ((size_t)blockIdx.x * 16 + n) * W + 32 * w + 16 * t + 4 * q);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, it ^= 1) {
        float* xt = xl[it];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            float4 v = vnext[t];
            if (tile + (int)gridDim.x < ntiles) vnext[t] = *reinterpret_cast<const float4*>(Xg + ((size_t)(tile + gridDim.x) * 16 + n) * W + 32 * w + 16 * t + 4 * q);
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            *reinterpret_cast<float4*>(xt + n * LDW + 32 * w + 16 * t + 4 * q) = v;
        }
        __syncthreads();
        float4 xr[8];
#pragma unroll
        for (int j = 0; j < 8; j++) xr[j] = *reinterpret_cast<const float4*>(xt + n * LDW + 16 * j + 4 * q);
#pragma unroll
        for (int h = 0; h < NH; h++) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float4 x = xr[j];
                a0 = mm16(wr[h][0][j].x, x.x, a0); b0 = mm16(wr[h][1][j].x, x.x, b0);
                a1 = mm16(wr[h][0][j].y, x.y, a1); b1 = mm16(wr[h][1][j].y, x.y, b1);
                a0 = mm16(wr[h][0][j].z, x.z, a0); b0 = mm16(wr[h][1][j].z, x.z, b0);
                a1 = mm16(wr[h][0][j].w, x.w, a1); b1 = mm16(wr[h][1][j].w, x.w, b1);
            }
            f32x4 y0 = a0 + a1, y1 = b0 + b1;
#pragma unroll
            for (int r = 0; r < 4; r++) { y0[r] = fmaxf(y0[r], 0.f); y1[r] = fmaxf(y1[r], 0.f); }
            float* dst = saved + (((size_t)h * ntiles + tile) * 16) * W;
            if (SAVE == 1) {
                __builtin_nontemporal_store(y0, reinterpret_cast<f32x4*>(dst + n * W + 32 * w + 4 * q));
                __builtin_nontemporal_store(y1, reinterpret_cast<f32x4*>(dst + n * W + 32 * w + 16 + 4 * q));
            }
            keep += y0 + y1;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    *reinterpret_cast<f32x4*>(out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = keep;
    if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
}

template <int NH, bool XREG, int SAVE>
static void run(const char* name, const float* W1, const float* X, float* out, float* saved, unsigned long long* cyc, int ntiles, int wgs) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((ws_kernel<NH, XREG, SAVE>), dim3(wgs), dim3(512), 0, 0, W1, X, out, saved, ntiles, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(wgs * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mx = 0, sum = 0;
    for (auto c : h) { mx = c > mx ? c : mx; sum += c; }
    const double tiles_per_wg = (double)ntiles / wgs;
    const double mfma_per_tile_simd = 2.0 * NH * 32;        // two waves per SIMD, 32 MFMAs per head and wave
    const double flop = (double)ntiles * 16 * 2.0 * W * W * NH;
    printf("%-40s %7.3f ms  %6.1f TF (%.3f of 157.3)  s_memtime ticks per tile: mean %.1f max %.1f;  MFMA floor %.0f cycles/tile\n", name, ms,
           flop / ms / 1e9, flop / ms / 1e9 / 157.3, sum / h.size() / tiles_per_wg, mx / tiles_per_wg, mfma_per_tile_simd * 32);
}

template <int NH, int SAVE>
static void run4(const char* name, const float* W1, const float* X, float* out, float* saved, unsigned long long* cyc, int ntiles, int wgs) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((ws4_kernel<NH, SAVE>), dim3(wgs), dim3(256), 0, 0, W1, X, out, saved, ntiles, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = (double)ntiles * 16 * 2.0 * W * W * NH;
    printf("%-40s %7.3f ms  %6.1f TF (%.3f of 157.3)\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main() {
    const int ntiles = 18750, NH = 5, wgs = 256;
    float *W1, *X, *out, *saved;
    unsigned long long* cyc;
    hipMalloc(&W1, sizeof(float) * NH * W * W);
    hipMalloc(&X, sizeof(float) * (size_t)ntiles * 16 * W);
    hipMalloc(&out, sizeof(float) * wgs * 512 * 4);
    hipMalloc(&saved, sizeof(float) * (size_t)NH * ntiles * 16 * W);
    hipMalloc(&cyc, 8 * wgs * 8);
    std::vector<float> hw(NH * W * W), hx((size_t)ntiles * 16 * W);
    for (size_t i = 0; i < hw.size(); i++) hw[i] = 0.01f * (float)((i * 2654435761u) % 97) - 0.5f;
    for (size_t i = 0; i < hx.size(); i++) hx[i] = 0.01f * (float)((i * 40503u) % 89) - 0.3f;
    hipMemcpy(W1, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000;      // 32 MFMAs per iteration and wave: ~0.55 ms
        float ms = 0.f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(bare_kernel, dim3(wgs), dim3(512), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double flop = (double)wgs * 8 * iters * 32 * 2048.0;
        printf("bare MFMA loop, two waves per SIMD: %.3f ms, %.1f TF (%.3f of 157.3)\n", ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
    }
    run<5, false, 0>("5 heads, X from LDS per head", W1, X, out, saved, cyc, ntiles, wgs);
    run<5, true, 0>("5 heads, X in 32 registers", W1, X, out, saved, cyc, ntiles, wgs);
    run<5, false, 1>("5 heads, X from LDS, h1 saved NT 64 B", W1, X, out, saved, cyc, ntiles, wgs);
    run<5, false, 2>("5 heads, X from LDS, h1 saved plain 64 B", W1, X, out, saved, cyc, ntiles, wgs);
    run<5, false, 3>("5 heads, X from LDS, h1 saved rows via LDS", W1, X, out, saved, cyc, ntiles, wgs);
    run<3, false, 0>("3 heads, X from LDS per head", W1, X, out, saved, cyc, ntiles, wgs);
    run4<5, 0>("4 waves x 512 regs, 5 heads", W1, X, out, saved, cyc, ntiles, wgs);
    run4<5, 1>("4 waves x 512 regs, 5 heads, saved NT", W1, X, out, saved, cyc, ntiles, wgs);
    return 0;
}
