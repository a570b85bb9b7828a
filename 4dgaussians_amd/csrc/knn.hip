// knn.hip -- mean squared distance to the three nearest neighbours (drop-in for simple_knn._C.distCUDA2).
//
// The reference initialises the Gaussian scales from it once per run (scene/gaussian_model.py:148:
// dist2 = clamp_min(distCUDA2(points), 1e-7); scales = log(sqrt(dist2))); the function lives in the un-vendored submodule
// submodules/simple-knn (.gitmodules:1-3), CUDA only.  Semantics restated from that call site and the published kernel:
// for every point, the three smallest squared Euclidean distances to OTHER points (self excluded by index, coincident
// points count with distance 0), averaged; slots that cannot be filled (fewer than 4 points) stay at FLT_MAX.
// Exact brute force: 256 queries per workgroup in registers, all points streamed through LDS in tiles of 1024 (broadcast
// ds_read_b128), 13 VALU operations per pair.  O(N^2) is fine for an init-time function: 300 k points = 9e10 pairs.
#include "common.h"

#include <float.h>

namespace fdgs {

constexpr int KNN_TILE = 1024;

template <bool SELF>
__device__ __forceinline__ void knn_tile(const float4* __restrict__ tile, int count, int tile_base, int self, float qx, float qy,
                                         float qz, float& b0, float& b1, float& b2) {
#pragma unroll 4
    for (int j = 0; j < count; j++) {
        const float4 p = tile[j];
        const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
        float d = dx * dx + dy * dy + dz * dz;
        if (SELF) d = (tile_base + j == self) ? FLT_MAX : d;
        // sorted insert into (b0 <= b1 <= b2)
        b2 = fminf(b2, d);
        float t = fminf(b1, b2); b2 = fmaxf(b1, b2); b1 = t;
        t = fminf(b0, b1); b1 = fmaxf(b0, b1); b0 = t;
    }
}

__global__ void __launch_bounds__(256) knn3_kernel(int N, const float* __restrict__ pts, float* __restrict__ out) {
    __shared__ float4 tile[KNN_TILE];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int q = i < N ? i : N - 1;
    const float qx = pts[3 * (size_t)q], qy = pts[3 * (size_t)q + 1], qz = pts[3 * (size_t)q + 2];
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    const int my_tile = (blockIdx.x * 256) / KNN_TILE;   // the one tile that contains this workgroup's own points
    for (int base = 0, t = 0; base < N; base += KNN_TILE, t++) {
        const int count = N - base < KNN_TILE ? N - base : KNN_TILE;
        __syncthreads();
        for (int j = threadIdx.x; j < count; j += 256)
            tile[j] = make_float4(pts[3 * (size_t)(base + j)], pts[3 * (size_t)(base + j) + 1], pts[3 * (size_t)(base + j) + 2], 0.f);
        __syncthreads();
        if (t == my_tile) knn_tile<true>(tile, count, base, i, qx, qy, qz, b0, b1, b2);
        else knn_tile<false>(tile, count, base, i, qx, qy, qz, b0, b1, b2);
    }
    if (i < N) out[i] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace fdgs

using namespace fdgs;

extern "C" int fdgs_knn3_mean_dist2(void* stream_, int N, const float* points, float* mean_dist2) {
    FDGS_REQUIRE(N >= 0 && (N == 0 || (points && mean_dist2)), "bad arguments");
    if (N == 0) return FDGS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    { FDGS_TIMED("knn3", stream); hipLaunchKernelGGL(knn3_kernel, dim3(cdiv(N, 256)), dim3(256), 0, stream, N, points, mean_dist2); }
    FDGS_LAUNCH_CHECK("knn3", 0, stream);
    return FDGS_OK;
}
