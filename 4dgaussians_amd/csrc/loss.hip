// loss.hip -- image losses on the rendered frame: L1 / L2 statistics and the 11x11 Gaussian-window SSIM, value and
// gradient with respect to the rendered image.
//
// Replaces, for the step right after render() in train.py:201-214 of the reference,
//     l1_loss(image, gt)            utils/loss_utils.py:20-21
//     psnr(image, gt)               utils/image_utils.py:17-38  (from the same sum of squares)
//     ssim(image, gt)               utils/loss_utils.py:40-66   (six depthwise 11x11 conv2d calls + ~15 elementwise kernels
//                                                                 and their autograd backward)
// by two launches: the forward pass produces all four sums and, per pixel, the three partial derivatives of the SSIM map
// that the backward pass needs; the backward pass convolves those three maps once more with the (symmetric) window.
//
//   mu1 = w*x, mu2 = w*y, e11 = w*(x x), e22 = w*(y y), e12 = w*(x y)          (zero padding, as conv2d(padding=5))
//   s1 = e11 - mu1^2, s2 = e22 - mu2^2, s12 = e12 - mu1 mu2
//   S  = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)) = A1 A2 / (B1 B2)
//   dS/dmu1 = (2 mu2 (A2 - A1) - S * 2 mu1 (B2 - B1)) / (B1 B2),  dS/de11 = -S / B2,  dS/de12 = 2 A1 / (B1 B2)
//   d(sum S)/dx = w*(dS/dmu1) + 2 x (w*(dS/de11)) + y (w*(dS/de12))
//
// The window is separable (outer product of an 11-tap Gaussian, sigma 1.5), so each pass is a row filter into LDS followed
// by a column filter: one workgroup = one 32x32 output tile of one channel plane, 42x42 haloed inputs in LDS.  HBM-bound:
// forward reads 2 planes and writes 3 maps, backward reads 5 and writes 1 (9 floats per pixel-channel per iteration).
#include "common.h"

#include <math.h>

namespace fdgs {

constexpr int LW = 11, LR = 5;       // window taps, radius
constexpr int LT = 32;               // output tile edge
constexpr int LH = LT + 2 * LR;      // haloed tile edge (42)
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

struct LossArgs {
    int planes, channels, H, W;
    const float* x; const float* y;
    float* maps;                    // [3][planes][H][W]: dS/dmu1, dS/de11, dS/de12 (forward writes, backward reads)
    float* acc;                     // [items][4]
    float* dimg;
    float w_l1, w_ssim; const float* grad_scale_dev;
    float g[LW];
};

// haloed tile of one plane into LDS, zeros outside the image
__device__ __forceinline__ void load_halo(float (*s)[LH + 1], const float* __restrict__ p, int H, int W, int x0, int y0) {
    for (int i = threadIdx.x; i < LH * LH; i += 256) {
        const int r = i / LH, c = i - r * LH;
        const int gy = y0 + r - LR, gx = x0 + c - LR;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        s[r][c] = in ? p[(size_t)gy * W + gx] : 0.f;
    }
}

template <bool MAPS>
__global__ void __launch_bounds__(256) image_loss_fwd_kernel(LossArgs a) {
    __shared__ float sx[LH][LH + 1], sy[LH][LH + 1];
    __shared__ float hz[5][LH][LT];
    __shared__ float red[4][3];
    const int plane = blockIdx.z, x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
    const size_t po = (size_t)plane * a.H * a.W;
    load_halo(sx, a.x + po, a.H, a.W, x0, y0);
    load_halo(sy, a.y + po, a.H, a.W, x0, y0);
    __syncthreads();
    for (int i = threadIdx.x; i < LH * LT; i += 256) {        // row filter
        const int r = i / LT, c = i - r * LT;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < LW; k++) {
            const float u = sx[r][c + k], v = sy[r][c + k], w = a.g[k];
            m1 = fmaf(w, u, m1); m2 = fmaf(w, v, m2);
            e11 = fmaf(w, u * u, e11); e22 = fmaf(w, v * v, e22); e12 = fmaf(w, u * v, e12);
        }
        hz[0][r][c] = m1; hz[1][r][c] = m2; hz[2][r][c] = e11; hz[3][r][c] = e22; hz[4][r][c] = e12;
    }
    __syncthreads();
    // column filter: thread (tx, ty) owns output rows 4 ty .. 4 ty + 3 of column tx
    const int tx = threadIdx.x & (LT - 1), ty = threadIdx.x >> 5;
    float o[5][4];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        float col[LW + 3];
#pragma unroll
        for (int k = 0; k < LW + 3; k++) col[k] = hz[q][4 * ty + k][tx];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < LW; k++) s = fmaf(a.g[k], col[j + k], s);
            o[q][j] = s;
        }
    }
    float sum_s = 0.f, sum_a = 0.f, sum_q = 0.f;
    const int gx = x0 + tx;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int gy = y0 + 4 * ty + j;
        if (gx < a.W && gy < a.H) {
            const float m1 = o[0][j], m2 = o[1][j];
            const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
            const float s1 = o[2][j] - m11, s2 = o[3][j] - m22, s12 = o[4][j] - m12;
            const float A1 = 2.f * m12 + SSIM_C1, A2 = 2.f * s12 + SSIM_C2;
            const float B1 = m11 + m22 + SSIM_C1, B2 = s1 + s2 + SSIM_C2;
            const float D = B1 * B2;
            const float S = A1 * A2 / D;
            sum_s += S;
            const float d = sx[4 * ty + j + LR][tx + LR] - sy[4 * ty + j + LR][tx + LR];
            sum_a += fabsf(d); sum_q += d * d;
            if (MAPS) {
                const size_t n = (size_t)a.planes * a.H * a.W, at = po + (size_t)gy * a.W + gx;
                a.maps[at] = (2.f * m2 * (A2 - A1) - S * 2.f * m1 * (B2 - B1)) / D;
                a.maps[n + at] = -S / B2;
                a.maps[2 * n + at] = 2.f * A1 / D;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum_s += __shfl_xor(sum_s, off, 64); sum_a += __shfl_xor(sum_a, off, 64); sum_q += __shfl_xor(sum_q, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sum_a; red[threadIdx.x >> 6][1] = sum_q; red[threadIdx.x >> 6][2] = sum_s; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float* acc = a.acc + 4 * (plane / a.channels);
        const int k = threadIdx.x;
        atomicAdd(&acc[k == 2 ? 3 : k], red[0][k] + red[1][k] + red[2][k] + red[3][k]);
    }
    if (threadIdx.x == 3 && blockIdx.x == 0 && blockIdx.y == 0)
        atomicAdd(&a.acc[4 * (plane / a.channels) + 2], (float)a.H * (float)a.W);
}

__global__ void __launch_bounds__(256) image_loss_bwd_kernel(LossArgs a) {
    __shared__ float sm[3][LH][LH + 1];
    __shared__ float hz[3][LH][LT];
    const int plane = blockIdx.z, x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
    const size_t po = (size_t)plane * a.H * a.W, n = (size_t)a.planes * a.H * a.W;
#pragma unroll
    for (int q = 0; q < 3; q++) load_halo(sm[q], a.maps + q * n + po, a.H, a.W, x0, y0);
    __syncthreads();
    for (int i = threadIdx.x; i < LH * LT; i += 256) {
        const int r = i / LT, c = i - r * LT;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LW; k++) {
            const float w = a.g[k];
            s0 = fmaf(w, sm[0][r][c + k], s0); s1 = fmaf(w, sm[1][r][c + k], s1); s2 = fmaf(w, sm[2][r][c + k], s2);
        }
        hz[0][r][c] = s0; hz[1][r][c] = s1; hz[2][r][c] = s2;
    }
    __syncthreads();
    const int tx = threadIdx.x & (LT - 1), ty = threadIdx.x >> 5;
    float o[3][4];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        float col[LW + 3];
#pragma unroll
        for (int k = 0; k < LW + 3; k++) col[k] = hz[q][4 * ty + k][tx];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < LW; k++) s = fmaf(a.g[k], col[j + k], s);
            o[q][j] = s;
        }
    }
    const float gs = a.grad_scale_dev ? *a.grad_scale_dev : 1.f;
    const float kl = a.w_l1 * gs, ks = a.w_ssim * gs;
    const int gx = x0 + tx;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int gy = y0 + 4 * ty + j;
        if (gx < a.W && gy < a.H) {
            const size_t at = po + (size_t)gy * a.W + gx;
            const float x = a.x[at], y = a.y[at], d = x - y;
            const float ds = o[0][j] + 2.f * x * o[1][j] + y * o[2][j];
            a.dimg[at] = ks * ds + (d > 0.f ? kl : (d < 0.f ? -kl : 0.f));
        }
    }
}

// the reference's window: exp() in double, stored as float, normalised by the float sum (utils/loss_utils.py:26-28)
inline void fill_window(float* g) {
    float v[LW], s = 0.f;
    for (int i = 0; i < LW; i++) { v[i] = (float)exp(-(double)((i - LR) * (i - LR)) / (2.0 * 1.5 * 1.5)); s += v[i]; }
    for (int i = 0; i < LW; i++) g[i] = v[i] / s;
}
}  // namespace fdgs

using namespace fdgs;

extern "C" int fdgs_image_loss_fwd(void* stream_, int items, int channels, int H, int W, const float* img, const float* gt,
                                   float* ssim_maps_opt, float* acc) {
    FDGS_REQUIRE(items >= 0 && channels >= 1 && H >= 0 && W >= 0, "bad sizes");
    if (items == 0 || H == 0 || W == 0) return FDGS_OK;
    FDGS_REQUIRE(img && gt && acc, "NULL pointer");
    FDGS_REQUIRE((long long)items * channels <= 65535, "too many image planes for one launch");
    hipStream_t stream = (hipStream_t)stream_;
    LossArgs a{};
    a.planes = items * channels; a.channels = channels; a.H = H; a.W = W; a.x = img; a.y = gt; a.maps = ssim_maps_opt; a.acc = acc;
    fill_window(a.g);
    const dim3 grid(cdiv(W, LT), cdiv(H, LT), a.planes);
    { FDGS_TIMED("image_loss_fwd", stream);
      if (ssim_maps_opt) hipLaunchKernelGGL(image_loss_fwd_kernel<true>, grid, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL(image_loss_fwd_kernel<false>, grid, dim3(256), 0, stream, a); }
    FDGS_LAUNCH_CHECK("image_loss_fwd", 0, stream);
    return FDGS_OK;
}

extern "C" int fdgs_image_loss_bwd(void* stream_, int items, int channels, int H, int W, const float* img, const float* gt,
                                   const float* ssim_maps, float w_l1, float w_ssim, const float* grad_scale_dev_opt, float* dimg) {
    FDGS_REQUIRE(items >= 0 && channels >= 1 && H >= 0 && W >= 0, "bad sizes");
    if (items == 0 || H == 0 || W == 0) return FDGS_OK;
    FDGS_REQUIRE(img && gt && ssim_maps && dimg, "NULL pointer");
    FDGS_REQUIRE((long long)items * channels <= 65535, "too many image planes for one launch");
    hipStream_t stream = (hipStream_t)stream_;
    LossArgs a{};
    a.planes = items * channels; a.channels = channels; a.H = H; a.W = W; a.x = img; a.y = gt; a.maps = const_cast<float*>(ssim_maps);
    a.dimg = dimg; a.w_l1 = w_l1; a.w_ssim = w_ssim; a.grad_scale_dev = grad_scale_dev_opt;
    fill_window(a.g);
    const dim3 grid(cdiv(W, LT), cdiv(H, LT), a.planes);
    { FDGS_TIMED("image_loss_bwd", stream); hipLaunchKernelGGL(image_loss_bwd_kernel, grid, dim3(256), 0, stream, a); }
    FDGS_LAUNCH_CHECK("image_loss_bwd", 0, stream);
    return FDGS_OK;
}
