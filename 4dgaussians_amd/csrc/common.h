// common.h -- shared host/device helpers of libfdgs (gfx950 only; no CUDA compatibility paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fdgs.h"

namespace fdgs {

// ---- error plumbing (thread-local message, never throws across the ABI) ----
extern thread_local char g_err[512];
inline int fail(int code, const char* fmt, const char* a = "", const char* b = "") {
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}
#define FDGS_HIP_CHECK(expr)                                                                       \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fdgs::fail(FDGS_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define FDGS_LAUNCH_CHECK(name, dbg, stream)                                                       \
    do {                                                                                           \
        hipError_t e_ = hipGetLastError();                                                         \
        if (e_ == hipSuccess && (dbg)) e_ = hipStreamSynchronize(stream);                          \
        if (e_ != hipSuccess) return fdgs::fail(FDGS_E_HIP, "kernel %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)
#define FDGS_REQUIRE(cond, msg)                                        \
    do {                                                               \
        if (!(cond)) return fdgs::fail(FDGS_E_INVALID, "%s", msg);    \
    } while (0)

// ---- optional per-kernel HIP-event timing (bench.py's live roofline measurement; off by default) ----
extern bool g_timing_on;
void* timing_begin(const char* name, hipStream_t stream);
void timing_end(void* rec, hipStream_t stream);
struct KernelTimer {
    void* rec;
    hipStream_t s;
    KernelTimer(const char* name, hipStream_t stream) : rec(g_timing_on ? timing_begin(name, stream) : nullptr), s(stream) {}
    ~KernelTimer() { if (rec) timing_end(rec, s); }
};
#define FDGS_CAT2(a, b) a##b
#define FDGS_CAT(a, b) FDGS_CAT2(a, b)
#define FDGS_TIMED(name, stream) fdgs::KernelTimer FDGS_CAT(fdgs_kt_, __LINE__)(name, stream)

constexpr int TILE = FDGS_TILE;
constexpr int TILE_PIX = TILE * TILE;  // 256 threads = 4 wave64 per tile
constexpr int WAVE = 64;

// ---- development / test knobs: ONE table (api.hip).  The environment is read ONCE, when the library is loaded; at run time a knob changes
// only through fdgs_tuning_set() -- no C-ABI call reads the environment.  Everything else that used to be an FDGS_* variable is a constant
// at its point of use (the tuned value; the sweeps are in profiles/r01b_tuning_sweep.txt).  Documented in INTEGRATION.md.
struct Tuning {
    int d1_form;     // FDGS_D1_FORM     0 (default: by shape, 8 at net_width 128 with up to two HexPlane levels, else 16) | 8 ( the weight-stationary form, deform_fwd_ws.h, where it applies) | 16 (deform_fwd16_kernel) | 32 (deform_fwd_kernel)
    int d1_wgs;      // FDGS_D1_WGS      workgroups of the forward kernel; -1 = two (form 16) / one (form 32) per CU, 0 = one per four tiles
    int d1_split;    // FDGS_D1_SPLIT    1 = the leftover tiles of the persistent loop are split by head over the waves
    int skip_dead;   // FDGS_SKIP_DEAD   1 = the deformation backward skips tiles without a gradient row (modes 0 / 2 of packed_rows_ready)
    int d4_mfma;     // FDGS_D4_MFMA     -1 = the caller's order hint picks the plane-gradient kernel, 1 / 0 force the splat / the per-corner form
    int d4_rows_kb;  // FDGS_D4_ROWS_KB  -1 = the time rows get all the LDS that is left; >= 0 caps it (tests: rows that do not fit)
    int tile_cull;   // FDGS_TILE_CULL   1 = exact tile culling, 0 = the reference's rectangle lists
    int rbwd_ppl;    // FDGS_RBWD_PPL    pixels per lane of the blending backward: -1 (default) = 2 up to 4 096 tiles, 4 above | 4 | 2 | 0 = the 256-thread form
    int tile_order;  // FDGS_TILE_ORDER  1 = the blending kernels take their tiles heaviest-first (per XCD), 0 = in image order
    int row_compact; // FDGS_ROW_COMPACT 1 = the deformation backward walks the non-zero ROWS (saved activations + ordered input), 0 = 32-row tiles
    int d2_form;     // FDGS_D2_FORM     0 (default: the weight-stationary backward-data kernel, deform_bwd_ws.h, where it applies) | 32 (the 32-row kernel everywhere)
};
extern Tuning g_tune;

// per-device host-side state (cached events, "function attribute already raised" flags) is indexed by the current device
constexpr int FDGS_MAX_DEVICES = 64;
inline int current_device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return d % FDGS_MAX_DEVICES;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// block-wide exclusive prefix of one value per thread (256 threads); *total = sum over the block
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* wtmp /*[4] LDS*/, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(inc, o, 64);
        if (lane >= (uint32_t)o) inc += u;
    }
    if (lane == 63) wtmp[w] = inc;
    __syncthreads();
    const uint32_t s0 = wtmp[0], s1 = wtmp[1], s2 = wtmp[2], s3 = wtmp[3];
    const uint32_t wp = (w > 0 ? s0 : 0u) + (w > 1 ? s1 : 0u) + (w > 2 ? s2 : 0u);
    *total = s0 + s1 + s2 + s3;
    __syncthreads();  // wtmp may be reused by the caller's next round
    return wp + inc - v;
}

// ---- radix sort geometry (binning.hip) ----
constexpr int SORT_THREADS = 256;
#ifndef FDGS_RSORT_ITEMS
#define FDGS_RSORT_ITEMS 16
#endif
#ifndef FDGS_NSORT_ITEMS
#define FDGS_NSORT_ITEMS 4
#endif
constexpr int SORT_ITEMS = FDGS_RSORT_ITEMS;         // keys per thread (tile sort of the R pairs)
constexpr int SORT_CHUNK = SORT_THREADS * SORT_ITEMS; // 4096 keys per workgroup
// The depth sort of the P Gaussians uses 1024-key workgroups: at 300 k Gaussians 4096-key blocks are 74 workgroups on 256 CUs, each
// pass a chain of three latency-bound launches; 293 smaller blocks fill the chip and a block's serial part is four times shorter.
constexpr int NSORT_ITEMS = FDGS_NSORT_ITEMS;
constexpr int NSORT_CHUNK = SORT_THREADS * NSORT_ITEMS;
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;

// ---- geom buffer layout (struct of arrays, 256-B aligned sections) ----
struct GeomLayout {
    size_t depth, recA, recB, recC, cov3D, tiles, clamped, rect, keys0, keys1, ids0, ids1, offsets, hist, total, cullmask, bytes;
    int sort_blocks;
};
inline GeomLayout geom_layout(int P) {
    GeomLayout g{};
    size_t o = 0;
    size_t n = (size_t)(P > 0 ? P : 1);
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes); return r; };
    g.total = take(256);  // [0] = total tiles touched (u32), [1] = scan carry scratch, [2] = 1 when tiles were culled
    g.depth = take(n * 4);
    g.recA = take(n * 16);
    g.recB = take(n * 16);
    g.recC = take(n * 16);
    g.cov3D = take(n * 24);
    g.tiles = take(n * 4);
    g.clamped = take(n * 4);
    g.rect = take(n * 8);
    g.keys0 = take(n * 4);
    g.keys1 = take(n * 4);
    g.ids0 = take(n * 4);
    g.ids1 = take(n * 4);
    g.offsets = take(n * 4);
    g.cullmask = take(n * 32);   // 256 bits per Gaussian: which tiles of its square can receive a contribution
    g.sort_blocks = cdiv((long long)n, NSORT_CHUNK);
    g.hist = take(((size_t)RADIX * g.sort_blocks + 1024) * 4);
    g.bytes = o;
    return g;
}
struct BinLayout {
    size_t tile0, tile1, gid0, gid1, hist, bytes;
    int sort_blocks;
};
inline BinLayout bin_layout(uint32_t R) {
    BinLayout b{};
    size_t o = 0, n = R > 0 ? R : 1;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes); return r; };
    b.tile0 = take(n * 4); b.tile1 = take(n * 4); b.gid0 = take(n * 4); b.gid1 = take(n * 4);
    b.sort_blocks = cdiv((long long)n, SORT_CHUNK);
    b.hist = take(((size_t)RADIX * b.sort_blocks + 1024) * 4);
    b.bytes = o;
    return b;
}
// tile rows per XCD group of the blending kernels' block -> tile map (render.hip: unit_of_block; sweep 0 / 1 / 2 / 4 in
// profiles/r03f_bench_cfg4_xcd*.json) and the largest per-XCD tile count the heaviest-first ordering sorts in LDS
constexpr int XCD_ROWS = 2;
constexpr int TILE_ORDER_MAX = 8192;
struct ImgLayout {
    size_t final_T, n_contrib, ranges, todo, order_f, order_b, bytes;
    int gx, gy, per_xcd;       // per_xcd = tiles (incl. padding) each of the 8 XCDs owns in the block -> tile map
};
inline ImgLayout img_layout(int W, int H) {
    ImgLayout m{};
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes); return r; };
    m.gx = (W + TILE - 1) / TILE; m.gy = (H + TILE - 1) / TILE;
    m.final_T = take((size_t)W * H * 4);
    m.n_contrib = take((size_t)W * H * 4);
    m.ranges = take((size_t)m.gx * m.gy * 8);
    const int groups = (m.gy + XCD_ROWS - 1) / XCD_ROWS;
    m.per_xcd = ((groups + 7) / 8) * XCD_ROWS * m.gx;
    m.todo = take((size_t)m.gx * m.gy * 4);            // per tile: entries the backward walks (max n_contrib), written by the forward
    m.order_f = take((size_t)8 * m.per_xcd * 4);       // heaviest-first tile order per XCD: forward (by list length) ...
    m.order_b = take((size_t)8 * m.per_xcd * 4);       // ... and backward (by `todo`)
    m.bytes = o;
    return m;
}
inline int tile_bits(int ntiles) {
    int b = 1;
    while ((1 << b) < ntiles) b++;
    return b;
}

template <typename T>
inline T* at(void* base, size_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + off); }
template <typename T>
inline const T* at(const void* base, size_t off) { return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off); }

}  // namespace fdgs
