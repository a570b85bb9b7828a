// binning.hip -- K2..K5: depth sort of the Gaussians, tile-pair expansion, stable tile sort, tile ranges.
//
// MI355X-first restructuring of the reference's binning (InclusiveSum + duplicateWithKeys +
// cub::DeviceRadixSort over 64-bit (tile|depth) keys + identifyTileRanges).  The 64-bit sort over R pairs is split
// into two stable LSD radix sorts that give the identical order:
//   (1) sort the P Gaussians by the fp32 bits of their view depth (4 passes over P 32-bit keys), then
//   (2) emit the (tile, Gaussian) pairs in that depth order and stable-sort them by tile id only
//       (ceil(log2(#tiles)/8) = 2 passes over R pairs instead of 6, 8 B per pair instead of 12 B).
// Ties (equal tile, equal depth) resolve by Gaussian index in both formulations because every pass is stable.
//
// Radix pass = wave64 ballot ranking (one ballot per digit bit gives each lane its rank among equal digits), per-wave
// digit counters in LDS, keys regrouped by digit in LDS so that the global scatter writes contiguous runs.  A sort over
// nbits key bits runs ceil(nbits / 8) passes of near-equal digit width (32 -> 8,8,8,8; the 13-bit tile id -> 7,6).
#include "common.h"

namespace fdgs {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ---------------------------------------------------------------- scans (all multi-workgroup)
// Radix offsets, step 1: workgroup d scans row d of hist[digit][block] (exclusive, in place) and writes the digit
// total.  256 workgroups instead of one: the R-pair sort of a 1352x1014 frame has 256 x 1556 counters per pass.
__global__ void __launch_bounds__(256) radix_digit_scan_kernel(uint32_t* __restrict__ hist, int nblocks,
                                                               uint32_t* __restrict__ dtot) {
    __shared__ uint32_t wtmp[4];
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    const int t = threadIdx.x;
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += 1024) {
        uint32_t v[4];
        const int i0 = base + t * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (i0 + k < nblocks) ? row[i0 + k] : 0u;
        uint32_t tot;
        uint32_t run = carry + block_excl_scan_256(v[0] + v[1] + v[2] + v[3], wtmp, &tot);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k < nblocks) row[i0 + k] = run;
            run += v[k];
        }
        carry += tot;
    }
    if (t == 0) dtot[blockIdx.x] = carry;
}

// Scan of n values (GATHER: value i = src[idx[i]]) in 4096-value chunks: every workgroup scans its chunk (inclusive) and publishes the
// chunk total; the consumer (expand_pairs_kernel) adds the totals of the chunks in front of the one it reads.
constexpr int SCAN_CHUNK = 4096;
template <bool GATHER>
__global__ void __launch_bounds__(256) scan_chunk_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx,
                                                         uint32_t* __restrict__ dst, uint32_t n, uint32_t* __restrict__ bsum) {
    __shared__ uint32_t wtmp[4];
    const uint32_t i0 = blockIdx.x * SCAN_CHUNK + threadIdx.x * 16;
    uint32_t v[16];
    if (i0 + 16 <= n) {
        if (GATHER) {
            uint32_t id[16];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 q = reinterpret_cast<const uint4*>(idx + i0)[k];
                id[4 * k] = q.x; id[4 * k + 1] = q.y; id[4 * k + 2] = q.z; id[4 * k + 3] = q.w;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = src[id[k]];
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 q = reinterpret_cast<const uint4*>(src + i0)[k];
                v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = (i0 + k < n) ? (GATHER ? src[idx[i0 + k]] : src[i0 + k]) : 0u;
    }
    uint32_t tsum = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { tsum += v[k]; v[k] = tsum; }   // thread-local inclusive
    uint32_t tot;
    const uint32_t excl = block_excl_scan_256(tsum, wtmp, &tot);
    if (i0 + 16 <= n) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            reinterpret_cast<uint4*>(dst + i0)[k] = make_uint4(v[4 * k] + excl, v[4 * k + 1] + excl, v[4 * k + 2] + excl, v[4 * k + 3] + excl);
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) if (i0 + k < n) dst[i0 + k] = v[k] + excl;
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
// ---------------------------------------------------------------- radix histogram: hist[digit*nblocks + block]
// `n_dev` (capacity mode, fdgs_raster_fwd_capacity): the key count lives on the device -- the launch is sized for the capacity `n`, the
// kernel works on min(*n_dev, n) keys; blocks behind the end still publish their (zero) counters.
__device__ __forceinline__ uint32_t live_count(const uint32_t* __restrict__ n_dev, uint32_t n) {
    if (!n_dev) return n;
    const uint32_t t = *n_dev;
    return t < n ? t : n;
}
template <int ITEMS>
__global__ void __launch_bounds__(SORT_THREADS) radix_hist_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ n_dev, uint32_t n,
                                                                  int shift, uint32_t mask, uint32_t* __restrict__ hist, int nblocks) {
    __shared__ uint32_t h[RADIX];
    n = live_count(n_dev, n);
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (SORT_THREADS * ITEMS);
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        uint32_t i = base + j * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// ---------------------------------------------------------------- radix scatter (stable)
// BITS = width of this pass's digit (<= RADIX_BITS): a sort over nbits key bits uses ceil(nbits / 8) passes of near-equal width -- the
// 13-bit tile sort of a 1352x1014 frame is a 7-bit and a 6-bit pass: 13 instead of 16 ballot rounds per key, and the runs a block
// writes per digit are 32 / 64 keys long (whole 128-byte lines) instead of 16.
template <int BITS, int ITEMS>
__global__ void __launch_bounds__(SORT_THREADS) radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                     const uint32_t* __restrict__ vals_in,
                                                                     uint32_t* __restrict__ keys_out,
                                                                     uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ n_dev,
                                                                     uint32_t n, int shift, const uint32_t* __restrict__ hist, int nblocks,
                                                                     const uint32_t* __restrict__ dtot) {
    __shared__ uint32_t wcnt[4][RADIX];   // per-wave digit counters
    __shared__ uint32_t lbase[RADIX];     // start of digit d inside the block-local regrouped array
    __shared__ uint32_t gbase[RADIX];     // global start of (digit d, this block)
    constexpr int CHUNK = SORT_THREADS * ITEMS;
    __shared__ uint32_t skey[CHUNK];
    __shared__ uint32_t sval[CHUNK];
    __shared__ uint32_t wtmp[4];
    __shared__ uint32_t wtmp2[4];
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    n = live_count(n_dev, n);
    if (blockIdx.x * (uint32_t)CHUNK >= n) return;          // (capacity mode: a block behind the end; uniform over the workgroup)
#pragma unroll
    for (int k = 0; k < 4; k++) wcnt[k][t] = 0;
    __syncthreads();
    const uint32_t wave_base = blockIdx.x * CHUNK + w * (CHUNK / 4);
    uint32_t key[ITEMS], val[ITEMS];
    uint16_t rank[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint32_t i = wave_base + j * 64 + lane;
        const bool valid = i < n;
        key[j] = valid ? keys_in[i] : 0xFFFFFFFFu;
        val[j] = valid ? vals_in[i] : 0u;
        const uint32_t d = (key[j] >> shift) & ((1u << BITS) - 1u);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        // lanes that are !valid keep a peers mask that still contains only valid lanes with their digit pattern;
        // they never use it.
        const uint32_t below = mbcnt(peers);
        const uint32_t cnt = (uint32_t)__popcll(peers);
        const int leader = __ffsll((unsigned long long)peers) - 1;
        uint32_t basev = 0;
        if (valid && (int)lane == leader) {
            basev = wcnt[w][d];
            wcnt[w][d] = basev + cnt;
        }
        basev = __shfl(basev, leader < 0 ? 0 : leader, 64);
        rank[j] = (uint16_t)(basev + below);
    }
    __syncthreads();
    // per digit (thread t = digit): wave prefixes, block count, block-local exclusive scan, global base
    {
        const uint32_t c0 = wcnt[0][t], c1 = wcnt[1][t], c2 = wcnt[2][t], c3 = wcnt[3][t];
        const uint32_t tot = c0 + c1 + c2 + c3;
        wcnt[0][t] = 0; wcnt[1][t] = c0; wcnt[2][t] = c0 + c1; wcnt[3][t] = c0 + c1 + c2;
        // two 256-wide exclusive scans at once: the block-local digit counts and the global digit totals
        const uint32_t dt = dtot[t];
        uint32_t inc = tot, dinc = dt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t u = __shfl_up(inc, o, 64), du = __shfl_up(dinc, o, 64);
            if (lane >= (uint32_t)o) { inc += u; dinc += du; }
        }
        if (lane == 63) { wtmp[w] = inc; wtmp2[w] = dinc; }
        __syncthreads();
        uint32_t wp = 0, dwp = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { wp += (k < (int)w) ? wtmp[k] : 0u; dwp += (k < (int)w) ? wtmp2[k] : 0u; }
        lbase[t] = wp + inc - tot;
        // global start of (digit t, this block) = start of digit t + keys of digit t in earlier blocks
        gbase[t] = (dwp + dinc - dt) + hist[(size_t)t * nblocks + blockIdx.x];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint32_t i = wave_base + j * 64 + lane;
        if (i < n) {
            const uint32_t d = (key[j] >> shift) & ((1u << BITS) - 1u);
            const uint32_t pos = lbase[d] + wcnt[w][d] + rank[j];
            skey[pos] = key[j];
            sval[pos] = val[j];
        }
    }
    __syncthreads();
    const uint32_t block_base = blockIdx.x * CHUNK;
    const uint32_t block_n = (n - block_base) < (uint32_t)CHUNK ? (n - block_base) : (uint32_t)CHUNK;
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint32_t li = j * SORT_THREADS + t;
        if (li < block_n) {
            const uint32_t k = skey[li];
            const uint32_t d = (k >> shift) & ((1u << BITS) - 1u);
            const uint32_t pos = gbase[d] + (li - lbase[d]);
            keys_out[pos] = k;
            vals_out[pos] = sval[li];
        }
    }
}

// LSD radix sort of n (key,val) pairs over bits [0,nbits). Ping-pongs between (k0,v0) and (k1,v1); returns which
// buffer holds the result (0 or 1) through *result_in.
int radix_sort_pairs(hipStream_t stream, uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, uint32_t n, int nbits,
                     uint32_t* hist, int nblocks, int debug, int* result_in, int items = SORT_ITEMS, const uint32_t* n_dev = nullptr) {
    int cur = 0;
    if (n > 0) {
        const int npass = (nbits + RADIX_BITS - 1) / RADIX_BITS;
        int shift = 0;
        for (int ps = 0; ps < npass; ps++) {
            const int bits = nbits / npass + (ps < nbits % npass ? 1 : 0);      // near-equal digit widths (32 -> 8,8,8,8; 13 -> 7,6)
            const uint32_t mask = (1u << bits) - 1u;
            uint32_t* ki = cur ? k1 : k0; uint32_t* vi = cur ? v1 : v0;
            uint32_t* ko = cur ? k0 : k1; uint32_t* vo = cur ? v0 : v1;
            {
                FDGS_TIMED("radix_hist", stream);
                if (items == NSORT_ITEMS) hipLaunchKernelGGL(radix_hist_kernel<NSORT_ITEMS>, dim3(nblocks), dim3(SORT_THREADS), 0, stream, ki, n_dev, n, shift, mask, hist, nblocks);
                else hipLaunchKernelGGL(radix_hist_kernel<SORT_ITEMS>, dim3(nblocks), dim3(SORT_THREADS), 0, stream, ki, n_dev, n, shift, mask, hist, nblocks);
            }
            FDGS_LAUNCH_CHECK("radix_hist", debug, stream);
            uint32_t* dtot = hist + (size_t)RADIX * nblocks;  // 256 digit totals live in the slack behind the counters
            { FDGS_TIMED("radix_scan", stream); hipLaunchKernelGGL(radix_digit_scan_kernel, dim3(RADIX), dim3(256), 0, stream, hist, nblocks, dtot); }
            FDGS_LAUNCH_CHECK("radix_scan", debug, stream);
            {
                FDGS_TIMED("radix_scatter", stream);
#define FDGS_SCATTER(B_)                                                                                                                  \
    do {                                                                                                                              \
        if (items == NSORT_ITEMS) hipLaunchKernelGGL((radix_scatter_kernel<B_, NSORT_ITEMS>), dim3(nblocks), dim3(SORT_THREADS), 0, stream, ki, vi, ko, vo, n_dev, n, shift, hist, nblocks, dtot); \
        else hipLaunchKernelGGL((radix_scatter_kernel<B_, SORT_ITEMS>), dim3(nblocks), dim3(SORT_THREADS), 0, stream, ki, vi, ko, vo, n_dev, n, shift, hist, nblocks, dtot); \
    } while (0)
                switch (bits) {
                    case 1: FDGS_SCATTER(1); break; case 2: FDGS_SCATTER(2); break; case 3: FDGS_SCATTER(3); break; case 4: FDGS_SCATTER(4); break;
                    case 5: FDGS_SCATTER(5); break; case 6: FDGS_SCATTER(6); break; case 7: FDGS_SCATTER(7); break; default: FDGS_SCATTER(8); break;
                }
#undef FDGS_SCATTER
            }
            FDGS_LAUNCH_CHECK("radix_scatter", debug, stream);
            shift += bits;
            cur ^= 1;
        }
    }
    *result_in = cur;
    return FDGS_OK;
}

// ---------------------------------------------------------------- pair expansion in depth order
// One workgroup per 256 consecutive depth-sorted Gaussians; every lane then walks the workgroup's pair range with a
// stride of 256 and finds the owning Gaussian by an 8-step binary search in LDS, so the pair stream is written
// fully coalesced.
// With exact tile culling (gs_math.h) a Gaussian's pairs are the SET bits of its 256-bit tile mask, in bit order: local
// pair index -> position of the local-th set bit (squares of more than 256 tiles are not culled and keep the direct map).
__device__ __forceinline__ uint32_t nth_set_bit(const uint4 lo, const uint4 hi, uint32_t n) {
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t base = 0, word = w[0];
#pragma unroll
    for (int q = 0; q < 7; q++) {
        const uint32_t c = (uint32_t)__popc(word);
        const bool next = n >= c && base == 32u * q;   // still walking: skip this word
        n = next ? n - c : n;
        word = next ? w[q + 1] : word;
        base = next ? base + 32u : base;
    }
    uint32_t pos = 0;
#pragma unroll
    for (int sh = 16; sh > 0; sh >>= 1) {
        const uint32_t c = (uint32_t)__popc(word & ((1u << sh) - 1u));
        const bool up = n >= c;
        n = up ? n - c : n;
        word = up ? word >> sh : word;
        pos = up ? pos + sh : pos;
    }
    return base + pos;
}

__global__ void __launch_bounds__(256) expand_pairs_kernel(int P, const uint32_t* __restrict__ sorted_ids,
                                                           const uint32_t* __restrict__ offsets_incl,
                                                           const uint32_t* __restrict__ tiles, const uint2* __restrict__ rect,
                                                           const uint4* __restrict__ cullmask, const uint32_t* __restrict__ total,
                                                           int gx, uint32_t* __restrict__ pair_tile,
                                                           uint32_t* __restrict__ pair_gid, uint32_t* __restrict__ ranges_zero, uint32_t ranges_n,
                                                           uint32_t cap /* capacity mode: pairs behind it are dropped (the FARTHEST: pairs come in depth order) */,
                                                           const uint32_t* __restrict__ bsum /* totals of the 4096-Gaussian scan chunks: `offsets_incl` is chunk-local */,
                                                           uint32_t* __restrict__ host_count /* opt (capacity mode): pinned host word that receives the pair count */) {
    // (the per-tile ranges tile_ranges_kernel fills at the end of the stage: cleared here, on the way, instead of by a memset node)
    if (blockIdx.y == 0)
        for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < ranges_n; k += gridDim.x * 256) ranges_zero[k] = 0u;
    __shared__ uint32_t s_end[256];
    __shared__ uint32_t s_gid[256];
    __shared__ uint32_t s_part[4];
    __shared__ uint32_t s_cnt0;
    __shared__ uint2 s_rect[256];
    __shared__ uint4 s_mask[256][2];
    const bool culled = total[2] != 0;
    const int t = threadIdx.x;
    const int i = blockIdx.x * 256 + t;
    uint32_t gid = 0, end = 0, cnt = 0;
    uint2 rc = make_uint2(0, 0);
    if (i < P) {
        gid = sorted_ids[i];
        end = offsets_incl[i];
        cnt = tiles[gid];
        rc = rect[gid];
        if (culled && cnt) { s_mask[threadIdx.x][0] = cullmask[2 * (size_t)gid]; s_mask[threadIdx.x][1] = cullmask[2 * (size_t)gid + 1]; }
    } else {
        end = offsets_incl[P - 1];
    }
    s_end[t] = end; s_gid[t] = gid; s_rect[t] = rc;
    if (t == 0) s_cnt0 = cnt;
    // the scan left chunk-LOCAL inclusive offsets (scan_chunk_kernel): the pairs in front of this block's chunk are the totals of the chunks
    // before it (the second scan kernel that used to add them to every offset is gone: one launch less)
    {
        const int chunk = (blockIdx.x * 256) / SCAN_CHUNK;
        uint32_t part = 0;
        for (int j = t; j < chunk; j += 256) part += bsum[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if ((t & 63) == 0) s_part[t >> 6] = part;
    }
    if (host_count && blockIdx.x == 0 && blockIdx.y == 0 && t == 0)
        __hip_atomic_store(host_count, total[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // (zero-copy: no copy node on the stream)
    __syncthreads();
    const uint32_t cp = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    // (everything below in chunk-local pair positions; `cp` makes them global where memory is addressed)
    const uint32_t start = s_end[0] - s_cnt0;          // local offset in front of the block's first Gaussian
    const uint32_t stop_g = s_end[255] + cp < cap ? s_end[255] + cp : cap;
    // gridDim.y workgroups share one block of 256 depth-consecutive Gaussians and interleave its pairs: the nearest Gaussians (the first
    // blocks) cover 60+ tiles each, five times the average.  Measured: 2 workgroups per block 0.036 ms, 1: 0.042, 4: 0.042, 8: 0.064 (the
    // staging of the 256 Gaussians -- two dependent gathers -- is repeated per workgroup)
    for (uint32_t pg = cp + start + t + 256u * blockIdx.y; pg < stop_g; pg += 256u * gridDim.y) {
        const uint32_t p = pg - cp;
        // smallest j with s_end[j] > p
        int lo = 0, hi = 255;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            int mid = (lo + hi) >> 1;
            if (s_end[mid] > p) hi = mid; else lo = mid + 1;
        }
        const int j = lo;
        const uint2 r = s_rect[j];
        const uint32_t xmin = r.x & 0xFFFFu, ymin = r.x >> 16, xmax = r.y & 0xFFFFu;
        const uint32_t w = xmax - xmin;
        const uint32_t jstart = (j == 0) ? start : s_end[j - 1];
        uint32_t local = p - jstart;
        if (culled && w * ((r.y >> 16) - ymin) <= 256u) local = nth_set_bit(s_mask[j][0], s_mask[j][1], local);
        const uint32_t ty = ymin + local / w, tx = xmin + local % w;
        pair_tile[pg] = ty * (uint32_t)gx + tx;
        pair_gid[pg] = s_gid[j];
    }
}

// four pairs per thread (one 16-byte load + the two neighbours): a quarter of the threads of the one-pair form, whose 16.6 k workgroups of
// three dependent 4-byte loads each were launch / latency bound (10.8 us for 17 MB)
__global__ void __launch_bounds__(256) tile_ranges_kernel(const uint32_t* __restrict__ n_dev, uint32_t R, const uint32_t* __restrict__ pair_tile,
                                                          uint2* __restrict__ ranges) {
    R = live_count(n_dev, R);
    const uint32_t p0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= R) return;
    uint32_t t[6];          // t[0] = tile of pair p0 - 1, t[1..4] = pairs p0 .. p0 + 3, t[5] = pair p0 + 4
    if (p0 + 4 <= R) {
        const uint4 q = *reinterpret_cast<const uint4*>(pair_tile + p0);
        t[1] = q.x; t[2] = q.y; t[3] = q.z; t[4] = q.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) t[1 + k] = p0 + k < R ? pair_tile[p0 + k] : 0xFFFFFFFFu;
    }
    t[0] = p0 > 0 ? pair_tile[p0 - 1] : 0xFFFFFFFFu;
    t[5] = p0 + 4 < R ? pair_tile[p0 + 4] : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t p = p0 + k;
        if (p >= R) break;
        const uint32_t tcur = t[1 + k];
        if (p == 0 || t[k] != tcur) ranges[tcur].x = p;
        if (p == R - 1 || t[2 + k] != tcur) ranges[tcur].y = p + 1;
    }
}

int validate_raster_params(const fdgs_raster_params* p);

}  // namespace fdgs

using namespace fdgs;

int fdgs_preprocess_fwd_impl(void* stream_, const fdgs_raster_params* p, void* geom, int32_t* radii, uint32_t* host_count_dev);   // preprocess.hip

extern "C" int fdgs_geom_bytes(int P, size_t* bytes) {
    FDGS_REQUIRE(P >= 0 && bytes, "bad arguments");
    *bytes = geom_layout(P).bytes;
    return FDGS_OK;
}
extern "C" int fdgs_img_bytes(int W, int H, size_t* bytes) {
    FDGS_REQUIRE(W > 0 && H > 0 && bytes, "bad arguments");
    *bytes = img_layout(W, H).bytes;
    return FDGS_OK;
}
extern "C" int fdgs_binning_bytes(uint32_t R, int W, int H, size_t* bytes) {
    FDGS_REQUIRE(W > 0 && H > 0 && bytes, "bad arguments");
    *bytes = bin_layout(R).bytes;
    return FDGS_OK;
}

// depth sort + offsets scan; the pair total goes to *num_rendered_host asynchronously.  wait = true: returns when it has arrived (the
// reference's one blocking read-back), false: never blocks (capacity mode)
// `skip_copy` (capacity mode): the pair count reaches the host word through a store of expand_pairs_kernel (zero-copy), not a copy node
static int bin_prepare_impl(void* stream_, const fdgs_raster_params* p, void* geom, uint32_t* num_rendered_host, bool wait, bool skip_copy = false) {
    int rc = validate_raster_params(p);
    if (rc) return rc;
    FDGS_REQUIRE(geom && num_rendered_host, "geom/num_rendered_host is NULL");
    hipStream_t stream = (hipStream_t)stream_;
    if (p->P == 0) { *num_rendered_host = 0; return FDGS_OK; }
    if (wait) *num_rendered_host = 0;
    GeomLayout gl = geom_layout(p->P);
    // the total was accumulated by preprocess: start its read-back now, sort while it is in flight
    // one event per host thread, created on first use and kept (the ABI contract is one host thread per stream)
    // (per DEVICE: an event belongs to the device that was current when it was created; a thread that later renders on another GPU
    // gets its own)
    static thread_local hipEvent_t evs[FDGS_MAX_DEVICES] = {};
    const int dev_ = current_device_slot();
    hipEvent_t& ev = evs[dev_];
    if (wait && !ev) FDGS_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (!skip_copy) FDGS_HIP_CHECK(hipMemcpyAsync(num_rendered_host, at<uint32_t>(geom, gl.total), 4, hipMemcpyDeviceToHost, stream));
    if (wait) FDGS_HIP_CHECK(hipEventRecord(ev, stream));
    int in = 0;
    rc = radix_sort_pairs(stream, at<uint32_t>(geom, gl.keys0), at<uint32_t>(geom, gl.ids0), at<uint32_t>(geom, gl.keys1),
                          at<uint32_t>(geom, gl.ids1), (uint32_t)p->P, 32, at<uint32_t>(geom, gl.hist), gl.sort_blocks, p->debug,
                          &in, NSORT_ITEMS);
    if (rc) return rc;
    // 4 passes: result is back in buffer 0
    const uint32_t* sorted_ids = at<uint32_t>(geom, in ? gl.ids1 : gl.ids0);
    if (in != 0) {  // keep the contract "sorted ids live in ids0" for any pass count
        FDGS_HIP_CHECK(hipMemcpyAsync(at<uint32_t>(geom, gl.ids0), sorted_ids, (size_t)p->P * 4, hipMemcpyDeviceToDevice, stream));
    }
    {
        // point_offsets = inclusive scan of tiles_touched in depth order; chunk totals reuse the (now idle) sort counters
        FDGS_TIMED("scan_tiles", stream);
        const int nchunks = cdiv(p->P, SCAN_CHUNK);
        uint32_t* bsum = at<uint32_t>(geom, gl.hist);
        // (chunk-local offsets + chunk totals: expand_pairs_kernel adds the totals of the chunks in front of its block itself)
        hipLaunchKernelGGL((scan_chunk_kernel<true>), dim3(nchunks), dim3(256), 0, stream, at<uint32_t>(geom, gl.tiles),
                           at<uint32_t>(geom, gl.ids0), at<uint32_t>(geom, gl.offsets), (uint32_t)p->P, bsum);
    }
    {
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) { return fail(FDGS_E_HIP, "kernel %s failed: %s", "scan_tiles", hipGetErrorString(e_)); }
    }
    if (wait) {
        hipError_t e = hipEventSynchronize(ev);
        if (e != hipSuccess) return fail(FDGS_E_HIP, "%s failed: %s", "hipEventSynchronize", hipGetErrorString(e));
    }
    if (p->debug) FDGS_HIP_CHECK(hipStreamSynchronize(stream));
    return FDGS_OK;
}

extern "C" int fdgs_bin_prepare(void* stream_, const fdgs_raster_params* p, void* geom, uint32_t* num_rendered_host) {
    return bin_prepare_impl(stream_, p, geom, num_rendered_host, true);
}

// pair expansion + tile sort + ranges.  from_device: R is the CAPACITY of `binning`; the kernels read the true pair count from the geom
// buffer and work on min(true, capacity) pairs
static int bin_sort_impl(void* stream_, const fdgs_raster_params* p, void* geom, void* binning, void* img, uint32_t R, bool from_device,
                         uint32_t* host_count_dev = nullptr) {
    int rc = validate_raster_params(p);
    if (rc) return rc;
    FDGS_REQUIRE(geom && img && (binning || R == 0), "geom/binning/img is NULL");
    hipStream_t stream = (hipStream_t)stream_;
    ImgLayout il = img_layout(p->W, p->H);
    if (p->P == 0 || R == 0) {
        FDGS_HIP_CHECK(hipMemsetAsync(at<char>(img, il.ranges), 0, (size_t)il.gx * il.gy * 8, stream));
        return FDGS_OK;
    }
    GeomLayout gl = geom_layout(p->P);
    BinLayout bl = bin_layout(R);
    const uint32_t* n_dev = from_device ? at<uint32_t>(geom, gl.total) : nullptr;
    { FDGS_TIMED("expand_pairs", stream); hipLaunchKernelGGL(expand_pairs_kernel, dim3(cdiv(p->P, 256), 2), dim3(256), 0, stream, p->P, at<uint32_t>(geom, gl.ids0),
                       at<uint32_t>(geom, gl.offsets), at<uint32_t>(geom, gl.tiles), at<uint2>(geom, gl.rect), at<uint4>(geom, gl.cullmask),
                       at<uint32_t>(geom, gl.total), il.gx,
                       at<uint32_t>(binning, bl.tile0), at<uint32_t>(binning, bl.gid0), at<uint32_t>(img, il.ranges), (uint32_t)(il.gx * il.gy * 2),
                       from_device ? R : 0xFFFFFFFFu, at<uint32_t>(geom, gl.hist), host_count_dev); }
    FDGS_LAUNCH_CHECK("expand_pairs", p->debug, stream);
    int in = 0;
    rc = radix_sort_pairs(stream, at<uint32_t>(binning, bl.tile0), at<uint32_t>(binning, bl.gid0), at<uint32_t>(binning, bl.tile1),
                          at<uint32_t>(binning, bl.gid1), R, tile_bits(il.gx * il.gy), at<uint32_t>(binning, bl.hist), bl.sort_blocks,
                          p->debug, &in, SORT_ITEMS, n_dev);
    if (rc) return rc;
    if (in != 0) {  // odd number of passes: bring the result back to buffer 0 (contract used by render + accessors)
        FDGS_HIP_CHECK(hipMemcpyAsync(at<uint32_t>(binning, bl.tile0), at<uint32_t>(binning, bl.tile1), (size_t)R * 4,
                                      hipMemcpyDeviceToDevice, stream));
        FDGS_HIP_CHECK(hipMemcpyAsync(at<uint32_t>(binning, bl.gid0), at<uint32_t>(binning, bl.gid1), (size_t)R * 4,
                                      hipMemcpyDeviceToDevice, stream));
    }
    { FDGS_TIMED("tile_ranges", stream); hipLaunchKernelGGL(tile_ranges_kernel, dim3(cdiv(R, 1024)), dim3(256), 0, stream, n_dev, R, at<uint32_t>(binning, bl.tile0),
                       at<uint2>(img, il.ranges)); }
    FDGS_LAUNCH_CHECK("tile_ranges", p->debug, stream);
    return FDGS_OK;
}

extern "C" int fdgs_bin_sort(void* stream_, const fdgs_raster_params* p, void* geom, void* binning, void* img, uint32_t R) {
    return bin_sort_impl(stream_, p, geom, binning, img, R, false);
}

extern "C" int fdgs_raster_fwd_capacity(void* stream, const fdgs_raster_params* p, void* geom, void* binning, void* img, uint32_t capacity,
                                        uint32_t* num_rendered_host, int32_t* radii, float* out_color, float* out_depth) {
    FDGS_REQUIRE(capacity > 0 && binning, "capacity mode needs a binning buffer of fdgs_binning_bytes(capacity) bytes, capacity > 0");
    // the count's way to the host: a store by the LAST workgroup of the projection kernel into the (pinned, device-visible) word -- it is in
    // host memory while the depth sort is still running; a copy node if the word is not device-visible (a 4-byte copy costs the stream
    // 4 us + a launch gap in the middle of the frame)
    void* dp = nullptr;
    const bool zero_copy = p && p->P > 0 && hipHostGetDevicePointer(&dp, num_rendered_host, 0) == hipSuccess && dp != nullptr;
    if (!zero_copy) (void)hipGetLastError();
    int rc = fdgs_preprocess_fwd_impl(stream, p, geom, radii, zero_copy ? reinterpret_cast<uint32_t*>(dp) : nullptr);
    if (rc) return rc;
    rc = bin_prepare_impl(stream, p, geom, num_rendered_host, false, zero_copy);
    if (rc) return rc;
    rc = bin_sort_impl(stream, p, geom, binning, img, capacity, true, nullptr);
    if (rc) return rc;
    return fdgs_render_fwd(stream, p, geom, binning, img, capacity, out_color, out_depth);
}

// Host side of the verified capacity path: wait until the pair count of a frame queued by fdgs_raster_fwd_capacity has reached its pinned
// host word (0xFFFFFFFF = not yet).  Spins on the word (it arrives with the projection kernel, long before the frame ends); every ~50 us
// the stream is queried so that a frame that finished without delivering a count is an error instead of a hang.
extern "C" int fdgs_pair_count_wait(void* stream_, const uint32_t* num_rendered_host, uint32_t* value) {
    FDGS_REQUIRE(num_rendered_host && value, "NULL pointer");
    const volatile uint32_t* w = num_rendered_host;
    hipStream_t stream = (hipStream_t)stream_;
    for (unsigned spins = 1;; spins++) {
        const uint32_t v = *w;
        if (v != 0xFFFFFFFFu) { *value = v; return FDGS_OK; }
        if ((spins & 0x3FFu) == 0) {
            const hipError_t e = hipStreamQuery(stream);
            if (e == hipSuccess) {
                const uint32_t v2 = *w;
                if (v2 != 0xFFFFFFFFu) { *value = v2; return FDGS_OK; }
                return fail(FDGS_E_INVALID, "%s", "the stream is idle and no pair count has arrived: was the frame queued with fdgs_raster_fwd_capacity on this stream?");
            }
            if (e != hipErrorNotReady) return fail(FDGS_E_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

extern "C" int fdgs_geom_field(void* geom, int P, int which, void** ptr) {
    FDGS_REQUIRE(geom && ptr && P >= 0, "bad arguments");
    GeomLayout gl = geom_layout(P);
    size_t off;
    switch (which) {
        case 0: off = gl.depth; break; case 1: off = gl.recA; break; case 2: off = gl.recB; break; case 3: off = gl.recC; break;
        case 4: off = gl.cov3D; break; case 5: off = gl.tiles; break; case 6: off = gl.clamped; break; case 7: off = gl.rect; break;
        case 8: off = gl.ids0; break; case 9: off = gl.offsets; break; case 10: off = gl.cullmask; break;
        default: return fail(FDGS_E_INVALID, "%s", "unknown geom field");
    }
    *ptr = at<char>(geom, off);
    return FDGS_OK;
}
extern "C" int fdgs_binning_field(void* binning, uint32_t R, int W, int H, int which, void** ptr) {
    (void)W; (void)H;
    FDGS_REQUIRE(binning && ptr, "bad arguments");
    BinLayout bl = bin_layout(R);
    if (which == 0) *ptr = at<char>(binning, bl.gid0);
    else if (which == 1) *ptr = at<char>(binning, bl.tile0);
    else return fail(FDGS_E_INVALID, "%s", "unknown binning field");
    return FDGS_OK;
}
extern "C" int fdgs_img_field(void* img, int W, int H, int which, void** ptr) {
    FDGS_REQUIRE(img && ptr && W > 0 && H > 0, "bad arguments");
    ImgLayout il = img_layout(W, H);
    if (which == 0) *ptr = at<char>(img, il.final_T);
    else if (which == 1) *ptr = at<char>(img, il.n_contrib);
    else if (which == 2) *ptr = at<char>(img, il.ranges);
    else if (which == 3) *ptr = at<char>(img, il.todo);
    else if (which == 4) *ptr = at<char>(img, il.order_f);
    else if (which == 5) *ptr = at<char>(img, il.order_b);
    else return fail(FDGS_E_INVALID, "%s", "unknown img field");
    return FDGS_OK;
}
