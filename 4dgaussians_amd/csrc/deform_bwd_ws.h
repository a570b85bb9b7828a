// deform_bwd_ws.h -- D2 (backward-data of the deformation MLP) in a WEIGHT-STATIONARY form (round 6).  Included by deform.hip inside namespace
// fdgs, after the 32-row kernel whose structures (BwdDev, head_k / head_off, ROW_PAD) and memory formats it shares: the row list, the compact
// gradient rows G [position][64], the saved relu(h1) / ReLU bits of the forward, DH1 / DHID / DFEAT indexed by list position -- D3 and D4 do
// not know which form ran.  Results differ from the 32-row kernel by summation order only.
//
// Why.  The 32-row kernel gives a wave its 32 rows and streams every weight of every layer past them (W1^T of five heads = 320 KB per 32
// rows from L2), one wave per SIMD with a long tail of stages nothing overlaps: 0.42 of the f32 MFMA peak on the shell scene, where it is
// the largest kernel of the frame (0.75 ms).  This is D1's medicine (deform_fwd_ws.h) for the transposed products: the weights stay, the
// rows move -- and because the reduction of dhid = sum_h W1_h^T dh1_h runs over the OUTPUT features of the heads' first layers, the split
// over the four waves is a split of K:
//     wave w owns first-layer features 32 w .. 32 w + 31 of every head: their W1 rows (all 128 columns: 5 x 64 registers, never reloaded),
//     their W2 columns, their slice of relu(h1), their columns of dW2;
//     per 16-row tile and head: dh1 = (W2^T G) . relu'(h1) for its 32 features (the D layout of v_mfma_f32_16x16x4_f32 IS the B layout of
//     the next product: no exchange), stored as DH1; dW2 += G^T relu(h1) for its 32 columns; then 64 MFMAs add W1^T dh1 into ITS partial
//     sum of dhid [128 x 16] (8 accumulator tiles);
//     ONE exchange per tile: the four partial dhid meet in LDS, every wave adds up the 32 features it owns, applies relu'(hidden), stores
//     DHID, and multiplies them with its rows of W0 (partial dfeat, summed by a short epilogue).
// Nothing streams from L2 in the tile loop except the tile's own data (16 gradient rows, 5 x 16 x 32 saved activations per wave, 32 bytes of
// ReLU bits per row), and that arrives by LDS-DMA (global_load_lds_dwordx4, guide: "glds"): 320 registers of weights + 32 of partial sums +
// 32 of dW2 sums leave no room for loads in flight.  A DMA writes its wave's LDS park lane-linearly, so the park's swizzle (conflict-free
// transposed reads for the dW2 product, whose reduction runs over the ROWS) is applied to the SOURCE addresses.  Every park is wave-private,
// filled a whole iteration before it is read, and retired by `s_waitcnt vmcnt(8)` -- at least 16 vector-memory operations (DMAs and stores)
// are issued between a DMA and the first read of its park, completion is in order, so "at most 8 in flight" covers it without ever waiting
// for the stores just issued (a vmcnt(0) would: deform_fwd_ws.h, lesson 1).  The compiler sees no load in the loop: the DMAs are asm.
//
// Applies to: row-list form (saved activations, ordered input), net_width 128, C*L in {32, 48}, all five heads (dynerf) or the three of position /
// scale / rotation (hypernerf: template parameter HM, the kernel is written for any set of three or more heads that starts with a k <= 4 head).
// Everything else runs the 32-row kernel (tuning knob d2_form = 32 forces it everywhere).

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {       // 64 lanes x 16 bytes -> LDS [lds_dst + 16 lane]
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {        // 64 lanes x 4 bytes -> LDS [lds_dst + 4 lane]
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// (a wave issues 25 - 26 vector-memory operations per tile; between a DMA and the first read of its park lie at least 19 of them: "at most 16
// in flight" retires the DMA and never waits for a store younger than half a tile -- stores take microseconds to be acknowledged)
// (with K of the five heads active the numbers are 4 K + 5 per tile and 4 K - 1 between a DMA and its park's first read: vmcnt(4 K - 4))
#ifdef WS2_X_NO_WAIT
#define WS2_WAIT_PARKS() do { } while (0)
#else
#define WS2_WAIT_PARKS() asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * NH - 4) : "memory")
#endif
#define WS2_WAIT_G() asm volatile("s_waitcnt vmcnt(10)" ::: "memory")      // (twelve operations follow the DMA of the gradient rows up to the barrier behind the second product)
#define WS2_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define WS2_LDS_DONE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WS2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// one MFMA, then up to five vector-ALU instructions, two LDS reads, one store -- sixteen times (8 MFMAs of the product + up to 8 of the slot)
#define WS2_IL1 __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0); \
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
#define WS2_INTERLEAVE WS2_IL1 WS2_IL1 WS2_IL1 WS2_IL1 WS2_IL1 WS2_IL1 WS2_IL1 WS2_IL1 WS2_IL1 WS2_IL1 WS2_IL1 WS2_IL1 WS2_IL1 WS2_IL1 WS2_IL1 WS2_IL1
#ifdef FDGS_PROFILE_D2WS
#define WS2_TICK(ph) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[ph] += t_ - pt; pt = t_; } while (0)
#else
#define WS2_TICK(ph) do { } while (0)
#endif

// The main products of the first WS2_ASM_MFMA heads are inline assembly with the A operand IN an accumulation register: the register
// allocator parks those heads' weights in AGPRs as spill slots and copies each one to a VGPR in front of its MFMA (v_accvgpr_read + two wait
// states, ~15 cycles out of every MFMA's shadow: 128 of a tile's 320); told that the operand lives there ("a"), it uses it in place: shell
// scene 0.614 -> 0.561 ms.  (Three heads measure the same as two; all five would need 320 AGPRs.)
#ifndef WS2_ASM_MFMA
#define WS2_ASM_MFMA 2
#endif

// HM: the active heads (bit hd; at least three, the SH head -- if active -- last by construction of the head numbering).  Arrays per head are
// indexed by the head's POSITION among the active ones (= its slab of saved activations / DH1, fdgs_deform_bwd's head_slot).
__host__ __device__ constexpr int ws2_popc(int m) { return m == 0 ? 0 : (m & 1) + ws2_popc(m >> 1); }
__host__ __device__ constexpr int ws2_nth(int m, int i, int hd = 0) { return m == 0 ? 0 : (m & 1) ? (i == 0 ? hd : ws2_nth(m >> 1, i - 1, hd + 1)) : ws2_nth(m >> 1, i, hd + 1); }

template <int FU, int HM>
__global__ void __launch_bounds__(256, 1) deform_bwd_data_ws_kernel(BwdDev d) {
    constexpr int W = 128, NH = ws2_popc(HM), LDW = W + 4, XLD = W + 4, F = 16 * FU, PLD = F + 4;
    static_assert(NH >= 3 && HM < 32 && ws2_nth(HM, 0) != FDGS_HEAD_SHS, "three or more heads, a k <= 4 head first");
    constexpr bool HAS_SH = (HM >> FDGS_HEAD_SHS) & 1;
    // (a TABLE of the active heads, evaluated here: called with a loop variable the recursion above is compiled, not folded -- a loop per call)
    constexpr int HID[FDGS_NUM_HEADS] = {ws2_nth(HM, 0), ws2_nth(HM, 1), ws2_nth(HM, 2), ws2_nth(HM, 3), ws2_nth(HM, 4)};
    const fdgs_deform_params& p = d.p;
    // parks (LDS-DMA destinations, lane-linear 16-byte chunks, swizzled through the source addresses):
    //   h1 [head][wave]: chunk (row g, c8 = features 4 c8 .. + 3 of the wave's 32) at slot 8 g + (c8 ^ (g & 7))
    //   g  [buffer]: the tile's 16 gradient rows, chunk (g, c16) at slot 16 g + (c16 ^ g) -- ONE copy per workgroup, a quarter DMA'd by each wave:
    //      readable after a barrier that every wave passed behind its own wait (the top of the next iteration)
    //   ri [stage][wave]: the tile's 16 row-list entries;  hm [wave]: uint4 (g, half t) at slot 16 t + g: the forward's ReLU bits of the trunk
    __shared__ __attribute__((aligned(16))) float park_h1[NH][4][512];
    __shared__ __attribute__((aligned(16))) float park_g[2][1024];
    __shared__ __attribute__((aligned(16))) uint32_t park_ri[2][4][64];
    __shared__ __attribute__((aligned(16))) uint32_t park_hm[4][256];
    __shared__ __attribute__((aligned(16))) float xp[4][16 * XLD];      // the waves' partial dhid [wave][row][feature]
    __shared__ __attribute__((aligned(16))) float pl[4][16 * PLD];      // the waves' partial dfeat [wave][row][input feature]
    __shared__ __attribute__((aligned(16))) float w2l[59 * LDW];        // the heads' W2, row-major (rows 0 .. 10: the k <= 4 heads, 11 .. 58: SH)
    __shared__ __attribute__((aligned(16))) float w0l[W * PLD];         // W0, row-major
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, q = lane >> 4;
    for (int hd = 0; hd < FDGS_NUM_HEADS; hd++)
        for (int i = tid; ((HM >> hd) & 1) && i < head_k(hd) * (W / 4); i += 256) {
            const int r = i / (W / 4), c4 = i - r * (W / 4);
            *reinterpret_cast<float4*>(w2l + (head_row0(hd) + r) * LDW + 4 * c4) = reinterpret_cast<const float4*>(p.w2[hd])[i];
        }
    for (int i = tid; i < W * (F / 4); i += 256) {
        const int r = i / (F / 4), c4 = i - r * (F / 4);
        *reinterpret_cast<float4*>(w0l + r * PLD + 4 * c4) = reinterpret_cast<const float4*>(p.w0)[i];
    }
    // ---- the stationary operands.  A-lane (i = n, k = q) of v_mfma_f32_16x16x4_f32:
    //   w1r[h][to][t][r] = W1_h[out = 32 w + 16 t + 4 q + r][in = 16 to + n]     dhid[in] += W1[out][in] dh1[out]
    // (W0 and the W2 are read from LDS where they are used: 320 registers of W1 + 64 of running sums are what the register file holds)
    float w1r[NH][8][2][4];
#pragma unroll
    for (int h = 0; h < NH; h++)
#pragma unroll
        for (int to = 0; to < 8; to++)
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) w1r[h][to][t][r] = p.w1[HID[h]][(size_t)(32 * w + 16 * t + 4 * q + r) * W + 16 * to + n];
    // sums that live in registers for the whole launch: dW2 of the wave's 32 columns (D-lane (n, q) register r: row 4 q + r of the column
    // group, column 32 w + 16 t + n) -- the four k <= 4 heads share one group (their 11 rows ARE columns 0 .. 10 of G) -- and db2
    f32x4 dws[2] = {zero4(), zero4()}, dwh[3][2];
#pragma unroll
    for (int og = 0; og < 3; og++) { dwh[og][0] = zero4(); dwh[og][1] = zero4(); }
    float dbs[4] = {0.f, 0.f, 0.f, 0.f};

    const int ntiles = (int)as_const(d.s.counters)[5] * 2;      // 16-row tiles of the (padded) row list
    const int G_ = (int)gridDim.x;
    const unsigned a_h1 = (unsigned)(size_t)&park_h1[0][w][0], a_g = (unsigned)(size_t)&park_g[0][256 * w], a_ri = (unsigned)(size_t)&park_ri[0][w][0],
                   a_hm = (unsigned)(size_t)&park_hm[w][0];
    constexpr unsigned S_H1 = 4 * 512 * 4, S_G = 1024 * 4, S_RI = 4 * 64 * 4;      // bytes between heads / buffers / stages
    // per-lane constants of the DMA source addresses
    const int h1_g[2] = {lane >> 3, 8 + (lane >> 3)};
    const int h1_c8[2] = {(lane & 7) ^ ((lane >> 3) & 7), (lane & 7) ^ ((lane >> 3) & 7)};      // (g & 7 is the same for g and g + 8)
    auto dma_ri = [&](int tile, int stage) {
        glds4(d.s.rows + (size_t)tile * 16 + n, a_ri + S_RI * stage);
    };
    auto dma_g = [&](int tile, int buf) {      // (this wave's quarter: rows 4 w .. 4 w + 3)
        const int slot = 64 * w + lane, g = slot >> 4, c16 = (slot & 15) ^ g;
        glds16(d.s.G + ((size_t)tile * 16 + g) * GCOLS + 4 * c16, a_g + S_G * buf);
    };
    auto dma_hm = [&](int stage) {
        const uint32_t row = park_ri[stage][w][n] & ~ROW_PAD;
        const int t = q & 1;
        glds16(reinterpret_cast<const uint4*>(d.sv_hmask) + ((size_t)(row >> 5) * 64 + 32 * t + (row & 31u)), a_hm);
    };
    auto dma_h1 = [&](int h, int stage) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const uint32_t row = park_ri[stage][w][h1_g[e]] & ~ROW_PAD;
            glds16(d.sv_h1 + ((size_t)h * d.s.Npad + row) * W + 32 * w + 4 * h1_c8[e], a_h1 + S_H1 * h + 1024u * e);
        }
    };

    // ---- one head's share of a tile up to the operand of the main product, cut into SLOTS: one wave per SIMD means nothing else runs while
    // an LDS read returns or a dependent MFMA chain drains, so every piece rides between two groups of eight independent MFMAs of the main
    // product of the head before it (a slot issues its LDS reads; the slot after it consumes them).
    // dh[t] (D-lane (n = row, q) register r) = dh1[row n][feature 32 w + 16 t + 4 q + r] -- also the B operand (k = q, step (t, r)) of W1^T dh1
    // (opaque copies of the lane coordinates per slot: the dozens of LDS offsets below are loop invariants, and hoisted out of the tile loop
    // they would each hold a register for the whole launch -- this kernel has none to spare; recomputed they are VALU work inside MFMA shadows)
    const int n_lane = n, q_lane = q;
    struct Rider {      // what a head's slots hand to each other (registers; every index is a compile-time constant after unrolling)
        float b, wa[8];
        float4 bv, hv[2];
        float bt[2][4], ar[4];
    };
// (NOT volatile: a volatile asm keeps its place between the stores and barriers around it, and everything computed from it queues up behind
// the last store of its slot; an asm that merely depends on the tile and on a key of its own cannot leave the iteration, cannot be merged
// with its siblings, and moves freely inside the slot)
#define WS2_COORDS_K(key) int n = n_lane, q = q_lane; asm("" : "+v"(n), "+v"(q) : "s"(tile), "s"(key));
    // Six per-lane BYTE offsets stay in registers for the whole launch; every LDS address of the slots is one of them, at most one XOR with a
    // constant (the parks' swizzles are XORs) and an immediate.  (Whole addresses would be loop invariants too -- ~70 of them, hoisted out of
    // the tile loop and held in registers this kernel does not have; recomputed from the lane id they were ~50 vector-ALU instructions per
    // slot, a block that runs in the open behind the last MFMA of its group.  Opaque copies per slot keep the XORs from being hoisted.)
    //   h1 park, natural (row n, features 16 t + 4 q ..):            (o_hn ^ 64 t)                    + 8192 h
    //   h1 park, transposed (row 4 q + c, feature 16 t + n):         ((o_ht ^ 64 t) ^ 16 c) + 128 c   + 8192 h
    //   G park, natural SH (row n, columns 16 + 16 s + 4 q ..):      (o_gn ^ 64 (1 + s))              + 4096 buffer
    //   G park, transposed (row 4 q + c, column 16 cg + n):          (o_gt ^ (16 c + 64 cg)) + 256 c  + 4096 buffer
    //   W2 of the SH head (row 11 + 16 s + 4 q + c, feature 32 w + 16 t + n):   o_w2 + 4 LDW (16 s + c) + 64 t
    const int o_hn = 16 * (8 * n + (q ^ (n & 7))) + 2048 * w, o_ht = 512 * q + 4 * n + 64 * (q & 1) + 2048 * w;
    const int o_gn = 256 * n + 16 * ((n & 12) | ((q ^ n) & 3)), o_gt = 1088 * q + 4 * n;
    const int o_w2 = 4 * ((11 + 4 * q) * LDW + 32 * w + n);
    const char* const lds_h1 = reinterpret_cast<const char*>(&park_h1[0][0][0]);
    const char* const lds_g = reinterpret_cast<const char*>(&park_g[0][0]);
    const char* const lds_w2 = reinterpret_cast<const char*>(&w2l[0]);
#define WS2_LDF(base, off) (*reinterpret_cast<const float*>((base) + (off)))
#define WS2_LDF4(base, off) (*reinterpret_cast<const float4*>((base) + (off)))
#define WS2_OPAQUE(v, key) asm("" : "+v"(v) : "s"(tile), "s"(key))
    // slots of a k <= 4 head: 0 reads | 1 dh1 | 2 mask, DH1, transposed reads | 3 dW2
    auto small_slot = [&](int slot, int h, int pi, int tile, int gbuf, f32x4* dh, Rider& R) {      // (h: the head, pi: its position among the active ones)
        const int k = head_k(h), off = head_off(h);
        if (slot == 0) {
            WS2_COORDS_K(64 + 8 * h)
            const int o = q < k ? q : k - 1, col = off + o;
            R.b = WS2_LDF(lds_g, 4096 * gbuf + 256 * n + 16 * ((col >> 2) ^ n) + 4 * (col & 3));
            const int wo = 4 * ((head_row0(h) + o) * LDW + 32 * w + n);
            R.wa[0] = WS2_LDF(lds_w2, wo); R.wa[1] = WS2_LDF(lds_w2, wo + 64);
            int hn = o_hn;
            WS2_OPAQUE(hn, 65 + 8 * h);
            R.hv[0] = WS2_LDF4(lds_h1, hn + 8192 * pi); R.hv[1] = WS2_LDF4(lds_h1, (hn ^ 64) + 8192 * pi);
        } else if (slot == 1) {
            const float b = q_lane < k ? R.b : 0.f;
            dh[0] = mm16(R.wa[0], b, zero4());
            dh[1] = mm16(R.wa[1], b, zero4());
        } else if (slot == 2) {
            // dW2[col][feature] += sum_rows G[row][col] relu(h1)[row][feature]: the reduction runs over the ROWS -- both operands are read
            // TRANSPOSED from the parks (A-lane (i = n: column, k = q) step c: row 4 q + c; B-lane (j = n: feature 16 t + n, k = q) likewise)
            int ht = o_ht, gt = o_gt;
            WS2_OPAQUE(ht, 66 + 8 * h); WS2_OPAQUE(gt, 67 + 8 * h);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                R.bt[0][c] = WS2_LDF(lds_h1, (ht ^ (16 * c)) + 128 * c + 8192 * pi);
                R.bt[1][c] = WS2_LDF(lds_h1, (ht ^ (64 + 16 * c)) + 128 * c + 8192 * pi);
                R.ar[c] = WS2_LDF(lds_g, (gt ^ (16 * c)) + 256 * c + 4096 * gbuf);
            }
            float* slab = d.s.DH1 + ((size_t)pi * d.s.Npad + (size_t)tile * 16 + n_lane) * W + 32 * w + 4 * q_lane;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const float4 hv = R.hv[t];
                dh[t][0] = hv.x > 0.f ? dh[t][0] : 0.f; dh[t][1] = hv.y > 0.f ? dh[t][1] : 0.f;
                dh[t][2] = hv.z > 0.f ? dh[t][2] : 0.f; dh[t][3] = hv.w > 0.f ? dh[t][3] : 0.f;
#ifndef WS2_X_NO_DH1_STORES
                *reinterpret_cast<float4*>(slab + 16 * t) = make_float4(dh[t][0], dh[t][1], dh[t][2], dh[t][3]);
#endif
            }
        } else {
            const bool mine = n_lane >= off && n_lane < off + k;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (pi == 0) dbs[0] += R.ar[c];       // (db2 of the four k <= 4 heads: all of columns 0 .. 15, once per tile)
                const float a = mine ? R.ar[c] : 0.f;
#ifndef WS2_X_NO_DW2
                dws[0] = mm16(a, R.bt[0][c], dws[0]);
                dws[1] = mm16(a, R.bt[1][c], dws[1]);
#else
                dbs[0] += a + R.bt[0][c] + R.bt[1][c];
#endif
            }
        }
    };
    // slots of the 48-row SH head: 0 reads | 1 - 3 dh1, 16 rows of W2 each | 4 mask, DH1, transposed reads | 5 - 7 dW2, 16 columns each
    auto sh_slot = [&](int slot, int tile, int gbuf, f32x4* dh, Rider& R) {
        constexpr int pi = NH - 1;      // (the SH head is the last active one)
        auto read_dh1_operands = [&](int s) {
            int gn = o_gn, w2 = o_w2;
            WS2_OPAQUE(gn, 128 + 2 * s); WS2_OPAQUE(w2, 129 + 2 * s);
            R.bv = WS2_LDF4(lds_g, (gn ^ (64 * (1 + s))) + 4096 * gbuf);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                R.wa[2 * c] = WS2_LDF(lds_w2, w2 + 4 * LDW * (16 * s + c)); R.wa[2 * c + 1] = WS2_LDF(lds_w2, w2 + 4 * LDW * (16 * s + c) + 64);
            }
        };
        auto read_g_t = [&](int og) {
            int gt = o_gt;
            WS2_OPAQUE(gt, 136 + og);
#pragma unroll
            for (int c = 0; c < 4; c++) R.ar[c] = WS2_LDF(lds_g, (gt ^ (16 * c + 64 * (1 + og))) + 256 * c + 4096 * gbuf);
        };
        if (slot == 0) {
            read_dh1_operands(0);
            int hn = o_hn;
            WS2_OPAQUE(hn, 140);
            R.hv[0] = WS2_LDF4(lds_h1, hn + 8192 * pi); R.hv[1] = WS2_LDF4(lds_h1, (hn ^ 64) + 8192 * pi);
        } else if (slot <= 3) {
            if (slot == 1) { dh[0] = zero4(); dh[1] = zero4(); }
            const float bb[4] = {R.bv.x, R.bv.y, R.bv.z, R.bv.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                dh[0] = mm16(R.wa[2 * c], bb[c], dh[0]);
                dh[1] = mm16(R.wa[2 * c + 1], bb[c], dh[1]);
            }
            if (slot < 3) read_dh1_operands(slot);
        } else if (slot == 4) {
            int ht = o_ht;
            WS2_OPAQUE(ht, 141);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                R.bt[0][c] = WS2_LDF(lds_h1, (ht ^ (16 * c)) + 128 * c + 8192 * pi);
                R.bt[1][c] = WS2_LDF(lds_h1, (ht ^ (64 + 16 * c)) + 128 * c + 8192 * pi);
            }
            read_g_t(0);
            float* slab = d.s.DH1 + ((size_t)pi * d.s.Npad + (size_t)tile * 16 + n_lane) * W + 32 * w + 4 * q_lane;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const float4 hv = R.hv[t];
                dh[t][0] = hv.x > 0.f ? dh[t][0] : 0.f; dh[t][1] = hv.y > 0.f ? dh[t][1] : 0.f;
                dh[t][2] = hv.z > 0.f ? dh[t][2] : 0.f; dh[t][3] = hv.w > 0.f ? dh[t][3] : 0.f;
#ifndef WS2_X_NO_DH1_STORES
                *reinterpret_cast<float4*>(slab + 16 * t) = make_float4(dh[t][0], dh[t][1], dh[t][2], dh[t][3]);
#endif
            }
        } else {
            const int og = slot - 5;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                dbs[1 + og] += R.ar[c];
#ifndef WS2_X_NO_DW2
                dwh[og][0] = mm16(R.ar[c], R.bt[0][c], dwh[og][0]);
                dwh[og][1] = mm16(R.ar[c], R.bt[1][c], dwh[og][1]);
#else
                dbs[0] += R.bt[0][c] + R.bt[1][c];
#endif
            }
            if (og < 2) read_g_t(og + 1);
        }
    };

#ifdef FDGS_PROFILE_D2WS
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pt = __builtin_amdgcn_s_memtime();
#endif
    int tile = blockIdx.x;
    if (tile < ntiles) {
        // ---- prologue: the first tile's parks, the second tile's row-list entries
        const int t1 = tile + G_ < ntiles ? tile + G_ : tile;
        dma_ri(tile, 0); dma_ri(t1, 1);
        WS2_WAIT_ALL();
        dma_g(tile, 0); dma_hm(0);
#pragma unroll
        for (int h = 0; h < NH; h++) dma_h1(h, 0);
        WS2_WAIT_ALL();
        __syncthreads();      // (w2l, w0l; every wave's quarter of the first tile's gradient rows)
        f32x4 dh_cur[2];
        int it = 0, prev = -1;
        uint32_t hbits_prev = 0u;
        // ---- the exchange of the previous tile, in three pieces that ride inside this tile's first products (the barrier at the top of the
        // iteration completed xp; the one behind the second product completes pl and frees xp):
        //   A: every wave adds up the partial dhid of the 32 hidden features it owns, applies relu'(hidden), stores DHID;
        //   B: ... multiplies them with its rows of W0 (partial dfeat);     C: the partial dfeat are added and stored
        f32x4 dhid[2];
        auto finish_a = [&](int t) {
            WS2_COORDS_K(200 + t)
            float4 s = *reinterpret_cast<const float4*>(&xp[0][n * XLD + 32 * w + 16 * t + 4 * q]);
#pragma unroll
            for (int ww = 1; ww < 4; ww++) {
                const float4 v = *reinterpret_cast<const float4*>(&xp[ww][n * XLD + 32 * w + 16 * t + 4 * q]);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const uint32_t b = hbits_prev >> (4 * t);
            dhid[t][0] = (b & 1u) ? s.x : 0.f; dhid[t][1] = (b & 2u) ? s.y : 0.f; dhid[t][2] = (b & 4u) ? s.z : 0.f; dhid[t][3] = (b & 8u) ? s.w : 0.f;
            *reinterpret_cast<float4*>(d.s.DHID + ((size_t)prev * 16 + n) * W + 32 * w + 4 * q + 16 * t) = make_float4(dhid[t][0], dhid[t][1], dhid[t][2], dhid[t][3]);
        };
        auto finish_b = [&]() {
            WS2_COORDS_K(202)
            f32x4 df[FU];
#pragma unroll
            for (int ct = 0; ct < FU; ct++) df[ct] = zero4();
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int ct = 0; ct < FU; ct++) df[ct] = mm16(w0l[(32 * w + 16 * t + 4 * q + r) * PLD + 16 * ct + n], dhid[t][r], df[ct]);
#pragma unroll
            for (int ct = 0; ct < FU; ct++)
                *reinterpret_cast<float4*>(&pl[w][n * PLD + 16 * ct + 4 * q]) = make_float4(df[ct][0], df[ct][1], df[ct][2], df[ct][3]);
        };
        auto finish_c = [&]() {
            if (tid < 16 * (F / 4)) {
                const int g = tid / (F / 4), c = tid - g * (F / 4);
                float4 s = *reinterpret_cast<const float4*>(&pl[0][g * PLD + 4 * c]);
#pragma unroll
                for (int ww = 1; ww < 4; ww++) {
                    const float4 v = *reinterpret_cast<const float4*>(&pl[ww][g * PLD + 4 * c]);
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
                *reinterpret_cast<float4*>(d.s.DFEAT + ((size_t)prev * 16 + g) * F + 4 * c) = s;
            }
        };
#pragma nounroll      // (nor peeled: the first iteration differs by `prev < 0` only)
        for (; tile < ntiles; tile += G_, it ^= 1) {
            WS2_TICK(0);
            const bool has_next = tile + G_ < ntiles;
            const int nxt = has_next ? tile + G_ : tile, nxt2 = tile + 2 * G_ < ntiles ? tile + 2 * G_ : nxt;
            // (every wave waits for ITS quarter of this tile's gradient rows before the barrier that opens them to the others)
            WS2_WAIT_PARKS();
            if (prev >= 0) WS2_BARRIER();            // every wave's partial sums of tile `prev` are in xp
            WS2_TICK(1);
            // this tile's ReLU bits of the trunk: D-lane (n, q), tile t, register r = hidden feature 32 w + 16 t + 4 q + r = word r, bit q + 4 w of the
            // uint4 (row, half t) in the forward's layout (deform_fwd_ws.h: trunk); the next tile's rows of the saved activations (this wave's
            // two 16-byte chunks per head are chunks of the same two rows for every head)
            uint32_t hbits = 0u;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const uint4 hm = *reinterpret_cast<const uint4*>(&park_hm[w][4 * (16 * t + n)]);
                const uint32_t sh = (uint32_t)(q + 4 * w);
                hbits |= (((hm.x >> sh) & 1u) | (((hm.y >> sh) & 1u) << 1) | (((hm.z >> sh) & 1u) << 2) | (((hm.w >> sh) & 1u) << 3)) << (4 * t);
            }
            uint32_t h1_off[2];      // float offset of the wave's chunk inside a head's slab
#pragma unroll
            for (int e = 0; e < 2; e++) h1_off[e] = (park_ri[it ^ 1][w][h1_g[e]] & ~ROW_PAD) * (uint32_t)W + (uint32_t)(32 * w + 4 * h1_c8[e]);
            auto dma_h1_next = [&](int h) {
#ifdef WS2_X_NO_DMA
                return;
#endif
#pragma unroll
                for (int e = 0; e < 2; e++) glds16(d.sv_h1 + (size_t)h * d.s.Npad * W + h1_off[e], a_h1 + S_H1 * h + 1024u * e);
            };
            WS2_LDS_DONE();
            // the next tile's gradient rows and ReLU bits, the row-list entries of the tile after it (stage `it` held this tile's: no longer needed)
            dma_g(nxt, it ^ 1); dma_hm(it ^ 1); dma_ri(nxt2, it);
            Rider R;
            // the first head's slots have no product to ride in (inside the previous tile's last product they cost 20 registers that are not there)
            constexpr int H0 = HID[0];
            small_slot(0, H0, 0, tile, it, dh_cur, R); small_slot(1, H0, 0, tile, it, dh_cur, R); small_slot(2, H0, 0, tile, it, dh_cur, R); small_slot(3, H0, 0, tile, it, dh_cur, R);
            WS2_TICK(2);
            f32x4 acc[8];
#pragma unroll
            for (int to = 0; to < 8; to++) acc[to] = zero4();
#pragma unroll
            for (int h = 0; h < NH; h++) {
                // park h is free (its slots ran inside the previous product): the next tile's rows of this head
                WS2_LDS_DONE();
                dma_h1_next(h);
                f32x4 dh_nxt[2];
                // the main product of head h: 8 groups of 8 MFMAs on independent accumulators; between the groups the slots of the next head and
                // the pieces of the previous tile's exchange
#pragma unroll
                for (int grp = 0; grp < 8; grp++) {
                    const int t = grp >> 2, r = grp & 3;
                    if (grp == 0) WS2_WAIT_PARKS();
#pragma unroll
                    for (int to = 0; to < 8; to++) {
#if WS2_ASM_MFMA > 0
                        // (A operand straight from an accumulation register: the compiler parks these weights in AGPRs anyway and copies each
                        // one to a VGPR in front of its MFMA -- a VALU instruction + two wait states out of every MFMA's shadow)
                        // (the compiler's hazard recogniser does not look into the statement: the B operand may have been written by the VALU
                        // instruction right in front of a group -- two wait states in front of the first MFMA of every group)
                        if (h < WS2_ASM_MFMA) {
                            if (to == 0) asm("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[to]) : "a"(w1r[h][to][t][r]), "v"(dh_cur[t][r]));
                            else asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[to]) : "a"(w1r[h][to][t][r]), "v"(dh_cur[t][r]));
                        }
                        else
#endif
                        acc[to] = mm16(w1r[h][to][t][r], dh_cur[t][r], acc[to]);
                    }
                    // (h: position of the product's head; its rider is the head at position h + 1: the SH head's eight slots or a k <= 4 head's four)
                    if (h + 1 < NH) {
                        if (HAS_SH && h + 1 == NH - 1) sh_slot(grp, tile, it, dh_nxt, R);
                        else if (grp < 4) small_slot(grp, HID[h + 1 < NH ? h + 1 : 0], h + 1, tile, it, dh_nxt, R);
                    }
                    if (prev >= 0) {
                        if (h == 0 && (grp == 4 || grp == 5)) finish_a(grp - 4);
                        if (h == 1 && grp == 4) finish_b();
                        if (h == 2 && grp == 4) finish_c();
                    }
                    // the slot's instructions go BETWEEN the MFMAs (a wave alone on its SIMD issues an MFMA every 32 cycles: four or five other
                    // instructions fit behind each, a block of fifty behind the eighth runs in the open)
                    WS2_INTERLEAVE
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (h == 0) WS2_TICK(5);
                // xp may be overwritten; the partial dfeat are complete; every wave's quarter of the NEXT tile's gradient rows has arrived (16 operations ago)
                if (h == 1) { WS2_WAIT_G(); WS2_BARRIER(); }
                if (h == 1) WS2_TICK(6);
                if (h == 3) WS2_TICK(7);
                if (h + 1 < NH) { dh_cur[0] = dh_nxt[0]; dh_cur[1] = dh_nxt[1]; }
            }
            WS2_TICK(3);
            // this wave's partial dhid into the exchange (the second barrier freed it)
#pragma unroll
            for (int to = 0; to < 8; to++)
                *reinterpret_cast<float4*>(&xp[w][n * XLD + 16 * to + 4 * q]) = make_float4(acc[to][0], acc[to][1], acc[to][2], acc[to][3]);
            hbits_prev = hbits;
            prev = tile;
            WS2_TICK(4);
        }
        WS2_BARRIER();
        finish_a(0); finish_a(1); finish_b();
        WS2_BARRIER();
        finish_c();
    }
    WS2_WAIT_ALL();
    // ---- flush: the wave's columns of dW2 (one global atomic per element and workgroup), db2 from wave 0
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int f = 32 * w + 16 * t + n;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int col = 4 * q + r;
            const int hd = col < 3 ? 0 : col < 6 ? 1 : col < 10 ? 2 : 3;
            if (col <= 10 && ((HM >> hd) & 1) && dws[t][r] != 0.f) atomicAdd(&d.d_w2[hd][(size_t)(col - head_off(hd)) * W + f], dws[t][r]);
#pragma unroll
            for (int og = 0; og < 3; og++)
                if (HAS_SH && dwh[og][t][r] != 0.f) atomicAdd(&d.d_w2[FDGS_HEAD_SHS][(size_t)(16 * og + col) * W + f], dwh[og][t][r]);
        }
    }
#pragma unroll
    for (int gq = 0; gq < 4; gq++) dbs[gq] = sum_lane_groups(dbs[gq]);
    if (w == 0 && q == 0) {
        const int hd = n < 3 ? 0 : n < 6 ? 1 : n < 10 ? 2 : 3;
        if (n <= 10 && ((HM >> hd) & 1) && dbs[0] != 0.f) atomicAdd(&d.d_b2[hd][n - head_off(hd)], dbs[0]);
#pragma unroll
        for (int og = 0; og < 3; og++)
            if (HAS_SH && dbs[1 + og] != 0.f) atomicAdd(&d.d_b2[FDGS_HEAD_SHS][16 * og + n], dbs[1 + og]);
    }
#ifdef FDGS_PROFILE_D2WS
    if (d.prof && lane == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(&d.prof[i], pacc[i]);
        atomicAdd(&d.prof[8], 1ull);
    }
#endif
}
