// api.hip -- library-wide entry points of libfdgs: error string, version, device query, L1 statistics kernel.
#include "common.h"

#include <map>
#include <mutex>
#include <string>
#include <strings.h>
#include <vector>

namespace fdgs {
thread_local char g_err[512] = {0};

bool g_timing_on = false;

// knob table: name (the FDGS_<NAME> environment variable, also the fdgs_tuning_set key in either spelling), field, default
namespace {
struct KnobDesc { const char* name; int Tuning::*field; int dflt; };
const KnobDesc kKnobs[] = {
    {"D1_FORM", &Tuning::d1_form, 0}, {"D1_WGS", &Tuning::d1_wgs, -1}, {"D1_SPLIT", &Tuning::d1_split, 1}, {"SKIP_DEAD", &Tuning::skip_dead, 1},
    {"D4_MFMA", &Tuning::d4_mfma, -1}, {"D4_ROWS_KB", &Tuning::d4_rows_kb, -1}, {"TILE_CULL", &Tuning::tile_cull, 1}, {"RBWD_PPL", &Tuning::rbwd_ppl, -1},
    {"TILE_ORDER", &Tuning::tile_order, 1}, {"ROW_COMPACT", &Tuning::row_compact, 1}, {"D2_FORM", &Tuning::d2_form, 0},
};
Tuning tuning_from_environment() {       // runs once, from the static initialiser below (library load)
    Tuning t{};
    for (const KnobDesc& k : kKnobs) {
        char env[64];
        snprintf(env, sizeof(env), "FDGS_%s", k.name);
        const char* v = getenv(env);
        t.*(k.field) = (v && *v) ? atoi(v) : k.dflt;
    }
    return t;
}
const KnobDesc* find_knob(const char* name) {
    if (!name) return nullptr;
    if (strncasecmp(name, "FDGS_", 5) == 0) name += 5;
    for (const KnobDesc& k : kKnobs)
        if (strcasecmp(name, k.name) == 0) return &k;
    return nullptr;
}
}  // namespace
static const Tuning g_tune_at_load = tuning_from_environment();
Tuning g_tune = g_tune_at_load;
namespace {
struct TimingRec { const char* name; hipEvent_t e0, e1; };
std::mutex g_tmu;
std::vector<TimingRec> g_pool;   // event pairs, reused between reports
size_t g_used = 0;
std::map<std::string, std::pair<long, double>> g_totals;
}  // namespace

void* timing_begin(const char* name, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_tmu);
    if (g_used == g_pool.size()) {
        TimingRec r{};
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return nullptr;
        g_pool.push_back(r);
    }
    TimingRec* r = &g_pool[g_used++];
    r->name = name;
    (void)hipEventRecord(r->e0, stream);
    return (void*)(g_used);  // 1-based index (the vector may reallocate)
}
void timing_end(void* rec, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_tmu);
    (void)hipEventRecord(g_pool[(size_t)rec - 1].e1, stream);
}

// [sum |a-b|, sum (a-b)^2, n] with optional dL/da = sign(a-b)*scale (utils/loss_utils.py:20-21 of the reference)
// One 1024-thread workgroup per CU, two float4 pairs in flight per thread (64 KB of requests per CU), ONE pair of atomics per workgroup:
// same-address float atomics retire at ~80 M/s on MI355X, so the 1 024 of the 512 x 256-thread form of rounds 1-3 were more
// than half of its 22 us.
constexpr int L1S_THREADS = 1024;
// ASSIGN: the workgroups leave their partial sums in `scratch` (float2 per workgroup) and take a ticket; the last one adds the partials in
// workgroup order and ASSIGNS acc[0..2] -- deterministic, no same-address float atomics, no zero fill of `acc` by the caller; the ticket
// counter (scratch word 2 * L1S_MAX_BLOCKS, zero at allocation) resets itself.
constexpr int L1S_MAX_BLOCKS = 256;
template <bool ASSIGN>
__global__ void __launch_bounds__(L1S_THREADS) l1_stats_kernel(size_t n, size_t n4, const float* __restrict__ a, const float* __restrict__ b,
                                                               float scale, float* __restrict__ grad, float* __restrict__ acc,
                                                               float* __restrict__ scratch) {
    float s1 = 0.f, s2 = 0.f;
    // n4 = number of 16-byte groups handled with float4 accesses (0 when a pointer is not 16-byte aligned), scalar tail
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float4* g4 = reinterpret_cast<float4*>(grad);
    auto one = [&](float x, float y) {
        const float d = x - y;
        s1 += fabsf(d); s2 += d * d;
        return d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
    };
    const size_t stride = (size_t)gridDim.x * L1S_THREADS;
    size_t i = (size_t)blockIdx.x * L1S_THREADS + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const float4 x0 = a4[i], y0 = b4[i], x1 = a4[i + stride], y1 = b4[i + stride];
        const float4 g0 = make_float4(one(x0.x, y0.x), one(x0.y, y0.y), one(x0.z, y0.z), one(x0.w, y0.w));
        const float4 g1 = make_float4(one(x1.x, y1.x), one(x1.y, y1.y), one(x1.z, y1.z), one(x1.w, y1.w));
        if (grad) { g4[i] = g0; g4[i + stride] = g1; }
    }
    for (; i < n4; i += stride) {
        const float4 x = a4[i], y = b4[i];
        const float4 g = make_float4(one(x.x, y.x), one(x.y, y.y), one(x.z, y.z), one(x.w, y.w));
        if (grad) g4[i] = g;
    }
    for (size_t k = n4 * 4 + (size_t)blockIdx.x * L1S_THREADS + threadIdx.x; k < n; k += stride) {
        const float g = one(a[k], b[k]);
        if (grad) grad[k] = g;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    __shared__ float w1[L1S_THREADS / 64], w2[L1S_THREADS / 64];
    if ((threadIdx.x & 63) == 0) { w1[threadIdx.x >> 6] = s1; w2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    __shared__ unsigned last_block;
    if (threadIdx.x == 0) {
        float t1 = 0.f, t2 = 0.f;
        for (int k = 0; k < L1S_THREADS / 64; k++) { t1 += w1[k]; t2 += w2[k]; }
        if constexpr (ASSIGN) {
            scratch[2 * blockIdx.x] = t1; scratch[2 * blockIdx.x + 1] = t2;
            __threadfence();
            unsigned* ticket = reinterpret_cast<unsigned*>(scratch + 2 * L1S_MAX_BLOCKS);
            last_block = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
        } else {
            atomicAdd(&acc[0], t1);
            atomicAdd(&acc[1], t2);
            if (blockIdx.x == 0) atomicAdd(&acc[2], (float)n);
        }
    }
    if constexpr (ASSIGN) {
        __syncthreads();
        if (last_block) {       // (uniform over the workgroup) the partials of all workgroups, one per thread, added in a fixed tree order
            __threadfence();
            float u1 = 0.f, u2 = 0.f;
            if (threadIdx.x < gridDim.x) {
                u1 = __hip_atomic_load(&scratch[2 * threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                u2 = __hip_atomic_load(&scratch[2 * threadIdx.x + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { u1 += __shfl_xor(u1, o, 64); u2 += __shfl_xor(u2, o, 64); }
            __syncthreads();
            if ((threadIdx.x & 63) == 0) { w1[threadIdx.x >> 6] = u1; w2[threadIdx.x >> 6] = u2; }
            __syncthreads();
            if (threadIdx.x == 0) {
                float v1 = 0.f, v2 = 0.f;
                for (int k = 0; k < L1S_MAX_BLOCKS / 64; k++) { v1 += w1[k]; v2 += w2[k]; }
                acc[0] = v1; acc[1] = v2; acc[2] = (float)n;
                *reinterpret_cast<unsigned*>(scratch + 2 * L1S_MAX_BLOCKS) = 0u;
            }
        }
    }
}
}  // namespace fdgs

using namespace fdgs;

extern "C" const char* fdgs_last_error(void) { return g_err; }
extern "C" int fdgs_abi_version(void) { return 6; }

extern "C" int fdgs_tuning_set(const char* name, int value) {
    const KnobDesc* k = find_knob(name);
    if (!k) return fail(FDGS_E_INVALID, "unknown tuning knob '%s'", name ? name : "(null)");
    g_tune.*(k->field) = value;
    return FDGS_OK;
}
extern "C" int fdgs_tuning_get(const char* name, int* value) {
    const KnobDesc* k = find_knob(name);
    if (!k || !value) return fail(FDGS_E_INVALID, "unknown tuning knob '%s'", name ? name : "(null)");
    *value = g_tune.*(k->field);
    return FDGS_OK;
}
extern "C" int fdgs_tuning_reset(void) {
    g_tune = g_tune_at_load;      // (the environment as it was when the library was loaded)
    return FDGS_OK;
}

extern "C" int fdgs_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(g_tmu);
    g_timing_on = on != 0;
    return FDGS_OK;
}

// Synchronises the device, folds the recorded event pairs into per-kernel totals and prints "name count total_ms" lines.
extern "C" int fdgs_timing_report(char* buf, size_t buflen, int reset) {
    FDGS_REQUIRE(buf && buflen > 0, "bad arguments");
    FDGS_HIP_CHECK(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_tmu);
    for (size_t i = 0; i < g_used; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_pool[i].e0, g_pool[i].e1) == hipSuccess) {
            auto& t = g_totals[g_pool[i].name];
            t.first += 1; t.second += ms;
        }
    }
    g_used = 0;
    size_t off = 0;
    buf[0] = 0;
    for (auto& kv : g_totals) {
        int n = snprintf(buf + off, buflen - off, "%s %ld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        if (n < 0 || (size_t)n >= buflen - off) break;
        off += (size_t)n;
    }
    if (reset) g_totals.clear();
    return FDGS_OK;
}

extern "C" int fdgs_device_arch(int dev, char* buf, size_t buflen) {
    FDGS_REQUIRE(buf && buflen > 0, "bad arguments");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || dev < 0 || dev >= n) return fail(FDGS_E_NOGPU, "%s", "no such HIP device");
    hipDeviceProp_t prop;
    FDGS_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, buflen, "%s", prop.gcnArchName);
    return FDGS_OK;
}

extern "C" int fdgs_l1_stats(void* stream_, size_t n, const float* a, const float* b, float grad_scale, float* grad_out_opt,
                             float* acc) {
    FDGS_REQUIRE(a && b && acc, "NULL pointer");
    if (n == 0) return FDGS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const bool aligned = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)grad_out_opt) & 15) == 0;
    const size_t n4 = aligned ? n / 4 : 0;
    int blocks = (int)(((aligned ? n / 4 : n) + L1S_THREADS - 1) / L1S_THREADS);
    if (blocks < 1) blocks = 1;
    if (blocks > 256) blocks = 256;   // one workgroup per CU; two atomics per workgroup on the same two words
    { FDGS_TIMED("l1_stats", stream); hipLaunchKernelGGL(l1_stats_kernel<false>, dim3(blocks), dim3(L1S_THREADS), 0, stream, n, n4, a, b, grad_scale, grad_out_opt, acc, (float*)nullptr); }
    FDGS_LAUNCH_CHECK("l1_stats", 0, stream);
    return FDGS_OK;
}

extern "C" int fdgs_l1_stats_scratch_bytes(size_t* bytes) {
    FDGS_REQUIRE(bytes, "bytes is NULL");
    *bytes = (2 * L1S_MAX_BLOCKS + 4) * sizeof(float);
    return FDGS_OK;
}

extern "C" int fdgs_l1_stats_assign(void* stream_, size_t n, const float* a, const float* b, float grad_scale, float* grad_out_opt,
                                    float* acc, void* scratch) {
    FDGS_REQUIRE(a && b && acc && scratch, "NULL pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const bool aligned = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)grad_out_opt) & 15) == 0;
    const size_t n4 = aligned ? n / 4 : 0;
    int blocks = (int)(((aligned ? n / 4 : n) + L1S_THREADS - 1) / L1S_THREADS);
    if (blocks < 1) blocks = 1;
    if (blocks > L1S_MAX_BLOCKS) blocks = L1S_MAX_BLOCKS;
    { FDGS_TIMED("l1_stats", stream); hipLaunchKernelGGL(l1_stats_kernel<true>, dim3(blocks), dim3(L1S_THREADS), 0, stream, n, n4, a, b, grad_scale, grad_out_opt, acc, reinterpret_cast<float*>(scratch)); }
    FDGS_LAUNCH_CHECK("l1_stats", 0, stream);
    return FDGS_OK;
}
