// Unrolled solver-in-the-loop training step, inference roll-out, loss and TF-style Adam.
//
// Replaces the msteps graph of /root/reference/karman-2d/karman_train.py:397-457 (forward
// unroll :399-426, loss :428-436, optimizer :449-457) and the roll-out loop of
// karman_apply.py:138-158.  The per-step order (step -> CNN correction -> add -> loss) and
// the reverse sweep follow SURVEY.md appendix C.9.
#include "common.hpp"
#include <stdlib.h>
#include <string.h>
#include <vector>

thread_local char g_sol_err[512] = "";

int sol_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_sol_err, sizeof(g_sol_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* sol_last_error(void) { return g_sol_err; }
extern "C" int sol_version(void) { return 215; }
// sizes of the ABI structs: the ctypes mirror in _lib.py checks them at load time
extern "C" int sol_abi_sizes(int32_t* karman_cfg, int32_t* burgers_cfg, int32_t* train_cfg) {
    if (karman_cfg) *karman_cfg = (int32_t)sizeof(sol_karman_cfg);
    if (burgers_cfg) *burgers_cfg = (int32_t)sizeof(sol_burgers_cfg);
    if (train_cfg) *train_cfg = (int32_t)sizeof(sol_train_cfg);
    return SOL_OK;
}

// ---- options --------------------------------------------------------------------------------
SolOptions& sol_opt() {
    static SolOptions o = [] {
        SolOptions d{};
        d.conv_precision = 0; d.conv_split3 = 0; d.conv_r3 = 1; d.conv_thin = 1; d.conv_bww32 = 1;
        d.correct_fuse = 1; d.bww_fuse = 1; d.bww_chunk = 0; d.bww_side = 1; d.streams = 1;
        d.density_mode = 0; d.cpt = 0; d.dbg_skip = 0; d.step_prof = 0; d.cnn_persistent = 0; d.graph_stream = 0; d.k3d_tile = 0; d.k3d_fused_tf = 1; d.k3d_conv_fused = 1; d.k3d_conv_rows = 8; d.conv_dx = 11; d.k3d_mfma_tf = 1; d.conv_thin_valu = 1; d.seed_fuse = 1; d.fwd_bands = 1; d.conv_thin_t3 = 1; d.k3d_bww_jobs = 2; d.k3d_conv_persist = 0; d.k3d_adj_tile = 1;
        return d;
    }();
    return o;
}

namespace {
struct OptName { const char* name; int SolOptions::*field; int lo, hi; };
const OptName OPT_NAMES[] = {
    {"conv_precision", &SolOptions::conv_precision, 0, 2}, {"conv_split3", &SolOptions::conv_split3, 0, 1},
    {"conv_r3", &SolOptions::conv_r3, 0, 1}, {"conv_thin", &SolOptions::conv_thin, 0, 1}, {"conv_bww32", &SolOptions::conv_bww32, 0, 1},
    {"correct_fuse", &SolOptions::correct_fuse, 0, 1}, {"bww_fuse", &SolOptions::bww_fuse, 0, 1},
    {"bww_chunk", &SolOptions::bww_chunk, 0, 1024}, {"bww_side", &SolOptions::bww_side, 0, 1}, {"streams", &SolOptions::streams, 1, 8},
    {"density_mode", &SolOptions::density_mode, 0, 2}, {"cpt", &SolOptions::cpt, 0, 16}, {"dbg_skip", &SolOptions::dbg_skip, 0, 1 << 30},
    {"step_prof", &SolOptions::step_prof, 0, 1}, {"cnn_persistent", &SolOptions::cnn_persistent, 0, 1},
    {"graph_stream", &SolOptions::graph_stream, 0, 1}, {"k3d_tile", &SolOptions::k3d_tile, 0, 1}, {"k3d_fused_tf", &SolOptions::k3d_fused_tf, 0, 1}, {"k3d_conv_fused", &SolOptions::k3d_conv_fused, 0, 1}, {"k3d_conv_rows", &SolOptions::k3d_conv_rows, 3, 8},
    {"conv_dx", &SolOptions::conv_dx, 0, 15}, {"conv_thin_valu", &SolOptions::conv_thin_valu, 0, 2}, {"k3d_mfma_tf", &SolOptions::k3d_mfma_tf, 0, 1},
    {"seed_fuse", &SolOptions::seed_fuse, 0, 1}, {"fwd_bands", &SolOptions::fwd_bands, 0, 1}, {"conv_thin_t3", &SolOptions::conv_thin_t3, 0, 1}, {"k3d_bww_jobs", &SolOptions::k3d_bww_jobs, 0, 2}, {"k3d_conv_persist", &SolOptions::k3d_conv_persist, 0, 1}, {"k3d_adj_tile", &SolOptions::k3d_adj_tile, 0, 1},
};
}  // namespace

extern "C" int sol_set_option(const char* name, int32_t value) {
    SOL_REQUIRE(name != nullptr, "sol_set_option: NULL name");
    for (const OptName& o : OPT_NAMES)
        if (!strcmp(o.name, name)) {
            SOL_REQUIRE(value >= o.lo && value <= o.hi, "sol_set_option: %s must be in [%d, %d] (got %d)", name, o.lo, o.hi, value);
            sol_opt().*(o.field) = value;
            return SOL_OK;
        }
    return sol_set_error(SOL_ERR_ARG, "sol_set_option: unknown option '%s'", name);
}

extern "C" int sol_get_option(const char* name, int32_t* value) {
    SOL_REQUIRE(name != nullptr && value != nullptr, "sol_get_option: NULL argument");
    for (const OptName& o : OPT_NAMES)
        if (!strcmp(o.name, name)) { *value = sol_opt().*(o.field); return SOL_OK; }
    return sol_set_error(SOL_ERR_ARG, "sol_get_option: unknown option '%s'", name);
}

// ---- launch profiler ------------------------------------------------------------------------
bool g_sol_prof_on = false;
namespace {
struct ProfRec { const char* name; hipEvent_t a, b; };
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
size_t g_prof_used = 0;
}  // namespace

bool sol_prof_events(const char* name, hipEvent_t* a, hipEvent_t* b) {
    while (g_prof_pool.size() < g_prof_used + 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return false;
        g_prof_pool.push_back(e);
    }
    *a = g_prof_pool[g_prof_used++];
    *b = g_prof_pool[g_prof_used++];
    g_prof_recs.push_back(ProfRec{name, *a, *b});
    return true;
}

extern "C" int sol_prof_begin(void) {
    g_prof_recs.clear();
    g_prof_used = 0;
    g_sol_prof_on = true;
    return SOL_OK;
}

// Stops profiling, waits for the device and sums the records per kernel name.  names: max_classes x 64 chars.
// Returns the number of classes (<= max_classes) or < 0.
extern "C" int sol_prof_end(int32_t max_classes, char* names, double* total_us, int32_t* calls) {
    g_sol_prof_on = false;
    SOL_REQUIRE(max_classes >= 1 && names && total_us && calls, "sol_prof_end: bad arguments");
    SOL_HIP_CHECK(hipDeviceSynchronize());
    int n = 0;
    for (const ProfRec& r : g_prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) { (void)hipGetLastError(); continue; }
        int k = 0;
        while (k < n && strncmp(names + (size_t)k * 64, r.name, 63)) ++k;
        if (k == n) {
            if (n == max_classes) continue;
            strncpy(names + (size_t)k * 64, r.name, 63);
            names[(size_t)k * 64 + 63] = 0;
            total_us[k] = 0.0; calls[k] = 0;
            ++n;
        }
        total_us[k] += (double)ms * 1e3;
        calls[k] += 1;
    }
    g_prof_recs.clear();
    g_prof_used = 0;
    return n;
}

namespace {

constexpr int NL = 12;   // conv layers of model_mars_moon
int pick_bww_chunk(int ms);
inline int layer_cin(int l) { return l == 0 ? 3 : 32; }
inline int layer_cout(int l) { return l == NL - 1 ? 2 : 32; }
inline int64_t layer_koff(int l) {
    int64_t off = 0;
    for (int k = 0; k < l; ++k) off += 25 * layer_cin(k) * layer_cout(k) + layer_cout(k);
    return off;
}

// ---- velocity += CNN correction (to_staggered pad, karman_train.py:88-90,413-426) + l2 loss ----
__global__ void k_correct_loss(float* __restrict__ vy, float* __restrict__ vx, const float* __restrict__ O,
                               const float* __restrict__ gt_vy, const float* __restrict__ gt_vx,
                               float s0, float s1, float l0, float l1, unsigned long long* __restrict__ loss, int B, int Y, int X, int tr) {
    __shared__ float red[64];
    const int N = Y * X, nVy = (Y + 1) * X, nVx = Y * (X + 1), XP = X + 1;
    auto cell = [&](int j, int i) { return tr ? i * Y + j : j * X + i; };      // O is [B][X][Y][2] in transposed CNN mode
    const int total = B * (nVy + nVx);
    float l = 0.f;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        if (e < B * nVy) {
            const int b = e / nVy, k = e - b * nVy;
            float v = vy[e];
            if (k < N) v += s0 * O[((size_t)b * N + (tr ? cell(k / X, k % X) : k)) * 2];
            vy[e] = v;
            if (gt_vy) { const float d = (gt_vy[e] - v) / l0; l += 0.5f * d * d; }
        } else {
            const int e2 = e - B * nVy;
            const int b = e2 / nVx, k = e2 - b * nVx;
            const int j = k / XP, i = k - j * XP;
            float v = vx[e2];
            if (i < X) v += s1 * O[((size_t)b * N + cell(j, i)) * 2 + 1];
            vx[e2] = v;
            if (gt_vx) { const float d = (gt_vx[e2] - v) / l1; l += 0.5f * d * d; }
        }
    }
    if (loss) {                                    // one exact integer add per workgroup (bit reproducible; loss_add_exact)
        const float s = block_sum(l, red, 0);
        if (threadIdx.x == 0) loss_add_exact(s, loss);
    }
}

// the per-step losses out of their exact accumulators (one thread per unrolled step)
__global__ void k_loss_finish(const unsigned long long* __restrict__ acc, float* __restrict__ loss_steps, int ms) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ms) loss_steps[i] = loss_acc_value(acc + (size_t)i * SOL_LOSS_ACC_WORDS);
}

// ---- backward seed: g_prd = g_next + (prd - gt)/(std^2 msteps);  dO = std * g_prd on cells ----
__global__ void k_seed(float* __restrict__ gvy, float* __restrict__ gvx, const float* __restrict__ vy,
                       const float* __restrict__ vx, const float* __restrict__ gt_vy, const float* __restrict__ gt_vx,
                       float s0, float s1, float l0, float l1, float inv_m, float* __restrict__ dO4, float* __restrict__ dO2,
                       int first, int B, int Y, int X, int tr) {
    const int N = Y * X, nVy = (Y + 1) * X, nVx = Y * (X + 1), XP = X + 1;
    auto cell = [&](int j, int i) { return tr ? i * Y + j : j * X + i; };
    const int total = B * (nVy + nVx);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        if (e < B * nVy) {
            const int b = e / nVy, k = e - b * nVy;
            float g = (vy[e] - gt_vy[e]) * inv_m / (l0 * l0);
            if (!first) g += gvy[e];
            gvy[e] = g;
            if (k < N) {
                const size_t c = (size_t)b * N + (tr ? cell(k / X, k % X) : k);
                dO4[c * 4] = s0 * g;
                dO2[c * 2] = s0 * g;
            }
        } else {
            const int e2 = e - B * nVy;
            const int b = e2 / nVx, k = e2 - b * nVx;
            const int j = k / XP, i = k - j * XP;
            float g = (vx[e2] - gt_vx[e2]) * inv_m / (l1 * l1);
            if (!first) g += gvx[e2];
            gvx[e2] = g;
            if (i < X) {
                const size_t c = (size_t)b * N + cell(j, i);
                dO4[c * 4 + 1] = s1 * g;
                dO2[c * 2 + 1] = s1 * g;
            }
        }
    }
}

// ---- zero-fill / copy of several regions in ONE launch -------------------------------------------
// The training graph contains NO memset / memcpy nodes: on ROCm 7.2 a replayed hipGraph whose memset nodes follow a
// large device-to-host copy on the same stream filled with garbage instead of zeros (tools/debug_sol32c.py: identical
// garbage losses on consecutive replays after a 200 KB .cpu() of a trainer buffer; eager hipMemsetAsync is fine).
// Kernels are immune, and one launch replaces four to seven nodes.
struct MemJobs {
    uint32_t* dst[8];
    const uint32_t* src[8];     // nullptr: fill with zero
    size_t words[8];
    const uint32_t* once[8];    // nullptr, or a control word: the job is skipped when it already holds `once_magic`
    uint32_t once_magic;
    int n;
};
__global__ void k_mem_jobs(MemJobs j) {
    for (int k = 0; k < j.n; ++k) {
        if (j.once[k] && *j.once[k] == j.once_magic) continue;       // uniform: written by k_chain_tick in an EARLIER launch
        uint32_t* d = j.dst[k];
        const uint32_t* s = j.src[k];
        const size_t n = j.words[k];
        const size_t n4 = (((uintptr_t)d | (uintptr_t)s) & 15) == 0 ? n / 4 : 0;      // 16-byte pieces when both are aligned
        for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x)
            reinterpret_cast<uint4*>(d)[e] = s ? reinterpret_cast<const uint4*>(s)[e] : make_uint4(0u, 0u, 0u, 0u);
        for (size_t e = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
            d[e] = s ? s[e] : 0u;
    }
}
// control words of the persistent CNN chain's hand-off regions: [0] epoch (one tick per use of the regions), [1] "zeroed" magic
__global__ void k_chain_tick(uint32_t* ctl, uint32_t magic) {
    if (threadIdx.x == 0) { ctl[0] = ctl[0] + 1u; ctl[1] = magic; }
}
struct MemList {
    MemJobs j{};
    void zero(void* p, size_t bytes) { copy(p, nullptr, bytes); }
    void zero_once(void* p, size_t bytes, const uint32_t* ctl_magic, uint32_t magic) {      // zero unless *ctl_magic == magic
        if (j.n < 8) { j.once[j.n] = ctl_magic; j.once_magic = magic; }
        copy(p, nullptr, bytes);
    }
    void copy(void* d, const void* s, size_t bytes) {
        if (!d || bytes == 0 || j.n >= 8) return;
        j.dst[j.n] = static_cast<uint32_t*>(d); j.src[j.n] = static_cast<const uint32_t*>(s); j.words[j.n] = bytes / 4; ++j.n;
    }
    int launch(hipStream_t hs) {
        if (j.n == 0) return SOL_OK;
        size_t mx = 0;
        for (int k = 0; k < j.n; ++k) mx = j.words[k] > mx ? j.words[k] : mx;
        const int grid = (int)((mx / 4 + 255) / 256 < 1 ? 1 : ((mx / 4 + 255) / 256 > 512 ? 512 : (mx / 4 + 255) / 256));
        SOL_LAUNCH(k_mem_jobs, dim3(grid), dim3(256), 0, hs, j);
        SOL_LAUNCH_CHECK();
        return SOL_OK;
    }
};

}  // namespace
// Device-to-device copy / zero fill of 32-bit words AS A KERNEL (k_mem_jobs): what a host uses inside a stream capture instead of
// hipMemcpyAsync / hipMemsetAsync (torch: tensor.copy_, clone, multi-block reductions), whose graph nodes sol_graph_check refuses.
extern "C" int sol_copy_words(void* stream, void* dst, const void* src, int64_t nwords) {
    SOL_REQUIRE(dst && nwords >= 0 && (((uintptr_t)dst | (uintptr_t)src) & 3) == 0, "sol_copy_words: dst (4-byte aligned), nwords >= 0; src NULL = zero fill");
    if (nwords == 0 || dst == src) return SOL_OK;
    MemList m;
    m.copy(dst, src, (size_t)nwords * 4);
    return m.launch((hipStream_t)stream);
}
namespace {
// ---- clock probe (bench.py: is THIS box slow, or did the kernels get slower?) ----------------------------------------------------
// Every SIMD of the chip runs one wave with a chain of `iters` DEPENDENT v_mfma_f32_16x16x16_f16 (a fixed number of matrix-pipe cycles
// per instruction whatever the operands are; the operands are non-trivial so that the power state is the one a convolution sees).
// Workgroup 0 reports the chain's duration on the constant 100 MHz clock (s_memrealtime) and on s_memtime, so that the host can quote
// ns per dependent MFMA -- a number that moves with the engine clock the box actually holds under matrix load and with nothing else.
typedef _Float16 probe_h4 __attribute__((ext_vector_type(4)));
typedef float probe_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_clock_probe(unsigned long long* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    probe_h4 a = {(_Float16)(0.5f + 0.001f * lane), (_Float16)(seed), (_Float16)(-0.25f), (_Float16)(0.125f * (lane & 3))};
    probe_h4 b = {(_Float16)(1.0f - 0.002f * lane), (_Float16)(0.75f), (_Float16)(seed * 0.5f), (_Float16)(-0.5f)};
    probe_f4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = wall_clock64(), c0 = clock64();
#pragma unroll 8
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);
    const unsigned long long t1 = wall_clock64(), c1 = clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = c1 - c0; out[2] = (unsigned long long)iters; }
    if (acc[0] == 1.2345e-30f) out[3] = 1;      // never true: keeps the chain alive
}
}  // namespace
// Memory-latency probe: ONE lane walks `steps` dependent 4-byte loads through a zero-filled buffer of `nlines` 128-byte lines (power of two) in a
// pseudo-random (LCG) order -- every load's address depends on the previous load's value.  ns per load = the round trip a kernel's FIRST
// loads pay (L2 is invalid at every kernel boundary): working sets of 64 MB / 1 GB answer "MALL hit" / "HBM".
__global__ void __launch_bounds__(64) k_latency_probe(const unsigned* __restrict__ buf, unsigned nlines_mask, int steps, unsigned long long* out) {
    if (threadIdx.x != 0) return;
    unsigned idx = 12345u & nlines_mask;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < steps; ++i) {
        const unsigned v = __builtin_nontemporal_load(buf + (size_t)idx * 32);
        idx = (idx * 1664525u + 1013904223u + v) & nlines_mask;
    }
    const unsigned long long t1 = wall_clock64();
    out[0] = t1 - t0; out[1] = (unsigned long long)steps; out[2] = idx;
}
extern "C" int sol_latency_probe(void* stream, const uint32_t* buf, int64_t nlines, int32_t steps, uint64_t* out4) {
    SOL_REQUIRE(buf && out4 && nlines >= 2 && (nlines & (nlines - 1)) == 0 && nlines <= (1ll << 31) && steps >= 1, "sol_latency_probe: buf (zero filled, nlines x 128 B, nlines a power of two), steps >= 1");
    SOL_LAUNCH(k_latency_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, buf, (unsigned)(nlines - 1), steps, reinterpret_cast<unsigned long long*>(out4));
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}
// (s_memtime, s_memrealtime) at this point of the stream: two stamps bracket a region whose AVERAGE shader clock is then
// 100 MHz x d(memtime) / d(memrealtime) -- the clock the device actually held while the region's kernels ran (bench.py: the timed steps)
// (every XCD has its own s_memtime counter with its own offset: the stamp is taken once per XCD -- 32 workgroups are dealt round robin to the
//  eight of them -- and filed under the XCD's id, so that differences are formed between stamps of ONE counter)
__global__ void k_clock_stamp(unsigned long long* out16) {
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        out16[2 * xcc] = clock64();
        out16[2 * xcc + 1] = wall_clock64();
    }
}
extern "C" int sol_clock_stamp(void* stream, uint64_t* out16) {
    SOL_REQUIRE(out16 != nullptr, "sol_clock_stamp: out16 (device, sixteen 64-bit words: {s_memtime, s_memrealtime} per XCD)");
    SOL_LAUNCH(k_clock_stamp, dim3(32), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(out16));
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}
extern "C" int sol_clock_probe(void* stream, uint64_t* out4, int32_t iters) {
    SOL_REQUIRE(out4 && iters >= 1 && iters <= (1 << 24), "sol_clock_probe: out4 (device, four 64-bit words), 1 <= iters <= 2^24");
    SOL_LAUNCH(k_clock_probe, dim3(256), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(out4), iters, 0.3f);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}
namespace {

// ---- transposed CNN mode ------------------------------------------------------------------------
// The split-precision convolution / weight-gradient kernels want image rows of 64 pixels.  A 64 x 32 grid (the reference's own
// training recipe, karman-2d/Makefile:78-80) has rows of 32 -- but columns of 64, and conv(x^T, w^T) = conv(x, w)^T: the
// whole CNN runs on TRANSPOSED images [B][X][Y][C] with tap-transposed packed weights (sol_pack_jobs), the weight gradients
// are transposed back in their reduction.  Only the two ends touch the solver's layout: the solver kernels write the features
// and read the feature gradient in the transposed cell order (sol_karman_feat_transposed; two k_transpose_cells launches per
// unrolled step until round 4), and the correction / loss / seed kernels index the transposed cell order.
__host__ __device__ inline bool cnn_transposed(int Y, int X) { return X % 64 != 0 && Y % 64 == 0; }

// ---- TF1 AdamOptimizer ------------------------------------------------------------------
__global__ void k_tensor_scale(const float* __restrict__ g, float* __restrict__ scale, int64_t off, int64_t n,
                               float clip_norm) {
    __shared__ float red[64];
    float s = 0.f;
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) { const float v = g[off + e]; s += v * v; }
    s = block_sum(s, red, 0);
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(s);
        *scale = nrm > clip_norm ? clip_norm / nrm : 1.f;   // tf.clip_by_norm
    }
}

struct AdamTensors {
    int n;
    int64_t off[40];
};

__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       int64_t n, float lr_t, float b1, float b2, float eps, const float* __restrict__ scale, AdamTensors T) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        float gr = g[e];
        if (scale) {
            int t = 0;
            while (t + 1 < T.n && e >= T.off[t + 1]) ++t;
            gr *= scale[t];
        }
        const float mi = b1 * m[e] + (1.f - b1) * gr;
        const float vi = b2 * v[e] + (1.f - b2) * gr * gr;
        m[e] = mi;
        v[e] = vi;
        p[e] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// ---- workspace carving ------------------------------------------------------------------
struct Ws {
    // sizes in floats
    size_t nVy, nVx, N, cells, st_vy, st_vx, st_d;
    float *vy, *vx, *d;            // [msteps][B][...] states after step i (index i = state i+1)
    float *svy, *svx;              // [msteps] saved post-diffusion velocity
    float *feat;                   // [msteps][cells][4]  (cell order of the CNN: transposed in transposed CNN mode)
    float *acts;                   // [msteps][11][cells][32]
    float *O;                      // [cells][2]
    float *gA, *gB;                // [cells][32]
    float *dO4, *dO2, *dF;         // [cells][4], [msteps][cells][2], [cells][2]
    float *dzb;                    // [msteps][11][cells][32]: pre-activation gradients kept for the batched weight gradient
    uint32_t *amax_act, *amax_dz;  // [msteps][11][SOL_AMAX_SLOTS]: absmax slots of every 32-channel activation / gradient tensor
    size_t amax_words;             // (both arrays are contiguous: one memset per training step)
    uint32_t* chain_flags;         // [msteps (or ROLLOUT_AMAX_SETS)][2 passes][chain_words]: hand-off flags of the persistent CNN launches
    size_t chain_words;            // words per chain launch (0: chain not usable for this shape)
    uint32_t* chain_ctl;           // [0] epoch, [1] magic ("the regions have been zeroed for THIS configuration")
    uint32_t chain_magic;
    float *gvy[2], *gvx[2];
    float *wf[NL], *wb[NL], *bias[NL];
    float *part[NL];
    size_t part_floats[NL];
    float *adam_scale;
    uint32_t* xchg;                // hand-off region of the band-split forward solver launches (k_karman_fwd_bands): zero between uses
    size_t xchg_words;
    unsigned long long* loss_acc;  // [msteps][SOL_LOSS_ACC_WORDS] exact accumulators of the per-step losses (loss_add_exact)
    size_t total_floats;
};

constexpr int ROLLOUT_AMAX_SETS = 8;      // roll-out: absmax slot sets used round robin, zeroed by ONE memset per 8 steps

size_t carve_ws(const sol_train_cfg* c, float* base, Ws& w, bool training) {
    const int B = c->karman.B, Y = c->karman.Y, X = c->karman.X;
    const int ms = training ? c->msteps : 1;
    w.N = (size_t)Y * X;
    w.cells = (size_t)B * w.N;
    w.nVy = (size_t)(Y + 1) * X;
    w.nVx = (size_t)Y * (X + 1);
    w.st_vy = B * w.nVy; w.st_vx = B * w.nVx; w.st_d = w.cells;
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += align_up(n, 64); return p; };
    w.vy = take(ms * w.st_vy); w.vx = take(ms * w.st_vx); w.d = take(ms * w.st_d);
    w.svy = take(ms * w.st_vy); w.svx = take(ms * w.st_vx);
    w.feat = take(ms * w.cells * 4);
    w.acts = take((size_t)ms * 11 * w.cells * 32);
    w.O = take(w.cells * 2);
    w.gA = take(w.cells * 32); w.gB = take(w.cells * 32);
    w.dO4 = take(w.cells * 4); w.dO2 = take((size_t)ms * w.cells * 2); w.dF = take(w.cells * 2);
    w.dzb = take(training ? (size_t)ms * 11 * w.cells * 32 : 0);
    w.amax_words = (size_t)(training ? ms : ROLLOUT_AMAX_SETS) * 11 * SOL_AMAX_SLOTS;
    w.amax_act = reinterpret_cast<uint32_t*>(take(w.amax_words));
    w.amax_dz = reinterpret_cast<uint32_t*>(take(training ? w.amax_words : 0));
    // hand-off regions of the persistent CNN chain: sized by the image height the CNN kernels see (X in transposed-CNN mode) and
    // carved only while the option is on (like bww_chunk, the option then enters the workspace size: a workspace sized with
    // the option off is refused, not overrun, when it is switched on later)
    w.chain_words = sol_opt().cnn_persistent ? sol_cnn_chain_flag_words(B, cnn_transposed(Y, X) ? X : Y, 10) : 0;
    w.chain_flags = reinterpret_cast<uint32_t*>(take((training ? (size_t)ms * 2 : (size_t)ROLLOUT_AMAX_SETS) * w.chain_words));   // roll-out: forward pass only
    w.chain_ctl = reinterpret_cast<uint32_t*>(take(64));
    w.chain_magic = 0x5017C4A1u ^ ((uint32_t)B * 2654435761u + (uint32_t)Y * 40503u + (uint32_t)X * 97u + (uint32_t)ms * 7u + (training ? 1u : 0u));
    for (int k = 0; k < 2; ++k) { w.gvy[k] = take(w.st_vy); w.gvx[k] = take(w.st_vx); }
    for (int l = 0; l < NL; ++l) {
        const int cin = layer_cin(l), cout = layer_cout(l);
        w.wf[l] = take(sol_conv5x5_packed_floats(cin == 3 ? 4 : cin, cout, SOL_CONV_FWD));
        w.wb[l] = take(sol_conv5x5_packed_floats(cout == 2 ? 4 : cout, cin, SOL_CONV_BWD_DATA));
        w.bias[l] = take(32);
        w.part_floats[l] = training ? sol_bww_batched_ws_floats(pick_bww_chunk(ms), B, cnn_transposed(Y, X) ? X : Y, cin == 3 ? 4 : cin, cout) : 0;
        w.part[l] = take(w.part_floats[l]);
    }
    w.adam_scale = take(64);
    w.xchg_words = sol_karman_fwd_bands_words(B);       // (carved whatever the option says: a workspace stays valid when fwd_bands is switched)
    w.xchg = reinterpret_cast<uint32_t*>(take(w.xchg_words));
    w.loss_acc = reinterpret_cast<unsigned long long*>(take((size_t)ms * SOL_LOSS_ACC_WORDS * 2));   // 64-bit words
    w.total_floats = off;
    return off * sizeof(float);
}

int pack_all(const sol_train_cfg* c, void* stream, const float* params, Ws& w, bool bwd) {
    // every layer, forward and backward-data form, fp32 + split-bf16 + split-fp16 sections and the padded biases: ONE launch
    const float* src[24]; float* out[24]; float* bo[24]; const float* bi[24];
    int cin[24], cout[24], mode[24], n = 0;
    for (int l = 0; l < NL; ++l) {
        const int ci = layer_cin(l), co = layer_cout(l);
        const int64_t koff = layer_koff(l), boff = koff + 25 * ci * co;
        src[n] = params + koff; out[n] = w.wf[l]; bo[n] = w.bias[l]; bi[n] = params + boff; cin[n] = ci; cout[n] = co; mode[n] = SOL_CONV_FWD; ++n;
        if (bwd) {   // run-conv of backward-data: input channels = forward cout, output channels = forward cin
            src[n] = params + koff; out[n] = w.wb[l]; bo[n] = nullptr; bi[n] = nullptr; cin[n] = co; cout[n] = ci; mode[n] = SOL_CONV_BWD_DATA; ++n;
        }
    }
    return sol_pack_jobs((hipStream_t)stream, n, src, out, bo, bi, cin, cout, mode, cnn_transposed(c->karman.Y, c->karman.X) ? 1 : 0);
}

inline float in_s0(const sol_train_cfg* c) { return c->in_std_v0 > 0.f ? c->in_std_v0 : c->std_v0; }
inline float in_s1(const sol_train_cfg* c) { return c->in_std_v1 > 0.f ? c->in_std_v1 : c->std_v1; }
inline float out_s0(const sol_train_cfg* c) { return c->out_std_v0 > 0.f ? c->out_std_v0 : c->std_v0; }
inline float out_s1(const sol_train_cfg* c) { return c->out_std_v1 > 0.f ? c->out_std_v1 : c->std_v1; }

// CNN forward (model_mars_moon, karman_train.py:101-138).  acts: 11 buffers [cells][32].
// amax: [11][SOL_AMAX_SLOTS] absmax slots of act[0..10] (zeroed by the caller): every producer publishes max|y| and every
// 32-channel consumer derives its fp16 operand scale from it (sol_conv5x5_scaled).
// scope guard: feature I/O of the solver launches issued inside the scope in the CNN's transposed cell order
struct FeatOrder {
    int old; bool active;
    explicit FeatOrder(bool tr) : old(sol_karman_feat_transposed(tr ? 1 : 0)), active(true) {}
    void restore() { if (active) { sol_karman_feat_transposed(old); active = false; } }
    ~FeatOrder() { restore(); }
};

// corr != nullptr: the last layer applies its output to the velocity and accumulates the loss (sol_conv5x5_correct) instead of storing O
struct Correct { float *vy, *vx; const float *gt_vy, *gt_vx; unsigned long long* loss; };
int net_forward(const sol_train_cfg* c, void* s, const Ws& w, const float* feat, float* const* act, float* O, uint32_t* amax, const Correct* corr = nullptr,
                uint32_t* chain_flags = nullptr) {
    const bool tr = cnn_transposed(c->karman.Y, c->karman.X);      // then `feat` and every CNN tensor are [B][X][Y][C]
    const int B = c->karman.B, Y = tr ? c->karman.X : c->karman.Y, X = tr ? c->karman.Y : c->karman.X;
    const float sl = c->lrelu_slope;
    auto am = [&](int k) { return amax ? amax + (size_t)k * SOL_AMAX_SLOTS : nullptr; };
    if (int e = sol_conv5x5_scaled(s, feat, w.wf[0], w.bias[0], nullptr, nullptr, act[0], B, Y, X, 4, 32, SOL_EPI_LRELU, sl, nullptr, am(0))) return e;
    if (amax && chain_flags && sol_cnn_chain_usable(B, Y, X)) {
        // layers 1..10 as ONE persistent launch (cnn_chain.hip)
        ChainLayer L[10];
        for (int k = 0; k < 5; ++k) {
            L[2 * k] = ChainLayer{sol_conv_packed_wsh(w.wf[1 + 2 * k], 32), w.bias[1 + 2 * k], nullptr, nullptr, act[1 + 2 * k], am(1 + 2 * k), SOL_EPI_LRELU};
            L[2 * k + 1] = ChainLayer{sol_conv_packed_wsh(w.wf[2 + 2 * k], 32), w.bias[2 + 2 * k], act[2 * k], nullptr, act[2 + 2 * k], am(2 + 2 * k), SOL_EPI_LRELU};
        }
        if (int e = sol_cnn_chain_launch((hipStream_t)s, L, 10, act[0], chain_flags, w.chain_ctl, B, Y, X, sl)) return e;
    } else
    for (int k = 0; k < 5; ++k) {
        const float* h = act[2 * k];
        if (int e = sol_conv5x5_scaled(s, h, w.wf[1 + 2 * k], w.bias[1 + 2 * k], nullptr, nullptr, act[1 + 2 * k], B, Y, X, 32, 32, SOL_EPI_LRELU, sl,
                                       am(2 * k), am(1 + 2 * k))) return e;
        if (int e = sol_conv5x5_scaled(s, act[1 + 2 * k], w.wf[2 + 2 * k], w.bias[2 + 2 * k], h, nullptr, act[2 + 2 * k], B, Y, X, 32, 32, SOL_EPI_LRELU, sl,
                                       am(1 + 2 * k), am(2 + 2 * k))) return e;
    }
    if (corr)
        return sol_conv5x5_correct(s, act[10], w.wf[11], w.bias[11], B, Y, X, am(10), corr->vy, corr->vx, corr->gt_vy, corr->gt_vx,
                                   out_s0(c), out_s1(c), c->std_v0, c->std_v1, corr->loss, tr ? 1 : 0);
    return sol_conv5x5_scaled(s, act[10], w.wf[11], w.bias[11], nullptr, nullptr, O, B, Y, X, 32, 2, SOL_EPI_NONE, sl, am(10), nullptr);
}

int check_train_cfg(const sol_train_cfg* c) {
    SOL_REQUIRE(c != nullptr, "train cfg is NULL");
    SOL_REQUIRE(c->msteps >= 1 && c->msteps <= 1024, "msteps out of range (%d)", c->msteps);
    SOL_REQUIRE(c->std_v0 > 0.f && c->std_v1 > 0.f && c->std_re > 0.f,
                "std values must be > 0 (the reference divides by them, karman_train.py:416-421; -n 1 gives std(Re)=0)");
    const int Y = c->karman.Y, X = c->karman.X;
    SOL_REQUIRE((Y * X) % 64 == 0, "Y*X must be a multiple of 64");
    return SOL_OK;
}

}  // namespace

extern "C" int sol_mars_moon_layer(int32_t l, int64_t* kernel_off, int64_t* bias_off, int32_t* cin, int32_t* cout) {
    SOL_REQUIRE(l >= 0 && l < NL, "layer index out of range (%d)", l);
    const int64_t k = layer_koff(l);
    if (kernel_off) *kernel_off = k;
    if (bias_off) *bias_off = k + 25 * layer_cin(l) * layer_cout(l);
    if (cin) *cin = layer_cin(l);
    if (cout) *cout = layer_cout(l);
    return SOL_OK;
}

extern "C" size_t sol_rollout_workspace_bytes(const sol_train_cfg* cfg) {
    if (!cfg) return 0;
    Ws w;
    return carve_ws(cfg, nullptr, w, false);
}

namespace {

// ---- sub-batch chains on their own HIP streams ----------------------------------------------
// The simulations of a batch are independent through the whole unroll (only the weight gradient is
// summed at the end), and the solver kernels occupy one CU per simulation for hundreds of
// microseconds while the conv kernels want the whole chip.  The batch is therefore split into S
// chains (S | B; env SOL_STREAMS, default 1) that run on S streams: the solver step of one simulation overlaps
// with the convolutions of the others, and the prologue/epilogue of one conv launch with the MFMA
// phase of another.  Each chain has its own workspace slice and weight-gradient partials.
struct StreamPool {
    hipStream_t s[8];
    hipEvent_t fork, join[8];
    bool ok;
};
StreamPool* pool() {
    static StreamPool p = [] {
        StreamPool q{};
        q.ok = true;
        for (int k = 0; k < 8; ++k) {
            q.ok &= hipStreamCreateWithFlags(&q.s[k], hipStreamNonBlocking) == hipSuccess;
            q.ok &= hipEventCreateWithFlags(&q.join[k], hipEventDisableTiming) == hipSuccess;
        }
        q.ok &= hipEventCreateWithFlags(&q.fork, hipEventDisableTiming) == hipSuccess;
        return q;
    }();
    return &p;
}

// Default: ONE weight-gradient launch per layer over all unrolled steps, after the sweep (chunk = msteps).
// SOL_BWW_CHUNK=n covers n steps per launch on a side stream, meant to fill the ~250 CUs that idle while the
// one-workgroup-per-simulation solver adjoint runs; measured on MI355X/ROCm 7.2 it does NOT overlap
// (39.8 ms vs 38.3 ms per step, eager and graph alike), so it is off.
int pick_bww_chunk(int ms) {
    int ch = sol_opt().bww_chunk;
    if (ch <= 0 || ch > ms) ch = ms;
    return ch;
}

int pick_chains(int B) {
    int want = sol_opt().streams;   // default 1; measured on MI355X/ROCm 7.2: concurrent chains are SLOWER (57 -> 105..167 ms/step), see DESIGN.md
    if (want < 1) want = 1;
    if (want > 8) want = 8;
    while (B % want) --want;
    return want;
}

struct TrainIO {
    const float *params, *d0, *vy0, *vx0, *re, *active, *inflow, *bcv, *bcm, *gt_vy, *gt_vx;
    int64_t bc_stride;
    float *loss_steps, *d_final, *vy_final, *vx_final;
    int32_t *iters_fwd, *iters_bwd;
};

// The weight gradients of the 32 -> 32 layers of unrolled step i ride in the solver-adjoint launch of step i (k_karman_bwd_bww at 128x64,
// k_karman_bwd_bww_small at 64x32 -- there the CNN runs on the transposed images, rows = B * X).  Returns the image rows per gradient
// workgroup (32 / 16: sized to last about as long as the adjoint), or 0 where the per-layer launches after the sweep are used.
// run_chain and the reduce at the end (which must know the partial layout) both ask here.
int train_fused_rb(const sol_train_cfg* c, const Ws& w, int ms) {
    const sol_karman_cfg* kc = &c->karman;
    const bool tr = cnn_transposed(kc->Y, kc->X);
    const int cY = tr ? kc->X : kc->Y, cX = tr ? kc->Y : kc->X;
    if (pick_bww_chunk(ms) != ms || cX != 64 || !sol_opt().bww_fuse || sol_opt().conv_precision != 0) return 0;
    const int rb = sol_karman_bwd_fusable(kc) ? 32 : (sol_karman_bwd_fusable_small(kc) ? 16 : 0);
    if (!rb || (kc->B * cY) % rb != 0 || sol_bww_step_ws_floats(kc->B, cY, rb) > w.part_floats[1]) return 0;
    return rb;
}

// forward unroll + reverse sweep of the simulations [b0, b0 + c.karman.B) on stream hs
int run_chain(const sol_train_cfg* c, const Ws& w, const Ws& shared, int Btot, int b0, hipStream_t hs, const TrainIO& io) {
    void* stream = hs;
    const sol_karman_cfg* kc = &c->karman;
    const int B = kc->B, Y = kc->Y, X = kc->X, ms = c->msteps;
    const bool tr = cnn_transposed(Y, X);              // CNN tensors are [B][X][Y][C]; cY x cX = image shape the CNN kernels see
    const int cY = tr ? X : Y, cX = tr ? Y : X;
    const float fscale[3] = {1.f / in_s0(c), 1.f / in_s1(c), 1.f / c->std_re};
    const int egrid = (int)((w.st_vy + w.st_vx + 255) / 256);
    const size_t gVy = (size_t)Btot * w.nVy, gVx = (size_t)Btot * w.nVx;       // per-step stride of the gt frames
    const float* d0 = io.d0 + (size_t)b0 * w.N;
    const float* vy0 = io.vy0 + (size_t)b0 * w.nVy;
    const float* vx0 = io.vx0 + (size_t)b0 * w.nVx;
    const float* re = io.re + b0;
    const float* bcv = io.bcv + (size_t)b0 * io.bc_stride;
    const float* bcm = io.bcm + (size_t)b0 * io.bc_stride;
    const float* gt_vy = io.gt_vy + (size_t)b0 * w.nVy;
    const float* gt_vx = io.gt_vx + (size_t)b0 * w.nVx;
    const float sl = c->lrelu_slope;
    Ws wn = w;                         // packed weights / padded biases are shared by all chains
    for (int l = 0; l < NL; ++l) { wn.wf[l] = shared.wf[l]; wn.wb[l] = shared.wb[l]; wn.bias[l] = shared.bias[l]; }

    // ---------------- forward unroll ----------------
    const bool dens_inline = sol_opt().density_mode == 1;
    const bool dens_fused = !dens_inline && io.d_final && sol_karman_bwd_fusable(kc) && sol_opt().density_mode == 0;
    // 64x32: one density advection per launch of the REVERSE sweep (k_karman_bwd_bww_small), msteps launches for msteps advections
    const bool dens_ride_bwd = !dens_inline && !dens_fused && io.d_final && sol_opt().density_mode == 0 && sol_karman_bwd_fusable_small(kc) &&
                               train_fused_rb(c, w, ms) != 0;
    for (int i = 0; i < ms; ++i) {
        const float* din = i == 0 ? d0 : w.d + (size_t)(i - 1) * w.st_d;
        const float* vyin = i == 0 ? vy0 : w.vy + (size_t)(i - 1) * w.st_vy;
        const float* vxin = i == 0 ? vx0 : w.vx + (size_t)(i - 1) * w.st_vx;
        float* dcur = w.d + (size_t)i * w.st_d;
        float* vycur = w.vy + (size_t)i * w.st_vy;
        float* vxcur = w.vx + (size_t)i * w.st_vx;
        float* feat_cnn = w.feat + (size_t)i * w.cells * 4;
        // Transposed CNN mode: the solver kernels write the features (and read the feature gradient) in the CNN's cell order themselves
        // (sol_karman_feat_transposed) -- it was a k_transpose_cells launch each way per unrolled step: 63 launches of 4.3 us in the 64x32 recipe.
        float* feat = feat_cnn;
        FeatOrder feat_order(tr);
        // The passive density leaves the critical path.  With the direct-solver kernels the density advection of step
        // i-1 (it only needs that step's saved velocity) rides in the solver launch of step i as B extra workgroups;
        // otherwise the whole chain is one launch after the unroll (sol_density_chain below).
        float* svy_i = w.svy + (size_t)i * w.st_vy;
        float* svx_i = w.svx + (size_t)i * w.st_vx;
        int32_t* it_i = io.iters_fwd ? io.iters_fwd + (size_t)i * Btot + b0 : nullptr;
        const bool bands = !dens_inline && !tr && sol_karman_fwd_bands_usable(kc);
        if (dens_fused && i >= 1) {
            const float* dprev = i == 1 ? d0 : w.d + (size_t)(i - 2) * w.st_d;
            if (int e = sol_karman_step_fwd_dens(kc, stream, vyin, vxin, re, io.active, io.inflow, bcv, bcm, io.bc_stride, vycur, vxcur, svy_i, svx_i,
                                                 feat, fscale, it_i, dprev, w.svy + (size_t)(i - 1) * w.st_vy, w.svx + (size_t)(i - 1) * w.st_vx,
                                                 w.d + (size_t)(i - 1) * w.st_d, bands ? w.xchg : nullptr)) return e;
        } else if (bands) {     // no density workgroups in this launch (first step, or the density runs as one chain behind the unroll)
            if (int e = sol_karman_step_fwd_dens(kc, stream, vyin, vxin, re, io.active, io.inflow, bcv, bcm, io.bc_stride, vycur, vxcur, svy_i, svx_i,
                                                 feat, fscale, it_i, nullptr, nullptr, nullptr, nullptr, w.xchg)) return e;
        } else if (int e = sol_karman_step_fwd(kc, stream, din, vyin, vxin, re, io.active, io.inflow, bcv, bcm, io.bc_stride,
                                               dens_inline ? dcur : nullptr, vycur, vxcur, svy_i, svx_i, feat, fscale, it_i)) return e;
        feat_order.restore();
        float* act[11];
        for (int k = 0; k < 11; ++k) act[k] = w.acts + ((size_t)i * 11 + k) * w.cells * 32;
        if (sol_conv_correct_fusable(cX, B * cY)) {            // correction + loss ride in the epilogue of the last CNN layer
            const Correct corr{vycur, vxcur, gt_vy + (size_t)i * gVy, gt_vx + (size_t)i * gVx, shared.loss_acc + (size_t)i * SOL_LOSS_ACC_WORDS};
            if (int e = net_forward(c, stream, wn, feat_cnn, act, w.O, w.amax_act + (size_t)i * 11 * SOL_AMAX_SLOTS, &corr, w.chain_flags + (size_t)(2 * i) * w.chain_words)) return e;
        } else {
            if (int e = net_forward(c, stream, wn, feat_cnn, act, w.O, w.amax_act + (size_t)i * 11 * SOL_AMAX_SLOTS, nullptr, w.chain_flags + (size_t)(2 * i) * w.chain_words)) return e;
            SOL_LAUNCH(k_correct_loss, dim3(egrid), dim3(256), 0, hs, vycur, vxcur, w.O,
                               gt_vy + (size_t)i * gVy, gt_vx + (size_t)i * gVx,
                               out_s0(c), out_s1(c), c->std_v0, c->std_v1, shared.loss_acc + (size_t)i * SOL_LOSS_ACC_WORDS, B, Y, X, tr ? 1 : 0);
            SOL_LAUNCH_CHECK();
        }
    }
    MemList fin;
    if (dens_inline) {
        if (io.d_final) fin.copy(io.d_final + (size_t)b0 * w.N, w.d + (size_t)(ms - 1) * w.st_d, w.st_d * sizeof(float));
    } else if (dens_fused) {
        // steps 0 .. ms-2 were advected inside the solver launches; the last one has no launch to ride with
        const float* dprev = ms == 1 ? d0 : w.d + (size_t)(ms - 2) * w.st_d;
        if (int e = sol_density_step(kc, stream, dprev, w.svy + (size_t)(ms - 1) * w.st_vy, w.svx + (size_t)(ms - 1) * w.st_vx, io.inflow,
                                     io.d_final + (size_t)b0 * w.N)) return e;
    } else if (io.d_final && !dens_ride_bwd) {
        // all saved velocities exist now: the whole density chain is ONE launch (one workgroup per simulation, ~0.3 ms)
        // on the main stream (as a concurrent graph branch it takes 6 CUs away from the 256-workgroup conv launches)
        if (int e = sol_density_chain(kc, stream, ms, d0, w.svy, w.svx, (long)w.st_vy, (long)w.st_vx, io.inflow, nullptr, (long)w.st_d,
                                      io.d_final + (size_t)b0 * w.N)) return e;
    }
    if (io.vy_final) fin.copy(io.vy_final + (size_t)b0 * w.nVy, w.vy + (size_t)(ms - 1) * w.st_vy, w.st_vy * sizeof(float));
    if (io.vx_final) fin.copy(io.vx_final + (size_t)b0 * w.nVx, w.vx + (size_t)(ms - 1) * w.st_vx, w.st_vx * sizeof(float));
    if (int e = fin.launch(hs)) return e;

    // ---------------- reverse sweep ----------------
    // The weight gradients of the 32 unrolled steps are NOT computed step by step: every pre-activation
    // gradient dz is kept (2.2 GB at C3, HBM is 288 GB) and each layer's dW is ONE launch over all steps
    // after the sweep (K = 32x more pixels per launch: no per-step prologue/epilogue/partial traffic).
    int cur = 0;
    const size_t cl32 = w.cells * 32;
    const long seg32 = (long)(11 * cl32);
    const int CH = pick_bww_chunk(ms);
    // weight gradients of the 32 -> 32 layers ride in the solver-adjoint launches (see k_karman_bwd_bww): 32 rows per workgroup
    const int FRB = train_fused_rb(c, w, ms);
    const bool fuse = FRB != 0;
    const int wg_per = fuse ? (B * cY) / FRB : 1;
    const bool use_side = CH < ms && pool()->ok && sol_opt().bww_side;
    hipStream_t side = pool()->s[7];
    bool side_used = false;
    for (int i = ms - 1; i >= 0; --i) {
        float* gvy = w.gvy[cur];
        float* gvx = w.gvx[cur];
        const float* vycur = w.vy + (size_t)i * w.st_vy;
        const float* vxcur = w.vx + (size_t)i * w.st_vx;
        float* dO2 = w.dO2 + (size_t)i * w.cells * 2;
        const float* act[11];
        float* D[11];
        for (int k = 0; k < 11; ++k) {
            act[k] = w.acts + ((size_t)i * 11 + k) * cl32;
            D[k] = w.dzb + ((size_t)i * 11 + k) * cl32;
        }
        uint32_t* amd = w.amax_dz + (size_t)i * 11 * SOL_AMAX_SLOTS;
        auto am = [&](int k) { return amd + (size_t)k * SOL_AMAX_SLOTS; };
        // Round 6: on 64-pixel rows the loss-gradient seed is computed by the 2 -> 32 backward-data launch itself (sol_conv5x5_seed): it reads the
        // adjoint output of step i+1 from gvy[cur] and writes G = d loss / d v_i into gvy[cur ^ 1] (whose previous content, G of step i+1,
        // the adjoint launch of step i+1 has consumed) -- one launch less per unrolled step.  Otherwise: k_seed, in place on gvy[cur].
        const bool seed_fused = !tr && cX == 64 && sol_opt().seed_fuse;
        if (seed_fused) {
            if (int e = sol_conv5x5_seed(stream, wn.wb[11], act[10], D[10], B, cY, cX, sl, am(10), vycur, vxcur, gt_vy + (size_t)i * gVy, gt_vx + (size_t)i * gVx,
                                         i == ms - 1 ? nullptr : w.gvy[cur], i == ms - 1 ? nullptr : w.gvx[cur], w.gvy[cur ^ 1], w.gvx[cur ^ 1], dO2,
                                         out_s0(c), out_s1(c), c->std_v0, c->std_v1, 1.f / (float)ms)) return e;
            gvy = w.gvy[cur ^ 1];
            gvx = w.gvx[cur ^ 1];
        } else {
            SOL_LAUNCH(k_seed, dim3(egrid), dim3(256), 0, hs, gvy, gvx, vycur, vxcur,
                               gt_vy + (size_t)i * gVy, gt_vx + (size_t)i * gVx,
                               out_s0(c), out_s1(c), c->std_v0, c->std_v1, 1.f / (float)ms, w.dO4, dO2, i == ms - 1 ? 1 : 0, B, Y, X, tr ? 1 : 0);
            SOL_LAUNCH_CHECK();
            if (int e = sol_conv5x5_scaled(stream, w.dO4, wn.wb[11], nullptr, nullptr, act[10], D[10], B, cY, cX, 4, 32, SOL_EPI_DLRELU, sl, nullptr, am(10))) return e;
        }
        if (sol_cnn_chain_usable(B, cY, cX)) {
            // the ten backward-data convolutions as ONE persistent launch
            ChainLayer L[10];
            for (int k = 4, n = 0; k >= 0; --k) {
                L[n++] = ChainLayer{sol_conv_packed_wsh(wn.wb[2 + 2 * k], 32), nullptr, nullptr, act[1 + 2 * k], D[1 + 2 * k], am(1 + 2 * k), SOL_EPI_DLRELU};
                L[n++] = ChainLayer{sol_conv_packed_wsh(wn.wb[1 + 2 * k], 32), nullptr, D[2 + 2 * k], act[2 * k], D[2 * k], am(2 * k), SOL_EPI_DLRELU};
            }
            if (int e = sol_cnn_chain_launch(hs, L, 10, D[10], w.chain_flags + (size_t)(2 * i + 1) * w.chain_words, w.chain_ctl, B, cY, cX, sl)) return e;
        } else
        for (int k = 4; k >= 0; --k) {
            const float* h = act[2 * k];
            const float* a = act[1 + 2 * k];
            if (int e = sol_conv5x5_scaled(stream, D[2 + 2 * k], wn.wb[2 + 2 * k], nullptr, nullptr, a, D[1 + 2 * k], B, cY, cX, 32, 32, SOL_EPI_DLRELU, sl,
                                           am(2 + 2 * k), am(1 + 2 * k))) return e;
            if (int e = sol_conv5x5_scaled(stream, D[1 + 2 * k], wn.wb[1 + 2 * k], nullptr, D[2 + 2 * k], h, D[2 * k], B, cY, cX, 32, 32, SOL_EPI_DLRELU, sl,
                                           am(1 + 2 * k), am(2 * k))) return e;
        }
        if (i % CH == 0) {
            // every dz of steps [i, i+CH) is final: their weight gradients go to the side stream
            const int n = (i + CH <= ms ? CH : ms - i), first = (i + CH >= ms) ? 1 : 0;
            hipStream_t bs = hs;
            if (use_side) {
                SOL_HIP_CHECK(hipEventRecord(pool()->fork, hs));
                SOL_HIP_CHECK(hipStreamWaitEvent(side, pool()->fork, 0));
                bs = side;
                side_used = true;
            }
            const float* feat_i = w.feat + (size_t)i * w.cells * 4;
            const float* acts_i = w.acts + (size_t)i * 11 * cl32;
            const float* dz_i = w.dzb + (size_t)i * 11 * cl32;
            // (cin_real = 3: the features are (v_y, v_x, Re) + one zero channel, whose gradient rows the thin kernel then skips)
            if (int e = sol_bww_batched(bs, feat_i, dz_i, w.part[0], n, CH, first, (long)(w.cells * 4), seg32, B, cY, cX, 4, 32, nullptr, nullptr, 0, 0, 3)) return e;
            const long amseg = 11 * SOL_AMAX_SLOTS;      // absmax slots: [step][11][SOL_AMAX_SLOTS]
            for (int l = 1; l <= 10 && !fuse; ++l)
                if (int e = sol_bww_batched(bs, acts_i + (size_t)(l - 1) * cl32, dz_i + (size_t)l * cl32, w.part[l], n, CH, first, seg32, seg32, B, cY, cX, 32, 32,
                                            w.amax_act + ((size_t)i * 11 + (l - 1)) * SOL_AMAX_SLOTS, w.amax_dz + ((size_t)i * 11 + l) * SOL_AMAX_SLOTS,
                                            amseg, amseg)) return e;
            if (int e = sol_bww_batched(bs, acts_i + (size_t)10 * cl32, w.dO2 + (size_t)i * w.cells * 2, w.part[11], n, CH, first, seg32, (long)(w.cells * 2), B, cY, cX, 32, 2)) return e;
        }
        BwArgs jobs[10];
        if (fuse) {       // this step's dz tensors are complete: one job per 32 -> 32 layer
            for (int l = 1; l <= 10; ++l)
                // (dbg_skip & 1024, timing experiment, results invalid: every step OVERWRITES its partial slice -- the launch without the
                //  read of the old partials = the upper bound of what accumulators kept resident across the steps could save)
                if (int e = sol_bww_step_job(&jobs[l - 1], act[l - 1], D[l], w.part[l], (i == ms - 1 || (sol_opt().dbg_skip & 1024)) ? 1 : 0, B, cY, cX, FRB,
                                             w.amax_act + ((size_t)i * 11 + (l - 1)) * SOL_AMAX_SLOTS, am(l))) return e;
        }
        SolDensRide ride{};
        if (dens_ride_bwd) {          // density advection s = ms - 1 - i (forward order) rides in this step's launch
            const int sd = ms - 1 - i;
            ride = SolDensRide{sd == 0 ? d0 : w.d + (size_t)(sd - 1) * w.st_d, w.svy + (size_t)sd * w.st_vy, w.svx + (size_t)sd * w.st_vx, io.inflow,
                               sd == ms - 1 ? io.d_final + (size_t)b0 * w.N : w.d + (size_t)sd * w.st_d};
        }
        if (i > 0) {
            if (int e = sol_conv5x5_scaled(stream, D[0], wn.wb[0], nullptr, nullptr, nullptr, w.dF, B, cY, cX, 32, 2, SOL_EPI_NONE, sl, am(0), nullptr)) return e;
            const float* dF = w.dF;
            FeatOrder feat_order(tr);       // the adjoint reads the feature gradient in the CNN's (transposed) cell order
            if (int e = sol_karman_step_bwd_fused(kc, stream, w.svy + (size_t)i * w.st_vy, w.svx + (size_t)i * w.st_vx, re, io.active,
                                                  bcm, io.bc_stride, gvy, gvx, dF, fscale, seed_fused ? w.gvy[cur] : w.gvy[cur ^ 1], seed_fused ? w.gvx[cur] : w.gvx[cur ^ 1],
                                                  io.iters_bwd ? io.iters_bwd + (size_t)i * Btot + b0 : nullptr,
                                                  jobs, fuse ? 10 : 0, wg_per, dens_ride_bwd ? &ride : nullptr)) return e;
            if (!seed_fused) cur ^= 1;        // (seed fused: G lives in gvy[cur ^ 1], the adjoint's output goes back to gvy[cur] -- no toggle)
        } else if (fuse) {
            if (int e = sol_bww_jobs_launch(stream, jobs, 10, wg_per, kc, dens_ride_bwd ? &ride : nullptr)) return e;     // step 0 has no adjoint to ride with
        }
    }
    if (side_used) {      // join the side stream
        SOL_HIP_CHECK(hipEventRecord(pool()->join[7], side));
        SOL_HIP_CHECK(hipStreamWaitEvent(hs, pool()->join[7], 0));
    }
    return SOL_OK;
}

int train_fwd_bwd_impl(const sol_train_cfg* cfg, hipStream_t hs, const TrainIO& io, void* workspace, size_t workspace_bytes, float* grads) {
    const int B = cfg->karman.B, Y = cfg->karman.Y, X = cfg->karman.X, ms = cfg->msteps;
    const int S = pick_chains(B);
    sol_train_cfg sub = *cfg;
    sub.karman.B = B / S;
    Ws w[8];
    const size_t sub_bytes = carve_ws(&sub, nullptr, w[0], true);
    if (workspace_bytes < sub_bytes * S)
        return sol_set_error(SOL_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", workspace_bytes, sub_bytes * S);
    for (int k = 0; k < S; ++k) carve_ws(&sub, reinterpret_cast<float*>(static_cast<char*>(workspace) + k * sub_bytes), w[k], true);
    if (int e = sol_init_karman_kernels()) return e;
    if (int e = sol_init_conv_kernels()) return e;
    StreamPool* sp = pool();
    if (S > 1 && !sp->ok) return sol_set_error(SOL_ERR_HIP, "could not create the internal HIP streams/events");

    if (int e = pack_all(cfg, hs, io.params, w[0], true)) return e;
    for (int k = 0; k < S; ++k) {
        MemList z;
        if (k == 0) z.zero(io.iters_bwd, B * sizeof(int32_t));                       // step 0 needs no adjoint  (io.loss_steps: every entry is written by k_loss_finish)
        z.zero(w[k].dO4, w[k].cells * 4 * sizeof(float));
        if (k == 0) z.zero(w[0].loss_acc, (size_t)ms * SOL_LOSS_ACC_WORDS * sizeof(unsigned long long));   // every chain adds into chain 0's accumulators
        z.zero(w[k].amax_act, 2 * w[k].amax_words * sizeof(uint32_t));                 // activation + gradient absmax slots
        if (sol_karman_fwd_bands_usable(&sub.karman)) z.zero(w[k].xchg, w[k].xchg_words * sizeof(uint32_t));   // (the launches restore the zeros themselves; this covers an aborted step)
        const bool chain = sol_cnn_chain_usable(sub.karman.B, cnn_transposed(Y, X) ? X : Y, cnn_transposed(Y, X) ? Y : X);
        // hand-off regions of the persistent CNN launches: zero ONCE (tag 0 = "never written"); afterwards the tags do the work
        if (chain) z.zero_once(w[k].chain_flags, (size_t)ms * 2 * w[k].chain_words * sizeof(uint32_t), w[k].chain_ctl + 1, w[k].chain_magic);
        if (int e = z.launch(hs)) return e;
        if (chain) {
            SOL_LAUNCH(k_chain_tick, dim3(1), dim3(64), 0, hs, w[k].chain_ctl, w[k].chain_magic);
            SOL_LAUNCH_CHECK();
        }
    }
    if (S == 1) {
        if (int e = run_chain(&sub, w[0], w[0], B, 0, hs, io)) return e;
    } else {
        SOL_HIP_CHECK(hipEventRecord(sp->fork, hs));
        for (int k = 0; k < S; ++k) {
            SOL_HIP_CHECK(hipStreamWaitEvent(sp->s[k], sp->fork, 0));
            if (int e = run_chain(&sub, w[k], w[0], B, k * (B / S), sp->s[k], io)) return e;
            SOL_HIP_CHECK(hipEventRecord(sp->join[k], sp->s[k]));
            SOL_HIP_CHECK(hipStreamWaitEvent(hs, sp->join[k], 0));
        }
    }
    SOL_LAUNCH(k_loss_finish, dim3((ms + 63) / 64), dim3(64), 0, hs, (const unsigned long long*)w[0].loss_acc, io.loss_steps, ms);
    SOL_LAUNCH_CHECK();
    // all twelve layers of a chain in two launches (chain k > 0 accumulates onto chain k-1: one pair of launches per chain)
    const bool trn = cnn_transposed(Y, X);
    const int fused_rb = train_fused_rb(&sub, w[0], ms);
    for (int k = 0; k < S; ++k) {
        float *part[NL], *dw[NL], *db[NL];
        int rows[NL], rbs[NL], cins[NL], couts[NL];
        for (int l = 0; l < NL; ++l) {
            const int cin = layer_cin(l), cout = layer_cout(l);
            const int64_t koff = layer_koff(l), boff = koff + 25 * cin * cout;
            const bool fused = l >= 1 && l <= 10 && fused_rb != 0;
            part[l] = w[k].part[l]; dw[l] = grads + koff; db[l] = grads + boff; cins[l] = cin; couts[l] = cout;
            rows[l] = (fused ? 1 : pick_bww_chunk(ms)) * (B / S) * (trn ? X : Y);
            rbs[l] = fused ? fused_rb : 0;
        }
        if (int e = sol_bww_reduce_layers(hs, NL, part, dw, db, rows, rbs, cins, couts, k > 0 ? 1 : 0, trn ? 1 : 0)) return e;
    }
    return SOL_OK;
}

}  // namespace

extern "C" size_t sol_train_workspace_bytes(const sol_train_cfg* cfg) {
    if (!cfg || cfg->karman.B < 1) return 0;
    const int S = pick_chains(cfg->karman.B);
    sol_train_cfg sub = *cfg;
    sub.karman.B = cfg->karman.B / S;
    Ws w;
    return carve_ws(&sub, nullptr, w, true) * S;
}

extern "C" int sol_train_fwd_bwd(const sol_train_cfg* cfg, void* stream, const float* params,
                                 const float* d0, const float* vy0, const float* vx0, const float* re,
                                 const float* active, const float* inflow,
                                 const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                                 const float* gt_vy, const float* gt_vx,
                                 void* workspace, size_t workspace_bytes,
                                 float* grads, float* loss_steps,
                                 float* d_final, float* vy_final, float* vx_final,
                                 int32_t* iters_fwd, int32_t* iters_bwd) {
    if (int e = check_train_cfg(cfg)) return e;
    SOL_REQUIRE(params && d0 && vy0 && vx0 && re && active && inflow && velBCy && velBCyMask && gt_vy && gt_vx &&
                workspace && grads && loss_steps, "sol_train_fwd_bwd: NULL pointer argument");
    TrainIO io{params, d0, vy0, vx0, re, active, inflow, velBCy, velBCyMask, gt_vy, gt_vx, bc_batch_stride,
               loss_steps, d_final, vy_final, vx_final, iters_fwd, iters_bwd};
    return train_fwd_bwd_impl(cfg, (hipStream_t)stream, io, workspace, workspace_bytes, grads);
}

// ---- graph-capture guard ---------------------------------------------------------------------------
// A replayed hipGraph on this path may contain KERNEL nodes (and empty / event / child-graph nodes) only.  Memset and memcpy nodes
// are refused: on ROCm 7.2 a captured memset node is unreliable under replay -- round 4: a torch reduction's semaphore memset inside a
// captured trainer made the reported per-step losses 0.5x / 2x the true values after a few replays while state and gradient stayed
// right -- and a memcpy node means somebody staged data inside the timed region.  Every capture site of the
// package (sol_train_graph_create here; the torch captures of GraphTrainer, BurgersTrainer, BurgersRollout, Karman3DTrainer through
// _lib.capture_graph) passes its graph through this check BEFORE instantiating it, so the defect class is refused, not tested for.
// Reference shape it protects: "build the graph once, sess.run many" (karman_train.py:385-391, 502).
static const char* graph_node_type_name(hipGraphNodeType t) {
    switch (t) {
        case hipGraphNodeTypeKernel: return "kernel";
        case hipGraphNodeTypeMemcpy: return "memcpy";
        case hipGraphNodeTypeMemset: return "memset";
        case hipGraphNodeTypeHost: return "host";
        case hipGraphNodeTypeGraph: return "child-graph";
        case hipGraphNodeTypeEmpty: return "empty";
        case hipGraphNodeTypeWaitEvent: return "wait-event";
        case hipGraphNodeTypeEventRecord: return "event-record";
        case hipGraphNodeTypeExtSemaphoreSignal: return "ext-semaphore-signal";
        case hipGraphNodeTypeExtSemaphoreWait: return "ext-semaphore-wait";
        case hipGraphNodeTypeMemAlloc: return "mem-alloc";
        case hipGraphNodeTypeMemFree: return "mem-free";
        case hipGraphNodeTypeMemcpyFromSymbol: return "memcpy-from-symbol";
        case hipGraphNodeTypeMemcpyToSymbol: return "memcpy-to-symbol";
        default: return "other";
    }
}

// counts[hipGraphNodeType] += nodes of `graph` (child graphs are entered); returns SOL_OK or a HIP error
static int graph_census(hipGraph_t graph, int32_t* counts, int ncounts, int depth) {
    size_t n = 0;
    SOL_HIP_CHECK(hipGraphGetNodes(graph, nullptr, &n));
    if (n == 0) return SOL_OK;
    hipGraphNode_t* nodes = new hipGraphNode_t[n];
    hipError_t e = hipGraphGetNodes(graph, nodes, &n);
    int rc = SOL_OK;
    for (size_t i = 0; i < n && e == hipSuccess && rc == SOL_OK; ++i) {
        hipGraphNodeType t;
        e = hipGraphNodeGetType(nodes[i], &t);
        if (e != hipSuccess) break;
        if ((int)t >= 0 && (int)t < ncounts) counts[(int)t]++;
        if (t == hipGraphNodeTypeGraph && depth < 8) {
            hipGraph_t child = nullptr;
            e = hipGraphChildGraphNodeGetGraph(nodes[i], &child);
            if (e == hipSuccess && child) rc = graph_census(child, counts, ncounts, depth + 1);
        }
    }
    delete[] nodes;
    if (e != hipSuccess) return sol_set_error(SOL_ERR_HIP, "graph node enumeration failed: %s", hipGetErrorString(e));
    return rc;
}

extern "C" int sol_graph_census(void* graph, int32_t* counts, int32_t ncounts) {
    SOL_REQUIRE(graph && counts && ncounts >= 1 && ncounts <= 64, "sol_graph_census: graph, counts[1..64]");
    for (int i = 0; i < ncounts; ++i) counts[i] = 0;
    return graph_census((hipGraph_t)graph, counts, ncounts, 0);
}

extern "C" const char* sol_graph_node_type_name(int32_t type) { return graph_node_type_name((hipGraphNodeType)type); }

// Text for sol_graph_check: the first few refused nodes of `g` -- child graphs entered, like the census -- each with the node before and
// after it ("memcpy after <kernel> before <kernel>").  Edge lists are sized by a count query (NULL array) first.
static void graph_node_label(hipGraphNode_t nd, char* out, size_t cap) {
    hipGraphNodeType t;
    out[0] = 0;
    if (hipGraphNodeGetType(nd, &t) != hipSuccess) return;
    if (t == hipGraphNodeTypeKernel) {
        hipKernelNodeParams kp;
        const char* nm = nullptr;
        if (hipGraphKernelNodeGetParams(nd, &kp) == hipSuccess && kp.func) nm = hipKernelNameRefByPtr(kp.func, nullptr);
        snprintf(out, cap, "%.60s", nm ? nm : "kernel");
    } else snprintf(out, cap, "%s", graph_node_type_name(t));
}
static bool graph_node_refused(hipGraphNodeType t) {
    return t == hipGraphNodeTypeMemset || t == hipGraphNodeTypeMemcpy || t == hipGraphNodeTypeMemcpyFromSymbol || t == hipGraphNodeTypeMemcpyToSymbol ||
           t == hipGraphNodeTypeHost || t == hipGraphNodeTypeMemAlloc || t == hipGraphNodeTypeMemFree;
}
static void graph_offenders(hipGraph_t g, char* det, size_t cap, size_t& dl, int& shown, int depth) {
    size_t n = 0;
    if (depth > 8 || hipGraphGetNodes(g, nullptr, &n) != hipSuccess || !n) return;
    std::vector<hipGraphNode_t> nodes(n);
    if (hipGraphGetNodes(g, nodes.data(), &n) != hipSuccess) return;
    for (size_t i = 0; i < n && shown < 4 && dl + 160 < cap; ++i) {
        hipGraphNodeType t;
        if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess) return;
        if (t == hipGraphNodeTypeGraph) {
            hipGraph_t child = nullptr;
            if (hipGraphChildGraphNodeGetGraph(nodes[i], &child) == hipSuccess && child) graph_offenders(child, det, cap, dl, shown, depth + 1);
            continue;
        }
        if (!graph_node_refused(t)) continue;
        char before[64] = "", after[64] = "";
        size_t k = 0;
        if (hipGraphNodeGetDependencies(nodes[i], nullptr, &k) == hipSuccess && k) {
            std::vector<hipGraphNode_t> nb(k);
            if (hipGraphNodeGetDependencies(nodes[i], nb.data(), &k) == hipSuccess && k) graph_node_label(nb[0], before, sizeof(before));
        }
        k = 0;
        if (hipGraphNodeGetDependentNodes(nodes[i], nullptr, &k) == hipSuccess && k) {
            std::vector<hipGraphNode_t> nb(k);
            if (hipGraphNodeGetDependentNodes(nodes[i], nb.data(), &k) == hipSuccess && k) graph_node_label(nb[0], after, sizeof(after));
        }
        size_t bytes = 0;
        if (t == hipGraphNodeTypeMemset) {
            hipMemsetParams mp;
            if (hipGraphMemsetNodeGetParams(nodes[i], &mp) == hipSuccess) bytes = mp.width * (mp.height ? mp.height : 1) * mp.elementSize;
        }
        dl += (size_t)snprintf(det + dl, cap - dl, "%s%s%s", shown++ ? "; " : "", graph_node_type_name(t), depth ? " (in a child graph)" : "");
        if (bytes && dl < cap) dl += (size_t)snprintf(det + dl, cap - dl, " of %zu B", bytes);
        if (dl < cap) dl += (size_t)snprintf(det + dl, cap - dl, " after '%s' before '%s'", before[0] ? before : "-", after[0] ? after : "-");
        if (dl >= cap) dl = cap - 1;
    }
}

extern "C" int sol_graph_check(void* graph, const char* what) {
    SOL_REQUIRE(graph != nullptr, "sol_graph_check: NULL graph");
    int32_t counts[32];
    if (int e = sol_graph_census(graph, counts, 32)) return e;
    // refused: everything that moves or clears memory without being one of this library's (or the caller's) kernels
    const hipGraphNodeType refused[] = {hipGraphNodeTypeMemset, hipGraphNodeTypeMemcpy, hipGraphNodeTypeMemcpyFromSymbol, hipGraphNodeTypeMemcpyToSymbol,
                                        hipGraphNodeTypeHost, hipGraphNodeTypeMemAlloc, hipGraphNodeTypeMemFree};
    char found[256];
    size_t len = 0;
    int bad = 0;
    found[0] = 0;
    for (hipGraphNodeType t : refused)
        if ((int)t < 32 && counts[(int)t] > 0) {
            bad += counts[(int)t];
            len += (size_t)snprintf(found + len, len < sizeof(found) ? sizeof(found) - len : 0, "%s%d %s", len ? ", " : "", counts[(int)t], graph_node_type_name(t));
            if (len >= sizeof(found)) len = sizeof(found) - 1;
        }
    if (bad) {
        // the first few offenders with their neighbours in the graph ("memcpy after <kernel> before <kernel>"): that is what locates
        // the host line -- the copy follows the kernel that produced its source and precedes the first consumer of its destination
        char det[600];
        size_t dl = 0;
        int shown = 0;
        det[0] = 0;
        graph_offenders((hipGraph_t)graph, det, sizeof(det), dl, shown, 0);
        return sol_set_error(SOL_ERR_GRAPH, "%s: the captured graph holds %s node(s) among %d kernel nodes [first: %s] -- only kernel nodes replay reliably on this "
                             "path (a memset node is what a multi-workgroup torch reduction or torch.zeros() inside the capture leaves behind: use the "
                             "library's kernels -- ops.L2LossFn for the loss, _lib.dcopy_ / _lib.dclone for copies -- or allocate and clear before the capture)",
                             what ? what : "graph", found, counts[(int)hipGraphNodeTypeKernel], det);
    }
    return SOL_OK;
}

// ---- the same step as a replayable hipGraph (removes ~1000 host launches per step) --------------
struct sol_train_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

extern "C" int sol_train_graph_create(const sol_train_cfg* cfg, const float* params,
                                      const float* d0, const float* vy0, const float* vx0, const float* re,
                                      const float* active, const float* inflow,
                                      const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                                      const float* gt_vy, const float* gt_vx,
                                      void* workspace, size_t workspace_bytes,
                                      float* grads, float* loss_steps,
                                      float* d_final, float* vy_final, float* vx_final,
                                      int32_t* iters_fwd, int32_t* iters_bwd, sol_train_graph** out) {
    if (int e = check_train_cfg(cfg)) return e;
    SOL_REQUIRE(params && d0 && vy0 && vx0 && re && active && inflow && velBCy && velBCyMask && gt_vy && gt_vx &&
                workspace && grads && loss_steps && out, "sol_train_graph_create: NULL pointer argument");
    if (int e = sol_init_karman_kernels()) return e;
    if (int e = sol_init_conv_kernels()) return e;
    if (!pool()->ok) return sol_set_error(SOL_ERR_HIP, "could not create the internal HIP streams/events");
    TrainIO io{params, d0, vy0, vx0, re, active, inflow, velBCy, velBCyMask, gt_vy, gt_vx, bc_batch_stride,
               loss_steps, d_final, vy_final, vx_final, iters_fwd, iters_bwd};
    hipStream_t cs;
    SOL_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipStreamDestroy(cs);
        return sol_set_error(SOL_ERR_HIP, "hipStreamBeginCapture failed");
    }
    const int rc = train_fwd_bwd_impl(cfg, cs, io, workspace, workspace_bytes, grads);
    const hipError_t ee = hipStreamEndCapture(cs, &graph);
    (void)hipStreamDestroy(cs);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (ee != hipSuccess || !graph) return sol_set_error(SOL_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ee));
    if (int e = sol_graph_check(graph, "sol_train_graph_create")) { (void)hipGraphDestroy(graph); return e; }      // kernel nodes only
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (ei != hipSuccess) { (void)hipGraphDestroy(graph); return sol_set_error(SOL_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ei)); }
    *out = new sol_train_graph{graph, exec};
    return SOL_OK;
}

extern "C" int sol_train_graph_launch(sol_train_graph* g, void* stream) {
    SOL_REQUIRE(g && g->exec, "sol_train_graph_launch: NULL graph");
    hipStream_t s = (hipStream_t)stream;
    if (sol_opt().graph_stream) {
        // replay on an internal non-blocking stream, fenced against the caller's stream with two events
        StreamPool* sp = pool();
        SOL_REQUIRE(sp->ok, "internal streams unavailable");
        SOL_HIP_CHECK(hipEventRecord(sp->fork, s));
        SOL_HIP_CHECK(hipStreamWaitEvent(sp->s[6], sp->fork, 0));
        SOL_HIP_CHECK(hipGraphLaunch(g->exec, sp->s[6]));
        SOL_HIP_CHECK(hipEventRecord(sp->join[6], sp->s[6]));
        SOL_HIP_CHECK(hipStreamWaitEvent(s, sp->join[6], 0));
        return SOL_OK;
    }
    SOL_HIP_CHECK(hipGraphLaunch(g->exec, s));
    return SOL_OK;
}

extern "C" int sol_train_graph_destroy(sol_train_graph* g) {
    if (!g) return SOL_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return SOL_OK;
}

extern "C" int sol_rollout(const sol_train_cfg* cfg, void* stream, const float* params,
                           float* d, float* vy, float* vx, const float* re,
                           const float* active, const float* inflow,
                           const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                           int32_t nsteps, void* workspace, size_t workspace_bytes, int32_t* iters) {
    if (int e = check_train_cfg(cfg)) return e;
    SOL_REQUIRE(params && d && vy && vx && re && active && inflow && velBCy && velBCyMask && workspace,
                "sol_rollout: NULL pointer argument");
    SOL_REQUIRE(nsteps >= 0, "nsteps must be >= 0");
    Ws w;
    const size_t need = carve_ws(cfg, static_cast<float*>(workspace), w, false);
    if (workspace_bytes < need)
        return sol_set_error(SOL_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    hipStream_t hs = (hipStream_t)stream;
    const sol_karman_cfg* kc = &cfg->karman;
    const int B = kc->B, Y = kc->Y, X = kc->X;
    const float fscale[3] = {1.f / in_s0(cfg), 1.f / in_s1(cfg), 1.f / cfg->std_re};
    const int egrid = (int)((w.st_vy + w.st_vx + 255) / 256);
    if (int e = pack_all(cfg, stream, params, w, false)) return e;
    float* act[11];
    // two ping-pong activation buffers suffice without the backward pass (+1 for the skip input)
    for (int k = 0; k < 11; ++k) act[k] = w.acts + (size_t)(k % 3) * w.cells * 32;
    // the state ping-pongs between the caller's buffers and the workspace (no copy-back per step); an odd step count ends
    // with one copy into the caller's buffers
    // The passive density leaves the solver kernel's critical path as in the training unroll (12 us of a 48 us launch at B = 1): the
    // advection of step i-1 -- it only needs that step's post-diffusion velocity, kept in two slots (the gradient buffers of the
    // training workspace are free here) -- rides in the solver launch of step i as B extra workgroups (k_karman_fwd_dens), the last
    // one is its own small launch behind the loop.  Same buffers, same values.
    const bool dens_ride = sol_karman_bwd_fusable(kc) && sol_opt().density_mode == 0;
    const bool bands = dens_ride && !cnn_transposed(Y, X) && sol_karman_fwd_bands_usable(kc);      // four workgroups per simulation (k_karman_fwd_bands)
    if (bands && nsteps > 0) {
        MemList z;
        z.zero(w.xchg, w.xchg_words * sizeof(uint32_t));
        if (int e = z.launch(hs)) return e;
    }
    for (int i = 0; i < nsteps; ++i) {
        float* sd = (i & 1) ? w.d : d;   float* svy = (i & 1) ? w.vy : vy;   float* svx = (i & 1) ? w.vx : vx;
        float* td = (i & 1) ? d : w.d;   float* tvy = (i & 1) ? vy : w.vy;   float* tvx = (i & 1) ? vx : w.vx;
        const bool tr = cnn_transposed(Y, X);
        {
            FeatOrder feat_order(tr);       // transposed CNN mode: features straight in the CNN's cell order
            int32_t* it_i = iters ? iters + (size_t)i * B : nullptr;
            if (!dens_ride) {
                if (int e = sol_karman_step_fwd(kc, stream, sd, svy, svx, re, active, inflow, velBCy, velBCyMask, bc_batch_stride,
                                                td, tvy, tvx, nullptr, nullptr, w.feat, fscale, it_i)) return e;
            } else if (i == 0) {
                if (bands) {
                    if (int e = sol_karman_step_fwd_dens(kc, stream, svy, svx, re, active, inflow, velBCy, velBCyMask, bc_batch_stride, tvy, tvx,
                                                         w.gvy[0], w.gvx[0], w.feat, fscale, it_i, nullptr, nullptr, nullptr, nullptr, w.xchg)) return e;
                } else if (int e = sol_karman_step_fwd(kc, stream, sd, svy, svx, re, active, inflow, velBCy, velBCyMask, bc_batch_stride,
                                                       nullptr, tvy, tvx, w.gvy[0], w.gvx[0], w.feat, fscale, it_i)) return e;
            } else {
                float* psd = ((i - 1) & 1) ? w.d : d;           // density in / out of step i-1 (the buffers it would have used itself)
                float* ptd = ((i - 1) & 1) ? d : w.d;
                if (int e = sol_karman_step_fwd_dens(kc, stream, svy, svx, re, active, inflow, velBCy, velBCyMask, bc_batch_stride, tvy, tvx,
                                                     w.gvy[i & 1], w.gvx[i & 1], w.feat, fscale, it_i,
                                                     psd, w.gvy[(i - 1) & 1], w.gvx[(i - 1) & 1], ptd, bands ? w.xchg : nullptr)) return e;
            }
        }
        if (i % ROLLOUT_AMAX_SETS == 0) {
            MemList z;
            const bool chain = sol_cnn_chain_usable(B, tr ? X : Y, tr ? Y : X);
            if (chain) z.zero_once(w.chain_flags, (size_t)ROLLOUT_AMAX_SETS * w.chain_words * sizeof(uint32_t), w.chain_ctl + 1, w.chain_magic);
            z.zero(w.amax_act, w.amax_words * sizeof(uint32_t));
            if (int e = z.launch(hs)) return e;
            if (chain) {
                SOL_LAUNCH(k_chain_tick, dim3(1), dim3(64), 0, hs, w.chain_ctl, w.chain_magic);
                SOL_LAUNCH_CHECK();
            }
        }
        uint32_t* amax = w.amax_act + (size_t)(i % ROLLOUT_AMAX_SETS) * 11 * SOL_AMAX_SLOTS;
        if (sol_conv_correct_fusable(tr ? Y : X, tr ? B * X : B * Y)) {
            const Correct corr{tvy, tvx, nullptr, nullptr, nullptr};
            if (int e = net_forward(cfg, stream, w, w.feat, act, w.O, amax, &corr, w.chain_flags + (size_t)(i % ROLLOUT_AMAX_SETS) * w.chain_words)) return e;
        } else {
            if (int e = net_forward(cfg, stream, w, w.feat, act, w.O, amax, nullptr, w.chain_flags + (size_t)(i % ROLLOUT_AMAX_SETS) * w.chain_words)) return e;
            SOL_LAUNCH(k_correct_loss, dim3(egrid), dim3(256), 0, hs, tvy, tvx, w.O,
                               (const float*)nullptr, (const float*)nullptr, out_s0(cfg), out_s1(cfg), cfg->std_v0, cfg->std_v1, (unsigned long long*)nullptr, B, Y, X, tr ? 1 : 0);
            SOL_LAUNCH_CHECK();
        }
    }
    if (dens_ride && nsteps > 0) {      // the density of the last step
        const int k = nsteps - 1;
        if (int e = sol_density_step(kc, stream, (k & 1) ? w.d : d, w.gvy[k & 1], w.gvx[k & 1], inflow, (k & 1) ? d : w.d)) return e;
    }
    if (nsteps & 1) {
        MemList c;
        c.copy(d, w.d, w.st_d * sizeof(float));
        c.copy(vy, w.vy, w.st_vy * sizeof(float));
        c.copy(vx, w.vx, w.st_vx * sizeof(float));
        if (int e = c.launch(hs)) return e;
    }
    return SOL_OK;
}

// ---- stand-alone l2 loss of one unrolled step (karman_train.py:428-436), SURVEY 8b2 -------------------------------------
// loss (+)= 0.5 * sum_c sum_e ((gt_c[e] - v_c[e]) / std_c)^2;  g_c[e] (+)= gscale * (v_c[e] - gt_c[e]) / std_c^2.
// The trainers never call it (their loss is fused into the last CNN layer / k_seed); it exists so that a host binding the
// per-op ABI can compose the reference's loss.  Deterministic: per-workgroup partial sums in a fixed layout, folded in a fixed
// order by one wave (no floating-point atomics).
namespace {
constexpr int L2_BLOCKS = 256;
struct L2Args { const float* v[3]; const float* gt[3]; float* g[3]; long n[3]; float il[3]; int nc; };
__global__ void __launch_bounds__(256) k_l2_loss(L2Args a, float gscale, int acc_g, float* __restrict__ part) {
    __shared__ float red[64];
    float s = 0.f;
    for (int c = 0; c < a.nc; ++c) {
        const float il = a.il[c], gs = gscale * il * il;
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < a.n[c]; e += (long)gridDim.x * blockDim.x) {
            const float d = a.v[c][e] - a.gt[c][e], q = d * il;
            s += 0.5f * q * q;
            if (a.g[c]) a.g[c][e] = (acc_g ? a.g[c][e] : 0.f) + gs * d;
        }
    }
    s = block_sum(s, red, 0);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void __launch_bounds__(64) k_l2_fold(const float* __restrict__ part, int n, float* __restrict__ loss, int acc) {
    float s = 0.f;
    for (int k = threadIdx.x; k < n; k += 64) s += part[k];
    s = wave_sum(s);
    if (threadIdx.x == 0) loss[0] = (acc ? loss[0] : 0.f) + s;
}
}  // namespace
extern "C" int32_t sol_l2_loss_scratch_floats(void) { return L2_BLOCKS; }
extern "C" int sol_l2_loss_fwd_bwd(void* stream, int32_t ncomp, const float* const* v, const float* const* gt, float* const* g,
                                   const int64_t* n, const float* std, float gscale, int32_t accumulate_g,
                                   float* loss, int32_t accumulate_loss, float* scratch) {
    SOL_REQUIRE(ncomp >= 1 && ncomp <= 3 && v && gt && n && std && loss && scratch, "sol_l2_loss_fwd_bwd: bad arguments (1..3 components)");
    L2Args a{};
    a.nc = ncomp;
    long total = 0;
    for (int c = 0; c < ncomp; ++c) {
        SOL_REQUIRE(v[c] && gt[c] && n[c] > 0 && std[c] > 0.f, "sol_l2_loss_fwd_bwd: component %d: NULL pointer, empty or std <= 0", c);
        a.v[c] = v[c]; a.gt[c] = gt[c]; a.g[c] = g ? g[c] : nullptr; a.n[c] = (long)n[c]; a.il[c] = 1.f / std[c];
        total += (long)n[c];
    }
    hipStream_t hs = (hipStream_t)stream;
    const int grid = (int)std::min<long>(L2_BLOCKS, (total + 255) / 256);
    SOL_LAUNCH(k_l2_loss, dim3(grid), dim3(256), 0, hs, a, gscale, (int)accumulate_g, scratch);
    SOL_LAUNCH(k_l2_fold, dim3(1), dim3(64), 0, hs, (const float*)scratch, grid, loss, (int)accumulate_loss);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

extern "C" int sol_adam_tf_step(void* stream, float* params, const float* grads, float* m, float* v,
                                int64_t n, int32_t t, float lr, float beta1, float beta2, float eps,
                                float clip_norm, const int64_t* tensor_offsets, int32_t n_tensors, float* scratch) {
    SOL_REQUIRE(params && grads && m && v && n > 0 && t >= 1, "sol_adam_tf_step: bad arguments");
    hipStream_t hs = (hipStream_t)stream;
    AdamTensors T{};
    const float* scale = nullptr;
    if (clip_norm > 0.f) {
        SOL_REQUIRE(tensor_offsets && scratch && n_tensors >= 1 && n_tensors < 40, "clip_norm needs tensor_offsets/scratch (n_tensors < 40)");
        T.n = n_tensors;
        for (int k = 0; k <= n_tensors; ++k) T.off[k] = tensor_offsets[k];
        for (int k = 0; k < n_tensors; ++k) {
            SOL_LAUNCH(k_tensor_scale, dim3(1), dim3(256), 0, hs, grads, scratch + k, T.off[k], T.off[k + 1] - T.off[k], clip_norm);
            SOL_LAUNCH_CHECK();
        }
        scale = scratch;
    }
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t));
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    SOL_LAUNCH(k_adam, dim3(grid), dim3(256), 0, hs, params, grads, m, v, n, (float)lr_t, beta1, beta2, eps, scale, T);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}
