// Shared helpers for the gfx950 kernels of libsol_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <initializer_list>

#include "../../include/sol_hip.h"

int sol_set_error(int code, const char* fmt, ...);
int sol_init_karman_kernels();
int sol_karman_feat_transposed(int on);      // internal: feature I/O of the step kernels in the CNN's transposed cell order (karman_step.hip)
int sol_init_conv_kernels();
int sol_density_chain(const sol_karman_cfg* c, void* stream, int ms, const float* d0, const float* svy, const float* svx,
                      long st_vy, long st_vx, const float* inflow, float* d_steps, long st_d, float* d_final);
// internal (train.hip): last CNN layer (32 -> 2, no activation) fused with `velocity += std * to_staggered(output)` and the
// l2 loss of the step (karman_train.py:413-447); needs the split-precision kernels (sol_conv_correct_fusable)
bool sol_conv_correct_fusable(int W, int rows);      // W, rows = B * H of the CNN's images (the transposed ones in transposed CNN mode)
int sol_conv5x5_seed(void* stream, const float* packed_bwd, const float* act_ref, float* y, int B, int H, int W, float slope, unsigned* y_absmax,
                     const float* vy, const float* vx, const float* gt_vy, const float* gt_vx, const float* gin_vy, const float* gin_vx,
                     float* g_vy, float* g_vx, float* dO2, float s0, float s1, float l0, float l1, float inv_m);
int sol_conv5x5_correct(void* stream, const float* x, const float* packed, const float* bias, int B, int H, int W,
                        const unsigned* x_absmax, float* vy, float* vx, const float* gt_vy, const float* gt_vx,
                        float s0, float s1, float l0, float l1, unsigned long long* loss_acc, int transposed);
size_t sol_bww_batched_ws_floats(int nseg, int B, int H, int cin, int cout);
int sol_bww_batched(void* stream, const float* x, const float* dz, float* partial, int nseg, int nseg_layout, int overwrite,
                    long x_seg, long dz_seg, int B, int H, int W, int cin, int cout,
                    const unsigned* xmax = nullptr, const unsigned* zmax = nullptr, long xmax_seg = 0, long zmax_seg = 0, int cin_real = 0);
int sol_bww_batched_reduce(void* stream, const float* partial, float* dw_hwio, float* db, int nseg, int B, int H,
                           int cin, int cout, int accumulate, int taps_transposed = 0);

// C[b] (M x N) = (accumulate ? C[b] : 0) + A[b] (M x K) * B[b] (K x N), row major, fp32, batch strides may be 0 (karman_large.hip)
int sol_gemm_f32(hipStream_t s, int batch, const float* A, int lda, long sA, const float* Bm, int ldb, long sB, float* C, int ldc, long sC,
                 int M, int N, int K, int accumulate);

// forward / backward-data convolution arguments (conv5x5.hip, conv5x5_sb.hip)
struct ConvArgs {
    const float *x, *wp, *bias, *res, *act;
    float* y;
    int B, H, W, CO;   // CO = real number of stored output channels
    int epi;
    float slope;
    int TW, RPW, tiles_x;
    const void* wsb;   // split-bf16 weight planes (second section of the packed buffer), cin == 32 only
    const void* wsh;   // split-fp16 weight planes + header (third section), cin == 32 only
    const unsigned* xmax;   // [SOL_AMAX_SLOTS] slots, max over them = bits of max|x| (non-negative float) -> fp16 path; NULL: bf16 path
    unsigned* ymax;         // [SOL_AMAX_SLOTS] slots updated with max|y| (atomic max on the float bits), or NULL
    // trainer only (sol_conv5x5_correct): the 32 -> 2 layer applies its output as the velocity correction instead of storing it
    float *cvy, *cvx;               // staggered velocity [B,H+1,W] / [B,H,W+1], updated in place (channel 0 -> v_y rows < H, 1 -> v_x columns < W)
    const float *gty, *gtx;         // ground-truth frames for the l2 loss, or NULL
    float cs0, cs1;                 // correction scale (out.std, = std_v unless --pretf)
    float ls0, ls1;                 // loss scale (std_v)
    unsigned long long* closs;      // exact accumulator ([SOL_LOSS_ACC_WORDS], loss_add_exact) of 0.5 * sum(((gt - v) / std)^2) over ALL faces, or NULL
    int ctr;                        // 1: the CNN runs on the transposed grid -- image row = the solver's x index, pixel = its y index (velocity [B,W+1,H] / [B,W,H+1])
    // trainer only (sol_conv5x5_seed): the 2 -> 32 backward-data layer of the reverse sweep computes its OWN input -- the loss-gradient seed of
    // the unrolled step (k_seed until round 6: one launch per unrolled step for 1.2 MB of elementwise work).  x is not read: the halo tile's
    // two channels are dO_c = cs_c * G_c with G_c = (v_c - gt_c) * sinv_m / ls_c^2 (+ gin_c), evaluated per halo pixel at the cell's low faces;
    // the workgroup of image row y also writes G (its row, plus v_y row Y from the last row's workgroup and the v_x column X face) and dO2
    const float *svy, *svx, *sginy, *sginx;   // v_i [B,H+1,W] / [B,H,W+1]; adjoint output of step i+1 (NULL at the last unrolled step); gt = gty / gtx
    float *sgy, *sgx, *sdO2;                  // G out (same shapes as v), dO2 out [B,H,W,2] (operand of the last layer's weight gradient)
    float sinv_m;                             // 1 / msteps
    int CI;                         // input channels per pixel of x as the caller declared them (0: not stated).  Kernels that read a fixed number of
                                    // channels per pixel whatever the caller meant (k_conv5x5_thin32: eight float4 = 32) check it before they are chosen
};
// correction mode: offsets of the faces of CNN pixel (image row jj, pixel px) inside a simulation's v_y / v_x, and of the face without
// a correction that this pixel also owns for the loss (v_y row Y, v_x column X of the SOLVER grid), or -1
struct CorrFaces { int oy, ox, ey, ex; };
__device__ __forceinline__ CorrFaces corr_faces(int tr, int H, int W, int jj, int px) {
    CorrFaces f;
    if (!tr) {          // solver grid Y = H, X = W
        f.oy = jj * W + px; f.ox = jj * (W + 1) + px;
        f.ey = jj == H - 1 ? f.oy + W : -1;
        f.ex = px == W - 1 ? f.ox + 1 : -1;
    } else {            // solver grid Y = W, X = H: cell (j = px, i = jj)
        f.oy = px * H + jj; f.ox = px * (H + 1) + jj;
        f.ey = px == W - 1 ? f.oy + H : -1;
        f.ex = jj == H - 1 ? f.ox + 1 : -1;
    }
    return f;
}
constexpr int SOL_AMAX_SLOTS = 256;   // one per workgroup of a 256-WG launch: same-address atomics serialise in L2 (~0.3 us each)
// backward-weight arguments
struct BwArgs {
    const float *x, *dz;
    float* partial;
    int B, H, W, cin, cout;
    int nblk;
    int nseg, rb;                 // nseg tensors of B*H rows each (the unrolled steps), rb image rows per workgroup
    long x_seg, dz_seg;           // element strides between consecutive segments
    int overwrite;                // 1: partial = acc (single launch), 0: partial += acc (accumulate over launches)
    const unsigned *xmax, *zmax;  // absmax slots of segment 0 of x / dz (NULL: bf16 six-product kernel) ...
    long xmax_seg, zmax_seg;      // ... and their strides (uint32 words) between consecutive segments
    int cin_real;                 // thin first layer (cin == 4): 1..3 = the channels beyond it are zero padding (their gradient rows are not computed); 0: all four are data
};
int sol_bww_sb_launch(hipStream_t s, const BwArgs& a, int nblk_run);
struct BwJobs {
    BwArgs a[5];
    int nrun[5];            // workgroups job k needs (<= wg_per; the surplus ones end at once)
    int n, wg_per;
};
int sol_bww_sb_jobs_launch(hipStream_t s, const BwJobs& p);
// n <= 5 independent 32 -> 32 weight-gradient passes (x[k], dz[k] -> partial[k], nplanes[k] x H rows of 64 pixels) in one launch (conv5x5.hip)
int sol_bww_batched_jobs(void* stream, int n, const float* const* x, const float* const* dz, float* const* partial, const int* nplanes, int overwrite,
                         int H, int W, const unsigned* xmax, const unsigned* zmax, int rb);
int sol_bww_pick_rb(int rows);
// ---- persistent chain of 32 -> 32 layers (cnn_chain.hip) ----
constexpr int SOL_CHAIN_MAXL = 12;
struct ChainLayer {
    const void* wsh;        // fp16 section of the packed weights: header {2^shift, 2^-shift, 0, 0} + planes [dy][dx][2][32 co][64 B]
    const float* bias;      // [32] or NULL
    const float* res;       // residual tensor [rows][64][32] or NULL
    const float* act;       // activation reference (SOL_EPI_DLRELU) or NULL
    float* y;               // output [rows][64][32]
    unsigned* ymax;         // per-tensor absmax slots (consumed by the weight-gradient kernels and the thin last layer) or NULL
    int epi;
};
struct ChainArgs {
    ChainLayer L[SOL_CHAIN_MAXL];
    int nl;
    const float* x0;        // input of the first layer
    unsigned* flags;        // hand-off region (sol_cnn_chain_flag_words)
    unsigned* err;          // |= 1 when a neighbour never delivered
    const unsigned* epoch;  // training-step epoch (part of the hand-off tag)
    unsigned long long* rmx;// [2 parities][rows]: {max|y| bits of the row, tag}
    uint4* xbuf;            // [2 parities][rows][64 px][8]: 16-byte granules of the rows' fp16 planes, tagged
    int B, H, nrows, ntiles;
    float slope;
};
bool sol_cnn_chain_usable(int B, int H, int W);
size_t sol_cnn_chain_flag_words(int B, int H, int nl);
int sol_cnn_chain_launch(hipStream_t s, const ChainLayer* layers, int nl, const float* x0, unsigned* flags, const unsigned* epoch, int B, int H, int W, float slope);
// fp16 section of a packed 32-input-channel weight buffer (sol_conv5x5_packed_floats layout)
const void* sol_conv_packed_wsh(const float* packed, int cout);
int sol_karman_step_bwd_fused(const sol_karman_cfg* cfg, void* stream,
                              const float* saved_vy, const float* saved_vx, const float* re, const float* active,
                              const float* velBCyMask, int64_t bc_batch_stride,
                              const float* g_vy_out, const float* g_vx_out, const float* dfeat, const float* feat_scale,
                              float* g_vy_in, float* g_vx_in, int32_t* iters, const BwArgs* bw, int nbw, int wg_per,
                              const struct SolDensRide* dens = nullptr);
// one passive-density advection (post-diffusion velocity svy / svx of its step) as B extra workgroups of a fused adjoint launch
// (k_karman_bwd_bww_small only: the 64x32 training path, where no forward launch can carry it)
struct SolDensRide { const float *d_in, *svy, *svx, *inflow; float* d_out; };
int sol_karman_bwd_fusable(const sol_karman_cfg* cfg);
int sol_karman_bwd_fusable_small(const sol_karman_cfg* cfg);     // the same for the 64 x 32 grid (k_karman_bwd_bww_small: 16 rows per gradient workgroup)
int sol_karman_step_fwd_dens(const sol_karman_cfg* cfg, void* stream,
                             const float* vy_in, const float* vx_in, const float* re, const float* active, const float* inflow,
                             const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                             float* vy_out, float* vx_out, float* saved_vy, float* saved_vx,
                             float* feat_out, const float* feat_scale, int32_t* iters,
                             const float* dens_d_in, const float* dens_svy, const float* dens_svx, float* dens_d_out, uint32_t* xchg = nullptr);
// band-split forward launch (k_karman_fwd_bands: four workgroups per simulation): usable for this configuration? / words of its hand-off region
int sol_karman_fwd_bands_usable(const sol_karman_cfg* cfg);
size_t sol_karman_fwd_bands_words(int B);
int sol_density_step(const sol_karman_cfg* c, void* stream, const float* d_in, const float* svy, const float* svx, const float* inflow, float* d_out);
int sol_bww_jobs_launch(void* stream, const BwArgs* bw, int nbw, int wg_per, const sol_karman_cfg* cfg = nullptr, const SolDensRide* dens = nullptr);
// weight-gradient job description for one layer of ONE unrolled step with `rb` rows per workgroup (train.hip -> fused launch)
int sol_bww_step_job(BwArgs* out, const float* x, const float* dz, float* partial, int overwrite, int B, int H, int W, int rb,
                     const unsigned* xmax, const unsigned* zmax);
int sol_bww_step_reduce(void* stream, const float* partial, float* dw_hwio, float* db, int B, int H, int rb, int cin, int cout, int accumulate);
int sol_bww_reduce_layers(void* stream, int n, float* const* partial, float* const* dw_hwio, float* const* db, const int* rows, const int* rb,
                          const int* cin, const int* cout, int accumulate, int taps_transposed);
size_t sol_bww_step_ws_floats(int B, int H, int rb);
// split-bf16 section of the packed weights and the kernels that consume it (conv5x5_sb.hip)
size_t sol_conv_sb_packed_floats(int OP);
size_t sol_conv_sh_packed_floats(int OP);
int sol_pack_jobs(hipStream_t s, int n, const float* const* w, float* const* out, float* const* bias_out, const float* const* bias_in,
                  const int* cin, const int* cout, const int* mode, int taps_transposed = 0);
int sol_conv_sh_pack(hipStream_t s, const float* w_hwio, int cin, int cout, int mode, void* out);
int sol_conv_sb_pack(hipStream_t s, const float* w_hwio, int cin, int cout, int mode, void* out);
int sol_conv_sb_launch(hipStream_t s, const ConvArgs& a, int NT, int ntiles);
// dx-major form of the 32 -> 32 fp16 three-product kernel (conv5x5_dx.hip; option conv_dx)
bool sol_conv_dx_usable(const ConvArgs& a, int NT, int ntiles);
// exact-fp32 VALU form of the thin 32 -> (<= 4) layers (conv5x5_thin.hip; option conv_thin_valu)
bool sol_conv_thin32_usable(const ConvArgs& a, int NT);
int sol_conv_thin32_launch(hipStream_t s, const ConvArgs& a, int ntiles);
int sol_conv_dx_launch(hipStream_t s, const ConvArgs& a, int ntiles);

// fused 5x5x5 convolution, 32 -> 32 or 32 -> (<= 16) channels, W == 64 (conv3d_sb.hip)
size_t sol_conv3d_sh_packed_floats(int cout);
int sol_conv3d_sh_pack(hipStream_t s, const float* w_dhwio, int mode, int cout, float* out);
int sol_conv3d_sb_launch(hipStream_t s, const float* x, const float* wsh, const float* bias, const float* residual, const float* act_ref, float* y,
                         int B, int D, int H, int cout, int epilogue, float slope, const unsigned* x_absmax, unsigned* y_absmax);

// ---- process-wide options (sol_set_option / sol_get_option, include/sol_hip.h) ----------------------------------
// Read at every call (plain ints, no caching in function-local statics), so a host may switch e.g. the convolution
// precision between two trainers of one process.  The library itself never reads the environment.
struct SolOptions {
    int conv_precision;   // 0 split (default): fp16 x3 where the operand's absmax is known, bf16 x6 otherwise; 1: bf16 x6 always; 2: strict fp32 MFMA
    int conv_split3;      // experiment: three leading bf16 products only (NOT fp32 equivalent)
    int conv_r3;          // 1: 3-row fp32 MFMA kernel for 32-channel layers (strict path)
    int conv_thin;        // 1: folded-tap fp32 kernels for the thin weight gradients
    int conv_bww32;       // 1: 32x32x2 fp32 MFMA weight gradient (strict path)
    int correct_fuse;     // 1: correction + loss in the last CNN layer's epilogue
    int bww_fuse;         // 1: 32->32 weight gradients ride in the solver-adjoint launches
    int bww_chunk;        // 0: one weight-gradient launch per layer over all unrolled steps; n: chunks of n steps
    int bww_side;         // 1: chunked weight gradients on a side stream
    int streams;          // sub-batch chains on separate streams (measured slower; default 1)
    int density_mode;     // 0: density advection rides in the next solver launch; 1: inline in the step kernel; 2: one chain launch after the unroll
    int cpt;              // 0: automatic strip height of the CG kernels, 8 / 16: forced
    int dbg_skip;         // timing experiments only
    int step_prof;        // debugging: synchronous phase times of the solver step kernels on stderr
    int cnn_persistent;   // 1: the ten 32 -> 32 layers of a CNN pass as ONE persistent launch (cnn_chain.hip) where the shape allows it; default 0:
                          //    measured equal to the per-layer launches end to end, kept as a verified experiment
    int graph_stream;     // 1: sol_train_graph_launch replays on an internal stream fenced by events against the caller's stream
    int k3d_fused_tf;     // 1 (default): the sine transforms of the karman-3d pressure solve as LDS-resident plane / column-slab kernels; 0: batched GEMMs
    int k3d_conv_rows;    // rows per workgroup of the one-launch Conv3D kernel: 8 (k_conv3d_sb8, 64 x 32 wave tiles), 6 (k_conv3d_sb6, 32 x 32), 3 (k_conv3d_sb, 16 x 32)
    int k3d_conv_fused;   // 1 (default): 32 -> 32 Conv3D layers with W == 64 and a known operand absmax as ONE launch (conv3d_sb.hip); 0: five passes of the 2-D kernel
    int k3d_mfma_tf;      // 1 (default): the LDS-resident sine transforms of the karman-3d pressure solve on the fp32 matrix cores (k3_ty_mfma, k3_tzx_mfma)
    int conv_dx;          // bit 0: the 32 -> 32 fp16 three-product convolutions run the dx-major kernel (conv5x5_dx.hip: a wave owns a pixel segment of all
                          //    three output rows, 0.53 LDS operand reads per MFMA); 0: k_conv5x5_sb (one output row per wave, 1.0 reads per MFMA).
                          //    bit 1: also the thin 32 -> (<= 16) layers where a workgroup owns one row (small launches); bit 2: those layers in every
                          //    launch (measured slower where the launch fills the chip); bit 3: the 32 -> 32 layers of one-row-per-workgroup launches as two
                          //    half-channel workgroups per tile (k_conv5x5_dx<1, 1, true>).  Default 11.
    int conv_thin_valu;   // 1 (default): the thin 32 -> (<= 4) layers of 64-pixel images (the trainer's output layer with the correction epilogue, the first
                          //    layer's data gradient) in exact fp32 on the vector ALU (conv5x5_thin.hip: no absmax wait, no operand split); 0: k_conv5x5_sb<1, KIND>
    int seed_fuse;        // 1 (default): on 64-pixel rows the trainer's loss-gradient seed is computed inside the 2 -> 32 backward-data launch (sol_conv5x5_seed)
                          //    instead of a k_seed launch per unrolled step
    int conv_thin_t3;     // 1 (default): the thin-INPUT layers of 64-pixel images (first layer, last backward-data layer incl. seed mode) as three image rows per
                          //    twelve-wave workgroup (k_conv5x5_t3); 0: k_conv5x5<4, NT>, one row per workgroup
    int k3d_adj_tile;     // 1 (default): the karman-3d advection adjoint scatters into an int64 LDS window per workgroup (k3b_advect_adj_tile); 0: global atomics only
    int k3d_conv_persist; // 1: the one-launch Conv3D kernel runs consecutive tiles per workgroup (256 workgroups) when the tile count is a multiple of 256.
                          //    Default 0: measured SLOWER (SOL-16 174.8 vs 163.8 ms same box, profiles/r06_k3d_conv_persist_ab.txt) -- the tile loop around the
                          //    unrolled tap rows costs 68 spilled VGPRs (8 without) in a kernel that sits at the 256-register limit of two waves per SIMD
    int k3d_bww_jobs;     // the five depth slices of a 32 -> 32 Conv3D weight gradient: 2 (default) ONE launch of ONE round of workgroups (51 per slice, each
                          //    with a fifth of the rows of its slice: a fifth of the prologues and of the partial read-modify-writes); 1: one launch,
                          //    five rounds of 32-row workgroups (k_conv5x5_bww_sb_jobs); 0: five launches
    int fwd_bands;        // 1 (default): the 128 x 64 forward solver step of the training / roll-out path as FOUR workgroups per simulation (k_karman_fwd_bands:
                          //    stencil phases on row bands with recomputed halos, the direct solve on band 0's CU, two hand-offs through global memory); 0: one workgroup
    int k3d_tile;         // 1: karman-3d advection from LDS tiles holding the full z column + halo; 0 (default, measured faster at B <= 2): wave-per-column gathers from global memory
};
SolOptions& sol_opt();

// ---- launch profiler (sol_prof_begin / sol_prof_end) ---------------------------------------------------------------
// While active, every kernel launched through SOL_LAUNCH carries its own start/stop HIP events (hipExtLaunchKernelGGL:
// the dispatch packet's begin / end timestamps, i.e. what rocprofv3 --kernel-trace reports) on the stream it is
// launched on.  Not usable under stream capture: the trainer's eager path is profiled.
bool sol_prof_events(const char* name, hipEvent_t* a, hipEvent_t* b);
extern bool g_sol_prof_on;

template <typename... KA, typename... A>
inline void sol_launch_impl(const char* name, void (*kernel)(KA...), dim3 g, dim3 b, size_t lds, hipStream_t s, A&&... args) {
    hipEvent_t ea, eb;
    if (g_sol_prof_on && sol_prof_events(name, &ea, &eb))
        hipExtLaunchKernelGGL(kernel, g, b, (uint32_t)lds, s, ea, eb, 0u, static_cast<KA>(args)...);
    else
        hipLaunchKernelGGL(kernel, g, b, lds, s, static_cast<KA>(args)...);
}
#define SOL_LAUNCH(kernel, ...) sol_launch_impl(#kernel, kernel, __VA_ARGS__)
#define SOL_LAUNCH_NAMED(name, kernel, ...) sol_launch_impl(name, kernel, __VA_ARGS__)

#define SOL_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return sol_set_error(SOL_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                \
                                 hipGetErrorString(e_), __FILE__, __LINE__);                 \
    } while (0)
#define SOL_LAUNCH_CHECK() SOL_HIP_CHECK(hipGetLastError())
#define SOL_REQUIRE(cond, ...)                                                               \
    do {                                                                                     \
        if (!(cond)) return sol_set_error(SOL_ERR_ARG, __VA_ARGS__);                         \
    } while (0)

// Dynamic LDS beyond 64 KB is an opt-in PER KERNEL AND PER DEVICE (hipFuncSetAttribute acts on the current device): every call
// site keeps one bit per device ordinal and repeats the opt-in when the process has switched devices (a function-local
// `static int rc = hipFuncSetAttribute(...)` only ever covered the device of the first call).  minus_static: the kernel's static
// LDS counts against the same 160 KB.  Not a stream operation: legal during graph capture.
#define SOL_K(...) reinterpret_cast<const void*>(__VA_ARGS__)
inline int sol_lds_optin(std::atomic<unsigned long long>& done, std::initializer_list<const void*> kernels, const char* what, bool minus_static = false) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return sol_set_error(SOL_ERR_HIP, "hipGetDevice failed (%s)", what);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return SOL_OK;
    for (const void* k : kernels) {
        int bytes = 160 * 1024;
        if (minus_static) {
            hipFuncAttributes fa;
            if (hipFuncGetAttributes(&fa, k) != hipSuccess) return sol_set_error(SOL_ERR_HIP, "hipFuncGetAttributes(%s) failed", what);
            bytes -= (int)fa.sharedSizeBytes;
        }
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
            return sol_set_error(SOL_ERR_HIP, "hipFuncSetAttribute(%s) failed on device %d", what, dev);
    }
    done.fetch_or(bit, std::memory_order_release);
    return SOL_OK;
}

// ---- write-through output stores (round 5) ------------------------------------------------------------------------------------
// gfx950 = 8 XCDs, each with its own L2 that is NOT coherent with the others: a kernel's plain stores stay dirty in the XCD's L2 and
// the release at the END of the kernel writes them all back in one burst -- in front of the next launch of the dependency chain, which
// is what every launch of this path is.  Measured on the 32 -> 32 convolution (6.3 MB of output per launch, 640 launches per SOL-32
// step; tools/ab_lib.py, one box, three alternations): plain 13.128 ms per step, nontemporal (nt) 13.103, agent-scope write-through (sc1)
// 12.775, system scope (sc0 sc1) 12.808.  With sc1 the lines leave the L2 while the other workgroups still compute and the end-of-kernel
// write-back finds nothing to do.  Use for a kernel's BULK outputs that the NEXT launch consumes, in full 16-byte pieces (scattered
// 4-byte write-through stores -- the thin layers' strided channels, the velocity correction -- measured SLOWER: 10.5 -> 11.0 us per launch); not for read-modify-write sequences of
// one thread on one address (the compiler does not see these stores in its vmcnt bookkeeping: extra stores only make its waits more
// conservative, but it will not order a later load of the same address behind them; the 16-byte form carries its own wait states).
// SOL_WT_STORES=0 (A/B builds): plain stores everywhere.
#ifndef SOL_WT_STORES
#define SOL_WT_STORES 1
#endif
typedef float sol_f32x4 __attribute__((ext_vector_type(4)));
typedef float sol_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_wt(float4* p, const float4& v) {
#if SOL_WT_STORES
    const sol_f32x4 q = {v.x, v.y, v.z, v.w};
    // s_nop 1: a VMEM store of more than 8 bytes followed by a VALU write of its data VGPRs needs 2 wait states on gfx940+; the
    // compiler's hazard recognizer does not look inside inline asm (without it: an unrolled store loop reused the data registers
    // at once and the weight-gradient partials came out wrong, tests/test_gpu_parity.py::test_conv5x5_against_oracle)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(q) : "memory");
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_wt(float2* p, const float2& v) {
#if SOL_WT_STORES
    const sol_f32x2 q = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(q) : "memory");
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_wt(float* p, float v) {
#if SOL_WT_STORES
    asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}

// wave64 all-reduce (every lane gets the sum)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Bit-reproducible sum over launches (the per-step l2 loss): an EXACT accumulator.  A workgroup's sum s >= 0 (itself formed in a fixed
// order) is the integer m * 2^(e - 150) (24-bit mantissa m, biased exponent e); it is added as m << (e & 7) to the 64-bit integer
// acc[e >> 3] (32 buckets of eight binades, units 2^(8 b - 150)): integer addition is exact and associative, so the result does not
// depend on the order in which workgroups -- or launches, or sub-batch streams -- arrive, and nothing is rounded before the final
// conversion (k_loss_finish: the buckets in ascending order, in double).  One fire-and-forget atomic per workgroup: no ticket, no
// fence, nobody waits.  (First form of the round: per-workgroup partials + a release/acquire ticket, the last workgroup folding them
// in order -- also reproducible, but every workgroup's tail carried an agent-scope release (L2 write-back) and an atomic round trip.)
// Capacity: m << 7 < 2^31 per addition, 2^32 additions per bucket.
// Non-finite sums are NOT dropped: a workgroup sum that is inf / nan (a diverged run) -- or negative, which no sum of squares can be --
// sets the POISON bit (bit 63 of the last bucket, out of reach of 2^32 finite additions) and loss_acc_value() then returns NaN, like the
// float atomics this replaced and like the reference's tf.nn.l2_loss (karman_train.py:430-436): monitoring keyed on a non-finite loss fires.
constexpr int SOL_LOSS_ACC_WORDS = 32;                           // unsigned long long per loss value
constexpr unsigned long long SOL_LOSS_POISON = 1ull << 63;       // in acc[SOL_LOSS_ACC_WORDS - 1]
__device__ __forceinline__ void loss_add_exact(float wg_sum, unsigned long long* __restrict__ acc) {
    const unsigned bits = __float_as_uint(wg_sum);
    const unsigned e = (bits >> 23) & 0xffu;
    if ((bits << 1) == 0u) return;                               // +-0: nothing to add
    if (e == 0xffu || (bits >> 31)) { atomicOr(&acc[SOL_LOSS_ACC_WORDS - 1], SOL_LOSS_POISON); return; }
    const unsigned long long m = (bits & 0x7fffffu) | (e ? 0x800000u : 0u);
    const unsigned ee = e ? e : 1u;                              // denormals share the exponent of the smallest normal
    atomicAdd(&acc[ee >> 3], m << (ee & 7u));
}
__device__ __forceinline__ float loss_acc_value(const unsigned long long* __restrict__ acc) {
    if (acc[SOL_LOSS_ACC_WORDS - 1] & SOL_LOSS_POISON) return __uint_as_float(0x7fc00000u);
    double v = 0.0;
    for (int b = 0; b < SOL_LOSS_ACC_WORDS; ++b) v += ldexp((double)acc[b], 8 * b - 150);
    return (float)v;
}
// Barrier-free workgroup stage in front of it: every wave (all lanes) calls this with its lanes' sums; `lds` = {wave ticket (zeroed before
// the kernel's first barrier), pad, pad, pad, one slot per wave}.  The wave that draws the last ticket adds the slots in wave order.
__device__ __forceinline__ void loss_publish_last(float lsum, unsigned long long* __restrict__ acc, unsigned* lds) {
    lsum = wave_sum(lsum);
    const int nw = blockDim.x >> 6;
    unsigned last = 0u;
    if ((threadIdx.x & 63) == 0) {
        reinterpret_cast<float*>(lds)[4 + (threadIdx.x >> 6)] = lsum;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        last = atomicAdd(&lds[0], 1u) == (unsigned)(nw - 1) ? 1u : 0u;       // the LDS unit executes in arrival order: the slot is written before the ticket is drawn
    }
    last = (unsigned)__shfl((int)last, 0, 64);
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += reinterpret_cast<volatile float*>(lds)[4 + w];
    if ((threadIdx.x & 63) == 0) loss_add_exact(s, acc);
}

// XCD-aware tile order: the hardware deals consecutive workgroup ids round robin to the 8 XCDs (each with its own L2), so
// workgroup w runs on XCD w % 8.  Mapping it to tile (w % 8) * (n / 8) + w / 8 gives every XCD one CONTIGUOUS block of
// image rows: a convolution's halo rows, and the rows the previous layer's launch wrote, are then in the SAME L2 except
// at the 8 block seams.  (n not a multiple of 8: identity.)
__device__ __forceinline__ int xcd_tile(int w, int n) {
#ifdef SOL_NO_XCD_REMAP
    return w;
#else
    return (n & 7) == 0 ? (w & 7) * (n >> 3) + (w >> 3) : w;
#endif
}

// Workgroup all-reduce through LDS.  `red` holds 2 x 32 floats; `slot` alternates 0/1
// between consecutive calls so that no extra barrier is needed against the previous use.
// Every thread returns the bit-identical sum (same summation order), so branches on it
// are workgroup-uniform.
__device__ __forceinline__ float block_sum(float v, float* red, int slot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    float* r = red + slot * 32;
    if (lane == 0) r[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += r[w];
    return s;
}

// max over the SOL_AMAX_SLOTS slots of an absmax array (bits of non-negative floats) -> power-of-two scale 2^shift with
// max * 2^shift in [2^14, 2^15), and its inverse
// two halves so that a kernel can put other loads between the slot load and its first use
__device__ __forceinline__ uint4 amax_load(const unsigned* slots) {
    static_assert(SOL_AMAX_SLOTS == 256, "one uint4 per lane");
    return reinterpret_cast<const uint4*>(slots)[threadIdx.x & 63];
}
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ unsigned amax_dpp(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, true);
}
__device__ __forceinline__ void amax_scale_of(const uint4& q, float& scale, float& inv) {
    unsigned m = max(max(q.x, q.y), max(q.z, q.w));
    // wave maximum with DPP row operations (six dependent ds_bpermute round trips of a __shfl_xor ladder cost ~0.3 us per launch)
    m = max(m, amax_dpp<0xB1>(m));          // quad_perm [1,0,3,2]
    m = max(m, amax_dpp<0x4E>(m));          // quad_perm [2,3,0,1]
    m = max(m, amax_dpp<0x141>(m));         // row_half_mirror
    m = max(m, amax_dpp<0x140>(m));         // row_mirror
    m = max(m, amax_dpp<0x142, 0xa>(m));    // row_bcast:15
    m = max(m, amax_dpp<0x143, 0xc>(m));    // row_bcast:31 -> lane 63 = wave maximum
    m = (unsigned)__builtin_amdgcn_readlane((int)m, 63);
    int e = (int)(m >> 23) - 127;                     // max in [2^e, 2^(e+1))
    e = m == 0u ? 0 : min(max(e, -100), 100);
    scale = __uint_as_float((unsigned)(14 - e + 127) << 23);
    inv = __uint_as_float((unsigned)(e - 14 + 127) << 23);
}
__device__ __forceinline__ void amax_scale(const unsigned* slots, float& scale, float& inv) {
    const uint4 q = amax_load(slots);
    amax_scale_of(q, scale, inv);
}
__device__ __forceinline__ unsigned amax_wave_max(unsigned m) {       // lane 63 (and the return value) = wave maximum
    m = max(m, amax_dpp<0xB1>(m));
    m = max(m, amax_dpp<0x4E>(m));
    m = max(m, amax_dpp<0x141>(m));
    m = max(m, amax_dpp<0x140>(m));
    m = max(m, amax_dpp<0x142, 0xa>(m));
    m = max(m, amax_dpp<0x143, 0xc>(m));
    return (unsigned)__builtin_amdgcn_readlane((int)m, 63);
}
// one atomic per workgroup: `v` = this thread's max|y|; `red` = 16 floats of LDS
__device__ __forceinline__ void amax_publish(float v, unsigned* slots, float* red) {
    v = __uint_as_float(amax_wave_max(__float_as_uint(fabsf(v))));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
        atomicMax(&slots[blockIdx.x & (SOL_AMAX_SLOTS - 1)], __float_as_uint(m));
    }
}


// barrier-free form: `lds2` = {running max, wave counter}, zeroed before the kernel's first barrier.  Every wave folds its max
// into lds2[0] (LDS integer atomics are cheap) and counts itself; the LDS unit executes these in arrival order, so the wave
// that draws the last ticket reads the complete workgroup max and issues the single global atomic.  Nobody waits for anybody.
__device__ __forceinline__ void amax_publish_last(float v, unsigned* slots, unsigned* lds2) {
    const unsigned wm = amax_wave_max(__float_as_uint(fabsf(v)));
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&lds2[0], wm);
        const unsigned ticket = atomicAdd(&lds2[1], 1u);
        if (ticket == (blockDim.x >> 6) - 1) atomicMax(&slots[blockIdx.x & (SOL_AMAX_SLOTS - 1)], atomicMax(&lds2[0], 0u));
    }
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
