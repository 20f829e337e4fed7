// 5x5 SAME convolution, 32 -> 32 channels, fp16 three-product arithmetic (KIND 2 of conv5x5_sb.hip: same operand splits, same
// per-product arithmetic, same packed weight buffer) in a "dx-major" schedule that makes the tap loop matrix-pipe bound.
//
// Why (round-3 verdict): in k_conv5x5_sb every wave owns one 16 px x 32 co tile of ONE output row and reads
// six 16-byte operands from LDS for six MFMAs per tap -- 72 ds_read_b128 per tap and CU (~390 clocks of the LDS pipe) for 288
// clocks of MFMA per SIMD: the loop is operand bound.  The reuse that is left lies ACROSS OUTPUT ROWS: input row s meets output
// row j at tap row dy = s - j, so one pixel operand A(s, dx) serves up to three output rows, and one weight operand B(dy, dx)
// serves all seven input rows of a workgroup.  Here a wave owns a 16-pixel segment of ALL THREE output rows of the workgroup
// and one 16-channel output tile (8 waves = 4 segments x 2 channel tiles, two waves per SIMD):
//   for dx in 0..4:                                    (phase: one barrier each; the 20 KB weight set of a phase is double buffered)
//       B(dy, dx), dy = 0..4        -> 40 VGPRs        (10 ds_read_b128)
//       for s in 0..6:  A(s, dx)                       ( 2 ds_read_b128)
//           for j with 0 <= s - j <= 4:  three MFMAs   (45 MFMAs per phase)
// = 24 operand reads per 45 MFMAs (0.53 per MFMA instead of 1.0): 960 reads per launch and CU instead of 1800, under 7200
// clocks of MFMA per SIMD.  All seven input rows are staged ONCE in the prologue (no row staging inside the loop), the
// accumulators come out as D[co][px] (weights as the A operand of the MFMA), i.e. a lane holds four consecutive output channels
// of one pixel: the epilogue stores 16-byte pieces straight from registers -- no LDS transposition, no epilogue barrier, and the
// residual / activation-reference operands are prefetched into registers in the same layout.
//
// Forms (template arguments R, COT, SPLIT; chosen by sol_conv_dx_launch, option conv_dx):
//   <3, 2>        the form above: launches that fill the chip with three-row workgroups (128x64, B = 6: 256 workgroups);
//   <1, 2>        one row per workgroup for small launches; five LDS rows leave room for ALL five weight sets (resident weights: one
//                 barrier behind the prologue's, phases 1..4 barrier free);
//   <1, 1, true>  ... as two half-channel workgroups per tile (each stages 50 of the 100 KB of weights; waves 4..7 stage only);
//   <R, 1>        thin layers (<= 16 output channels: the 32 -> 2 output layer in correction mode -- velocity update + l2 loss in the
//                 epilogue --, the 32 -> 3 data gradient); used in one-row launches, k_conv5x5_sb<1, 2> is faster in chip-filling ones.
//
// Replaces keras.layers.Conv2D(32, 5, padding='same') (+ bias, LeakyReLU, residual add) of model_mars_moon
// (/root/reference/karman-2d/karman_train.py:101-138) for the ten 32 -> 32 layers and the output layer, forward and backward-data.
#include "split_kernels.hpp"

namespace {

using namespace sbk;

constexpr int DX_HWP = 68;                         // halo pixels per row (64 + 4)
constexpr int DX_PLANE = DX_HWP * 64;              // bytes per fp16 plane of one halo row
constexpr int DX_SLOT = 2 * DX_PLANE;              // hi + lo plane
// R output rows per workgroup need R + 4 input rows.  R = 3 (one workgroup per CU at 128x64 x 6) is the form described above; R = 1
// serves launches with few image rows (the reference's 64x32 x 3 recipe: 96 rows = 32 workgroups of three rows on 256 CUs, every
// launch pure latency): three times the workgroups, a third of the loop each -- the rows no longer share pixel operands, which does
// not matter where the loop is a fraction of the launch.
constexpr int dx_wpl(int COT) { return COT * 16 * 64; }        // bytes per (tap, plane) weight block: 16 COT co x 32 ci fp16
constexpr int dx_wph(int COT) { return 5 * 2 * dx_wpl(COT); }  // bytes per dx phase: five dy x two planes
constexpr int dx_wbufs(int R) { return R == 1 ? 5 : 2; }       // one row per workgroup: all five weight sets stay resident (see RW in the kernel)
constexpr int dx_lds(int R, int COT) { return (R + 4) * DX_SLOT + dx_wbufs(R) * dx_wph(COT) + 16 + 64; }    // + absmax words, loss_publish_last's 4 + 12

// DX_ACL2: the two cross products of a tap (h1 h2', h2' h1) accumulate into SEPARATE registers that are added in the epilogue.  With one
// accumulator a row's two cross-term MFMAs of step s and the first of step s + 1 are 6 and then 3 issue slots apart (1-row steps: 2), and a
// dependent v_mfma_f32_16x16x32_f16 issues ~4 slots after its producer (tools/mfma_dep_distance.py): bubbles only the other wave can fill.
#ifndef DX_ACL2
#define DX_ACL2 0
#endif
#define DX_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// -DSOL_CONV_PROF (tools/conv_dx_probe.py builds such a library next to the product one): phase stamps (100 MHz s_memrealtime,
// thread 0 of every workgroup, 16 per workgroup) into the buffer set with sol_conv_dx_prof_set().
#ifdef SOL_CONV_PROF
__device__ long long* g_dx_prof = nullptr;          // [cap launches][grid][16] stamps; g_dx_prof_ctl = {launch counter, cap}
__device__ unsigned g_dx_prof_ctl[2] = {0u, 1u};
// (the buffer pointer is read ONCE into scalar registers: a stamp that reloads it costs a scalar-memory round trip, ~0.5 us each)
#define DX_STAMP(k) do { if (threadIdx.x == 0 && dx_prof) dx_prof[(k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DX_STAMP(k) do { } while (0)
#endif

// COT: 16-channel output tiles.  2: the 32 -> 32 layers (waves = 4 pixel segments x 2 tiles).  1: the thin layers (<= 16 output channels:
// the 32 -> 2 output layer in correction mode and its data gradient's counterpart) -- waves 0..3 own the four pixel segments and all the
// MFMAs, waves 4..7 are the "late" staging role only.
// SPLIT (COT = 1 only): a 32 -> 32 layer as TWO workgroups per tile, each with one 16-channel half of the output -- for the small launches
// (one row per workgroup, 96..128 tiles on 256 CUs): a workgroup then stages half the weights (50 instead of 100 KB; the weights are
// 70 % of what a one-row workgroup pulls through its CU's 64 B/clock vector-memory path) and the launch uses twice the CUs.
template <int R, int COT, bool SPLIT = false>
__global__ void __launch_bounds__(512) k_conv5x5_dx(const float* __restrict__ arg_x, const void* __restrict__ arg_wsh, const unsigned* __restrict__ arg_xmax,
                                                      int nrows, int arg_H, int arg_W, int arg_tiles_x, int arg_hshift, ConvArgs a) {
    // What the prologue needs before its first request -- the three operand pointers and the tile geometry -- are LEADING scalar
    // kernel arguments: the file is compiled with -amdgpu-kernarg-preload-count=11, so the command processor delivers them in SGPRs
    // with the wave (gfx950 kernel-argument preload) and no scalar-memory round trip stands in front of the requests; the rest
    // of ConvArgs (epilogue operands) is fetched lazily in their shadow.
    a.x = arg_x; a.wsh = arg_wsh; a.xmax = arg_xmax; a.H = arg_H; a.W = arg_W; a.tiles_x = arg_tiles_x; a.RPW = arg_hshift;
    // (Tried: all kernel arguments fetched in ONE batch of scalar loads pinned at the entry instead of the compiler's three dependent
    // batches.  The timeline behind the first stamp improved by 0.1 us -- and every launch became 0.5 us LONGER (rocprof 11.2 ->
    // 11.7 us in the step): the one big batch is waited for before anything is issued, while the lazy batches overlap with the
    // first vector requests.  Reverted.)
    constexpr int DX_NS = R + 4;                                // input rows (LDS slots) of R output rows
    constexpr int DX_WPL = dx_wpl(COT), DX_WPH = dx_wph(COT);
    constexpr int NBLK = COT * 128;                             // uint4 per tap of the packed weights: 2 planes x 16 COT co x 4
    constexpr int TAPQ = (COT == 2 || SPLIT) ? 256 : 128;       // uint4 per tap of the packed SOURCE (32 or 16 padded output channels)
    static_assert(COT == 1 || COT == 2, "one or two 16-channel output tiles");
    static_assert(!SPLIT || COT == 1, "SPLIT: one output tile per workgroup");
    static_assert(R >= 1 && R <= 3, "one to three output rows per workgroup");
    extern __shared__ __align__(16) unsigned char smem_dx[];
    unsigned char* const ring = smem_dx;                        // [7 slots][2 planes][68 px][64 B]
    unsigned char* const Wt = smem_dx + DX_NS * DX_SLOT;        // [2 (RW: 5) buffers][5 dy][2 planes][16 COT co][64 B]
    // RW ("resident weights", one-row workgroups): the five-slot ring leaves room for ALL five weight sets (43.5 + 100 KB), so the sets of
    // phases 2..4 get their own buffers, are written once (phase 1) and the phases 2, 3 run without a barrier and read their next set
    // a whole phase ahead.  In those launches (96..128 workgroups: the 64x32 recipe, roll-outs at B = 1) a phase is five steps of three
    // MFMAs and the barrier + store + exposed operand read of a double-buffered phase cost as much as its MFMAs (phase timeline,
    // tools/conv_dx_probe.py 3 32 64: phases 1..3 0.6-0.7 us each, phase 4 -- no barrier -- 0.26 us).
    constexpr bool RW = dx_wbufs(R) == 5;
    unsigned* const amax_lds = reinterpret_cast<unsigned*>(smem_dx + DX_NS * DX_SLOT + dx_wbufs(R) * DX_WPH);
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // (Tried, round 5: gridDim.x is one scalar load from the hidden part of the kernarg segment, and the tile coordinates -- every row request --
    //  wait for it.  The grid size as a twelfth preloaded argument: 13.09 ms per step against 12.96 with the load; packed into the spare
    //  bits of arg_hshift (no extra preload dword): 12.53 against 12.46.  Same-box A/Bs of four alternations each, both SLOWER without
    //  the load, like the one-batch argument fetch below.  gridDim.x stays.)
    int bid = blockIdx.x, ngrid = gridDim.x, half = 0;          // SPLIT: workgroups [0, n) own output channels 0..15 of tile bid, [n, 2n) channels 16..31
    if (SPLIT) { ngrid >>= 1; half = bid >= ngrid ? 1 : 0; bid -= half * ngrid; }
    const int seg = wid & 3;                                    // 16-pixel segment (wave uniform)
    const int wt = COT == 2 ? wid >> 2 : 0;                     // the wave's 16-channel tile inside the workgroup's staged weights
    const int cot = SPLIT ? half : wid >> 2;                    // ... and among the layer's output channels
    // position r of a staged tap block [2 planes][16 COT co][4] inside the source's tap block [2 planes][32 or 16 co][4]
    auto wpos = [&](int r) { return SPLIT ? ((r >> 6) * 128 + half * 64 + (r & 63)) : r; };
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H, W = a.W;
#ifdef SOL_CONV_PROF
    long long* __restrict__ dx_prof = g_dx_prof;
    if (dx_prof) dx_prof += ((size_t)(g_dx_prof_ctl[0] % g_dx_prof_ctl[1]) * gridDim.x + blockIdx.x) * 16;
#endif
    DX_STAMP(0);
    if (tid == 0) *reinterpret_cast<uint4*>(amax_lds) = make_uint4(0u, 0u, 0u, 0u);      // absmax word + ticket, loss sum + ticket
    // What needs no tile coordinates is requested first, in the shadow of the scalar preamble: the absmax slots and the role's weight set
    // (phase 0 / phase 1: piece k = dy block k, position t).  NAMED registers: an array captured by the staging lambda was put into
    // scratch memory (a store behind every load: the requests serialised, first MFMA at 5.4 us).
    const bool late = wid >= 4;                                  // wave uniform (SGPR): staging role, see below
    const int t = tid & 255;
    const uint4 am = amax_load(a.xmax);
    const uint4* gw = reinterpret_cast<const uint4*>(a.wsh) + 1;          // behind the header {2^shift_w, 2^-shift_w, 0, 0}
    // item i = t + 256 k of the role's phase (5 NBLK items: tap (i / NBLK, dxr), position i % NBLK); COT == 1 has 640: k = 3, 4 and the
    // upper half of k = 2 repeat item 0 and are dropped
    auto wsrc_of = [&](int k) {
        int i = t + 256 * k;
        if (i >= 5 * NBLK) i = 0;
        return gw + (size_t)((i / NBLK) * 5 + (late ? 1 : 0)) * TAPQ + wpos(i % NBLK);
    };
    const uint4 wq0 = *wsrc_of(0), wq1 = *wsrc_of(1), wq2 = *wsrc_of(2), wq3 = *wsrc_of(COT == 2 ? 3 : 0), wq4 = *wsrc_of(COT == 2 ? 4 : 0);
    const float winv = reinterpret_cast<const float*>(a.wsh)[1];
    // Scalar preamble, kept short: it runs before the first request can go out (its first form -- four integer divisions on the
    // VALU with read-first-lane round trips and a compare chain per (s, j) pair, 330 instructions -- cost 1.2 us of every launch).
    const int bx = xcd_tile(bid, ngrid);
    int tx = 0, ty = bx;
    if (a.tiles_x != 1) { tx = bx % a.tiles_x; ty = bx / a.tiles_x; }      // (W == 64: one column block, no division)
    const int G0 = ty * R, x0 = tx * 64;
    // image of row G0: H is a power of two in every shipped configuration (a.RPW = log2 H, set by the launcher; -1: one division)
    const int b0 = a.RPW >= 0 ? (G0 >> a.RPW) : G0 / H;
    const int lo0 = b0 * H, hi0 = lo0 + H;
    // validity of (input slot s, output row j): the input row G0 - 2 + s must lie in the image of output row G0 + j; the three
    // rows span at most two images, and for one j the valid s form a range
    unsigned vmask = 0u;                                        // bit j * 8 + s  (wave uniform: SGPR)
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int gj = G0 + j;
        const int lo = gj >= hi0 ? hi0 : lo0;                   // first row of gj's image
        const int smin = max(j, lo - (G0 - 2)), smax = min(j + 4, lo + H - 1 - (G0 - 2));
        if (gj < nrows && smax >= smin) vmask |= (((2u << smax) - 1u) & ~((1u << smin) - 1u)) << (j * 8);
    }
    vmask = __builtin_amdgcn_readfirstlane(vmask);

    const float4* gx = reinterpret_cast<const float4*>(a.x);
    // ---- prologue: every request goes out before anything is waited for ------------------------------------------------------
    // Staging by ROLE.  A CU's vector-memory path moves ~64 B per clock and a wave issues in order: the 105 KB a workgroup needs
    // before its second barrier (seven input rows, the weight sets of phases 0 and 1) take ~0.85 us to pass through it, and a wave
    // that has to push its share of ALL of it through before it may touch the first row that came back sits on the critical path
    // for all of that time (first form of this kernel: first MFMA 3.7 us after the launch).  So waves 0..3 ("early") request, split
    // and stage only what the first three steps read -- rows 0..2, their halo pixels, the weight set of phase 0 -- and waves 4..7
    // ("late") request rows 3..6, their halo and the weight set of phase 1, go straight to the first barrier, run steps 0..2 and
    // stage their share behind them (stage_late, in front of the phase's own barrier), when it has long arrived.
    // Thread map of a role (256 threads): channel quad c4 = t & 7 of pixels p, p + 32 (p = t >> 3) of each of the role's rows --
    // one base address, constant LDS destination + immediates, no predicate (W % 64 == 0; rows outside [0, nrows) come from a
    // clamped row and are never used: `vmask` masks every pair that would read them).  Halo pixels (x0 - 2, x0 - 1, x0 + 64,
    // x0 + 65; zero unless a neighbouring column block exists): one item for the first 32 threads per row of the role.
    const int c4 = t & 7, p = t >> 3;
    const int s_base = late ? 3 : 0, n_rows = late ? DX_NS - 3 : 3;
    const int hrow = t >> 5, hp = (t >> 3) & 3, hc_h = hp < 2 ? hp : hp + 64;     // halo item: row s_base + hrow (if hrow < n_rows), halo position hc_h
    const int hxx = x0 + hc_h - 2;
    const bool hok = hxx >= 0 && hxx < W;
    // weight set of phase dx as three 16-byte pieces of every thread (later phases): 1280 uint4 = five (dy) blocks of 256
    auto load_w = [&](int dx, uint4& p0, uint4& p1, uint4& p2) {
        const int i0 = tid, i1 = tid + 512 < 5 * NBLK ? tid + 512 : 0, i2 = (tid & 255) + 1024;
        p0 = gw[(size_t)((i0 / NBLK) * 5 + dx) * TAPQ + wpos(i0 % NBLK)];
        p1 = gw[(size_t)((i1 / NBLK) * 5 + dx) * TAPQ + wpos(i1 % NBLK)];
        if (COT == 2) p2 = gw[(size_t)((i2 / NBLK) * 5 + dx) * TAPQ + wpos(i2 % NBLK)];
    };
    auto store_w = [&](int buf, const uint4& p0, const uint4& p1, const uint4& p2) {
        uint4* dst = reinterpret_cast<uint4*>(Wt + buf * DX_WPH);
        dst[tid] = p0;
        if (tid + 512 < 5 * NBLK) dst[tid + 512] = p1;
        if (COT == 2 && tid < 256) dst[tid + 1024] = p2;
    };
    auto row_of = [&](int s_) { const int gr = G0 - 2 + s_; return gr < 0 ? 0 : (gr >= nrows ? nrows - 1 : gr); };    // scalar clamp
    float4 hv[8], hh;                                            // item n: row s_base + (n >> 1), pixel p + 32 (n & 1); the early role has six
    const float4* src = gx + ((size_t)(x0 + p)) * 8 + c4;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int s_ = s_base + ((n >> 1) < n_rows ? (n >> 1) : n_rows - 1);          // (early role, n = 6, 7: a repeat of row 2, dropped)
        hv[n] = src[(size_t)row_of(s_) * W * 8 + (n & 1) * 32 * 8];
    }
    hh = gx[((size_t)row_of(s_base + (hrow < n_rows ? hrow : n_rows - 1)) * W + (hok ? hxx : 0)) * 8 + c4];
    uint4 wB0, wB1, wB2, wC0, wC1, wC2, wD0, wD1, wD2;          // the weight sets of phases 2, 3, 4 in flight (named: arrays end up in scratch)
    wB0 = wB1 = wB2 = wC0 = wC1 = wC2 = wD0 = wD1 = wD2 = make_uint4(0u, 0u, 0u, 0u);
    if (RW) {                                                    // one-row workgroups: all of it is wanted within ~1.5 us -- requested here, behind the role's own
        load_w(2, wB0, wB1, wB2);
        load_w(3, wC0, wC1, wC2);
        load_w(4, wD0, wD1, wD2);
    }
    __builtin_amdgcn_sched_barrier(0);
    DX_STAMP(1);
    float sa, sai;
    amax_scale_of(am, sa, sai);
    const float out_scale = sai * winv;
    DX_STAMP(2);
    auto put = [&](unsigned char* q, const float4& v) __attribute__((always_inline)) {
        unsigned p00, p10, p01, p11;
        split2h(v.x, v.y, sa, p00, p10);
        split2h(v.z, v.w, sa, p01, p11);
        *reinterpret_cast<uint2*>(q) = make_uint2(p00, p01);
        *reinterpret_cast<uint2*>(q + DX_PLANE) = make_uint2(p10, p11);
    };
    auto stage_role = [&]() __attribute__((always_inline)) {
        unsigned char* q0 = ring + s_base * DX_SLOT;
        const int hcA = p + 2, hcB = p + 34;
        unsigned char* qA = q0 + hcA * 64 + ((((c4 >> 1) ^ swzb(hcA)) << 4) | ((c4 & 1) << 3));
        unsigned char* qB = q0 + hcB * 64 + ((((c4 >> 1) ^ swzb(hcB)) << 4) | ((c4 & 1) << 3));
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if ((n >> 1) < n_rows) put(((n & 1) ? qB : qA) + (n >> 1) * DX_SLOT, hv[n]);       // (uniform: the role's row count)
        if (!hok) hh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hrow < n_rows) put(q0 + hrow * DX_SLOT + hc_h * 64 + ((((c4 >> 1) ^ swzb(hc_h)) << 4) | ((c4 & 1) << 3)), hh);
        uint4* dst = reinterpret_cast<uint4*>(Wt + (late ? 1 : 0) * DX_WPH) + t;
        dst[0] = wq0; dst[256] = wq1;
        if (COT == 2) { dst[512] = wq2; dst[768] = wq3; dst[1024] = wq4; }
        else if (t < 128) dst[512] = wq2;
    };
    auto stage_late = [&]() __attribute__((always_inline)) { if (late) stage_role(); };
    if (!late) stage_role();
    DX_STAMP(12);
    DX_BARRIER();
    DX_STAMP(3);
    float4 biasv = make_float4(0.f, 0.f, 0.f, 0.f);           // output channels cot * 16 + 4 g .. + 3 (epilogue): requested behind the barrier,
    const bool mma = COT == 2 || !late;                         // wave uniform: this wave owns an output tile
    if (a.bias && mma) {                                         // off the critical path (four dword loads: a caller's bias slice need not be 16-byte aligned)
        const float* bp = a.bias + cot * 16 + 4 * g;
        if (COT == 2 || SPLIT) biasv = make_float4(bp[0], bp[1], bp[2], bp[3]);
        else biasv = make_float4(4 * g < a.CO ? bp[0] : 0.f, 4 * g + 1 < a.CO ? bp[1] : 0.f, 4 * g + 2 < a.CO ? bp[2] : 0.f, 4 * g + 3 < a.CO ? bp[3] : 0.f);
    }

    f32x4 acc[R], acl[R], acm[R];                               // acl (+ acm, DX_ACL2): the 2^-11 weighted cross terms
#pragma unroll
    for (int j = 0; j < R; ++j) { acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acl[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acm[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // epilogue operands (residual, activation reference): requested at the start of the last phase, consumed after the last MFMA
    float4 resv[R], actv[R];
#pragma unroll
    for (int j = 0; j < R; ++j) { resv[j] = make_float4(0.f, 0.f, 0.f, 0.f); actv[j] = make_float4(1.f, 1.f, 1.f, 1.f); }
    // correction mode (thin form): the faces' velocities and ground-truth values of every row, requested during phases 1..3 like the
    // residual above -- the ground-truth frames are read once per training step (HBM-cold) and sat, with the velocity read-modify-write
    // behind them, in the launch's tail (9.8 us per launch against 5.0 us for the same layer with a plain store)
    float cpv[R][6];                                            // v_y, v_x, gt_y, gt_x, and the two loss-only faces (or a repeat)
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
        for (int q = 0; q < 6; ++q) cpv[j][q] = 0.f;
    const int pcc = seg * 16 + li;                              // this lane's pixel inside the 64-pixel row
    const int co_l = wt * 16 + li;                              // this lane's weight row (output channel) of the A operand
    auto out_f4 = [&](int j) { return ((size_t)(G0 + j) * W + x0 + pcc) * 8 + cot * 4 + g; };   // float4 index of (row j, this pixel, co 4g..)
    const unsigned char* const wlane = Wt + co_l * 64 + ((g ^ swzb(co_l)) << 4);
    auto a_base = [&](int dx) { const int hc = pcc + dx; return ring + hc * 64 + ((g ^ swzb(hc)) << 4); };

    // Operand registers: TWO weight sets (the set of phase dx + 1 is read during steps 4..6 of phase dx, behind the phase's only
    // barrier) and two pixel operands (one step of look-ahead).  The stream of MFMAs therefore runs across the phase boundary
    // without a bubble; the one barrier of a phase sits in its middle (after step 2, where the next phase's weights are
    // written), when every wave is deep inside a run of MFMAs.
    uint4 bw[2][5][2], ao[2][2];
    auto read_b = [&](int set, int dx, int dy) __attribute__((always_inline)) {
        bw[set][dy][0] = *reinterpret_cast<const uint4*>(wlane + (RW ? dx : (dx & 1)) * DX_WPH + dy * 2 * DX_WPL);
        bw[set][dy][1] = *reinterpret_cast<const uint4*>(wlane + (RW ? dx : (dx & 1)) * DX_WPH + dy * 2 * DX_WPL + DX_WPL);
    };
    auto read_a = [&](int slot, int dx, int s_) __attribute__((always_inline)) {
        const unsigned char* ab = a_base(dx);
        ao[slot][0] = *reinterpret_cast<const uint4*>(ab + s_ * DX_SLOT);
        ao[slot][1] = *reinterpret_cast<const uint4*>(ab + s_ * DX_SLOT + DX_PLANE);
    };
    if (mma) {
        read_b(0, 0, 0);
        read_a(0, 0, 0);
#pragma unroll
        for (int dy = 1; dy < 5; ++dy) read_b(0, 0, dy);
    }

    // ---- epilogue of ONE output row: lane (li, g) holds output channels cot*16 + 4g .. + 3 of pixel pcc -----------------------------
    // (Tried: row j's epilogue behind the MFMAs of step j + 5 of the last phase, in their shadow -- row j has its last MFMA in step
    // j + 4.  The phase grew by what the tail shrank: 9.42 -> 9.81 us per launch in the pipeline probe.  All rows at the end.)
    float vmax = 0.f;
    auto epilogue_row = [&](const int j) __attribute__((always_inline)) {
        if (G0 + j < nrows) {                                   // wave uniform
            float4 v;
            v.x = (acc[j][0] + acl[j][0] * (1.f / 2048.f)) * out_scale + biasv.x;
            v.y = (acc[j][1] + acl[j][1] * (1.f / 2048.f)) * out_scale + biasv.y;
            v.z = (acc[j][2] + acl[j][2] * (1.f / 2048.f)) * out_scale + biasv.z;
            v.w = (acc[j][3] + acl[j][3] * (1.f / 2048.f)) * out_scale + biasv.w;
            if (a.res) { v.x += resv[j].x; v.y += resv[j].y; v.z += resv[j].z; v.w += resv[j].w; }     // (uniform branch; without a residual the prefetched dummy is dropped)
            if (a.epi == SOL_EPI_LRELU) {
                v.x = v.x > 0.f ? v.x : a.slope * v.x; v.y = v.y > 0.f ? v.y : a.slope * v.y;
                v.z = v.z > 0.f ? v.z : a.slope * v.z; v.w = v.w > 0.f ? v.w : a.slope * v.w;
            } else if (a.epi == SOL_EPI_DLRELU) {
                v.x *= actv[j].x > 0.f ? 1.f : a.slope; v.y *= actv[j].y > 0.f ? 1.f : a.slope;
                v.z *= actv[j].z > 0.f ? 1.f : a.slope; v.w *= actv[j].w > 0.f ? 1.f : a.slope;
            }
            vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            st_wt(reinterpret_cast<float4*>(a.y) + out_f4(j), v);      // write-through (common.hpp): no dirty lines left for the end-of-kernel release
        }
    };

    // ---- thin layers (COT == 1): lane (li, g) holds output channels 4g .. 4g+3 of pixel pcc --------------------------------------
    // correction mode (a.cvy: the 32 -> 2 output layer of the trainer, lanes g == 0): the output is applied to the staggered velocity and
    // never stored; the faces without a correction (v_y row H, v_x column W) only enter the loss: the owner of the image's last row / of
    // column W - 1 adds them.  Otherwise a strided store of the a.CO <= 16 channels.
    float lsum = 0.f;
    auto epilogue_thin = [&](const int j) __attribute__((always_inline)) {
        const int gy = G0 + j;
        if (gy >= nrows) return;                                // wave uniform
        float o4[4];
        o4[0] = (acc[j][0] + acl[j][0] * (1.f / 2048.f)) * out_scale + biasv.x;
        o4[1] = (acc[j][1] + acl[j][1] * (1.f / 2048.f)) * out_scale + biasv.y;
        o4[2] = (acc[j][2] + acl[j][2] * (1.f / 2048.f)) * out_scale + biasv.z;
        o4[3] = (acc[j][3] + acl[j][3] * (1.f / 2048.f)) * out_scale + biasv.w;
        const int i = x0 + pcc;
        if (a.cvy) {
            if (g == 0) {
                const int b = a.RPW >= 0 ? (gy >> a.RPW) : gy / H;
                const int jj = gy - b * H;
                const int nVy = a.ctr ? (W + 1) * H : (H + 1) * W, nVx = a.ctr ? W * (H + 1) : H * (W + 1);
                const CorrFaces f = corr_faces(a.ctr, H, W, jj, i);
                float* vy = a.cvy + (size_t)b * nVy;
                float* vx = a.cvx + (size_t)b * nVx;
                const float v0 = cpv[j][0] + a.cs0 * o4[0], v1 = cpv[j][1] + a.cs1 * o4[1];       // (operands prefetched during the tap loop)
                vy[f.oy] = v0;
                vx[f.ox] = v1;
                if (a.gty) {
                    const float d = (cpv[j][2] - v0) / a.ls0;
                    lsum += 0.5f * d * d;
                    if (f.ey >= 0) { const float d2 = cpv[j][4] / a.ls0; lsum += 0.5f * d2 * d2; }       // v_y row Y of the solver grid
                }
                if (a.gtx) {
                    const float d = (cpv[j][3] - v1) / a.ls1;
                    lsum += 0.5f * d * d;
                    if (f.ex >= 0) { const float d2 = cpv[j][5] / a.ls1; lsum += 0.5f * d2 * d2; }       // v_x column X
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 4 * g + r;
                if (co < a.CO) {
                    const size_t o = ((size_t)gy * W + i) * a.CO + co;
                    float v = o4[r];
                    if (a.res) v += a.res[o];
                    if (a.epi == SOL_EPI_LRELU) v = v > 0.f ? v : a.slope * v;
                    else if (a.epi == SOL_EPI_DLRELU) v *= (a.act[o] > 0.f ? 1.f : a.slope);
                    vmax = fmaxf(vmax, fabsf(v));
                    a.y[o] = v;
                }
            }
        }
    };

    // The weight sets of phases 2, 3, 4 are ALL requested during phase 0 (three named register sets, one request group at the phase's
    // start and one each behind steps 3 and DX_NS - 2: spread out, a burst of requests stalls every wave at issue) so that the
    // HBM-cold epilogue operands can follow early: vmcnt retires in order, and a weight set requested AFTER them could not be waited
    // for without them.
    auto phase = [&](const int dx) __attribute__((always_inline)) {
        const int set = dx & 1;
        if (dx == 0 && !RW) load_w(2, wB0, wB1, wB2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < DX_NS; ++s) {
            // operands of the next step: A(s + 1) of this phase, or A(0) of the next one; behind the barrier also the next weight set
            const int aslot = (dx * DX_NS + s) & 1;             // (7 steps per phase: the slot parity runs on across phases)
            const bool late_a = dx == 0 && s == 2;              // row 3 is only staged behind this step (first phase)
            if (mma) {
                if (s + 1 < DX_NS && !late_a) read_a(aslot ^ 1, dx, s + 1);
                else if (dx < 4) read_a(aslot ^ 1, dx + 1, 0);
                if (dx < 4) {                                   // the next weight set, over the steps behind the barrier
#pragma unroll
                    for (int k = 0; k < 5; ++k)
                        if (s == (RW && dx >= 1 ? k : (DX_NS == 7 ? 4 + k / 2 : (DX_NS == 6 ? 3 + k / 2 : 3 + k / 3)))) read_b(set ^ 1, dx + 1, k);
                }
            }
            if (dx == 0 && s == 3 && !RW) load_w(3, wC0, wC1, wC2);
            if (dx == 0 && s == DX_NS - 2 && !RW) load_w(4, wD0, wD1, wD2);
            // Epilogue operands (residual: phase 1, activation reference: phase 2), ONE request per step 3.. -- behind the LAST weight
            // request and spread out: a lane's 16 bytes sit 128 bytes from its neighbour's, so each request occupies the CU's
            // vector-memory path for 16 cache lines, and six of them back to back from all eight waves stalled every wave at issue
            // (+0.8 us in that phase).  UNCONDITIONAL requests -- a launch without a residual / activation reference reads the same
            // positions of x and drops the values: requests inside `if (a.res)` blocks make the compiler's next vmcnt wait a wait
            // for everything in flight.  They are HBM-cold in the training pipeline (written hundreds of launches ago): ~3 us.
            if ((COT == 2 || SPLIT) && (dx == 1 || dx == 2) && s >= 3 && s - 3 < R) {
                const int j = s - 3;
                // (scalar clamp) rows beyond the tensor -- the last workgroup's tail and the padding workgroups that own no rows at
                // all (grid rounded up to a multiple of 8) -- read the tensor's last row: any valid position
                const size_t o4 = ((size_t)min(G0 + j, nrows - 1) * W + x0 + pcc) * 8 + cot * 4 + g;
                if (dx == 1) resv[j] = reinterpret_cast<const float4*>(a.res ? a.res : a.x)[o4];
                else actv[j] = reinterpret_cast<const float4*>(a.epi == SOL_EPI_DLRELU ? a.act : a.x)[o4];
            }
            if (COT == 1 && !SPLIT && dx >= 1 && dx <= 3 && s >= DX_NS - R) {      // (one request group per step, as above)
                const int j = s - (DX_NS - R);
                const int gy = min(G0 + j, nrows - 1);
                const int b = a.RPW >= 0 ? (gy >> a.RPW) : gy / H;
                const CorrFaces f = corr_faces(a.ctr, H, W, gy - b * H, x0 + pcc);
                const size_t nVy = a.ctr ? (size_t)(W + 1) * H : (size_t)(H + 1) * W, nVx = a.ctr ? (size_t)W * (H + 1) : (size_t)H * (W + 1);
                const bool cm = a.cvy != nullptr;                                  // (uniform) not in correction mode: dummy reads of x
                if (dx == 1) {
                    cpv[j][0] = (cm ? a.cvy + b * nVy : a.x)[cm ? f.oy : 0];
                    cpv[j][1] = (cm ? a.cvx + b * nVx : a.x)[cm ? f.ox : 0];
                } else if (dx == 2) {
                    const bool gm = cm && a.gty != nullptr;
                    cpv[j][2] = (gm ? a.gty + b * nVy : a.x)[gm ? f.oy : 0];
                    cpv[j][3] = (gm ? a.gtx + b * nVx : a.x)[gm ? f.ox : 0];
                } else {
                    const bool gm = cm && a.gty != nullptr;
                    const int ey = f.ey >= 0 ? f.ey : f.oy, ex = f.ex >= 0 ? f.ex : f.ox;
                    cpv[j][4] = (gm ? a.gty + b * nVy : a.x)[gm ? ey : 0] - (gm ? a.cvy + b * nVy : a.x)[gm ? ey : 0];      // gt - v of the loss-only faces
                    cpv[j][5] = (gm ? a.gtx + b * nVx : a.x)[gm ? ex : 0] - (gm ? a.cvx + b * nVx : a.x)[gm ? ex : 0];
                }
            }
            __builtin_amdgcn_sched_barrier(0);                  // keep the prefetch reads above this step's MFMAs
            const f16x8 x1 = __builtin_bit_cast(f16x8, ao[aslot][0]), x2 = __builtin_bit_cast(f16x8, ao[aslot][1]);
            const int jlo = s - 4 > 0 ? s - 4 : 0, jhi = s < R - 1 ? s : R - 1;      // (constants once the s loop is unrolled)
            unsigned need = 0u;
#pragma unroll
            for (int j = jlo; j <= jhi; ++j) need |= 1u << (j * 8 + s);
            if (!mma) {
                // (COT == 1: a staging-only wave)
            } else if ((vmask & need) == need) {
                // the common case (every pair of this step valid): one straight block, the three rows' chains interleaved
#pragma unroll
                for (int j = jlo; j <= jhi; ++j)
                    acl[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bw[set][s - j][0]), x2, acl[j], 0, 0, 0);
#pragma unroll
                for (int j = jlo; j <= jhi; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bw[set][s - j][0]), x1, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = jlo; j <= jhi; ++j) {
                    if (DX_ACL2) acm[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bw[set][s - j][1]), x1, acm[j], 0, 0, 0);
                    else acl[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bw[set][s - j][1]), x1, acl[j], 0, 0, 0);
                }
            } else {
                // border / image-straddling workgroups: the pairs whose input row lies in another image are skipped
#pragma unroll
                for (int j = jlo; j <= jhi; ++j)
                    if (vmask & (1u << (j * 8 + s))) {
                        const f16x8 b1 = __builtin_bit_cast(f16x8, bw[set][s - j][0]), b2 = __builtin_bit_cast(f16x8, bw[set][s - j][1]);
                        acl[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, x2, acl[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, x1, acc[j], 0, 0, 0);
                        if (DX_ACL2) acm[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2, x1, acm[j], 0, 0, 0);
                        else acl[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2, x1, acl[j], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s == 2 && dx < (RW ? 1 : 4)) {
                // next phase's weights: their buffer was last read (as the set of phase dx - 1) before the previous phase's barrier;
                // in the first phase also the input rows 3..6 (steps 3..6 read them)
                if (dx == 0) {
                    stage_late();                                 // rows 3..6 AND the weight set of phase 1 (late role)
                    if (RW) { store_w(2, wB0, wB1, wB2); store_w(3, wC0, wC1, wC2); store_w(4, wD0, wD1, wD2); }      // (the last barrier of the launch)
                }
                else if (dx == 1) store_w(0, wB0, wB1, wB2);
                else if (dx == 2) store_w(1, wC0, wC1, wC2);
                else store_w(0, wD0, wD1, wD2);
                DX_BARRIER();
                if (late_a && mma) read_a(aslot ^ 1, dx, s + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        DX_STAMP(4 + dx);
    };
    phase(0);
    phase(1);
    phase(2);
    phase(3);
    phase(4);

    // The wave that is through with its MFMAs issues its epilogue AHEAD of the other wave's MFMA stream on the same SIMD (the ~60 VALU
    // instructions per row otherwise wait behind it, 10-20 clocks apiece).  Same-box A/B, four alternations: 12.49 -> 12.45 ms per step.
    __builtin_amdgcn_s_setprio(3);
    if (DX_ACL2) {
#pragma unroll
        for (int j = 0; j < R; ++j) acl[j] += acm[j];
    }
    if (COT == 2 || SPLIT) {
        if (mma) {
#pragma unroll
            for (int j = 0; j < R; ++j) epilogue_row(j);
        }
    } else {
        if (mma) {
#pragma unroll
            for (int j = 0; j < R; ++j) epilogue_thin(j);
        }
        // workgroup uniform: wave sums -> LDS slots -> the last wave's fixed-order sum -> ONE exact integer add per workgroup (loss_add_exact: bit reproducible)
        if (a.cvy && a.closs) loss_publish_last(lsum, a.closs, amax_lds + 2);
    }
    DX_STAMP(9);
    if (a.ymax) amax_publish_last(vmax, a.ymax, amax_lds);
    DX_STAMP(10);
#ifdef SOL_CONV_PROF
    __builtin_amdgcn_s_waitcnt(0);
    DX_STAMP(11);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_dx_prof_ctl[0], 1u);
#endif
}

}  // namespace

#ifdef SOL_CONV_PROF
extern "C" int sol_conv_dx_prof_set(long long* buf, unsigned cap) {
    const unsigned ctl[2] = {0u, cap ? cap : 1u};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_dx_prof_ctl), ctl, sizeof(ctl)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_dx_prof), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

// three rows per workgroup where that fills the chip; one row where the launch is small (< 128 three-row workgroups)
static int dx_rows_per_wg(int nrows, int tiles_x) { return ((nrows + 2) / 3) * tiles_x >= 128 ? 3 : 1; }

bool sol_conv_dx_usable(const ConvArgs& a, int NT, int ntiles) {
    if (!sol_opt().conv_dx || !a.xmax || !a.wsh || a.W % 64 != 0) return false;
    if (NT == 2) return (sol_opt().conv_dx & 1) && a.CO == 32 && !a.cvy;      // option bit 0 (one-row launches: as two half-channel workgroups per tile, bit 3)
    // Thin layers (<= 16 output channels; option bit 1, bit 2 = also in big launches): half the waves of the COT = 1 form only stage, and
    // where the launch fills the chip with three-row workgroups k_conv5x5_sb<1, 2> (twelve waves, a row each) is the faster kernel
    // (SOL-32 step at 128x64, B = 6: 13.26 ms against 13.43); in the small launches (one row per workgroup: the 64x32 recipe, roll-outs)
    // the dx form wins (8.94 against 9.04 ms per recipe step; tools/conv_thin_ab.py).
    if (NT != 1 || a.CO > 16 || !(sol_opt().conv_dx & 2)) return false;
    return (sol_opt().conv_dx & 4) != 0 || dx_rows_per_wg(ntiles / a.tiles_x, a.tiles_x) == 1;
}

int sol_conv_dx_launch(hipStream_t s, const ConvArgs& a_in, int ntiles) {
    ConvArgs a = a_in;
    a.RPW = -1;                                           // (field unused by this kernel otherwise) log2 H, or -1
    for (int k = 0; k < 20; ++k) if ((1 << k) == a.H) a.RPW = k;
    static std::atomic<unsigned long long> optin{0};
    if (int e = sol_lds_optin(optin, {SOL_K((k_conv5x5_dx<1, 2>)), SOL_K((k_conv5x5_dx<3, 2>)), SOL_K((k_conv5x5_dx<1, 1>)), SOL_K((k_conv5x5_dx<3, 1>)),
                                         SOL_K((k_conv5x5_dx<1, 1, true>))}, "k_conv5x5_dx")) return e;
    const bool thin = a.CO <= 16;
    const int nrows = ntiles / a.tiles_x;                 // global image rows B*H
    const int R = dx_rows_per_wg(nrows, a.tiles_x);
    int grid = ((nrows + R - 1) / R) * a.tiles_x;
    if (grid > 64) grid = (grid + 7) / 8 * 8;             // XCD-aware tile order needs a multiple of 8 (xcd_tile); padding tiles own no rows
    if (!thin && R == 1 && (sol_opt().conv_dx & 8))
        SOL_LAUNCH((k_conv5x5_dx<1, 1, true>), dim3(2 * grid), dim3(512), dx_lds(1, 1), s, a.x, a.wsh, a.xmax, nrows, a.H, a.W, a.tiles_x, a.RPW, a);
    else if (thin) {
        if (R == 3) SOL_LAUNCH((k_conv5x5_dx<3, 1>), dim3(grid), dim3(512), dx_lds(3, 1), s, a.x, a.wsh, a.xmax, nrows, a.H, a.W, a.tiles_x, a.RPW, a);
        else SOL_LAUNCH((k_conv5x5_dx<1, 1>), dim3(grid), dim3(512), dx_lds(1, 1), s, a.x, a.wsh, a.xmax, nrows, a.H, a.W, a.tiles_x, a.RPW, a);
    } else if (R == 3) SOL_LAUNCH((k_conv5x5_dx<3, 2>), dim3(grid), dim3(512), dx_lds(3, 2), s, a.x, a.wsh, a.xmax, nrows, a.H, a.W, a.tiles_x, a.RPW, a);
    else SOL_LAUNCH((k_conv5x5_dx<1, 2>), dim3(grid), dim3(512), dx_lds(1, 2), s, a.x, a.wsh, a.xmax, nrows, a.H, a.W, a.tiles_x, a.RPW, a);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}
