// 5x5 SAME convolution, 32 input channels, on the gfx950 bf16 matrix cores with fp32-EQUIVALENT
// arithmetic ("split bf16", the bf16 analogue of 3xTF32): every fp32 operand v is written as the
// exact-to-2^-27 sum of three bf16 numbers v = v1 + v2 + v3 (v1 = rn(v), v2 = rn(v - v1),
// v3 = rn(v - v1 - v2); the subtractions are exact in fp32) and a product a*b is accumulated as the
// six bf16 x bf16 MFMA products whose weight is >= 2^-18:  a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1.
// bf16 x bf16 products are exact in fp32 and the MFMA accumulates in fp32, so the result differs from
// an fp32 FMA chain by O(2^-24) per term -- the parity tests hold it to the same tolerances as the
// fp32-MFMA kernels of conv5x5.hip (which remain for the other shapes and as SOL_CONV_NO_SB=1).
//
// Why: v_mfma_f32_16x16x4_f32 sustains 92 TFLOP/s on random operands on this part, while
// v_mfma_f32_16x16x32_bf16 sustains 1580 TFLOP/s (tools/ubench/mfma_bf16_rand.hip): six bf16
// products per fp32 product are still 2.8x the fp32 matrix-core rate.
//
// Replaces the same keras.layers.Conv2D(32, 5, padding='same') (+bias, LeakyReLU, residual add) of
// model_mars_moon (/root/reference/karman-2d/karman_train.py:101-138) as conv5x5.hip does.
//
// Work decomposition: three CONSECUTIVE 64-pixel image rows per workgroup (12 waves, one workgroup per CU
// at 128x64 x 6).  They need seven input rows; each is split into its three bf16 planes and staged ONCE
// into a shared 4-slot LDS ring (one new row per tap row instead of three), the weights arrive pre-split from the
// packed buffer ([dy][dx][plane][cout][cin] bf16, already in LDS image order) and are double
// buffered per tap row.  One tap = ONE K = 32 MFMA per split product: lane (li, g) holds
// A[pixel li][cin 8g..8g+7] and B[cin 8g..8g+7][cout li] as one 16-byte ds_read_b128 each.
// LDS rows are 64 B per pixel / per cout; the 16-B chunk index is XOR-ed with ((idx >> 2) & 1) << 1,
// which makes every ds_read_b128 lane group of the CDNA4 LDS (MI355X_MICROARCH.md, LDS table) hit 16
// distinct bank quads for all five dx shifts (brute-forced, tools/lds_swizzle_search.py).
#include "split_kernels.hpp"
#include <stdlib.h>

namespace {

using namespace sbk;

__global__ void k_absmax(const float* __restrict__ x, size_t n, unsigned* __restrict__ slots) {
    __shared__ float red[16];
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
    amax_publish(m, slots, red);
}

// fp16 weight planes: header {2^shift_w, 2^-shift_w, 0, 0} then out[dy][dx][plane 2][o][chunk s][j]
__global__ void k_pack_sh(const float* __restrict__ w, const unsigned* __restrict__ wmax, float* __restrict__ hdr,
                          unsigned short* __restrict__ out, int cin, int cout, int OP, int mode) {
    float sc, inv;
    amax_scale(wmax, sc, inv);
    if (blockIdx.x == 0 && threadIdx.x == 0) { hdr[0] = sc; hdr[1] = inv; hdr[2] = 0.f; hdr[3] = 0.f; }
    const int total = 25 * OP * 16;   // pairs of channels
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int jp = e & 3, s = (e >> 2) & 3, o = (e >> 4) % OP, tap = e / (16 * OP);
        const int i0 = 8 * (s ^ swzb(o)) + 2 * jp;
        float v[2] = {0.f, 0.f};
        for (int q = 0; q < 2; ++q) {
            const int i = i0 + q;
            if (i < cin && o < cout)
                v[q] = mode == SOL_CONV_FWD ? w[(tap * cin + i) * cout + o] : w[((24 - tap) * cout + o) * cin + i];
        }
        unsigned p[2];
        split2h(v[0], v[1], sc, p[0], p[1]);
        for (int pl = 0; pl < 2; ++pl) {
            const size_t idx = ((((size_t)tap * 2 + pl) * OP + o) * 4 + s) * 8 + 2 * jp;
            *reinterpret_cast<unsigned*>(out + idx) = p[pl];
        }
    }
}

// ------------------------------------------------------------------------------------
// weight packing: out[dy][dx][plane][o][chunk s][j] (bf16), chunk s holds cin 8*(s ^ swzb(o)) + j
// FWD / BWD_DATA source index as k_pack (conv5x5.hip); `cin` = 32 channels of the convolution being run
// ------------------------------------------------------------------------------------
__global__ void k_pack_sb(const float* __restrict__ w, unsigned short* __restrict__ out, int cin, int cout, int OP, int mode) {
    const int total = 25 * OP * 16;   // pairs of channels
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int jp = e & 3, s = (e >> 2) & 3, o = (e >> 4) % OP, tap = e / (16 * OP);
        const int i0 = 8 * (s ^ swzb(o)) + 2 * jp;
        float v[2] = {0.f, 0.f};
        for (int q = 0; q < 2; ++q) {
            const int i = i0 + q;
            if (i < cin && o < cout)
                v[q] = mode == SOL_CONV_FWD ? w[(tap * cin + i) * cout + o] : w[((24 - tap) * cout + o) * cin + i];
        }
        unsigned p[3];
        split3(v[0], v[1], p[0], p[1], p[2]);
        for (int pl = 0; pl < 3; ++pl) {
            const size_t idx = ((((size_t)tap * 3 + pl) * OP + o) * 4 + s) * 8 + 2 * jp;
            *reinterpret_cast<unsigned*>(out + idx) = p[pl];
        }
    }
}

// ------------------------------------------------------------------------------------
// forward / backward-data kernel, W % 64 == 0, CIN = 32
// ------------------------------------------------------------------------------------
// KIND 0: bf16, six products (default without absmax);  1: bf16, three leading products (experiment);
// KIND 2: fp16, three products, per-tensor power-of-two scaling (needs a.xmax)
// -DSOL_CONV_PROF (tools/conv_phase_probe.py builds such a library next to the product one): phase stamps (100 MHz
// s_memrealtime, thread 0 of every workgroup, 16 per workgroup) into the buffer set with sol_conv_prof_set().
// -DSOL_CONV_TRUNC=0|1|2 (tools/conv_variants.py): return at kernel entry / after the prologue / after the tap-row loop.
#ifdef SOL_CONV_PROF
__device__ long long* g_conv_prof = nullptr;        // [cap launches][grid][16] stamps; g_conv_prof_ctl = {launch counter, cap}
__device__ unsigned g_conv_prof_ctl[2] = {0u, 1u};
extern "C" int sol_conv_prof_set(long long* buf, unsigned cap) {
    const unsigned ctl[2] = {0u, cap ? cap : 1u};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_conv_prof_ctl), ctl, sizeof(ctl)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_conv_prof), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define SOL_CSTAMP(k) do { if (NT == 2 && KIND == 2 && threadIdx.x == 0 && cv_prof) cv_prof[(k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SOL_CSTAMP(k) do { } while (0)
#endif
#ifndef SOL_CONV_TRUNC
#define SOL_CONV_TRUNC 9
#endif

template <int NT, int KIND>
__global__ void __launch_bounds__(768) k_conv5x5_sb(ConvArgs a, int nrows) {
    constexpr int OP = NT * 16;
    constexpr int HWP = 68;                       // halo pixels per row (64 + 4)
    constexpr int PLANE = HWP * 64;               // bytes per bf16 plane of one halo row
    constexpr int NPL = KIND == 2 ? 2 : 3;        // operand planes
    constexpr int SLOT = NPL * PLANE;             // bytes per halo row
    constexpr int WPL = OP * 64;                  // bytes per (dx, plane) weight block
    constexpr int WBUF = 5 * NPL * WPL;           // bytes per tap-row weight phase
    constexpr int AMAX_LDS = 4 * SLOT + 2 * WBUF;  // behind the ring and the weight buffers: workgroup max|y|, wave counter; loss_publish_last's 4 + 12 words
    extern __shared__ __align__(16) unsigned char smem_sb[];
    if (SOL_CONV_TRUNC == 0) return;
#ifdef SOL_CONV_PROF
    long long* __restrict__ cv_prof = g_conv_prof;            // read once into scalar registers (a reload per stamp costs ~0.5 us each)
    if (cv_prof) cv_prof += ((size_t)(g_conv_prof_ctl[0] % g_conv_prof_ctl[1]) * gridDim.x + blockIdx.x) * 16;
#endif
    SOL_CSTAMP(0);
    if (threadIdx.x == 0) *reinterpret_cast<uint4*>(smem_sb + AMAX_LDS) = make_uint4(0u, 0u, 0u, 0u);      // see amax_publish_last, loss_publish_last
    // the wave index is read into an SGPR: everything derived from it (tile row, image bounds, "does this wave have taps in
    // this tap row") is then provably wave uniform -- scalar branches instead of exec-masked regions that the compiler
    // executes back to back with conservative waits at the joins
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wid >> 2, t = tid & 255, lane = tid & 63, wave = wid & 3;
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H, W = a.W;
    // workgroup = three CONSECUTIVE global image rows G0 .. G0+2 (row = b*H + y) of one 64-pixel column block:
    // they need the seven input rows G0-2 .. G0+4, each of which is staged ONCE into a shared 4-slot ring
    // (row r lives in slot (r - G0 + 2) & 3); a tile skips the tap rows whose input row belongs to another image.
    const int bx = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = bx % a.tiles_x, G0 = (bx / a.tiles_x) * 3;
    const int gy = G0 + grp;                          // this tile's global row
    const bool tvalid = gy < nrows;
    const int b = (tvalid ? gy : 0) / H, x0 = tx * 64;
    const int row_lo = b * H, row_hi = row_lo + H;    // rows of this tile's image
    unsigned char* ring = smem_sb;                    // [4][3 planes][68][64 B], shared by the three tiles
    unsigned char* Wt = smem_sb + 4 * SLOT;           // [2][5][3 planes][OP][64 B], shared
    const float4* gx = reinterpret_cast<const float4*>(a.x);
    const uint4* gw = KIND == 2 ? reinterpret_cast<const uint4*>(a.wsh) + 1 : reinterpret_cast<const uint4*>(a.wsb);
    float sa = 1.f, out_scale = 1.f;                  // KIND 2: input scale 2^shift and 2^-(shift_x + shift_w)
    SOL_CSTAMP(10);
    constexpr int WV = WBUF / 16;                     // uint4 per weight phase
    constexpr int WPT = (WV + 767) / 768;

    // one float4 (4 channels of one halo pixel) of global row `gr` per thread (threads 0..543)
    // The loads are UNCONDITIONAL (lanes without a pixel read the tensor's first 16 bytes and are zeroed when the row is
    // written to LDS): a predicated load sits in its own basic block, and the register allocator's copies at the join made
    // the compiler wait for it right there -- a full memory round trip in the prologue, four in a row in the unrolled loop.
    auto row_ok = [&](int gr, int e) {
        const int xx = x0 + (e >> 3) - 2;
        return e < HWP * 8 && gr >= 0 && gr < nrows && xx >= 0 && xx < W;
    };
    auto load_row = [&](int gr, int e) {
        const int xx = x0 + (e >> 3) - 2;
        return gx[row_ok(gr, e) ? ((size_t)gr * W + xx) * 8 + (e & 7) : (size_t)0];
    };
    auto store_row = [&](int slot, int gr, float4 v, int e) {
        if (!row_ok(gr, e)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < HWP * 8) {
            const int hc = e >> 3, c4 = e & 7;
            unsigned p[3][2];
            if constexpr (KIND == 2) {
                split2h(v.x, v.y, sa, p[0][0], p[1][0]);
                split2h(v.z, v.w, sa, p[0][1], p[1][1]);
            } else {
                split3(v.x, v.y, p[0][0], p[1][0], p[2][0]);
                split3(v.z, v.w, p[0][1], p[1][1], p[2][1]);
            }
            unsigned char* q = ring + slot * SLOT + hc * 64 + ((((c4 >> 1) ^ swzb(hc)) << 4) | ((c4 & 1) << 3));
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint2*>(q + pl * PLANE) = make_uint2(p[pl][0], p[pl][1]);
        }
    };
    static_assert(WPT <= 3, "weight phase: at most three 16-byte pieces per thread");
    // one thread's share of a tap-row weight set travels as up to three NAMED uint4 (arrays and structs passed by reference
    // through the lambdas ended up in scratch memory); clamped index instead of a predicate: see load_row
    auto load_w = [&](int dy, uint4& p0, uint4& p1, uint4& p2) {
        p0 = gw[(size_t)dy * WV + (tid < WV ? tid : WV - 1)];
        if constexpr (WPT >= 2) p1 = gw[(size_t)dy * WV + (tid + 768 < WV ? tid + 768 : WV - 1)];
        if constexpr (WPT >= 3) p2 = gw[(size_t)dy * WV + (tid + 1536 < WV ? tid + 1536 : WV - 1)];
    };
    auto store_w = [&](int buf, const uint4& p0, const uint4& p1, const uint4& p2) {
        uint4* dst = reinterpret_cast<uint4*>(Wt + buf * WBUF);
        if (tid < WV) dst[tid] = p0;
        if constexpr (WPT >= 2) { if (tid + 768 < WV) dst[tid + 768] = p1; }
        if constexpr (WPT >= 3) { if (tid + 1536 < WV) dst[tid + 1536] = p2; }
    };

    // Two tap rows of look-ahead: the input row and the weight set of tap row dy+2 are requested at the start of tap row dy
    // (two named register sets in flight), the row/weights of tap row dy+1 are written to LDS BETWEEN the taps of row dy (their
    // ring slot / weight buffer is free for the whole tap row), and the barriers of the loop wait for LDS traffic only
    // (__syncthreads() would also wait for the loads in flight).
#define SB_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    float biasv[NT];                                  // bias of output channel n*16 + li (0 without a bias / beyond CO)
    float4 hvA = make_float4(0.f, 0.f, 0.f, 0.f), hvB = hvA;   // the two register sets in flight (named, not an array: an
    uint4 wA0, wA1, wA2, wB0, wB1, wB2;                         // indexed pair was left in scratch memory by the compiler)
    wA0 = wA1 = wA2 = wB0 = wB1 = wB2 = make_uint4(0u, 0u, 0u, 0u);
    {   // prologue: input rows G0-2, G0-1, G0 (one per tile group, 3 float4 per thread) and the weights of tap row 0
        float4 hv[3];
        uint4 w0 = make_uint4(0u, 0u, 0u, 0u), w1 = w0, w2 = w0;
        // every global load of the prologue goes out before anything is waited for: the absmax slots of x (KIND 2) travel
        // together with the rows and the weights instead of costing a round trip of their own
        uint4 am = make_uint4(0u, 0u, 0u, 0u);
        float winv = 1.f;
        if constexpr (KIND == 2) { am = amax_load(a.xmax); winv = reinterpret_cast<const float*>(a.wsh)[1]; }
        {   // the biases travel with the prologue's requests too (they were two serial round trips in the epilogue); without a
            // bias the lanes read x[0] and the value is dropped
            const float* bp = a.bias ? a.bias : a.x;
#pragma unroll
            for (int n = 0; n < NT; ++n) biasv[n] = bp[a.bias ? (n * 16 + li < a.CO ? n * 16 + li : 0) : 0];
        }
#pragma unroll
        for (int n = 0; n < 3; ++n) hv[n] = load_row(G0 - 2 + grp, t + n * 256);
        load_w(0, w0, w1, w2);
        hvA = load_row(G0 + 1, tid);
        load_w(1, wA0, wA1, wA2);
        __builtin_amdgcn_sched_barrier(0);             // (the scheduler sank a weight load to its use otherwise)
        if constexpr (KIND == 2) {
            float sai;
            amax_scale_of(am, sa, sai);
            out_scale = sai * winv;
        }
#pragma unroll
        for (int n = 0; n < 3; ++n) store_row(grp, G0 - 2 + grp, hv[n], t + n * 256);
        store_w(0, w0, w1, w2);
    }
    SB_BARRIER();
    SOL_CSTAMP(1);
    if (SOL_CONV_TRUNC == 1) return;

    f32x4 acc[NT], acl[NT];                           // acl: KIND 2 accumulator of the 2^-11 weighted cross terms
#pragma unroll
    for (int n = 0; n < NT; ++n) { acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; acl[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int pcc = wave * 16 + li;                   // this lane's A-row pixel inside the tile
    // split products in order of increasing weight (small terms first)
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};

    // the epilogue's residual / activation-reference operands (NT == 2 path) are fetched during the last tap rows: in the
    // training pipeline they come from HBM (the forward activations are hundreds of launches old), a 2 us round trip that
    // otherwise sits between the last MFMA and the first store
    constexpr int EF4 = 16 * OP / 4 / 64;             // float4 per lane of the wave's [16 px][OP] tile
    // They travel by LDS-DMA into a per-wave 2 x EF4 KB region ([res | act][n][lane] float4: every lane reads back exactly what
    // it requested), not into registers: with register destinations the register allocator copied a freshly requested value
    // (`global_load; s_waitcnt vmcnt(0); v_mov`) -- a memory round trip in the open, once per launch.  The request is inline
    // assembly: after the BUILTIN the compiler makes the next ds_read wait for the request (it cannot tell that the operand
    // reads do not alias the DMA target), i.e. the first operand read of tap row 4 waited for the round trip.  Requests the
    // compiler does not know about only make its own vmcnt waits longer, never shorter (the counter retires in order); the
    // read-back in the epilogue is ordered by an explicit s_waitcnt vmcnt(0).
    __shared__ __align__(16) unsigned char pf_lds[12 * 2 * EF4 * 1024];
    unsigned char* pf = pf_lds + wid * (2 * EF4 * 1024);
    auto lds_dma16 = [&](const void* src, unsigned char* dst_wave_uniform) __attribute__((always_inline)) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dst_wave_uniform);
        unsigned keep;                                    // M0 is saved and restored inside the block (it may not be clobbered)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lo));
    };
#ifndef SOL_CONV_LATE_EPI                             // (A/B switch: 14.85 -> 14.77 ms per training step)
    constexpr bool EPI_PREFETCH = true;
#else
    constexpr bool EPI_PREFETCH = false;
#endif

#ifndef SOL_CONV_STAGE_AT
#define SOL_CONV_STAGE_AT 1, 3
#endif
    constexpr int STAGE_AT[2] = {SOL_CONV_STAGE_AT};
    constexpr int STAGE_ROW_AT = STAGE_AT[0], STAGE_W_AT = STAGE_AT[1];   // after which tap (dx) the row / the weights are written
    // one tap row: requests (row, weights) of tap row `nxt` into (hin, win) -- nxt < 0: nothing --, runs the five taps, writes
    // (hout, wout) = row G0+dy+1 and the weights of tap row dy+1 to LDS
    auto tap_row = [&](const int dy, const int nxt, float4& hin, uint4& wi0, uint4& wi1, uint4& wi2,
                       const float4& hout, const uint4& wo0, const uint4& wo1, const uint4& wo2)
                       __attribute__((always_inline)) {
        // requested AFTER the last staging store of tap row 3: the compiler's wait counts are conservative across the
        // staging branches, and a request before them was waited for (a round trip in the open) when the weights were written
        auto epi_prefetch = [&]() __attribute__((always_inline)) {
            if (EPI_PREFETCH && tvalid && a.CO == OP) {
#pragma unroll
                for (int n = 0; n < EF4; ++n) {
                    const int e = lane + n * 64, px = e / (OP / 4), c4 = e % (OP / 4);
                    const size_t o4 = ((size_t)gy * W + x0 + wave * 16 + px) * (OP / 4) + c4;
                    if (a.res) lds_dma16(reinterpret_cast<const float4*>(a.res) + o4, pf + n * 1024);
                    if (a.epi == SOL_EPI_DLRELU) lds_dma16(reinterpret_cast<const float4*>(a.act) + o4, pf + (EF4 + n) * 1024);
                }
            }
        };
        if (nxt >= 0) {
            hin = load_row(G0 + nxt, tid);            // the one new input row of tap row nxt: G0-2 + nxt + 2
            load_w(nxt, wi0, wi1, wi2);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int src = gy + dy - 2;                  // input row of this tile for this tap row
        // The staging / prefetch code stands ONCE, outside the "has taps" conditionals (a wave of a border tile skips the taps
        // of tap rows that lie in another image): with a second copy in an else branch the compiler merged the two paths'
        // outstanding-load state conservatively and waited (vmcnt(0)) at the start of the next tap row.
        const bool has_taps = tvalid && src >= row_lo && src < row_hi;   // wave uniform (SGPR)
        const unsigned char* hrow = ring + ((grp + dy) & 3) * SLOT;
        const unsigned char* wbuf = Wt + (dy & 1) * WBUF;
        uint4 ao[2][NPL], bo[2][NT][NPL];
        {
            auto load_ops = [&](int dx, uint4 (&ar)[NPL], uint4 (&br)[NT][NPL]) {
                const int hc = pcc + dx;
                const unsigned char* ap = hrow + hc * 64 + ((g ^ swzb(hc)) << 4);
#ifdef SOL_CONV_EXP_NOLDS                             // experiment (tools/conv_variants.py): operands from registers, no ds_read
                unsigned fake = 0x2c112e37u + (unsigned)(size_t)ap * 0x00010001u + dx;
                asm volatile("" : "+v"(fake));
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) ar[pl] = make_uint4(fake, fake ^ 0x01000100u, fake + 0x00030002u, fake ^ 0x00100010u);
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) br[n][pl] = make_uint4(fake ^ 0x02000200u, fake, fake ^ 0x00010100u, fake + 0x00010001u);
                return;
#endif
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) ar[pl] = *reinterpret_cast<const uint4*>(ap + pl * PLANE);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int co = n * 16 + li;
                    const unsigned char* bp = wbuf + dx * NPL * WPL + co * 64 + ((g ^ swzb(co)) << 4);
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) br[n][pl] = *reinterpret_cast<const uint4*>(bp + pl * WPL);
                }
            };
            auto taps = [&](const int dx0, const int dx1) __attribute__((always_inline)) {
            if (dx0 == 0) load_ops(0, ao[0], bo[0]);
#pragma unroll
            for (int dx = dx0; dx < dx1; ++dx) {
                if (dx < 4) load_ops(dx + 1, ao[(dx + 1) & 1], bo[(dx + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ds_reads above this tap's MFMAs
#ifdef SOL_CONV_EXP_NOMFMA                            // experiment: the ds_reads are consumed, no MFMA issued
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    { const uint4 q = ao[dx & 1][pl]; asm volatile("" :: "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w)); }
#pragma unroll
                    for (int n = 0; n < NT; ++n) { const uint4 q = bo[dx & 1][n][pl]; asm volatile("" :: "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w)); }
                }
                continue;
#endif
                if constexpr (KIND == 2) {
                    const f16x8 a1 = __builtin_bit_cast(f16x8, ao[dx & 1][0]), a2 = __builtin_bit_cast(f16x8, ao[dx & 1][NPL - 1]);
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const f16x8 b1 = __builtin_bit_cast(f16x8, bo[dx & 1][n][0]), b2 = __builtin_bit_cast(f16x8, bo[dx & 1][n][NPL - 1]);
                        acl[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, acl[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[n], 0, 0, 0);
                        acl[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, acl[n], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int pr = (KIND == 1 ? 3 : 0); pr < 6; ++pr) {
                        const bf16x8 av = __builtin_bit_cast(bf16x8, ao[dx & 1][PA[pr]]);
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const bf16x8 bv = __builtin_bit_cast(bf16x8, bo[dx & 1][n][PB[pr]]);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[n], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            };
            // staging of the next tap row in the shadow of this row's MFMAs (slot of row G0+dy+1: its previous tenant,
            // row G0+dy-3, is dead; weight buffer (dy+1)&1: set dy-1 is dead)
            static_assert(STAGE_ROW_AT <= STAGE_W_AT && STAGE_W_AT <= 4, "staging positions");
            if (has_taps) taps(0, STAGE_ROW_AT + 1);
            if (dy < 4) { store_row((dy + 3) & 3, G0 + dy + 1, hout, tid); __builtin_amdgcn_sched_barrier(0); }
            if (has_taps) taps(STAGE_ROW_AT + 1, STAGE_W_AT + 1);
            if (dy < 4) { store_w((dy + 1) & 1, wo0, wo1, wo2); __builtin_amdgcn_sched_barrier(0); }
            if (dy == 3) { epi_prefetch(); __builtin_amdgcn_sched_barrier(0); }
            if (has_taps) taps(STAGE_W_AT + 1, 5);
        }
        SB_BARRIER();
        SOL_CSTAMP(2 + dy);
    };
    {
        tap_row(0, 2, hvB, wB0, wB1, wB2, hvA, wA0, wA1, wA2);
        tap_row(1, 3, hvA, wA0, wA1, wA2, hvB, wB0, wB1, wB2);
        tap_row(2, 4, hvB, wB0, wB1, wB2, hvA, wA0, wA1, wA2);
        tap_row(3, -1, hvA, wA0, wA1, wA2, hvB, wB0, wB1, wB2);
        tap_row(4, -1, hvA, wA0, wA1, wA2, hvB, wB0, wB1, wB2);
    }
    if (SOL_CONV_TRUNC == 2) { if (acc[0][0] + acc[NT - 1][3] == 1.2345f) a.y[tid] = acl[0][1]; return; }
    if constexpr (KIND == 2) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[n][r] = (acc[n][r] + acl[n][r] * (1.f / 2048.f)) * out_scale;
    }
    unsigned char* halo = ring + (size_t)grp * 4 * 16 * OP * sizeof(float);   // epilogue scratch: 4 waves x [16 px][OP] floats per tile
    float vmax = 0.f;                                 // max|y| of this thread (for a.ymax)
    // ---- epilogue: transpose the wave's [16 px][OP] tile through LDS (the ring is free after the last
    //      barrier) so that every lane moves 16-byte pieces of full 128-byte pixels ----
    if (a.CO == OP) {
        float* tb = reinterpret_cast<float*>(halo) + wave * (16 * OP);   // 16 px x OP floats per wave
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float bias = a.bias ? biasv[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) tb[(4 * g + r) * OP + n * 16 + li] = acc[n][r] + bias;
        }
        // same-wave LDS round trip: the compiler's s_waitcnt lgkmcnt orders write -> read
        if (EPI_PREFETCH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA of res / act has landed
        if (tvalid) {
            constexpr int F4 = 16 * OP / 4 / 64;          // float4 per lane
#pragma unroll
            for (int n = 0; n < F4; ++n) {
                const int e = lane + n * 64;              // float4 index inside the tile
                const int px = e / (OP / 4), c4 = e % (OP / 4);
                float4 v = *reinterpret_cast<const float4*>(&tb[px * OP + c4 * 4]);
                const size_t o4 = ((size_t)gy * W + x0 + wave * 16 + px) * (OP / 4) + c4;
                if (a.res) { const float4 q = EPI_PREFETCH ? *reinterpret_cast<const float4*>(pf + n * 1024 + lane * 16) : reinterpret_cast<const float4*>(a.res)[o4]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
                if (a.epi == SOL_EPI_LRELU) {
                    v.x = v.x > 0.f ? v.x : a.slope * v.x; v.y = v.y > 0.f ? v.y : a.slope * v.y;
                    v.z = v.z > 0.f ? v.z : a.slope * v.z; v.w = v.w > 0.f ? v.w : a.slope * v.w;
                } else if (a.epi == SOL_EPI_DLRELU) {
                    const float4 q = EPI_PREFETCH ? *reinterpret_cast<const float4*>(pf + (EF4 + n) * 1024 + lane * 16) : reinterpret_cast<const float4*>(a.act)[o4];
                    v.x *= q.x > 0.f ? 1.f : a.slope; v.y *= q.y > 0.f ? 1.f : a.slope;
                    v.z *= q.z > 0.f ? 1.f : a.slope; v.w *= q.w > 0.f ? 1.f : a.slope;
                }
                vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                st_wt(reinterpret_cast<float4*>(a.y) + o4, v);
            }
        }
    } else if (a.cvy) {
        // correction mode (32 -> 2 layer of the trainer): lanes li = 0 / 1 hold output channel 0 / 1 of pixels 4g..4g+3 of the
        // wave's 16-pixel segment; the output is applied to the staggered velocity and never stored.  The faces without a
        // correction (v_y row H, v_x column W) only enter the loss: the owner of the image's last row / of column W-1 adds them.
        float lsum = 0.f;
        if (tvalid && li < 2) {
            const int jj = gy - b * H;
            const int nV = li == 0 ? (a.ctr ? (W + 1) * H : (H + 1) * W) : (a.ctr ? W * (H + 1) : H * (W + 1));      // faces per simulation
            const float bias = a.bias ? biasv[0] : 0.f;
            const float s = li == 0 ? a.cs0 : a.cs1, ls = li == 0 ? a.ls0 : a.ls1;
            float* vf = (li == 0 ? a.cvy : a.cvx) + (size_t)b * nV;
            const float* gt = li == 0 ? a.gty : a.gtx;
            if (gt) gt += (size_t)b * nV;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const CorrFaces f = corr_faces(a.ctr, H, W, jj, x0 + wave * 16 + 4 * g + r);
                const int o = li == 0 ? f.oy : f.ox, e = li == 0 ? f.ey : f.ex;
                const float v = vf[o] + s * (acc[0][r] + bias);
                vf[o] = v;
                if (gt) { const float d = (gt[o] - v) / ls; lsum += 0.5f * d * d; }
                if (gt && e >= 0) { const float d = (gt[e] - vf[e]) / ls; lsum += 0.5f * d * d; }        // the face without a correction (v_y row Y / v_x column X)
            }
        }
        // workgroup uniform: wave sums -> LDS slots -> the last wave's fixed-order sum -> ONE exact integer add per workgroup (loss_add_exact: bit reproducible)
        if (a.closs) loss_publish_last(lsum, a.closs, reinterpret_cast<unsigned*>(smem_sb + AMAX_LDS) + 2);
    } else if (tvalid) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = n * 16 + li;
            if (co >= a.CO) continue;
            const float bias = a.bias ? biasv[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cc = wave * 16 + 4 * g + r;
                const size_t o = ((size_t)gy * W + x0 + cc) * a.CO + co;
                float v = acc[n][r] + bias;
                if (a.res) v += a.res[o];
                if (a.epi == SOL_EPI_LRELU) v = v > 0.f ? v : a.slope * v;
                else if (a.epi == SOL_EPI_DLRELU) v *= (a.act[o] > 0.f ? 1.f : a.slope);
                vmax = fmaxf(vmax, fabsf(v));
                a.y[o] = v;
            }
        }
    }
    SOL_CSTAMP(7);
    // no barrier: the last wave publishes (two barriers + a serial 12-value reduction cost 0.5 us per launch)
    if (a.ymax) amax_publish_last(vmax, a.ymax, reinterpret_cast<unsigned*>(smem_sb + AMAX_LDS));
    SOL_CSTAMP(8);
#ifdef SOL_CONV_PROF
    __builtin_amdgcn_s_waitcnt(0);
    SOL_CSTAMP(9);
    if (NT == 2 && KIND == 2 && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_conv_prof_ctl[0], 1u);
#endif
}


// weight-gradient kernel: body in split_kernels.hpp (bww_sb_body), also fused into the solver-adjoint launch
template <int KIND>
__global__ void __launch_bounds__(512) k_conv5x5_bww_sb(BwArgs a) {
    extern __shared__ __align__(16) unsigned char smem_sb[];
    bww_sb_body<KIND>(a, blockIdx.x, smem_sb);
}
// n <= 5 weight-gradient jobs in ONE launch (karman-3d: the five depth slices of a Conv3D layer, which were five launches of one round of
// workgroups each -- every one with its own dispatch gap, cold start and tail): workgroup u runs block u % wg_per of job u / wg_per
template <int KIND>
__global__ void __launch_bounds__(512) k_conv5x5_bww_sb_jobs(BwJobs p) {
    extern __shared__ __align__(16) unsigned char smem_sb[];
    const int job = (int)blockIdx.x / p.wg_per, sub = (int)blockIdx.x % p.wg_per;
    if (sub >= p.nrun[job]) return;
    bww_sb_body<KIND>(p.a[job], sub, smem_sb);
}

#ifdef BWW_PROF
extern "C" int sol_bww_prof_set(unsigned* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(sbk::g_bww_prof), &buf, sizeof(buf)) == hipSuccess ? 0 : -1; }
#endif

// ------------------------------------------------------------------------------------
// all layers, all operand forms, one launch (training / roll-out path)
// ------------------------------------------------------------------------------------
// Job = (layer, mode).  Every workgroup of a job first finds the layer's max|w| itself (<= 25,600 values), then packs
// its share of the three sections of the job's buffer: fp32 [tap][o][i], bf16 planes, fp16 header + planes
// (layouts as k_pack / k_pack_sb / k_pack_sh).  Replaces ~130 tiny launches per training step.
struct PackJob { const float* w; float* out; float* bias_out; const float* bias_in; int cin, cout, mode; int tt; };   // cin/cout of the convolution being RUN; tt: taps transposed (dy <-> dx)
struct PackJobs { PackJob j[24]; int n; };
constexpr int PACK_WG = 8;        // workgroups per job

__device__ __forceinline__ float pack_src(const PackJob& jb, int tap, int i, int o) {
    if (i >= jb.cin || o >= jb.cout) return 0.f;
    if (jb.tt) tap = (tap % 5) * 5 + tap / 5;        // the convolution runs on TRANSPOSED images (train.hip): conv(x^T, w^T) = conv(x, w)^T
    return jb.mode == SOL_CONV_FWD ? jb.w[(tap * jb.cin + i) * jb.cout + o] : jb.w[((24 - tap) * jb.cout + o) * jb.cin + i];
}

__global__ void __launch_bounds__(256) k_pack_jobs(PackJobs jobs) {
    __shared__ float red[8];
    const PackJob jb = jobs.j[blockIdx.x / PACK_WG];
    const int part = blockIdx.x % PACK_WG, tid = threadIdx.x;
    const int IP = jb.cin <= 4 ? 4 : 32, OP = jb.cout <= 16 ? 16 : 32;
    // fp32 section
    const int total = 25 * OP * IP;
    for (int e = part * 256 + tid; e < total; e += PACK_WG * 256) {
        const int i = e % IP, o = (e / IP) % OP, tap = e / (IP * OP);
        jb.out[e] = pack_src(jb, tap, i, o);
    }
    if (part == 0 && jb.bias_out && tid < 32) jb.bias_out[tid] = (jb.bias_in && tid < jb.cout) ? jb.bias_in[tid] : 0.f;
    if (IP != 32) return;
    // max |w| of the layer (every workgroup of the job computes the same value)
    float m = 0.f;
    for (int e = tid; e < 25 * jb.cin * jb.cout; e += 256) m = fmaxf(m, fabsf(jb.w[e]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const unsigned mb = __float_as_uint(m);
    int ex = (int)(mb >> 23) - 127;
    ex = mb == 0u ? 0 : min(max(ex, -100), 100);
    const float sc = __uint_as_float((unsigned)(14 - ex + 127) << 23), inv = __uint_as_float((unsigned)(ex - 14 + 127) << 23);
    unsigned short* sb = reinterpret_cast<unsigned short*>(jb.out + total);
    float* hdr = jb.out + total + (size_t)25 * 3 * OP * 16;
    unsigned short* sh = reinterpret_cast<unsigned short*>(hdr + 4);
    if (part == 0 && tid == 0) { hdr[0] = sc; hdr[1] = inv; hdr[2] = 0.f; hdr[3] = 0.f; }
    const int pairs = 25 * OP * 16;
    for (int e = part * 256 + tid; e < pairs; e += PACK_WG * 256) {
        const int jp = e & 3, s = (e >> 2) & 3, o = (e >> 4) % OP, tap = e / (16 * OP);
        const int i0 = 8 * (s ^ swzb(o)) + 2 * jp;
        const float v0 = pack_src(jb, tap, i0, o), v1 = pack_src(jb, tap, i0 + 1, o);
        unsigned p[3], h[2];
        split3(v0, v1, p[0], p[1], p[2]);
        split2h(v0, v1, sc, h[0], h[1]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<unsigned*>(sb + ((((size_t)tap * 3 + pl) * OP + o) * 4 + s) * 8 + 2 * jp) = p[pl];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<unsigned*>(sh + ((((size_t)tap * 2 + pl) * OP + o) * 4 + s) * 8 + 2 * jp) = h[pl];
    }
}

// dynamic part: ring + two weight buffers (three planes: the bf16 kinds) + absmax words (the kernels add 24 x OP/16 KB static)
constexpr size_t sb_lds(int OP) { return (size_t)4 * 3 * 68 * 64 + 2 * (size_t)5 * 3 * OP * 64 + 16 + 64; }

int init_sb_kernels() {
    // static LDS (the conv kernels' epilogue-prefetch regions) counts against the same 160 KB
    static std::atomic<unsigned long long> optin{0};
    return sol_lds_optin(optin, {SOL_K(k_conv5x5_bww_sb<0>), SOL_K(k_conv5x5_bww_sb<2>), SOL_K(k_conv5x5_bww_sb_jobs<0>), SOL_K(k_conv5x5_bww_sb_jobs<2>), SOL_K(k_conv5x5_sb<1, 0>), SOL_K(k_conv5x5_sb<2, 0>),
                                 SOL_K(k_conv5x5_sb<1, 1>), SOL_K(k_conv5x5_sb<2, 1>), SOL_K(k_conv5x5_sb<1, 2>), SOL_K(k_conv5x5_sb<2, 2>)},
                         "split conv kernels", true);
}

}  // namespace

// max|x| of a tensor into the SOL_ABSMAX_SLOTS-slot form the scaled kernels consume (sol_conv5x5_scaled, sol_conv3d, the weight
// gradients): for callers whose producer did not publish it.  One pass over x at HBM speed.
namespace {
__global__ void k_zero_slots(unsigned* slots) { slots[threadIdx.x] = 0u; }
}
extern "C" int sol_absmax(void* stream, const float* x, int64_t n, uint32_t* slots) {
    SOL_REQUIRE(x && slots && n > 0, "sol_absmax: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    SOL_LAUNCH(k_zero_slots, dim3(1), dim3(SOL_AMAX_SLOTS), 0, s, slots);
    const size_t blocks = ((size_t)n + 4095) / 4096;
    SOL_LAUNCH(k_absmax, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks))), dim3(256), 0, s, x, (size_t)n, slots);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

size_t sol_conv_sb_packed_floats(int OP) { return (size_t)25 * 3 * OP * 16; }   // 25 taps x 3 planes x OP x 32 bf16

int sol_conv_sb_pack(hipStream_t s, const float* w_hwio, int cin, int cout, int mode, void* out) {
    const int OP = cout <= 16 ? 16 : 32;
    const int total = 25 * OP * 16;
    SOL_LAUNCH(k_pack_sb, dim3((total + 255) / 256), dim3(256), 0, s, w_hwio, reinterpret_cast<unsigned short*>(out), cin, cout, OP, mode);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

// fp16 section: header (4 floats) + 25 taps x 2 planes x OP x 32 fp16, + SOL_AMAX_SLOTS absmax slots of the weights (scratch)
size_t sol_conv_sh_packed_floats(int OP) { return 4 + (size_t)25 * 2 * OP * 16 + SOL_AMAX_SLOTS; }

int sol_conv_sh_pack(hipStream_t s, const float* w_hwio, int cin, int cout, int mode, void* out) {
    const int OP = cout <= 16 ? 16 : 32;
    float* hdr = reinterpret_cast<float*>(out);
    unsigned* slots = reinterpret_cast<unsigned*>(hdr + 4 + (size_t)25 * 2 * OP * 16);
    SOL_LAUNCH(k_zero_slots, dim3(1), dim3(SOL_AMAX_SLOTS), 0, s, slots);       // (a kernel, not hipMemsetAsync: memset nodes are refused in captured graphs, sol_graph_check)
    SOL_LAUNCH(k_absmax, dim3(8), dim3(256), 0, s, w_hwio, (size_t)25 * cin * cout, slots);
    SOL_LAUNCH_CHECK();
    const int total = 25 * OP * 16;
    SOL_LAUNCH(k_pack_sh, dim3((total + 255) / 256), dim3(256), 0, s, w_hwio, slots, hdr, reinterpret_cast<unsigned short*>(hdr + 4), cin, cout, OP, mode);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

// SOL_CONV_SPLIT=3 runs the three leading products only (~2^-17 relative error per product): an
// experiment knob, NOT the default and not what bench.py or the parity tests use.
int sol_conv_sb_launch(hipStream_t s, const ConvArgs& a, int NT, int ntiles) {
    if (sol_conv_thin32_usable(a, NT)) return sol_conv_thin32_launch(s, a, ntiles);      // thin layers of 64-pixel images: exact fp32 on the VALU
    if (sol_conv_dx_usable(a, NT, ntiles)) return sol_conv_dx_launch(s, a, ntiles);
    if (int e = init_sb_kernels()) return e;
    const int nprod = sol_opt().conv_split3 ? 3 : 6;
    const int nrows = ntiles / a.tiles_x;             // global image rows B*H
    int grid3 = ((nrows + 2) / 3) * a.tiles_x;        // three consecutive rows of one column block per workgroup
    if (grid3 > 64) grid3 = (grid3 + 7) / 8 * 8;      // XCD-aware tile order needs a multiple of 8 (xcd_tile); padding tiles own no rows
    const size_t lds = sb_lds(NT * 16);
    if (a.xmax) {                                     // per-tensor absmax known: fp16 three-product kernels
        if (NT == 2) SOL_LAUNCH((k_conv5x5_sb<2, 2>), dim3(grid3), dim3(768), lds, s, a, nrows);
        else SOL_LAUNCH((k_conv5x5_sb<1, 2>), dim3(grid3), dim3(768), lds, s, a, nrows);
    }
    else if (NT == 2 && nprod == 6) SOL_LAUNCH((k_conv5x5_sb<2, 0>), dim3(grid3), dim3(768), lds, s, a, nrows);
    else if (NT == 2) SOL_LAUNCH((k_conv5x5_sb<2, 1>), dim3(grid3), dim3(768), lds, s, a, nrows);
    else if (nprod == 6) SOL_LAUNCH((k_conv5x5_sb<1, 0>), dim3(grid3), dim3(768), lds, s, a, nrows);
    else SOL_LAUNCH((k_conv5x5_sb<1, 1>), dim3(grid3), dim3(768), lds, s, a, nrows);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

int sol_bww_sb_launch(hipStream_t s, const BwArgs& a, int nblk_run) {
    if (int e = init_sb_kernels()) return e;
    const bool use_sh = sol_opt().conv_precision == 0;
    // fp16 three-product kernel: absmax of both operands known and no workgroup straddles two segments
    if (use_sh && a.xmax && a.zmax && (a.B * a.H) % a.rb == 0) SOL_LAUNCH(k_conv5x5_bww_sb<2>, dim3(nblk_run), dim3(512), BW_LDS, s, a);
    else SOL_LAUNCH(k_conv5x5_bww_sb<0>, dim3(nblk_run), dim3(512), BW_LDS, s, a);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

int sol_bww_sb_jobs_launch(hipStream_t s, const BwJobs& p) {
    if (int e = init_sb_kernels()) return e;
    SOL_REQUIRE(p.n >= 1 && p.n <= 5 && p.wg_per >= 1, "sol_bww_sb_jobs_launch: 1..5 jobs");
    bool sh = sol_opt().conv_precision == 0;
    // (a workgroup must not straddle two SEGMENTS -- their scales differ; with one segment a ragged last block is fine)
    for (int k = 0; k < p.n; ++k) sh = sh && p.a[k].xmax && p.a[k].zmax && (p.a[k].nseg == 1 || (p.a[k].B * p.a[k].H) % p.a[k].rb == 0);
    if (sh) SOL_LAUNCH(k_conv5x5_bww_sb_jobs<2>, dim3(p.n * p.wg_per), dim3(512), BW_LDS, s, p);
    else SOL_LAUNCH(k_conv5x5_bww_sb_jobs<0>, dim3(p.n * p.wg_per), dim3(512), BW_LDS, s, p);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

// internal (train.hip): pack `n` (layer, mode) jobs in one launch; out buffers sized by sol_conv5x5_packed_floats
int sol_pack_jobs(hipStream_t s, int n, const float* const* w, float* const* out, float* const* bias_out, const float* const* bias_in,
                  const int* cin, const int* cout, const int* mode, int taps_transposed) {
    SOL_REQUIRE(n >= 1 && n <= 24, "sol_pack_jobs: 1..24 jobs (got %d)", n);
    PackJobs jobs{};
    jobs.n = n;
    for (int k = 0; k < n; ++k) jobs.j[k] = PackJob{w[k], out[k], bias_out[k], bias_in[k], cin[k], cout[k], mode[k], taps_transposed};
    SOL_LAUNCH(k_pack_jobs, dim3(n * PACK_WG), dim3(256), 0, s, jobs);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}
