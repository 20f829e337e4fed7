// 5 x 5 x 5 SAME convolution, 32 -> 32 channels, NDHWC with W == 64, as ONE launch on the gfx950 16-bit matrix cores with
// fp32-equivalent arithmetic (fp16 three-product operand splits, per-tensor power-of-two scale: KIND 2 of conv5x5_sb.hip).
//
// Replaces keras.layers.Conv3D(32, 5, padding='same') (+ bias, LeakyReLU, residual add) of the 3-D model_mars_moon -- the
// dimension-generic form of /root/reference/karman-2d/karman_train.py:101-138 (the reference has no 3-D code, README.md:37-38).
//
// Why a kernel of its own: the five-pass form (sol_conv3d in karman3d.hip: one launch of the 2-D kernel per depth slice,
// running sum in HBM) pays the 2-D kernel's per-workgroup skeleton (prologue round trip, epilogue, write drain) and a
// read-modify-write of the 64 MB output per slice.  Here a workgroup keeps its three (x, z)-plane rows' accumulators in
// registers across ALL 125 taps: for each depth slice kd it runs the 2-D kernel's five tap rows on plane d + kd - 2 (shared
// 4-slot LDS ring of split input rows, double-buffered tap-row weight sets, one barrier per five taps), and the first rows /
// weight sets of slice kd + 1 are REQUESTED during tap rows 3 and 4 of slice kd, so that the only exposed cost of a slice
// boundary is one LDS write round + barrier.  One prologue round trip and one epilogue per 125 taps instead of per 25.
//
// Work decomposition, operand layout (ds_read_b128 fragments, XOR swizzle), epilogue of the first form: as k_conv5x5_sb<2, 2>.
// Three forms live here (option k3d_conv_rows; DESIGN.md 4.8 has the measurements that led from one to the next):
//   k_conv3d_sb          3 rows per workgroup, 16 px x 32 co per wave, 12 waves     412 us per launch at 128 x 64 x 64
//   k_conv3d_sb6         6 rows, 32 x 32, 12 waves, padded strides, LDS-DMA weights 334 us
//   k_conv3d_sb8<NT>     8 rows, 64 x (16 NT), 8 waves, operand prefetch            279-292 us (default; NT = 1: the thin 32 -> 3 / 4 layers)
#include "split_kernels.hpp"

namespace {

using namespace sbk;

#define C3_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ void __launch_bounds__(768) k_conv3d_sb(ConvArgs a, int nrows, int D) {
    constexpr int NT = 2, OP = 32, HWP = 68;
    constexpr int PLANE = HWP * 64;               // bytes per fp16 plane of one halo row
    constexpr int SLOT = 2 * PLANE;               // hi + lo plane
    constexpr int WPL = OP * 64;                  // bytes per (dx, plane) weight block
    constexpr int WBUF = 5 * 2 * WPL;             // bytes per tap-row weight set (20 480)
    constexpr int AMAX_LDS = 4 * SLOT + 2 * WBUF;
    extern __shared__ __align__(16) unsigned char smem_c3[];
    if (threadIdx.x == 0) *reinterpret_cast<uint2*>(smem_c3 + AMAX_LDS) = make_uint2(0u, 0u);      // see amax_publish_last
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wid >> 2, t = tid & 255, lane = tid & 63, wave = wid & 3;
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H;
    constexpr int W = 64;
    // workgroup = three consecutive global rows G0 .. G0+2 (row = plane * H + x-row, plane = b * D + d) of the [B*D*H][64][32] tensor
    const int bx = xcd_tile(blockIdx.x, gridDim.x);
    const int G0 = bx * 3;
    const int gy = G0 + grp;
    const bool tvalid = gy < nrows;
    const int plane = (tvalid ? gy : 0) / H, dpl = plane % D;
    const int row_lo = plane * H, row_hi = row_lo + H;
    unsigned char* ring = smem_c3;                    // [4][2 planes][68][64 B]
    unsigned char* Wt = smem_c3 + 4 * SLOT;           // [2][5][2 planes][32][64 B]
    const float4* gx = reinterpret_cast<const float4*>(a.x);
    const uint4* gw = reinterpret_cast<const uint4*>(a.wsh) + 1;      // header {2^shift_w, 2^-shift_w, 0, 0} then 25 tap-row sets
    float sa = 1.f, out_scale = 1.f;
    constexpr int WV = WBUF / 16;                     // 1280 uint4 per tap-row set
    static_assert((WV + 767) / 768 == 2, "two 16-byte pieces per thread and weight set");

    // unconditional loads (clamped index, zeroed when written to LDS): see conv5x5_sb.hip
    auto row_ok = [&](int gr, int e) {
        const int xx = (e >> 3) - 2;
        return e < HWP * 8 && gr >= 0 && gr < nrows && xx >= 0 && xx < W;
    };
    auto load_row = [&](int gr, int e) {
        const int xx = (e >> 3) - 2;
        return gx[row_ok(gr, e) ? ((size_t)gr * W + xx) * 8 + (e & 7) : (size_t)0];
    };
    auto store_row = [&](int slot, int gr, float4 v, int e) {
        if (!row_ok(gr, e)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < HWP * 8) {
            const int hc = e >> 3, c4 = e & 7;
            unsigned p0[2], p1[2];
            split2h(v.x, v.y, sa, p0[0], p1[0]);
            split2h(v.z, v.w, sa, p0[1], p1[1]);
            unsigned char* q = ring + slot * SLOT + hc * 64 + ((((c4 >> 1) ^ swzb(hc)) << 4) | ((c4 & 1) << 3));
            *reinterpret_cast<uint2*>(q) = make_uint2(p0[0], p0[1]);
            *reinterpret_cast<uint2*>(q + PLANE) = make_uint2(p1[0], p1[1]);
        }
    };
    auto load_w = [&](int set, uint4& p0, uint4& p1) {
        p0 = gw[(size_t)set * WV + tid];
        p1 = gw[(size_t)set * WV + (tid + 768 < WV ? tid + 768 : WV - 1)];
    };
    auto store_w = [&](int buf, const uint4& p0, const uint4& p1) {
        uint4* dst = reinterpret_cast<uint4*>(Wt + buf * WBUF);
        dst[tid] = p0;
        if (tid + 768 < WV) dst[tid + 768] = p1;
    };

    float biasv[NT];
    float4 hvA = make_float4(0.f, 0.f, 0.f, 0.f), hvB = hvA;      // the two register sets of the tap-row pipeline
    float4 hvP0 = hvA, hvP1 = hvA, hvP2 = hvA;                     // first three rows of the NEXT depth slice
    uint4 wA0, wA1, wB0, wB1, wP0, wP1;
    wA0 = wA1 = wB0 = wB1 = wP0 = wP1 = make_uint4(0u, 0u, 0u, 0u);
    {   // prologue of slice kd = 0: rows G0-2 .. G0 of plane d - 2 (one per tile group) and tap-row weight set 0
        const int sh = -2 * H;
        uint4 am = amax_load(a.xmax);
        const float winv = reinterpret_cast<const float*>(a.wsh)[1];
        {
            const float* bp = a.bias ? a.bias : a.x;
#pragma unroll
            for (int n = 0; n < NT; ++n) biasv[n] = bp[a.bias ? n * 16 + li : 0];
        }
        hvP0 = load_row(G0 - 2 + grp + sh, t);
        hvP1 = load_row(G0 - 2 + grp + sh, t + 256);
        hvP2 = load_row(G0 - 2 + grp + sh, t + 512);
        load_w(0, wP0, wP1);
        hvA = load_row(G0 + 1 + sh, tid);
        load_w(1, wA0, wA1);
        __builtin_amdgcn_sched_barrier(0);
        float sai;
        amax_scale_of(am, sa, sai);
        out_scale = sai * winv;
        store_row(grp, G0 - 2 + grp + sh, hvP0, t);
        store_row(grp, G0 - 2 + grp + sh, hvP1, t + 256);
        store_row(grp, G0 - 2 + grp + sh, hvP2, t + 512);
        store_w(0, wP0, wP1);
    }
    C3_BARRIER();

    f32x4 acc[NT], acl[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) { acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; acl[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int pcc = wave * 16 + li;

    // residual of the epilogue: fetched by LDS-DMA during the last tap rows (conv5x5_sb.hip explains the form)
    constexpr int EF4 = 16 * OP / 4 / 64;             // 2 float4 per lane of the wave's [16 px][32] tile
    __shared__ __align__(16) unsigned char pf_lds[12 * EF4 * 1024];
    unsigned char* pf = pf_lds + wid * (EF4 * 1024);
    auto lds_dma16 = [&](const void* src, unsigned char* dst_wave_uniform) __attribute__((always_inline)) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dst_wave_uniform);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lo));
    };

    // one tap row of depth slice kd.  MODE 0: request row / weights of tap row dy + 2 of this slice into (hin, wi);
    // MODE 1 (dy == 3): request the next slice's first three rows + weight set 0 into (hvP, wP);
    // MODE 2 (dy == 4): request the next slice's row G0+1 + weight set 1 into (hin, wi).  last: no next slice.
    // Between the taps: (hout, wo) = row G0+dy+1 / weight set dy+1 of this slice go to LDS (dy < 4).
    auto tap_row = [&](const int kd, const int dy, const int mode, const bool last, float4& hin, uint4& wi0, uint4& wi1,
                       const float4& hout, const uint4& wo0, const uint4& wo1) __attribute__((always_inline)) {
        const int sh = (kd - 2) * H, shn = (kd - 1) * H;
        if (mode == 0) {
            hin = load_row(G0 + dy + 2 + sh, tid);
            load_w(kd * 5 + dy + 2, wi0, wi1);
        } else if (!last) {
            if (mode == 1) {
                hvP0 = load_row(G0 - 2 + grp + shn, t);
                hvP1 = load_row(G0 - 2 + grp + shn, t + 256);
                hvP2 = load_row(G0 - 2 + grp + shn, t + 512);
                load_w((kd + 1) * 5, wP0, wP1);
            } else {
                hin = load_row(G0 + 1 + shn, tid);
                load_w((kd + 1) * 5 + 1, wi0, wi1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int src = gy + sh + dy - 2;             // input row of this tile for this tap row
        const bool plane_ok = dpl + kd - 2 >= 0 && dpl + kd - 2 < D;
        const bool has_taps = tvalid && plane_ok && src >= row_lo + sh && src < row_hi + sh;      // wave uniform
        const unsigned char* hrow = ring + ((grp + dy) & 3) * SLOT;
        const unsigned char* wbuf = Wt + (dy & 1) * WBUF;
        uint4 ao[2][2], bo[2][NT][2];
        auto load_ops = [&](int dx, uint4 (&ar)[2], uint4 (&br)[NT][2]) {
            const int hc = pcc + dx;
            const unsigned char* ap = hrow + hc * 64 + ((g ^ swzb(hc)) << 4);
            ar[0] = *reinterpret_cast<const uint4*>(ap);
            ar[1] = *reinterpret_cast<const uint4*>(ap + PLANE);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co = n * 16 + li;
                const unsigned char* bp = wbuf + dx * 2 * WPL + co * 64 + ((g ^ swzb(co)) << 4);
                br[n][0] = *reinterpret_cast<const uint4*>(bp);
                br[n][1] = *reinterpret_cast<const uint4*>(bp + WPL);
            }
        };
        auto taps = [&](const int dx0, const int dx1) __attribute__((always_inline)) {
            if (dx0 == 0) load_ops(0, ao[0], bo[0]);
#pragma unroll
            for (int dx = dx0; dx < dx1; ++dx) {
                if (dx < 4) load_ops(dx + 1, ao[(dx + 1) & 1], bo[(dx + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const f16x8 a1 = __builtin_bit_cast(f16x8, ao[dx & 1][0]), a2 = __builtin_bit_cast(f16x8, ao[dx & 1][1]);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const f16x8 b1 = __builtin_bit_cast(f16x8, bo[dx & 1][n][0]), b2 = __builtin_bit_cast(f16x8, bo[dx & 1][n][1]);
                    acl[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, acl[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[n], 0, 0, 0);
                    acl[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, acl[n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (has_taps) taps(0, 2);
        if (dy < 4) { store_row((dy + 3) & 3, G0 + dy + 1 + sh, hout, tid); __builtin_amdgcn_sched_barrier(0); }
        if (has_taps) taps(2, 4);
        if (dy < 4) { store_w((dy + 1) & 1, wo0, wo1); __builtin_amdgcn_sched_barrier(0); }
        if (dy == 3 && last && tvalid && a.res) {     // residual prefetch, after the last staging store of the launch
#pragma unroll
            for (int n = 0; n < EF4; ++n) {
                const int e = lane + n * 64, px = e / (OP / 4), c4 = e % (OP / 4);
                const size_t o4 = ((size_t)gy * W + wave * 16 + px) * (OP / 4) + c4;
                lds_dma16(reinterpret_cast<const float4*>(a.res) + o4, pf + n * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (has_taps) taps(4, 5);
        C3_BARRIER();
    };

    for (int kd = 0; kd < 5; ++kd) {
        const bool last = kd == 4;
        tap_row(kd, 0, 0, last, hvB, wB0, wB1, hvA, wA0, wA1);
        tap_row(kd, 1, 0, last, hvA, wA0, wA1, hvB, wB0, wB1);
        tap_row(kd, 2, 0, last, hvB, wB0, wB1, hvA, wA0, wA1);
        tap_row(kd, 3, 1, last, hvA, wA0, wA1, hvB, wB0, wB1);
        tap_row(kd, 4, 2, last, hvA, wA0, wA1, hvB, wB0, wB1);
        if (!last) {
            // slice boundary: every wave is past the barrier of tap row 4, so the ring slots 0..2 and weight buffer 0 are free;
            // the rows / weights were requested two tap rows ago
            const int shn = (kd - 1) * H;
            store_row(grp, G0 - 2 + grp + shn, hvP0, t);
            store_row(grp, G0 - 2 + grp + shn, hvP1, t + 256);
            store_row(grp, G0 - 2 + grp + shn, hvP2, t + 512);
            store_w(0, wP0, wP1);
            C3_BARRIER();
        }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[n][r] = (acc[n][r] + acl[n][r] * (1.f / 2048.f)) * out_scale;

    // ---- epilogue: transpose the wave's [16 px][32] tile through LDS (the ring is free after the last barrier) ----
    unsigned char* halo = ring + (size_t)grp * 4 * 16 * OP * sizeof(float);
    float vmax = 0.f;
    float* tb = reinterpret_cast<float*>(halo) + wave * (16 * OP);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const float bias = a.bias ? biasv[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[(4 * g + r) * OP + n * 16 + li] = acc[n][r] + bias;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's LDS-DMA of the residual has landed
    if (tvalid) {
#pragma unroll
        for (int n = 0; n < EF4; ++n) {
            const int e = lane + n * 64;
            const int px = e / (OP / 4), c4 = e % (OP / 4);
            float4 v = *reinterpret_cast<const float4*>(&tb[px * OP + c4 * 4]);
            const size_t o4 = ((size_t)gy * W + wave * 16 + px) * (OP / 4) + c4;
            if (a.res) { const float4 q = *reinterpret_cast<const float4*>(pf + n * 1024 + lane * 16); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
            if (a.epi == SOL_EPI_LRELU) {
                v.x = v.x > 0.f ? v.x : a.slope * v.x; v.y = v.y > 0.f ? v.y : a.slope * v.y;
                v.z = v.z > 0.f ? v.z : a.slope * v.z; v.w = v.w > 0.f ? v.w : a.slope * v.w;
            } else if (a.epi == SOL_EPI_DLRELU) {     // backward: times LeakyReLU'(activation reference)
                const float4 q = reinterpret_cast<const float4*>(a.act)[o4];
                v.x *= q.x > 0.f ? 1.f : a.slope; v.y *= q.y > 0.f ? 1.f : a.slope;
                v.z *= q.z > 0.f ? 1.f : a.slope; v.w *= q.w > 0.f ? 1.f : a.slope;
            }
            vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            st_wt(reinterpret_cast<float4*>(a.y) + o4, v);
        }
    }
    if (a.ymax) amax_publish_last(vmax, a.ymax, reinterpret_cast<unsigned*>(smem_c3 + AMAX_LDS));
}

// ------------------------------------------------------------------------------------------------------------------------
// k_conv3d_sb6: the same convolution with SIX rows per workgroup and a 32-pixel x 32-channel tile per wave.
// The 3-row kernel above is bound by its LDS operand stream: per tap a wave issues 6 ds_read_b128 (2 A + 4 B) for 6 MFMAs, and
// all twelve waves re-read the SAME weight fragments.  A wave that owns two 16-pixel tiles reads the weight fragments once for
// both: 8 reads (4 A + 4 B) per 12 MFMAs -- one third less LDS traffic per matrix instruction.  With three rows per workgroup that
// would leave six waves for four SIMDs (why the 2-D kernel cannot do it at 768 rows); the 3-D launches have 8 192 rows per
// simulation, so six rows per workgroup still fill the chip five times over.  12 waves: wave = (row r = wid / 2, pixel half).
// Input rows live in an 8-slot ring (row G0-2+rr in slot rr & 7: six live rows + the incoming one) with a padded 160-byte pixel
// stride (operand addresses = one VGPR + immediates, see the kernel body); the tap-row weight sets arrive by LDS-DMA into three
// rotating buffers; split staging of the rows, next-slice prefetch during tap rows 3 / 4, one LDS-only barrier per five taps and
// the epilogue through LDS are as above.  The residual is read straight from global memory in the epilogue.
// Kept as option k3d_conv_rows = 6 (the default is the eight-row kernel below) and as the subject of tools/c6_phase_probe.py.
// ------------------------------------------------------------------------------------------------------------------------
// -DSOL_C6_PROF (tools/c6_phase_probe.py builds such a library next to the product one): per-wave s_memtime stamps (low 32 bits,
// kept in LDS, dumped at the end) -- 8 per tap row -- into the buffer set with sol_c6_prof_set().
#ifdef SOL_C6_PROF
__device__ unsigned* g_c6_prof = nullptr;
extern "C" int sol_c6_prof_set(unsigned* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_c6_prof), &buf, sizeof(buf)) == hipSuccess ? 0 : -1; }
#define C6_NSTAMP 216
#define C6_STAMP(i) do { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); if (lane == 0) c6_st[(i)] = t_; } while (0)
#else
#define C6_STAMP(i) do { } while (0)
#endif
__global__ void __launch_bounds__(768) k_conv3d_sb6(ConvArgs a, int nrows, int D) {
    constexpr int OP = 32, HWP = 68;
    // Input rows in LDS with a PADDED pixel stride instead of the XOR swizzle of the kernels above: a pixel is 160 bytes = 64 B hi
    // plane (four 16-byte channel chunks) + 64 B lo plane + 32 B pad.  160-byte strides are conflict free for the ds_read_b128
    // lane groups of gfx950, and -- the point -- every A address of a tap row is ONE lane-dependent VGPR plus an instruction
    // immediate ((16 m + dx) * 160 + 64 * plane), where the XOR form needs address arithmetic (or 10 live address registers)
    // per tap: 29 -> 13 VALU instructions per 12 MFMAs.  The weight sets keep the XOR image of k_pack3_sh (its swizzle depends
    // on the lane only, so the B addresses are base + immediate as well) and travel global -> LDS by LDS-DMA: no staging
    // registers, no ds_write, three rotating 20 KB buffers (set t + 2 is requested during tap row t).
    constexpr int PS = 160, SLOT = HWP * PS, WPL = OP * 64, WSET = 5 * 2 * WPL;
    constexpr int NSLOT = 8, ROWS = 6, NWB = 3;
    constexpr int AMAX_LDS = NSLOT * SLOT + NWB * WSET;
    extern __shared__ __align__(16) unsigned char smem_c6[];
    if (threadIdx.x == 0) *reinterpret_cast<uint2*>(smem_c6 + AMAX_LDS) = make_uint2(0u, 0u);
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = wid >> 1, half = wid & 1;           // this wave's row of the workgroup and its 32-pixel half
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H;
    constexpr int W = 64;
    const int bx = xcd_tile(blockIdx.x, gridDim.x);
    const int G0 = bx * ROWS;
    const int gy = G0 + r;
    const bool tvalid = gy < nrows;
    const int plane = (tvalid ? gy : 0) / H, dpl = plane % D;
    const int row_lo = plane * H, row_hi = row_lo + H;
    unsigned char* Wt = smem_c6;                      // weight buffers first: LDS-DMA destinations (M0) below 64 KB
    unsigned char* ring = smem_c6 + NWB * WSET;
    const float4* gx = reinterpret_cast<const float4*>(a.x);
    const unsigned char* gw = reinterpret_cast<const unsigned char*>(a.wsh) + 16;
    float sa = 1.f, out_scale = 1.f;
#ifdef SOL_C6_PROF
    unsigned* c6_st = reinterpret_cast<unsigned*>(smem_c6 + AMAX_LDS + 16) + wid * C6_NSTAMP;
    C6_STAMP(200);
    if (lane == 0) c6_st[204] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif

    auto row_ok = [&](int gr, int e) {
        const int xx = (e >> 3) - 2;
        return e < HWP * 8 && gr >= 0 && gr < nrows && xx >= 0 && xx < W;
    };
    auto load_row = [&](int gr, int e) {
        const int xx = (e >> 3) - 2;
        return gx[row_ok(gr, e) ? ((size_t)gr * W + xx) * 8 + (e & 7) : (size_t)0];
    };
    auto store_row = [&](int slot, int gr, float4 v, int e) {
        if (!row_ok(gr, e)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < HWP * 8) {
            const int hc = e >> 3, c4 = e & 7;
            unsigned p0[2], p1[2];
            split2h(v.x, v.y, sa, p0[0], p1[0]);
            split2h(v.z, v.w, sa, p0[1], p1[1]);
            unsigned char* q = ring + slot * SLOT + hc * PS + c4 * 8;
            *reinterpret_cast<uint2*>(q) = make_uint2(p0[0], p0[1]);
            *reinterpret_cast<uint2*>(q + 64) = make_uint2(p1[0], p1[1]);
        }
    };
    // LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to dst + 16 * lane (conv5x5_sb.hip explains the form)
    auto lds_dma16 = [&](const void* src, unsigned char* dst_wave_uniform) __attribute__((always_inline)) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dst_wave_uniform);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lo));
    };
    // weight set t (one tap row: 20 one-KB pieces) into buffer buf: piece wid by every wave, piece wid + 12 by waves 0..7
    auto dma_w = [&](int t, int buf) __attribute__((always_inline)) {
        const unsigned char* src = gw + (size_t)t * WSET + lane * 16;
        unsigned char* dst = Wt + buf * WSET;
        lds_dma16(src + wid * 1024, dst + wid * 1024);
        if (wid < 8) lds_dma16(src + (wid + 12) * 1024, dst + (wid + 12) * 1024);
    };
    // the six rows rr = 0..5 of a depth slice as ONE item list: item e = tid + n*768 (n < 5) -> row e / 544, position e % 544
    constexpr int ROW_ITEMS = HWP * 8;                // 544 float4 per row
    auto pro_row = [&](int e) { return e / ROW_ITEMS; };
    auto pro_load = [&](int sh, int n) {
        const int e = tid + n * 768, rr = pro_row(e);
        return rr < ROWS ? load_row(G0 - 2 + rr + sh, e - rr * ROW_ITEMS) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto pro_store = [&](int sh, int n, const float4& v) {
        const int e = tid + n * 768, rr = pro_row(e);
        if (rr < ROWS) store_row(rr, G0 - 2 + rr + sh, v, e - rr * ROW_ITEMS);
    };

    float biasv[2];
    float4 hvA = make_float4(0.f, 0.f, 0.f, 0.f), hvB = hvA, hvP0 = hvA, hvP1 = hvA, hvP2 = hvA, hvP3 = hvA, hvP4 = hvA;
    {
        const int sh = -2 * H;
        uint4 am = amax_load(a.xmax);
        const float winv = reinterpret_cast<const float*>(a.wsh)[1];
        {
            const float* bp = a.bias ? a.bias : a.x;
#pragma unroll
            for (int n = 0; n < 2; ++n) biasv[n] = bp[a.bias ? n * 16 + li : 0];
        }
        hvP0 = pro_load(sh, 0); hvP1 = pro_load(sh, 1); hvP2 = pro_load(sh, 2); hvP3 = pro_load(sh, 3); hvP4 = pro_load(sh, 4);
        hvA = load_row(G0 + 4 + sh, tid);             // rr = 6: the new row of tap row 1
        __builtin_amdgcn_sched_barrier(0);
        dma_w(0, 0);
        dma_w(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        float sai;
        amax_scale_of(am, sa, sai);
        out_scale = sai * winv;
        pro_store(sh, 0, hvP0); pro_store(sh, 1, hvP1); pro_store(sh, 2, hvP2); pro_store(sh, 3, hvP3); pro_store(sh, 4, hvP4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    C6_STAMP(201);
    C3_BARRIER();

    f32x4 acc[2][2], acl[2][2];                       // [pixel tile][channel tile]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) { acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; acl[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int a_lane = (half * 32 + li) * PS + g * 16;                     // A: pixel tile 0, tap 0 (tile m, tap dx: + (16 m + dx) * PS)
    const unsigned char* b_lane = Wt + li * 64 + ((g ^ swzb(li)) << 4);   // B: channel tile 0 (tile n: + 16 n * 64; tap dx: + dx * 2 * WPL)

    // one tap row of depth slice kd (weight set t = 5 kd + dy in buffer wb).  MODE 0: request the row of tap row dy + 2 of this
    // slice into hin; MODE 1 (dy == 3): request the next slice's first six rows into hvP; MODE 2 (dy == 4): request the next
    // slice's row G0+4 into hin.  last: no next slice.  Between the taps: hout = row G0+dy+4 of this slice goes to LDS (dy < 4)
    // and weight set t + 2 is requested into buffer wb2 (free since the barrier of tap row t - 1).
    auto tap_row = [&](const int kd, const int dy, const int mode, const bool last, const int wb, const int wb2, float4& hin,
                       const float4& hout) __attribute__((always_inline)) {
        const int sh = (kd - 2) * H, shn = (kd - 1) * H;
        C6_STAMP((kd * 5 + dy) * 8 + 0);
        if (mode == 0) {
            hin = load_row(G0 + dy + 5 + sh, tid);    // rr = dy + 7: the new row of tap row dy + 2
        } else if (!last) {
            if (mode == 1) {
                hvP0 = pro_load(shn, 0); hvP1 = pro_load(shn, 1); hvP2 = pro_load(shn, 2); hvP3 = pro_load(shn, 3); hvP4 = pro_load(shn, 4);
            } else {
                hin = load_row(G0 + 4 + shn, tid);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int src = gy + sh + dy - 2;
        const bool plane_ok = dpl + kd - 2 >= 0 && dpl + kd - 2 < D;
        const bool has_taps = tvalid && plane_ok && src >= row_lo + sh && src < row_hi + sh;
        const unsigned char* hrow = ring + ((r + dy) & 7) * SLOT + a_lane;
        const unsigned char* wbuf = b_lane + wb * WSET;
        // operands are NOT double buffered across taps (the registers are not there at three waves per SIMD); three waves per
        // SIMD cover the LDS latency
        // operands are not double buffered across taps (measured: no gain; three waves per SIMD cover the LDS latency)
        uint4 ao[2][2], bo[2][2];                     // [tile][plane]
        auto taps = [&](const int dx0, const int dx1) __attribute__((always_inline)) {
#pragma unroll
            for (int dx = dx0; dx < dx1; ++dx) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    bo[n][0] = *reinterpret_cast<const uint4*>(wbuf + dx * 2 * WPL + 16 * n * 64);
                    bo[n][1] = *reinterpret_cast<const uint4*>(wbuf + dx * 2 * WPL + 16 * n * 64 + WPL);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    ao[m][0] = *reinterpret_cast<const uint4*>(hrow + (16 * m + dx) * PS);
                    ao[m][1] = *reinterpret_cast<const uint4*>(hrow + (16 * m + dx) * PS + 64);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f16x8 a1 = __builtin_bit_cast(f16x8, ao[m][0]), a2 = __builtin_bit_cast(f16x8, ao[m][1]);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const f16x8 b1 = __builtin_bit_cast(f16x8, bo[n][0]), b2 = __builtin_bit_cast(f16x8, bo[n][1]);
                        acl[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, acl[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[m][n], 0, 0, 0);
                        acl[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, acl[m][n], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (has_taps) taps(0, 2);
        C6_STAMP((kd * 5 + dy) * 8 + 1);
        if (dy < 4) { store_row((dy + 6) & 7, G0 + dy + 4 + sh, hout, tid); __builtin_amdgcn_sched_barrier(0); }
        C6_STAMP((kd * 5 + dy) * 8 + 2);
        if (has_taps) taps(2, 4);
        C6_STAMP((kd * 5 + dy) * 8 + 3);
        const bool more_w = !last || dy < 3;          // set t + 2 exists (t + 2 <= 124)
        if (more_w) { dma_w(kd * 5 + dy + 2, wb2); __builtin_amdgcn_sched_barrier(0); }
        C6_STAMP((kd * 5 + dy) * 8 + 4);
        if (has_taps) taps(4, 5);
        C6_STAMP((kd * 5 + dy) * 8 + 5);
        // weight set t + 1 (requested during tap row t - 1) must have landed before the barrier publishes it: every vector-memory
        // operation of THIS tap row is younger -- the row request(s) at its head and this wave's one or two DMA pieces
        if (mode == 1 && !last) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (mode != 0 && last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        C6_STAMP((kd * 5 + dy) * 8 + 6);
        C3_BARRIER();
        C6_STAMP((kd * 5 + dy) * 8 + 7);
    };

    int wb0 = 0;                                      // buffer of weight set 5 kd
    for (int kd = 0; kd < 5; ++kd) {
        const bool last = kd == 4;
        const int w1 = wb0 == 2 ? 0 : wb0 + 1, w2 = w1 == 2 ? 0 : w1 + 1;     // (wb0 + 1) % 3, (wb0 + 2) % 3
        tap_row(kd, 0, 0, last, wb0, w2, hvB, hvA);
        tap_row(kd, 1, 0, last, w1, wb0, hvA, hvB);
        tap_row(kd, 2, 0, last, w2, w1, hvB, hvA);
        tap_row(kd, 3, 1, last, wb0, w2, hvA, hvB);
        tap_row(kd, 4, 2, last, w1, wb0, hvA, hvB);
        wb0 = w2;                                     // (wb0 + 5) % 3
        if (!last) {
            const int shn = (kd - 1) * H;
            pro_store(shn, 0, hvP0); pro_store(shn, 1, hvP1); pro_store(shn, 2, hvP2); pro_store(shn, 3, hvP3); pro_store(shn, 4, hvP4);
            C3_BARRIER();
        }
    }
    C6_STAMP(202);
    // ---- epilogue: the wave's [32 px][32 co] tile through LDS (the ring is free after the last barrier), 16-byte stores ----
    float* tb = reinterpret_cast<float*>(ring) + wid * (32 * OP);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float bias = a.bias ? biasv[n] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                tb[(16 * m + 4 * g + q) * OP + n * 16 + li] = (acc[m][n][q] + acl[m][n][q] * (1.f / 2048.f)) * out_scale + bias;
        }
    float vmax = 0.f;
    if (tvalid) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {                 // 256 float4 per wave
            const int e = lane + n * 64, px = e >> 3, c4 = e & 7;
            float4 v = *reinterpret_cast<const float4*>(&tb[px * OP + c4 * 4]);
            const size_t o4 = ((size_t)gy * W + half * 32 + px) * (OP / 4) + c4;
            if (a.res) { const float4 q = reinterpret_cast<const float4*>(a.res)[o4]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
            if (a.epi == SOL_EPI_LRELU) {
                v.x = v.x > 0.f ? v.x : a.slope * v.x; v.y = v.y > 0.f ? v.y : a.slope * v.y;
                v.z = v.z > 0.f ? v.z : a.slope * v.z; v.w = v.w > 0.f ? v.w : a.slope * v.w;
            } else if (a.epi == SOL_EPI_DLRELU) {     // backward: times LeakyReLU'(activation reference)
                const float4 q = reinterpret_cast<const float4*>(a.act)[o4];
                v.x *= q.x > 0.f ? 1.f : a.slope; v.y *= q.y > 0.f ? 1.f : a.slope;
                v.z *= q.z > 0.f ? 1.f : a.slope; v.w *= q.w > 0.f ? 1.f : a.slope;
            }
            vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            st_wt(reinterpret_cast<float4*>(a.y) + o4, v);
        }
    }
    if (a.ymax) amax_publish_last(vmax, a.ymax, reinterpret_cast<unsigned*>(smem_c6 + AMAX_LDS));
#ifdef SOL_C6_PROF
    C6_STAMP(203);
    if (lane == 0) c6_st[205] = (unsigned)__builtin_amdgcn_s_memrealtime();
    if (g_c6_prof)
        for (int i = lane; i < C6_NSTAMP; i += 64) g_c6_prof[((size_t)blockIdx.x * 12 + wid) * C6_NSTAMP + i] = c6_st[i];
#endif
}

#ifdef SOL_C6_PROF
constexpr size_t c6_lds() { return (size_t)8 * 68 * 160 + 3 * (size_t)5 * 2 * 32 * 64 + 32 + 12 * C6_NSTAMP * 4; }
#else
constexpr size_t c6_lds() { return (size_t)8 * 68 * 160 + 3 * (size_t)5 * 2 * 32 * 64 + 16; }
#endif

// ------------------------------------------------------------------------------------------------------------------------
// k_conv3d_sb8: EIGHT rows per workgroup, one whole 64-pixel row x 32 channels per wave, eight waves (two per SIMD, 256 VGPRs).
// What the phase stamps and the timing experiments on k_conv3d_sb6 show (profiles/r03_conv3d_sb6_phase_timeline.txt):
// no single bound -- operand reads (LDS 55 % busy), the staging blocks and the per-tap-row barrier each cost 13-29 %, and
// removing any one of them alone recovers only its share.  This form lowers all three per MFMA:
//   * a wave's weight fragments serve four pixel tiles: 12 ds_read_b128 (8 A + 4 B) per 24 MFMAs (0.5 per MFMA; 0.67 / 1.0 above);
//   * a weight set (LDS-DMA, three rotating buffers) and one new input row serve eight output rows; eight waves per barrier;
//   * the always-zero halo pixels are written once, so a row is exactly one float4 per thread;
//   * 8 192 rows / 8 = 1 024 workgroups = four full rounds of the 256 CUs at one simulation (1 366 six-row workgroups: 5.3);
//   * the operands of tap dx + 1 are read before the MFMAs of tap dx, and the A fragments of the NEXT tap row before the barrier
//     (rows are staged two tap rows ahead, so they are published a barrier early) -- the registers are there at two waves per SIMD.
// Measured at 128 x 64 x 64 (per launch): 279 us against 334 (six rows) and 412 (three rows); the tap loop alone (no staging,
// no barriers) runs at the matrix pipe's pace (198 us = 24.6 M MFMAs x 16 cycles / 1024 SIMDs at the 1.95 GHz the chip holds
// under this load).  What is left: row staging 61 us (29 the loads, 35 the VGPR -> LDS writes; the fp16 split itself is free
// once the SLP vectoriser is kept from packing it into v_pk_*_f32), weight DMA 32 us, barriers 40 us.  Neutral when tried:
// staggering the staging block between the two waves of a SIMD (kept), reading the next tap row's B fragments before the
// barrier as well, waiting for a weight set in the tap row that requests it.
// Input rows rr = 0..11 of a depth slice (global row G0 - 2 + rr) live in 11 LDS slots in the XOR layout of the kernels above
// (slot rr; row 11 reuses slot 0, dead since tap row 0): the A address of tap dx is one of five lane-dependent registers plus
// immediates (pixel tile, plane).
// DBG (option dbg_skip, timing experiments only, results invalid): 1 no taps, 8 no barriers, 32 no weight DMA, 64 no rows.
// ------------------------------------------------------------------------------------------------------------------------
template <int DBG> __device__ __forceinline__ void c8_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DBG) : "memory"); }
// NT = output-channel tiles of 16: 2 for the 32 -> 32 layers, 1 for the thin 32 -> (<= 16) layers (32 -> 3 forward, 32 -> 4 data
// gradient of the first layer): half the MFMAs and weight bytes on the same staging skeleton, outputs stored channel by channel.
// PERS (round 6, option k3d_conv_persist): a workgroup runs `tpw` CONSECUTIVE tiles (256 workgroups x 4 tiles at 128 x 64 x 64 instead of four
// rounds of 256): the tile boundary is treated like a depth-slice boundary -- the next tile's first nine rows and weight sets 0, 1 are requested
// during the last two tap rows of the tile -- with the epilogue in between, so that only the first tile of a workgroup pays the cold round trip
// in front of its first MFMA.  Same products in the same order per output: bit-identical to the one-tile form.  MEASURED SLOWER (default off):
// SOL-16 174.8 vs 163.8 ms -- the runtime tile loop around the unrolled tap rows spills 68 VGPRs (8 in the one-tile form).
template <int NT, int DBG, bool PERS = false>
__global__ void __launch_bounds__(512) k_conv3d_sb8(ConvArgs a, int nrows, int D, int tpw) {
    constexpr int OP = NT * 16, HWP = 68;
    constexpr int PLANE = HWP * 64, SLOT = 2 * PLANE, WPL = OP * 64, WSET = 5 * 2 * WPL;
    constexpr int NSLOT = 11, ROWS = 8, NWB = 3;
    constexpr int AMAX_LDS = NWB * WSET + NSLOT * SLOT;
    extern __shared__ __align__(16) unsigned char smem_c8[];
    if (threadIdx.x == 0) *reinterpret_cast<uint2*>(smem_c8 + AMAX_LDS) = make_uint2(0u, 0u);
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = wid;                                // this wave's row of the workgroup
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H;
    constexpr int W = 64;
    const int bx = xcd_tile(blockIdx.x, gridDim.x);
    const int ntile = PERS ? tpw : 1;
    int G0 = bx * ntile * ROWS;                       // (per tile: G0 .. row_hi change at a tile boundary)
    int gy = G0 + r;
    bool tvalid = gy < nrows;
    int plane = (tvalid ? gy : 0) / H, dpl = plane % D;
    int row_lo = plane * H, row_hi = row_lo + H;
    unsigned char* Wt = smem_c8;                      // weight buffers first: LDS-DMA destinations (M0) below 64 KB
    unsigned char* ring = smem_c8 + NWB * WSET;
    const float4* gx = reinterpret_cast<const float4*>(a.x);
    const unsigned char* gw = reinterpret_cast<const unsigned char*>(a.wsh) + 16;
    float sa = 1.f, out_scale = 1.f;

    // halo pixels hc = 0, 1, 66, 67 (x = -2, -1, 64, 65) of every slot and plane are zero for the whole launch (PERS: written again behind
    // every epilogue, whose transposition buffers lie over the ring)
    auto zero_halo = [&]() {
        if (tid < NSLOT * 2 * 4 * 4) {
            const int c = tid & 3, px = (tid >> 2) & 3, pl = (tid >> 4) & 1, sl = tid >> 5;
            *reinterpret_cast<uint4*>(ring + sl * SLOT + pl * PLANE + (px < 2 ? px : 64 + px) * 64 + c * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
    };
    zero_halo();
    // a row = 64 pixels x 8 float4 = one item per thread: pixel x = tid >> 3 (LDS column hc = x + 2), channels 4 (tid & 7) ..
    const int st_off = ((tid >> 3) + 2) * 64 + (((((tid & 7) >> 1) ^ swzb((tid >> 3) + 2)) << 4) | ((tid & 1) << 3));
    auto row_in = [&](int gr) { return gr >= 0 && gr < nrows; };
    auto load_row = [&](int gr) {
        if (DBG & 64) return make_float4(0.f, 0.f, 0.f, 0.f);
        return gx[row_in(gr) ? (size_t)gr * (W * 8) + tid : (size_t)tid];
    };
    auto store_row = [&](int slot, int gr, float4 v) {
        if (DBG & 64) return;
        const float sc = row_in(gr) ? sa : 0.f;       // rows outside the tensor: the (finite) clamped load times zero
        unsigned p0[2], p1[2];
        split2h(v.x, v.y, sc, p0[0], p1[0]);
        split2h(v.z, v.w, sc, p0[1], p1[1]);
        unsigned char* q = ring + slot * SLOT + st_off;
        *reinterpret_cast<uint2*>(q) = make_uint2(p0[0], p0[1]);
        *reinterpret_cast<uint2*>(q + PLANE) = make_uint2(p1[0], p1[1]);
    };
    // LDS-DMA: 16 bytes per active lane from per-lane global addresses to dst + 16 * lane (conv5x5_sb.hip explains the form)
    auto lds_dma16 = [&](const void* src, unsigned char* dst_wave_uniform) __attribute__((always_inline)) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dst_wave_uniform);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lo));
    };
    // weight set t (one tap row, WSET = 20 480 / 10 240 bytes) into buffer buf: wave w moves bytes [CW w, CW (w + 1)), CW = WSET / 8,
    // in NDMA = 3 / 2 pieces of at most 1 KB (NDMA vector-memory operations per wave and set: the count the waits below rely on)
    constexpr int CW = WSET / 8, NDMA = (CW + 1023) / 1024;
    auto dma_w = [&](int t, int buf) __attribute__((always_inline)) {
        if (DBG & 32) return;
        const unsigned char* src = gw + (size_t)t * WSET + wid * CW + lane * 16;
        unsigned char* dst = Wt + buf * WSET + wid * CW;
#pragma unroll
        for (int pc = 0; pc < NDMA; ++pc) {
            constexpr int full = 64;
            const int lanes = (CW - pc * 1024) / 16;          // 64 except for the last piece
            if (lanes >= full || lane < lanes) lds_dma16(src + pc * 1024, dst + pc * 1024);
        }
    };

    float biasv[NT];
    float4 hvA = make_float4(0.f, 0.f, 0.f, 0.f), hvB = hvA;
    float4 hvP[ROWS + 1];
    {
        const int sh = -2 * H;
        uint4 am = amax_load(a.xmax);
        const float winv = reinterpret_cast<const float*>(a.wsh)[1];
        {
            const float* bp = a.bias ? a.bias : a.x;
#pragma unroll
            for (int n = 0; n < NT; ++n) biasv[n] = bp[a.bias ? (n * 16 + li < a.CO ? n * 16 + li : 0) : 0];
        }
#pragma unroll
        for (int n = 0; n <= ROWS; ++n) hvP[n] = load_row(G0 - 2 + n + sh);     // rr = 0..8: the rows of tap rows 0 and 1
        hvA = load_row(G0 + 7 + sh);                  // rr = 9: the new row of tap row 2
        __builtin_amdgcn_sched_barrier(0);
        dma_w(0, 0);
        dma_w(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        float sai;
        amax_scale_of(am, sa, sai);
        out_scale = sai * winv;
#pragma unroll
        for (int n = 0; n <= ROWS; ++n) store_row(n, G0 - 2 + n + sh, hvP[n]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    C3_BARRIER();

    f32x4 acc[4][NT], acl[4][NT];                     // [pixel tile][channel tile]
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) { acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; acl[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    int a_off[5];                                     // A: pixel tile 0 of tap dx (tile m: + 1024 m; lo plane: + PLANE)
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) a_off[dx] = (li + dx) * 64 + ((g ^ swzb(li + dx)) << 4);
    const unsigned char* b_lane = Wt + li * 64 + ((g ^ swzb(li)) << 4);   // B: channel tile 0 (tile n: + 1024 n; tap dx: + dx * 2 * WPL)
    uint4 ao[2][4][2], bo[2][NT][2];                   // operands, [buffer][tile][plane]: tap (dy, dx) uses buffer (dy + dx) & 1
    const bool early = wid >= 4;                      // waves 4..7 (the SIMD partners of waves 0..3) stage at the head of a tap row

    // One tap row of depth slice kd (weight set t = 5 kd + dy in buffer wb).
    // Staging block (once per tap row; waves 4..7 at the head, waves 0..3 after tap 1, so that the two waves of a SIMD never
    // stage at the same time): the row of tap row dy + 2 (requested a tap row ago) goes to LDS, the row after it is requested,
    // weight set t + 2 is requested into buffer wb2 (free since the barrier of tap row t - 1).  dy == 3 / 4 request the next
    // slice's rows rr = 0..8 / rr = 9 instead (last: no next slice).
    // Before tap 4 the A fragments of tap (dy + 1, 0) are read: every row of the next tap row was published a barrier ago
    // (rows are staged TWO tap rows ahead), so after the barrier only the four B reads stand before the first MFMA.
    auto tap_row = [&](const int kd, const int dy, const bool last, const int wb, const int wb2) __attribute__((always_inline)) {
        const int sh = (kd - 2) * H;
        // rr = 2 of the NEXT slice: of this tile's slice kd + 1, or (PERS, kd == 4, not the last tile) of the next tile's slice 0
        const int nb = kd == 4 ? G0 + ROWS - 2 * H : G0 + (kd - 1) * H;
        const bool more_w = !last || dy < 3;          // set t + 2 exists (t + 2 <= 24; PERS: sets 0, 1 of the next tile)
        const int t2 = kd * 5 + dy + 2 >= 25 ? kd * 5 + dy + 2 - 25 : kd * 5 + dy + 2;      // (25 tap-row sets of five taps)
        auto block = [&]() __attribute__((always_inline)) {
            if (dy == 0) { store_row(9, G0 + 7 + sh, hvA); hvB = load_row(G0 + 8 + sh); }
            if (dy == 1) { store_row(10, G0 + 8 + sh, hvB); hvA = load_row(G0 + 9 + sh); }
            if (dy == 2) { store_row(0, G0 + 9 + sh, hvA); }                      // rr = 11 in slot 0 (row 0: dead since tap row 0)
            if (dy == 3 && !last) {
#pragma unroll
                for (int n = 0; n <= ROWS; ++n) hvP[n] = load_row(nb - 2 + n);
            }
            if (dy == 4 && !last) hvA = load_row(nb + 7);
            __builtin_amdgcn_sched_barrier(0);
            if (more_w) dma_w(t2, wb2);
            __builtin_amdgcn_sched_barrier(0);
        };
        const int src = gy + sh + dy - 2;
        const bool plane_ok = dpl + kd - 2 >= 0 && dpl + kd - 2 < D;
        const bool has_taps = !(DBG & 1) && tvalid && plane_ok && src >= row_lo + sh && src < row_hi + sh;
        const int rs = r + dy == NSLOT ? 0 : r + dy, rsn = r + dy + 1 == NSLOT ? 0 : r + dy + 1;
        const unsigned char* hrow = ring + rs * SLOT;
        const unsigned char* hnext = ring + rsn * SLOT;
        const unsigned char* wbuf = b_lane + wb * WSET;
        auto load_b = [&](int dx, int q) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                bo[q][n][0] = *reinterpret_cast<const uint4*>(wbuf + dx * 2 * WPL + 1024 * n);
                bo[q][n][1] = *reinterpret_cast<const uint4*>(wbuf + dx * 2 * WPL + 1024 * n + WPL);
            }
        };
        auto load_a = [&](const unsigned char* row, int dx, int q) {
            const unsigned char* ap = row + a_off[dx];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                ao[q][m][0] = *reinterpret_cast<const uint4*>(ap + 1024 * m);
                ao[q][m][1] = *reinterpret_cast<const uint4*>(ap + 1024 * m + PLANE);
            }
        };
        auto taps = [&](const int dx0, const int dx1) __attribute__((always_inline)) {
#pragma unroll
            for (int dx = dx0; dx < dx1; ++dx) {
                const int q = (dy + dx) & 1;
                if (dx == 0) { load_b(0, q); if (dy == 0) load_a(hrow, 0, q); }     // A of tap 0: read before the last barrier (dy > 0)
                if (dx < 4) { load_b(dx + 1, q ^ 1); load_a(hrow, dx + 1, q ^ 1); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f16x8 a1 = __builtin_bit_cast(f16x8, ao[q][m][0]), a2 = __builtin_bit_cast(f16x8, ao[q][m][1]);
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const f16x8 b1 = __builtin_bit_cast(f16x8, bo[q][n][0]), b2 = __builtin_bit_cast(f16x8, bo[q][n][1]);
                        acl[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, acl[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[m][n], 0, 0, 0);
                        acl[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, acl[m][n], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (early) block();
        if (has_taps) taps(0, 2);
        if (!early) block();
        if (has_taps) taps(2, 4);
        if (dy < 4) {
            load_a(hnext, 0, (dy + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (has_taps) taps(4, 5);
        // Weight set t + 1 (requested during tap row t - 1) must have landed before the barrier publishes it.  Younger vector-memory
        // operations of this wave: the row requests of this tap row's staging block and its three DMA pieces.
        if (dy < 2) c8_wait_vm<1 + NDMA>();
        else if (dy == 2) c8_wait_vm<NDMA>();
        else if (last) c8_wait_vm<0>();
        else if (dy == 3) c8_wait_vm<9 + NDMA>();
        else c8_wait_vm<1 + NDMA>();
        if (!(DBG & 8)) {
            // LDS operations complete in order: lgkmcnt(8) leaves the eight prefetched A reads in flight and drains the staging stores
            if (dy < 4) asm volatile("s_waitcnt lgkmcnt(8)\n\ts_barrier" ::: "memory");
            else C3_BARRIER();
        }
    };

    int wb0 = 0;                                      // buffer of weight set 5 kd
    float vmax = 0.f;
    for (int ti = 0; ti < ntile; ++ti) {
    const bool last_tile = ti == ntile - 1;
    for (int kd = 0; kd < 5; ++kd) {
        const bool last = kd == 4 && last_tile;
        const int w1 = wb0 == 2 ? 0 : wb0 + 1, w2 = w1 == 2 ? 0 : w1 + 1;     // (wb0 + 1) % 3, (wb0 + 2) % 3
        tap_row(kd, 0, last, wb0, w2);
        tap_row(kd, 1, last, w1, wb0);
        tap_row(kd, 2, last, w2, w1);
        tap_row(kd, 3, last, wb0, w2);
        tap_row(kd, 4, last, w1, wb0);
        wb0 = w2;                                     // (wb0 + 5) % 3
        if (kd < 4) {
            // slice boundary: every wave is past the barrier of tap row 4, slots 0..8 are free; the rows were requested two tap rows ago
            const int shn = (kd - 1) * H;
#pragma unroll
            for (int n = 0; n <= ROWS; ++n) store_row(n, G0 - 2 + n + shn, hvP[n]);
            C3_BARRIER();
        }
    }
    // ---- epilogue: the wave's [64 px][OP] tile through LDS (the ring is free after the last barrier) ----
    float* tb = reinterpret_cast<float*>(ring) + wid * (64 * OP);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float bias = a.bias ? biasv[n] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                tb[(16 * m + 4 * g + q) * OP + n * 16 + li] = (acc[m][n][q] + acl[m][n][q] * (1.f / 2048.f)) * out_scale + bias;
        }
    if (tvalid && a.CO == OP) {                       // full channel tiles: 16-byte stores
#pragma unroll
        for (int n = 0; n < OP / 4; ++n) {            // 64 * OP / 4 float4 per wave
            const int e = lane + n * 64, px = e / (OP / 4), c4 = e % (OP / 4);
            float4 v = *reinterpret_cast<const float4*>(&tb[px * OP + c4 * 4]);
            const size_t o4 = ((size_t)gy * W + px) * (OP / 4) + c4;
            if (a.res) { const float4 q = reinterpret_cast<const float4*>(a.res)[o4]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
            if (a.epi == SOL_EPI_LRELU) {
                v.x = v.x > 0.f ? v.x : a.slope * v.x; v.y = v.y > 0.f ? v.y : a.slope * v.y;
                v.z = v.z > 0.f ? v.z : a.slope * v.z; v.w = v.w > 0.f ? v.w : a.slope * v.w;
            } else if (a.epi == SOL_EPI_DLRELU) {     // backward: times LeakyReLU'(activation reference)
                const float4 q = reinterpret_cast<const float4*>(a.act)[o4];
                v.x *= q.x > 0.f ? 1.f : a.slope; v.y *= q.y > 0.f ? 1.f : a.slope;
                v.z *= q.z > 0.f ? 1.f : a.slope; v.w *= q.w > 0.f ? 1.f : a.slope;
            }
            vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            st_wt(reinterpret_cast<float4*>(a.y) + o4, v);
        }
    } else if (tvalid) {                              // CO < OP stored channels per pixel (the thin layers): the row's 64 * CO floats are contiguous
        const int n_out = 64 * a.CO;
        for (int e = lane; e < n_out; e += 64) {
            const int px = e / a.CO, c = e - px * a.CO;
            float v = tb[px * OP + c];
            const size_t o = (size_t)gy * W * a.CO + e;
            if (a.res) v += a.res[o];
            if (a.epi == SOL_EPI_LRELU) v = v > 0.f ? v : a.slope * v;
            else if (a.epi == SOL_EPI_DLRELU) v *= a.act[o] > 0.f ? 1.f : a.slope;
            vmax = fmaxf(vmax, fabsf(v));
            a.y[o] = v;
        }
    }
    if (PERS && !last_tile) {
        // tile boundary: the transposition buffers are read (barrier), the halo columns they covered are zeroed again, the next tile's rows
        // rr = 0..8 of slice 0 (requested during tap row 3 of the last slice) go to slots 0..8
        C3_BARRIER();
        zero_halo();
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) { acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; acl[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        G0 += ROWS;
        gy = G0 + r;
        tvalid = gy < nrows;
        plane = (tvalid ? gy : 0) / H; dpl = plane % D;
        row_lo = plane * H; row_hi = row_lo + H;
#pragma unroll
        for (int n = 0; n <= ROWS; ++n) store_row(n, G0 - 2 + n - 2 * H, hvP[n]);
        C3_BARRIER();
    }
    }   // tiles
    if (a.ymax) amax_publish_last(vmax, a.ymax, reinterpret_cast<unsigned*>(smem_c8 + AMAX_LDS));
}

constexpr size_t c8_lds(int NT) { return (size_t)3 * 5 * 2 * (NT * 16) * 64 + (size_t)11 * 2 * 68 * 64 + 16; }

// fp16 weight planes of all 125 taps with ONE power-of-two scale: header {2^shift_w, 2^-shift_w, 0, 0}, then
// out[tap = (kd*5 + dy)*5 + dx][plane 2][o 32][chunk s][j] in the LDS image order of the 2-D kernels (k_pack_sh).
// mode SOL_CONV_BWD_DATA: the flipped kernel with swapped channel axes (w is the FORWARD kernel [125][cout_run][cin_run]).
__global__ void __launch_bounds__(256) k_pack3_sh(const float* __restrict__ w, float* __restrict__ hdr, unsigned short* __restrict__ out, int mode,
                                                  int cout_run, int OP) {
    // run channels: 32 in, cout_run <= OP out (OP = 16 or 32 rows per plane, rows >= cout_run are zero).  FORWARD kernel layout of w:
    // SOL_CONV_FWD [125][32][cout_run]; SOL_CONV_BWD_DATA [125][cout_run][32] (the forward layer's cin = cout_run)
    __shared__ float red[4];
    float m = 0.f;
    // every workgroup finds the same maximum (128 000 floats for a 32 -> 32 layer: 16-byte loads when the tensor allows them -- the scalar loop
    // was 40 of this launch's 49 us, 22 launches per karman-3d training step)
    const int nw = 125 * 32 * cout_run;
    if ((reinterpret_cast<size_t>(w) & 15) == 0 && (nw & 3) == 0) {
        const float4* w4 = reinterpret_cast<const float4*>(w);
        for (int e = threadIdx.x; e < nw / 4; e += 256) {
            const float4 v = w4[e];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (int e = threadIdx.x; e < nw; e += 256) m = fmaxf(m, fabsf(w[e]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const unsigned mb = __float_as_uint(m);
    int ex = (int)(mb >> 23) - 127;
    ex = mb == 0u ? 0 : min(max(ex, -100), 100);
    const float sc = __uint_as_float((unsigned)(14 - ex + 127) << 23), inv = __uint_as_float((unsigned)(ex - 14 + 127) << 23);
    if (blockIdx.x == 0 && threadIdx.x == 0) { hdr[0] = sc; hdr[1] = inv; hdr[2] = 0.f; hdr[3] = 0.f; }
    const int total = 125 * OP * 16;                 // pairs of input channels
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int jp = e & 3, s = (e >> 2) & 3, o = (e >> 4) % OP, tap = e / (16 * OP);
        const int i0 = 8 * (s ^ swzb(o)) + 2 * jp;
        float v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = i0 + q;
            v[q] = o >= cout_run ? 0.f : mode == SOL_CONV_FWD ? w[((size_t)tap * 32 + i) * cout_run + o] : w[((size_t)(124 - tap) * cout_run + o) * 32 + i];
        }
        unsigned p[2];
        split2h(v[0], v[1], sc, p[0], p[1]);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            *reinterpret_cast<unsigned*>(out + ((((size_t)tap * 2 + pl) * OP + o) * 4 + s) * 8 + 2 * jp) = p[pl];
    }
}

constexpr size_t c3_lds() { return (size_t)4 * 2 * 68 * 64 + 2 * (size_t)5 * 2 * 32 * 64 + 16; }

}  // namespace

// floats of the fused kernel's weight section: header (4) + 125 taps x 2 planes x OP x 32 fp16 (OP = 32, or 16 for cout <= 16)
size_t sol_conv3d_sh_packed_floats(int cout) { return 4 + (size_t)125 * 2 * (cout <= 16 ? 16 : 32) * 16; }

int sol_conv3d_sh_pack(hipStream_t s, const float* w_dhwio, int mode, int cout, float* out) {
    SOL_LAUNCH(k_pack3_sh, dim3(64), dim3(256), 0, s, w_dhwio, out, reinterpret_cast<unsigned short*>(out + 4), mode, cout, cout <= 16 ? 16 : 32);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

// y = epi(conv3d(x, w) + bias (+ residual)), x / y [nplanes = B*D][H][64][32]; wsh from sol_conv3d_sh_pack; x_absmax required
int sol_conv3d_sb_launch(hipStream_t s, const float* x, const float* wsh, const float* bias, const float* residual, const float* act_ref, float* y,
                         int B, int D, int H, int cout, int epilogue, float slope, const unsigned* x_absmax, unsigned* y_absmax) {
    static std::atomic<unsigned long long> optin3{0}, optin8{0}, optin6{0};
    if (int e = sol_lds_optin(optin3, {SOL_K(k_conv3d_sb)}, "k_conv3d_sb", true)) return e;
    ConvArgs a{};
    a.x = x; a.bias = bias; a.res = residual; a.act = act_ref; a.y = y; a.B = B * D; a.H = H; a.W = 64; a.CO = cout; a.epi = epilogue; a.slope = slope;
    a.wsh = wsh; a.xmax = x_absmax; a.ymax = y_absmax; a.tiles_x = 1;
    const int nrows = B * D * H;
    if (cout <= 16) {                                 // thin layers: the eight-row kernel with one output-channel tile
        if (int e = sol_lds_optin(optin8, {SOL_K(k_conv3d_sb8<1, 0>), SOL_K(k_conv3d_sb8<2, 0>), SOL_K((k_conv3d_sb8<1, 0, true>)), SOL_K((k_conv3d_sb8<2, 0, true>)), SOL_K(k_conv3d_sb8<2, 1>), SOL_K(k_conv3d_sb8<2, 8>),
                                           SOL_K(k_conv3d_sb8<2, 32>), SOL_K(k_conv3d_sb8<2, 64>), SOL_K(k_conv3d_sb8<2, 104>)}, "k_conv3d_sb8")) return e;
        const int nt8 = (nrows + 7) / 8, grid8 = (nt8 + 7) / 8 * 8;
        if (sol_opt().k3d_conv_persist && nt8 % 256 == 0 && nt8 >= 512) SOL_LAUNCH((k_conv3d_sb8<1, 0, true>), dim3(256), dim3(512), c8_lds(1), s, a, nrows, D, nt8 / 256);
        else SOL_LAUNCH((k_conv3d_sb8<1, 0>), dim3(grid8), dim3(512), c8_lds(1), s, a, nrows, D, 1);
        SOL_LAUNCH_CHECK();
        return SOL_OK;
    }
    if (sol_opt().k3d_conv_rows == 8) {               // eight rows per workgroup, 64 x 32 tile per wave, two waves per SIMD
        if (int e = sol_lds_optin(optin8, {SOL_K(k_conv3d_sb8<1, 0>), SOL_K(k_conv3d_sb8<2, 0>), SOL_K((k_conv3d_sb8<1, 0, true>)), SOL_K((k_conv3d_sb8<2, 0, true>)), SOL_K(k_conv3d_sb8<2, 1>), SOL_K(k_conv3d_sb8<2, 8>),
                                           SOL_K(k_conv3d_sb8<2, 32>), SOL_K(k_conv3d_sb8<2, 64>), SOL_K(k_conv3d_sb8<2, 104>)}, "k_conv3d_sb8")) return e;
        const int nt8 = (nrows + 7) / 8, grid8 = (nt8 + 7) / 8 * 8;
#define C8_DBG(N) case N: { \
            SOL_LAUNCH((k_conv3d_sb8<2, N>), dim3(grid8), dim3(512), c8_lds(2), s, a, nrows, D, 1); break; }
        switch (sol_opt().dbg_skip) { C8_DBG(1) C8_DBG(8) C8_DBG(32) C8_DBG(64) C8_DBG(104) default: break; }
        if (sol_opt().dbg_skip) { SOL_LAUNCH_CHECK(); return SOL_OK; }
        if (sol_opt().k3d_conv_persist && nt8 % 256 == 0 && nt8 >= 512) SOL_LAUNCH((k_conv3d_sb8<2, 0, true>), dim3(256), dim3(512), c8_lds(2), s, a, nrows, D, nt8 / 256);
        else SOL_LAUNCH((k_conv3d_sb8<2, 0>), dim3(grid8), dim3(512), c8_lds(2), s, a, nrows, D, 1);
        SOL_LAUNCH_CHECK();
        return SOL_OK;
    }
    if (sol_opt().k3d_conv_rows == 6) {               // six rows per workgroup, 32 x 32 tile per wave
        const int nt6 = (nrows + 5) / 6, grid6 = (nt6 + 7) / 8 * 8;
        if (int e = sol_lds_optin(optin6, {SOL_K(k_conv3d_sb6)}, "k_conv3d_sb6")) return e;
        SOL_LAUNCH(k_conv3d_sb6, dim3(grid6), dim3(768), c6_lds(), s, a, nrows, D);
        SOL_LAUNCH_CHECK();
        return SOL_OK;
    }
    // a multiple of 8 workgroups, so that the XCD-aware tile order applies (xcd_tile): every XCD then owns a contiguous block of
    // planes and the five depth slices of a row come from ITS L2 (2.5 MB of reuse distance) instead of being fetched by all
    // eight L2s (measured without it: 745 MB of fabric reads per launch for 64 MB of input).  Padding tiles own no rows.
    const int ntiles = (nrows + 2) / 3, grid = (ntiles + 7) / 8 * 8;
    SOL_LAUNCH(k_conv3d_sb, dim3(grid), dim3(768), c3_lds(), s, a, nrows, D);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}
