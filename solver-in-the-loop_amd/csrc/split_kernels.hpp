// Split-operand helpers and the weight-gradient workgroup body of the 16-bit MFMA kernels, shared by conv5x5_sb.hip
// (stand-alone launches) and karman_step.hip (the weight gradients of an unrolled step ride in the SAME launch as
// that step's solver adjoint: 6 of 256 CUs would otherwise idle for its whole duration).
#pragma once
#include "common.hpp"

namespace sbk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ __forceinline__ int swzb(int idx) { return ((idx >> 2) & 1) << 1; }
// chunk swizzles of the weight-gradient staging planes (bww_sb_body): see the layout comment there
#ifdef BW_SWZ_READS_ONLY      // A/B switch (tools/ab_lib.py): the swizzles of rounds 1-4, conflict-free for the reads only
__host__ __device__ __forceinline__ int bw_swx(int ch) { return ch & 15; }
__host__ __device__ __forceinline__ int bw_swz(int ch) { return (ch >> 1) & 7; }
#else
__host__ __device__ __forceinline__ int bw_swx(int ch) { return ((ch ^ (ch >> 2)) & 3) | ((((ch >> 2) ^ (ch >> 4)) & 1) << 2) | (((ch >> 3) & 1) << 3); }
__host__ __device__ __forceinline__ int bw_swz(int ch) { return (((ch >> 1) ^ (ch >> 4)) & 1) | (((ch >> 2) & 3) << 1); }
#endif

// two fp32 -> packed pair of bf16 (round to nearest even), low half = first element
__device__ __forceinline__ unsigned pk_bf16(float x, float y) {
    const f32x2 f = {x, y};
    const bf16x2 b = __builtin_convertvector(f, bf16x2);
    return __builtin_bit_cast(unsigned, b);
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// (x, y) -> three packed bf16 pairs with x = x1 + x2 + x3 (+ O(2^-27 x))
__device__ __forceinline__ void split3(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = pk_bf16(x, y);
    const float rx = x - bf_lo(p1), ry = y - bf_hi(p1);
    p2 = pk_bf16(rx, ry);
    p3 = pk_bf16(rx - bf_lo(p2), ry - bf_hi(p2));
}

// ---- split fp16 ("KIND 2"): v * 2^shift = h1 + h2 / 2048 with two fp16 numbers (22 significant bits), 2^shift chosen per
// TENSOR from its max|v| so that the scaled tensor spans fp16's 29 normal binades below 2^15.  Three MFMA products
// (h1h1 into one accumulator, h1h2' + h2'h1 into a second one that is folded in with 2^-11 at the end): the dropped
// terms are <= 2^-22 relative to max|a| max|b| -- an ABSOLUTE error bound per tensor, which is what the relative-L2
// parity criterion measures (float64 check: 7.5e-8 before the fp32 accumulation, tools/conv_accuracy.py).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pk_f16(float x, float y) {
    const f32x2 f = {x, y};
    const f16x2 h = __builtin_convertvector(f, f16x2);
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void split2h(float x, float y, float scale, unsigned& p1, unsigned& p2) {
    const float xs = x * scale, ys = y * scale;
    p1 = pk_f16(xs, ys);
    const f16x2 h = __builtin_bit_cast(f16x2, p1);
    p2 = pk_f16((xs - (float)h.x) * 2048.f, (ys - (float)h.y) * 2048.f);
}
// KIND-2 split of the weight-gradient kernel: lo plane NOT pre-scaled (one accumulator for all three products);
// elements below 2^-19 of the tensor maximum lose their lo part to fp16 underflow -- an absolute error < 2^-30 max|v|.
__device__ __forceinline__ void split2u(float x, float y, float scale, unsigned& p1, unsigned& p2) {
    const float xs = x * scale, ys = y * scale;
    p1 = pk_f16(xs, ys);
    const f16x2 h = __builtin_bit_cast(f16x2, p1);
    p2 = pk_f16(xs - (float)h.x, ys - (float)h.y);
}

// ------------------------------------------------------------------------------------
// backward-weight, 32 -> 32 channels, W == 64:  dW[dy][dx][ci][co] = sum_px x[y+dy-2][px+dx-2][ci] * dz[y][px][co]
// ------------------------------------------------------------------------------------
// GEMM per tap with K = pixels: A[m = ci][k = px] = x^T, B[k = px][n = co] = dz.  The K = 32 bf16 MFMA
// wants 8 consecutive PIXELS of one channel per lane, so rows are transposed to [channel][pixel] bf16
// planes while they are split and staged (4 px x 4 ch per thread item, ds_write_b64).  One workgroup
// (8 waves) owns `rb` consecutive image rows and ALL 25 taps: an x row is staged once and meets the five
// dz rows y+2-dy of a 6-slot dz ring, so x and dz are read, split and staged once instead of five times.
// Wave = (ci tile, co tile, pixel half): 25 accumulator tiles each; the two pixel halves are folded
// through LDS at the end.  The dx shift along K is done in registers: a lane reads pixels 8g..8g+11 of
// its channel (b128 + b64) and builds the five shifted operands with v_alignbit (dx odd) or by register
// renaming (dx even) -- no unaligned LDS access and the x operand is reused for all five tap rows.
// LDS rows: x 16 chunks of 16 B per channel (68 px used), chunk ^= bw_swx(ci); dz 8 chunks, chunk ^= bw_swz(co).
// The swizzles serve BOTH sides (round 5; until then chunk ^= ci & 15 / (co >> 1) & 7 served the reads only and the transposed
// staging writes ran 4-way / 2-way conflicted: 3.1 M SQ_LDS_BANK_CONFLICT cycles per fused launch, profiles/r04_pmc_sq.txt):
//   reads   ds_read_b128, lane groups of 16 on 64 banks: the 16 channels li of a wave's tile (channel bits 0..3) must spread over the
//           16 chunk slots of a 256-byte bank row -- the map (bits 0..3) -> chunk must be a bijection;
//   writes  ds_write_b32, lane groups of 32 on 32 banks ((a / 4) mod 32 = (chunk & 7) * 4 + pixel pair & 3): a group holds one
//           pixel-pair quad and the eight channel QUADS c4 of a row item (channel = 4 c4 + c: bits 2..4 vary) -- the map
//           (bits 2..4) -> chunk & 7 must be a bijection too.
// Both are linear over GF(2) in the channel bits b0..b4:  x: (b0^b2, b1^b3, b2^b4, b3);  dz: (b1^b4, b2, b3)  [dz rows are 128 B:
// b0 selects the half of the bank row].  A lane group's two chunk bases differ by 1 (g, g+1): the sets stay disjoint because the
// channel pairs that differ by the pre-image of 1 (li ^ 1 for x, li ^ 2 for dz) lie in the same half of the group.
#ifndef BWW_B_UPFRONT
#define BWW_B_UPFRONT 1
#endif
#ifndef BWW_PHASE_SHIFT
#define BWW_PHASE_SHIFT 1
#endif
#ifndef BWW_PIPE        // KIND 2: rows are staged ONE ROW FURTHER AHEAD (x row gr + 2 / dz row gr + 4 during row gr) and a wave reads the operand
#define BWW_PIPE 0      // fragments of row gr + 1 at the TAIL of row gr, in front of the row barrier: the first MFMA of a row waits for no LDS round trip
#endif
#ifndef BWW_REQ_LATE    // BWW_PIPE: the dz role (which stages at the head of the row, on the row's critical path) issues its requests BEHIND its staging
#define BWW_REQ_LATE 1
#endif
#ifndef BWW_PIN_ORDER
#define BWW_PIN_ORDER 1
#endif
#ifndef BWW_Z_ROTATE     // KIND 2: the dz fragments live in registers ACROSS rows (row gr's tap row dy is row gr+1's tap row dy+1): one new
#define BWW_Z_ROTATE 1   // fragment pair is read per row instead of five (0: all five re-read from the ring every row, rounds 4-6)
#endif
#ifndef BWW_DBG        // timing experiments (tools/ab_lib.py variants + tools/bww3d_time.py; results invalid): 1 no MFMAs, 2 no staging (the
#define BWW_DBG 0      // requests die with it), 4 no row barrier, 8 no requests.  Measured per 32 -> 32 Conv3D layer at 128 x 64 x 64 (five
#endif                 // passes + reduce, 464 us): 187 / 345 / 404 / 354 us; staging interleaved into the MFMA stream (branch-free, one basic block): 460 us
// -DBWW_PROF (tools/bww_row_probe.py): per-WAVE s_memtime stamps (low 32 bits), six per image row, kept in the unused LDS behind the KIND-2
// ring (81,920 .. 122,880) and dumped before the fold: [workgroup][wave][row][8] into the buffer set with sol_bww_prof_set().
#ifdef BWW_PROF
static __device__ unsigned* g_bww_prof = nullptr;
#define BWW_STAMP(row, k) do { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); if (lane == 0 && (row) - r0 < 40) bww_st[((row) - r0) * 8 + (k)] = t_; } while (0)
#else
#define BWW_STAMP(row, k) do { } while (0)
#endif
constexpr int BW_XPL = 32 * 256;                          // bytes: x plane of one row stage
constexpr int BW_ZPL = 32 * 128;                          // bytes: dz plane of one row stage
constexpr int BW_LDS = 2 * 3 * BW_XPL + 6 * 3 * BW_ZPL;   // 122,880 B (three planes; also >= the 102,400 B fold buffer)

// KIND 0: bf16 planes x3, six products;  KIND 2: fp16 planes x2 scaled per segment by the absmax of x / dz, three products
// (the host guarantees that a workgroup's rows lie in ONE segment)
template <int KIND>
__device__ __forceinline__ void bww_sb_body(const BwArgs& a, const int blk, unsigned char* smem_sb) {
    constexpr int W = 64;
    constexpr int NPL = KIND == 2 ? 2 : 3;
    constexpr int BW_XST = NPL * BW_XPL, BW_ZST = NPL * BW_ZPL;
    unsigned char* const XS = smem_sb;
    unsigned char* const ZS = smem_sb + 2 * BW_XST;
    // the wave index is read into an SGPR: staging role, tile coordinates and every row-range test below are then scalar
    // branches (an exec-masked region per role made the compiler wait for a load inside the region it was issued in)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int mt = wave & 1, nt = (wave >> 1) & 1, kb = wave >> 2;
    const int H = a.H;
    const int R = a.nseg * a.B * H, RPS = a.B * H;
    const int r0 = blk * a.rb, r1 = min(r0 + a.rb, R);

    // Staging roles: waves 0..3 move the x row, waves 4..7 the dz row.  One item = 2 pixels x 4 channels (two float4 requests);
    // both roles have exactly 256 items per row: x items are the image pixel pairs (2i, 2i+1) -> halo positions (2i+2, 2i+3);
    // the halo positions 0, 1, 66, 67 of every x stage are zero for good (written once below).  No request is predicated.
    const bool xrole = wave < 4;
#ifdef BWW_PROF
    unsigned* const bww_st = reinterpret_cast<unsigned*>(smem_sb + 81920) + wave * 320;      // 8 waves x 40 rows x 8 stamps x 4 B = 10 KB
#endif
    const int it = tid & 255;
    const int pxg = it >> 3, c4 = it & 7;          // 2-pixel group, channel quad
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    float sx = 1.f, sz = 1.f, out_scale = 1.f;
    if constexpr (KIND == 2) {
        const int seg0 = r0 / RPS;
        float ix, iz;
        amax_scale(a.xmax + (size_t)seg0 * a.xmax_seg, sx, ix);
        amax_scale(a.zmax + (size_t)seg0 * a.zmax_seg, sz, iz);
        out_scale = ix * iz;
    }

    // this thread's item of x row gx_row (x role) or dz row gz_row (dz role); rows outside [0, R) are replaced by row r0 (their
    // item is requested and dropped)
    // the role's tensor and segment stride, selected ONCE and made opaque (left as `xrole ? a.x : a.dz` at the point of use the
    // compiler built a two-entry table in scratch memory and read it back in every row)
    unsigned long long role_base = (unsigned long long)(xrole ? a.x : a.dz);
    long role_seg = xrole ? a.x_seg : a.dz_seg;
    asm volatile("" : "+s"(role_base), "+s"(role_seg));
    auto request = [&](int gx_row, int gz_row, float4& v0, float4& v1) __attribute__((always_inline)) {
        const int gr = xrole ? gx_row : gz_row;
        // (global address space spelled out: a pointer made from the opaque integer is generic, and flat loads also count in
        // lgkmcnt -- the LDS-only barriers below would wait for them)
        typedef const f32x4 __attribute__((address_space(1)))* gf4p;
        const int rr = gr >= 0 && gr < R ? gr : r0, seg = rr / RPS, grs = rr - seg * RPS;
        gf4p rp = (gf4p)(role_base + ((unsigned long long)seg * role_seg + (unsigned long long)grs * W * 32) * sizeof(float));
        if (BWW_DBG & 8) { v0 = v1 = make_float4(1e-3f, 2e-3f, -1e-3f, 5e-4f); return; }
        const f32x4 q0 = rp[(2 * pxg) * 8 + c4], q1 = rp[(2 * pxg + 1) * 8 + c4];
        v0 = make_float4(q0[0], q0[1], q0[2], q0[3]);
        v1 = make_float4(q1[0], q1[1], q1[2], q1[3]);
    };
    // split the 2 px x 4 ch item and write it transposed: per channel one 4-byte piece (2 pixels) per plane
    auto store_item = [&](const float4& v0, const float4& v1, int pg, unsigned char* base, int plane_bytes, int row_bytes, bool is_x)
                          __attribute__((always_inline)) {
        const float e[4][2] = {{v0.x, v1.x}, {v0.y, v1.y}, {v0.z, v1.z}, {v0.w, v1.w}};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ch = 4 * c4 + c;
            unsigned p[3];
            if constexpr (KIND == 2) split2u(e[c][0], e[c][1], is_x ? sx : sz, p[0], p[1]);
            else split3(e[c][0], e[c][1], p[0], p[1], p[2]);
            const int sw = is_x ? bw_swx(ch) : bw_swz(ch);
            unsigned char* q = base + ch * row_bytes + ((((pg >> 2) ^ sw) << 4) | ((pg & 3) << 2));
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<unsigned*>(q + pl * plane_bytes) = p[pl];
        }
    };
    // write the item as x row gx_row / dz row gz_row (callers pass rows that are in range for the thread's role)
    auto stage = [&](int gx_row, int gz_row, const float4& v0, const float4& v1) __attribute__((always_inline)) {
        if (xrole) store_item(v0, v1, pxg + 1, XS + (gx_row & 1) * BW_XST, BW_XPL, 256, true);
        else {
            store_item(v0, v1, pxg, ZS + ((gz_row + 6) % 6) * BW_ZST, BW_ZPL, 128, false);
            if (gz_row >= r0 && gz_row < r1) {   // bias gradient: every owned dz row is staged exactly once
                bs[0] += v0.x + v1.x;
                bs[1] += v0.y + v1.y;
                bs[2] += v0.z + v1.z;
                bs[3] += v0.w + v1.w;
            }
        }
    };
    auto z_in_range = [&](int gz) { return gz >= 0 && gz < R && gz <= r1 + 1; };

    // ---- prologue: dz rows r0-2 .. r0+2 and x row r0 go to LDS; the items of x rows r0+1, r0+2 / dz rows r0+3, r0+4 are
    //      requested into the register sets B and C (three rows of look-ahead: the operands come from HBM -- forward activations
    //      and gradients written hundreds of launches ago -- and a row of MFMA work covers about half of that round trip).
    //      PIPE (KIND 2): everything one row further -- x rows r0, r0+1 and dz rows r0-2 .. r0+3 staged, sets B / C = x rows r0+2, r0+3 /
    //      dz rows r0+4, r0+5 ----
    constexpr bool PIPE = KIND == 2 && BWW_B_UPFRONT && BWW_Z_ROTATE && BWW_PIPE;
    constexpr int XD = PIPE ? 2 : 1, ZD = PIPE ? 4 : 3;          // row gr stages x row gr + XD and dz row gr + ZD
    float4 sA0, sA1, sB0, sB1, sC0, sC1;
    sA0 = sA1 = sB0 = sB1 = sC0 = sC1 = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        constexpr int NZ0 = ZD + 2;                              // dz rows staged here: r0-2 .. r0+ZD-1
        float4 pv[NZ0][2];
        if (xrole) {
#pragma unroll
            for (int k = 0; k < XD; ++k) request(r0 + k < r1 ? r0 + k : r0, 0, pv[k][0], pv[k][1]);
        } else {
#pragma unroll
            for (int k = 0; k < NZ0; ++k) request(0, r0 - 2 + k, pv[k][0], pv[k][1]);
        }
        request(r0 + XD < r1 ? r0 + XD : r0, r0 + ZD, sB0, sB1);
        request(r0 + XD + 1 < r1 ? r0 + XD + 1 : r0, r0 + ZD + 1, sC0, sC1);
        __builtin_amdgcn_sched_barrier(0);
        if (xrole) {
#pragma unroll
            for (int k = 0; k < XD; ++k)
                if (r0 + k < r1) stage(r0 + k, 0, pv[k][0], pv[k][1]);
            // the zero halo positions (0, 1) and (66, 67) of both stages, all planes: 2 x NPL x 32 channels x 2 pieces
            for (int e = tid; e < 2 * NPL * 32 * 2; e += 256) {
                const int side = e & 1, ch = (e >> 1) & 31, pl = (e >> 6) % NPL, st = e / (64 * NPL);
                const int pg = side ? 33 : 0;
                *reinterpret_cast<unsigned*>(XS + st * BW_XST + pl * BW_XPL + ch * 256 + ((((pg >> 2) ^ bw_swx(ch)) << 4) | ((pg & 3) << 2))) = 0u;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NZ0; ++k)
                if (z_in_range(r0 - 2 + k)) stage(0, r0 - 2 + k, pv[k][0], pv[k][1]);
        }
    }
#define BW_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")   /* LDS only: requests stay in flight */
    BW_BARRIER();

    f32x4 acc[25];
#pragma unroll
    for (int tp = 0; tp < 25; ++tp) acc[tp] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    const int c0 = 4 * kb + g;
    const int swx_l = bw_swx(16 * mt + li), swz_l = bw_swz(16 * nt + li);      // this lane's channel rows of the x / dz operands

    // KIND 2, BWW_Z_ROTATE: Bz[dy] = this lane's fragments of dz row gr + 2 - dy.  Row gr + 1 meets the same dz rows one tap row further
    // down, so the set is SHIFTED at the head of a row and only Bz[0] (dz row gr + 2, staged during the previous row) is read: 6
    // ds_reads (13 -> 5 KB) per wave and row instead of 14 -- the operand reads of a row were 106 KB per CU, ~900 clocks of the LDS
    // pipe in front of every row's first MFMA.  Same products in the same order: bit-identical sums.
    uint4 Bz[5][NPL];
    auto read_z = [&](int gz, uint4 (&dst)[NPL]) __attribute__((always_inline)) {
        const unsigned char* zs = ZS + ((gz + 6) % 6) * BW_ZST + (16 * nt + li) * 128 + ((c0 ^ swz_l) << 4);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) dst[pl] = *reinterpret_cast<const uint4*>(zs + pl * BW_ZPL);
    };
    // PIPE: the raw x fragments (16 + 8 bytes per plane) of the NEXT row and the dz fragments of dz row gr + 3, read at the tail of row gr
    uint4 Aq[NPL], Bn[NPL];
    uint2 Ae[NPL];
    auto read_x = [&](int gx) __attribute__((always_inline)) {
        const unsigned char* xs = XS + (gx & 1) * BW_XST + (16 * mt + li) * 256;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            Aq[pl] = *reinterpret_cast<const uint4*>(xs + pl * BW_XPL + ((c0 ^ swx_l) << 4));
            Ae[pl] = *reinterpret_cast<const uint2*>(xs + pl * BW_XPL + (((c0 + 1) ^ swx_l) << 4));
        }
    };
    if constexpr (KIND == 2 && BWW_B_UPFRONT && BWW_Z_ROTATE) {
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) read_z(r0 + 1 - dy, Bz[dy]);      // rows r0+1 .. r0-2: do_row(r0) shifts them to dy = 1 .. 4
        if constexpr (PIPE) { read_x(r0); read_z(r0 + 2, Bn); }
    }

    // one image row: request the items of x row gr+XD+2 / dz row gr+ZD+2 into (i0, i1), run the 25 taps of row gr, write the items
    // held in (o0, o1) -- x row gr+XD / dz row gr+ZD, requested two iterations ago -- to LDS
    auto do_row = [&](const int gr, float4& i0, float4& i1, const float4& o0, const float4& o1) __attribute__((always_inline)) {
        const int y = gr % H;
        BWW_STAMP(gr, 0);
#ifdef BWW_PROF
        { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memrealtime(); if (lane == 0 && gr - r0 < 40) bww_st[(gr - r0) * 8 + 6] = t_; }      // 100 MHz: the clock the CU holds
#endif
        const bool req_late = PIPE && BWW_REQ_LATE && !xrole;            // (wave uniform)
        if (!req_late) request(gr + XD + 2 < r1 ? gr + XD + 2 : r0, gr + ZD + 2, i0, i1);
        __builtin_amdgcn_sched_barrier(0);
        // Phase shift between the two waves of a SIMD (waves w and w + 4): the dz role stages its row at the HEAD of the iteration, the x role
        // at its tail -- one wave's split / ds_write phase then lies under the other wave's MFMA block instead of both staging (matrix pipe
        // idle) and both multiplying at the same time.  Legal: the slot written (dz row gr+ZD) is read by nobody during this iteration.
        // Same-box A/B, three alternations (tools/ab_lib.py): 0 (both late) 11.554, 1 (dz early) 11.430, 2 (x early) 11.612, 3 (both early) 11.521 ms per step.
        if (BWW_PHASE_SHIFT == 1 && !xrole) {
            if (!(BWW_DBG & 2) && z_in_range(gr + ZD)) stage(gr + XD, gr + ZD, o0, o1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BWW_PHASE_SHIFT == 2 && xrole) {            // (variant: the x role stages early, the dz role late)
            if (!(BWW_DBG & 2) && gr + XD < r1) stage(gr + XD, gr + ZD, o0, o1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BWW_PHASE_SHIFT == 3) {                     // (variant: both roles stage early)
            if (!(BWW_DBG & 2) && (xrole ? gr + XD < r1 : z_in_range(gr + ZD))) stage(gr + XD, gr + ZD, o0, o1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (req_late) {
            request(gr + XD + 2 < r1 ? gr + XD + 2 : r0, gr + ZD + 2, i0, i1);
            __builtin_amdgcn_sched_barrier(0);
        }

        // x operand of this lane: pixels 8*c0 .. 8*c0+11 (halo coordinates) of channel 16*mt + li, three planes
        uint4 A[NPL][5];
        {
            if constexpr (!PIPE) read_x(gr);            // (PIPE: read at the tail of the previous row)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                const uint4 q = Aq[pl];
                const uint2 e = Ae[pl];
                A[pl][0] = q;
                A[pl][2] = make_uint4(q.y, q.z, q.w, e.x);
                A[pl][4] = make_uint4(q.z, q.w, e.x, e.y);
                A[pl][1] = make_uint4(__builtin_amdgcn_alignbit(q.y, q.x, 16), __builtin_amdgcn_alignbit(q.z, q.y, 16),
                                      __builtin_amdgcn_alignbit(q.w, q.z, 16), __builtin_amdgcn_alignbit(e.x, q.w, 16));
                A[pl][3] = make_uint4(__builtin_amdgcn_alignbit(q.z, q.y, 16), __builtin_amdgcn_alignbit(q.w, q.z, 16),
                                      __builtin_amdgcn_alignbit(e.x, q.w, 16), __builtin_amdgcn_alignbit(e.y, e.x, 16));
            }
        }
        if constexpr (KIND == 2 && BWW_B_UPFRONT) {
            // the dz fragments of ALL five tap rows are in registers at the head of the row (40 VGPRs): one exposed LDS
            // round trip per row instead of one per tap row in front of every group of 15 MFMAs (rows outside the image: the read
            // hits a valid ring slot and is dropped)
            uint4 (&Bv)[5][NPL] = Bz;
            BWW_STAMP(gr, 1);                           // (early staging issued; the stamp itself waits for lgkmcnt(0): LDS writes done)
            if constexpr (BWW_Z_ROTATE) {
#pragma unroll
                for (int dy = 4; dy > 0; --dy)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) Bv[dy][pl] = Bv[dy - 1][pl];
                if constexpr (PIPE) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) Bv[0][pl] = Bn[pl];
                } else read_z(gr + 2, Bv[0]);
            } else {
#pragma unroll
                for (int dy = 0; dy < 5; ++dy) read_z(gr + 2 - dy, Bv[dy]);
            }
            __builtin_amdgcn_sched_barrier(0);
            BWW_STAMP(gr, 2);                           // operand fragments in registers
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) {
                const int yz = y + 2 - dy;
                if (yz < 0 || yz >= H || (BWW_DBG & 1)) continue;        // workgroup uniform
                constexpr int QA[3] = {1, 0, 0}, QB[3] = {0, 1, 0};     // a2 b1, a1 b2, a1 b1
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int dx = 0; dx < 5; ++dx) {
                        acc[dy * 5 + dx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[QA[pr]][dx]), __builtin_bit_cast(f16x8, Bv[dy][QB[pr]]),
                                                                                  acc[dy * 5 + dx], 0, 0, 0);
                        // SOURCE ORDER pinned: the scheduler emitted the three groups of five as a snake (dx 0..4, 4..0, 0..4), i.e. two
                        // pairs of back-to-back DEPENDENT MFMAs per tap row; a dependent 16x16x32 MFMA issues ~4 slots after its
                        // producer (round-robin over five accumulators hides that, the snake does not: 21.6 instead of 16 clocks per
                        // MFMA while a wave has the matrix pipe to itself, tools/bww_row_probe.py)
                        if (BWW_PIN_ORDER) __builtin_amdgcn_sched_barrier(0);
                    }
            }
        } else {
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) {
                const int yz = y + 2 - dy;
                if (yz < 0 || yz >= H) continue;            // workgroup uniform
                const int gz = gr + 2 - dy;
                const unsigned char* zs = ZS + ((gz + 6) % 6) * BW_ZST + (16 * nt + li) * 128 + ((c0 ^ swz_l) << 4);
                uint4 Bv[NPL];
    #pragma unroll
                for (int pl = 0; pl < NPL; ++pl) Bv[pl] = *reinterpret_cast<const uint4*>(zs + pl * BW_ZPL);
                if constexpr (KIND == 2) {
                    constexpr int QA[3] = {1, 0, 0}, QB[3] = {0, 1, 0};     // a2 b1, a1 b2, a1 b1
    #pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
    #pragma unroll
                        for (int dx = 0; dx < 5; ++dx)
                            acc[dy * 5 + dx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[QA[pr]][dx]), __builtin_bit_cast(f16x8, Bv[QB[pr]]),
                                                                                      acc[dy * 5 + dx], 0, 0, 0);
                } else {
    #pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
    #pragma unroll
                        for (int dx = 0; dx < 5; ++dx)
                            acc[dy * 5 + dx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[PA[pr]][dx]), __builtin_bit_cast(bf16x8, Bv[PB[pr]]),
                                                                                       acc[dy * 5 + dx], 0, 0, 0);
                }
            }
}
        __builtin_amdgcn_sched_barrier(0);
        BWW_STAMP(gr, 3);                               // last MFMA issued
        __builtin_amdgcn_sched_barrier(0);
        if (!(BWW_DBG & 2) && (xrole ? ((BWW_PHASE_SHIFT & 2) == 0 && gr + XD < r1) : ((BWW_PHASE_SHIFT & 1) == 0 && z_in_range(gr + ZD)))) stage(gr + XD, gr + ZD, o0, o1);
        if constexpr (PIPE) {
            // operands of row gr + 1: x row gr + 1 and dz row gr + 3 were staged during row gr - 1 (visible since its barrier); the slots written
            // during THIS row (x row gr + 2, dz row gr + 4) are other slots.  The x role waits for the barrier here anyway; for the dz role the
            // round trip replaces the one that stood in front of its MFMA block.
            __builtin_amdgcn_sched_barrier(0);
            read_x(gr + 1);
            read_z(gr + 3, Bn);
        }
        BWW_STAMP(gr, 4);                               // late staging written (lgkmcnt(0))
        if (!(BWW_DBG & 4)) BW_BARRIER();
        BWW_STAMP(gr, 5);
    };
    // (six rows per trip: at the loop's back edge the compiler's wait-count pass gives up on the requests in flight and waits
    // for all of them -- once per six rows instead of once per three)
#pragma unroll 1
    for (int gr = r0; gr < r1; gr += 6) {
        // (early exits, not `if (gr + k < r1) do_row(...)`: behind a conditional row every register-resident dz fragment would be a
        // phi of its shifted and unshifted place -- 32 copies and their temporaries per row, 227 spilled registers)
        do_row(gr, sA0, sA1, sB0, sB1);
        if (gr + 1 >= r1) break;
        do_row(gr + 1, sB0, sB1, sC0, sC1);
        if (gr + 2 >= r1) break;
        do_row(gr + 2, sC0, sC1, sA0, sA1);
        if (gr + 3 >= r1) break;
        do_row(gr + 3, sA0, sA1, sB0, sB1);
        if (gr + 4 >= r1) break;
        do_row(gr + 4, sB0, sB1, sC0, sC1);
        if (gr + 5 >= r1) break;
        do_row(gr + 5, sC0, sC1, sA0, sA1);
    }
#ifdef BWW_PROF
    if (g_bww_prof)
        for (int i = lane; i < 320; i += 64) g_bww_prof[((size_t)blk * 8 + wave) * 320 + i] = bww_st[i];
#endif
    __syncthreads();                                     // (also retires the requests of rows beyond r1)

    // ---- fold the two pixel halves through LDS and add into this block's partial slice --------
    // All 512 threads move 16-byte pieces (4 consecutive co of one ci) into the block's partial slice.  Accumulating launches first
    // request ALL of the thread's thirteen old pieces (the slice was written an unrolled step ago: HBM-cold), then add and store --
    // as a load-add-store loop the compiler kept one or two requests in flight.  BWW_OLD_EARLY: the requests go out BEFORE the fold
    // (they depend on nothing it does), and the fold's two barriers are LDS-only, so the cold round trip lies under the fold's
    // LDS traffic instead of behind it.
    float* pw = a.partial + (size_t)blk * (25 * 1024);
    constexpr int NP = (4 * 25 * 64 + 511) / 512;       // 13 (the last one for threads < 256)
    auto dst_of = [&](int e) {
        const int c4 = e & 3, row = (e >> 2) & 15, tp = (e >> 6) % 25, wt = e / (25 * 64);      // wt = (mt, nt) tile
        return reinterpret_cast<float4*>(&pw[(tp * 32 + 16 * (wt & 1) + row) * 32 + 16 * (wt >> 1) + c4 * 4]);
    };
    float4 old[NP];
    auto request_old = [&]() __attribute__((always_inline)) {
        if (!a.overwrite) {
#pragma unroll
            for (int k = 0; k < NP; ++k) old[k] = *dst_of(min(tid + k * 512, 4 * 25 * 64 - 1));
        } else {
#pragma unroll
            for (int k = 0; k < NP; ++k) old[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#ifndef BWW_OLD_EARLY      // same-box A/B (tools/ab_lib.py, three alternations): 11.437 -> 11.392 ms per C3 step, karman-3d neutral; same sums bit for bit
#define BWW_OLD_EARLY 1
#endif
#if BWW_OLD_EARLY
#define BW_FOLD_BARRIER() BW_BARRIER()
    request_old();
#else
#define BW_FOLD_BARRIER() __syncthreads()
#endif
    float* red = reinterpret_cast<float*>(smem_sb);      // [4 waves][25 taps][256]
    if (kb == 1) {
#pragma unroll
        for (int tp = 0; tp < 25; ++tp)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave & 3) * 25 + tp) * 256 + (4 * g + r) * 16 + li] = acc[tp][r];
    }
    BW_FOLD_BARRIER();
    if (kb == 0) {      // second pixel half added in LDS, scaled: red = this block's complete [tile][tap][16 ci][16 co] sums
#pragma unroll
        for (int tp = 0; tp < 25; ++tp)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* q = &red[((wave & 3) * 25 + tp) * 256 + (4 * g + r) * 16 + li];
                *q = (acc[tp][r] + *q) * out_scale;
            }
    }
    BW_FOLD_BARRIER();
    {
        // (round 6, measured: the read-modify-write as no-return global_atomic_add_f32 -- the L2 does it, one add per address and launch, still bit
        //  reproducible -- 12.97 vs 11.12 ms per step: 100 KB of dword atomics per workgroup take ~58 us longer than load - add - store)
        // (round 6, measured: pulling the old slice into the L2 with one dword per line and thread under the MFMAs of the last six rows made the
        //  step SLOWER, 11.76 vs 11.67 ms over three alternations: the requests compete with the row loads the MFMAs are waiting for)
#if !BWW_OLD_EARLY
        request_old();
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int e = tid + k * 512;
            if (e < 4 * 25 * 64) {
                const int c4 = e & 3, row = (e >> 2) & 15, tp = (e >> 6) % 25, wt = e / (25 * 64);
                const float4 v = *reinterpret_cast<const float4*>(&red[(wt * 25 + tp) * 256 + row * 16 + c4 * 4]);
                *dst_of(e) = make_float4(old[k].x + v.x, old[k].y + v.y, old[k].z + v.z, old[k].w + v.w);      // (write-through stores measured no gain here: 12.86 vs 12.91 ms per step)
            }
        }
    }
    __syncthreads();
    float* redb = reinterpret_cast<float*>(smem_sb);     // [256 dz items][4]
    if (!xrole) {
#pragma unroll
        for (int c = 0; c < 4; ++c) redb[it * 4 + c] = bs[c];
    }
    __syncthreads();
    if (tid < 32) {
        float v = 0.f;
        for (int p = 0; p < 32; ++p) v += redb[((p << 3) | (tid >> 2)) * 4 + (tid & 3)];
        float* pb = a.partial + (size_t)a.nblk * (25 * 1024) + (size_t)blk * 32;
        pb[tid] = a.overwrite ? v : pb[tid] + v;
    }
}



}  // namespace sbk
