// The 32 -> 32 layers of one CNN pass (model_mars_moon, /root/reference/karman-2d/karman_train.py:101-138: ten 5x5 SAME
// convolutions between the thin first and last layer; the reverse sweep's backward-data pass has the same shape) as ONE
// launch.  One workgroup per CU keeps its three image rows in LDS from layer to layer; only the two halo rows above and
// below cross workgroups, through global memory with one flag per (layer, workgroup):
//
//   The hand-off carries its own validity (form R2 of cdna_hip_programming.md guideline 16: "the data IS the flag"), because
//   a flag costs a second ~2.5 us memory round trip behind the data's and a drain of the producer's stores before it:
//   producer  after its epilogue knows the row's fp16 scale it writes every (pixel, 4 channels) item of its three rows as ONE
//             16-byte granule {hi0 lo0 hi1 lo1 | hi2 lo2 hi3 lo3} -- the split planes the consumer needs anyway -- into an
//             exchange buffer with write-through (sc1) stores.  The least significant bit of every lo half carries a tag bit:
//             each naturally aligned 8-byte half holds a 2-bit tag (8-byte stores are single-copy atomic), tag = 1 + (write
//             sequence number of the buffer mod 3); the buffer is double-buffered by layer parity, so the only stale content a
//             reader can meet (the buffer's previous writer: two layers ago, or the region's previous use one training step ago)
//             carries a different tag, and zero-filled memory carries 0.
//             The row's max|y| travels as an 8-byte {bits, tag} word.  No drain, no flag, no fence.
//   consumer  requests the granules of its four halo rows (sc1 loads, L1 bypassed) after tap step 0, checks the tags after
//             tap step 2 (re-requests what was not there yet; bounded), strips the tag bits and writes the planes to LDS.
//   Every reader of a row -- its owner and both neighbours -- uses the lo plane with the tag bit cleared (21 instead of 22
//   significant bits of the split; the results of neighbouring workgroups stay consistent bit for bit).
//   (MI355X_MICROARCH.md, inter-workgroup visibility and the hand-off price list.)
//
// A layer's 15 (output row, tap row) pairs run own-rows-first: steps 0..2 need only the workgroup's own input rows
// (row G0+s for all three tiles with tap row dy = 2+s-t), steps 3..4 the halo rows -- 4 us of MFMA work cover the
// neighbours' publish + the halo fetch.  What a kernel boundary used to cost per layer (dispatch, write drain, prologue
// round trips: ~5 us of a 13.5 us launch) is gone.
//
// Arithmetic: the fp16 three-product split of conv5x5_sb.hip (KIND 2) with the power-of-two operand scale taken per
// image ROW instead of per tensor (a per-tensor maximum would be a grid-wide dependency between layers): every (tile, tap
// row) pair accumulates into fresh accumulators that are folded in with the row's 2^-shift.  Never less accurate than the
// per-tensor form.  Weight planes: the packed fp16 section (conv5x5_sb.hip k_pack_jobs), all five tap-row sets resident.
#include "split_kernels.hpp"

namespace {

using namespace sbk;

constexpr int CH_HWP = 68;                          // halo pixels per row
constexpr int CH_PLANE = CH_HWP * 64;               // bytes per fp16 plane of a row
constexpr int CH_SLOT = 2 * CH_PLANE;               // 8,704 B per row (hi + lo plane)
constexpr int CH_WPL = 32 * 64;                     // bytes per (dx, plane) weight block
constexpr int CH_WBUF = 5 * 2 * CH_WPL;             // 20,480 B per tap-row weight set
constexpr int CH_W_OFF = 7 * CH_SLOT;               // 60,928
constexpr int CH_MISC_OFF = CH_W_OFF + 5 * CH_WBUF; // 163,328
constexpr int CH_LDS = CH_MISC_OFF + 128;           // 163,456 of 163,840
constexpr int CH_SPIN_LIMIT = 1 << 17;
// relative input row rr = row - (G0 - 2) in 0..6 -> LDS slot: the four halo rows first (contiguous: the epilogue's
// transposition buffers alias them), then the own rows
__device__ __forceinline__ int ch_slot(int rr) { return rr < 2 ? rr : (rr < 5 ? rr + 2 : rr - 3); }

// misc words: [0..6] row max bits, [8] workgroup absmax, [9] wave ticket, [16..22] row 2^-shift (float)
__device__ __forceinline__ void ch_scale(unsigned m, float& scale, float& inv) {
    int e = (int)(m >> 23) - 127;
    e = m == 0u ? 0 : min(max(e, -100), 100);
    scale = __uint_as_float((unsigned)(14 - e + 127) << 23);
    inv = __uint_as_float((unsigned)(e - 14 + 127) << 23);
}
__device__ __forceinline__ unsigned ch_max4(const float4& v) {
    return __float_as_uint(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// Row maximum of staged items: the 64 consecutive items of a wave lie in at most TWO rows (544 items per row), so two
// wave reductions and two LDS atomics per wave replace 64 same-address atomics (which the LDS serialises lane by lane:
// 36 instructions x 64 lanes were 1.1 us per layer).  m = bits of this lane's max (0 for lanes without an item), k = its row.
// wave maximum of non-negative float bit patterns with DPP row operations (__shfl_xor goes through the LDS crossbar: six
// dependent ds_bpermute round trips per reduction cost more than the whole split of a halo row)
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ unsigned ch_dpp(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, true);
}
__device__ __forceinline__ unsigned ch_wave_max(unsigned v) {
    v = max(v, ch_dpp<0xB1>(v));          // quad_perm [1,0,3,2]
    v = max(v, ch_dpp<0x4E>(v));          // quad_perm [2,3,0,1]
    v = max(v, ch_dpp<0x141>(v));         // row_half_mirror
    v = max(v, ch_dpp<0x140>(v));         // row_mirror: every lane holds its row-of-16 maximum
    v = max(v, ch_dpp<0x142, 0xa>(v));    // row_bcast:15
    v = max(v, ch_dpp<0x143, 0xc>(v));    // row_bcast:31: lane 63 holds the wave maximum
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ void ch_rowmax_publish(unsigned m, int k, unsigned* rowmax_words, const int* slot_of_k, int lane) {
    const int k0 = __builtin_amdgcn_readfirstlane(k);
    const unsigned a = ch_wave_max(k == k0 ? m : 0u), b = ch_wave_max(k == k0 ? 0u : m);
    if (lane == 0) {
        atomicMax(&rowmax_words[slot_of_k[k0]], a);
        if (b) atomicMax(&rowmax_words[slot_of_k[k0 + 1]], b);
    }
}
// Loop-invariant-code-motion fence: address arithmetic derived from the returned value is recomputed where it is used
// instead of being hoisted out of the layer loop and kept (spilled) across the MFMA steps.
__device__ __forceinline__ int ch_opaque(int x) { asm volatile("" : "+v"(x)); return x; }

// -DSOL_CHAIN_PROF (tools/chain_phase_probe.py builds such a library next to the product one): 100 MHz s_memrealtime stamps
// of thread 0 of every workgroup, 8 per layer: layer start, after each of the five tap steps, stores issued, layer end.
#ifdef SOL_CHAIN_PROF
__device__ long long* g_chain_prof = nullptr;
extern "C" int sol_chain_prof_set(long long* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_chain_prof), &buf, sizeof(buf)) == hipSuccess ? 0 : -1; }
#define SOL_CHSTAMP(l, k) do { if (threadIdx.x == 0 && g_chain_prof) g_chain_prof[((size_t)blockIdx.x * SOL_CHAIN_MAXL + (l)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SOL_CHSTAMP(l, k) do { } while (0)
#endif

__global__ void __launch_bounds__(768) k_cnn_chain(ChainArgs a) {
    extern __shared__ __align__(16) unsigned char smem_ch[];
    unsigned char* const rows = smem_ch;
    unsigned char* const wts = smem_ch + CH_W_OFF;
    unsigned* const misc = reinterpret_cast<unsigned*>(smem_ch + CH_MISC_OFF);
    float* const rowinv = reinterpret_cast<float*>(misc + 16);
    const int tid = threadIdx.x, grp = tid >> 8, lane = tid & 63, wave = (tid >> 6) & 3;
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H;
    constexpr int W = 64;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int G0 = tile * 3, gy = G0 + grp;
    const bool tvalid = gy < a.nrows;
    const int b = (tvalid ? gy : 0) / H;
    const int row_lo = b * H, row_hi = row_lo + H;
    const bool has_up = tile > 0, has_dn = tile + 1 < a.ntiles;
    const size_t tensor_bytes = (size_t)a.nrows * W * 32 * sizeof(float);
    const int pcc = wave * 16 + li;

    if (tid < 32) misc[tid] = 0u;

    // ---- helpers -----------------------------------------------------------------------------
    // item e in [0, 544): 4 channels (c4) of halo pixel hc of one row
    auto item_off = [&](int gr, int e) -> int {          // byte offset inside a [rows][64][32] tensor, or -1 (zero fill)
        const int hc = e >> 3, c4 = e & 7, xx = hc - 2;
        if (gr < 0 || gr >= a.nrows || xx < 0 || xx >= W) return -1;
        return ((gr * W + xx) * 8 + c4) * 16;
    };
    // planes of one item: hi pairs p0[2], lo pairs p1[2] with the tag bit (LSB of every lo half) cleared
    auto split_item = [&](const float4& v, float sc, unsigned (&p0)[2], unsigned (&p1)[2]) {
        split2h(v.x, v.y, sc, p0[0], p1[0]);
        split2h(v.z, v.w, sc, p0[1], p1[1]);
        p1[0] &= 0xfffefffeu; p1[1] &= 0xfffefffeu;
    };
    auto put_item = [&](int rr, int e, const unsigned (&p0)[2], const unsigned (&p1)[2]) {
        const int hc = e >> 3, c4 = e & 7;
        unsigned char* q = rows + ch_slot(rr) * CH_SLOT + hc * 64 + ((((c4 >> 1) ^ swzb(hc)) << 4) | ((c4 & 1) << 3));
        *reinterpret_cast<uint2*>(q) = make_uint2(p0[0], p0[1]);
        *reinterpret_cast<uint2*>(q + CH_PLANE) = make_uint2(p1[0], p1[1]);
    };
    auto write_item = [&](int rr, int e, const float4& v, float sc) {
        unsigned p0[2], p1[2];
        split_item(v, sc, p0, p1);
        put_item(rr, e, p0, p1);
    };
#ifndef SOL_CHAIN_NO_DMA
    constexpr bool CH_DMA = true;                     // weight sets by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write
#else
    constexpr bool CH_DMA = false;
#endif
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef __attribute__((address_space(1))) const void glb_cvoid_t;
    // set `dy` of `gw` -> weight slot `dy`: 20 chunks of 1 KB (64 lanes x 16 B, already in LDS image order), two per wave at most;
    // completion is tracked by vmcnt: the step ends with s_waitcnt vmcnt(0) before its barrier
    auto dma_wset = [&](int tidv, const uint4* gw, int dy) {
        const int wv = tidv >> 6, ln = tidv & 63;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int c = wv + 12 * n;
            if (c < 20)
                __builtin_amdgcn_global_load_lds((glb_cvoid_t*)(gw + (size_t)dy * (CH_WBUF / 16) + c * 64 + ln),
                                                 (lds_void_t*)(wts + dy * CH_WBUF + c * 1024), 16, 0, 0);
        }
    };
    constexpr int WV = CH_WBUF / 16;                  // 1280 uint4 per set
    auto load_wset = [&](int tid, const uint4* gw, int dy, uint4 (&v)[2]) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int e = tid + n * 768;
            v[n] = e < WV ? gw[(size_t)dy * WV + e] : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_wset = [&](int tid, int dy, const uint4 (&v)[2]) {
        uint4* dst = reinterpret_cast<uint4*>(wts + dy * CH_WBUF);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int e = tid + n * 768;
            if (e < WV) dst[e] = v[n];
        }
    };

    f32x4 total[2];
    // one (tile, tap row) pair: 5 taps x 3 split products x 2 column tiles, folded into `total` with the row's 2^-shift
    auto do_step = [&](int dy, int rr) {
        const int src = G0 - 2 + rr;                  // input row
        if (!(tvalid && src >= row_lo && src < row_hi)) return;     // wave uniform
        const unsigned char* hrow = rows + ch_slot(rr) * CH_SLOT;
        const unsigned char* wbuf = wts + dy * CH_WBUF;
        f32x4 acc[2], acl[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) { acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; acl[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        uint4 ao[2][2], bo[2][2][2];
        auto load_ops = [&](int dx, uint4 (&ar)[2], uint4 (&br)[2][2]) {
            const int hc = pcc + dx;
            const unsigned char* ap = hrow + hc * 64 + ((g ^ swzb(hc)) << 4);
            ar[0] = *reinterpret_cast<const uint4*>(ap);
            ar[1] = *reinterpret_cast<const uint4*>(ap + CH_PLANE);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int co = n * 16 + li;
                const unsigned char* bp = wbuf + dx * 2 * CH_WPL + co * 64 + ((g ^ swzb(co)) << 4);
                br[n][0] = *reinterpret_cast<const uint4*>(bp);
                br[n][1] = *reinterpret_cast<const uint4*>(bp + CH_WPL);
            }
        };
        load_ops(0, ao[0], bo[0]);
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
            if (dx < 4) load_ops(dx + 1, ao[(dx + 1) & 1], bo[(dx + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 a1 = __builtin_bit_cast(f16x8, ao[dx & 1][0]), a2 = __builtin_bit_cast(f16x8, ao[dx & 1][1]);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const f16x8 b1 = __builtin_bit_cast(f16x8, bo[dx & 1][n][0]), b2 = __builtin_bit_cast(f16x8, bo[dx & 1][n][1]);
                acl[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, acl[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[n], 0, 0, 0);
                acl[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, acl[n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const float ri = rowinv[rr];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) total[n][r] += (acc[n][r] + acl[n][r] * (1.f / 2048.f)) * ri;
    };

    // ---- chain prologue: the seven input rows of the first layer (written by an earlier launch: plain loads) and its
    //      weight sets 0..3 ----
    __syncthreads();                                  // misc zeroed
    {
        const float4* gx = reinterpret_cast<const float4*>(a.x0);
        float4 hv[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            const int e = tid + n * 768;
            hv[n] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 7 * 544) {
                const int rr = e / 544, off = item_off(G0 - 2 + rr, e - rr * 544);
                if (off >= 0) hv[n] = gx[off >> 4];
            }
        }
        const uint4* gw0 = reinterpret_cast<const uint4*>(a.L[0].wsh) + 1;
        uint4 wv[2][2];
        load_wset(tid, gw0, 0, wv[0]);
        load_wset(tid, gw0, 1, wv[1]);
        {
            const int all_rr[8] = {0, 1, 2, 3, 4, 5, 6, 6};
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const int e = tid + n * 768;
                const bool in = e < 7 * 544;
                ch_rowmax_publish(in ? ch_max4(hv[n]) : 0u, in ? e / 544 : 6, misc, all_rr, lane);
            }
        }
        store_wset(tid, 0, wv[0]);
        store_wset(tid, 1, wv[1]);
        load_wset(tid, gw0, 2, wv[0]);
        load_wset(tid, gw0, 3, wv[1]);
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            const int e = tid + n * 768;
            if (e < 7 * 544) {
                const int rr = e / 544;
                float sc, inv;
                ch_scale(misc[rr], sc, inv);
                write_item(rr, e - rr * 544, hv[n], sc);
                if (e - rr * 544 == 0) rowinv[rr] = inv;
            }
        }
        store_wset(tid, 2, wv[0]);
        store_wset(tid, 3, wv[1]);
        __syncthreads();
    }

    // Barriers inside the layer loop wait for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() also
    // waits for every global load / store in flight, which would serialise the prefetches (weights, flags, halo rows,
    // epilogue operands) and the write-through output stores with the MFMA steps they are meant to hide behind.
#define CH_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    uint4 stg[4];                                     // staging registers: a different load in every step (see the table below)
    // per-lane constants of the three halo items (item e = tid + n*768 of 4 rows x 544): kept in registers on purpose -- the
    // index arithmetic (two divisions, clamps, swizzle) would otherwise run ~100 VALU instructions per layer on 12 waves
    int h_goff[3], h_loff[3], h_row[3], h_flag[3];    // granule byte offset in a parity block, LDS byte offset, row index, flags
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        const int e = tid + n * 768, k = min(e / 544, 3), it = e - k * 544, rr = k < 2 ? k : k + 3, gr = G0 - 2 + rr;
        const int hc = it >> 3, c4 = it & 7, xx = hc - 2;
        const bool in = e < 4 * 544, rowreal = in && gr >= 0 && gr < a.nrows, real = rowreal && xx >= 0 && xx < W;
        const int grc = min(max(gr, 0), a.nrows - 1), xc = min(max(xx, 0), W - 1);
        h_goff[n] = ((grc * W + xc) * 8 + c4) * 16;
        h_row[n] = grc;
        h_loff[n] = ch_slot(rr) * CH_SLOT + hc * 64 + ((((c4 >> 1) ^ swzb(hc)) << 4) | ((c4 & 1) << 3));
        h_flag[n] = (in ? 1 : 0) | (real ? 2 : 0) | (rowreal ? 4 : 0) | ((in && it == 0) ? 8 : 0) | (rr << 4);
    }
    const unsigned ep = *a.epoch;                     // training-step epoch (written by an earlier launch of the step)
    // Tag of layer l's hand-off.  Each parity buffer sees the writers l = p, p+2, ... of one launch after those of the region's
    // previous use (epoch - 1); a reader must reject exactly the writer BEFORE the one it waits for, so the tags follow the
    // buffer's write sequence number modulo 3 (consecutive writers differ; 0 = never written).
    const unsigned nw0 = (unsigned)(a.nl) / 2u, nw1 = (unsigned)(a.nl - 1) / 2u;      // writers per launch (layers 0 .. nl-2) with parity 0 / 1
    auto chain_tag = [&](int l) -> unsigned { return 1u + (ep * ((l & 1) ? nw1 : nw0) + (unsigned)(l >> 1)) % 3u; };
    const size_t xrow = (size_t)W * 8;                // granules per row
    const int xbytes = (int)((size_t)2 * a.nrows * xrow * 16);
#pragma unroll 1
    for (int l = 0; l < a.nl; ++l) {
        const ChainLayer L = a.L[l];
        const bool more = l + 1 < a.nl;
        const uint4* gw = reinterpret_cast<const uint4*>(L.wsh) + 1;
        const uint4* gwn = more ? reinterpret_cast<const uint4*>(a.L[l + 1].wsh) + 1 : gw;
        const float winv = reinterpret_cast<const float*>(L.wsh)[1];
#pragma unroll
        for (int n = 0; n < 2; ++n) total[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        SOL_CHSTAMP(l, 0);
#ifdef CH_EXP_NOHALO                                    // timing experiment only (wrong results): what the layer costs without the exchange
        const bool exch = false;
#else
        const bool exch = l > 0;
#endif
        // hand-off of layer l-1's output: parity buffer (l-1)&1, tag 1 + (((l-1)/2 + epoch) mod 3)
        const unsigned tag_in = chain_tag(l - 1);
        const int par_in = (l - 1) & 1;
        unsigned long long hrm[3];                    // {row max bits, tag} of this lane's three halo items' rows
        float4 pres[2], pact[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) { pres[n] = make_float4(0.f, 0.f, 0.f, 0.f); pact[n] = make_float4(1.f, 1.f, 1.f, 1.f); }
        // request (again) the granules + row words of the halo rows: item e -> halo row k (0,1: above; 2,3: below), 544 items per row
        auto halo_issue = [&]() {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, xbytes, 0x00020000);
            const int pbase = par_in * (xbytes >> 1);
#pragma unroll
            for (int n = 0; n < 3; ++n) {      // unconditional loads (clamped addresses); items without data are zeroed
                stg[n] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, pbase + h_goff[n], 0, 16));
                hrm[n] = __hip_atomic_load(&a.rmx[(size_t)par_in * a.nrows + h_row[n]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!(h_flag[n] & 2)) stg[n] = make_uint4(0u, 0u, 0u, 0u);
            }
        };
        // Five tap steps.  Steps 0..2: all three tiles read the own row G0+s (tap row dy = 2+s-t); steps 3, 4: the halo rows.
        // One rolled loop (one copy of the MFMA body):
        //   s   loads issued before the MFMAs                    after them
        //   0   this layer's weight set 4 (LDS-DMA)              DMA landed; the halo granules + row words are requested
        //   1   --                                               --
        //   2   --                                               tags checked (re-request until valid), planes -> LDS
        //   3   next layer's set 2; residual / act reference     --
        //   4   next layer's sets 1 and 3                        --
        // (next layer's set 0 follows in the epilogue: it is the last one this layer reads.)
#pragma unroll 1
        for (int s = 0; s < 5; ++s) {
            const int tids = ch_opaque(tid);          // staging addresses of this step: recomputed here, not carried across the MFMAs
            if (s == 0) {
                if (CH_DMA) dma_wset(tids, gw, 4);
                else load_wset(tids, gw, 4, reinterpret_cast<uint4(&)[2]>(stg[0]));
            } else if (s == 3) {
                if (more) { if (CH_DMA) dma_wset(tids, gwn, 2); else load_wset(tids, gwn, 2, reinterpret_cast<uint4(&)[2]>(stg[0])); }
                if (tvalid) {
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int e = (tids & 63) + n * 64, px = e >> 3, c4 = e & 7;
                        const size_t o4 = ((size_t)(G0 + (tids >> 8)) * W + ((tids >> 6) & 3) * 16 + px) * 8 + c4;
                        if (L.res) pres[n] = reinterpret_cast<const float4*>(L.res)[o4];
                        if (L.epi == SOL_EPI_DLRELU) pact[n] = reinterpret_cast<const float4*>(L.act)[o4];
                    }
                }
            } else if (s == 4) {
                if (more) {
                    if (CH_DMA) { dma_wset(tids, gwn, 1); dma_wset(tids, gwn, 3); }
                    else { load_wset(tids, gwn, 1, reinterpret_cast<uint4(&)[2]>(stg[0])); load_wset(tids, gwn, 3, reinterpret_cast<uint4(&)[2]>(stg[2])); }
                }
            }

            int dy, rr;
            if (s < 3) { dy = 2 + s - grp; rr = 2 + s; }
            else if (s == 3) { dy = grp == 0 ? 1 : (grp == 1 ? 4 : 3); rr = grp == 0 ? 1 : 5; }
            else { dy = grp == 2 ? 4 : 0; rr = grp == 0 ? 0 : (grp == 1 ? 1 : 6); }
            do_step(dy, rr);

            const int tidp = ch_opaque(tid);
            // the step's LDS-DMA (issued before the MFMAs) has landed; waited BEFORE the halo request goes out
            // (step 3: the residual / activation-reference loads were issued AFTER the DMA and may stay in flight until the epilogue)
            if (CH_DMA && (s == 0 || s == 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (CH_DMA && s == 3) {
                const int nlate = tvalid ? ((L.res ? 2 : 0) + (L.epi == SOL_EPI_DLRELU ? 2 : 0)) : 0;
                if (nlate == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (nlate == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            if (s == 0) {
                if (!CH_DMA) store_wset(tidp, 4, reinterpret_cast<uint4(&)[2]>(stg[0]));
                if (tid == 0) { misc[2] = 0u; misc[3] = 0u; misc[4] = 0u; }      // own (output) row maxima of this layer
                if (exch) halo_issue();
            } else if (s == 2) {
                if (exch) {
                    unsigned spins = 0;
                    for (;;) {
                        bool ok = true;
#pragma unroll
                        for (int n = 0; n < 3; ++n) {
                            const uint4 g4 = stg[n];
                            const unsigned t0 = ((g4.x >> 16) & 1u) | ((g4.y >> 15) & 2u), t1 = ((g4.z >> 16) & 1u) | ((g4.w >> 15) & 2u);
                            ok = ok && (!(h_flag[n] & 2) || (t0 == tag_in && t1 == tag_in)) && (!(h_flag[n] & 4) || (unsigned)(hrm[n] >> 32) == tag_in);
                        }
#ifdef CH_EXP_NOCHECK                                    // timing experiment only: never wait for a neighbour (results may be stale)
                        ok = true;
#endif
                        if (__all(ok)) break;
                        if (++spins > CH_SPIN_LIMIT) {            // a neighbour never delivered (not resident?): flag the launch and go on
                            if (lane == 0) atomicOr(a.err, 1u);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(8);
                        halo_issue();
                    }
#pragma unroll
                    for (int n = 0; n < 3; ++n) {
                        if (h_flag[n] & 1) {
                            const uint4 g4 = stg[n];
                            // de-interleave {hi lo hi lo | hi lo hi lo} and strip the tag bits
                            unsigned char* q = rows + h_loff[n];
                            *reinterpret_cast<uint2*>(q) = make_uint2((g4.x & 0xffffu) | (g4.y << 16), (g4.z & 0xffffu) | (g4.w << 16));
                            *reinterpret_cast<uint2*>(q + CH_PLANE) = make_uint2(((g4.x >> 16) | (g4.y & 0xffff0000u)) & 0xfffefffeu,
                                                                                   ((g4.z >> 16) | (g4.w & 0xffff0000u)) & 0xfffefffeu);
                            if (h_flag[n] & 8) {
                                float sc, inv;
                                ch_scale((h_flag[n] & 4) ? (unsigned)hrm[n] : 0u, sc, inv);
                                rowinv[h_flag[n] >> 4] = inv;
                            }
                        }
                    }
                }
            } else if (s == 3) {
                if (more && !CH_DMA) store_wset(tidp, 2, reinterpret_cast<uint4(&)[2]>(stg[0]));
            } else if (s == 4) {
                if (more && !CH_DMA) { store_wset(tidp, 1, reinterpret_cast<uint4(&)[2]>(stg[0])); store_wset(tidp, 3, reinterpret_cast<uint4(&)[2]>(stg[2])); }
            }
            if (s != 0) CH_BARRIER();      // step 0 stages only weight set 4 (read from step 2 on): the barrier after step 1 covers it
            SOL_CHSTAMP(l, 1 + s);
        }
        // ---------------- epilogue ----------------------------------------------------------------------
        {
            const int tide = ch_opaque(tid);
            const int lane = tide & 63, wave = (tide >> 6) & 3, grp = tide >> 8, g = lane >> 4, li = lane & 15, gy = G0 + grp;
            if (more) { if (CH_DMA) dma_wset(tide, gwn, 0); else load_wset(tide, gwn, 0, reinterpret_cast<uint4(&)[2]>(stg[0])); }
            float* tb = reinterpret_cast<float*>(rows) + (grp * 4 + wave) * (16 * 32);      // aliases the (dead) halo slots
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float bias = L.bias ? L.bias[n * 16 + li] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) tb[(4 * g + r) * 32 + n * 16 + li] = total[n][r] * winv + bias;
            }
            float4 v[2];
            float vmax = 0.f;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int e = lane + n * 64, px = e >> 3, c4 = e & 7;
                float4 q = *reinterpret_cast<const float4*>(&tb[px * 32 + c4 * 4]);      // same-wave LDS round trip
                q.x += pres[n].x; q.y += pres[n].y; q.z += pres[n].z; q.w += pres[n].w;
                if (L.epi == SOL_EPI_LRELU) {
                    q.x = q.x > 0.f ? q.x : a.slope * q.x; q.y = q.y > 0.f ? q.y : a.slope * q.y;
                    q.z = q.z > 0.f ? q.z : a.slope * q.z; q.w = q.w > 0.f ? q.w : a.slope * q.w;
                } else if (L.epi == SOL_EPI_DLRELU) {
                    q.x *= pact[n].x > 0.f ? 1.f : a.slope; q.y *= pact[n].y > 0.f ? 1.f : a.slope;
                    q.z *= pact[n].z > 0.f ? 1.f : a.slope; q.w *= pact[n].w > 0.f ? 1.f : a.slope;
                }
                v[n] = q;
                vmax = fmaxf(vmax, __uint_as_float(ch_max4(q)));
                if (tvalid) reinterpret_cast<float4*>(L.y)[((size_t)gy * W + wave * 16 + px) * 8 + c4] = q;   // the tensor itself: plain store
            }
            const unsigned rm = ch_wave_max(tvalid ? __float_as_uint(vmax) : 0u);
            if (lane == 0) atomicMax(&misc[2 + grp], rm);
            if (more && !CH_DMA) store_wset(tide, 0, reinterpret_cast<uint4(&)[2]>(stg[0]));
            SOL_CHSTAMP(l, 6);
            CH_BARRIER();
            // per-tensor absmax (consumed by the weight-gradient kernels and the thin last layer): one atomic per workgroup
            if (tid == 0 && L.ymax) atomicMax(&L.ymax[blockIdx.x & (SOL_AMAX_SLOTS - 1)], max(max(misc[2], misc[3]), misc[4]));
            if (more) {          // own rows of the next layer's input: split with the row's scale -> own LDS slots AND the hand-off buffer
                float sc, inv;
                const unsigned rmax = misc[2 + grp];
                ch_scale(rmax, sc, inv);
                const unsigned tag_out = chain_tag(l);
                const unsigned tb0 = (tag_out & 1u) << 16, tb1 = (tag_out >> 1) << 16;
                const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, xbytes, 0x00020000);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int e = lane + n * 64, px = e >> 3, c4 = e & 7;
                    unsigned p0[2], p1[2];
                    split_item(tvalid ? v[n] : make_float4(0.f, 0.f, 0.f, 0.f), sc, p0, p1);
                    put_item(2 + grp, (wave * 16 + px + 2) * 8 + c4, p0, p1);
                    if (tvalid) {    // granule {hi0 lo0 | hi1 lo1 || hi2 lo2 | hi3 lo3}, tag bits in the lo halves' LSBs
                        u32x4 gq;
                        gq.x = (p0[0] & 0xffffu) | (p1[0] << 16) | tb0;
                        gq.y = (p0[0] >> 16) | (p1[0] & 0xffff0000u) | tb1;
                        gq.z = (p0[1] & 0xffffu) | (p1[1] << 16) | tb0;
                        gq.w = (p0[1] >> 16) | (p1[1] & 0xffff0000u) | tb1;
                        __builtin_amdgcn_raw_buffer_store_b128(gq, rx, (int)((((size_t)(l & 1) * a.nrows + gy) * xrow + (wave * 16 + px) * 8 + c4) * 16), 0, 16);
                    }
                }
                if ((tide & 255) == 0) {
                    rowinv[2 + grp] = inv;
                    if (tvalid) __hip_atomic_store(&a.rmx[(size_t)(l & 1) * a.nrows + gy], ((unsigned long long)tag_out << 32) | rmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (CH_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // next layer's weight set 0 (LDS-DMA) has landed
                CH_BARRIER();
            }
            SOL_CHSTAMP(l, 7);
        }
    }
#undef CH_BARRIER
}

int init_chain_kernel() {
    static std::atomic<unsigned long long> optin{0};
    return sol_lds_optin(optin, {SOL_K(k_cnn_chain)}, "k_cnn_chain");
}

int chain_cus() {
    static int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return cus;
    }();
    return n;
}

}  // namespace

// one workgroup per CU, all co-resident: the chain needs ceil(B*H/3) <= #CUs
bool sol_cnn_chain_usable(int B, int H, int W) {
    if (!sol_opt().cnn_persistent || sol_opt().conv_precision != 0 || W != 64) return false;
    const int ntiles = (B * H + 2) / 3;
    return ntiles >= 2 && ntiles <= chain_cus();
}

// hand-off region of one chain launch, in 32-bit words: 64 (error word) + row words [2 parities][rows] x 8 B + granules
// [2 parities][rows][64 px][8] x 16 B.  Must be ZERO before its first use (tag 0 = invalid); never needs zeroing again.
size_t sol_cnn_chain_flag_words(int B, int H, int /*nl*/) { return 64 + (size_t)2 * B * H * 2 + (size_t)2 * B * H * 64 * 8 * 4; }

// region: sol_cnn_chain_flag_words() words (zero before the first use); epoch: device word that changes by one between
// consecutive uses of the same region (training step counter)
int sol_cnn_chain_launch(hipStream_t s, const ChainLayer* layers, int nl, const float* x0, unsigned* flags, const unsigned* epoch, int B, int H, int W, float slope) {
    SOL_REQUIRE(layers && nl >= 1 && nl <= SOL_CHAIN_MAXL && x0 && flags && epoch && sol_cnn_chain_usable(B, H, W), "sol_cnn_chain_launch: bad arguments");
    if (int e = init_chain_kernel()) return e;
    ChainArgs a{};
    for (int l = 0; l < nl; ++l) a.L[l] = layers[l];
    a.nl = nl; a.x0 = x0; a.flags = flags; a.B = B; a.H = H; a.nrows = B * H; a.ntiles = (B * H + 2) / 3; a.slope = slope;
    a.err = flags;
    a.epoch = epoch;
    a.rmx = reinterpret_cast<unsigned long long*>(flags + 64);
    a.xbuf = reinterpret_cast<uint4*>(flags + 64 + (size_t)2 * a.nrows * 2);
    SOL_LAUNCH(k_cnn_chain, dim3(a.ntiles), dim3(768), (size_t)CH_LDS, s, a);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}
