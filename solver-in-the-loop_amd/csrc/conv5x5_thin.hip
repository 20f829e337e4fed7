// 5x5 SAME convolution, 32 -> (<= 4) channels, W == 64: the THIN layers of model_mars_moon where a launch fills the chip -- the 32 -> 2
// output layer of the trainer (velocity correction + l2 loss in its epilogue, nothing stored) and the 32 -> 3 data gradient of the first
// layer -- in EXACT fp32 on the vector ALU (round 5).
//
// Why not the matrix cores: with <= 4 of 16 output columns in use an MFMA tile does 1/8 .. 1/4 useful work, and the split-precision
// kernels pay for their operand format in front of the first MFMA -- the absmax slots of the input must have arrived before the fp16
// split can start, every element is split on the VALU and written to LDS as two planes.  These launches are pure latency
// (k_conv5x5_sb<1, 2>: 10.5 us for 0.16 GFLOP), so the cheapest arithmetic that needs NO preparation wins: fp32 rows go to LDS as they
// are (no scale, no split, no absmax dependency: the requests leave as soon as the tile coordinates are known), and the K = 800 sum of a
// pixel runs as v_pk_fma_f32 with the WEIGHTS AS SGPR PAIRS (they are wave uniform).
//
// Decomposition: workgroup = three consecutive image rows x 64 pixels (the tiling of conv5x5_dx.hip: 256 workgroups at 128x64 x 6,
// XCD-aware order), 1024 threads = 16 waves (56 VGPRs: four waves per SIMD hide the scalar-load and LDS round trips of the unit loop,
// which the compiler has to wait for together -- both count in lgkmcnt).  Staging: seven input rows, 16-byte pieces, pixel pitch 144 B (128 + 16: with
// lanes = consecutive pixels a ds_read_b128 lane group then covers 16 distinct bank quads; 36 words is coprime enough with 64).
// Work unit = (output row r, tap row dy): 15 units of 5 dx x 32 ci, lanes = the 64 pixels of the row; wave u < 15 runs unit u.
// A unit reads 40 ds_read_b128 per lane for 80 x CON v_pk_fma_f32: the packed pair is (even, odd) input channel -- the natural halves
// of the b128 read and of the packed fp32 weights [tap][co][ci] (k_pack_jobs) -- so each output channel keeps two partial sums, folded
// at the end.  The units' sums meet in LDS ([unit][pixel][4]), one barrier, then 192 threads add the five tap rows of their pixel in a
// fixed order (bit reproducible) and run the epilogue.  Units whose input row lies in another image (or outside the tensor) are skipped
// by a scalar branch: the staging has no predicate, such rows are fetched from a clamped row and never read.
//
// Replaces keras.layers.Conv2D(2, 5, padding='same') of model_mars_moon (/root/reference/karman-2d/karman_train.py:137) + to_staggered +
// the velocity update + tf.nn.l2_loss (:88-90, 424-436), and the data gradient of the first Conv2D (:103).
#include "common.hpp"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(4)))* th_cfp;      // constant address space: uniform loads become scalar loads

constexpr int TH_PITCH = 144;                    // bytes per staged pixel
constexpr int TH_ROW = 68 * TH_PITCH;            // 64 pixels + 2 halo pixels on each side
constexpr int TH_NROWS = 7;
constexpr int TH_PART = 15 * 64 * 16;            // [unit][pixel][4 floats]
constexpr int TH_LDS = TH_NROWS * TH_ROW + TH_PART + 128;    // + {absmax word, ticket, loss ticket, pad x3, 16 wave slots}

template <int CON, int NW>                       // CON: output channels computed, 2 (CO <= 2) or 4 (CO <= 4); NW: waves per workgroup, 16 or 8
__global__ void __launch_bounds__(NW * 64) k_conv5x5_thin32(ConvArgs a, int nrows) {
    extern __shared__ __align__(16) unsigned char smem_th[];
    unsigned char* const rows = smem_th;
    float* const part = reinterpret_cast<float*>(smem_th + TH_NROWS * TH_ROW);
    unsigned* const slots = reinterpret_cast<unsigned*>(smem_th + TH_NROWS * TH_ROW + TH_PART);     // [0..1] absmax, [2..] loss ticket + wave slots
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H;
    constexpr int W = 64;
    const int G0 = xcd_tile(blockIdx.x, gridDim.x) * 3;
    if (tid < 32) slots[tid] = 0u;

    // ---- staging: every request goes out at once, no predicate (rows outside the tensor: a clamped row, never read) ----------------------
    const int c4 = tid & 7, p = (tid >> 3) & 63, hb = tid >> 9;      // NW = 16: thread = channel quad of pixel p in rows hb, hb + 2, hb + 4 (, 6); NW = 8: all seven rows
    const float4* gx = reinterpret_cast<const float4*>(a.x) + (size_t)p * 8 + c4;
    // (NAMED registers: an indexed array is left in scratch memory by the compiler -- a store behind every load)
    auto grow = [&](int s) { return (size_t)min(max(G0 - 2 + s, 0), nrows - 1) * W * 8; };      // scalar clamp
    const float4 v0 = gx[grow(NW == 16 ? hb : 0)], v1 = gx[grow(NW == 16 ? hb + 2 : 1)], v2 = gx[grow(NW == 16 ? hb + 4 : 2)], v3 = gx[grow(NW == 16 ? 6 : 3)];      // (NW = 16, row 6: both halves request it, the upper half drops it)
    float4 v4 = v3, v5 = v3, v6 = v3;
    if (NW == 8) { v4 = gx[grow(4)]; v5 = gx[grow(5)]; v6 = gx[grow(6)]; }
    // epilogue operands of the thread's pixel (threads 0..191: row r = wave 0..2, pixel = lane), requested with the rows: the velocity faces
    // and the ground-truth frames are HBM cold (read once per training step)
    const int er = wave, gy = G0 + er;
    const bool ep = wave < 3 && gy < nrows;                   // wave uniform
    const int eb = ep ? gy / H : 0, ejj = gy - eb * H;
    const CorrFaces f = corr_faces(a.ctr, H, W, ep ? ejj : 0, lane);
    const bool cm = a.cvy != nullptr && ep, gm = cm && a.gty != nullptr;
    const size_t nVy = a.ctr ? (size_t)(W + 1) * H : (size_t)(H + 1) * W, nVx = a.ctr ? (size_t)W * (H + 1) : (size_t)H * (W + 1);
    float pvy = 0.f, pvx = 0.f, pgy = 0.f, pgx = 0.f, pey = 0.f, pex = 0.f;
    if (cm) {
        pvy = a.cvy[eb * nVy + f.oy];
        pvx = a.cvx[eb * nVx + f.ox];
    }
    if (gm) {
        pgy = a.gty[eb * nVy + f.oy];
        pgx = a.gtx[eb * nVx + f.ox];
        if (f.ey >= 0) pey = a.gty[eb * nVy + f.ey] - a.cvy[eb * nVy + f.ey];      // the faces without a correction only enter the loss
        if (f.ex >= 0) pex = a.gtx[eb * nVx + f.ex] - a.cvx[eb * nVx + f.ex];
    }
    // pull the weight lines of this wave's unit into the L2 while the rows are in flight (one dword per 64-byte line and lane): the L2 is
    // invalid at kernel start, and the unit loop below has to wait for every scalar load together with its LDS reads (both count in
    // lgkmcnt) -- ten exposed round trips per unit, each to the MALL otherwise.  (A plain vector load: the value only keeps the request alive.)
    float wwarm = 0.f;
    if (wave < 15 && lane < 5 * CON * 2)         // (NW = 8: the wave's first unit; its second one is warmed by the wave that has the same tap row)
        wwarm = a.wp[(size_t)(wave / 3) * 5 * 16 * 32 + ((lane / (CON * 2)) * 16 + (lane / 2) % CON) * 32 + (lane & 1) * 16];
    __builtin_amdgcn_sched_barrier(0);
    if (tid < 224) {                                             // zero halo pixels 0, 1, 66, 67 of every row
        const int s = tid >> 5, q = tid & 31, hp = q >> 3, px = hp < 2 ? hp : hp + 64;
        *reinterpret_cast<float4*>(rows + s * TH_ROW + px * TH_PITCH + (q & 7) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {
        float4* q = reinterpret_cast<float4*>(rows + (p + 2) * TH_PITCH + c4 * 16);
        constexpr int RS = TH_ROW / 16;
        if (NW == 16) {
            q[hb * RS] = v0; q[(hb + 2) * RS] = v1; q[(hb + 4) * RS] = v2;
            if (hb == 0) q[6 * RS] = v3;
        } else {
            q[0] = v0; q[RS] = v1; q[2 * RS] = v2; q[3 * RS] = v3; q[4 * RS] = v4; q[5 * RS] = v5; q[6 * RS] = v6;
        }
    }
    asm volatile("" :: "v"(wwarm));
    __syncthreads();

    // ---- the wave's units (output row r, tap row dy) ----------------------------------------------------------------------------------------
#pragma unroll 1
    for (int u = wave; u < 15; u += NW) {
        const int r = u % 3, dy = u / 3;
        const int oy = G0 + r, iy = oy + dy - 2;
        const int b = oy < nrows ? oy / H : 0;
        const bool valid = oy < nrows && iy >= b * H && iy < (b + 1) * H;      // scalar: input row in the image of the output row
        f32x2 acc[CON];
#pragma unroll
        for (int c = 0; c < CON; ++c) acc[c] = (f32x2){0.f, 0.f};
        if (valid) {
            const unsigned char* rb = rows + (r + dy) * TH_ROW + lane * TH_PITCH;
            th_cfp wt = (th_cfp)(a.wp + (size_t)dy * 5 * 16 * 32);      // packed fp32 weights [tap][16 co][32 ci]: wave uniform -> scalar loads
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
#pragma unroll
                for (int cq = 0; cq < 8; ++cq) {
                    const float4 x = *reinterpret_cast<const float4*>(rb + dx * TH_PITCH + cq * 16);
                    const f32x2 x01 = {x.x, x.y}, x23 = {x.z, x.w};
#pragma unroll
                    for (int c = 0; c < CON; ++c) {
                        th_cfp wc = wt + (dx * 16 + c) * 32 + cq * 4;
                        const f32x2 w01 = {wc[0], wc[1]}, w23 = {wc[2], wc[3]};
                        acc[c] = __builtin_elementwise_fma(x01, w01, acc[c]);
                        acc[c] = __builtin_elementwise_fma(x23, w23, acc[c]);
                    }
                }
            }
        }
        float4 o = make_float4(acc[0].x + acc[0].y, acc[1].x + acc[1].y, 0.f, 0.f);
        if (CON == 4) { o.z = acc[CON - 2].x + acc[CON - 2].y; o.w = acc[CON - 1].x + acc[CON - 1].y; }
        *reinterpret_cast<float4*>(part + (u * 64 + lane) * 4) = o;
    }
    __syncthreads();

    // ---- fold the five tap rows (fixed order) and the epilogue: threads 0..191 = (row r = wave, pixel = lane) ------------------------------
    float lsum = 0.f, vmax = 0.f;
    if (ep) {
        float4 o = *reinterpret_cast<const float4*>(part + ((0 * 3 + er) * 64 + lane) * 4);
#pragma unroll
        for (int dy = 1; dy < 5; ++dy) {
            const float4 q = *reinterpret_cast<const float4*>(part + ((dy * 3 + er) * 64 + lane) * 4);
            o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
        }
        if (a.bias) {                                            // (a caller's bias has a.CO entries: no read beyond them)
            o.x += a.bias[0];
            if (a.CO > 1) o.y += a.bias[1];
            if (CON == 4 && a.CO > 2) o.z += a.bias[2];
            if (CON == 4 && a.CO > 3) o.w += a.bias[3];
        }
        if (cm) {
            // correction mode: the output is applied to the staggered velocity and never stored (karman_train.py:88-90, 424-426)
            const float v0 = pvy + a.cs0 * o.x, v1 = pvx + a.cs1 * o.y;
            a.cvy[eb * nVy + f.oy] = v0;
            a.cvx[eb * nVx + f.ox] = v1;
            if (gm) {
                const float d0 = (pgy - v0) / a.ls0, d1 = (pgx - v1) / a.ls1;
                lsum = 0.5f * d0 * d0 + 0.5f * d1 * d1;
                if (f.ey >= 0) { const float d2 = pey / a.ls0; lsum += 0.5f * d2 * d2; }
                if (f.ex >= 0) { const float d2 = pex / a.ls1; lsum += 0.5f * d2 * d2; }
            }
        } else {
            const float oc[4] = {o.x, o.y, o.z, o.w};
            float* yp = a.y + ((size_t)gy * W + lane) * a.CO;
#pragma unroll
            for (int c = 0; c < CON; ++c)
                if (c < a.CO) { yp[c] = oc[c]; vmax = fmaxf(vmax, fabsf(oc[c])); }
        }
    }
    // workgroup uniform: wave sums -> LDS slots -> the last wave's fixed-order sum -> ONE exact integer add per workgroup (loss_add_exact)
    if (a.cvy && a.closs) loss_publish_last(lsum, a.closs, slots + 2);
    if (a.ymax) amax_publish_last(vmax, a.ymax, slots);
}

}  // namespace

// Which launches take this kernel: 32 input channels, <= 4 output channels, images of exactly 64 pixels per row, no residual, no
// activation (the two thin layers of the trainer), option conv_thin_valu (default 1).  Everything else stays on k_conv5x5_sb<1, KIND> /
// the thin form of the dx kernel.
bool sol_conv_thin32_usable(const ConvArgs& a, int NT) {
    // a.CI == 32: the kernel reads eight float4 per pixel and a [tap][16][32] fp32 weight section unconditionally (a caller with another
    // channel count -- none exists today: sol_conv_sb_launch is reached with 32 input channels only -- must not end up here)
    return sol_opt().conv_thin_valu && NT == 1 && a.CI == 32 && a.CO >= 1 && a.CO <= 4 && a.W == 64 && a.tiles_x == 1 && !a.res && a.epi == SOL_EPI_NONE && a.wp &&
           (!a.cvy || a.CO == 2);
}

int sol_conv_thin32_launch(hipStream_t s, const ConvArgs& a, int ntiles) {
    SOL_REQUIRE(a.CI == 32 && a.W == 64 && a.CO >= 1 && a.CO <= 4 && a.wp, "k_conv5x5_thin32: 32 input channels, 64-pixel rows, <= 4 output channels (got %d, %d, %d)", a.CI, a.W, a.CO);
    static std::atomic<unsigned long long> optin{0};
    if (int e = sol_lds_optin(optin, {SOL_K((k_conv5x5_thin32<2, 16>)), SOL_K((k_conv5x5_thin32<4, 16>)), SOL_K((k_conv5x5_thin32<2, 8>)), SOL_K((k_conv5x5_thin32<4, 8>))}, "k_conv5x5_thin32")) return e;
    const int nrows = ntiles;                             // tiles_x == 1: one tile per image row
    int grid = (nrows + 2) / 3;
    if (grid > 64) grid = (grid + 7) / 8 * 8;             // XCD-aware tile order (xcd_tile); padding workgroups own no rows
    const bool w8 = sol_opt().conv_thin_valu == 2;     // (A/B: eight waves, two units each)
    if (a.CO <= 2) {
        if (w8) SOL_LAUNCH((k_conv5x5_thin32<2, 8>), dim3(grid), dim3(512), TH_LDS, s, a, nrows);
        else SOL_LAUNCH((k_conv5x5_thin32<2, 16>), dim3(grid), dim3(1024), TH_LDS, s, a, nrows);
    } else {
        if (w8) SOL_LAUNCH((k_conv5x5_thin32<4, 8>), dim3(grid), dim3(512), TH_LDS, s, a, nrows);
        else SOL_LAUNCH((k_conv5x5_thin32<4, 16>), dim3(grid), dim3(1024), TH_LDS, s, a, nrows);
    }
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}
