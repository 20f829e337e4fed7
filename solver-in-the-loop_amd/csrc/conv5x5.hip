// 5x5 SAME convolution (NHWC, fp32) on the gfx950 fp32 matrix cores, forward / backward-data
// (same kernel, differently packed weights) and backward-weight.
//
// Replaces keras.layers.Conv2D(filters, 5, padding='same') (+bias, LeakyReLU, residual add)
// of model_mars_moon (/root/reference/karman-2d/karman_train.py:101-138) and TF's conv
// gradients.  fp32 in / fp32 accumulate (v_mfma_f32_16x16x4_f32): plain bf16 would break the 1e-5
// parity bar of the solver-in-the-loop loss.  The 32-channel, W % 64 == 0 shapes run on the
// fp32-equivalent split-bf16 kernels of conv5x5_sb.hip by default; the kernels here serve the
// other shapes (and SOL_CONV_NO_SB=1).
//
// Implicit GEMM, per workgroup: M = 64 output pixels (4 waves x 16), N = 16*NT output
// channels, K = 25 taps x CIN.  The input halo tile is staged once in LDS (pixel stride
// padded for conflict-free ds_read_b128), the weights stream from L1/L2 in the packed
// [tap][cout][cin] layout so that one float4 load yields 4 K-steps of the B operand.
// MFMA operand maps (cdna guide section 3):  A: lane l -> A[i = l&15][k = l>>4],
// B: lane l -> B[k = l>>4][j = l&15],  C/D: lane l, reg r -> C[row = 4*(l>>4)+r][col = l&15].
// The K index inside one MFMA is permuted (k-step kk of lane group g uses channel 8g+kk);
// A and B use the same permutation so the product is unchanged.
#include "common.hpp"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int pad_in(int c) { return c <= 4 ? 4 : 32; }       // K-side channel padding
__host__ __device__ inline int pad_out(int c) { return c <= 16 ? 16 : 32; }    // N-side channel padding

// ------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------
// FWD:      packed[tap][o][i] = w[tap][i][o]            (o = cout, i = cin)
// BWD_DATA: packed[tap][o][i] = w[24-tap][o][i]         (o = forward cin, i = forward cout)
// `cin`/`cout` are the channel counts of the convolution being RUN (for BWD_DATA: cin =
// channels of dy = forward cout).
__global__ void k_pack(const float* __restrict__ w, float* __restrict__ packed, int cin, int cout, int mode) {
    const int IP = pad_in(cin), OP = pad_out(cout);
    const int total = 25 * OP * IP;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int i = e % IP, o = (e / IP) % OP, tap = e / (IP * OP);
        float v = 0.f;
        if (i < cin && o < cout) {
            if (mode == SOL_CONV_FWD) v = w[(tap * cin + i) * cout + o];      // HWIO, I = cin, O = cout
            else v = w[((24 - tap) * cout + o) * cin + i];                     // HWIO, I = cout(run), O = cin(run)
        }
        packed[e] = v;
    }
}

// ------------------------------------------------------------------------------------
// forward / backward-data kernel
// ------------------------------------------------------------------------------------
// ConvArgs: common.hpp
template <int CIN, int NT>
__global__ void __launch_bounds__(256) k_conv5x5(ConvArgs a) {
    constexpr int CP = CIN == 4 ? 4 : 36;   // LDS pixel stride in floats
    constexpr int OP = NT * 16;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H, W = a.W, TW = a.TW, RPW = a.RPW;
    const int HW = TW + 4;   // halo width in pixels
    int blk = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = blk % a.tiles_x; blk /= a.tiles_x;
    const int rows_blk = H / RPW;
    const int ty = blk % rows_blk;
    const int b = blk / rows_blk;
    const int y0 = ty * RPW, x0 = tx * TW;

    // ---- stage the zero padded halo tile -------------------------------------------------
    {
        constexpr int C4 = CIN / 4;
        const int npix = (RPW + 4) * HW;
        const float4* gx = reinterpret_cast<const float4*>(a.x);
        if (CIN == 4 && a.svy) {
            // seed mode (see ConvArgs): RPW == 1, TW == W == 64.  Same arithmetic, operation for operation, as k_seed
            const float l00 = a.ls0 * a.ls0, l11 = a.ls1 * a.ls1;
            const size_t by = (size_t)b * (H + 1) * W, bx = (size_t)b * H * (W + 1);
            for (int pix = tid; pix < npix; pix += 256) {
                const int hr = pix / HW, hc = pix - hr * HW;
                const int yy = y0 + hr - 2, xx = hc - 2;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const size_t ky = by + (size_t)yy * W + xx, kx = bx + (size_t)yy * (W + 1) + xx;
                    float gy = (a.svy[ky] - a.gty[ky]) * a.sinv_m / l00, gxx = (a.svx[kx] - a.gtx[kx]) * a.sinv_m / l11;
                    if (a.sginy) { gy += a.sginy[ky]; gxx += a.sginx[kx]; }
                    v.x = a.cs0 * gy; v.y = a.cs1 * gxx;
                    if (hr == 2) {                       // this workgroup's own row: publish G and dO2
                        a.sgy[ky] = gy; a.sgx[kx] = gxx;
                        *reinterpret_cast<float2*>(a.sdO2 + ((size_t)(b * H + yy) * W + xx) * 2) = make_float2(v.x, v.y);
                    }
                }
                *reinterpret_cast<float4*>(&smem[pix * CP]) = v;
            }
            // the faces without a cell of their own: v_x column X of this row, v_y row Y (by the workgroup of the last image row)
            if (tid == 0) {
                const size_t kx = bx + (size_t)y0 * (W + 1) + W;
                float g = (a.svx[kx] - a.gtx[kx]) * a.sinv_m / l11;
                if (a.sginx) g += a.sginx[kx];
                a.sgx[kx] = g;
            }
            if (y0 == H - 1 && tid >= 64 && tid < 64 + W) {
                const size_t ky = by + (size_t)H * W + (tid - 64);
                float g = (a.svy[ky] - a.gty[ky]) * a.sinv_m / l00;
                if (a.sginy) g += a.sginy[ky];
                a.sgy[ky] = g;
            }
        } else
        for (int e = tid; e < npix * C4; e += 256) {
            const int pix = e / C4, c4 = e - pix * C4;
            const int hr = pix / HW, hc = pix - hr * HW;
            const int yy = y0 + hr - 2, xx = x0 + hc - 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = gx[((size_t)(b * H + yy) * W + xx) * C4 + c4];
            *reinterpret_cast<float4*>(&smem[pix * CP + c4 * 4]) = v;
        }
        // thin layers: the 25 x OP x 4 weight block goes through LDS with 16-byte loads.  (Per-lane dword loads of the B
        // operands -- 50 per lane, the same 12.8 KB in all 3072 waves -- kept the texture path busy for 6.5 us of a 12.3 us launch.)
        if constexpr (CIN == 4) {
            const float4* gw = reinterpret_cast<const float4*>(a.wp);
            float* wl = smem + npix * CP;
            for (int e = tid; e < 25 * OP; e += 256) *reinterpret_cast<float4*>(&wl[e * 4]) = gw[e];
        }
    }
    __syncthreads();

    // ---- implicit GEMM ------------------------------------------------------------------
    const int q = wave * 16 + li;            // this lane's A-row pixel inside the tile
    const int prr = q / TW, pcc = q - prr * TW;
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (CIN == 32) {
        const float* abase = &smem[(prr * HW + pcc) * CP + 8 * g];
        const float* wbase = a.wp + (size_t)li * 32 + 8 * g;
#pragma unroll 1
        for (int dy = 0; dy < 5; ++dy) {
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                const int tap = dy * 5 + dx;
                const float* ap = abase + (dy * HW + dx) * CP;
                const float4 a0 = *reinterpret_cast<const float4*>(ap);
                const float4 a1 = *reinterpret_cast<const float4*>(ap + 4);
                const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float* wp = wbase + ((size_t)tap * OP + n * 16) * 32;
                    const float4 b0 = *reinterpret_cast<const float4*>(wp);
                    const float4 b1 = *reinterpret_cast<const float4*>(wp + 4);
                    const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk)
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv[kk], acc[n], 0, 0, 0);
                }
            }
        }
    } else {   // CIN == 4: one MFMA per tap, lane group g = channel
        const float* abase = &smem[(prr * HW + pcc) * CP + g];
        const float* wl = smem + (RPW + 4) * HW * CP + li * 4 + g;
#pragma unroll
        for (int tap = 0; tap < 25; ++tap) {
            const int dy = tap / 5, dx = tap - dy * 5;
            const float av = abase[(dy * HW + dx) * CP];
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wl[(tap * OP + n * 16) * 4], acc[n], 0, 0, 0);
        }
    }

    // ---- epilogue: bias, residual, LeakyReLU / LeakyReLU' ---------------------------------
    float vmax = 0.f;                                // max|y| of this thread (a.ymax)
    if (CIN == 4 && a.CO == OP && TW == 64) {
        // thin first layer / last backward-data layer (4 -> 32 channels): the launch is bound by its 6 MB of output, so the
        // wave's [16 px][OP] tile is transposed through LDS and leaves as 16-byte pieces of full 128-byte pixels
        __syncthreads();                             // every wave is done with the halo tile
        float* tb = smem + wave * (16 * OP);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float bias = a.bias ? a.bias[n * 16 + li] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) tb[(4 * g + r) * OP + n * 16 + li] = acc[n][r] + bias;
        }
        constexpr int F4 = 16 * OP / 4 / 64;          // float4 per lane
#pragma unroll
        for (int n = 0; n < F4; ++n) {
            const int e = lane + n * 64;
            const int px = e / (OP / 4), c4 = e % (OP / 4);
            float4 v = *reinterpret_cast<const float4*>(&tb[px * OP + c4 * 4]);
            const size_t o4 = ((size_t)(b * H + y0) * W + x0 + wave * 16 + px) * (OP / 4) + c4;
            if (a.res) { const float4 q = reinterpret_cast<const float4*>(a.res)[o4]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
            if (a.epi == SOL_EPI_LRELU) {
                v.x = v.x > 0.f ? v.x : a.slope * v.x; v.y = v.y > 0.f ? v.y : a.slope * v.y;
                v.z = v.z > 0.f ? v.z : a.slope * v.z; v.w = v.w > 0.f ? v.w : a.slope * v.w;
            } else if (a.epi == SOL_EPI_DLRELU) {
                const float4 q = reinterpret_cast<const float4*>(a.act)[o4];
                v.x *= q.x > 0.f ? 1.f : a.slope; v.y *= q.y > 0.f ? 1.f : a.slope;
                v.z *= q.z > 0.f ? 1.f : a.slope; v.w *= q.w > 0.f ? 1.f : a.slope;
            }
            vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            st_wt(reinterpret_cast<float4*>(a.y) + o4, v);
        }
    } else {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = n * 16 + li;
            if (co >= a.CO) continue;
            const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = wave * 16 + 4 * g + r;
                const int rr = qq / TW, cc = qq - rr * TW;
                const size_t o = ((size_t)(b * H + y0 + rr) * W + x0 + cc) * a.CO + co;
                float v = acc[n][r] + bias;
                if (a.res) v += a.res[o];
                if (a.epi == SOL_EPI_LRELU) v = v > 0.f ? v : a.slope * v;
                else if (a.epi == SOL_EPI_DLRELU) v *= (a.act[o] > 0.f ? 1.f : a.slope);
                vmax = fmaxf(vmax, fabsf(v));
                a.y[o] = v;
            }
        }
    }
    if (a.ymax) {                                    // workgroup uniform
        __syncthreads();
        amax_publish(vmax, a.ymax, smem);
    }
}

// ------------------------------------------------------------------------------------
// thin-INPUT layer (4 -> 16 NT channels, 64-pixel rows), THREE image rows per workgroup (round 6)
// ------------------------------------------------------------------------------------
// The trainer's first layer (3 -> 32) and its last backward-data layer (2 -> 32, seed mode): 0.24 GFLOP per launch, pure launch latency.  As
// k_conv5x5<4, NT> they ran one image row per 256-thread workgroup: 768 workgroups (three per CU), each staging FIVE halo rows and the
// whole 12.8 KB weight block for one row of output -- and, in seed mode, evaluating the loss gradient of every pixel five times.  Here a
// workgroup of twelve waves owns three consecutive rows of the (batch x height) row stack (the dx kernel's tiling: 256 workgroups, one per
// CU): seven halo rows, one weight block, wave = (row, 16-pixel segment).  Rows of the stack that belong to another image than the output row
// (the stack is cut anywhere, also between two images) contribute zeros: the test is wave uniform, one v_cndmask per MFMA operand.
// Arithmetic and summation order per output pixel are k_conv5x5<4, NT>'s (same MFMA sequence over the taps): bit-identical results.
template <int NT>
__global__ void __launch_bounds__(768) k_conv5x5_t3(ConvArgs a) {
    constexpr int CP = 4, OP = NT * 16, W = 64, HW = W + 4, NR = 3, NH = NR + 4, NTHR = 768;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H, R = a.B * H;
    const int r0 = xcd_tile(blockIdx.x, gridDim.x) * NR;          // first row of the stack owned by this workgroup
    // the epilogue's operands (residual, activation reference of LeakyReLU') are requested FIRST: they are HBM-cold and nothing but the
    // launch's own latency chain (stage -> 50 MFMAs -> transpose -> store) is there to cover their round trip
    constexpr int F4 = 16 * OP / 4 / 64;              // float4 per lane of the wave's [16 px][OP] output tile
    float4 eres[F4], eact[F4];
    {
        const int gro_ = min(r0 + (wave >> 2), R - 1), seg_ = wave & 3;          // (a stack that is no multiple of three: the last workgroup's surplus rows load row R-1 and store nothing)
#pragma unroll
        for (int n = 0; n < F4; ++n) {
            const int e = lane + n * 64, px = e / (OP / 4), c4 = e % (OP / 4);
            const size_t o4 = ((size_t)gro_ * W + seg_ * 16 + px) * (OP / 4) + c4;
            eres[n] = a.res ? reinterpret_cast<const float4*>(a.res)[o4] : make_float4(0.f, 0.f, 0.f, 0.f);
            eact[n] = a.epi == SOL_EPI_DLRELU ? reinterpret_cast<const float4*>(a.act)[o4] : make_float4(1.f, 1.f, 1.f, 1.f);
        }
    }

    // ---- stage the seven halo rows (zero outside the stack / outside the row) and the weight block ----
    if (a.svy) {
        // seed mode (ConvArgs): the halo pixel IS the loss gradient, same arithmetic, operation for operation, as k_seed / k_conv5x5<4, 2>
        const float l00 = a.ls0 * a.ls0, l11 = a.ls1 * a.ls1;
        for (int pix = tid; pix < NH * HW; pix += NTHR) {
            const int hr = pix / HW, hc = pix - hr * HW;
            const int gr = r0 + hr - 2, xx = hc - 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr >= 0 && gr < R && xx >= 0 && xx < W) {
                const int b = gr / H, yy = gr - b * H;
                const size_t ky = (size_t)b * (H + 1) * W + (size_t)yy * W + xx, kx = (size_t)b * H * (W + 1) + (size_t)yy * (W + 1) + xx;
                float gy = (a.svy[ky] - a.gty[ky]) * a.sinv_m / l00, gxx = (a.svx[kx] - a.gtx[kx]) * a.sinv_m / l11;
                if (a.sginy) { gy += a.sginy[ky]; gxx += a.sginx[kx]; }
                v.x = a.cs0 * gy; v.y = a.cs1 * gxx;
                if (hr >= 2 && hr < 2 + NR) {            // the workgroup's own rows: publish G and dO2
                    a.sgy[ky] = gy; a.sgx[kx] = gxx;
                    *reinterpret_cast<float2*>(a.sdO2 + ((size_t)gr * W + xx) * 2) = make_float2(v.x, v.y);
                }
            }
            *reinterpret_cast<float4*>(&smem[pix * CP]) = v;
        }
        // the faces without a cell of their own: v_x column X of every own row, v_y row Y behind the last row of an image
        if (tid < NR && r0 + tid < R) {
            const int gr = r0 + tid, b = gr / H, yy = gr - b * H;
            const size_t kx = (size_t)b * H * (W + 1) + (size_t)yy * (W + 1) + W;
            float gq = (a.svx[kx] - a.gtx[kx]) * a.sinv_m / l11;
            if (a.sginx) gq += a.sginx[kx];
            a.sgx[kx] = gq;
        }
        if (tid >= 64 && tid < 64 + NR * W) {
            const int o = (tid - 64) / W, xx = (tid - 64) - o * W, gr = r0 + o, b = gr / H, yy = gr - b * H;
            if (yy == H - 1 && gr < R) {
                const size_t ky = (size_t)b * (H + 1) * W + (size_t)H * W + xx;
                float gq = (a.svy[ky] - a.gty[ky]) * a.sinv_m / l00;
                if (a.sginy) gq += a.sginy[ky];
                a.sgy[ky] = gq;
            }
        }
    } else {
        const float4* gx = reinterpret_cast<const float4*>(a.x);
        for (int pix = tid; pix < NH * HW; pix += NTHR) {
            const int hr = pix / HW, hc = pix - hr * HW;
            const int gr = r0 + hr - 2, xx = hc - 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr >= 0 && gr < R && xx >= 0 && xx < W) v = gx[(size_t)gr * W + xx];
            *reinterpret_cast<float4*>(&smem[pix * CP]) = v;
        }
    }
    {
        const float4* gw = reinterpret_cast<const float4*>(a.wp);
        float* wl = smem + NH * HW * CP;
        for (int e = tid; e < 25 * OP; e += NTHR) *reinterpret_cast<float4*>(&wl[e * 4]) = gw[e];
    }
    __syncthreads();

    // ---- implicit GEMM: wave = (output row, 16-pixel segment), one MFMA per tap and channel tile, lane group g = input channel ----
    const int orow = wave >> 2, seg = wave & 3;
    const int gro = r0 + orow, yo = gro % H;
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        const float* abase = &smem[(orow * HW + seg * 16 + li) * CP + g];
        const float* wl = smem + NH * HW * CP + li * 4 + g;
#pragma unroll
        for (int tap = 0; tap < 25; ++tap) {
            const int dy = tap / 5, dx = tap - dy * 5;
            const bool ok = yo + dy - 2 >= 0 && yo + dy - 2 < H;       // the tap's row lies in the output row's image (wave uniform)
            const float av = ok ? abase[(dy * HW + dx) * CP] : 0.f;
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wl[(tap * OP + n * 16) * 4], acc[n], 0, 0, 0);
        }
    }

    // ---- epilogue: bias, residual, LeakyReLU / LeakyReLU'; the wave's [16 px][OP] tile leaves as 16-byte pieces of full pixels ----
    float vmax = 0.f;
    __syncthreads();                                 // every wave is done with the halo tile
    float* tb = smem + wave * (16 * OP);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const float bias = a.bias ? a.bias[n * 16 + li] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[(4 * g + r) * OP + n * 16 + li] = acc[n][r] + bias;
    }
#pragma unroll
    for (int n = 0; n < F4; ++n) {
        const int e = lane + n * 64;
        const int px = e / (OP / 4), c4 = e % (OP / 4);
        float4 v = *reinterpret_cast<const float4*>(&tb[px * OP + c4 * 4]);
        const size_t o4 = ((size_t)gro * W + seg * 16 + px) * (OP / 4) + c4;
        if (a.res) { const float4 q = eres[n]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
        if (a.epi == SOL_EPI_LRELU) {
            v.x = v.x > 0.f ? v.x : a.slope * v.x; v.y = v.y > 0.f ? v.y : a.slope * v.y;
            v.z = v.z > 0.f ? v.z : a.slope * v.z; v.w = v.w > 0.f ? v.w : a.slope * v.w;
        } else if (a.epi == SOL_EPI_DLRELU) {
            const float4 q = eact[n];
            v.x *= q.x > 0.f ? 1.f : a.slope; v.y *= q.y > 0.f ? 1.f : a.slope;
            v.z *= q.z > 0.f ? 1.f : a.slope; v.w *= q.w > 0.f ? 1.f : a.slope;
        }
        if (gro < R) {
            vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            st_wt(reinterpret_cast<float4*>(a.y) + o4, v);
        }
    }
    if (a.ymax) {                                    // workgroup uniform
        __syncthreads();
        amax_publish(vmax, a.ymax, smem);
    }
}
// usable where the launch is a SMALL thin-input layer of 64-pixel rows (any number of rows: the last workgroup's surplus rows store nothing)
static bool conv_t3_usable(int B, int H, int W, int cin, int cout_padded) {
    // small launches only (<= 4 workgroups per CU in the one-row form): at 8192 rows (a Conv3D tap pass at 128 x 64 x 64) the one-row kernel is
    // the faster one, 45.3 vs 50.8 us per launch (tools/k3d_time.py)
    return sol_opt().conv_thin_t3 && cin == 4 && W == 64 && (cout_padded == 32 || cout_padded == 16) && B * H <= 1024;
}
static size_t conv_t3_lds(int OP) {
    size_t lds = ((size_t)7 * 68 * 4 + (size_t)25 * OP * 4) * sizeof(float);
    const size_t epi = (size_t)12 * 16 * OP * sizeof(float);
    return lds < epi ? epi : lds;
}

// ------------------------------------------------------------------------------------
// 32-input-channel kernel (the 22 FLOP-dominant launches per sim-step): software pipelined.
//   * halo tile UNPADDED with an XOR swizzle of the 16-byte channel chunks
//     (chunk ^= ((pix >> 1) & 3) << 1) -> conflict-free ds_read_b128 for the A operand, and
//     43.5 KB instead of 49 KB so that three workgroups + their weight buffers fit one CU;
//   * the per-tap weight tile [cout][32] is staged ONCE per workgroup in LDS (double buffered,
//     same swizzle) instead of being re-read from L1/L2 by each of the 4 waves;
//   * the loads of tap t+1 (weights) and of the next halo row are issued before the MFMAs of
//     tap t and written to LDS after them: one barrier per tap, the first MFMA starts after
//     one halo row instead of the whole tile.
// Lane group g consumes channel chunks g and g+4 (K permutation shared by A and B).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int swz(int pix) { return ((pix >> 1) & 3) << 1; }

template <int NT>
__global__ void __launch_bounds__(256) k_conv5x5_c32(ConvArgs a) {
    constexpr int OP = NT * 16;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H, W = a.W, TW = a.TW, RPW = a.RPW;
    const int HW = TW + 4, NR = RPW + 4;
    int blk = blockIdx.x;
    const int tx = blk % a.tiles_x; blk /= a.tiles_x;
    const int rows_blk = H / RPW;
    const int ty = blk % rows_blk;
    const int b = blk / rows_blk;
    const int y0 = ty * RPW, x0 = tx * TW;
    float* halo = smem;                      // [NR*HW][32] swizzled
    float* Bs = smem + NR * HW * 32;         // [2][OP][32] swizzled
    const float4* gx = reinterpret_cast<const float4*>(a.x);
    const float4* gw = reinterpret_cast<const float4*>(a.wp);
    const int row_f4 = HW * 8;               // float4 per halo row
    constexpr int HPT = 3;                   // float4 per thread per halo row (HW <= 68 -> 544 <= 768)

    auto load_row = [&](int hr, float4 (&v)[HPT]) {
        const int yy = y0 + hr - 2;
#pragma unroll
        for (int n = 0; n < HPT; ++n) {
            const int e = tid + n * 256;
            v[n] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < row_f4) {
                const int hc = e >> 3, c4 = e & 7;
                const int xx = x0 + hc - 2;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) v[n] = gx[((size_t)(b * H + yy) * W + xx) * 8 + c4];
            }
        }
    };
    auto store_row = [&](int hr, const float4 (&v)[HPT]) {
#pragma unroll
        for (int n = 0; n < HPT; ++n) {
            const int e = tid + n * 256;
            if (e < row_f4) {
                const int hc = e >> 3, c4 = e & 7;
                const int pix = hr * HW + hc;
                *reinterpret_cast<float4*>(&halo[pix * 32 + ((c4 ^ swz(pix)) << 2)]) = v[n];
            }
        }
    };
    const int bco = tid >> 3, bc4 = tid & 7;    // weight tile element of this thread
    const bool bact = tid < OP * 8;
    float* bdst = Bs + bco * 32 + ((bc4 ^ swz(bco)) << 2);

    // ---- prologue: halo rows [0, RPW) and the weights of tap 0 -----------------------------
    for (int hr = 0; hr < RPW; ++hr) {
        float4 v[HPT];
        load_row(hr, v);
        store_row(hr, v);
    }
    if (bact) *reinterpret_cast<float4*>(bdst) = gw[(size_t)bco * 8 + bc4];
    __syncthreads();

    const int q = wave * 16 + li;            // this lane's A-row pixel inside the tile
    const int prr = q / TW, pcc = q - prr * TW;
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int dy = 0; dy < 5; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
            const int tap = dy * 5 + dx;
            // issue the prefetches for the next tap / next halo row
            float4 bnext = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool more = tap + 1 < 25;
            if (more && bact) bnext = gw[((size_t)(tap + 1) * OP + bco) * 8 + bc4];
            float4 hv[HPT];
            const bool hrow = dx == 0 && dy < 4;
            if (hrow) load_row(RPW + dy, hv);
            // MFMAs of this tap
            const int pix = (prr + dy) * HW + pcc + dx;
            const float* ap = &halo[pix * 32];
            const int sa = swz(pix);
            const float4 a0 = *reinterpret_cast<const float4*>(ap + ((g ^ sa) << 2));
            const float4 a1 = *reinterpret_cast<const float4*>(ap + (((g + 4) ^ sa) << 2));
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const float* bbuf = Bs + (tap & 1) * OP * 32;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co = n * 16 + li;
                const float* bp = bbuf + co * 32;
                const int sb = swz(co);
                const float4 b0 = *reinterpret_cast<const float4*>(bp + ((g ^ sb) << 2));
                const float4 b1 = *reinterpret_cast<const float4*>(bp + (((g + 4) ^ sb) << 2));
                const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv[kk], acc[n], 0, 0, 0);
            }
            // publish the prefetched data for the next tap
            if (more && bact) *reinterpret_cast<float4*>(bdst + ((tap + 1) & 1) * OP * 32) = bnext;
            if (hrow) store_row(RPW + dy, hv);
            __syncthreads();
        }
    }

    // ---- epilogue: bias, residual, LeakyReLU / LeakyReLU' ---------------------------------
    float vmax = 0.f;                                // max|y| of this thread (a.ymax)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 16 + li;
        if (co >= a.CO) continue;
        const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = wave * 16 + 4 * g + r;
            const int rr = qq / TW, cc = qq - rr * TW;
            const size_t o = ((size_t)(b * H + y0 + rr) * W + x0 + cc) * a.CO + co;
            float v = acc[n][r] + bias;
            if (a.res) v += a.res[o];
            if (a.epi == SOL_EPI_LRELU) v = v > 0.f ? v : a.slope * v;
            else if (a.epi == SOL_EPI_DLRELU) v *= (a.act[o] > 0.f ? 1.f : a.slope);
            vmax = fmaxf(vmax, fabsf(v));
            a.y[o] = v;
        }
    }
    if (a.ymax) {                                    // workgroup uniform
        __syncthreads();
        amax_publish(vmax, a.ymax, smem);
    }
}

// ------------------------------------------------------------------------------------
// W % 64 == 0 variant: three independent 64-pixel row tiles per workgroup (768 threads = 12
// waves = 3 per SIMD, one workgroup per CU at C3) that SHARE the staged weights.  A row tile
// needs exactly one halo row per tap row dy, so the halo is a 2-slot ring (17 KB per tile) and
// the weights of a whole tap row (5 taps, 20 KB) are double buffered: ONE barrier per 5 taps
// (5 per launch instead of 25) and 80 back-to-back MFMAs per wave between barriers.
// ------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(768) k_conv5x5_r3(ConvArgs a, int ntiles) {
    constexpr int OP = NT * 16;
    constexpr int HWP = 68;                       // halo pixels per row (64 + 4)
    constexpr int SLOT = HWP * 32;                // floats per halo row
    constexpr int WBUF = 5 * OP * 32;             // floats per weight phase
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, grp = tid >> 8, t = tid & 255, lane = tid & 63, wave = (tid >> 6) & 3;
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H, W = a.W;
    const int tile = blockIdx.x * 3 + grp;
    const bool tvalid = tile < ntiles;
    const int tx = tvalid ? tile % a.tiles_x : 0;
    const int gy = tvalid ? tile / a.tiles_x : 0;     // global row index b*H + y
    const int b = gy / H, y = gy - b * H, x0 = tx * 64;
    float* halo = smem + grp * 2 * SLOT;              // [2][68][32] swizzled, private to the tile
    float* Wt = smem + 3 * 2 * SLOT;                  // [2][5][OP][32] swizzled, shared
    const float4* gx = reinterpret_cast<const float4*>(a.x);
    const float4* gw = reinterpret_cast<const float4*>(a.wp);
    constexpr int HPT = 3;                            // 544 float4 per halo row / 256 threads
    constexpr int WPT = (5 * OP * 8 + 767) / 768;     // float4 per thread per weight phase

    auto load_row = [&](int dy, float4 (&v)[HPT]) {
        const int yy = y + dy - 2;
#pragma unroll
        for (int n = 0; n < HPT; ++n) {
            const int e = t + n * 256;
            v[n] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < HWP * 8) {
                const int hc = e >> 3, c4 = e & 7, xx = x0 + hc - 2;
                if (tvalid && yy >= 0 && yy < H && xx >= 0 && xx < W) v[n] = gx[((size_t)(b * H + yy) * W + xx) * 8 + c4];
            }
        }
    };
    auto store_row = [&](int slot, const float4 (&v)[HPT]) {
        float* dst = halo + slot * SLOT;
#pragma unroll
        for (int n = 0; n < HPT; ++n) {
            const int e = t + n * 256;
            if (e < HWP * 8) {
                const int hc = e >> 3, c4 = e & 7;
                *reinterpret_cast<float4*>(&dst[hc * 32 + ((c4 ^ swz(hc)) << 2)]) = v[n];
            }
        }
    };
    auto load_w = [&](int dy, float4 (&v)[WPT]) {
#pragma unroll
        for (int n = 0; n < WPT; ++n) {
            const int e = tid + n * 768;
            v[n] = e < 5 * OP * 8 ? gw[(size_t)dy * (5 * OP * 8) + e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_w = [&](int buf, const float4 (&v)[WPT]) {
        float* dst = Wt + buf * WBUF;
#pragma unroll
        for (int n = 0; n < WPT; ++n) {
            const int e = tid + n * 768;
            if (e < 5 * OP * 8) {
                const int c4 = e & 7, rowi = e >> 3;           // rowi = tapl*OP + co
                const int co = rowi % OP;
                *reinterpret_cast<float4*>(&dst[rowi * 32 + ((c4 ^ swz(co)) << 2)]) = v[n];
            }
        }
    };

    {   // prologue: tap row 0
        float4 hv[HPT];
        float4 wv[WPT];
        load_row(0, hv);
        load_w(0, wv);
        store_row(0, hv);
        store_w(0, wv);
    }
    __syncthreads();

    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int pcc = wave * 16 + li;                   // this lane's A-row pixel inside the tile

#pragma unroll 1
    for (int dy = 0; dy < 5; ++dy) {
        float4 hv[HPT];
        float4 wv[WPT];
        if (dy < 4) {
            load_row(dy + 1, hv);
            load_w(dy + 1, wv);
        }
        const float* hrow = halo + (dy & 1) * SLOT;
        const float* wbuf = Wt + (dy & 1) * WBUF;
        // register double buffering of the MFMA operands: the ds_reads of tap dx+1 are issued before the
        // 8*NT MFMAs of tap dx, so one LDS latency per tap row is exposed instead of one per 4 MFMAs
        float4 ao[2][2], bo[2][NT][2];
        auto load_ops = [&](int dx, float4 (&ar)[2], float4 (&br)[NT][2]) {
            const int hc = pcc + dx;
            const float* ap = hrow + hc * 32;
            const int sa = swz(hc);
            ar[0] = *reinterpret_cast<const float4*>(ap + ((g ^ sa) << 2));
            ar[1] = *reinterpret_cast<const float4*>(ap + (((g + 4) ^ sa) << 2));
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co = n * 16 + li;
                const float* bp = wbuf + (dx * OP + co) * 32;
                const int sb = swz(co);
                br[n][0] = *reinterpret_cast<const float4*>(bp + ((g ^ sb) << 2));
                br[n][1] = *reinterpret_cast<const float4*>(bp + (((g + 4) ^ sb) << 2));
            }
        };
        load_ops(0, ao[0], bo[0]);
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
            if (dx < 4) load_ops(dx + 1, ao[(dx + 1) & 1], bo[(dx + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ds_reads above this tap's MFMAs
            const float4 a0 = ao[dx & 1][0], a1 = ao[dx & 1][1];
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float4 b0 = bo[dx & 1][n][0], b1 = bo[dx & 1][n][1];
                const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv[kk], acc[n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (dy < 4) {
            store_row((dy + 1) & 1, hv);
            store_w((dy + 1) & 1, wv);
        }
        __syncthreads();
    }
    // ---- epilogue: transpose the wave's [16 px][OP] tile through LDS (the tile's halo ring is free
    //      after the last barrier) so that every lane moves 16-byte pieces of full 128-byte pixels ----
    if (a.CO == OP) {
        float* tb = halo + wave * (16 * OP);              // 16 px x OP floats per wave
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float bias = a.bias ? a.bias[n * 16 + li] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) tb[(4 * g + r) * OP + n * 16 + li] = acc[n][r] + bias;
        }
        // same-wave LDS round trip: the compiler's s_waitcnt lgkmcnt orders write -> read
        if (tvalid) {
            constexpr int F4 = 16 * OP / 4 / 64;          // float4 per lane
#pragma unroll
            for (int n = 0; n < F4; ++n) {
                const int e = lane + n * 64;              // float4 index inside the tile
                const int px = e / (OP / 4), c4 = e % (OP / 4);
                float4 v = *reinterpret_cast<const float4*>(&tb[px * OP + c4 * 4]);
                const size_t o4 = ((size_t)gy * W + x0 + wave * 16 + px) * (OP / 4) + c4;
                if (a.res) { const float4 q = reinterpret_cast<const float4*>(a.res)[o4]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
                if (a.epi == SOL_EPI_LRELU) {
                    v.x = v.x > 0.f ? v.x : a.slope * v.x; v.y = v.y > 0.f ? v.y : a.slope * v.y;
                    v.z = v.z > 0.f ? v.z : a.slope * v.z; v.w = v.w > 0.f ? v.w : a.slope * v.w;
                } else if (a.epi == SOL_EPI_DLRELU) {
                    const float4 q = reinterpret_cast<const float4*>(a.act)[o4];
                    v.x *= q.x > 0.f ? 1.f : a.slope; v.y *= q.y > 0.f ? 1.f : a.slope;
                    v.z *= q.z > 0.f ? 1.f : a.slope; v.w *= q.w > 0.f ? 1.f : a.slope;
                }
                st_wt(reinterpret_cast<float4*>(a.y) + o4, v);
            }
        }
        return;
    }
    if (!tvalid) return;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 16 + li;
        if (co >= a.CO) continue;
        const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cc = wave * 16 + 4 * g + r;
            const size_t o = ((size_t)gy * W + x0 + cc) * a.CO + co;
            float v = acc[n][r] + bias;
            if (a.res) v += a.res[o];
            if (a.epi == SOL_EPI_LRELU) v = v > 0.f ? v : a.slope * v;
            else if (a.epi == SOL_EPI_DLRELU) v *= (a.act[o] > 0.f ? 1.f : a.slope);
            a.y[o] = v;
        }
    }
}

// ------------------------------------------------------------------------------------
// backward-weight
// ------------------------------------------------------------------------------------
// Workgroup (dy, row block): accumulates dW[dy][0..4][ci][co] over RB image rows in MFMA
// accumulators (GEMM with K = pixels), then adds them into its private slice of `partial`
// (no atomics; the slice is owned by the workgroup index, calls on one stream serialise).
constexpr int RB = 8;   // image rows per workgroup

// BwArgs: common.hpp

template <int CIN, int COUT>   // real channel counts: CIN in {3,4,32} staged as pad_in, COUT in {2,32}
__global__ void __launch_bounds__(256) k_conv5x5_bww(BwArgs a) {
    constexpr int CI = CIN <= 4 ? 4 : 32;         // channels per pixel in global x
    constexpr int CPX = CI == 4 ? 4 : 48;         // LDS strides ( = 16 mod 32 -> conflict free b32 reads)
    constexpr int CPZ = COUT <= 4 ? 4 : 48;
    constexpr int NTA = CI == 4 ? 1 : 2;          // 16-wide tiles along ci / co
    constexpr int NTC = COUT <= 16 ? 1 : 2;
    constexpr int IP = CI == 4 ? 16 : 32;         // padded dims of the partial buffer
    constexpr int OP = NTC * 16;
    constexpr int XF4 = CI / 4;                   // float4 per pixel of x
    constexpr int NXR = CI == 32 ? 3 : 1;         // float4 registers per thread per x row  (W <= 92 / 252)
    constexpr int NZR = COUT > 4 ? 2 : 1;         // registers per thread per dz row        (W <= 64 / 256)
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int H = a.H, W = a.W;
    const int dy = blockIdx.x % 5, blk = blockIdx.x / 5;
    const int bufsz = (W + 4) * CPX + W * CPZ;    // floats per stage: x row then dz row
    const int ta = wave / NTC, tc = wave % NTC;
    const bool active_wave = wave < NTA * NTC;

    f32x4 acc[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) acc[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    const int R = a.nseg * a.B * H, RPS = a.B * H;
    const int gr_end = min((blk + 1) * a.rb, R);
    // rows this workgroup visits: dy == 2 always (bias needs every dz row), else only valid taps
    auto row_valid = [&](int gr) { const int y = gr % H, yy = y + dy - 2; return yy >= 0 && yy < H; };
    auto next_row = [&](int gr) { while (gr < gr_end && !(dy == 2 || row_valid(gr))) ++gr; return gr; };

    float4 xr[NXR];
    float4 zr[NZR];
    auto load_row = [&](int gr) {
        const int seg = gr / RPS, grs = gr - seg * RPS;
        const int b = grs / H, y = grs - b * H, yy = y + dy - 2;
        const bool valid = yy >= 0 && yy < H;
        const float4* gx = reinterpret_cast<const float4*>(a.x + (size_t)seg * a.x_seg) + (size_t)(b * H + (valid ? yy : 0)) * W * XF4;
#pragma unroll
        for (int n = 0; n < NXR; ++n) {
            const int e = tid + n * 256;
            const int px = e / XF4, c4 = e - px * XF4, xx = px - 2;
            xr[n] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid && px < W + 4 && xx >= 0 && xx < W) xr[n] = gx[xx * XF4 + c4];
        }
        if constexpr (COUT > 4) {
            const float4* gz = reinterpret_cast<const float4*>(a.dz + (size_t)seg * a.dz_seg) + (size_t)(b * H + y) * W * (COUT / 4);
#pragma unroll
            for (int n = 0; n < NZR; ++n) {
                const int e = tid + n * 256;
                zr[n] = e < W * (COUT / 4) ? gz[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const float2* gz = reinterpret_cast<const float2*>(a.dz + (size_t)seg * a.dz_seg) + (size_t)(b * H + y) * W;
            const float2 v = tid < W ? gz[tid] : make_float2(0.f, 0.f);
            zr[0] = make_float4(v.x, v.y, 0.f, 0.f);
        }
    };
    auto store_row = [&](float* buf) {
        float* xs = buf;
        float* zs = buf + (W + 4) * CPX;
#pragma unroll
        for (int n = 0; n < NXR; ++n) {
            const int e = tid + n * 256;
            const int px = e / XF4, c4 = e - px * XF4;
            if (px < W + 4) *reinterpret_cast<float4*>(&xs[px * CPX + c4 * 4]) = xr[n];
        }
        if constexpr (COUT > 4) {
#pragma unroll
            for (int n = 0; n < NZR; ++n) {
                const int e = tid + n * 256;
                const int px = e / (COUT / 4), c4 = e % (COUT / 4);
                if (e < W * (COUT / 4)) *reinterpret_cast<float4*>(&zs[px * CPZ + c4 * 4]) = zr[n];
            }
        } else {
            if (tid < W) *reinterpret_cast<float4*>(&zs[tid * CPZ]) = zr[0];
        }
    };

    int gr = next_row(blk * a.rb);
    int cur = 0;
    if (gr < gr_end) {
        load_row(gr);
        store_row(smem);
    }
    __syncthreads();
    while (gr < gr_end) {
        const int gnext = next_row(gr + 1);
        if (gnext < gr_end) load_row(gnext);           // global -> registers, overlaps the MFMAs below
        const bool valid = row_valid(gr);              // workgroup uniform
        const float* xs = smem + cur * bufsz;
        const float* zs = xs + (W + 4) * CPX;
        if (active_wave) {
            for (int p0 = 0; p0 < W; p0 += 4) {
                const int px = p0 + g;
                float bv;
                if constexpr (COUT > 4) bv = zs[px * CPZ + tc * 16 + li];
                else bv = li < 4 ? zs[px * CPZ + li] : 0.f;
                if (dy == 2 && ta == 0) bsum += bv;
                if (valid) {
#pragma unroll
                    for (int d = 0; d < 5; ++d) {
                        float av;
                        if constexpr (CI == 32) av = xs[(px + d) * CPX + ta * 16 + li];
                        else av = li < 4 ? xs[(px + d) * CPX + li] : 0.f;
                        acc[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[d], 0, 0, 0);
                    }
                }
            }
        }
        if (gnext < gr_end) store_row(smem + (cur ^ 1) * bufsz);
        __syncthreads();
        cur ^= 1;
        gr = gnext;
    }
    if (!active_wave) return;
    // partial layout: [blk][25 taps][IP][OP] then [blk][OP] bias sums
    float* pw = a.partial + (size_t)blk * (25 * IP * OP);
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        const int tap = dy * 5 + d;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = ta * 16 + 4 * g + r, co = tc * 16 + li;
            float* dst = &pw[(tap * IP + ci) * OP + co];
            *dst = a.overwrite ? acc[d][r] : *dst + acc[d][r];
        }
    }
    if (dy == 2 && ta == 0) {
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (g == 0) {
            float* pb = a.partial + (size_t)a.nblk * (25 * IP * OP) + (size_t)blk * OP;
            pb[tc * 16 + li] = a.overwrite ? bsum : pb[tc * 16 + li] + bsum;
        }
    }
}

// 32x32-channel, W == 64 variant on v_mfma_f32_32x32x2_f32: the whole [ci 32][co 32] tile of a tap
// is ONE accumulator (A[i=ci][k=px], B[k=px][j=co]; lanes 0-31 / 32-63 take two consecutive pixels),
// so both operands are conflict-free ds_read_b32 straight from the NHWC rows -- no transpose, no
// padding, 6 LDS reads per 5 MFMAs of 64 cycles.  The 4 waves split the 64 pixels of a row (K split)
// and fold their accumulators through LDS once at the end.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) k_conv5x5_bww32(BwArgs a) {
    constexpr int W = 64, STG = (W + 4) * 32 + W * 32;      // floats per stage: x row (68 px) + dz row
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, kpx = lane >> 5;
    const int H = a.H;
    const int dy = blockIdx.x % 5, blk = blockIdx.x / 5;
    f32x16 acc[5];
#pragma unroll
    for (int d = 0; d < 5; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    float bsum = 0.f;
    const int R = a.nseg * a.B * H, RPS = a.B * H;
    const int gr_end = min((blk + 1) * a.rb, R);
    auto row_valid = [&](int gr) { const int y = gr % H, yy = y + dy - 2; return yy >= 0 && yy < H; };
    auto next_row = [&](int gr) { while (gr < gr_end && !(dy == 2 || row_valid(gr))) ++gr; return gr; };
    float4 xr[3], zr[2];
    auto load_row = [&](int gr) {
        const int seg = gr / RPS, grs = gr - seg * RPS;
        const int b = grs / H, y = grs - b * H, yy = y + dy - 2;
        const bool valid = yy >= 0 && yy < H;
        const float4* gx = reinterpret_cast<const float4*>(a.x + (size_t)seg * a.x_seg) + (size_t)(b * H + (valid ? yy : 0)) * W * 8;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int e = tid + n * 256, px = e >> 3, c4 = e & 7, xx = px - 2;
            xr[n] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid && px < W + 4 && xx >= 0 && xx < W) xr[n] = gx[xx * 8 + c4];
        }
        const float4* gz = reinterpret_cast<const float4*>(a.dz + (size_t)seg * a.dz_seg) + (size_t)(b * H + y) * W * 8;
#pragma unroll
        for (int n = 0; n < 2; ++n) zr[n] = gz[tid + n * 256];
    };
    auto store_row = [&](float* buf) {
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int e = tid + n * 256;
            if (e < (W + 4) * 8) reinterpret_cast<float4*>(buf)[e] = xr[n];
        }
        float4* zs4 = reinterpret_cast<float4*>(buf + (W + 4) * 32);
#pragma unroll
        for (int n = 0; n < 2; ++n) zs4[tid + n * 256] = zr[n];
    };
    int gr = next_row(blk * a.rb);
    int cur = 0;
    if (gr < gr_end) { load_row(gr); store_row(smem); }
    __syncthreads();
    while (gr < gr_end) {
        const int gnext = next_row(gr + 1);
        if (gnext < gr_end) load_row(gnext);
        const bool valid = row_valid(gr);
        const float* xs = smem + cur * STG + (16 * wave + kpx) * 32 + c;
        const float* zs = smem + cur * STG + (W + 4) * 32 + (16 * wave + kpx) * 32 + c;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const float bv = zs[ks * 64];
            if (dy == 2) bsum += bv;
            if (valid) {
#pragma unroll
                for (int d = 0; d < 5; ++d)
                    acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[(2 * ks + d) * 32], bv, acc[d], 0, 0, 0);
            }
        }
        if (gnext < gr_end) store_row(smem + (cur ^ 1) * STG);
        __syncthreads();
        cur ^= 1;
        gr = gnext;
    }
    // fold the 4 K-split accumulators through LDS, one tap at a time, into this block's partial slice
    float* red = smem;                                  // [4][32][32]
    float* pw = a.partial + (size_t)blk * (25 * 1024);
    for (int d = 0; d < 5; ++d) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kpx;
            red[wave * 1024 + row * 32 + c] = acc[d][r];
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int idx = tid + n * 256;
            const float v = (red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx]);
            float* dst = &pw[(dy * 5 + d) * 1024 + idx];
            *dst = a.overwrite ? v : *dst + v;
        }
        __syncthreads();
    }
    if (dy == 2) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (kpx == 0) red[wave * 32 + c] = bsum;
        __syncthreads();
        if (tid < 32) {
            float* pb = a.partial + (size_t)a.nblk * (25 * 1024) + (size_t)blk * 32;
            const float v = (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]);
            pb[tid] = a.overwrite ? v : pb[tid] + v;
        }
    }
}

// Weight gradient of the two thin layers (first: 3(+1 pad) -> 32 channels, last: 32 -> 2) on
// v_mfma_f32_16x16x4_f32 with the five dx taps folded into the otherwise empty matrix dimension:
//   MODE 0 (cin 4, cout 32):  A[i = (dx, ci)][k = px] = x[px+dx][ci],   B[k = px][j = co]      = dz[px][co]
//   MODE 1 (cin 32, cout 2):  A[i = ci][k = q]        = x[q][ci],       B[k = q][j = 2*dx+co]  = dz[q-dx][co]
// so one accumulator set per tap row dy holds all five dx taps and a workgroup (4 waves, K split over the
// pixels) keeps the accumulators of ALL tap rows: x and dz are read once instead of 5 times.  Rows are double
// buffered in LDS with register prefetch.
// Round 6: 16 x 16 tiles instead of one 32 x 32 x 2 tile per tap row.  The folded dimension has 5 x 3 = 15 entries when the
// fourth input channel is zero padding (MODE 0, MT = 1: ONE 16-row tile, 15 / 16 used instead of 20 / 32) and 10 entries in
// MODE 1 (ONE 16-column tile, 10 / 16 used instead of 10 / 32): half the matrix-pipe cycles per pixel for the same products
// (the kernels are bound by the fp32 matrix pipe: 40 x 64 clocks per row and wave before, 40 x 32 / 50 x 32 now).  MT = 2 (four
// real input channels: 20 folded entries, two 16-row tiles) costs what the 32 x 32 form did.
template <int MODE, int MT>
__global__ void __launch_bounds__(256) k_conv5x5_bww_thin(BwArgs a) {
    constexpr int W = 64;
    constexpr int XC = MODE == 0 ? 4 : 32, ZC = MODE == 0 ? 32 : 2;            // channels per pixel of x / dz
    constexpr int XROWS = MODE == 0 ? 5 : 1, ZROWS = MODE == 0 ? 1 : 5;        // rows staged per iteration
    constexpr int XSZ = (W + 4) * XC;
    constexpr int ZSZ = (MODE == 0 ? W + 4 : W + 8) * ZC;                      // MODE 1: dz rows staged with 4 zero pixels in front and behind
    constexpr int STG = XROWS * XSZ + ZROWS * ZSZ;
    constexpr int IP = MODE == 0 ? 16 : 32, OP = MODE == 0 ? 32 : 16;
    constexpr int XF4 = XROWS * XSZ / 4, ZF4 = ZROWS * ZSZ / 4;                // float4 per stage
    constexpr int NXR = (XF4 + 255) / 256, NZR = (ZF4 + 255) / 256;
    constexpr int NTM = MODE == 0 ? MT : 2, NTN = MODE == 0 ? 2 : 1;           // 16 x 16 tiles of a tap row's accumulator
    constexpr int CI0 = MT == 1 ? 3 : 4;                                       // MODE 0: input channels folded with dx (15 or 20 rows)
    typedef float bt_f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    const int H = a.H, blk = blockIdx.x;
    bt_f4 acc[5][NTM][NTN];
#pragma unroll
    for (int d = 0; d < 5; ++d)
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn) acc[d][tm][tn] = (bt_f4){0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};
    const int R = a.nseg * a.B * H, RPS = a.B * H;
    const int gr_end = min((blk + 1) * a.rb, R);
    float4 xr[NXR], zr[NZR];
    // iteration row gr: MODE 0 = dz (output) row, needs x rows y+dy-2; MODE 1 = x (input) row, needs dz rows y+2-dy
    auto load_row = [&](int gr) {
        const int seg = gr / RPS, grs = gr - seg * RPS;
        const int b = grs / H, y = grs - b * H;
        const float* xb = a.x + (size_t)seg * a.x_seg + (size_t)b * H * W * XC;
        const float* zb = a.dz + (size_t)seg * a.dz_seg + (size_t)b * H * W * ZC;
#pragma unroll
        for (int n = 0; n < NXR; ++n) {
            const int e = tid + n * 256;              // float4 index inside the x stage
            xr[n] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < XF4) {
                const int row = e / (XSZ / 4), f = e - row * (XSZ / 4);
                const int px = f / (XC / 4), c4 = f - px * (XC / 4), xx = px - 2;
                const int yy = MODE == 0 ? y + row - 2 : y;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) xr[n] = reinterpret_cast<const float4*>(xb + ((size_t)yy * W + xx) * XC)[c4];
            }
        }
#pragma unroll
        for (int n = 0; n < NZR; ++n) {
            const int e = tid + n * 256;
            zr[n] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < ZF4) {
                if (MODE == 0) {                      // one dz row, no halo: pixels 0..63 at stage pixels 0..63
                    if (e < W * ZC / 4) zr[n] = reinterpret_cast<const float4*>(zb + (size_t)y * W * ZC)[e];
                } else {                              // five dz rows y+2-dy, each [72 px][2] with pixel p at stage pixel p+4
                    const int row = e / (ZSZ / 4), f = e - row * (ZSZ / 4);      // f: float4 = 2 pixels
                    const int yy = y + 2 - row, p = 2 * f - 4;
                    if (yy >= 0 && yy < H && p >= 0 && p < W) zr[n] = reinterpret_cast<const float4*>(zb + ((size_t)yy * W + p) * ZC)[0];
                }
            }
        }
    };
    auto store_row = [&](float* buf) {
#pragma unroll
        for (int n = 0; n < NXR; ++n) { const int e = tid + n * 256; if (e < XF4) reinterpret_cast<float4*>(buf)[e] = xr[n]; }
#pragma unroll
        for (int n = 0; n < NZR; ++n) { const int e = tid + n * 256; if (e < ZF4) reinterpret_cast<float4*>(buf + XROWS * XSZ)[e] = zr[n]; }
    };
    int gr = blk * a.rb, cur = 0;
    if (gr < gr_end) { load_row(gr); store_row(smem); }
    __syncthreads();
    for (; gr < gr_end; ++gr) {
        if (gr + 1 < gr_end) load_row(gr + 1);
        const float* xs = smem + cur * STG;
        const float* zs = xs + XROWS * XSZ;
        if (MODE == 0) {
            // A operand lane (m = n16, k = kq): folded row i = 16 tm + n16 = (dx, ci); B operand lane (k = kq, n = n16): co = 16 tn + n16.
            // A wave handles pixels [16 wave, 16 wave + 16): four k-steps of four pixels.
            int aoff[NTM];
            bool aok[NTM];
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm) {
                const int i = 16 * tm + n16;
                aok[tm] = i < 5 * CI0;
                aoff[tm] = aok[tm] ? (i / CI0) * 4 + i % CI0 : 0;              // (px + dx) * 4 + ci relative to pixel px
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int px = 16 * wave + 4 * ks + kq;
                float bv[2];
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) { bv[tn] = zs[px * 32 + 16 * tn + n16]; bsum[tn] += bv[tn]; }
#pragma unroll
                for (int d = 0; d < 5; ++d) {        // tap row dy = d uses x row slot d (= image row y+d-2, zero if outside)
#pragma unroll
                    for (int tm = 0; tm < NTM; ++tm) {
                        const float av = aok[tm] ? xs[d * XSZ + px * 4 + aoff[tm]] : 0.f;
#pragma unroll
                        for (int tn = 0; tn < NTN; ++tn) acc[d][tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[tn], acc[d][tm][tn], 0, 0, 0);
                    }
                }
            }
        } else {
            // K runs over the 68 halo pixels q of the x row in 17 groups of four; wave w takes the groups g = w, w + 4, ... (5, 4, 4, 4 of them)
            const int dxi = n16 >> 1, co = n16 & 1;
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const int g = wave + 4 * ks;
                const bool ok = g < 17;
                const int qq = ok ? 4 * g + kq : 0;
                float av[2];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) av[tm] = ok ? xs[qq * 32 + 16 * tm + n16] : 0.f;
                // B[q][j = (dx, co)]: x halo pixel q is image pixel q-2; output pixel = q-2-(dx-2) = q-dx -> stage pixel q-dx+4
#pragma unroll
                for (int d = 0; d < 5; ++d) {        // tap row dy = d pairs x row y with dz row y+2-d (stage slot d)
                    const float bv = (ok && n16 < 10) ? zs[d * ZSZ + (qq - dxi + 4) * 2 + co] : 0.f;
                    if (d == 2 && n16 < 2) bsum[0] += bv;    // dx = 0 columns of the dy = 2 slot: every dz pixel of the row once
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm) acc[d][tm][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm], bv, acc[d][tm][0], 0, 0, 0);
                }
            }
        }
        if (gr + 1 < gr_end) store_row(smem + (cur ^ 1) * STG);
        __syncthreads();
        cur ^= 1;
    }
    // fold the 4 K-split accumulators through LDS, one tap row at a time: red[wave][row = folded / input-channel index][32 columns]
    float* red = smem;                                  // [4][32][32]
    float* pw = a.partial + (size_t)blk * (25 * IP * OP);
    for (int d = 0; d < 5; ++d) {
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * 1024 + (16 * tm + 4 * kq + r) * 32 + 16 * tn + n16] = acc[d][tm][tn][r];
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int idx = tid + n * 256, row = idx >> 5, col = idx & 31;
            const float v = (red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx]);
            int tap, ci, co;
            bool keep;
            if (MODE == 0) { tap = d * 5 + row / CI0; ci = row % CI0; co = col; keep = row < 5 * CI0; }
            else { tap = d * 5 + (col >> 1); ci = row; co = col & 1; keep = col < 10; }
            if (keep) {
                float* dst = &pw[(tap * IP + ci) * OP + co];
                *dst = a.overwrite ? v : *dst + v;
            }
        }
        __syncthreads();
    }
    {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bsum[t] += __shfl_xor(bsum[t], 16, 64);
            bsum[t] += __shfl_xor(bsum[t], 32, 64);
            if (kq == 0) red[wave * 32 + 16 * t + n16] = bsum[t];
        }
        __syncthreads();
        const int nb = MODE == 0 ? 32 : 2;
        if (tid < nb) {
            float* pb = a.partial + (size_t)a.nblk * (25 * IP * OP) + (size_t)blk * OP;
            const float v = (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]);
            pb[tid] = a.overwrite ? v : pb[tid] + v;
        }
    }
}

// partial [nblk][25][IP][OP] (+ [nblk][OP] bias sums) -> dw [25][cin][cout], db [cout], deterministic:
// stage 1 (to_dw = 0): grid.y chunks of `chunk` blocks are summed in parallel, the sum is written back into
// the first block of the chunk; stage 2 (to_dw = 1): the chunk heads (stride `chunk`) are summed in order.
__global__ void k_bww_reduce(float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db,
                             int nblk, int cin, int cout, int IP, int OP, int chunk, int stride, int to_dw, int accumulate, int tt) {
    const int nw = 25 * cin * cout;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int k0 = blockIdx.y * chunk * stride, k1 = min(k0 + chunk * stride, nblk);
    if (e < nw) {
        const int co = e % cout, ci = (e / cout) % cin, tap = e / (cout * cin);
        const size_t off = (size_t)(tap * IP + ci) * OP + co;
        float s = 0.f;
        for (int k = k0; k < k1; k += stride) s += partial[(size_t)k * (25 * IP * OP) + off];
        // tt: the gradient was accumulated on transposed images: its tap (dy, dx) is the weight's tap (dx, dy)
        const int eo = tt ? (((tap % 5) * 5 + tap / 5) * cin + ci) * cout + co : e;
        if (to_dw) dw[eo] = accumulate ? dw[eo] + s : s;
        else partial[(size_t)k0 * (25 * IP * OP) + off] = s;
    } else if (e < nw + cout) {
        const int co = e - nw;
        float* pb = partial + (size_t)nblk * (25 * IP * OP);
        float s = 0.f;
        for (int k = k0; k < k1; k += stride) s += pb[(size_t)k * OP + co];
        if (to_dw) db[co] = accumulate ? db[co] + s : s;
        else pb[(size_t)k0 * OP + co] = s;
    }
}

// the same two stages for SEVERAL layers in one launch each (blockIdx.z = layer): a training step reduces twelve layers, 24
// launches of a few microseconds of work each
struct ReduceJob { float *partial, *dw, *db; int nblk, cin, cout, IP, OP, ny, accumulate, tt; };
struct ReduceJobs { ReduceJob j[12]; };
__global__ void k_bww_reduce_jobs(ReduceJobs J, int stage2) {
    const ReduceJob& r = J.j[blockIdx.z];
    constexpr int CHUNK = 16;
    const int nw = 25 * r.cin * r.cout;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    int k0, k1, stride, to_dw;
    if (!stage2) {                      // chunk sums, folded into the first block of every chunk (layers with one chunk: nothing)
        if (r.ny <= 1 || (int)blockIdx.y >= r.ny) return;
        k0 = blockIdx.y * CHUNK; k1 = min(k0 + CHUNK, r.nblk); stride = 1; to_dw = 0;
    } else {                            // chunk heads (or, with one chunk, the blocks themselves) in order -> dw, db
        if (blockIdx.y > 0) return;
        k0 = 0; k1 = r.nblk; stride = r.ny > 1 ? CHUNK : 1; to_dw = 1;
    }
    if (e < nw) {
        const int co = e % r.cout, ci = (e / r.cout) % r.cin, tap = e / (r.cout * r.cin);
        const size_t off = (size_t)(tap * r.IP + ci) * r.OP + co;
        float s = 0.f;
        for (int k = k0; k < k1; k += stride) s += r.partial[(size_t)k * (25 * r.IP * r.OP) + off];
        const int eo = r.tt ? (((tap % 5) * 5 + tap / 5) * r.cin + ci) * r.cout + co : e;
        if (to_dw) r.dw[eo] = r.accumulate ? r.dw[eo] + s : s;
        else r.partial[(size_t)k0 * (25 * r.IP * r.OP) + off] = s;
    } else if (e < nw + r.cout) {
        const int co = e - nw;
        float* pb = r.partial + (size_t)r.nblk * (25 * r.IP * r.OP);
        float s = 0.f;
        for (int k = k0; k < k1; k += stride) s += pb[(size_t)k * r.OP + co];
        if (to_dw) r.db[co] = r.accumulate ? r.db[co] + s : s;
        else pb[(size_t)k0 * r.OP + co] = s;
    }
}

int check_shape(int B, int H, int W, int cin, int cout) {
    SOL_REQUIRE(B >= 1 && H >= 1 && W >= 4, "conv5x5: bad image shape B=%d H=%d W=%d", B, H, W);
    SOL_REQUIRE((W <= 64 && 64 % W == 0 && H % (64 / W) == 0) || W % 64 == 0,
                "conv5x5: W must divide 64 (with H %% (64/W) == 0) or be a multiple of 64 (H=%d W=%d)", H, W);
    SOL_REQUIRE(cin == 4 || cin == 32, "conv5x5: input channels must be 4 (zero padded) or 32 (got %d)", cin);
    SOL_REQUIRE(cout >= 1 && cout <= 32, "conv5x5: output channels must be in [1,32] (got %d)", cout);
    return SOL_OK;
}

}  // namespace

extern "C" int32_t sol_absmax_slots(void) {
    static_assert(SOL_AMAX_SLOTS == SOL_ABSMAX_SLOTS, "include/sol_hip.h and common.hpp disagree");
    return SOL_AMAX_SLOTS;
}

extern "C" size_t sol_conv5x5_packed_floats(int32_t cin, int32_t cout, int32_t /*mode*/) {
    // fp32 section (all shapes) + split-bf16 planes for the 32-input-channel kernels (conv5x5_sb.hip)
    return (size_t)25 * pad_in(cin) * pad_out(cout) +
           (pad_in(cin) == 32 ? sol_conv_sb_packed_floats(pad_out(cout)) + sol_conv_sh_packed_floats(pad_out(cout)) : 0);
}

extern "C" int sol_conv5x5_pack(void* stream, const float* w_hwio, int32_t cin, int32_t cout, int32_t mode, float* packed) {
    SOL_REQUIRE(w_hwio && packed, "sol_conv5x5_pack: NULL pointer");
    SOL_REQUIRE(cin >= 1 && cin <= 32 && cout >= 1 && cout <= 32, "sol_conv5x5_pack: channels out of range");
    SOL_REQUIRE(mode == SOL_CONV_FWD || mode == SOL_CONV_BWD_DATA, "sol_conv5x5_pack: bad mode %d", mode);
    const int total = 25 * pad_in(cin) * pad_out(cout);
    SOL_LAUNCH(k_pack, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_hwio, packed, cin, cout, mode);
    SOL_LAUNCH_CHECK();
    if (pad_in(cin) == 32) {
        if (int e = sol_conv_sb_pack((hipStream_t)stream, w_hwio, cin, cout, mode, packed + total)) return e;
        return sol_conv_sh_pack((hipStream_t)stream, w_hwio, cin, cout, mode, packed + total + sol_conv_sb_packed_floats(pad_out(cout)));
    }
    return SOL_OK;
}

// n <= 24 (layer, mode) pack jobs in ONE launch (k_pack_jobs: what the C++ trainer packs its step's weights with): the per-step weight
// packing of a host-composed schedule (schedule2d.NetSchedule2D) is one launch instead of two to ten per layer and mode.
extern "C" int sol_conv5x5_pack_jobs(void* stream, int32_t n, const float* const* w_hwio, const int32_t* cin, const int32_t* cout,
                                     const int32_t* mode, float* const* packed) {
    SOL_REQUIRE(n >= 1 && n <= 24 && w_hwio && cin && cout && mode && packed, "sol_conv5x5_pack_jobs: 1 <= n <= 24 jobs, no NULL array");
    const float* bi[24];
    float* bo[24];
    int ci[24], co[24], md[24];
    for (int k = 0; k < n; ++k) {
        SOL_REQUIRE(w_hwio[k] && packed[k], "sol_conv5x5_pack_jobs: NULL pointer in job %d", k);
        SOL_REQUIRE(cin[k] >= 1 && cin[k] <= 32 && cout[k] >= 1 && cout[k] <= 32, "sol_conv5x5_pack_jobs: channels out of range in job %d", k);
        SOL_REQUIRE(mode[k] == SOL_CONV_FWD || mode[k] == SOL_CONV_BWD_DATA, "sol_conv5x5_pack_jobs: bad mode %d in job %d", mode[k], k);
        bi[k] = nullptr; bo[k] = nullptr; ci[k] = cin[k]; co[k] = cout[k]; md[k] = mode[k];
    }
    return sol_pack_jobs((hipStream_t)stream, n, w_hwio, packed, bo, bi, ci, co, md, 0);
}

int sol_init_conv_kernels() {
    static std::atomic<unsigned long long> optin{0};
    return sol_lds_optin(optin, {SOL_K(k_conv5x5_r3<1>), SOL_K(k_conv5x5_r3<2>), SOL_K(k_conv5x5_c32<1>), SOL_K(k_conv5x5_c32<2>)}, "conv kernels");
}

static int conv_impl(void* stream, const float* x, const float* packed, const float* bias,
                     const float* residual, const float* act_ref, float* y,
                     int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout,
                     int32_t epilogue, float slope, const uint32_t* x_absmax, uint32_t* y_absmax) {
    if (int e = check_shape(B, H, W, cin, cout)) return e;
    SOL_REQUIRE(x && packed && y, "sol_conv5x5: NULL pointer");
    SOL_REQUIRE(epilogue != SOL_EPI_DLRELU || act_ref, "sol_conv5x5: SOL_EPI_DLRELU needs act_ref");
    if (int e = sol_init_conv_kernels()) return e;
    ConvArgs a{};
    a.x = x; a.wp = packed; a.bias = bias; a.res = residual; a.act = act_ref; a.y = y;
    a.B = B; a.H = H; a.W = W; a.CO = cout; a.CI = cin; a.epi = epilogue; a.slope = slope;
    a.TW = W < 64 ? W : 64;
    a.RPW = 64 / a.TW;
    a.tiles_x = W / a.TW;
    const int grid = B * (H / a.RPW) * a.tiles_x;
    const int CP = cin == 4 ? 4 : 36;
    size_t lds = (size_t)(a.RPW + 4) * (a.TW + 4) * CP * sizeof(float);
    if (cin == 4) {
        lds += (size_t)25 * pad_out(cout) * 4 * sizeof(float);                                // weight block
        if (lds < 4 * 16 * 32 * sizeof(float)) lds = 4 * 16 * 32 * sizeof(float);             // epilogue transposition buffers
    }
    const int NT = pad_out(cout) / 16;
    hipStream_t s = (hipStream_t)stream;
    const bool use_sb = sol_opt().conv_precision != 2;
    const bool use_sh = sol_opt().conv_precision == 0;
    if (use_sb) a.ymax = y_absmax;                  // published by the split kernels and the thin fp32 kernel below
    if (cin == 32 && W % 64 == 0 && use_sb) {
        a.wsb = packed + (size_t)25 * 32 * pad_out(cout);
        a.wsh = packed + (size_t)25 * 32 * pad_out(cout) + sol_conv_sb_packed_floats(pad_out(cout));
        a.xmax = use_sh ? x_absmax : nullptr;       // absmax of x known -> fp16 three-product kernel, else bf16 six-product
        return sol_conv_sb_launch(s, a, NT, B * H * (W / 64));
    }
    if (cin == 32 && W % 64 == 0 && sol_opt().conv_r3) {
        const int ntiles = B * H * (W / 64);
        const size_t ldsr = ((size_t)3 * 2 * 68 * 32 + 2 * (size_t)5 * NT * 16 * 32) * sizeof(float);
        const int grid3 = (ntiles + 2) / 3;
        if (NT == 2) {
            SOL_LAUNCH((k_conv5x5_r3<2>), dim3(grid3), dim3(768), ldsr, s, a, ntiles);
        } else {
            SOL_LAUNCH((k_conv5x5_r3<1>), dim3(grid3), dim3(768), ldsr, s, a, ntiles);
        }
    }
    else if (cin == 32) {
        const size_t lds32 = ((size_t)(a.RPW + 4) * (a.TW + 4) * 32 + 2 * (size_t)NT * 16 * 32) * sizeof(float);
        if (NT == 2) {
            SOL_LAUNCH((k_conv5x5_c32<2>), dim3(grid), dim3(256), lds32, s, a);
        } else {
            SOL_LAUNCH((k_conv5x5_c32<1>), dim3(grid), dim3(256), lds32, s, a);
        }
    }
    else if (conv_t3_usable(B, H, W, cin, pad_out(cout)) && cout == pad_out(cout)) {
        if (NT == 2) SOL_LAUNCH((k_conv5x5_t3<2>), dim3((B * H + 2) / 3), dim3(768), conv_t3_lds(32), s, a);
        else SOL_LAUNCH((k_conv5x5_t3<1>), dim3((B * H + 2) / 3), dim3(768), conv_t3_lds(16), s, a);
    }
    else if (cin == 4 && NT == 2) SOL_LAUNCH((k_conv5x5<4, 2>), dim3(grid), dim3(256), lds, s, a);
    else SOL_LAUNCH((k_conv5x5<4, 1>), dim3(grid), dim3(256), lds, s, a);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

extern "C" int sol_conv5x5(void* stream, const float* x, const float* packed, const float* bias,
                           const float* residual, const float* act_ref, float* y,
                           int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout,
                           int32_t epilogue, float slope) {
    return conv_impl(stream, x, packed, bias, residual, act_ref, y, B, H, W, cin, cout, epilogue, slope, nullptr, nullptr);
}

const void* sol_conv_packed_wsh(const float* packed, int cout) {
    return packed + (size_t)25 * 32 * pad_out(cout) + sol_conv_sb_packed_floats(pad_out(cout));
}

// The 2 -> 32 backward-data layer of the trainer's reverse sweep with the loss-gradient seed computed in its staging phase (ConvArgs seed
// fields): y = conv(dO, packed_bwd) * lrelu'(act_ref), dO = out_std * G, G = (v - gt) / (std^2 msteps) (+ gin); writes G and dO2.  W == 64.
int sol_conv5x5_seed(void* stream, const float* packed_bwd, const float* act_ref, float* y, int B, int H, int W, float slope, unsigned* y_absmax,
                     const float* vy, const float* vx, const float* gt_vy, const float* gt_vx, const float* gin_vy, const float* gin_vx,
                     float* g_vy, float* g_vx, float* dO2, float s0, float s1, float l0, float l1, float inv_m) {
    if (int e = check_shape(B, H, W, 4, 32)) return e;
    SOL_REQUIRE(W == 64 && packed_bwd && act_ref && y && vy && vx && gt_vy && gt_vx && g_vy && g_vx && dO2 && !gin_vy == !gin_vx, "sol_conv5x5_seed: bad arguments");
    if (int e = sol_init_conv_kernels()) return e;
    ConvArgs a{};
    a.x = nullptr; a.wp = packed_bwd; a.act = act_ref; a.y = y;
    a.B = B; a.H = H; a.W = W; a.CO = 32; a.CI = 4; a.epi = SOL_EPI_DLRELU; a.slope = slope;
    a.TW = 64; a.RPW = 1; a.tiles_x = 1;
    if (sol_opt().conv_precision != 2) a.ymax = y_absmax;
    a.svy = vy; a.svx = vx; a.gty = gt_vy; a.gtx = gt_vx; a.sginy = gin_vy; a.sginx = gin_vx; a.sgy = g_vy; a.sgx = g_vx; a.sdO2 = dO2;
    a.cs0 = s0; a.cs1 = s1; a.ls0 = l0; a.ls1 = l1; a.sinv_m = inv_m;
    size_t lds = (size_t)5 * 68 * 4 * sizeof(float) + (size_t)25 * 32 * 4 * sizeof(float);
    if (lds < 4 * 16 * 32 * sizeof(float)) lds = 4 * 16 * 32 * sizeof(float);
    if (conv_t3_usable(B, H, W, 4, 32)) SOL_LAUNCH((k_conv5x5_t3<2>), dim3((B * H + 2) / 3), dim3(768), conv_t3_lds(32), (hipStream_t)stream, a);
    else SOL_LAUNCH((k_conv5x5<4, 2>), dim3(B * H), dim3(256), lds, (hipStream_t)stream, a);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

bool sol_conv_correct_fusable(int W, int rows) {
    (void)rows;
    return sol_opt().conv_precision == 0 && sol_opt().correct_fuse && W % 64 == 0;
}

int sol_conv5x5_correct(void* stream, const float* x, const float* packed, const float* bias, int B, int H, int W,
                        const unsigned* x_absmax, float* vy, float* vx, const float* gt_vy, const float* gt_vx,
                        float s0, float s1, float l0, float l1, unsigned long long* loss_acc, int transposed) {
    if (int e = check_shape(B, H, W, 32, 2)) return e;
    SOL_REQUIRE(x && packed && x_absmax && vy && vx && sol_conv_correct_fusable(W, B * H), "sol_conv5x5_correct: bad arguments");
    // (the thin dx kernel's prefetch keys both ground-truth reads on gt_vy: one frame without the other would be a NULL-based device read)
    SOL_REQUIRE(!gt_vy == !gt_vx, "sol_conv5x5_correct: gt_vy and gt_vx must be given together (or both NULL: no loss)");
    SOL_REQUIRE(!loss_acc || gt_vy, "sol_conv5x5_correct: a loss accumulator needs the ground-truth frames");
    if (int e = sol_init_conv_kernels()) return e;
    ConvArgs a{};
    a.x = x; a.wp = packed; a.bias = bias; a.B = B; a.H = H; a.W = W; a.CO = 2; a.CI = 32; a.epi = SOL_EPI_NONE;
    a.TW = 64; a.RPW = 1; a.tiles_x = W / 64;
    a.wsb = packed + (size_t)25 * 32 * pad_out(2);
    a.wsh = packed + (size_t)25 * 32 * pad_out(2) + sol_conv_sb_packed_floats(pad_out(2));
    a.xmax = x_absmax;
    a.cvy = vy; a.cvx = vx; a.gty = gt_vy; a.gtx = gt_vx; a.cs0 = s0; a.cs1 = s1; a.ls0 = l0; a.ls1 = l1; a.closs = loss_acc; a.ctr = transposed ? 1 : 0;
    return sol_conv_sb_launch((hipStream_t)stream, a, 1, B * H * (W / 64));
}

extern "C" int sol_conv5x5_scaled(void* stream, const float* x, const float* packed, const float* bias,
                                  const float* residual, const float* act_ref, float* y,
                                  int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout,
                                  int32_t epilogue, float slope, const uint32_t* x_absmax, uint32_t* y_absmax) {
    return conv_impl(stream, x, packed, bias, residual, act_ref, y, B, H, W, cin, cout, epilogue, slope, x_absmax, y_absmax);
}

static int bww_dims(int rows, int rb, int cin, int cout, int* nblk, int* IP, int* OP) {
    *nblk = (rows + rb - 1) / rb;
    *IP = cin <= 4 ? 16 : 32;
    *OP = cout <= 16 ? 16 : 32;
    return 0;
}

// image rows per workgroup: 8 for a single tensor; for the batched (all unrolled steps in one launch)
// form enough rows that ~1300-2600 workgroups exist and the partial buffer stays small
static int pick_rb(int rows, int cin = 32, int cout = 32) {
    const bool thin = cin <= 4 || cout <= 4;      // thin layers: 4-wave workgroups with 20 KB of LDS, latency bound per row:
    if (rows > 4096) return thin ? (rows + 1023) / 1024 : (rows + 255) / 256;   // 4 workgroups per CU there; one per CU for the split kernels (they own a CU's LDS)
    return RB;
}

extern "C" size_t sol_conv5x5_bwd_weight_ws_floats(int32_t B, int32_t H, int32_t /*W*/, int32_t cin, int32_t cout) {
    int nblk, IP, OP;
    bww_dims(B * H, RB, cin, cout, &nblk, &IP, &OP);
    return (size_t)nblk * (25 * IP * OP + OP);
}

int sol_bww_pick_rb(int rows) { return pick_rb(rows); }

size_t sol_bww_batched_ws_floats(int nseg, int B, int H, int cin, int cout) {
    int nblk, IP, OP;
    const int rows = nseg * B * H;
    bww_dims(rows, pick_rb(rows, cin, cout), cin, cout, &nblk, &IP, &OP);
    return (size_t)nblk * (25 * IP * OP + OP);
}

static int bww_launch(void* stream, const float* x, const float* dz, float* partial, int nseg, long x_seg, long dz_seg,
                      int rb, int overwrite, int B, int H, int W, int cin, int cout, int nblk_layout = 0,
                      const unsigned* xmax = nullptr, const unsigned* zmax = nullptr, long xmax_seg = 0, long zmax_seg = 0, int cin_real = 0) {
    SOL_REQUIRE(x && dz && partial, "sol_conv5x5_bwd_weight: NULL pointer");
    SOL_REQUIRE(B >= 1 && H >= 1 && W >= 4 && W % 4 == 0 && W <= 64, "sol_conv5x5_bwd_weight: need 4 <= W <= 64, W %% 4 == 0 (got %d)", W);
    SOL_REQUIRE((cin == 4 || cin == 32) && (cout == 2 || cout == 32),
                "sol_conv5x5_bwd_weight: supported (cin,cout) are {4,32}x{2,32} (got %d,%d)", cin, cout);
    BwArgs a{};
    a.x = x; a.dz = dz; a.partial = partial; a.B = B; a.H = H; a.W = W; a.cin = cin; a.cout = cout;
    a.nseg = nseg; a.rb = rb; a.x_seg = x_seg; a.dz_seg = dz_seg; a.overwrite = overwrite;
    a.xmax = xmax; a.zmax = zmax; a.xmax_seg = xmax_seg; a.zmax_seg = zmax_seg; a.cin_real = cin_real;
    int IP, OP;
    bww_dims(nseg * B * H, rb, cin, cout, &a.nblk, &IP, &OP);
    const int nblk_run = a.nblk;                    // workgroups needed for this launch's rows
    if (nblk_layout > 0) a.nblk = nblk_layout;      // partial buffer laid out for a (larger) reference launch
    const int CPX = cin == 4 ? 4 : 48, CPZ = cout <= 4 ? 4 : 48;
    const size_t lds = 2 * ((size_t)(W + 4) * CPX + (size_t)W * CPZ) * sizeof(float);   // double buffered rows
    const int grid = nblk_run * 5;
    hipStream_t s = (hipStream_t)stream;
    if (W == 64 && ((cin == 4 && cout == 32) || (cin == 32 && cout == 2)) && sol_opt().conv_thin) {
        // all five tap rows in one workgroup: grid = nblk; LDS = 2 stages (>= the 16 KB fold buffer)
        if (cin == 4) {
            const size_t l0 = 2 * (size_t)(5 * 68 * 4 + 68 * 32) * sizeof(float);
            // a.cin_real <= 3: the fourth channel of x is zero padding (the karman features): its gradient rows are not computed
            if (a.cin_real >= 1 && a.cin_real <= 3) SOL_LAUNCH((k_conv5x5_bww_thin<0, 1>), dim3(nblk_run), dim3(256), l0, s, a);
            else SOL_LAUNCH((k_conv5x5_bww_thin<0, 2>), dim3(nblk_run), dim3(256), l0, s, a);
        } else {
            const size_t l1 = 2 * (size_t)(68 * 32 + 5 * 72 * 2) * sizeof(float);
            SOL_LAUNCH((k_conv5x5_bww_thin<1, 2>), dim3(nblk_run), dim3(256), l1, s, a);
        }
        SOL_LAUNCH_CHECK();
        return SOL_OK;
    }
    if (cin == 32 && cout == 32 && W == 64 && sol_opt().conv_precision != 2) return sol_bww_sb_launch(s, a, nblk_run);
    if (cin == 32 && cout == 32 && W == 64 && sol_opt().conv_bww32) {
        const size_t lds3 = 2 * ((size_t)(64 + 4) * 32 + 64 * 32) * sizeof(float);
        SOL_LAUNCH(k_conv5x5_bww32, dim3(grid), dim3(256), lds3, s, a);
    }
    else if (cin == 32 && cout == 32) SOL_LAUNCH((k_conv5x5_bww<32, 32>), dim3(grid), dim3(256), lds, s, a);
    else if (cin == 32 && cout == 2) SOL_LAUNCH((k_conv5x5_bww<32, 2>), dim3(grid), dim3(256), lds, s, a);
    else if (cin == 4 && cout == 32) SOL_LAUNCH((k_conv5x5_bww<4, 32>), dim3(grid), dim3(256), lds, s, a);
    else SOL_LAUNCH((k_conv5x5_bww<4, 2>), dim3(grid), dim3(256), lds, s, a);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

extern "C" int sol_conv5x5_bwd_weight(void* stream, const float* x, const float* dz, float* partial,
                                      int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout) {
    return bww_launch(stream, x, dz, partial, 1, 0, 0, RB, 0, B, H, W, cin, cout);
}

// `nseg` unrolled steps of one layer in ONE launch: segment s reads x + s*x_seg, dz + s*dz_seg.  The partial
// buffer is laid out for `nseg_layout` >= nseg segments (several chunk launches accumulate into it:
// overwrite = 1 for the first chunk, 0 afterwards).
int sol_bww_batched(void* stream, const float* x, const float* dz, float* partial, int nseg, int nseg_layout, int overwrite,
                    long x_seg, long dz_seg, int B, int H, int W, int cin, int cout,
                    const unsigned* xmax, const unsigned* zmax, long xmax_seg, long zmax_seg, int cin_real) {
    const int rb = pick_rb(nseg_layout * B * H, cin, cout);
    int nblk, IP, OP;
    bww_dims(nseg_layout * B * H, rb, cin, cout, &nblk, &IP, &OP);
    return bww_launch(stream, x, dz, partial, nseg, x_seg, dz_seg, rb, overwrite, B, H, W, cin, cout, nblk, xmax, zmax, xmax_seg, zmax_seg, cin_real);
}

// n <= 5 passes of the 32 -> 32 split kernels in ONE launch: same blocks, partial layouts and arithmetic as n calls of
// sol_bww_batched(stream, x[k], dz[k], partial[k], 1, 1, overwrite, 0, 0, nplanes[k], H, 64, 32, 32, xmax, zmax, 0, 0)
// rb > 0: image rows per workgroup (every job; >= pick_rb of the job's rows, so that the partial layout fits the buffer sized for pick_rb);
// the reduce must then be called with the same rb
int sol_bww_batched_jobs(void* stream, int n, const float* const* x, const float* const* dz, float* const* partial, const int* nplanes, int overwrite,
                         int H, int W, const unsigned* xmax, const unsigned* zmax, int rb) {
    SOL_REQUIRE(n >= 1 && n <= 5 && W == 64 && H >= 1 && sol_opt().conv_precision != 2, "sol_bww_batched_jobs: 1..5 jobs of the split kernels, W == 64");
    BwJobs p{};
    p.n = n;
    for (int k = 0; k < n; ++k) {
        SOL_REQUIRE(x[k] && dz[k] && partial[k] && nplanes[k] >= 1, "sol_bww_batched_jobs: bad job %d", k);
        BwArgs& a = p.a[k];
        a.x = x[k]; a.dz = dz[k]; a.partial = partial[k]; a.B = nplanes[k]; a.H = H; a.W = W; a.cin = 32; a.cout = 32;
        a.nseg = 1; a.rb = rb > 0 ? rb : pick_rb(nplanes[k] * H); a.x_seg = 0;
        SOL_REQUIRE(a.rb >= pick_rb(nplanes[k] * H), "sol_bww_batched_jobs: %d rows per workgroup is below the layout's %d", a.rb, pick_rb(nplanes[k] * H)); a.dz_seg = 0; a.overwrite = overwrite;
        a.xmax = xmax; a.zmax = zmax; a.xmax_seg = 0; a.zmax_seg = 0; a.cin_real = 0;
        int IP, OP;
        bww_dims(nplanes[k] * H, a.rb, 32, 32, &a.nblk, &IP, &OP);
        p.nrun[k] = a.nblk;
        p.wg_per = p.wg_per > a.nblk ? p.wg_per : a.nblk;
    }
    return sol_bww_sb_jobs_launch((hipStream_t)stream, p);
}

static int bww_reduce(void* stream, const float* partial, float* dw_hwio, float* db, int rows, int rb, int cin, int cout, int accumulate, int tt = 0) {
    SOL_REQUIRE(partial && dw_hwio && db, "sol_conv5x5_bwd_weight_reduce: NULL pointer");
    SOL_REQUIRE(cin >= 1 && cin <= 32 && cout >= 1 && cout <= 32, "sol_conv5x5_bwd_weight_reduce: channels out of range");
    int nblk, IP, OP;
    bww_dims(rows, rb, cin <= 4 ? 4 : 32, cout, &nblk, &IP, &OP);
    const int total = 25 * cin * cout + cout;
    float* pm = const_cast<float*>(partial);       // the caller's workspace: chunk sums are folded in place
    const int chunk = 16, ny = (nblk + chunk - 1) / chunk;
    hipStream_t hs = (hipStream_t)stream;
    if (ny > 1) {
        SOL_LAUNCH(k_bww_reduce, dim3((total + 255) / 256, ny), dim3(256), 0, hs, pm, dw_hwio, db, nblk, cin, cout, IP, OP, chunk, 1, 0, 0, 0);
        SOL_LAUNCH_CHECK();
        SOL_LAUNCH(k_bww_reduce, dim3((total + 255) / 256, 1), dim3(256), 0, hs, pm, dw_hwio, db, nblk, cin, cout, IP, OP, ny, chunk, 1, accumulate, tt);
    } else {
        SOL_LAUNCH(k_bww_reduce, dim3((total + 255) / 256, 1), dim3(256), 0, hs, pm, dw_hwio, db, nblk, cin, cout, IP, OP, nblk, 1, 1, accumulate, tt);
    }
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

extern "C" int sol_conv5x5_bwd_weight_reduce(void* stream, const float* partial, float* dw_hwio, float* db,
                                             int32_t B, int32_t H, int32_t /*W*/, int32_t cin, int32_t cout,
                                             int32_t accumulate) {
    return bww_reduce(stream, partial, dw_hwio, db, B * H, RB, cin, cout, accumulate);
}

// ---- per-step jobs for the fused solver-adjoint + weight-gradient launch (32 -> 32 layers, W == 64) ----------------
// One job = one layer of ONE unrolled step, `rb` image rows per workgroup; the partial buffer has (B*H)/rb blocks and is
// accumulated over the steps of the reverse sweep (overwrite = 1 for the first one), then reduced once.
size_t sol_bww_step_ws_floats(int B, int H, int rb) { return (size_t)((B * H) / rb) * (25 * 32 * 32 + 32); }

int sol_bww_step_job(BwArgs* out, const float* x, const float* dz, float* partial, int overwrite, int B, int H, int W, int rb,
                     const unsigned* xmax, const unsigned* zmax) {
    SOL_REQUIRE(out && x && dz && partial && xmax && zmax, "sol_bww_step_job: NULL pointer");
    SOL_REQUIRE(W == 64 && rb >= 1 && (B * H) % rb == 0, "sol_bww_step_job: W == 64 and rb | B*H required (W=%d, B*H=%d, rb=%d)", W, B * H, rb);
    BwArgs a{};
    a.x = x; a.dz = dz; a.partial = partial; a.B = B; a.H = H; a.W = W; a.cin = 32; a.cout = 32;
    a.nblk = (B * H) / rb; a.nseg = 1; a.rb = rb; a.x_seg = 0; a.dz_seg = 0; a.overwrite = overwrite;
    a.xmax = xmax; a.zmax = zmax; a.xmax_seg = 0; a.zmax_seg = 0;
    *out = a;
    return SOL_OK;
}

// n <= 12 layers at once: rows[i] / rb[i] as in bww_reduce (rb <= 0: pick_rb), same summation order as the per-layer launches
int sol_bww_reduce_layers(void* stream, int n, float* const* partial, float* const* dw_hwio, float* const* db, const int* rows, const int* rb,
                          const int* cin, const int* cout, int accumulate, int taps_transposed) {
    SOL_REQUIRE(n >= 1 && n <= 12, "sol_bww_reduce_layers: 1 <= n <= 12");
    ReduceJobs J{};
    int max_total = 0, max_ny = 1;
    for (int i = 0; i < n; ++i) {
        SOL_REQUIRE(partial[i] && dw_hwio[i] && db[i] && cin[i] >= 1 && cin[i] <= 32 && cout[i] >= 1 && cout[i] <= 32, "sol_bww_reduce_layers: bad layer %d", i);
        ReduceJob& r = J.j[i];
        const int rbi = rb[i] > 0 ? rb[i] : pick_rb(rows[i], cin[i] <= 4 ? 4 : 32, cout[i]);
        bww_dims(rows[i], rbi, cin[i] <= 4 ? 4 : 32, cout[i], &r.nblk, &r.IP, &r.OP);
        r.partial = partial[i]; r.dw = dw_hwio[i]; r.db = db[i]; r.cin = cin[i]; r.cout = cout[i];
        r.ny = (r.nblk + 15) / 16; r.accumulate = accumulate; r.tt = taps_transposed;
        max_total = max_total > 25 * cin[i] * cout[i] + cout[i] ? max_total : 25 * cin[i] * cout[i] + cout[i];
        max_ny = max_ny > r.ny ? max_ny : r.ny;
    }
    hipStream_t hs = (hipStream_t)stream;
    if (max_ny > 1) {
        SOL_LAUNCH(k_bww_reduce_jobs, dim3((max_total + 255) / 256, max_ny, n), dim3(256), 0, hs, J, 0);
        SOL_LAUNCH_CHECK();
    }
    SOL_LAUNCH(k_bww_reduce_jobs, dim3((max_total + 255) / 256, 1, n), dim3(256), 0, hs, J, 1);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

// the reduction of sol_conv5x5_bwd_weight's partial sums for n <= 12 layers of one image size in two launches (same summation order as n calls
// of sol_conv5x5_bwd_weight_reduce); dw_hwio[k] / db[k] may point straight into a flat gradient buffer
extern "C" int sol_conv5x5_bwd_weight_reduce_jobs(void* stream, int32_t n, float* const* partial, float* const* dw_hwio, float* const* db,
                                                  int32_t B, int32_t H, int32_t /*W*/, const int32_t* cin, const int32_t* cout, int32_t accumulate) {
    SOL_REQUIRE(n >= 1 && n <= 12 && partial && dw_hwio && db && cin && cout && B >= 1 && H >= 1, "sol_conv5x5_bwd_weight_reduce_jobs: 1 <= n <= 12 layers, no NULL array");
    int rows[12], rb[12], ci[12], co[12];
    for (int k = 0; k < n; ++k) { rows[k] = B * H; rb[k] = RB; ci[k] = cin[k]; co[k] = cout[k]; }
    return sol_bww_reduce_layers(stream, n, partial, dw_hwio, db, rows, rb, ci, co, accumulate, 0);
}

int sol_bww_step_reduce(void* stream, const float* partial, float* dw_hwio, float* db, int B, int H, int rb, int cin, int cout, int accumulate) {
    return bww_reduce(stream, partial, dw_hwio, db, B * H, rb, cin, cout, accumulate);
}

int sol_bww_batched_reduce(void* stream, const float* partial, float* dw_hwio, float* db, int nseg, int B, int H,
                           int cin, int cout, int accumulate, int taps_transposed) {
    const int rows = nseg * B * H;
    return bww_reduce(stream, partial, dw_hwio, db, rows, pick_rb(rows, cin <= 4 ? 4 : 32, cout), cin, cout, accumulate, taps_transposed);
}
