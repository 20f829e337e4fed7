// phase stamps of the thin 4 -> 32 convolution (one image row of 64 pixels per 256-thread workgroup)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int B = 6, H = 128, W = 64, OP = 32, NT = 2, HW = 68;
#define STAMP(k) if (tid == 0) st[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime()

template <int VAR>
__global__ void __launch_bounds__(256) k_thin(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                                              float* __restrict__ y, long long* __restrict__ st) {
    __shared__ __align__(16) float smem[1360 + 3200];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.x / H, y0 = blockIdx.x % H;
    STAMP(0);
    float bw4[25][NT];
    const float* wbase = wp + (size_t)li * 4 + g;
#pragma unroll
    for (int tap = 0; tap < 25; ++tap)
#pragma unroll
        for (int n = 0; n < NT; ++n) bw4[tap][n] = VAR == 3 || VAR >= 5 ? (float)(tap + n + lane) : wbase[((size_t)tap * OP + n * 16) * 4];
    if (VAR != 4) {
        const float4* gx = reinterpret_cast<const float4*>(x);
        for (int e = tid; e < 5 * HW; e += 256) {
            const int hr = e / HW, hc = e - hr * HW;
            const int yy = y0 + hr - 2, xx = hc - 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = gx[(size_t)(b * H + yy) * W + xx];
            *reinterpret_cast<float4*>(&smem[e * 4]) = v;
        }
    }
    if (VAR >= 5) {
        const float4* gw = reinterpret_cast<const float4*>(wp);
        for (int e = tid; e < 800; e += 256) *reinterpret_cast<float4*>(&smem[1360 + e * 4]) = gw[e];
    }
    __syncthreads();
    STAMP(1);
    const int q = wave * 16 + li;
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* abase = &smem[q * 4 + g];
    if (VAR == 6) {
        float av[25], bv[25][NT];
#pragma unroll
        for (int tap = 0; tap < 25; ++tap) {
            av[tap] = abase[((tap / 5) * HW + tap % 5) * 4];
#pragma unroll
            for (int n = 0; n < NT; ++n) bv[tap][n] = smem[1360 + (tap * OP + n * 16 + li) * 4 + g];
        }
#pragma unroll
        for (int tap = 0; tap < 25; ++tap)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tap], bv[tap][n], acc[n], 0, 0, 0);
    } else
#pragma unroll
    for (int tap = 0; tap < 25; ++tap) {
        const int dy = tap / 5, dx = tap - dy * 5;
        const float av = abase[(dy * HW + dx) * 4];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float bv = VAR == 5 ? smem[1360 + (tap * OP + n * 16 + li) * 4 + g] : bw4[tap][n];
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[n], 0, 0, 0);
        }
    }
    __syncthreads();
    STAMP(2);
    float* tb = smem + wave * (16 * OP);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const float bs = bias[n * 16 + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[(4 * g + r) * OP + n * 16 + li] = acc[n][r] + bs;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int e = lane + n * 64;
        const int px = e / 8, c4 = e % 8;
        float4 v = *reinterpret_cast<const float4*>(&tb[px * OP + c4 * 4]);
        const size_t o4 = ((size_t)(b * H + y0) * W + wave * 16 + px) * 8 + c4;
        v.x = v.x > 0.f ? v.x : 0.3f * v.x; v.y = v.y > 0.f ? v.y : 0.3f * v.y;
        v.z = v.z > 0.f ? v.z : 0.3f * v.z; v.w = v.w > 0.f ? v.w : 0.3f * v.w;
        if (VAR != 1) reinterpret_cast<float4*>(y)[o4] = v;
        else if (v.x == 12345.f) reinterpret_cast<float4*>(y)[o4] = v;
    }
    STAMP(3);
    if (VAR == 2) { __builtin_amdgcn_s_waitcnt(0); STAMP(4); }
}

template <int VAR>
void run(const char* name, float* x, float* wp, float* bias, float* y, long long* st) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_thin<VAR>, dim3(B * H), dim3(256), 0, 0, x, wp, bias, y, st);
    hipEventRecord(e0);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_thin<VAR>, dim3(B * H), dim3(256), 0, 0, x, wp, bias, y, st);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)B * H * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = h[0];
    for (int w = 0; w < B * H; ++w) t0 = std::min(t0, h[(size_t)w * 8]);
    printf("%-28s %.2f us/launch;", name, ms * 10.f);
    const int ns = VAR == 2 ? 5 : 4;
    for (int k = 0; k < ns; ++k) {
        std::vector<long long> v;
        for (int w = 0; w < B * H; ++w) v.push_back(h[(size_t)w * 8 + k] - t0);
        std::sort(v.begin(), v.end());
        printf("  s%d %.2f/%.2f/%.2f", k, v[0] * 0.01, v[v.size() / 2] * 0.01, v.back() * 0.01);
    }
    printf("  (min/med/max us after first WG start)\n");
}

int main() {
    float *x, *wp, *bias, *y; long long* st;
    hipMalloc(&x, (size_t)B * H * W * 4 * 4); hipMalloc(&wp, 25 * 32 * 4 * 4); hipMalloc(&bias, 128); hipMalloc(&y, (size_t)B * H * W * 32 * 4);
    hipMalloc(&st, (size_t)B * H * 8 * 8);
    hipMemset(x, 0, (size_t)B * H * W * 4 * 4); hipMemset(wp, 0, 25 * 32 * 4 * 4); hipMemset(bias, 0, 128);
    run<0>("baseline", x, wp, bias, y, st);
    run<1>("no stores", x, wp, bias, y, st);
    run<2>("stores + wait", x, wp, bias, y, st);
    run<3>("no weight loads", x, wp, bias, y, st);
    run<4>("no halo loads", x, wp, bias, y, st);
    run<5>("weights through LDS", x, wp, bias, y, st);
    run<6>("LDS weights, preloaded", x, wp, bias, y, st);
    return 0;
}
