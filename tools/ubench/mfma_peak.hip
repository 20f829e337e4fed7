// micro-benchmark: attainable fp32 MFMA rate (16x16x4 and 32x32x2) and shader clock on this box
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k16(float* out, long long* clk, int iters) {
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void __launch_bounds__(256) k32(float* out, long long* clk, int iters) {
    f32x16 acc[2];
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
    float* out; long long* clk;
    hipMalloc(&out, 256 * 1024 * 4 * 4); hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 768}) {
        for (int variant = 0; variant < 3; ++variant) {
            const int iters = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (variant == 0) hipLaunchKernelGGL(k16<2>, dim3(wgs), dim3(256), 0, 0, out, clk, iters);
                else if (variant == 1) hipLaunchKernelGGL(k16<4>, dim3(wgs), dim3(256), 0, 0, out, clk, iters);
                else hipLaunchKernelGGL(k32, dim3(wgs), dim3(256), 0, 0, out, clk, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
            const int nacc = variant == 0 ? 2 : (variant == 1 ? 4 : 2);
            const double flop_per = variant == 2 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 4;
            const double flops = (double)wgs * 4 * iters * nacc * flop_per;
            printf("wgs %d %s: %.3f ms  %.1f TF  clock64 ticks %lld (%.1f ticks per MFMA per wave; tick rate %.3f GHz if cycles)\n", wgs,
                   variant == 0 ? "16x16x4 x2acc" : (variant == 1 ? "16x16x4 x4acc" : "32x32x2 x2acc"), ms, flops / ms / 1e9,
                   c, (double)c / (iters * nacc), c / (ms * 1e6));
        }
    }
    return 0;
}
