// fp32 MFMA rate with RANDOM operands that change every instruction (DVFS-realistic), vs constant operands
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int RANDOM>
__global__ void __launch_bounds__(256) k16(const float* in, float* out, int iters) {
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float a[8], b[8];
    for (int k = 0; k < 8; ++k) { a[k] = RANDOM ? in[(threadIdx.x * 8 + k) & 4095] : 1.0f; b[k] = RANDOM ? in[(threadIdx.x * 8 + k + 2048) & 4095] : 0.5f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[k], a[(k + 1) & 7], acc1, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}
template <int RANDOM>
__global__ void __launch_bounds__(256) k32(const float* in, float* out, int iters) {
    f32x16 acc[2];
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0;
    float a[8], b[8];
    for (int k = 0; k < 8; ++k) { a[k] = RANDOM ? in[(threadIdx.x * 8 + k) & 4095] : 1.0f; b[k] = RANDOM ? in[(threadIdx.x * 8 + k + 2048) & 4095] : 0.5f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[k], a[(k + 1) & 7], acc[1], 0, 0, 0);
        }
    }
    float s = 0;
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 768 * 256 * 4);
    float h[4096]; srand(1); for (int i = 0; i < 4096; ++i) h[i] = (rand() / (float)RAND_MAX * 2 - 1) * 0.05f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 4; ++variant) {
        const int iters = 4000;   // x16 MFMAs per wave
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (variant == 0) hipLaunchKernelGGL(k16<0>, dim3(768), dim3(256), 0, 0, in, out, iters);
            if (variant == 1) hipLaunchKernelGGL(k16<1>, dim3(768), dim3(256), 0, 0, in, out, iters);
            if (variant == 2) hipLaunchKernelGGL(k32<0>, dim3(768), dim3(256), 0, 0, in, out, iters);
            if (variant == 3) hipLaunchKernelGGL(k32<1>, dim3(768), dim3(256), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double flop_per = variant >= 2 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 4;
        const double flops = 768.0 * 4 * iters * 16 * flop_per;
        printf("%s %s operands: %.3f ms  %.1f TF\n", variant >= 2 ? "32x32x2" : "16x16x4", (variant & 1) ? "random" : "constant", ms, flops / ms / 1e9);
    }
    return 0;
}
