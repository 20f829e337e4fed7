// bf16 MFMA (gfx950 v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16) rate with constant vs random operands
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union U { bf16x8 v; unsigned u[4]; };
template <int RANDOM>
__global__ void __launch_bounds__(256) k16(const unsigned* in, float* out, int iters) {
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    U a[4], b[4];
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 4; ++j) {
        a[k].u[j] = RANDOM ? in[(threadIdx.x * 16 + k * 4 + j) & 4095] : 0x3f803f80u;
        b[k].u[j] = RANDOM ? in[(threadIdx.x * 16 + k * 4 + j + 2048) & 4095] : 0x3f003f00u;
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k].v, b[k].v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[k].v, a[(k + 1) & 3].v, acc1, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}
template <int RANDOM>
__global__ void __launch_bounds__(256) k32(const unsigned* in, float* out, int iters) {
    f32x16 acc[2];
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0;
    U a[4], b[4];
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 4; ++j) {
        a[k].u[j] = RANDOM ? in[(threadIdx.x * 16 + k * 4 + j) & 4095] : 0x3f803f80u;
        b[k].u[j] = RANDOM ? in[(threadIdx.x * 16 + k * 4 + j + 2048) & 4095] : 0x3f003f00u;
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k].v, b[k].v, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[k].v, a[(k + 1) & 3].v, acc[1], 0, 0, 0);
        }
    }
    float s = 0;
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    unsigned *in; float* out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 768 * 256 * 4);
    unsigned h[4096]; srand(1);
    for (int i = 0; i < 4096; ++i) {   // two random bf16 in [-0.05, 0.05] incl. random low mantissa bits
        unsigned w = 0;
        for (int s = 0; s < 2; ++s) { float f = (rand() / (float)RAND_MAX * 2 - 1) * 0.05f; unsigned u; memcpy(&u, &f, 4); w |= (u >> 16) << (16 * s); }
        h[i] = w;
    }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 4; ++variant) {
        const int iters = 8000;   // x8 MFMAs per wave
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
        const double flop_per = 2.0 * 16 * 16 * 32;   // same for 32x32x16
        const double flops = 768.0 * 4 * iters * 8 * flop_per;
        printf("bf16 %s %s operands: %.3f ms  %.1f TF\n", variant >= 2 ? "32x32x16" : "16x16x32", (variant & 1) ? "random" : "constant", ms, flops / ms / 1e9);
    }
    return 0;
}
