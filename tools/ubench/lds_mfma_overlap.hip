// do ds_read_b128 operand reads and 16x16x32 f16 MFMAs overlap on one CU?  12 waves, [6 reads][6 MFMAs] per step as in k_conv5x5_sb
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NRD, int SWZ = 0>     // MODE bit 0: reads, bit 1: MFMAs;  NRD reads per step (6 = product kernel)
__global__ void __launch_bounds__(768) k(float* out, int steps) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int e = tid; e < 100 * 1024 / 16; e += 768) reinterpret_cast<uint4*>(smem)[e] = make_uint4(e, e * 3, e * 5, 0x3c003c00);
    __syncthreads();
    // SWZ: the product kernel's addressing (64-byte pixels, 16-byte chunk index XOR ((idx >> 2) & 1) << 1)
    const int li = lane & 15, g = lane >> 4, hc = ((tid >> 6) & 3) * 16 + li;
    const unsigned char* base = SWZ ? smem + hc * 64 + ((g ^ (((hc >> 2) & 1) << 1)) << 4)
                                    : smem + (lane & 15) * 64 + ((lane >> 4) << 4) + (tid >> 6) * 1024;
    uint4 cur[6], nxt[6];
    for (int q = 0; q < 6; ++q) cur[q] = make_uint4(tid, q, 0x3c003c00, 0x3c003c00);
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    if (MODE & 1) for (int q = 0; q < NRD; ++q) cur[q] = *reinterpret_cast<const uint4*>(base + q * 4352);
    for (int s = 0; s < steps; ++s) {
        if (MODE & 1) {
#pragma unroll
            for (int q = 0; q < NRD; ++q) nxt[q] = *reinterpret_cast<const uint4*>(base + q * 4352 + ((s & 7) << 10));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 2) {
            const f16x8 a1 = __builtin_bit_cast(f16x8, cur[0]), a2 = __builtin_bit_cast(f16x8, cur[1]);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const f16x8 b1 = __builtin_bit_cast(f16x8, cur[2 + 2 * n]), b2 = __builtin_bit_cast(f16x8, cur[3 + 2 * n]);
                acc[2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, acc[2 + n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[n], 0, 0, 0);
                acc[2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, acc[2 + n], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < NRD; ++q) asm volatile("" :: "v"(cur[q].x), "v"(cur[q].y), "v"(cur[q].z), "v"(cur[q].w));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 1) {
#pragma unroll
            for (int q = 0; q < NRD; ++q) cur[q] = nxt[q];
        }
    }
    float r = 0.f;
    for (int n = 0; n < 4; ++n) r += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    for (int q = 0; q < 6; ++q) r += __uint_as_float(cur[q].x & 0xffff);
    if (r == 1234.5f) out[tid] = r;
}

template <int MODE, int NRD, int SWZ = 0>
void run(const char* name, float* out) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NRD, SWZ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int steps = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NRD, SWZ>), dim3(256), dim3(768), 100 * 1024, 0, out, steps);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NRD, SWZ>), dim3(256), dim3(768), 100 * 1024, 0, out, steps);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.1f ns per step (12 waves x [%d reads][6 MFMAs])\n", name, ms * 1e6 / steps, NRD);
}
int main() {
    float* out; hipMalloc(&out, 4096);
    run<1, 6>("reads only", out);
    run<2, 6>("MFMAs only", out);
    run<3, 6>("reads + MFMAs", out);
    run<1, 6, 1>("reads only, swizzled", out);
    run<3, 6, 1>("reads + MFMAs, swizzled", out);
    run<1, 5, 1>("reads only, swizzled, 5/step", out);
    run<3, 5, 1>("reads + MFMAs, swizzled, 5/step", out);
    run<1, 4, 1>("reads only, swizzled, 4/step", out);
    run<3, 4, 1>("reads + MFMAs, swizzled, 4/step", out);
    run<1, 3, 1>("reads only, swizzled, 3/step", out);
    run<3, 3, 1>("reads + MFMAs, swizzled, 3/step", out);
    run<1, 4>("reads only (4 per step)", out);
    run<3, 4>("reads + MFMAs (4 per step)", out);
    run<1, 3>("reads only (3 per step)", out);
    run<3, 3>("reads + MFMAs (3 per step)", out);
    return 0;
}
