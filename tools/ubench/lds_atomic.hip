// micro-benchmark: LDS atomic add throughput, float vs int32 vs uint64, 512 threads, stride-1 addresses + offset
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    __shared__ float sf[16384];
    int* si = reinterpret_cast<int*>(sf);
    unsigned long long* sl = reinterpret_cast<unsigned long long*>(sf);
    for (int i = threadIdx.x; i < 16384; i += 512) sf[i] = 0.f;
    __syncthreads();
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a = (idx + u * 67 + it * 131) & 8191;
            if (MODE == 0) atomicAdd(&sf[a], 1.0f + u);
            else if (MODE == 1) atomicAdd(&si[a], 1 + u);
            else if (MODE == 2) atomicAdd(&sl[a], (unsigned long long)(1 + u));
            else sf[a] += 1.0f + u;   // non atomic reference (racy)
        }
    }
    __syncthreads();
    float s = 0;
    for (int i = threadIdx.x; i < 16384; i += 512) s += sf[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 512 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(6), dim3(512), 0, 0, out, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(6), dim3(512), 0, 0, out, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(6), dim3(512), 0, 0, out, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(6), dim3(512), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double ops = 512.0 * iters * 8;
        printf("mode %d (%s): %.3f ms, %.2f lane-ops per ns per CU\n", mode, mode == 0 ? "ds_add_f32" : mode == 1 ? "ds_add_u32" : mode == 2 ? "ds_add_u64" : "plain rmw", ms, ops / (ms * 1e6));
    }
    return 0;
}
