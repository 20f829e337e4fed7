// workgroup start-time spread as a function of workgroup size, LDS size and register allocation
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int NV>
__global__ void __launch_bounds__(1024) k_probe(long long* __restrict__ st, float* __restrict__ sink, int work) {
    extern __shared__ float dyn[];
    const long long t = __builtin_amdgcn_s_memrealtime();
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < work; ++it)
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = v[i] * 1.0001f + v[(i + 1) % NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    if (s == 12345.f) { sink[threadIdx.x] = s; dyn[threadIdx.x] = s; }
    if (threadIdx.x == 0) { st[blockIdx.x * 2] = t; st[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}

template <int NV>
void run(int grid, int threads, int lds, int work, long long* st, float* sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(k_probe<NV>, dim3(grid), dim3(threads), lds, 0, st, sink, work); hipDeviceSynchronize(); }
    std::vector<long long> h(grid * 2);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = h[0];
    for (int w = 0; w < grid; ++w) t0 = std::min(t0, h[w * 2]);
    std::vector<double> a, e;
    for (int w = 0; w < grid; ++w) { a.push_back((h[w * 2] - t0) * 0.01); e.push_back((h[w * 2 + 1] - t0) * 0.01); }
    std::sort(a.begin(), a.end()); std::sort(e.begin(), e.end());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_probe<NV>, dim3(grid), dim3(threads), lds, 0, st, sink, work);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("grid %4d x %4d thr, LDS %3d KB, %3d live regs: start med %.2f max %.2f us; end max %.2f us; %.2f us/launch\n",
           grid, threads, lds / 1024, NV, a[grid / 2], a.back(), e.back(), ms * 20.f);
}

int main() {
    long long* st; float* sink;
    hipMalloc(&st, 4096 * 16); hipMalloc(&sink, 4096);
    run<8>(256, 768, 110 * 1024, 1, st, sink);
    run<8>(256, 768, 1024, 1, st, sink);
    run<8>(256, 256, 110 * 1024, 1, st, sink);
    run<8>(256, 256, 1024, 1, st, sink);
    run<8>(768, 256, 1024, 1, st, sink);
    run<8>(768, 256, 50 * 1024, 1, st, sink);
    run<100>(256, 768, 110 * 1024, 1, st, sink);
    run<100>(256, 768, 1024, 1, st, sink);
    run<8>(256, 1024, 110 * 1024, 1, st, sink);
    run<8>(512, 384, 70 * 1024, 1, st, sink);
    return 0;
}
