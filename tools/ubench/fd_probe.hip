// feasibility probe: separable transform on one CU with wave-uniform coefficients in SGPRs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int Y = 128, X = 64;
__device__ __forceinline__ int sw(int m, int n) { return m * 64 + (n ^ (m & 63)); }

// strip-type transform along y: out[r] (rows 16w+r, column n) = sum_k Q[k][16w + r] * in[k][n]   (Q symmetric)
__device__ __forceinline__ void ytrans(const float* __restrict__ Q, const float* buf, int w, int n, f2 (&acc)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = (f2){0.f, 0.f};
    const float* Qw = Q + 16 * w;           // wave uniform
#pragma unroll 4
    for (int k = 0; k < Y; ++k) {
        const float v = buf[sw(k, n)];
        const f2 vv = {v, v};
        const f2* qr = reinterpret_cast<const f2*>(Qw + (size_t)k * Y);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += qr[q] * vv;
    }
}
// row-type transform along x: out[c'] (row m, columns 16cb + c') = sum_n in[m][n] * Q[n][16cb + c']
__device__ __forceinline__ void xtrans(const float* __restrict__ Q, const float* buf, int m, int cb, f2 (&acc)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = (f2){0.f, 0.f};
    const float* Qc = Q + 16 * cb;
#pragma unroll 4
    for (int n = 0; n < X; ++n) {
        const float v = buf[sw(m, n)];
        const f2 vv = {v, v};
        const f2* qr = reinterpret_cast<const f2*>(Qc + (size_t)n * X);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += qr[q] * vv;
    }
}
__global__ void __launch_bounds__(512) k_fd(const float* __restrict__ Qy, const float* __restrict__ Qx, const float* __restrict__ il,
                                            const float* __restrict__ b, float* __restrict__ x, int reps) {
    __shared__ float B0[Y * X], B1[Y * X];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = tid & 127, cb = __builtin_amdgcn_readfirstlane(tid >> 7);
    const float* gb = b + (size_t)blockIdx.x * Y * X;
    float r[16];
    for (int k = 0; k < 16; ++k) r[k] = gb[(16 * w + k) * X + lane];
    f2 acc[8];
    for (int rep = 0; rep < reps; ++rep) {
        for (int k = 0; k < 16; ++k) B0[sw(16 * w + k, lane)] = r[k];
        __syncthreads();
        ytrans(Qy, B0, w, lane, acc);                      // T1 rows 16w.., column lane
        __syncthreads();
        for (int q = 0; q < 8; ++q) { B0[sw(16 * w + 2 * q, lane)] = acc[q].x; B0[sw(16 * w + 2 * q + 1, lane)] = acc[q].y; }
        __syncthreads();
        xtrans(Qx, B0, m, cb, acc);                        // T2 row m, columns 16cb..
        for (int q = 0; q < 8; ++q) {
            const int c = 16 * cb + 2 * q;
            B1[sw(m, c)] = acc[q].x * il[c * Y + m];
            B1[sw(m, c + 1)] = acc[q].y * il[(c + 1) * Y + m];
        }
        __syncthreads();
        xtrans(Qx, B1, m, cb, acc);                        // T3 row m, columns 16cb..
        for (int q = 0; q < 8; ++q) { B0[sw(m, 16 * cb + 2 * q)] = acc[q].x; B0[sw(m, 16 * cb + 2 * q + 1)] = acc[q].y; }
        __syncthreads();
        ytrans(Qy, B0, w, lane, acc);                      // x rows 16w.., column lane
        for (int q = 0; q < 8; ++q) { r[2 * q] = rep + 1 < reps ? r[2 * q] : acc[q].x; r[2 * q + 1] = rep + 1 < reps ? r[2 * q + 1] : acc[q].y; }
        __syncthreads();
    }
    float* gx = x + (size_t)blockIdx.x * Y * X;
    for (int k = 0; k < 16; ++k) gx[(16 * w + k) * X + lane] = r[k];
}
int main() {
    std::vector<double> qy(Y * Y), qx(X * X), lam(Y * X);
    for (int i = 0; i < Y; ++i) for (int j = 0; j < Y; ++j) qy[i * Y + j] = sqrt(2.0 / (Y + 1)) * sin(M_PI * (i + 1) * (j + 1) / (Y + 1));
    for (int i = 0; i < X; ++i) for (int j = 0; j < X; ++j) qx[i * X + j] = sqrt(2.0 / (X + 1)) * sin(M_PI * (i + 1) * (j + 1) / (X + 1));
    std::vector<float> fqy(Y * Y), fqx(X * X), fil(Y * X), fb(6 * Y * X), fx(6 * Y * X);
    for (int i = 0; i < Y * Y; ++i) fqy[i] = (float)qy[i];
    for (int i = 0; i < X * X; ++i) fqx[i] = (float)qx[i];
    for (int m = 0; m < Y; ++m) for (int c = 0; c < X; ++c) {
        lam[m * X + c] = 4 - 2 * cos(M_PI * (m + 1) / (Y + 1)) - 2 * cos(M_PI * (c + 1) / (X + 1));
        fil[c * Y + m] = (float)(1.0 / lam[m * X + c]);
    }
    srand(1);
    for (auto& v : fb) v = rand() / (float)RAND_MAX - 0.5f;
    float *dqy, *dqx, *dil, *db, *dx;
    hipMalloc(&dqy, fqy.size() * 4); hipMalloc(&dqx, fqx.size() * 4); hipMalloc(&dil, fil.size() * 4); hipMalloc(&db, fb.size() * 4); hipMalloc(&dx, fx.size() * 4);
    hipMemcpy(dqy, fqy.data(), fqy.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dqx, fqx.data(), fqx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dil, fil.data(), fil.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, fb.data(), fb.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int reps : {1, 11}) {
        float ms = 0;
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_fd, dim3(6), dim3(512), 0, 0, dqy, dqx, dil, db, dx, reps);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("reps %d: %.2f us\n", reps, ms * 1e3);
    }
    hipMemcpy(fx.data(), dx, fx.size() * 4, hipMemcpyDeviceToHost);
    // residual check: A x = b with the 5-point Dirichlet Laplacian
    double en = 0, bn = 0;
    for (int j = 0; j < Y; ++j) for (int i = 0; i < X; ++i) {
        auto at = [&](int jj, int ii) { return (jj < 0 || jj >= Y || ii < 0 || ii >= X) ? 0.0 : (double)fx[jj * X + ii]; };
        const double ax = 4 * at(j, i) - at(j - 1, i) - at(j + 1, i) - at(j, i - 1) - at(j, i + 1);
        en += (ax - fb[j * X + i]) * (ax - fb[j * X + i]); bn += (double)fb[j * X + i] * fb[j * X + i];
    }
    printf("relative residual %.3e\n", sqrt(en / bn));
    return 0;
}
