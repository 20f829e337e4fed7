// Forward solver step for grids that do not fit one workgroup's LDS (the reference's data generation runs
// KarmanFlow.step at 256 x 128: /root/reference/karman-2d/karman.py:98-159, Makefile:19-28 `-r 128`).
//
// Same arithmetic as k_karman_fwd (csrc/karman_step.hip), decomposed into chip-wide launches on global memory
// (the whole state of a batch is a few MB, L2 resident):
//   k_l_diffuse   explicit diffusion (replicate padding, dx = 1) + velocity BC              -> sv_y, sv_x
//   k_l_advect    semi-Lagrangian advection of v_y, v_x (clamped) and density (zero ghost ring) + inflow,
//                 hard-BC face masks                                                           -> v_out (pre-projection), d_out
//   k_l_div       rhs = -div
//   pressure      DIRECT solve: x = G (b - U_S E_SS x_S) with the sine-transform diagonalisation of the rectangle and
//                 the capacitance correction of the obstacle (precond.direct_solver_blob, window 16/32/64): eight small
//                 fp32 GEMMs (k_l_gemm) + gather / K' / scatter kernels
//   k_l_project   v -= mask * grad p  (+ fused to_feature)
// Forward only (no saved state, no adjoint): this path generates reference data, it is not trained through.
#include "common.hpp"

namespace {

constexpr int FDL_HEADER = 16;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float acc_at(const float* act, int Y, int X, int j, int i) {   // 'boundary' extrapolation of the active mask
    return act[clampi(j, 0, Y - 1) * X + clampi(i, 0, X - 1)] != 0.f ? 1.f : 0.f;
}
// hard-BC face masks: a face is open iff both cells it separates are accessible (outside the OPEN domain counts as accessible)
__device__ __forceinline__ float mask_y(const float* act, int Y, int X, int j, int i) {   // face between rows j-1 and j
    return acc_at(act, Y, X, j - 1, i) * acc_at(act, Y, X, j, i);
}
__device__ __forceinline__ float mask_x(const float* act, int Y, int X, int j, int i) {   // face between columns i-1 and i
    return acc_at(act, Y, X, j, i - 1) * acc_at(act, Y, X, j, i);
}

struct LArgs {
    int B, Y, X;
    float dtdx, dt, adt;
    int grad_pad, inflow_before;
    const float *d_in, *vy_in, *vx_in, *re, *active, *inflow, *bcv, *bcm;
    long bc_stride;
    float *d_out, *vy_out, *vx_out, *svy, *svx, *rhs, *feat;
    const float* p;
    float fs0, fs1, fs2;
};

__global__ void k_l_diffuse(LArgs a) {
    const int Y = a.Y, X = a.X, XP = X + 1, nVy = (Y + 1) * X, nVx = Y * XP;
    const int b = blockIdx.y;
    const float alpha = a.adt / a.re[b];
    const float* vy = a.vy_in + (size_t)b * nVy;
    const float* vx = a.vx_in + (size_t)b * nVx;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nVy + nVx; k += gridDim.x * blockDim.x) {
        if (k < nVy) {
            const int j = k / X, i = k - j * X;
            const float c = vy[k];
            const float lap = vy[min(j + 1, Y) * X + i] + vy[max(j - 1, 0) * X + i] + vy[j * X + min(i + 1, X - 1)] + vy[j * X + max(i - 1, 0)] - 4.f * c;
            float v = c + alpha * lap;
            v = v * (1.f - a.bcm[(size_t)b * a.bc_stride + k]) + a.bcv[(size_t)b * a.bc_stride + k];
            a.svy[(size_t)b * nVy + k] = v;
        } else {
            const int q = k - nVy, j = q / XP, i = q - j * XP;
            const float c = vx[q];
            const float lap = vx[min(j + 1, Y - 1) * XP + i] + vx[max(j - 1, 0) * XP + i] + vx[j * XP + min(i + 1, X)] + vx[j * XP + max(i - 1, 0)] - 4.f * c;
            a.svx[(size_t)b * nVx + q] = c + alpha * lap;
        }
    }
}

struct Bil { int j0, j1, i0, i1; float wy, wx; };
__device__ __forceinline__ Bil bil_clamp(int H, int W, int jb, float oy, int ib, float ox) {
    Bil s;
    const float fy = floorf(oy), fx = floorf(ox);
    s.wy = oy - fy; s.wx = ox - fx;
    const int j0 = jb + (int)fy, i0 = ib + (int)fx;
    s.j0 = clampi(j0, 0, H - 1); s.j1 = clampi(j0 + 1, 0, H - 1);
    s.i0 = clampi(i0, 0, W - 1); s.i1 = clampi(i0 + 1, 0, W - 1);
    return s;
}
__device__ __forceinline__ float bil_eval(const float* f, int W, const Bil& s) {
    const float f00 = f[s.j0 * W + s.i0], f01 = f[s.j0 * W + s.i1], f10 = f[s.j1 * W + s.i0], f11 = f[s.j1 * W + s.i1];
    return (1.f - s.wy) * ((1.f - s.wx) * f00 + s.wx * f01) + s.wy * ((1.f - s.wx) * f10 + s.wx * f11);
}

__global__ void k_l_advect(LArgs a) {
    const int Y = a.Y, X = a.X, XP = X + 1, N = Y * X, nVy = (Y + 1) * X, nVx = Y * XP;
    const int b = blockIdx.y;
    const float* sy = a.svy + (size_t)b * nVy;
    const float* sx = a.svx + (size_t)b * nVx;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nVy + nVx + N; k += gridDim.x * blockDim.x) {
        if (k < nVy) {
            const int j = k / X, i = k - j * X;
            const float uy = sy[k];
            const int ja = max(j - 1, 0), jb = min(j, Y - 1);
            const float ux = 0.25f * (sx[ja * XP + i] + sx[ja * XP + i + 1] + sx[jb * XP + i] + sx[jb * XP + i + 1]);
            const Bil s = bil_clamp(Y + 1, X, j, -uy * a.dtdx, i, -ux * a.dtdx);
            a.vy_out[(size_t)b * nVy + k] = bil_eval(sy, X, s) * mask_y(a.active, Y, X, j, i);
        } else if (k < nVy + nVx) {
            const int q = k - nVy, j = q / XP, i = q - j * XP;
            const float ux = sx[q];
            const int ia = max(i - 1, 0), ib = min(i, X - 1);
            const float uy = 0.25f * (sy[j * X + ia] + sy[j * X + ib] + sy[(j + 1) * X + ia] + sy[(j + 1) * X + ib]);
            const Bil s = bil_clamp(Y, XP, j, -uy * a.dtdx, i, -ux * a.dtdx);
            a.vx_out[(size_t)b * nVx + q] = bil_eval(sx, XP, s) * mask_x(a.active, Y, X, j, i);
        } else if (a.d_out) {
            const int c = k - nVy - nVx, j = c / X, i = c - j * X;
            const float* gd = a.d_in + (size_t)b * N;
            const float uy = 0.5f * (sy[c] + sy[c + X]);
            const float ux = 0.5f * (sx[j * XP + i] + sx[j * XP + i + 1]);
            const float oy = -uy * a.dtdx, ox = -ux * a.dtdx;
            const float fy = floorf(oy), fx = floorf(ox);
            const float wy = oy - fy, wx = ox - fx;
            const int j0 = j + (int)fy, i0 = i + (int)fx;
            float f[2][2];
            for (int dj = 0; dj < 2; ++dj)
                for (int di = 0; di < 2; ++di) {
                    const int jj = j0 + dj, ii = i0 + di;
                    float v = 0.f;   // extrapolation 'constant': one ring of zero ghost cells
                    if (jj >= 0 && jj < Y && ii >= 0 && ii < X) {
                        v = gd[jj * X + ii];
                        if (a.inflow_before) v += a.inflow[jj * X + ii];
                    }
                    f[dj][di] = v;
                }
            float v = (1.f - wy) * ((1.f - wx) * f[0][0] + wx * f[0][1]) + wy * ((1.f - wx) * f[1][0] + wx * f[1][1]);
            if (!a.inflow_before) v += a.inflow[c] * a.dt;
            a.d_out[(size_t)b * N + c] = v;
        }
    }
}

__global__ void k_l_div(LArgs a) {
    const int Y = a.Y, X = a.X, XP = X + 1, N = Y * X, nVy = (Y + 1) * X, nVx = Y * XP;
    const int b = blockIdx.y;
    const float* vy = a.vy_out + (size_t)b * nVy;
    const float* vx = a.vx_out + (size_t)b * nVx;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < N; c += gridDim.x * blockDim.x) {
        const int j = c / X, i = c - j * X;
        const float div = (vy[(j + 1) * X + i] - vy[j * X + i]) + (vx[j * XP + i + 1] - vx[j * XP + i]);
        a.rhs[(size_t)b * N + c] = -div;        // M p = -div  <=>  A p = div
    }
}

__global__ void k_l_project(LArgs a) {
    const int Y = a.Y, X = a.X, XP = X + 1, N = Y * X, nVy = (Y + 1) * X, nVx = Y * XP;
    const int b = blockIdx.y;
    const float* P = a.p + (size_t)b * N;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nVy + nVx; k += gridDim.x * blockDim.x) {
        if (k < nVy) {
            const int j = k / X, i = k - j * X;
            float g = 0.f;
            if (j >= 1 && j <= Y - 1) g = P[j * X + i] - P[(j - 1) * X + i];
            else if (a.grad_pad == 1) g = (j == 0) ? P[i] : -P[(Y - 1) * X + i];
            const float v = a.vy_out[(size_t)b * nVy + k] - mask_y(a.active, Y, X, j, i) * g;
            a.vy_out[(size_t)b * nVy + k] = v;
            if (a.feat && j < Y) a.feat[((size_t)b * N + k) * 4 + 0] = v * a.fs0;
        } else {
            const int q = k - nVy, j = q / XP, i = q - j * XP;
            float g = 0.f;
            if (i >= 1 && i <= X - 1) g = P[j * X + i] - P[j * X + i - 1];
            else if (a.grad_pad == 1) g = (i == 0) ? P[j * X] : -P[j * X + X - 1];
            const float v = a.vx_out[(size_t)b * nVx + q] - mask_x(a.active, Y, X, j, i) * g;
            a.vx_out[(size_t)b * nVx + q] = v;
            if (a.feat && i < X) {
                float* f = a.feat + ((size_t)b * N + j * X + i) * 4;
                f[1] = v * a.fs1; f[2] = a.re[b] * a.fs2; f[3] = 0.f;
            }
        }
    }
}

// ---- small fp32 GEMM: C[b] (M x N) = (accumulate ? C[b] : 0) + scale (.) (A[b] (M x K) * B[b] (K x N)) ----------
// row-major with leading dimensions; batch strides may be 0 (shared operand); `scale` (M x N, ld = lds) optional.
// 64 x 64 tile per workgroup, 16 x 16 threads x (4 x 4) outputs, K in slabs of 16 through LDS.
struct GArgs {
    const float *A, *Bm, *scale;
    float* C;
    int M, N, K, lda, ldb, ldc, lds;
    long sA, sB, sC;
    int accumulate;
};
__global__ void __launch_bounds__(256) k_l_gemm(GArgs g) {
    __shared__ float As[16][65], Bs[16][65];
    const int b = blockIdx.z, m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const float* A = g.A + (size_t)b * g.sA;
    const float* Bm = g.Bm + (size_t)b * g.sB;
    float* C = g.C + (size_t)b * g.sC;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < g.K; k0 += 16) {
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            const int r = e >> 4, kk = e & 15;              // A tile: 64 rows x 16 k
            const int m = m0 + r, k = k0 + kk;
            As[kk][r] = (m < g.M && k < g.K) ? A[(size_t)m * g.lda + k] : 0.f;
            const int kr = e >> 6, c = e & 63;              // B tile: 16 k x 64 cols
            const int kb = k0 + kr, n = n0 + c;
            Bs[kr][c] = (kb < g.K && n < g.N) ? Bm[(size_t)kb * g.ldb + n] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { av[r] = As[kk][ty * 4 + r]; bv[r] = Bs[kk][tx * 4 + r]; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] += av[r] * bv[c];
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int m = m0 + ty * 4 + r, n = n0 + tx * 4 + c;
            if (m < g.M && n < g.N) {
                float v = acc[r][c];
                if (g.scale) v *= g.scale[(size_t)m * g.lds + n];
                float* dst = &C[(size_t)m * g.ldc + n];
                *dst = g.accumulate ? *dst + v : v;
            }
        }
}

// T[m][c] = (add ? T[m][c] + add[m][c] * il : T[m][c] * il) with il = ilT[c][m] (1 / eigenvalue, stored transposed in the blob)
__global__ void k_l_scale(float* __restrict__ T, const float* __restrict__ add, const float* __restrict__ ilT, int Y, int X) {
    const int b = blockIdx.y;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < Y * X; e += gridDim.x * blockDim.x) {
        const int m = e / X, c = e - m * X;
        const float il = ilT[(size_t)c * Y + m];
        float* t = T + (size_t)b * Y * X + e;
        *t = add ? *t + add[(size_t)b * Y * X + e] * il : *t * il;
    }
}

// capacitance correction on the window values: xs = x0w[sidx], c = K' xs (KpT stored transposed), W2[sidx] = -c
__global__ void k_l_capacitance(const float* __restrict__ x0w, const float* __restrict__ KpT, const int* __restrict__ sidx,
                                float* __restrict__ W2, int SP, int win) {
    extern __shared__ float xs[];
    const int b = blockIdx.x;
    const float* xw = x0w + (size_t)b * 64 * 64;      // per-simulation stride of the window scratch (host carve)
    float* w2 = W2 + (size_t)b * 64 * 64;
    for (int t = threadIdx.x; t < SP; t += blockDim.x) { const int si = sidx[t]; xs[t] = si >= 0 ? xw[si] : 0.f; }
    for (int t = threadIdx.x; t < win * win; t += blockDim.x) w2[t] = 0.f;
    __syncthreads();
    for (int s = threadIdx.x; s < SP; s += blockDim.x) {
        const int si = sidx[s];
        if (si < 0) continue;
        float c = 0.f;
        for (int q = 0; q < SP; ++q) c += KpT[(size_t)q * SP + s] * xs[q];
        w2[si] = -c;
    }
}

struct Header { int Y, X, wy0, wx0, nS, SP, win; };

int gemm(hipStream_t s, int batch, const float* A, int lda, long sA, const float* Bm, int ldb, long sB, float* C, int ldc, long sC,
         int M, int N, int K, int accumulate) {
    GArgs g{A, Bm, nullptr, C, M, N, K, lda, ldb, ldc, 0, sA, sB, sC, accumulate};
    SOL_LAUNCH(k_l_gemm, dim3((N + 63) / 64, (M + 63) / 64, batch), dim3(256), 0, s, g);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

}  // namespace

// shared with karman3d.hip (the sine transforms of the 3-D direct solve are batched products of this kind)
int sol_gemm_f32(hipStream_t s, int batch, const float* A, int lda, long sA, const float* Bm, int ldb, long sB, float* C, int ldc, long sC,
                 int M, int N, int K, int accumulate) {
    return gemm(s, batch, A, lda, sA, Bm, ldb, sB, C, ldc, sC, M, N, K, accumulate);
}

extern "C" size_t sol_karman_step_large_workspace_bytes(const sol_karman_cfg* c) {
    if (!c) return 0;
    const size_t B = c->B, Y = c->Y, X = c->X;
    // sv_y, sv_x, three N-sized buffers (rhs / transforms), window scratch (u, t2w: Y*64 each; x0w, W2: 64*64 each)
    const size_t floats = B * ((Y + 1) * X + Y * (X + 1) + 3 * Y * X + 2 * Y * 64 + 2 * 64 * 64) + 256;
    return floats * sizeof(float);
}

extern "C" int sol_karman_step_fwd_large(const sol_karman_cfg* c, void* stream,
                                         const float* d_in, const float* vy_in, const float* vx_in,
                                         const float* re, const float* active, const float* inflow,
                                         const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                                         float* d_out, float* vy_out, float* vx_out,
                                         float* feat_out, const float* feat_scale,
                                         const int32_t* direct_header_host,
                                         void* workspace, size_t workspace_bytes) {
    SOL_REQUIRE(c != nullptr, "cfg is NULL");
    SOL_REQUIRE(c->B >= 1 && c->Y >= 16 && c->X >= 16, "sol_karman_step_fwd_large: B >= 1, Y, X >= 16 (got %d, %d, %d)", c->B, c->Y, c->X);
    SOL_REQUIRE(vy_in && vx_in && re && active && velBCy && velBCyMask && vy_out && vx_out && workspace, "sol_karman_step_fwd_large: NULL pointer argument");
    SOL_REQUIRE((d_in && inflow) || !d_out, "density output requested without d_in/inflow");
    SOL_REQUIRE(!feat_out || feat_scale, "feat_out requires feat_scale");
    SOL_REQUIRE(c->direct && c->direct_n > 0, "sol_karman_step_fwd_large needs the direct-solver blob (cfg.direct)");
    SOL_REQUIRE(workspace_bytes >= sol_karman_step_large_workspace_bytes(c), "workspace too small");
    SOL_REQUIRE(vy_in != vy_out && vx_in != vx_out && d_in != d_out, "sol_karman_step_fwd_large: outputs must not alias the inputs");
    SOL_REQUIRE(direct_header_host && direct_header_host[0] == 0x46443032, "sol_karman_step_fwd_large: direct_header_host must be the first 16 words of the blob (host copy)");
    const Header h{direct_header_host[1], direct_header_host[2], direct_header_host[3], direct_header_host[4],
                   direct_header_host[5], direct_header_host[6], direct_header_host[7]};
    SOL_REQUIRE(h.SP >= h.nS && h.SP <= 4096 && h.wy0 >= 0 && h.wx0 >= 0 && h.wy0 + h.win <= c->Y && h.wx0 + h.win <= c->X,
                "direct-solver blob header is inconsistent");
    SOL_REQUIRE(h.Y == c->Y && h.X == c->X, "direct-solver blob is for a %dx%d grid, cfg is %dx%d", h.Y, h.X, c->Y, c->X);
    SOL_REQUIRE(h.win == 16 || h.win == 32 || h.win == 64, "direct-solver blob: unsupported window %d", h.win);
    const int B = c->B, Y = c->Y, X = c->X, N = Y * X, win = h.win, SP = h.SP;
    hipStream_t s = (hipStream_t)stream;
    // blob sections
    const float* Qy = c->direct + FDL_HEADER;
    const float* Qx = Qy + (size_t)Y * Y;
    const float* ilT = Qx + (size_t)X * X;             // [X][Y]
    const float* KpT = ilT + (size_t)X * Y;
    const int* sidx = reinterpret_cast<const int*>(KpT + (size_t)SP * SP);
    const float* QxW = reinterpret_cast<const float*>(sidx + SP);     // [X][win]
    // workspace carve
    float* w = static_cast<float*>(workspace);
    float* svy = w; w += (size_t)B * (Y + 1) * X;
    float* svx = w; w += (size_t)B * Y * (X + 1);
    float* T0 = w; w += (size_t)B * N;                 // rhs / T3 / pressure
    float* T1 = w; w += (size_t)B * N;
    float* T2 = w; w += (size_t)B * N;                 // spectral coefficients, stored TRANSPOSED [X][Y] (matches ilT)
    float* U = w; w += (size_t)B * Y * 64;             // [Y][win]
    float* V = w; w += (size_t)B * Y * 64;
    float* X0 = w; w += (size_t)B * 64 * 64;           // [win][win]
    float* W2 = w; w += (size_t)B * 64 * 64;

    LArgs a{};
    a.B = B; a.Y = Y; a.X = X; a.dtdx = c->dt / c->dx; a.dt = c->dt; a.adt = c->dt * c->res * c->res;
    a.grad_pad = c->grad_pad; a.inflow_before = c->inflow_before;
    a.d_in = d_in; a.vy_in = vy_in; a.vx_in = vx_in; a.re = re; a.active = active; a.inflow = inflow;
    a.bcv = velBCy; a.bcm = velBCyMask; a.bc_stride = bc_batch_stride;
    a.d_out = d_out; a.vy_out = vy_out; a.vx_out = vx_out; a.svy = svy; a.svx = svx; a.rhs = T0; a.feat = feat_out; a.p = T0;
    if (feat_scale) { a.fs0 = feat_scale[0]; a.fs1 = feat_scale[1]; a.fs2 = feat_scale[2]; }
    const int faces = (Y + 1) * X + Y * (X + 1);
    SOL_LAUNCH(k_l_diffuse, dim3((faces + 255) / 256, B), dim3(256), 0, s, a);
    SOL_LAUNCH(k_l_advect, dim3((faces + N + 255) / 256, B), dim3(256), 0, s, a);
    SOL_LAUNCH(k_l_div, dim3((N + 255) / 256, B), dim3(256), 0, s, a);
    SOL_LAUNCH_CHECK();

    // ---- direct pressure solve.  All matrices row-major [Y][X] per simulation; Qy, Qx symmetric.
    const long sN = N, sU = (long)Y * 64, sW = 64 * 64;
    // T1 = Qy rhs ;  T2 = (T1 Qx) / lam
    if (int e = gemm(s, B, Qy, Y, 0, T0, X, sN, T1, X, sN, Y, X, Y, 0)) return e;
    if (int e = gemm(s, B, T1, X, sN, Qx, X, 0, T2, X, sN, Y, X, X, 0)) return e;
    SOL_LAUNCH(k_l_scale, dim3((N + 255) / 256, B), dim3(256), 0, s, T2, (const float*)nullptr, ilT, Y, X);
    // window values of G b: U = T2 Qx[:, win] ; X0 = Qy[win, :] U
    if (int e = gemm(s, B, T2, X, sN, QxW, win, 0, U, win, sU, Y, win, X, 0)) return e;
    if (int e = gemm(s, B, Qy + (size_t)h.wy0 * Y, Y, 0, U, win, sU, X0, win, sW, win, win, Y, 0)) return e;
    // W2 = -scatter(K' gather(X0))
    SOL_LAUNCH(k_l_capacitance, dim3(B), dim3(256), SP * sizeof(float), s, X0, KpT, sidx, W2, SP, win);
    SOL_LAUNCH_CHECK();
    // spectral coefficients of the correction: V = Qy[:, win] W2 ; T2 += ((V Qx[win, :])) / lam
    if (int e = gemm(s, B, Qy + h.wy0, Y, 0, W2, win, sW, V, win, sU, Y, win, win, 0)) return e;
    if (int e = gemm(s, B, V, win, sU, Qx + (size_t)h.wx0 * X, X, 0, T1, X, sN, Y, X, win, 0)) return e;
    SOL_LAUNCH(k_l_scale, dim3((N + 255) / 256, B), dim3(256), 0, s, T2, (const float*)T1, ilT, Y, X);
    // p = Qy (T2 Qx)
    if (int e = gemm(s, B, T2, X, sN, Qx, X, 0, T1, X, sN, Y, X, X, 0)) return e;
    if (int e = gemm(s, B, Qy, Y, 0, T1, X, sN, T0, X, sN, Y, X, Y, 0)) return e;
    SOL_LAUNCH(k_l_project, dim3((faces + 255) / 256, B), dim3(256), 0, s, a);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}
