// Burgers step on the periodic staggered grid and its adjoint  --  gfx950 / CDNA4.
//
// Replaces BurgersTest.step / step_with_f (/root/reference/burgers/burgers_train.py:182-187)
// -> PhiFlow Burgers.step: v = semi_lagrangian(v, v, dt); v = diffuse(v, dt*nu) (periodic
// domain -> spectral branch); then v += dt*f.  One workgroup per simulation, all in LDS.
//
// The spectral diffusion  ifft(fft(f) * exp(-(2 pi)^2 |k|^2 a))  is separable and real, so it
// is applied as two small dense products with symmetric circulant matrices prepared by the
// host (out = Cy * f * Cx^T): no FFT library, and the adjoint is the same operator.
// PhiFlow-1.x quirk kept on purpose: the periodic staggered
// components keep the duplicated +1 face and every wrap-around is modulo the ARRAY length.
#include "common.hpp"

namespace {

constexpr int NT = 256;
constexpr int MAXTB = 17;   // ceil(65*64 / 256)

struct BArgs {
    int B, Y, X;
    float dtdx, dt;
    const float *vy_in, *vx_in, *fy, *fx, *cyp1, *cx, *cy, *cxp1, *g_vy_out, *g_vx_out;
    float *vy_out, *vx_out, *g_vy_in, *g_vx_in;
};

__host__ __device__ inline int al4b(int n) { return (n + 3) & ~3; }
__host__ inline size_t blds_bytes(int Y, int X) {
    const int nVy = (Y + 1) * X, nVx = Y * (X + 1);
    const int nmax = nVy > nVx ? nVy : nVx;
    return (size_t)(2 * (al4b(nVy) + al4b(nVx)) + al4b(nmax) + al4b((Y + 1) * (Y + 1)) + al4b(X * X) +
                    al4b(Y * Y) + al4b((X + 1) * (X + 1))) * sizeof(float);
}

struct BLds {
    float *Avy, *Avx, *Bvy, *Bvx, *T, *Cyp1, *Cx, *Cy, *Cxp1;
};
__device__ inline BLds bcarve(float* s, int Y, int X) {
    const int nVy = (Y + 1) * X, nVx = Y * (X + 1);
    const int nmax = nVy > nVx ? nVy : nVx;
    BLds l;
    l.Avy = s; l.Avx = l.Avy + al4b(nVy);
    l.Bvy = l.Avx + al4b(nVx); l.Bvx = l.Bvy + al4b(nVy);
    l.T = l.Bvx + al4b(nVx);
    l.Cyp1 = l.T + al4b(nmax);
    l.Cx = l.Cyp1 + al4b((Y + 1) * (Y + 1));
    l.Cy = l.Cx + al4b(X * X);
    l.Cxp1 = l.Cy + al4b(Y * Y);
    return l;
}

__device__ __forceinline__ int wrap(int v, int n) {
    v %= n;
    return v < 0 ? v + n : v;
}

struct BilP {
    int j0, j1, i0, i1;
    float wy, wx;
};
__device__ __forceinline__ BilP bil_wrap(int H, int W, int jb, float oy, int ib, float ox) {
    BilP s;
    const float fy = floorf(oy), fx = floorf(ox);
    s.wy = oy - fy;
    s.wx = ox - fx;
    const int j0 = jb + (int)fy, i0 = ib + (int)fx;
    s.j0 = wrap(j0, H); s.j1 = wrap(j0 + 1, H);
    s.i0 = wrap(i0, W); s.i1 = wrap(i0 + 1, W);
    return s;
}

// out[H,W] = Cl[H,H] * (in[H,W] * Cr[W,W]^T), T is scratch [H,W]; out may alias in.
__device__ void diffuse2(const float* in, float* out, float* T, const float* Cl, const float* Cr, int H, int W) {
    for (int e = threadIdx.x; e < H * W; e += NT) {
        const int j = e / W, i = e - j * W;
        float s = 0.f;
        for (int k = 0; k < W; ++k) s += in[j * W + k] * Cr[i * W + k];
        T[e] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < H * W; e += NT) {
        const int j = e / W, i = e - j * W;
        float s = 0.f;
        for (int k = 0; k < H; ++k) s += Cl[j * H + k] * T[k * W + i];
        out[e] = s;
    }
    __syncthreads();
}

__device__ void load_common(const BArgs& a, const BLds& L) {
    const int Y = a.Y, X = a.X;
    for (int e = threadIdx.x; e < (Y + 1) * (Y + 1); e += NT) L.Cyp1[e] = a.cyp1[e];
    for (int e = threadIdx.x; e < X * X; e += NT) L.Cx[e] = a.cx[e];
    for (int e = threadIdx.x; e < Y * Y; e += NT) L.Cy[e] = a.cy[e];
    for (int e = threadIdx.x; e < (X + 1) * (X + 1); e += NT) L.Cxp1[e] = a.cxp1[e];
}

__global__ void __launch_bounds__(NT) k_burgers_fwd(BArgs a) {
    extern __shared__ __align__(16) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Y = a.Y, X = a.X, nVy = (Y + 1) * X, nVx = Y * (X + 1), XP = X + 1;
    const BLds L = bcarve(smem, Y, X);
    for (int k = tid; k < nVy; k += NT) L.Avy[k] = a.vy_in[(size_t)b * nVy + k];
    for (int k = tid; k < nVx; k += NT) L.Avx[k] = a.vx_in[(size_t)b * nVx + k];
    load_common(a, L);
    __syncthreads();
    // semi-Lagrangian self-advection (A -> B), wrap modulo the array length
    for (int k = tid; k < nVy; k += NT) {
        const int j = k / X, i = k - j * X;
        const float uy = L.Avy[k];
        const int ja = wrap(j - 1, Y), jb = wrap(j, Y);
        const float ux = 0.25f * (L.Avx[ja * XP + i] + L.Avx[ja * XP + i + 1] + L.Avx[jb * XP + i] + L.Avx[jb * XP + i + 1]);
        const BilP s = bil_wrap(Y + 1, X, j, -uy * a.dtdx, i, -ux * a.dtdx);
        L.Bvy[k] = (1.f - s.wy) * ((1.f - s.wx) * L.Avy[s.j0 * X + s.i0] + s.wx * L.Avy[s.j0 * X + s.i1]) +
                   s.wy * ((1.f - s.wx) * L.Avy[s.j1 * X + s.i0] + s.wx * L.Avy[s.j1 * X + s.i1]);
    }
    for (int k = tid; k < nVx; k += NT) {
        const int j = k / XP, i = k - j * XP;
        const float ux = L.Avx[k];
        const int ia = wrap(i - 1, X), ib = wrap(i, X);
        const float uy = 0.25f * (L.Avy[j * X + ia] + L.Avy[j * X + ib] + L.Avy[(j + 1) * X + ia] + L.Avy[(j + 1) * X + ib]);
        const BilP s = bil_wrap(Y, XP, j, -uy * a.dtdx, i, -ux * a.dtdx);
        L.Bvx[k] = (1.f - s.wy) * ((1.f - s.wx) * L.Avx[s.j0 * XP + s.i0] + s.wx * L.Avx[s.j0 * XP + s.i1]) +
                   s.wy * ((1.f - s.wx) * L.Avx[s.j1 * XP + s.i0] + s.wx * L.Avx[s.j1 * XP + s.i1]);
    }
    __syncthreads();
    diffuse2(L.Bvy, L.Bvy, L.T, L.Cyp1, L.Cx, Y + 1, X);
    diffuse2(L.Bvx, L.Bvx, L.T, L.Cy, L.Cxp1, Y, XP);
    for (int k = tid; k < nVy; k += NT) {
        float v = L.Bvy[k];
        if (a.fy) v += a.dt * a.fy[(size_t)b * nVy + k];
        a.vy_out[(size_t)b * nVy + k] = v;
    }
    for (int k = tid; k < nVx; k += NT) {
        float v = L.Bvx[k];
        if (a.fx) v += a.dt * a.fx[(size_t)b * nVx + k];
        a.vx_out[(size_t)b * nVx + k] = v;
    }
}

__global__ void __launch_bounds__(NT) k_burgers_bwd(BArgs a) {
    extern __shared__ __align__(16) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Y = a.Y, X = a.X, nVy = (Y + 1) * X, nVx = Y * (X + 1), XP = X + 1;
    const BLds L = bcarve(smem, Y, X);
    for (int k = tid; k < nVy; k += NT) { L.Avy[k] = a.vy_in[(size_t)b * nVy + k]; L.Bvy[k] = a.g_vy_out[(size_t)b * nVy + k]; }
    for (int k = tid; k < nVx; k += NT) { L.Avx[k] = a.vx_in[(size_t)b * nVx + k]; L.Bvx[k] = a.g_vx_out[(size_t)b * nVx + k]; }
    load_common(a, L);
    __syncthreads();
    // diffusion adjoint = same symmetric operator
    diffuse2(L.Bvy, L.Bvy, L.T, L.Cyp1, L.Cx, Y + 1, X);
    diffuse2(L.Bvx, L.Bvx, L.T, L.Cy, L.Cxp1, Y, XP);
    float gy[MAXTB], gx[MAXTB];
#pragma unroll
    for (int n = 0; n < MAXTB; ++n) {
        const int k = tid + n * NT;
        gy[n] = k < nVy ? L.Bvy[k] : 0.f;
        gx[n] = k < nVx ? L.Bvx[k] : 0.f;
    }
    __syncthreads();
    // int32 fixed-point scatter (ds_add_f32 is ~37x slower than ds_add_u32 on gfx950; see karman_step.hip)
    int* Iy = reinterpret_cast<int*>(L.Bvy);
    int* Ix = reinterpret_cast<int*>(L.Bvx);
    float smax = 0.f, gmax = 0.f;
    for (int k = tid; k < nVy; k += NT) { Iy[k] = 0; smax = fmaxf(smax, fabsf(L.Avy[k])); }
    for (int k = tid; k < nVx; k += NT) { Ix[k] = 0; smax = fmaxf(smax, fabsf(L.Avx[k])); }
#pragma unroll
    for (int n = 0; n < MAXTB; ++n) gmax = fmaxf(gmax, fmaxf(fabsf(gy[n]), fabsf(gx[n])));
    {   // workgroup max through the (free) scratch row T
        for (int off = 32; off > 0; off >>= 1) { smax = fmaxf(smax, __shfl_xor(smax, off, 64)); gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64)); }
        if ((tid & 63) == 0) { L.T[tid >> 6] = smax; L.T[8 + (tid >> 6)] = gmax; }
        __syncthreads();
        smax = fmaxf(fmaxf(L.T[0], L.T[1]), fmaxf(L.T[2], L.T[3]));
        gmax = fmaxf(fmaxf(L.T[8], L.T[9]), fmaxf(L.T[10], L.T[11]));
    }
    const float bound = 16.f * gmax * fmaxf(1.f, 2.f * a.dtdx * smax);
    const float qs = bound > 0.f ? 2147483648.f / bound : 0.f, qi = bound > 0.f ? bound / 2147483648.f : 0.f;
    auto fx = [&](float v) { return __float2int_rn(v * qs); };
#pragma unroll
    for (int n = 0; n < MAXTB; ++n) {
        const int k = tid + n * NT;
        if (k < nVy && gy[n] != 0.f) {
            const int j = k / X, i = k - j * X;
            const float g = gy[n];
            const float uy = L.Avy[k];
            const int ja = wrap(j - 1, Y), jb = wrap(j, Y);
            const float ux = 0.25f * (L.Avx[ja * XP + i] + L.Avx[ja * XP + i + 1] + L.Avx[jb * XP + i] + L.Avx[jb * XP + i + 1]);
            const BilP s = bil_wrap(Y + 1, X, j, -uy * a.dtdx, i, -ux * a.dtdx);
            const float f00 = L.Avy[s.j0 * X + s.i0], f01 = L.Avy[s.j0 * X + s.i1];
            const float f10 = L.Avy[s.j1 * X + s.i0], f11 = L.Avy[s.j1 * X + s.i1];
            atomicAdd(&Iy[s.j0 * X + s.i0], fx((1.f - s.wy) * (1.f - s.wx) * g));
            atomicAdd(&Iy[s.j0 * X + s.i1], fx((1.f - s.wy) * s.wx * g));
            atomicAdd(&Iy[s.j1 * X + s.i0], fx(s.wy * (1.f - s.wx) * g));
            atomicAdd(&Iy[s.j1 * X + s.i1], fx(s.wy * s.wx * g));
            const float ddy = (1.f - s.wx) * (f10 - f00) + s.wx * (f11 - f01);
            const float ddx = (1.f - s.wy) * (f01 - f00) + s.wy * (f11 - f10);
            const float guy = -a.dtdx * g * ddy, gux = -0.25f * a.dtdx * g * ddx;
            atomicAdd(&Iy[k], fx(guy));
            atomicAdd(&Ix[ja * XP + i], fx(gux));
            atomicAdd(&Ix[ja * XP + i + 1], fx(gux));
            atomicAdd(&Ix[jb * XP + i], fx(gux));
            atomicAdd(&Ix[jb * XP + i + 1], fx(gux));
        }
        if (k < nVx && gx[n] != 0.f) {
            const int j = k / XP, i = k - j * XP;
            const float g = gx[n];
            const float ux = L.Avx[k];
            const int ia = wrap(i - 1, X), ib = wrap(i, X);
            const float uy = 0.25f * (L.Avy[j * X + ia] + L.Avy[j * X + ib] + L.Avy[(j + 1) * X + ia] + L.Avy[(j + 1) * X + ib]);
            const BilP s = bil_wrap(Y, XP, j, -uy * a.dtdx, i, -ux * a.dtdx);
            const float f00 = L.Avx[s.j0 * XP + s.i0], f01 = L.Avx[s.j0 * XP + s.i1];
            const float f10 = L.Avx[s.j1 * XP + s.i0], f11 = L.Avx[s.j1 * XP + s.i1];
            atomicAdd(&Ix[s.j0 * XP + s.i0], fx((1.f - s.wy) * (1.f - s.wx) * g));
            atomicAdd(&Ix[s.j0 * XP + s.i1], fx((1.f - s.wy) * s.wx * g));
            atomicAdd(&Ix[s.j1 * XP + s.i0], fx(s.wy * (1.f - s.wx) * g));
            atomicAdd(&Ix[s.j1 * XP + s.i1], fx(s.wy * s.wx * g));
            const float ddy = (1.f - s.wx) * (f10 - f00) + s.wx * (f11 - f01);
            const float ddx = (1.f - s.wy) * (f01 - f00) + s.wy * (f11 - f10);
            const float gux = -a.dtdx * g * ddx, guy = -0.25f * a.dtdx * g * ddy;
            atomicAdd(&Ix[k], fx(gux));
            atomicAdd(&Iy[j * X + ia], fx(guy));
            atomicAdd(&Iy[j * X + ib], fx(guy));
            atomicAdd(&Iy[(j + 1) * X + ia], fx(guy));
            atomicAdd(&Iy[(j + 1) * X + ib], fx(guy));
        }
    }
    __syncthreads();
    for (int k = tid; k < nVy; k += NT) L.Bvy[k] = (float)Iy[k] * qi;
    for (int k = tid; k < nVx; k += NT) L.Bvx[k] = (float)Ix[k] * qi;
    __syncthreads();
    for (int k = tid; k < nVy; k += NT) a.g_vy_in[(size_t)b * nVy + k] = L.Bvy[k];
    for (int k = tid; k < nVx; k += NT) a.g_vx_in[(size_t)b * nVx + k] = L.Bvx[k];
}

// ------------------------------------------------------------------------------------
// Large grids (forward only): the reference generates its Burgers training data at 128 x 128 (burgers/Makefile:19-29,
// `burgers.py -r 128`), beyond the one-workgroup-in-LDS kernels above.  Same arithmetic on global memory with the whole chip:
// one launch for the advection, two tiled products for the separable circulant diffusion per component, force added by the
// second.  Not differentiable (the reference does not train through its hi-res data either).
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_burgers_adv_large(int B, int Y, int X, float dtdx, const float* __restrict__ vy_in,
                                                           const float* __restrict__ vx_in, float* __restrict__ ay, float* __restrict__ ax) {
    const int nVy = (Y + 1) * X, nVx = Y * (X + 1), XP = X + 1;
    const int total = B * (nVy + nVx);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int b = e / (nVy + nVx), r = e - b * (nVy + nVx);
        const float* Vy = vy_in + (size_t)b * nVy;
        const float* Vx = vx_in + (size_t)b * nVx;
        if (r < nVy) {
            const int j = r / X, i = r - j * X;
            const float uy = Vy[r];
            const int ja = wrap(j - 1, Y), jb = wrap(j, Y);
            const float ux = 0.25f * (Vx[ja * XP + i] + Vx[ja * XP + i + 1] + Vx[jb * XP + i] + Vx[jb * XP + i + 1]);
            const BilP s = bil_wrap(Y + 1, X, j, -uy * dtdx, i, -ux * dtdx);
            ay[(size_t)b * nVy + r] = (1.f - s.wy) * ((1.f - s.wx) * Vy[s.j0 * X + s.i0] + s.wx * Vy[s.j0 * X + s.i1]) +
                                      s.wy * ((1.f - s.wx) * Vy[s.j1 * X + s.i0] + s.wx * Vy[s.j1 * X + s.i1]);
        } else {
            const int k = r - nVy, j = k / XP, i = k - j * XP;
            const float ux = Vx[k];
            const int ia = wrap(i - 1, X), ib = wrap(i, X);
            const float uy = 0.25f * (Vy[j * X + ia] + Vy[j * X + ib] + Vy[(j + 1) * X + ia] + Vy[(j + 1) * X + ib]);
            const BilP s = bil_wrap(Y, XP, j, -uy * dtdx, i, -ux * dtdx);
            ax[(size_t)b * nVx + k] = (1.f - s.wy) * ((1.f - s.wx) * Vx[s.j0 * XP + s.i0] + s.wx * Vx[s.j0 * XP + s.i1]) +
                                      s.wy * ((1.f - s.wx) * Vx[s.j1 * XP + s.i0] + s.wx * Vx[s.j1 * XP + s.i1]);
        }
    }
}

// One 16 x 16 output tile per workgroup of C[b] = L * In[b] * R^T restricted to one side per launch:
//   SIDE 0:  out[j][i] = sum_k in[j][k] * R[i][k]          (in [H][W], R [W][W])
//   SIDE 1:  out[j][i] = sum_k Lm[j][k] * in[k][i] (+ dt f) (Lm [H][H], in [H][W])
// summed over k in ascending order like diffuse2 above (same rounding as the one-workgroup kernel).
template <int SIDE>
__global__ void __launch_bounds__(256) k_burgers_circ_large(int H, int W, const float* __restrict__ in, const float* __restrict__ M,
                                                            float* __restrict__ out, const float* __restrict__ f, float dt) {
    __shared__ float ta[16][17], tb[16][17];
    const int b = blockIdx.z, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + tx, j = blockIdx.y * 16 + ty;
    const float* src = in + (size_t)b * H * W;
    const int K = SIDE == 0 ? W : H;
    float s = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        if (SIDE == 0) {
            ta[ty][tx] = (j < H && k0 + tx < K) ? src[(size_t)j * W + k0 + tx] : 0.f;              // in[j][k]
            const int ir = blockIdx.x * 16 + ty;
            tb[ty][tx] = (ir < W && k0 + tx < K) ? M[(size_t)ir * W + k0 + tx] : 0.f;              // R[i][k], row i = tile column ty
        } else {
            ta[ty][tx] = (j < H && k0 + tx < K) ? M[(size_t)j * H + k0 + tx] : 0.f;                // Lm[j][k]
            tb[ty][tx] = (k0 + ty < K && i < W) ? src[(size_t)(k0 + ty) * W + i] : 0.f;            // in[k][i]
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) s += SIDE == 0 ? ta[ty][k] * tb[tx][k] : ta[ty][k] * tb[k][tx];
        __syncthreads();
    }
    if (i < W && j < H) {
        const size_t o = (size_t)b * H * W + (size_t)j * W + i;
        out[o] = (SIDE == 1 && f) ? s + dt * f[o] : s;
    }
}

int bcheck(const sol_burgers_cfg* c) {
    SOL_REQUIRE(c != nullptr, "cfg is NULL");
    SOL_REQUIRE(c->B >= 1 && c->Y >= 2 && c->X >= 2 && c->Y <= 64 && c->X <= 64, "burgers: need 2 <= Y,X <= 64 (got %d,%d)", c->Y, c->X);
    SOL_REQUIRE(c->dx > 0.f, "dx must be > 0");
    SOL_REQUIRE(blds_bytes(c->Y, c->X) <= 160 * 1024, "burgers: grid does not fit the LDS");
    return SOL_OK;
}

template <typename K>
int blaunch(K kernel, const sol_burgers_cfg* c, void* stream, const BArgs& a) {
    const size_t lds = blds_bytes(c->Y, c->X);
    static std::atomic<unsigned long long> optin{0};
    if (int e = sol_lds_optin(optin, {SOL_K(k_burgers_fwd), SOL_K(k_burgers_bwd)}, "burgers kernels")) return e;
    SOL_LAUNCH_NAMED("k_burgers", kernel, dim3(c->B), dim3(NT), lds, (hipStream_t)stream, a);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

}  // namespace

extern "C" int sol_burgers_step_fwd(const sol_burgers_cfg* cfg, void* stream, const float* vy_in, const float* vx_in,
                                    const float* f_y, const float* f_x, const float* circ_yp1, const float* circ_x,
                                    const float* circ_y, const float* circ_xp1, float* vy_out, float* vx_out) {
    if (int e = bcheck(cfg)) return e;
    SOL_REQUIRE(vy_in && vx_in && circ_yp1 && circ_x && circ_y && circ_xp1 && vy_out && vx_out, "sol_burgers_step_fwd: NULL pointer");
    SOL_REQUIRE((f_y == nullptr) == (f_x == nullptr), "f_y and f_x must both be given or both be NULL");
    BArgs a{};
    a.B = cfg->B; a.Y = cfg->Y; a.X = cfg->X; a.dtdx = cfg->dt / cfg->dx; a.dt = cfg->dt;
    a.vy_in = vy_in; a.vx_in = vx_in; a.fy = f_y; a.fx = f_x;
    a.cyp1 = circ_yp1; a.cx = circ_x; a.cy = circ_y; a.cxp1 = circ_xp1;
    a.vy_out = vy_out; a.vx_out = vx_out;
    return blaunch(k_burgers_fwd, cfg, stream, a);
}

extern "C" int sol_burgers_step_bwd(const sol_burgers_cfg* cfg, void* stream, const float* vy_in, const float* vx_in,
                                    const float* circ_yp1, const float* circ_x, const float* circ_y, const float* circ_xp1,
                                    const float* g_vy_out, const float* g_vx_out, float* g_vy_in, float* g_vx_in) {
    if (int e = bcheck(cfg)) return e;
    SOL_REQUIRE(vy_in && vx_in && circ_yp1 && circ_x && circ_y && circ_xp1 && g_vy_out && g_vx_out && g_vy_in && g_vx_in,
                "sol_burgers_step_bwd: NULL pointer");
    BArgs a{};
    a.B = cfg->B; a.Y = cfg->Y; a.X = cfg->X; a.dtdx = cfg->dt / cfg->dx; a.dt = cfg->dt;
    a.vy_in = vy_in; a.vx_in = vx_in;
    a.cyp1 = circ_yp1; a.cx = circ_x; a.cy = circ_y; a.cxp1 = circ_xp1;
    a.g_vy_out = g_vy_out; a.g_vx_out = g_vx_out; a.g_vy_in = g_vy_in; a.g_vx_in = g_vx_in;
    return blaunch(k_burgers_bwd, cfg, stream, a);
}

// ---- large-grid forward step (data generation at the reference's 128 x 128 hi-res setting, burgers/Makefile:19-29) ----
extern "C" size_t sol_burgers_step_large_workspace_bytes(const sol_burgers_cfg* cfg) {
    if (!cfg || cfg->B < 1 || cfg->Y < 2 || cfg->X < 2) return 0;
    const size_t nVy = (size_t)(cfg->Y + 1) * cfg->X, nVx = (size_t)cfg->Y * (cfg->X + 1);
    return (size_t)cfg->B * 2 * (nVy + nVx) * sizeof(float);      // advected field + one product, both components
}

extern "C" int sol_burgers_step_fwd_large(const sol_burgers_cfg* cfg, void* stream, const float* vy_in, const float* vx_in,
                                          const float* f_y, const float* f_x, const float* circ_yp1, const float* circ_x,
                                          const float* circ_y, const float* circ_xp1, float* vy_out, float* vx_out,
                                          void* workspace, size_t workspace_bytes) {
    SOL_REQUIRE(cfg != nullptr, "cfg is NULL");
    SOL_REQUIRE(cfg->B >= 1 && cfg->Y >= 2 && cfg->X >= 2 && cfg->Y <= 1024 && cfg->X <= 1024, "burgers (large): need 2 <= Y,X <= 1024 (got %d,%d)", cfg->Y, cfg->X);
    SOL_REQUIRE(cfg->dx > 0.f, "dx must be > 0");
    SOL_REQUIRE(vy_in && vx_in && circ_yp1 && circ_x && circ_y && circ_xp1 && vy_out && vx_out && workspace, "sol_burgers_step_fwd_large: NULL pointer");
    SOL_REQUIRE((f_y == nullptr) == (f_x == nullptr), "f_y and f_x must both be given or both be NULL");
    SOL_REQUIRE(workspace_bytes >= sol_burgers_step_large_workspace_bytes(cfg), "sol_burgers_step_fwd_large: workspace too small");
    const int B = cfg->B, Y = cfg->Y, X = cfg->X;
    const size_t nVy = (size_t)(Y + 1) * X, nVx = (size_t)Y * (X + 1);
    float* ay = reinterpret_cast<float*>(workspace);
    float* ax = ay + B * nVy;
    float* ty = ax + B * nVx;
    float* tx = ty + B * nVy;
    hipStream_t hs = (hipStream_t)stream;
    const int total = (int)(B * (nVy + nVx));
    SOL_LAUNCH(k_burgers_adv_large, dim3((total + 255) / 256), dim3(256), 0, hs, B, Y, X, cfg->dt / cfg->dx, vy_in, vx_in, ay, ax);
    SOL_LAUNCH_CHECK();
    auto tiles = [](int H, int W, int B_) { return dim3((W + 15) / 16, (H + 15) / 16, B_); };
    // v_y: [Y+1][X]:  T = A * Cx^T, out = Cyp1 * T (+ dt f_y)
    SOL_LAUNCH(k_burgers_circ_large<0>, tiles(Y + 1, X, B), dim3(256), 0, hs, Y + 1, X, (const float*)ay, circ_x, ty, (const float*)nullptr, 0.f);
    SOL_LAUNCH_CHECK();
    SOL_LAUNCH(k_burgers_circ_large<1>, tiles(Y + 1, X, B), dim3(256), 0, hs, Y + 1, X, (const float*)ty, circ_yp1, vy_out, f_y, cfg->dt);
    SOL_LAUNCH_CHECK();
    // v_x: [Y][X+1]:  T = A * Cxp1^T, out = Cy * T (+ dt f_x)
    SOL_LAUNCH(k_burgers_circ_large<0>, tiles(Y, X + 1, B), dim3(256), 0, hs, Y, X + 1, (const float*)ax, circ_xp1, tx, (const float*)nullptr, 0.f);
    SOL_LAUNCH_CHECK();
    SOL_LAUNCH(k_burgers_circ_large<1>, tiles(Y, X + 1, B), dim3(256), 0, hs, Y, X + 1, (const float*)tx, circ_y, vx_out, f_x, cfg->dt);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}
