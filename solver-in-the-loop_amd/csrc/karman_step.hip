// Fused solver step for the karman-2d scene and its adjoint  --  gfx950 / CDNA4.
//
// Replaces KarmanFlow.step (/root/reference/karman-2d/karman_train.py:173-185) and the
// PhiFlow ops it calls (diffuse, advect.semi_lagrangian x2, effect_applied(Inflow),
// divergence_free -> SparseCG, StaggeredGrid.gradient) and their TF gradients.
//
// Design (MI355X first): one simulation = one workgroup.  All solver state of a sample
// (two staggered components, <= 33 KB each at 128x64) lives in that CU's 160 KB LDS from
// the input load to the output store.  The CG vectors x, r, p, Mp never touch LDS at all:
// each thread owns a column strip of 8 cells, so the y-neighbours are registers, the
// x-neighbours are one lane away (wave shift) and only the strip end rows are exchanged
// through LDS (2 floats per thread per iteration).  The dot products are wave64 shuffles
// plus one LDS slot per wave.
#include "split_kernels.hpp"
#include <stdlib.h>

namespace {

// CPT = cells per thread (strip height) is a template parameter (8 or 16): fewer, fatter
// threads amortise the per-wave reduction/halo overhead of a CG iteration (the loop is
// instruction-issue bound), more threads help the tiny grids.
typedef float f2 __attribute__((ext_vector_type(2)));

struct StepArgs {
    int B, Y, X;
    float dtdx;    // dt / dx  (index-space displacement per unit velocity)
    float dt;
    float adt;     // dt*res*res  -> alpha_b = adt / Re_b   (karman_train.py:175)
    float rtol2, atol2;
    int max_iter, grad_pad, inflow_before, dbg;
    const float *d_in, *vy_in, *vx_in, *re, *active, *inflow, *bcv, *bcm, *cinv, *fd;
    int fd_n;
    long bc_stride;
    float *d_out, *vy_out, *vx_out, *saved_vy, *saved_vx, *feat;
    float fs0, fs1, fs2;
    int feat_tr;   // features / feature gradients in the transposed cell order [B][X][Y][C] (the CNN's image order when the CNN runs on the transposed grid)
    int* iters;
    // backward only
    const float *g_vy_out, *g_vx_out, *dfeat;
    float *g_vy_in, *g_vx_in;
    long long* prof;   // SOL_STEP_PROF=1: phase time stamps of workgroup 0 (100 MHz wall clock), debugging only
};
#define SOL_STAMP(i) do { if (a.prof && blockIdx.x == 0 && threadIdx.x == 0) a.prof[i] = wall_clock64(); } while (0)

__host__ __device__ inline int al4(int n) { return (n + 3) & ~3; }

struct Lds {
    float *Avy, *Avx, *Bvy, *Bvx, *E, *red, *cs;
    unsigned char* act;
    float* fdx;          // fd_small_floats(Y, X) floats behind the mask (16-byte aligned)
};

__host__ __device__ inline size_t lds_floats(int Y, int X, int cpt) {
    const int nVy = (Y + 1) * X, nVx = Y * (X + 1);
    return 2 * (size_t)(al4(nVy) + al4(nVx)) + 4 * (size_t)(Y / cpt + 2) * X + 64 + 4 * (size_t)al4(Y * X / 64);
}
// extra LDS of the direct pressure solver on SMALL grids (fd_solve_small: Y*X <= 2048): the transform matrices and two
// field buffers live in LDS there (at 128x64 the register-tiled fd_solve streams them instead)
__host__ __device__ inline size_t fd_small_floats(int Y, int X) {
    return (Y * X <= 2048 && Y >= 16 && X >= 16) ? (size_t)Y * Y + (size_t)X * X + (size_t)X * Y + (size_t)X * 16 + 2 * (size_t)Y * X + (size_t)Y * 16 + 256 + 256 + 256 + 256 + 4096 + (size_t)Y * 16 : 0;
}
__host__ inline size_t lds_bytes(int Y, int X, int cpt) { return lds_floats(Y, X, cpt) * 4 + (size_t)al4(Y * X) + fd_small_floats(Y, X) * 4 + 16; }

__device__ inline Lds carve(float* smem, int Y, int X, int cpt) {
    const int nVy = (Y + 1) * X, nVx = Y * (X + 1);
    Lds l;
    l.Avy = smem;
    l.Avx = l.Avy + al4(nVy);
    l.Bvy = l.Avx + al4(nVx);
    l.Bvx = l.Bvy + al4(nVy);
    l.E = l.Bvx + al4(nVx);
    l.red = l.E + 4 * (Y / cpt + 2) * X;
    l.cs = l.red + 64;                                    // coarse-space scratch: rc[2], rcM, zc
    l.act = reinterpret_cast<unsigned char*>(l.cs + 4 * al4(Y * X / 64));
    l.fdx = reinterpret_cast<float*>(l.act + ((al4(Y * X) + 15) & ~15));
    return l;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// accessible(j,i): active mask with 'boundary' extrapolation (OPEN domain) [EXT-RECALL A.6]
__device__ __forceinline__ float acc_at(const unsigned char* act, int Y, int X, int j, int i) {
    return (float)act[clampi(j, 0, Y - 1) * X + clampi(i, 0, X - 1)];
}
__device__ __forceinline__ float mask_y(const unsigned char* act, int Y, int X, int j, int i) {
    return fminf(acc_at(act, Y, X, j - 1, i), acc_at(act, Y, X, j, i));
}
__device__ __forceinline__ float mask_x(const unsigned char* act, int Y, int X, int j, int i) {
    return fminf(acc_at(act, Y, X, j, i - 1), acc_at(act, Y, X, j, i));
}

// Bilinear sample with index clamping ('boundary' extrapolation).  The sample coordinate is
// (jb + oy, ib + ox) with small float offsets, so the interpolation weights keep full fp32
// precision even at large indices.
struct Bil {
    int j0, j1, i0, i1;
    float wy, wx;
};
__device__ __forceinline__ Bil bil_clamp(int H, int W, int jb, float oy, int ib, float ox) {
    Bil s;
    const float fy = floorf(oy), fx = floorf(ox);
    s.wy = oy - fy;
    s.wx = ox - fx;
    const int j0 = jb + (int)fy, i0 = ib + (int)fx;
    s.j0 = clampi(j0, 0, H - 1);
    s.j1 = clampi(j0 + 1, 0, H - 1);
    s.i0 = clampi(i0, 0, W - 1);
    s.i1 = clampi(i0 + 1, 0, W - 1);
    return s;
}
// (One expression for every caller.  OBSERVATION, round 6 (tools/debug_6432.py): with the three fused multiply-adds spelled out through
//  __builtin_fmaf -- algebraically the same blend -- k_karman_fwd<8, 2> (64 x 32, small-grid direct solve) missed its golden step by 2.3e-4 / 2.7e-3,
//  deterministically, with an error pattern centred on the obstacle window, while k_karman_fwd<8, 0> (CG) of the same build stayed at 5e-8.  The
//  same source built WITHOUT `-mllvm --amdgpu-sched-strategy=iterative-ilp` (_build.py) is correct (4.5e-8): LLVM's experimental scheduler
//  strategy mis-schedules that kernel for that spelling (`max-ilp` is correct too, and as slow as the default: 12.11 vs 11.60 ms per step).  The strategy is worth 0.49 ms of a 11.7 ms C3 step (the weight-gradient body in the
//  fused adjoint launch), so it stays; what guards the shipped binary is that every kernel of this file is compared with the oracle / golden
//  vectors by the GPU suite ON THE BUILD THAT SHIPS.  Do not respell numerics here without running `pytest -m gpu`.)
__device__ __forceinline__ float bil_mix(const Bil& s, float f00, float f01, float f10, float f11) {
#ifdef BIL_EXPLICIT       // the respelling of the observation above (variant builds only)
    const float ux = 1.f - s.wx, uy = 1.f - s.wy;
    const float a0 = __builtin_fmaf(s.wx, f01, ux * f00);
    const float a1 = __builtin_fmaf(s.wx, f11, ux * f10);
    return __builtin_fmaf(s.wy, a1, uy * a0);
#else
    return (1.f - s.wy) * ((1.f - s.wx) * f00 + s.wx * f01) + s.wy * ((1.f - s.wx) * f10 + s.wx * f11);
#endif
}
__device__ __forceinline__ float bil_eval(const float* f, int W, const Bil& s) {
    const float f00 = f[s.j0 * W + s.i0], f01 = f[s.j0 * W + s.i1];
    const float f10 = f[s.j1 * W + s.i0], f11 = f[s.j1 * W + s.i1];
    return bil_mix(s, f00, f01, f10, f11);
}

// explicit diffusion of one face (phase 2 of the forward kernels), fused multiply-adds spelled out (see bil_mix): v_y with the velocity BC blend
__device__ __forceinline__ float dif_y(float c, float up, float dn, float rt, float lf, float alpha, float bcm, float bcv) {
    const float lap = up + dn + rt + lf - 4.f * c;          // summation order ((up + down) + right) + left - 4 c  (4 c is exact)
    return __builtin_fmaf(__builtin_fmaf(alpha, lap, c), 1.f - bcm, bcv);
}
__device__ __forceinline__ float dif_x(float c, float lap, float alpha) { return __builtin_fmaf(alpha, lap, c); }

// ------------------------------------------------------------------------------------
// Strip ownership for the CG: thread -> (strip, column i), rows j0..j0+7
// ------------------------------------------------------------------------------------
struct Own {
    int strip, i, j0, nstrips;
    bool owner;
};
template <int CPT>
__device__ __forceinline__ Own ownership(int Y, int X) {
    Own o;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int spw = 64 / X;   // strips per wave (X <= 64, power of two)
    o.nstrips = Y / CPT;
    o.strip = wave * spw + lane / X;
    o.i = lane % X;
    o.j0 = o.strip * CPT;
    o.owner = o.strip < o.nstrips;
    return o;
}

// Solve M x = rhs with M = -A (SPD): M[c,c] = dg[c] (number of accessible neighbours,
// >= 1), M[c,n] = -active[c]*active[n]; p = 0 outside the OPEN domain.  Matrix-free CG
// from x0 = 0; per-sample stop |r|^2 <= max(rtol2*|b|^2, atol2).  rhs comes in r[], the
// solution leaves in x[].  Returns the iteration count (workgroup uniform).
//
// Classic (Hestenes-Stiefel) CG arithmetic with TWO workgroup barriers per iteration: the
// strip-end rows of the next search direction are exchanged as (r, p_old) pairs BEFORE the
// |r|^2 reduction barrier, and every thread forms p_halo = r_halo + beta*p_old_halo itself
// once beta is known.  The loop is VALU-issue bound (rocprof: SQ_ACTIVE_INST_ANY ~ 100% of
// one SIMD's issue slots), so it is written for instruction count: x-neighbours are folded
// DPP operands (v_add_f32_dpp wave_shr/shl), the vector updates are packed (v_pk_fma_f32),
// the dot products reduce with DPP row ops + row_bcast + ONE v_readlane, divisions are
// v_rcp_f32, the obstacle mask is applied only by waves that own obstacle cells.
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, true));
}
__device__ __forceinline__ float row_allsum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror  -> every lane holds its row-of-16 sum
    return v;
}
// all-reduce over the workgroup; `slot` (16 floats, entries >= #waves stay zero) must not be
// rewritten before every thread has passed the NEXT barrier.  Bit-identical in every lane of
// every wave (same association everywhere), so branches on the result are uniform.
__device__ __forceinline__ float cg_block_sum(float v, float* slot) {
    v = row_allsum(v);
    v += dpp_mov<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3 += rows 0,2
    v += dpp_mov<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3 += row 1   => lane 63 = wave total
    const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    slot[threadIdx.x >> 6] = s;    // every lane stores the same value to the same address
    __syncthreads();
    return row_allsum(slot[threadIdx.x & 15]);
}

template <int CPT, bool FULLROW>   // FULLROW: X == 64, one wave per strip row -> DPP bound_ctrl supplies the zero halo
__device__ __forceinline__ int cg_solve(const Own& o, int X, const float (&dgf)[CPT], const float (&acf)[CPT],
                                        float (&rf)[CPT], float (&xf)[CPT], float* E, float* red,
                                        float rtol2, float atol2, int max_iter) {
    constexpr int H = CPT / 2;
    f2 dg[H], ac[H], r[H], x[H], p[H], Mp[H];
    bool hasobst = false;
    f2 part2 = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < H; ++q) {
        dg[q] = (f2){dgf[2 * q], dgf[2 * q + 1]};
        ac[q] = (f2){acf[2 * q], acf[2 * q + 1]};
        r[q] = (f2){rf[2 * q], rf[2 * q + 1]};
        x[q] = (f2){0.f, 0.f};
        p[q] = r[q];
        part2 += r[q] * r[q];
        hasobst |= (acf[2 * q] == 0.f) | (acf[2 * q + 1] == 0.f);
    }
    const bool wobst = __ballot(hasobst && o.owner) != 0ull;   // wave uniform
    const int EW = (o.nstrips + 2) * X;
    // E: [4][nstrips+2][X] = r_top, r_bot, p_top, p_bot of every strip; rows 0 and nstrips+1 stay zero
    float* Ert = E; float* Erb = E + EW; float* Ept = E + 2 * EW; float* Epb = E + 3 * EW;
    for (int k = threadIdx.x; k < 4 * EW; k += blockDim.x) E[k] = 0.f;
    if (threadIdx.x < 64) red[threadIdx.x] = 0.f;
    __syncthreads();
    const int eo = (o.strip + 1) * X + o.i;
    if (FULLROW || o.owner) { Ert[eo] = r[0].x; Erb[eo] = r[H - 1].y; }
    float rr = cg_block_sum(part2.x + part2.y, red + 16);      // barrier: also publishes the halo rows
    float hprev = 0.f, hnext = 0.f;
    if (FULLROW || o.owner) { hprev = Erb[eo - X]; hnext = Ert[eo + X]; }   // p == r at start
    const float thresh = fmaxf(rtol2 * rr, atol2);
    const bool has_l = o.i > 0, has_r = o.i < X - 1;
    int it = 0;
    while (rr > thresh && it < max_iter) {
        float nb[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const float prev = k > 0 ? p[(k - 1) / 2][(k - 1) & 1] : hprev;
            const float next = k < CPT - 1 ? p[(k + 1) / 2][(k + 1) & 1] : hnext;
            nb[k] = prev + next;
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            float pl = dpp_mov<0x138>(p[k / 2][k & 1]);   // wave_shr:1  lane i <- lane i-1 (0 into lane 0)
            if (!FULLROW) pl = has_l ? pl : 0.f;
            nb[k] += pl;
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            float pr = dpp_mov<0x130>(p[k / 2][k & 1]);   // wave_shl:1  lane i <- lane i+1 (0 into lane 63)
            if (!FULLROW) pr = has_r ? pr : 0.f;
            nb[k] += pr;
        }
        part2 = (f2){0.f, 0.f};
#pragma unroll
        for (int q = 0; q < H; ++q) Mp[q] = dg[q] * p[q] - (f2){nb[2 * q], nb[2 * q + 1]};
        if (wobst) {
#pragma unroll
            for (int q = 0; q < H; ++q) Mp[q] *= ac[q];
        }
#pragma unroll
        for (int q = 0; q < H; ++q) part2 += p[q] * Mp[q];
        const float pMp = cg_block_sum(part2.x + part2.y, red);            // barrier A
        if (!(pMp > 0.f)) break;
        const float alpha = rr * __builtin_amdgcn_rcpf(pMp);
        part2 = (f2){0.f, 0.f};
#pragma unroll
        for (int q = 0; q < H; ++q) {
            x[q] += alpha * p[q];
            r[q] -= alpha * Mp[q];
            part2 += r[q] * r[q];
        }
        if (FULLROW || o.owner) { Ert[eo] = r[0].x; Erb[eo] = r[H - 1].y; Ept[eo] = p[0].x; Epb[eo] = p[H - 1].y; }
        const float rrn = cg_block_sum(part2.x + part2.y, red + 16);       // barrier B (publishes the halo rows too)
        const float beta = rrn * __builtin_amdgcn_rcpf(rr);
        rr = rrn;
#pragma unroll
        for (int q = 0; q < H; ++q) p[q] = r[q] + beta * p[q];
        if (FULLROW || o.owner) {
            hprev = Erb[eo - X] + beta * Epb[eo - X];
            hnext = Ert[eo + X] + beta * Ept[eo + X];
        }
        ++it;
    }
#pragma unroll
    for (int q = 0; q < H; ++q) { xf[2 * q] = x[q].x; xf[2 * q + 1] = x[q].y; }
    return it;
}

// two-value variant of cg_block_sum (one barrier)
__device__ __forceinline__ void cg_block_sum2(float& v0, float& v1, float* slot) {
    v0 = row_allsum(v0); v1 = row_allsum(v1);
    v0 += dpp_mov<0x142, 0xa>(v0); v1 += dpp_mov<0x142, 0xa>(v1);
    v0 += dpp_mov<0x143, 0xc>(v0); v1 += dpp_mov<0x143, 0xc>(v1);
    const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v0), 63));
    const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v1), 63));
    slot[threadIdx.x >> 6] = s0;
    slot[16 + (threadIdx.x >> 6)] = s1;
    __syncthreads();
    v0 = row_allsum(slot[threadIdx.x & 15]);
    v1 = row_allsum(slot[16 + (threadIdx.x & 15)]);
}
// workgroup max of a non-negative value (one barrier; `slot` = 16 floats, stale entries must be >= 0 and <= result)
template <int NT = 0>
__device__ __forceinline__ float block_max(float v, float* slot) {
    v = __uint_as_float(amax_wave_max(__float_as_uint(v)));      // DPP row operations (non-negative floats order like their bit patterns)
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = 0.f;
    const int nw = ((NT ? NT : (int)blockDim.x) + 63) >> 6;
    for (int w = 0; w < nw; ++w) m = fmaxf(m, slot[w]);
    return m;
}
// two workgroup maxima with ONE barrier (`slot`: 32 floats)
template <int NT = 0>
__device__ __forceinline__ void block_max2(float& a, float& b, float* slot) {
    a = __uint_as_float(amax_wave_max(__float_as_uint(a)));
    b = __uint_as_float(amax_wave_max(__float_as_uint(b)));
    if ((threadIdx.x & 63) == 0) { slot[threadIdx.x >> 6] = a; slot[16 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    float ma = 0.f, mb = 0.f;
    const int nw = ((NT ? NT : (int)blockDim.x) + 63) >> 6;
    for (int w = 0; w < nw; ++w) { ma = fmaxf(ma, slot[w]); mb = fmaxf(mb, slot[16 + w]); }
    a = ma; b = mb;
}
__device__ __forceinline__ float sum8lanes(float v) {   // all-reduce over aligned groups of 8 lanes
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    return v;
}

// Two-level preconditioned CG.  M^-1 = D^-1 + P (P^T A P)^-1 P^T with piecewise-constant
// aggregates of 8x8 cells (P) and the dense coarse inverse `cinv_g` [nc,nc] prepared by the host
// for the scene geometry (nc = Y*X/64 = 128 at 128x64).  The coarse space removes the smooth
// error modes that make plain CG need ~240 iterations on this grid: ~56 iterations instead, same
// converged solution.  Still TWO barriers per iteration: the restriction is linear, so
// P^T r_new = P^T r - alpha P^T(Mp) is formed from block sums published before the alpha barrier;
// r.z = sum r^2/d + (P^T r).(zc) needs no per-cell z before the beta barrier; each thread keeps its
// 32-entry slice of a coarse-inverse row in registers (4 threads per coarse row).
template <int CPT, bool FULLROW, int CC>   // CC = coarse columns per thread = nc / (64/CPT): 32 at 128x64, 8 at 64x32
__device__ __forceinline__ int pcg_solve(const Own& o, int Y, int X, const unsigned char* act,
                                         const float (&dgf)[CPT], const float (&acf)[CPT], float (&rf)[CPT], float (&xf)[CPT],
                                         float* E, float* red, float* cs, const float* __restrict__ cinv_g,
                                         float rtol2, float atol2, int max_iter) {
    constexpr int H = CPT / 2, NBS = CPT / 8, TPR = 64 / CPT, CPTc = CC;
    const int nbx = X >> 3, nc = (Y >> 3) * nbx, ncp = al4(nc);
    const int tid = threadIdx.x;
    const int I = tid / TPR, q = tid % TPR;
    const bool crow = I < nc;
    f2 dg[H], ac[H], idg[H], r[H], x[H], p[H], Mp[H];
    bool hasobst = false;
#pragma unroll
    for (int qq = 0; qq < H; ++qq) {
        dg[qq] = (f2){dgf[2 * qq], dgf[2 * qq + 1]};
        ac[qq] = (f2){acf[2 * qq], acf[2 * qq + 1]};
        idg[qq] = (f2){1.f / dgf[2 * qq], 1.f / dgf[2 * qq + 1]};
        r[qq] = (f2){rf[2 * qq], rf[2 * qq + 1]};
        x[qq] = (f2){0.f, 0.f};
        hasobst |= (acf[2 * qq] == 0.f) | (acf[2 * qq + 1] == 0.f);
    }
    const bool wobst = __ballot(hasobst && o.owner) != 0ull;
    // this thread's slice of coarse-inverse row I (registers, constant over the solve)
    f2 cinv[CC / 2];
#pragma unroll
    for (int c = 0; c < CC / 4; ++c) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (crow) v = reinterpret_cast<const float4*>(cinv_g + (size_t)I * nc + q * CC)[c];
        cinv[2 * c] = (f2){v.x, v.y};
        cinv[2 * c + 1] = (f2){v.z, v.w};
    }
    float* rcb = cs;               // [2][ncp]
    float* rcM = cs + 2 * ncp;     // [ncp]
    float* zc = cs + 3 * ncp;      // [ncp]
    const int EW = (o.nstrips + 2) * X;
    float* Ert = E; float* Erb = E + EW; float* Ept = E + 2 * EW; float* Epb = E + 3 * EW;
    for (int k = tid; k < 4 * EW; k += blockDim.x) E[k] = 0.f;
    for (int k = tid; k < 4 * ncp; k += blockDim.x) cs[k] = 0.f;
    if (tid < 64) red[tid] = 0.f;
    __syncthreads();
    const bool own = FULLROW || o.owner;
    const int eo = (o.strip + 1) * X + o.i;
    const bool has_prev = own && o.strip > 0, has_next = own && o.strip < o.nstrips - 1;
    const bool has_l = o.i > 0, has_r = o.i < X - 1;
    const int bcol = o.i >> 3, brow0 = o.strip * NBS;
    const int blk_prev = (brow0 - 1) * nbx + bcol, blk_next = (brow0 + NBS) * nbx + bcol;
    auto dg_at = [&](int j, int i) {
        const float n = acc_at(act, Y, X, j - 1, i) + acc_at(act, Y, X, j + 1, i) + acc_at(act, Y, X, j, i - 1) + acc_at(act, Y, X, j, i + 1);
        return fmaxf(n, 1.f);
    };
    const float idg_prev = has_prev ? 1.f / dg_at(o.j0 - 1, o.i) : 0.f;
    const float idg_next = has_next ? 1.f / dg_at(o.j0 + CPT, o.i) : 0.f;
    const float ac_prev = has_prev ? (float)act[(o.j0 - 1) * X + o.i] : 0.f;     // prolongation is masked (P = active * indicator)
    const float ac_next = has_next ? (float)act[(o.j0 + CPT) * X + o.i] : 0.f;
    const bool blk_writer = own && (o.i & 7) == 0;

    auto block_sums = [&](const f2 (&v)[H], float* dst) {    // dst[blk] = sum over the 8x8 aggregate
#pragma unroll
        for (int h = 0; h < NBS; ++h) {
            f2 s2 = v[4 * h] + v[4 * h + 1] + v[4 * h + 2] + v[4 * h + 3];
            const float s = sum8lanes(s2.x + s2.y);
            if (blk_writer) dst[(brow0 + h) * nbx + bcol] = s;
        }
    };
    // coarse solve: zc[I] = sum_J Cinv[I][J] * (rc_old[J] - alpha * rcM[J]); rc_new[I]; returns rc_new[I]*zc[I] on q == 0
    auto coarse_solve = [&](const float* rco, float* rcn, float alpha) {
        f2 acc2 = {0.f, 0.f};
        const f2 nal = {-alpha, -alpha};
#pragma unroll
        for (int c = 0; c < CC / 4; ++c) {
            const float4 a = reinterpret_cast<const float4*>(rco + q * CC)[c];
            const float4 m = reinterpret_cast<const float4*>(rcM + q * CC)[c];
            acc2 += cinv[2 * c] * ((f2){a.x, a.y} + nal * (f2){m.x, m.y});
            acc2 += cinv[2 * c + 1] * ((f2){a.z, a.w} + nal * (f2){m.z, m.w});
        }
        float acc = acc2.x + acc2.y;
        acc += dpp_mov<0xB1>(acc);
        acc += dpp_mov<0x4E>(acc);
        if (TPR == 8) acc += dpp_mov<0x141>(acc);
        float cd = 0.f;
        if (crow && q == 0) {
            const float rn = rco[I] - alpha * rcM[I];
            rcn[I] = rn;
            zc[I] = acc;
            cd = rn * acc;
        }
        return cd;
    };

    // ---- start: z0 = M^-1 r0, p0 = z0 -------------------------------------------------------
    block_sums(r, rcb);                        // rc[0] = P^T r0   (rcM == 0, alpha irrelevant)
    __syncthreads();
    f2 part2 = {0.f, 0.f}, part3 = {0.f, 0.f};
    float cd = coarse_solve(rcb, rcb + ncp, 0.f);   // also copies rc into buffer 1 -> use buffer 1 as "old" of iteration 0
#pragma unroll
    for (int qq = 0; qq < H; ++qq) { part3 += r[qq] * r[qq]; part2 += r[qq] * r[qq] * idg[qq]; }
    if (own) { Ert[eo] = r[0].x; Erb[eo] = r[H - 1].y; }
    float rz = part2.x + part2.y + cd, rr = part3.x + part3.y;
    cg_block_sum2(rz, rr, red + 16);
    {
#pragma unroll
        for (int h = 0; h < NBS; ++h) {
            const float zch = zc[(brow0 + h) * nbx + bcol];
#pragma unroll
            for (int qq = 4 * h; qq < 4 * h + 4; ++qq) p[qq] = r[qq] * idg[qq] + (wobst ? ac[qq] * zch : (f2){zch, zch});
        }
    }
    float hprev = has_prev ? Erb[eo - X] * idg_prev + ac_prev * zc[blk_prev] : 0.f;
    float hnext = has_next ? Ert[eo + X] * idg_next + ac_next * zc[blk_next] : 0.f;
    const float thresh = fmaxf(rtol2 * rr, atol2);
    int it = 0;
    while (rr > thresh && it < max_iter) {
        float nb[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const float prev = k > 0 ? p[(k - 1) / 2][(k - 1) & 1] : hprev;
            const float next = k < CPT - 1 ? p[(k + 1) / 2][(k + 1) & 1] : hnext;
            nb[k] = prev + next;
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            float pl = dpp_mov<0x138>(p[k / 2][k & 1]);
            if (!FULLROW) pl = has_l ? pl : 0.f;
            nb[k] += pl;
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            float pr = dpp_mov<0x130>(p[k / 2][k & 1]);
            if (!FULLROW) pr = has_r ? pr : 0.f;
            nb[k] += pr;
        }
#pragma unroll
        for (int qq = 0; qq < H; ++qq) Mp[qq] = dg[qq] * p[qq] - (f2){nb[2 * qq], nb[2 * qq + 1]};
        if (wobst) {
#pragma unroll
            for (int qq = 0; qq < H; ++qq) Mp[qq] *= ac[qq];
        }
        part2 = (f2){0.f, 0.f};
#pragma unroll
        for (int qq = 0; qq < H; ++qq) part2 += p[qq] * Mp[qq];
        block_sums(Mp, rcM);
        const float pMp = cg_block_sum(part2.x + part2.y, red);            // barrier A (publishes P^T Mp)
        if (!(pMp > 0.f)) break;
        const float alpha = rz * __builtin_amdgcn_rcpf(pMp);
        part2 = (f2){0.f, 0.f};
        part3 = (f2){0.f, 0.f};
#pragma unroll
        for (int qq = 0; qq < H; ++qq) {
            x[qq] += alpha * p[qq];
            r[qq] -= alpha * Mp[qq];
            const f2 r2 = r[qq] * r[qq];
            part3 += r2;
            part2 += r2 * idg[qq];
        }
        // iteration `it` reads rc buffer (it+1)&1 and writes buffer it&1
        cd = coarse_solve(rcb + ((it + 1) & 1) * ncp, rcb + (it & 1) * ncp, alpha);
        if (own) { Ert[eo] = r[0].x; Erb[eo] = r[H - 1].y; Ept[eo] = p[0].x; Epb[eo] = p[H - 1].y; }
        float rzn = part2.x + part2.y + cd, rrn = part3.x + part3.y;
        cg_block_sum2(rzn, rrn, red + 16);                                  // barrier B (publishes zc and the halo rows)
        const float beta = rzn * __builtin_amdgcn_rcpf(rz);
        rz = rzn;
        rr = rrn;
#pragma unroll
        for (int h = 0; h < NBS; ++h) {
            const float zch = zc[(brow0 + h) * nbx + bcol];
#pragma unroll
            for (int qq = 4 * h; qq < 4 * h + 4; ++qq) p[qq] = (r[qq] * idg[qq] + (wobst ? ac[qq] * zch : (f2){zch, zch})) + beta * p[qq];
        }
        hprev = has_prev ? (Erb[eo - X] * idg_prev + ac_prev * zc[blk_prev]) + beta * Epb[eo - X] : 0.f;
        hnext = has_next ? (Ert[eo + X] * idg_next + ac_next * zc[blk_next]) + beta * Ept[eo + X] : 0.f;
        ++it;
    }
#pragma unroll
    for (int qq = 0; qq < H; ++qq) { xf[2 * qq] = x[qq].x; xf[2 * qq + 1] = x[qq].y; }
    return it;
}


// ------------------------------------------------------------------------------------
// Direct pressure solve (128 x 64): fast diagonalisation + capacitance correction
// ------------------------------------------------------------------------------------
// M = M_r + U_S E_SS U_S^T with M_r the Dirichlet 5-point Laplacian of the rectangle, diagonalised by the
// orthonormal sine transforms Qy (128x128), Qx (64x64): M_r^-1 = G = (Qy (x) Qx) diag(1/lam) (Qy (x) Qx).
//     x = G (b - U_S E_SS x_S),   x_S = (I + G_SS E_SS)^-1 (G b)_S            (precond.direct_solver_blob)
// One forward 2-D transform of b, the spectral coefficients of the window-sized correction added in
// spectral space, one inverse transform.  ~7 MFLOP per solve instead of ~56 preconditioned CG
// iterations; no reductions, 13 barriers.  The transforms are 16 x 64-column (or 128-row x 16) register
// tiles per wave whose coefficients Q[k][16 values] are WAVE UNIFORM: they are fetched with
// s_load_dwordx16 and used as SGPR-pair operands of v_pk_fma_f32, the data element comes from one
// conflict-free ds_read_b32 per 8 packed FMAs (rows padded to 65 floats serve row and column access).
// The blob is constant for the lifetime of the kernel: reading it through the CONSTANT address space is what lets
// the compiler scalarise the wave-uniform coefficient loads even though the kernel has stored to global memory.
typedef const float __attribute__((address_space(4)))* fd_cfp;
typedef const f2 __attribute__((address_space(4)))* fd_cf2p;
struct FdView {
    const float *Qy, *Qx, *ilT, *KpT, *QxW;
    const int* sidx;
    int wy0, wx0, SP;
};
constexpr int FD_Y = 128, FD_X = 64, FD_LD = 65, FD_WIN = 16, FD_ULD = 17;
constexpr int FD_BUF = FD_Y * FD_LD;   // floats per transform buffer
#ifndef SOL_FD_MFMA
#define SOL_FD_MFMA 1                  // the sine transforms of fd_solve on the fp32 matrix cores (0: the packed-VALU forms, for A/B builds)
#endif
constexpr int FD_QYS = SOL_FD_MFMA ? 32 : FD_Y / 16;   // VGPRs of the wave's Qy coefficient slice, loaded at kernel start

__device__ __forceinline__ FdView fd_view(const float* __restrict__ blob) {
    const int* h = reinterpret_cast<const int*>(blob);
    FdView v;
    v.wy0 = h[3]; v.wx0 = h[4]; v.SP = h[6];
    v.Qy = blob + 16;
    v.Qx = v.Qy + FD_Y * FD_Y;
    v.ilT = v.Qx + FD_X * FD_X;
    v.KpT = v.ilT + FD_X * FD_Y;
    v.sidx = reinterpret_cast<const int*>(v.KpT + (size_t)v.SP * v.SP);
    v.QxW = reinterpret_cast<const float*>(v.sidx + v.SP);
    return v;
}
// Sine-transform symmetry: Q[NK-1-k][j] = (-1)^j Q[k][j] and Q[k][NK-1-j] = (-1)^k Q[k][j], so with
// s_k = v_k + v_{NK-1-k}, d_k = v_k - v_{NK-1-k} (k < NK/2):
//     out[j]      = sum_k Q[k][j] * (j even ? s_k : d_k)
//     out[NK-1-j] = sum_k (-1)^k Q[k][j] * (j even ? d_k : s_k)
// i.e. HALF the multiply-adds and only the [NK/2 x NK/2] corner of Q is read (16 KB of Qy, 4 KB of Qx: the
// wave-uniform s_load stream stays inside the scalar cache).  A thread produces the 8 outputs j = 8*g8 + r and
// their 8 mirrors NK-1-j:  lo[q] = (out[8g8+2q], out[8g8+2q+1]),  hi[q] = (out[NK-1-(8g8+2q)], out[NK-1-(8g8+2q+1)]).
// Input element k of the thread is bp[k * SK].
template <int NK, int SK>
__device__ __forceinline__ void fd_trans(const float* __restrict__ Q, const float* bp, int g8, f2 (&lo)[4], f2 (&hi)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { lo[q] = (f2){0.f, 0.f}; hi[q] = (f2){0.f, 0.f}; }
    fd_cfp Qg = (fd_cfp)(Q + 8 * g8);
#pragma unroll 4
    for (int k = 0; k < NK / 2; k += 2) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const float va = bp[(k + kk) * SK], vb = bp[(NK - 1 - k - kk) * SK];
            const float sk = va + vb, dk = va - vb;
            const f2 sd = {sk, dk};
            const f2 ds = kk == 0 ? (f2){dk, sk} : (f2){-dk, -sk};      // (-1)^k
            fd_cf2p qr = (fd_cf2p)(Qg + (size_t)(k + kk) * NK);
#pragma unroll
            for (int q = 0; q < 4; ++q) { lo[q] += qr[q] * sd; hi[q] += qr[q] * ds; }
        }
    }
}

// Same transform with the wave's coefficient slice Q[k < NK/2][8*g8 .. 8*g8+7] held in VGPRs (element 8k+r in register
// (8k+r) >> 6, lane (8k+r) & 63; loaded once per kernel by fd_load_slice) and moved to SGPR pairs with v_readlane: the
// 16 KB corner of Qy does not stay in the scalar cache, and an s_load miss costs more than 8 extra VALU instructions.
template <int NK>
__device__ __forceinline__ void fd_load_slice(const float* __restrict__ Q, int g8, float (&sl)[NK / 16]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < NK / 16; ++j) sl[j] = Q[(size_t)(8 * j + (lane >> 3)) * NK + 8 * g8 + (lane & 7)];
}
template <int NK, int SK>
__device__ __forceinline__ void fd_trans_rl(const float (&sl)[NK / 16], const float* bp, f2 (&lo)[4], f2 (&hi)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { lo[q] = (f2){0.f, 0.f}; hi[q] = (f2){0.f, 0.f}; }
#pragma unroll
    for (int k = 0; k < NK / 2; ++k) {
        const float va = bp[k * SK], vb = bp[(NK - 1 - k) * SK];
        const float sk = va + vb, dk = va - vb;
        const f2 sd = {sk, dk};
        const f2 ds = (k & 1) == 0 ? (f2){dk, sk} : (f2){-dk, -sk};      // (-1)^k
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = 8 * k + 2 * q;
            const float q0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl[e >> 6]), e & 63));
            const float q1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl[(e + 1) >> 6]), (e + 1) & 63));
            const f2 qq = {q0, q1};
            lo[q] += qq * sd;
            hi[q] += qq * ds;
        }
    }
}

// ---- the four sine transforms of a solve on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) ----------------------------------
// Same peak as the packed VALU, but the operands arrive as plain vector registers: no wave-uniform coefficient stream (v_readlane /
// s_load per packed FMA), which is what the VALU forms above are bound by (8.0 us for the two forward transforms of a solve against
// 2.6 us of multiply-adds).  Eight waves = eight 32 x 32 output tiles:
//   y transform (128 rows, symmetry as above): four products [32 j' x 64 k] . [64 k x 64 c], wave = (product p, column half):
//       p = 0: rows j = 2j'      = sum_k Q[k][j] s_k          p = 2: rows 127 - 2j'       = sum_k (-1)^k Q[k][2j'] d_k
//       p = 1: rows j = 2j' + 1  = sum_k Q[k][j] d_k          p = 3: rows 127 - (2j' + 1) = sum_k (-1)^k Q[k][2j'+1] s_k
//     A operand = the wave's coefficient slice, 32 VGPRs loaded once per kernel (fd_load_ay: register t, lane (i, kk) = coefficient
//     (k = 2t + kk, row j'(i))); B operand = s_k / d_k formed from two ds_read_b32 of the column (rows k and 127 - k).
//   x transform (64 columns, no symmetry: its parity split would halve the tile width): [128 m x 64 c] . [64 c x 64 c'],
//     wave = (32-row tile, column half); A operand = one ds_read_b32 of the row, B operand = Qx[c][32 half + j] in 32 VGPRs (fd_load_bx).
// Accumulator tile D[r], lane: row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31.
typedef float fd_f16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void fd_load_ay(const float* __restrict__ Qy, int w, float (&ay)[32]) {
    const int lane = threadIdx.x & 63, i = lane & 31, kk = lane >> 5, p = w >> 1;
    const float sgn = (p >= 2 && kk) ? -1.f : 1.f;               // (-1)^k, k = 2t + kk
#pragma unroll
    for (int t = 0; t < 32; ++t) ay[t] = sgn * Qy[(size_t)(2 * t + kk) * FD_Y + 2 * i + (p & 1)];
}
__device__ __forceinline__ void fd_load_bx(const float* __restrict__ Qx, int w, float (&bx)[32]) {
    const int lane = threadIdx.x & 63, j = lane & 31, kk = lane >> 5;
#pragma unroll
    for (int t = 0; t < 32; ++t) bx[t] = Qx[(size_t)(2 * t + kk) * FD_X + 32 * (w & 1) + j];
}
// y transform of src ([128][FD_LD]) -> dst ([128][LDD])
template <int LDD>
__device__ __forceinline__ void fd_mfma_y(const float (&ay)[32], const float* src, float* dst, int w) {
    const int lane = threadIdx.x & 63, kk = lane >> 5, p = w >> 1;
    const float* lo = src + 32 * (w & 1) + (lane & 31) + kk * FD_LD;                  // rows k = 2t + kk
    const float* hi = src + 32 * (w & 1) + (lane & 31) + (FD_Y - 1 - kk) * FD_LD;      // rows 127 - k
    const float dsg = (p == 0 || p == 3) ? 1.f : -1.f;                                  // s_k (p = 0, 3) or d_k (p = 1, 2)   (wave uniform)
    fd_f16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const float v = lo[2 * t * FD_LD] + dsg * hi[-2 * t * FD_LD];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[t], v, acc, 0, 0, 0);
    }
    float* d = dst + 32 * (w & 1) + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * kk;
        const int row = p < 2 ? 2 * i + p : FD_Y - 1 - 2 * i - (p - 2);
        d[row * LDD] = acc[r];
    }
}
// x transform of src ([128][FD_LD]) -> dst ([128][FD_LD]); src and dst must be different buffers
__device__ __forceinline__ void fd_mfma_x(const float (&bx)[32], const float* src, float* dst, int w) {
    const int lane = threadIdx.x & 63, kk = lane >> 5;
    const float* ap = src + (32 * (w >> 1) + (lane & 31)) * FD_LD + kk;                 // row m, columns c = 2t + kk
    fd_f16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 32; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * t], bx[t], acc, 0, 0, 0);
    float* d = dst + (32 * (w >> 1)) * FD_LD + 32 * (w & 1) + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * kk) * FD_LD] = acc[r];
}

// Touch every 128-byte line of the blob once at kernel start (one dword per line, summed into a value the caller
// keeps alive): the solver kernels run between convolution launches that stream hundreds of MB through the L2,
// so the 263 KB of coefficients would otherwise come from HBM inside the latency-bound scalar-load loops.
// All requests go out before the first is summed (clamped index, no predicate): as a loop `s += blob[i]` the five requests of a
// thread were issued one round trip after the other -- in the training pipeline, where the convolution launches have flushed
// the blob out of the L2, that made the load phase of the adjoint kernel 15 us instead of 3.4 (tools/step_phases.py ... train).
// The blob has at most 8 x 512 lines of 128 bytes (checked by the host: fd_n <= 131072 words).
__device__ __forceinline__ float fd_prefetch(const float* __restrict__ blob, int words) {
    float v[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) v[n] = blob[min((int)(threadIdx.x + n * blockDim.x) * 32, words - 1)];
    return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

__device__ __forceinline__ void fd_load_qys(const float* __restrict__ Qy, int w, float (&qys)[FD_QYS]) {
#if SOL_FD_MFMA
    fd_load_ay(Qy, w, qys);
#else
    fd_load_slice<FD_Y>(Qy, w, qys);
#endif
}

// rhs in rf[] (strip layout: rows 16*wave + k, column lane); returns the solution as a [128][64] LDS array (inside buf,
// complete for every thread).  buf = 2*FD_BUF floats of LDS.
// The first request group of fd_solve (x-transform operand, window operands, eigenvalue reciprocals: 68 registers), loadable at KERNEL START
// (k_karman_fwd_bands, band 0): inside the solve these requests cost ~3 us in front of the first transform (fwd-transform 8.1 us against 4.4 us
// for the inverse pair, which finds its operands in registers).
struct FdOps { float bx[32], qxw[16], qyk[4], il[16]; };
__device__ __forceinline__ void fd_load_ops(const float* __restrict__ blob, FdOps& o) {
    const FdView F = fd_view(blob);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = tid & 127, cb = __builtin_amdgcn_readfirstlane(tid >> 7);
    fd_load_bx(F.Qx, w, o.bx);
#pragma unroll
    for (int t = 0; t < 16; ++t) o.qxw[t] = F.QxW[(4 * t + (lane >> 4)) * FD_WIN + (lane & 15)];
#pragma unroll
    for (int t = 0; t < 4; ++t) o.qyk[t] = F.Qy[(size_t)(F.wy0 + (lane & 15)) * FD_Y + 16 * w + 4 * t + (lane >> 4)];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 8 * cb + 2 * q + e;
            o.il[2 * q + e] = F.ilT[c * FD_Y + m];
            o.il[8 + 2 * q + e] = F.ilT[(FD_X - 1 - c) * FD_Y + m];
        }
}
// RHS_STAGED: the caller has written the right-hand side into buf as [128][FD_LD] itself (k_karman_fwd_bands); rf is then unused
// pre != nullptr: the first request group was loaded by the caller (fd_load_ops)
template <bool RHS_STAGED = false, bool PRELOADED = false>
__device__ __forceinline__ float* fd_solve(const float* __restrict__ blob, const float (&qys)[FD_QYS], float* buf, const float (&rf)[16], long long* prof,
                                           const FdOps& pre) {
#define FD_STAMP(i) do { if (prof && (blockIdx.x == 0 || (RHS_STAGED && blockIdx.x == 32)) && threadIdx.x == 0) prof[i] = wall_clock64(); } while (0)   /* (32: the solver workgroup of simulation 0 in k_karman_fwd_bands) */
    const FdView F = fd_view(blob);
    float* B0 = buf;
    float* B1 = buf + FD_BUF;
    float* U = B0;                       // [128][17]  u / t2w
    float* XP = B0 + 2304;               // [2][256]   partial window values
    float* XS = B0 + 2816;               // [SP]       gathered x0 on S
    float* CP = B0 + 3072;               // [2][256]   partial K' x_S
    float* W2 = B0 + 3584;               // [16][16]   -E_SS x_S scattered into the window
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = tid & 127, cb = __builtin_amdgcn_readfirstlane(tid >> 7);
#if !SOL_FD_MFMA
    f2 lo[4], hi[4];
#endif
    // this thread's 16 spectral coefficients: index 2q+e -> column 8cb+2q+e, 8+2q+e -> column 63-(8cb+2q+e)
    float t2[16], il[16];
    int col[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        col[2 * q] = 8 * cb + 2 * q; col[2 * q + 1] = col[2 * q] + 1;
        col[8 + 2 * q] = FD_X - 1 - col[2 * q]; col[8 + 2 * q + 1] = FD_X - 2 - col[2 * q];
    }

    // ---- forward transform: T2 = (Qy b Qx) / lam ---------------------------------------
    if constexpr (!RHS_STAGED) {
#pragma unroll
        for (int k = 0; k < 16; ++k) B0[(16 * w + k) * FD_LD + lane] = rf[k];
    }
#if SOL_FD_MFMA
    float bx[32];
    float* XP8 = B0 + 4096;              // [8 waves][256] partial window values (MFMA form)
    // window products on v_mfma_f32_16x16x4_f32 (wave = 16 rows m; B operand lane: column lane & 15, k = lane >> 4).  Two request groups, so
    // that no more than 20 extra registers are live across the transforms: what the first two window phases need here, the rest behind them.
    float qxw[16], qyk[4];               // QxW[c][i'] (u = T2 QxW);  window rows of Qy as A operand [16 jw x 4 m] of x0w = Qy[win, :] u (this wave's K slice)
    if constexpr (PRELOADED) {
#pragma unroll
        for (int t = 0; t < 32; ++t) bx[t] = pre.bx[t];
#pragma unroll
        for (int t = 0; t < 16; ++t) { qxw[t] = pre.qxw[t]; il[t] = pre.il[t]; }
#pragma unroll
        for (int t = 0; t < 4; ++t) qyk[t] = pre.qyk[t];
    } else {
        fd_load_bx(F.Qx, w, bx);         // x-transform operand + the eigenvalue reciprocals: in flight behind the y transform
#pragma unroll
        for (int t = 0; t < 16; ++t) qxw[t] = F.QxW[(4 * t + (lane >> 4)) * FD_WIN + (lane & 15)];
#pragma unroll
        for (int t = 0; t < 4; ++t) qyk[t] = F.Qy[(size_t)(F.wy0 + (lane & 15)) * FD_Y + 16 * w + 4 * t + (lane >> 4)];
#pragma unroll
        for (int e = 0; e < 16; ++e) il[e] = F.ilT[col[e] * FD_Y + m];
    }
    __syncthreads();
    fd_mfma_y<FD_LD>(qys, B0, B1, w);    // B1 = Qy b
    __syncthreads();
    fd_mfma_x(bx, B1, B0, w);            // B0 = Qy b Qx
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        t2[e] = B0[m * FD_LD + col[e]] * il[e];
        B1[m * FD_LD + col[e]] = t2[e];
    }
    __syncthreads();                     // B1 = T2, B0 free
#else
    __syncthreads();
    fd_trans_rl<FD_Y, FD_LD>(qys, B0 + lane, lo, hi);          // rows 8w+r and 127-(8w+r), column lane
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r0 = 8 * w + 2 * q;
        B0[r0 * FD_LD + lane] = lo[q].x; B0[(r0 + 1) * FD_LD + lane] = lo[q].y;
        B0[(FD_Y - 1 - r0) * FD_LD + lane] = hi[q].x; B0[(FD_Y - 2 - r0) * FD_LD + lane] = hi[q].y;
    }
    __syncthreads();
    fd_trans<FD_X, 1>(F.Qx, B0 + m * FD_LD, cb, lo, hi);       // row m, columns 8cb+r and 63-(8cb+r)
#pragma unroll
    for (int q = 0; q < 4; ++q) { t2[2 * q] = lo[q].x; t2[2 * q + 1] = lo[q].y; t2[8 + 2 * q] = hi[q].x; t2[8 + 2 * q + 1] = hi[q].y; }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        il[e] = F.ilT[col[e] * FD_Y + m];
        t2[e] *= il[e];
        B1[m * FD_LD + col[e]] = t2[e];
    }
    __syncthreads();                     // B1 = T2, B0 free
#endif
    FD_STAMP(9);

    // ---- window values of G b:  u = T2 Qx[:, win] ;  x0w = Qy[win, :] u ----------------------
#if SOL_FD_MFMA
    {
        typedef float fd_f4 __attribute__((ext_vector_type(4)));
        fd_f4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* ap = B1 + (16 * w + (lane & 15)) * FD_LD + (lane >> 4);
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * t], qxw[t], acc, 0, 0, 0);
        float* up = U + (16 * w + 4 * (lane >> 4)) * FD_ULD + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) up[r * FD_ULD] = acc[r];
    }
#else
    {
        f2 u0 = {0.f, 0.f}, u1 = {0.f, 0.f};
        fd_cfp Qc = (fd_cfp)(F.QxW + 4 * cb);           // QxW[c][i'] = Qx[c][wx0 + i']: 64 B per c
        const float* bp = B1 + m * FD_LD;
#pragma unroll 8
        for (int c = 0; c < FD_X; ++c) {
            const float v = bp[c];
            const f2 vv = {v, v};
            fd_cfp qr = Qc + (size_t)c * FD_WIN;
            u0 += (f2){qr[0], qr[1]} * vv;
            u1 += (f2){qr[2], qr[3]} * vv;
        }
        float* up = U + m * FD_ULD + 4 * cb;
        up[0] = u0.x; up[1] = u0.y; up[2] = u1.x; up[3] = u1.y;
    }
#endif
    __syncthreads();
    FD_STAMP(10);
#if SOL_FD_MFMA
    {
        typedef float fd_f4 __attribute__((ext_vector_type(4)));
        fd_f4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* bp = U + (16 * w + (lane >> 4)) * FD_ULD + (lane & 15);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qyk[t], bp[4 * t * FD_ULD], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) XP8[w * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
        if (tid < 256) W2[tid] = 0.f;
    }
    // second request group: Qx[wx0 + i'][c] (spectral coefficients of the correction) and the window rows of Qy as A operand [16 m x 4 jw]
    // of t2w = Qy[:, win] W2 -- in flight behind the gather and the K' product
    float qxr[4][4], qym[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int t = 0; t < 4; ++t) qxr[ct][t] = F.Qx[(size_t)(F.wx0 + 4 * t + (lane >> 4)) * FD_X + 16 * ct + (lane & 15)];
#pragma unroll
    for (int t = 0; t < 4; ++t) qym[t] = F.Qy[(size_t)(F.wy0 + 4 * t + (lane >> 4)) * FD_Y + 16 * w + (lane & 15)];
    __syncthreads();
    FD_STAMP(11);
    if (tid < F.SP) {
        const int si = F.sidx[tid];
        float v = 0.f;
        if (si >= 0) {
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) v += XP8[ww * 256 + si];
        }
        XS[tid] = v;
    }
#else
    {
        const int h = tid >> 8, t = tid & 255, jw = t >> 4, iw = t & 15;
        const float* qrow = F.Qy + (size_t)(F.wy0 + jw) * FD_Y + 64 * h;
        const float* up = U + (64 * h) * FD_ULD + iw;
        float s = 0.f;
#pragma unroll 8
        for (int mm = 0; mm < 64; ++mm) s += qrow[mm] * up[mm * FD_ULD];
        XP[h * 256 + t] = s;
        if (tid < 256) W2[tid] = 0.f;
    }
    __syncthreads();
    FD_STAMP(11);
    if (tid < F.SP) {
        const int si = F.sidx[tid];
        XS[tid] = si >= 0 ? XP[si] + XP[256 + si] : 0.f;
    }
#endif
    __syncthreads();
    // ---- c = K' x0_S  (two halves of the sum), scattered with a minus sign into the window -----
    if (tid < 2 * F.SP) {
        const int h = tid >= F.SP ? 1 : 0, sidx_ = tid - h * F.SP, half = F.SP >> 1;
        const float* kp = F.KpT + (size_t)(h * half) * F.SP + sidx_;
        const float* xs = XS + h * half;
        float s = 0.f;
#pragma unroll 8         /* (unroll 32 measured: 2.4 -> 8.4 us) */
        for (int q = 0; q < half; ++q) s += kp[(size_t)q * F.SP] * xs[q];
        CP[h * 256 + sidx_] = s;
    }
    __syncthreads();
    FD_STAMP(12);
    if (tid < F.SP) {
        const int si = F.sidx[tid];
        if (si >= 0) W2[si] = -(CP[tid] + CP[256 + tid]);
    }
    __syncthreads();
    // ---- spectral coefficients of the correction: t2w = Qy[:, win] W2 ; T2 += (t2w Qx[win, :]) / lam
#if SOL_FD_MFMA
    {
        typedef float fd_f4 __attribute__((ext_vector_type(4)));
        fd_f4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* bp = W2 + (lane >> 4) * FD_WIN + (lane & 15);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qym[t], bp[4 * t * FD_WIN], acc, 0, 0, 0);
        float* up = U + (16 * w + 4 * (lane >> 4)) * FD_ULD + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) up[r * FD_ULD] = acc[r];
    }
#else
    {
        f2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
        float qw[FD_WIN];
#pragma unroll
        for (int jw = 0; jw < FD_WIN; ++jw) qw[jw] = F.Qy[(size_t)(F.wy0 + jw) * FD_Y + m];
        __builtin_amdgcn_sched_barrier(0);              // all 16 requests in flight before the first use (1.6 -> 0.7 us)
#pragma unroll
        for (int jw = 0; jw < FD_WIN; ++jw) {
            const float qv = qw[jw];
            const float4 wv = *reinterpret_cast<const float4*>(W2 + jw * FD_WIN + 4 * cb);
            const f2 qq = {qv, qv};
            a0 += qq * (f2){wv.x, wv.y};
            a1 += qq * (f2){wv.z, wv.w};
        }
        float* up = U + m * FD_ULD + 4 * cb;
        up[0] = a0.x; up[1] = a0.y; up[2] = a1.x; up[3] = a1.y;
    }
#endif
    __syncthreads();
    FD_STAMP(13);
#if SOL_FD_MFMA
    {
        // S[m][c] = sum_i' t2w[m][i'] Qx[wx0+i'][c] on the matrix cores (wave = rows 16w .. 16w+15, four 16-column tiles) -> XS2 (= B0 behind
        // the small scratch arrays ... B0 is free but U / W2 live in its head: the products go to B1's free twin rows, see below)
        typedef float fd_f4 __attribute__((ext_vector_type(4)));
        fd_f4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (fd_f4){0.f, 0.f, 0.f, 0.f};
        const float* ap = U + (16 * w + (lane & 15)) * FD_ULD + (lane >> 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float av = ap[4 * t];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, qxr[ct][t], acc[ct], 0, 0, 0);
        }
        __syncthreads();                 // every read of U (t2w) is done: S may overwrite B0
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) B0[(16 * w + 4 * (lane >> 4) + r) * FD_LD + 16 * ct + (lane & 15)] = acc[ct][r];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            t2[e] += il[e] * B0[m * FD_LD + col[e]];
            B1[m * FD_LD + col[e]] = t2[e];
        }
    }
#else
    {
        // T2[m][c] += il * sum_i' t2w[m][i'] Qx[wx0+i'][c] for this thread's 16 columns (8 contiguous + their 8 mirrors)
        f2 al[4], ah[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { al[q] = (f2){0.f, 0.f}; ah[q] = (f2){0.f, 0.f}; }
        fd_cfp Ql = (fd_cfp)(F.Qx + (size_t)F.wx0 * FD_X + 8 * cb);
        fd_cfp Qh = (fd_cfp)(F.Qx + (size_t)F.wx0 * FD_X + (FD_X - 8 - 8 * cb));
        const float* up = U + m * FD_ULD;
#pragma unroll 4
        for (int iw = 0; iw < FD_WIN; ++iw) {
            const float v = up[iw];
            const f2 vv = {v, v};
            fd_cfp ql = Ql + (size_t)iw * FD_X;
            fd_cfp qh = Qh + (size_t)iw * FD_X;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                al[q] += (f2){ql[2 * q], ql[2 * q + 1]} * vv;
                ah[q] += (f2){qh[7 - 2 * q], qh[6 - 2 * q]} * vv;       // columns 63-(8cb+2q), 62-(8cb+2q)
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            t2[2 * q] += il[2 * q] * al[q].x; t2[2 * q + 1] += il[2 * q + 1] * al[q].y;
            t2[8 + 2 * q] += il[8 + 2 * q] * ah[q].x; t2[8 + 2 * q + 1] += il[8 + 2 * q + 1] * ah[q].y;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) B1[m * FD_LD + col[e]] = t2[e];
    }
#endif
    __syncthreads();
    FD_STAMP(14);
    // ---- inverse transform: x = Qy (T2 Qx), written to B1 as [128][64] for the caller ---------------
#if SOL_FD_MFMA
    fd_mfma_x(bx, B1, B0, w);
    __syncthreads();                     // B0 = T3; every read of B1 (T2) is done
    FD_STAMP(15);
    fd_mfma_y<FD_X>(qys, B0, B1, w);
    __syncthreads();                     // solution complete in LDS
    return B1;
#else
    fd_trans<FD_X, 1>(F.Qx, B1 + m * FD_LD, cb, lo, hi);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c0 = 8 * cb + 2 * q;
        B0[m * FD_LD + c0] = lo[q].x; B0[m * FD_LD + c0 + 1] = lo[q].y;
        B0[m * FD_LD + FD_X - 1 - c0] = hi[q].x; B0[m * FD_LD + FD_X - 2 - c0] = hi[q].y;
    }
    __syncthreads();                     // B0 = T3; every read of B1 (T2) is done
    FD_STAMP(15);
    fd_trans_rl<FD_Y, FD_LD>(qys, B0 + lane, lo, hi);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r0 = 8 * w + 2 * q;
        B1[r0 * FD_X + lane] = lo[q].x; B1[(r0 + 1) * FD_X + lane] = lo[q].y;
        B1[(FD_Y - 1 - r0) * FD_X + lane] = hi[q].x; B1[(FD_Y - 2 - r0) * FD_X + lane] = hi[q].y;
    }
    __syncthreads();                     // solution complete in LDS
    return B1;
#endif
}

__device__ __forceinline__ float* fd_solve(const float* __restrict__ blob, const float (&qys)[FD_QYS], float* buf, const float (&rf)[16], long long* prof) {
    const FdOps none{};
    return fd_solve<false, false>(blob, qys, buf, rf, prof, none);
}

// ---- the same direct solve for small grids (Y*X <= 2048, e.g. the reference's 64x32 training recipe,
//      /root/reference/karman-2d/Makefile:78-80), any thread count ---------------------------------------------------------
// Follows precond.direct_solve_reference line by line:  T2 = (Qy b Qx) / lam;  x0w = Qy[win,:] T2 Qx[:,win];
// x_S = gather;  c = K' x_S;  W2 = -scatter(c);  T2 += (Qy[:,win] W2 Qx[win,:]) / lam;  x = Qy T2 Qx.
// Everything is LDS resident (the matrices are copied in first: 30 KB at 64x32); each product runs as 4x4 register tiles
// C[M][N] = sum_k At[k][M] B[k][N] with both operands read as 16-byte pieces (Qy, Qx are symmetric, so a transposed operand
// is the matrix itself; intermediate results are stored in the orientation their consumer needs).
// stage the matrices of the blob into the solver's LDS region; called at kernel start so that the copies travel behind the
// diffusion / advection phases (nothing else touches `ext`); the solve begins with a barrier
template <int NT = 0>     // NT: threads of the workgroup that run the solver (0: all of it)
__device__ __forceinline__ void fd_small_stage(const float* __restrict__ blob, int Y, int X, float* ext) {
    const int* h = reinterpret_cast<const int*>(blob);
    const int SP = h[6];
    const int tid = threadIdx.x, T = NT ? NT : (int)blockDim.x;
    const int n4 = (Y * Y + X * X + X * Y) / 4;              // Qy, Qx, 1/lam: contiguous in the blob and in LDS
    const float4* src = reinterpret_cast<const float4*>(blob + 16);
    float4* dst = reinterpret_cast<float4*>(ext);
#pragma unroll 8
    for (int e = tid; e < n4; e += T) dst[e] = src[e];
    const float* gKp = blob + 16 + (size_t)Y * Y + (size_t)X * X + (size_t)X * Y;
    const float* gQW = gKp + (size_t)SP * SP + SP;
    float* LQW = ext + Y * Y + X * X + X * Y;
    float* W2 = LQW + X * 16 + 2 * Y * X + Y * 16 + 256;
    float* KP = W2 + 256 + 256 + 256;
#pragma unroll 4
    for (int e = tid; e < X * 16; e += T) LQW[e] = gQW[e];
    for (int e = tid; e < 256; e += T) W2[e] = 0.f;
    if (SP <= 64) {
#pragma unroll 8
        for (int e = tid; e < SP * SP / 4; e += T) reinterpret_cast<float4*>(KP)[e] = reinterpret_cast<const float4*>(gKp)[e];
    }
    // QyW[m][j'] = Qy[m][wy0 + j'] (= Qy[wy0 + j'][m]: symmetric): the window rows as one compact, 16-byte aligned slab
    float* QyW = KP + 4096;
    const int wy0 = h[3];
    const float* gQy = blob + 16;
#pragma unroll 4
    for (int e = tid; e < Y * 16; e += T) QyW[e] = gQy[(size_t)(wy0 + (e & 15)) * Y + (e >> 4)];
}
template <int CPT, int NT = 0>
__device__ __forceinline__ float* fd_solve_small(const float* __restrict__ blob, int Y, int X, const Own& o, float* ext, const float (&rf)[CPT]) {
    const int* h = reinterpret_cast<const int*>(blob);
    const int wy0 = h[3], wx0 = h[4], SP = h[6];
    const float* gQy = blob + 16;
    const float* gQx = gQy + (size_t)Y * Y;
    const float* gIL = gQx + (size_t)X * X;            // [c][m]
    const float* gKp = gIL + (size_t)X * Y;            // KpT [SP][SP]
    const int* gsidx = reinterpret_cast<const int*>(gKp + (size_t)SP * SP);
    const float* gQW = reinterpret_cast<const float*>(gsidx + SP);      // [c][16]
    float* LQy = ext;
    float* LQx = LQy + Y * Y;
    float* LIL = LQx + X * X;
    float* LQW = LIL + X * Y;
    float* B0 = LQW + X * 16;          // [Y][X] / [X][Y]
    float* B1 = B0 + Y * X;
    float* U = B1 + Y * X;             // [Y][16] / [16][Y]
    float* XW = U + Y * 16;            // [16][16]
    float* W2 = XW + 256;
    float* XS = W2 + 256;              // [SP] (SP <= 256)
    float* CP = XS + 256;
    float* KP = CP + 256;              // K'^T [SP][SP] when SP <= 64 (fd_small_stage)
    float* QyW = KP + 4096;            // [Y][16] window rows of Qy
    const int tid = threadIdx.x, T = NT ? NT : (int)blockDim.x;
    (void)gQy; (void)gQW;
    if (o.owner) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) B0[(o.j0 + k) * X + o.i] = rf[k];
    }
    __syncthreads();
    // C tile loop: MODE 0: C[m][n] = v;  1: Ct[n][m] = v;  2: Ct[n][m] = v * S[n][m];  3: Ct[n][m] += v * S[n][m]
    // Grids whose sides are multiples of 16 (64x32, 32x16): the products run as 16 x 16 tiles of v_mfma_f32_16x16x4_f32, one tile per wave
    // and trip -- A operand lane (m = lane & 15, k = lane >> 4) = At[k][m0 + m], B operand = Bm[k][n0 + n], both contiguous LDS reads; tile
    // element r of a lane: row m0 + 4 (lane >> 4) + r, column n0 + (lane & 15).  The 4 x 4 register-tile form below kept half of the 256
    // threads busy in the large products and spent two LDS reads per 16 multiply-adds.
    const bool mfma_ok = SOL_FD_MFMA && (Y & 15) == 0 && (X & 15) == 0;
    auto mm = [&](const float* At, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N, int K, int mode, const float* S) {
        if (mfma_ok) {
            typedef float fd_f4 __attribute__((ext_vector_type(4)));
            const int lane = tid & 63, wv = tid >> 6, nw = T >> 6, tn16 = N >> 4, tiles16 = (M >> 4) * tn16;
            for (int t = wv; t < tiles16; t += nw) {
                const int m0 = (t / tn16) << 4, n0 = (t % tn16) << 4;
                const float* ap = At + (lane >> 4) * lda + m0 + (lane & 15);
                const float* bp = Bm + (lane >> 4) * ldb + n0 + (lane & 15);
                fd_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k * lda], bp[k * ldb], acc, 0, 0, 0);
                const int n = n0 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 4 * (lane >> 4) + r;
                    if (mode == 0) C[m * ldc + n] = acc[r];
                    else if (mode == 1) C[n * ldc + m] = acc[r];
                    else if (mode == 2) C[n * ldc + m] = acc[r] * S[n * ldc + m];
                    else if (mode == 3) C[n * ldc + m] += acc[r] * S[n * ldc + m];
                    else C[m * ldc + n] += acc[r] * S[m * ldc + n];
                }
            }
            return;
        }
        const int tn = N >> 2, tiles = (M >> 2) * tn;
        for (int t = tid; t < tiles; t += T) {
            const int m0 = (t / tn) << 2, n0 = (t % tn) << 2;
            float acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
#pragma unroll 4
            for (int k = 0; k < K; ++k) {
                const float4 av = *reinterpret_cast<const float4*>(At + k * lda + m0);
                const float4 bv = *reinterpret_cast<const float4*>(Bm + k * ldb + n0);
                const float aa[4] = {av.x, av.y, av.z, av.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += aa[i] * bb[j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (mode == 0) C[(m0 + i) * ldc + n0 + j] = acc[i][j];
                    else if (mode == 1) C[(n0 + j) * ldc + m0 + i] = acc[i][j];
                    else if (mode == 2) C[(n0 + j) * ldc + m0 + i] = acc[i][j] * S[(n0 + j) * ldc + m0 + i];
                    else if (mode == 3) C[(n0 + j) * ldc + m0 + i] += acc[i][j] * S[(n0 + j) * ldc + m0 + i];
                    else C[(m0 + i) * ldc + n0 + j] += acc[i][j] * S[(m0 + i) * ldc + n0 + j];      // 4: in place, not transposed
                }
        }
    };
    mm(LQy, Y, B0, X, B1, Y, Y, X, Y, 1, nullptr);          // P1t[c][j] = (Qy b)^T
    __syncthreads();
    mm(B1, Y, LQx, X, B0, Y, Y, X, X, 2, LIL);              // T2t[c][m] = (P1 Qx)[m][c] / lam
    __syncthreads();
    mm(B0, Y, LQW, 16, U, 16, Y, 16, X, 0, nullptr);        // u[m][i'] = sum_c T2[m][c] Qx[c][wx0+i']
    __syncthreads();
    mm(QyW, 16, U, 16, XW, 16, 16, 16, Y, 0, nullptr);      // x0w[j'][i'] = sum_m Qy[wy0+j'][m] u[m][i']
    __syncthreads();
    for (int e = tid; e < SP; e += T) { const int si = gsidx[e]; XS[e] = si >= 0 ? XW[si] : 0.f; }
    __syncthreads();
    for (int e = tid; e < SP; e += T) {                     // c = K' x_S
        float sacc = 0.f;
        if (SP <= 64) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
            for (int q = 0; q < SP; q += 4) {
                s0 += KP[q * SP + e] * XS[q]; s1 += KP[(q + 1) * SP + e] * XS[q + 1];
                s2 += KP[(q + 2) * SP + e] * XS[q + 2]; s3 += KP[(q + 3) * SP + e] * XS[q + 3];
            }
            sacc = (s0 + s1) + (s2 + s3);
        } else {
#pragma unroll 16
            for (int q = 0; q < SP; ++q) sacc += gKp[(size_t)q * SP + e] * XS[q];      // loads issued 16 deep
        }
        CP[e] = sacc;
    }
    __syncthreads();
    for (int e = tid; e < SP; e += T) { const int si = gsidx[e]; if (si >= 0) W2[si] = -CP[e]; }
    __syncthreads();
    mm(LQy + wy0 * Y, Y, W2, 16, U, Y, Y, 16, 16, 1, nullptr);      // t2wT[i'][m] = sum_j' Qy[wy0+j'][m] W2[j'][i']
    __syncthreads();
    mm(LQx + wx0 * X, X, U, Y, B0, Y, X, Y, 16, 4, LIL);            // T2t[c][m] += (sum_i' Qx[wx0+i'][c] t2w[m][i']) / lam
    __syncthreads();
    mm(B0, Y, LQx, X, B1, X, Y, X, X, 0, nullptr);          // P3[m][c] = T2 Qx
    __syncthreads();
    mm(LQy, Y, B1, X, B0, X, Y, X, Y, 0, nullptr);          // x[j][c] = Qy P3
    __syncthreads();
    return B0;
}

// per-cell matrix coefficients of the owned strip
template <int CPT>
__device__ __forceinline__ void cell_coeffs(const Own& o, const unsigned char* act, int Y, int X,
                                            float (&dg)[CPT], float (&ac)[CPT]) {
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        dg[k] = 1.f;
        ac[k] = 1.f;    // non-owner lanes: identity rows (r = 0 there)
        if (o.owner) {
            const int j = o.j0 + k, i = o.i;
            ac[k] = (float)act[j * X + i];
            const float nacc = acc_at(act, Y, X, j - 1, i) + acc_at(act, Y, X, j + 1, i) +
                               acc_at(act, Y, X, j, i - 1) + acc_at(act, Y, X, j, i + 1);
            dg[k] = fmaxf(nacc, 1.f);
        }
    }
}

// ------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------
// SOLVER: 0 = plain CG, 1 = two-level preconditioned CG, 2 = direct (fast diagonalisation); separate instantiations keep
// each variant's register allocation independent (1 and 2 exist for the 16-cell strips only)
template <int CPT, int SOLVER>
__device__ __forceinline__ void karman_fwd_body(const StepArgs& a, float* smem) {
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int Y = a.Y, X = a.X, N = Y * X, nVy = (Y + 1) * X, nVx = Y * (X + 1), XP = X + 1;
    const int lx = __ffs(X) - 1;                 // X is a power of two: k / X == k >> lx
    const float invXP = 1.f / (float)XP;         // k / XP == (int)((k + 0.5) * invXP), exact for k < 2^22
    const Lds L = carve(smem, Y, X, CPT);

    SOL_STAMP(0);
    float fdp = 0.f;
    if constexpr (SOLVER == 2) {
        if (Y == FD_Y) fdp = fd_prefetch(a.fd, a.fd_n);
        else fd_small_stage(a.fd, Y, X, L.fdx);
    }
    // ---- phase 1: load inputs (all global loads in flight before the first LDS store) ---
    // 16-byte loads: a dword load costs the texture path as much per wave as a dwordx4 one, and this phase is nothing else
    // (X, Y are multiples of 4, so every array and every batch offset is 16-byte aligned and a v_y quad stays in one row)
    constexpr int NV = (CPT + 1 + 3) / 4;      // float4 items of a (Y+1) x X field per thread
    const int nQy = nVy >> 2, nQx = nVx >> 2;
    float4 bcv_r[NV], bcm_r[NV];               // velocity BC of this thread's v_y quads: loaded here, used by phase 2
    {
        const float4* gvy = reinterpret_cast<const float4*>(a.vy_in + (size_t)b * nVy);
        const float4* gvx = reinterpret_cast<const float4*>(a.vx_in + (size_t)b * nVx);
        const float4* bcv = reinterpret_cast<const float4*>(a.bcv + (size_t)b * a.bc_stride);
        const float4* bcm = reinterpret_cast<const float4*>(a.bcm + (size_t)b * a.bc_stride);
        const float4* gact = reinterpret_cast<const float4*>(a.active);
        float4 ty[NV], tx[NV], ta[CPT / 4];
#pragma unroll
        for (int n = 0; n < NV; ++n) { const int q = tid + n * nthr; ty[n] = gvy[min(q, nQy - 1)]; tx[n] = gvx[min(q, nQx - 1)]; }   // branch-free: clamped index
#pragma unroll
        for (int n = 0; n < CPT / 4; ++n) ta[n] = gact[min(tid + n * nthr, (N >> 2) - 1)];
#pragma unroll
        for (int n = 0; n < NV; ++n) { const int q = min(tid + n * nthr, nQy - 1); bcv_r[n] = bcv[q]; bcm_r[n] = bcm[q]; }
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = tid + n * nthr;
            if (q < nQy) reinterpret_cast<float4*>(L.Avy)[q] = ty[n];
            if (q < nQx) reinterpret_cast<float4*>(L.Avx)[q] = tx[n];
        }
#pragma unroll
        for (int n = 0; n < CPT / 4; ++n) {
            const int q = tid + n * nthr;
            if (q < (N >> 2)) reinterpret_cast<uchar4*>(L.act)[q] = make_uchar4(ta[n].x != 0.f, ta[n].y != 0.f, ta[n].z != 0.f, ta[n].w != 0.f);
        }
    }
    __syncthreads();
    SOL_STAMP(1);

    // ---- phase 2: explicit diffusion (replicate padding, dx = 1) + velocity BC ------
    if (!(a.dbg & 8)) {
        const float alpha = a.adt / a.re[b];
#pragma unroll
        for (int n = 0; n < NV; ++n) {             // compile-time trip count: the BC values sit in registers
            const int q = tid + n * nthr;
            if (q >= nQy) continue;
            const int k = q << 2, j = k >> lx, i = k & (X - 1);
            const float4 c = reinterpret_cast<const float4*>(L.Avy)[q];
            const float4 up = *reinterpret_cast<const float4*>(&L.Avy[min(j + 1, Y) * X + i]);
            const float4 dn = *reinterpret_cast<const float4*>(&L.Avy[max(j - 1, 0) * X + i]);
            const float rt = L.Avy[j * X + min(i + 4, X - 1)], lf = L.Avy[j * X + max(i - 1, 0)];
            float4 v;
            v.x = dif_y(c.x, up.x, dn.x, c.y, lf, alpha, bcm_r[n].x, bcv_r[n].x);
            v.y = dif_y(c.y, up.y, dn.y, c.z, c.x, alpha, bcm_r[n].y, bcv_r[n].y);
            v.z = dif_y(c.z, up.z, dn.z, c.w, c.y, alpha, bcm_r[n].z, bcv_r[n].z);
            v.w = dif_y(c.w, up.w, dn.w, rt, c.z, alpha, bcm_r[n].w, bcv_r[n].w);
            reinterpret_cast<float4*>(L.Bvy)[q] = v;
            if (a.saved_vy) reinterpret_cast<float4*>(a.saved_vy + (size_t)b * nVy)[q] = v;
        }
        #pragma unroll 4
        for (int k = tid; k < nVx; k += nthr) {
            const int j = (int)(((float)k + 0.5f) * invXP), i = k - j * XP;
            const float c = L.Avx[k];
            const float lap = L.Avx[min(j + 1, Y - 1) * XP + i] + L.Avx[max(j - 1, 0) * XP + i] +
                              L.Avx[j * XP + min(i + 1, X)] + L.Avx[j * XP + max(i - 1, 0)] - 4.f * c;
            const float v = dif_x(c, lap, alpha);
            L.Bvx[k] = v;
            if (a.saved_vx) a.saved_vx[(size_t)b * nVx + k] = v;
        }
    }
    __syncthreads();
    SOL_STAMP(2);

    // ---- phase 3: semi-Lagrangian advection (B -> A), hard-BC face mask fused --------
    if (!(a.dbg & 1))
    #pragma unroll 4
    for (int k = tid; k < nVy; k += nthr) {
        const int j = k >> lx, i = k & (X - 1);
        const float uy = L.Bvy[k];
        const int ja = max(j - 1, 0), jb = min(j, Y - 1);
        const float ux = 0.25f * (L.Bvx[ja * XP + i] + L.Bvx[ja * XP + i + 1] + L.Bvx[jb * XP + i] + L.Bvx[jb * XP + i + 1]);
        const Bil s = bil_clamp(Y + 1, X, j, -uy * a.dtdx, i, -ux * a.dtdx);
        L.Avy[k] = bil_eval(L.Bvy, X, s) * mask_y(L.act, Y, X, j, i);
    }
    #pragma unroll 4
    for (int k = tid; k < nVx; k += nthr) {
        const int j = (int)(((float)k + 0.5f) * invXP), i = k - j * XP;
        const float ux = L.Bvx[k];
        const int ia = max(i - 1, 0), ib = min(i, X - 1);
        const float uy = 0.25f * (L.Bvy[j * X + ia] + L.Bvy[j * X + ib] + L.Bvy[(j + 1) * X + ia] + L.Bvy[(j + 1) * X + ib]);
        const Bil s = bil_clamp(Y, XP, j, -uy * a.dtdx, i, -ux * a.dtdx);
        L.Avx[k] = bil_eval(L.Bvx, XP, s) * mask_x(L.act, Y, X, j, i);
    }
    SOL_STAMP(3);
    if (a.d_out && !(a.dbg & 2)) {
        const float* gd = a.d_in + (size_t)b * N;
        #pragma unroll 8
        for (int k = tid; k < N; k += nthr) {
            const int j = k >> lx, i = k & (X - 1);
            const float uy = 0.5f * (L.Bvy[k] + L.Bvy[k + X]);
            const float ux = 0.5f * (L.Bvx[j * XP + i] + L.Bvx[j * XP + i + 1]);
            const float oy = -uy * a.dtdx, ox = -ux * a.dtdx;
            const float fy = floorf(oy), fx = floorf(ox);
            const float wy = oy - fy, wx = ox - fx;
            const int j0 = j + (int)fy, i0 = i + (int)fx;
            float f[2][2];
#pragma unroll
            for (int dj = 0; dj < 2; ++dj)
#pragma unroll
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
            if (!a.inflow_before) v += a.inflow[k] * a.dt;
            a.d_out[(size_t)b * N + k] = v;
        }
    }
    __syncthreads();
    SOL_STAMP(4);

    // ---- phase 4/5: divergence + CG pressure solve ----------------------------------
    float qys[FD_QYS];              // direct solver: this wave's slice of Qy (L2-warm: fd_prefetch), in flight behind the setup
#pragma unroll
    for (int j = 0; j < FD_QYS; ++j) qys[j] = 0.f;
    if constexpr (SOLVER == 2 && CPT == 16) { if (Y == FD_Y) fd_load_qys(a.fd + 16, __builtin_amdgcn_readfirstlane(tid >> 6), qys); }
    const Own o = ownership<CPT>(Y, X);
    float dg[CPT], ac[CPT], r[CPT], x[CPT];
    cell_coeffs<CPT>(o, L.act, Y, X, dg, ac);
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        r[k] = 0.f;
        if (o.owner) {
            const int j = o.j0 + k, i = o.i;
            const float div = (L.Avy[(j + 1) * X + i] - L.Avy[j * X + i]) + (L.Avx[j * XP + i + 1] - L.Avx[j * XP + i]);
            r[k] = -div;   // M p = -div  <=>  A p = div
        }
    }
    int it = 0;
    float* Pfd = nullptr;           // direct solver: the solution arrives as an LDS array
    SOL_STAMP(5);
    if constexpr (SOLVER == 2) {            // host guarantees 128 x 64 (register-tiled) or a small grid (LDS resident)
        if constexpr (CPT == 16) Pfd = Y == FD_Y ? fd_solve(a.fd, qys, L.Bvy, r, a.prof) : fd_solve_small<CPT>(a.fd, Y, X, o, L.fdx, r);
        else Pfd = fd_solve_small<CPT>(a.fd, Y, X, o, L.fdx, r);
    } else if constexpr (SOLVER == 1) {
        it = X == 64 ? pcg_solve<CPT, true, 32>(o, Y, X, L.act, dg, ac, r, x, L.E, L.red, L.cs, a.cinv, a.rtol2, a.atol2, a.max_iter)
                     : pcg_solve<CPT, false, 8>(o, Y, X, L.act, dg, ac, r, x, L.E, L.red, L.cs, a.cinv, a.rtol2, a.atol2, a.max_iter);
    } else {
        it = X == 64 ? cg_solve<CPT, true>(o, X, dg, ac, r, x, L.E, L.red, a.rtol2, a.atol2, a.max_iter)
                     : cg_solve<CPT, false>(o, X, dg, ac, r, x, L.E, L.red, a.rtol2, a.atol2, a.max_iter);
    }
    if (a.iters && tid == 0) a.iters[b] = it;
    SOL_STAMP(6);

    // ---- phase 6: v -= mask * grad p ;  outputs --------------------------------------
    float* P = L.Bvy;   // region B is free after the advection
    if constexpr (SOLVER == 2) P = Pfd;
    else {
        if (o.owner) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) P[(o.j0 + k) * X + o.i] = x[k];
        }
        __syncthreads();
    }
    {
        float* gvy = a.vy_out + (size_t)b * nVy;
        float* gvx = a.vx_out + (size_t)b * nVx;
        #pragma unroll 2
        for (int q = tid; q < nQy; q += nthr) {        // four faces of one row at a time
            const int k = q << 2, j = k >> lx, i = k & (X - 1);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j >= 1 && j <= Y - 1) {
                const float4 p1 = *reinterpret_cast<const float4*>(&P[j * X + i]), p0 = *reinterpret_cast<const float4*>(&P[(j - 1) * X + i]);
                g = make_float4(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z, p1.w - p0.w);
            } else if (a.grad_pad == 1) {
                const float4 p = *reinterpret_cast<const float4*>(&P[(j == 0 ? 0 : Y - 1) * X + i]);
                g = j == 0 ? p : make_float4(-p.x, -p.y, -p.z, -p.w);
            }
            const uchar4 m0 = *reinterpret_cast<const uchar4*>(&L.act[max(j - 1, 0) * X + i]), m1 = *reinterpret_cast<const uchar4*>(&L.act[min(j, Y - 1) * X + i]);
            float4 v = reinterpret_cast<const float4*>(L.Avy)[q];
            v.x -= (float)min(m0.x, m1.x) * g.x; v.y -= (float)min(m0.y, m1.y) * g.y;
            v.z -= (float)min(m0.z, m1.z) * g.z; v.w -= (float)min(m0.w, m1.w) * g.w;
            reinterpret_cast<float4*>(L.Avy)[q] = v;
            reinterpret_cast<float4*>(gvy)[q] = v;
        }
        #pragma unroll 4
        for (int k = tid; k < nVx; k += nthr) {
            const int j = (int)(((float)k + 0.5f) * invXP), i = k - j * XP;
            float g = 0.f;
            if (i >= 1 && i <= X - 1) g = P[j * X + i] - P[j * X + i - 1];
            else if (a.grad_pad == 1) g = (i == 0) ? P[j * X] : -P[j * X + X - 1];
            const float v = L.Avx[k] - mask_x(L.act, Y, X, j, i) * g;
            L.Avx[k] = v;
            gvx[k] = v;
        }
    }
    SOL_STAMP(7);
    if (a.feat) {   // fused to_feature + 1/std scaling (karman_train.py:77-86,416-419)
        __syncthreads();
        float4* gf = reinterpret_cast<float4*>(a.feat) + (size_t)b * N;
        const float rech = a.re[b] * a.fs2;
        if (a.feat_tr) {        // cell (j, i) -> position i * Y + j: coalesced stores, column reads of the LDS fields (a few bank-conflicted reads per thread)
            #pragma unroll 4
            for (int k = tid; k < N; k += nthr) {
                const int i = k / Y, j = k - i * Y;
                gf[k] = make_float4(L.Avy[j * X + i] * a.fs0, L.Avx[j * XP + i] * a.fs1, rech, 0.f);
            }
        } else {
            #pragma unroll 4
            for (int k = tid; k < N; k += nthr) {
                const int j = k >> lx, i = k & (X - 1);
                gf[k] = make_float4(L.Avy[k] * a.fs0, L.Avx[j * XP + i] * a.fs1, rech, 0.f);
            }
        }
    }
    SOL_STAMP(8);
    if (fdp == 1.2345678e-30f && a.iters) a.iters[b] = -2;      // never true: keeps the prefetch loads alive
}

template <int CPT, int SOLVER>
__global__ void __launch_bounds__(CPT == 16 ? 512 : 1024) k_karman_fwd(StepArgs a) {
    extern __shared__ __align__(16) float smem[];
    karman_fwd_body<CPT, SOLVER>(a, smem);
}

// One advection step of the passive density of ONE simulation straight from global memory (same arithmetic as phase 3 /
// k_density_chain).  In the training graph these workgroups ride in the NEXT step's solver launch (k_karman_fwd_dens): the
// density of step i-1 only needs the saved post-diffusion velocity of step i-1, and 250 CUs idle during a solver launch.
struct DensStep {
    int B, Y, X, inflow_before;
    float dtdx, dt;
    const float *d_in, *svy, *svx, *inflow;   // [B][...] of the step being advected
    float* d_out;
};
__device__ __forceinline__ void density_step_body(const DensStep& q, int b) {
    const int Y = q.Y, X = q.X, N = Y * X, XP = X + 1;
    const float* gd = q.d_in + (size_t)b * N;
    const float* sy = q.svy + (size_t)b * (Y + 1) * X;
    const float* sx = q.svx + (size_t)b * Y * XP;
#pragma unroll 4
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        const int j = k / X, i = k - j * X;
        const float uy = 0.5f * (sy[k] + sy[k + X]);
        const float ux = 0.5f * (sx[j * XP + i] + sx[j * XP + i + 1]);
        const float oy = -uy * q.dtdx, ox = -ux * q.dtdx;
        const float fy = floorf(oy), fx = floorf(ox);
        const float wy = oy - fy, wx = ox - fx;
        const int j0 = j + (int)fy, i0 = i + (int)fx;
        float f[2][2];
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int di = 0; di < 2; ++di) {
                const int jj = j0 + dj, ii = i0 + di;
                float v = 0.f;   // extrapolation 'constant': one ring of zero ghost cells
                if (jj >= 0 && jj < Y && ii >= 0 && ii < X) {
                    v = gd[jj * X + ii];
                    if (q.inflow_before) v += q.inflow[jj * X + ii];
                }
                f[dj][di] = v;
            }
        float v = (1.f - wy) * ((1.f - wx) * f[0][0] + wx * f[0][1]) + wy * ((1.f - wx) * f[1][0] + wx * f[1][1]);
        if (!q.inflow_before) v += q.inflow[k] * q.dt;
        q.d_out[(size_t)b * N + k] = v;
    }
}
__global__ void __launch_bounds__(512) k_karman_fwd_dens(StepArgs a, DensStep q) {
    extern __shared__ __align__(16) float smem[];
    if ((int)blockIdx.x < a.B) karman_fwd_body<16, 2>(a, smem);
    else density_step_body(q, (int)blockIdx.x - a.B);
}
__global__ void __launch_bounds__(512) k_density_step(DensStep q) { density_step_body(q, blockIdx.x); }

// ------------------------------------------------------------------------------------
// forward, 128 x 64, direct solver: FOUR workgroups per simulation (round 6)
// ------------------------------------------------------------------------------------
// The stencil phases of the one-workgroup kernel (load, diffusion + BC, advection, divergence, projection, outputs: 28 of its 44 us at
// B = 6) are bound by the vector ALU of ONE compute unit -- 2 waves per SIMD run ~100 instructions per face -- while 250 CUs idle.  Here a
// simulation is cut into BD_N bands of BD_R cell rows; every band's workgroup recomputes the diffusion on BD_HA halo rows each side
// (departure points of the semi-Lagrangian step are then band local for |u| dt / dx < BD_HA; a face whose departure point lies outside the
// halo recomputes the diffused corner values it needs from the step's input in global memory: same result, slow, and only the garbage
// states of an untrained network ever take it), advects its faces and forms its rows of the divergence.  Two hand-offs through global memory: the divergence rows of bands 1.. go to band 0's workgroup, which runs the
// direct solve on its CU as before (the transforms couple every cell with every other: DESIGN_HISTORY section 8 prices their split) and
// sends every band the pressure rows its projection needs.  "The data is the flag" (cdna_hip_programming.md, guideline 16): a word is
// stored as bits ^ BD_KEY with write-through (sc1) stores and the exchange region holds zeros otherwise, so the consumer polls its own
// 16-byte pieces with sc1 loads until no word is zero (bits == BD_KEY is a NaN payload no arithmetic produces), decodes, and restores the
// zeros for the next launch -- no flag, no fence, no drain.  A band that never arrives (spin limit) leaves NaNs, not stale numbers.
// Workgroup u = 32 g + 8 w + (b & 7) for simulation b = 8 g + (b & 7), band w: the four workgroups of a simulation run on ONE XCD
// (workgroups are dealt round robin to the eight XCDs) and meet in its L2; correctness does not depend on it (sc1 = agent scope).
// Arithmetic, operation order and therefore every output bit equal the one-workgroup kernel's (tests/test_gpu_parity.py).
constexpr int BD_N = 4, BD_R = FD_Y / BD_N, BD_HA = 8, BD_PROWS = BD_R + 1;
constexpr unsigned BD_KEY = 0x7fc0deadu;
constexpr int BD_DIV_WORDS = FD_Y * FD_X, BD_P_WORDS = BD_N * BD_PROWS * FD_X, BD_WORDS = BD_DIV_WORDS + BD_P_WORDS;
constexpr unsigned BD_SPIN_LIMIT = 1u << 15;
__device__ __forceinline__ uint4 bd_load(const __amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));      // aux 16 = sc1: agent scope, misses the vector L1
}
__device__ __forceinline__ void bd_store(const __amdgpu_buffer_rsrc_t r, int byte_off, const uint4& v) {
    typedef unsigned bd_u4 __attribute__((ext_vector_type(4)));
    const bd_u4 q = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(q, r, byte_off, 0, 16);
}
__device__ __forceinline__ bool bd_valid(const uint4& v) { return v.x != 0u && v.y != 0u && v.z != 0u && v.w != 0u; }
__device__ __forceinline__ float4 bd_decode(const uint4& v) {
    return make_float4(__uint_as_float(v.x ^ BD_KEY), __uint_as_float(v.y ^ BD_KEY), __uint_as_float(v.z ^ BD_KEY), __uint_as_float(v.w ^ BD_KEY));
}
__device__ __forceinline__ uint4 bd_encode(const float4& v) {
    return make_uint4(__float_as_uint(v.x) ^ BD_KEY, __float_as_uint(v.y) ^ BD_KEY, __float_as_uint(v.z) ^ BD_KEY, __float_as_uint(v.w) ^ BD_KEY);
}
// poll NQ 16-byte pieces (byte offsets off[], piece n wanted iff on[n]) until every word of every wanted piece of the WAVE is there
template <int NQ>
__device__ __forceinline__ void bd_recv(const __amdgpu_buffer_rsrc_t r, const int (&off)[NQ], const bool (&on)[NQ], uint4 (&v)[NQ]) {
    unsigned spins = 0;
    for (;;) {
#pragma unroll
        for (int n = 0; n < NQ; ++n) v[n] = bd_load(r, off[n]);
        bool ok = true;
#pragma unroll
        for (int n = 0; n < NQ; ++n) ok = ok && (!on[n] || bd_valid(v[n]));
        if (__all(ok) || ++spins > BD_SPIN_LIMIT) break;
        __builtin_amdgcn_s_sleep(2);
    }
}

#define BD_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")   /* LDS only: global requests stay in flight */
#define BD_STAMP(i) do { if (a.prof && blockIdx.x == 8 && threadIdx.x == 0) a.prof[20 + (i)] = wall_clock64(); } while (0)   /* band 1 of simulation 0 */
__device__ __forceinline__ void karman_fwd_band_body(const StepArgs& a, float* smem, const int b, const int w, unsigned* xw) {
    constexpr int Y = FD_Y, X = FD_X, XP = X + 1, N = Y * X, nVy = (Y + 1) * X, nVx = Y * XP, nthr = 512, lx = 6;
    const int tid = threadIdx.x;
    const float invXP = 1.f / (float)XP;
    const Lds L = carve(smem, Y, X, 16);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(xw, 0, BD_WORDS * 4, 0x00020000);
    // row ranges of this band (cells c0 .. c1-1)
    const int c0 = w * BD_R, c1 = c0 + BD_R;
    const int fy1 = w == BD_N - 1 ? Y + 1 : c1;                                  // owned v_y face rows [c0, fy1)
    const int ay1 = min(c1 + 1, Y + 1);                                          // advected v_y rows [c0, ay1): the divergence of row c1-1 needs face row c1
    const int dy0 = max(c0 - BD_HA, 0), dy1 = min(c1 + BD_HA + 1, Y + 1);        // diffused v_y rows
    const int dx0 = max(c0 - BD_HA, 0), dx1 = min(c1 + BD_HA, Y);                // diffused v_x rows
    const int ly0 = max(dy0 - 1, 0), ly1 = min(dy1 + 1, Y + 1);                  // loaded rows
    const int lx0 = max(dx0 - 1, 0), lx1 = min(dx1 + 1, Y);
    const int m0 = max(c0 - 1, 0), m1 = min(c1 + 1, Y);                          // mask rows

    SOL_STAMP(0);
    BD_STAMP(0);

    // ---- phase 1: load (16-byte pieces, all requests in flight before the first LDS store) ----
    constexpr int NV = 2;                       // <= 51 rows x 16 quads (v_y), <= 50 rows x 65 floats (v_x): at most 816 quads each
    float4 bcv_r[NV], bcm_r[NV];
    const int qy0 = ly0 * (X / 4), qy1 = ly1 * (X / 4);
    const int qx0 = (lx0 * XP) >> 2, qx1 = min((lx1 * XP + 3) >> 2, nVx >> 2);
    const int qd0 = dy0 * (X / 4), qd1 = dy1 * (X / 4);
    const int qm0 = m0 * (X / 4), qm1 = m1 * (X / 4);
    {
        const float4* gvy = reinterpret_cast<const float4*>(a.vy_in + (size_t)b * nVy);
        const float4* gvx = reinterpret_cast<const float4*>(a.vx_in + (size_t)b * nVx);
        const float4* bcv = reinterpret_cast<const float4*>(a.bcv + (size_t)b * a.bc_stride);
        const float4* bcm = reinterpret_cast<const float4*>(a.bcm + (size_t)b * a.bc_stride);
        const float4* gact = reinterpret_cast<const float4*>(a.active);
        float4 ty[NV], tx[NV], ta[NV];
#pragma unroll
        for (int n = 0; n < NV; ++n) { ty[n] = gvy[min(qy0 + tid + n * nthr, qy1 - 1)]; tx[n] = gvx[min(qx0 + tid + n * nthr, qx1 - 1)]; }
#pragma unroll
        for (int n = 0; n < NV; ++n) ta[n] = gact[min(qm0 + tid + n * nthr, qm1 - 1)];
#pragma unroll
        for (int n = 0; n < NV; ++n) { const int q = min(qd0 + tid + n * nthr, qd1 - 1); bcv_r[n] = bcv[q]; bcm_r[n] = bcm[q]; }
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int qy = qy0 + tid + n * nthr, qx = qx0 + tid + n * nthr, qm = qm0 + tid + n * nthr;
            if (qy < qy1) reinterpret_cast<float4*>(L.Avy)[qy] = ty[n];
            if (qx < qx1) reinterpret_cast<float4*>(L.Avx)[qx] = tx[n];
            if (qm < qm1) reinterpret_cast<uchar4*>(L.act)[qm] = make_uchar4(ta[n].x != 0.f, ta[n].y != 0.f, ta[n].z != 0.f, ta[n].w != 0.f);
        }
    }
    // the solver's coefficients (100 registers): requested behind the state, consumed 5 us later.  The barriers up to the solve wait for LDS
    // only (__syncthreads would wait for these requests as well)
    BD_BARRIER();
    SOL_STAMP(1);

    // ---- phase 2: explicit diffusion + velocity BC on the band and its halo ----
    {
        const float alpha = a.adt / a.re[b];
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = qd0 + tid + n * nthr;
            if (q >= qd1) continue;
            const int k = q << 2, j = k >> lx, i = k & (X - 1);
            const float4 c = reinterpret_cast<const float4*>(L.Avy)[q];
            const float4 up = *reinterpret_cast<const float4*>(&L.Avy[min(j + 1, Y) * X + i]);
            const float4 dn = *reinterpret_cast<const float4*>(&L.Avy[max(j - 1, 0) * X + i]);
            const float rt = L.Avy[j * X + min(i + 4, X - 1)], lf = L.Avy[j * X + max(i - 1, 0)];
            float4 v;
            v.x = dif_y(c.x, up.x, dn.x, c.y, lf, alpha, bcm_r[n].x, bcv_r[n].x);
            v.y = dif_y(c.y, up.y, dn.y, c.z, c.x, alpha, bcm_r[n].y, bcv_r[n].y);
            v.z = dif_y(c.z, up.z, dn.z, c.w, c.y, alpha, bcm_r[n].z, bcv_r[n].z);
            v.w = dif_y(c.w, up.w, dn.w, rt, c.z, alpha, bcm_r[n].w, bcv_r[n].w);
            reinterpret_cast<float4*>(L.Bvy)[q] = v;
            if (a.saved_vy && j >= c0 && j < fy1) reinterpret_cast<float4*>(a.saved_vy + (size_t)b * nVy)[q] = v;
        }
        #pragma unroll 4
        for (int k = dx0 * XP + tid; k < dx1 * XP; k += nthr) {
            const int j = (int)(((float)k + 0.5f) * invXP), i = k - j * XP;
            const float c = L.Avx[k];
            const float lap = L.Avx[min(j + 1, Y - 1) * XP + i] + L.Avx[max(j - 1, 0) * XP + i] +
                              L.Avx[j * XP + min(i + 1, X)] + L.Avx[j * XP + max(i - 1, 0)] - 4.f * c;
            const float v = dif_x(c, lap, alpha);
            L.Bvx[k] = v;
            if (a.saved_vx && j >= c0 && j < c1) a.saved_vx[(size_t)b * nVx + k] = v;
        }
    }
    BD_BARRIER();
    SOL_STAMP(2);

    // ---- phase 3: semi-Lagrangian advection (B -> A) of the band's faces, hard-BC face mask fused ----
    // Departure points outside the halo (|u| dt / dx >= BD_HA: an untrained network's first corrections can do that) take a slow path with the
    // SAME result: the diffused value of a far corner is recomputed from the step's input in global memory (phase 2's expressions).
    const float alpha_s = a.adt / a.re[b];
    const float* gvy_in = a.vy_in + (size_t)b * nVy;
    const float* gvx_in = a.vx_in + (size_t)b * nVx;
    const float* gbcv = a.bcv + (size_t)b * a.bc_stride;
    const float* gbcm = a.bcm + (size_t)b * a.bc_stride;
    auto far_y = [&](int j, int i) -> float {
        if (j >= dy0 && j < dy1) return L.Bvy[j * X + i];
        const float c = gvy_in[j * X + i], up = gvy_in[min(j + 1, Y) * X + i], dn = gvy_in[max(j - 1, 0) * X + i];
        const float rt = gvy_in[j * X + min(i + 1, X - 1)], lf = gvy_in[j * X + max(i - 1, 0)];
        return dif_y(c, up, dn, rt, lf, alpha_s, gbcm[j * X + i], gbcv[j * X + i]);
    };
    auto far_x = [&](int j, int i) -> float {
        if (j >= dx0 && j < dx1) return L.Bvx[j * XP + i];
        const float c = gvx_in[j * XP + i];
        const float lap = gvx_in[min(j + 1, Y - 1) * XP + i] + gvx_in[max(j - 1, 0) * XP + i] +
                          gvx_in[j * XP + min(i + 1, X)] + gvx_in[j * XP + max(i - 1, 0)] - 4.f * c;
        return dif_x(c, lap, alpha_s);
    };
    bool far = false;
    #pragma unroll 4
    for (int k = c0 * X + tid; k < ay1 * X; k += nthr) {
        const int j = k >> lx, i = k & (X - 1);
        const float uy = L.Bvy[k];
        const int ja = max(j - 1, 0), jb = min(j, Y - 1);
        const float ux = 0.25f * (L.Bvx[ja * XP + i] + L.Bvx[ja * XP + i + 1] + L.Bvx[jb * XP + i] + L.Bvx[jb * XP + i + 1]);
        const Bil s = bil_clamp(Y + 1, X, j, -uy * a.dtdx, i, -ux * a.dtdx);
        L.Avy[k] = bil_eval(L.Bvy, X, s) * mask_y(L.act, Y, X, j, i);
        far = far || !(s.j0 >= dy0 && s.j1 < dy1);
    }
    #pragma unroll 4
    for (int k = c0 * XP + tid; k < c1 * XP; k += nthr) {
        const int j = (int)(((float)k + 0.5f) * invXP), i = k - j * XP;
        const float ux = L.Bvx[k];
        const int ia = max(i - 1, 0), ib = min(i, X - 1);
        const float uy = 0.25f * (L.Bvy[j * X + ia] + L.Bvy[j * X + ib] + L.Bvy[(j + 1) * X + ia] + L.Bvy[(j + 1) * X + ib]);
        const Bil s = bil_clamp(Y, XP, j, -uy * a.dtdx, i, -ux * a.dtdx);
        L.Avx[k] = bil_eval(L.Bvx, XP, s) * mask_x(L.act, Y, X, j, i);
        far = far || !(s.j0 >= dx0 && s.j1 < dx1);
    }
    if (__builtin_amdgcn_ballot_w64(far) != 0ull) {     // (wave uniform) redo this wave's faces whose departure point left the halo
        for (int k = c0 * X + tid; k < ay1 * X; k += nthr) {
            const int j = k >> lx, i = k & (X - 1);
            const float uy = L.Bvy[k];
            const int ja = max(j - 1, 0), jb = min(j, Y - 1);
            const float ux = 0.25f * (L.Bvx[ja * XP + i] + L.Bvx[ja * XP + i + 1] + L.Bvx[jb * XP + i] + L.Bvx[jb * XP + i + 1]);
            const Bil s = bil_clamp(Y + 1, X, j, -uy * a.dtdx, i, -ux * a.dtdx);
            if (!(s.j0 >= dy0 && s.j1 < dy1))
                L.Avy[k] = bil_mix(s, far_y(s.j0, s.i0), far_y(s.j0, s.i1), far_y(s.j1, s.i0), far_y(s.j1, s.i1)) * mask_y(L.act, Y, X, j, i);
        }
        for (int k = c0 * XP + tid; k < c1 * XP; k += nthr) {
            const int j = (int)(((float)k + 0.5f) * invXP), i = k - j * XP;
            const float ux = L.Bvx[k];
            const int ia = max(i - 1, 0), ib = min(i, X - 1);
            const float uy = 0.25f * (L.Bvy[j * X + ia] + L.Bvy[j * X + ib] + L.Bvy[(j + 1) * X + ia] + L.Bvy[(j + 1) * X + ib]);
            const Bil s = bil_clamp(Y, XP, j, -uy * a.dtdx, i, -ux * a.dtdx);
            if (!(s.j0 >= dx0 && s.j1 < dx1))
                L.Avx[k] = bil_mix(s, far_x(s.j0, s.i0), far_x(s.j0, s.i1), far_x(s.j1, s.i0), far_x(s.j1, s.i1)) * mask_x(L.act, Y, X, j, i);
        }
    }
    SOL_STAMP(3);
    BD_STAMP(3);
    BD_BARRIER();
    SOL_STAMP(4);

    // ---- phase 4: divergence of the band (one 4-cell piece per thread); band 0 gathers the other bands' rows and solves ----
    float4 rq;
    {
        const int j = c0 + (tid >> 4), i = (tid & 15) << 2;
        const float4 y1 = *reinterpret_cast<const float4*>(&L.Avy[(j + 1) * X + i]), y0 = *reinterpret_cast<const float4*>(&L.Avy[j * X + i]);
        const float* xr = &L.Avx[j * XP + i];
        const float x0 = xr[0], x1 = xr[1], x2 = xr[2], x3 = xr[3], x4 = xr[4];
        rq = make_float4(-((y1.x - y0.x) + (x1 - x0)), -((y1.y - y0.y) + (x2 - x1)), -((y1.z - y0.z) + (x3 - x2)), -((y1.w - y0.w) + (x4 - x3)));
    }
    float* P = L.Bvy + FD_BUF;          // [128][64]: the received pressure rows c0-1 .. c1-1
    {
        bd_store(rx, ((c0 + (tid >> 4)) * X + ((tid & 15) << 2)) * 4, bd_encode(rq));
        SOL_STAMP(5);
        BD_STAMP(4);
        constexpr int NG = (BD_PROWS * (X / 4) + nthr - 1) / nthr;       // 2 (528 pieces; band 0 has no row -1: 512)
        const int skip = w == 0 ? X / 4 : 0;
        int off[NG]; bool on[NG]; uint4 v[NG];
#pragma unroll
        for (int n = 0; n < NG; ++n) {
            const int e = skip + tid + n * nthr;
            on[n] = e < BD_PROWS * (X / 4);
            off[n] = (BD_DIV_WORDS + w * BD_PROWS * X + 4 * min(e, BD_PROWS * (X / 4) - 1)) * 4;
        }
        bd_recv<NG>(rx, off, on, v);
#pragma unroll
        for (int n = 0; n < NG; ++n) {
            const int e = skip + tid + n * nthr;
            if (on[n]) {
                *reinterpret_cast<float4*>(&P[(c0 - 1 + (e >> 4)) * X + ((e & 15) << 2)]) = bd_decode(v[n]);
                bd_store(rx, off[n], make_uint4(0u, 0u, 0u, 0u));
            }
        }
        SOL_STAMP(6);
        BD_STAMP(6);
        __syncthreads();
    }

    // ---- phase 6: v -= mask * grad p on the band's faces; outputs ----
    {
        float* gvy = a.vy_out + (size_t)b * nVy;
        float* gvx = a.vx_out + (size_t)b * nVx;
        #pragma unroll 2
        for (int q = c0 * (X / 4) + tid; q < fy1 * (X / 4); q += nthr) {
            const int k = q << 2, j = k >> lx, i = k & (X - 1);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j >= 1 && j <= Y - 1) {
                const float4 p1 = *reinterpret_cast<const float4*>(&P[j * X + i]), p0 = *reinterpret_cast<const float4*>(&P[(j - 1) * X + i]);
                g = make_float4(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z, p1.w - p0.w);
            } else if (a.grad_pad == 1) {
                const float4 p = *reinterpret_cast<const float4*>(&P[(j == 0 ? 0 : Y - 1) * X + i]);
                g = j == 0 ? p : make_float4(-p.x, -p.y, -p.z, -p.w);
            }
            const uchar4 m0 = *reinterpret_cast<const uchar4*>(&L.act[max(j - 1, 0) * X + i]), m1 = *reinterpret_cast<const uchar4*>(&L.act[min(j, Y - 1) * X + i]);
            float4 v = reinterpret_cast<const float4*>(L.Avy)[q];
            v.x -= (float)min(m0.x, m1.x) * g.x; v.y -= (float)min(m0.y, m1.y) * g.y;
            v.z -= (float)min(m0.z, m1.z) * g.z; v.w -= (float)min(m0.w, m1.w) * g.w;
            reinterpret_cast<float4*>(L.Avy)[q] = v;
            reinterpret_cast<float4*>(gvy)[q] = v;
        }
        #pragma unroll 4
        for (int k = c0 * XP + tid; k < c1 * XP; k += nthr) {
            const int j = (int)(((float)k + 0.5f) * invXP), i = k - j * XP;
            float g = 0.f;
            if (i >= 1 && i <= X - 1) g = P[j * X + i] - P[j * X + i - 1];
            else if (a.grad_pad == 1) g = (i == 0) ? P[j * X] : -P[j * X + X - 1];
            const float v = L.Avx[k] - mask_x(L.act, Y, X, j, i) * g;
            L.Avx[k] = v;
            gvx[k] = v;
        }
    }
    SOL_STAMP(7);
    if (a.feat) {   // fused to_feature + 1/std scaling of the band's cells
        __syncthreads();
        float4* gf = reinterpret_cast<float4*>(a.feat) + (size_t)b * N;
        const float rech = a.re[b] * a.fs2;
        #pragma unroll 4
        for (int k = c0 * X + tid; k < c1 * X; k += nthr) {
            const int j = k >> lx, i = k & (X - 1);
            gf[k] = make_float4(L.Avy[k] * a.fs0, L.Avx[j * XP + i] * a.fs1, rech, 0.f);
        }
    }
    SOL_STAMP(8);
    BD_STAMP(8);
}

// the simulation's SOLVER workgroup: no stencil work -- its coefficients (100 registers) are in flight while the bands advect --, gathers
// the divergence rows of all bands, runs the direct solve, hands every band its pressure rows
#define BS_STAMP(i) do { if (a.prof && blockIdx.x == 8 * BD_N && threadIdx.x == 0) a.prof[i] = wall_clock64(); } while (0)   /* solver workgroup of simulation 0 */
__device__ __forceinline__ void karman_fwd_solver_wg(const StepArgs& a, float* smem, const int b, unsigned* xw) {
    constexpr int Y = FD_Y, X = FD_X, nthr = 512;
    const int tid = threadIdx.x;
    const Lds L = carve(smem, Y, X, 16);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(xw, 0, BD_WORDS * 4, 0x00020000);
    BS_STAMP(29);
    const float fdp = fd_prefetch(a.fd, a.fd_n);
    FdOps fops;
    float qys[FD_QYS];
    fd_load_qys(a.fd + 16, __builtin_amdgcn_readfirstlane(tid >> 6), qys);
    fd_load_ops(a.fd, fops);
    float* B0 = L.Bvy;                  // the solver's right-hand side, [128][FD_LD]
    float* P = L.Bvy + FD_BUF;          // its result, [128][64]
    constexpr int NG = Y * (X / 4) / nthr;      // 4 pieces per thread
    int off[NG]; bool on[NG]; uint4 v[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) { off[n] = 4 * (tid + n * nthr) * 4; on[n] = true; }
    bd_recv<NG>(rx, off, on, v);
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        const int e = tid + n * nthr, j = e >> 4, i = (e & 15) << 2;
        const float4 f = bd_decode(v[n]);
        float* d = &B0[j * FD_LD + i];
        d[0] = f.x; d[1] = f.y; d[2] = f.z; d[3] = f.w;
        bd_store(rx, off[n], make_uint4(0u, 0u, 0u, 0u));
    }
    BS_STAMP(30);
    const float rdummy[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    fd_solve<true, true>(a.fd, qys, L.Bvy, rdummy, a.prof, fops);
    if (a.iters && tid == 0) a.iters[b] = 0;
    BS_STAMP(31);
    // the pressure rows c0-1 .. c1-1 of every band, one copy per band (band 0 has no row -1)
    constexpr int NS = (BD_N * BD_PROWS * (X / 4) + nthr - 1) / nthr;
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const int e = tid + n * nthr;
        if (e < BD_N * BD_PROWS * (X / 4)) {
            const int h = e / (BD_PROWS * (X / 4)), r = e - h * (BD_PROWS * (X / 4));
            const int j = BD_R * h - 1 + (r >> 4), i = (r & 15) << 2;
            if (j >= 0) bd_store(rx, (BD_DIV_WORDS + (h * BD_PROWS + (r >> 4)) * X + i) * 4, bd_encode(*reinterpret_cast<const float4*>(&P[j * X + i])));
        }
    }
    BS_STAMP(27);
    if (fdp == 1.2345678e-30f && a.iters) a.iters[b] = -2;      // never true: keeps the prefetch loads alive
}

__global__ void __launch_bounds__(512) k_karman_fwd_bands(StepArgs a, DensStep q, unsigned* xch) {
    extern __shared__ __align__(16) float smem[];
    const int u = (int)blockIdx.x, nsol = 8 * (BD_N + 1) * ((a.B + 7) >> 3);
    if (u < nsol) {
        const int b = (u / (8 * (BD_N + 1))) * 8 + (u & 7), w = (u >> 3) % (BD_N + 1);
        if (b >= a.B) return;
        if (w < BD_N) karman_fwd_band_body(a, smem, b, w, xch + (size_t)b * BD_WORDS);
        else karman_fwd_solver_wg(a, smem, b, xch + (size_t)b * BD_WORDS);
    } else density_step_body(q, u - nsol);
}

// ------------------------------------------------------------------------------------
// backward (adjoint w.r.t. the input velocity)
// ------------------------------------------------------------------------------------
// NT != 0: the body runs on the first NT threads of a larger workgroup whose other waves have ended (k_karman_bwd_bww_small)
template <int CPT, int SOLVER, int NT = 0>
__device__ __forceinline__ void karman_bwd_body(const StepArgs& a, float* smem) {
    constexpr int MAXT = CPT + 1;   // face targets per thread: (Y+1)*X / (Y*X/CPT) <= CPT+1 for Y >= CPT
    const int b = blockIdx.x, tid = threadIdx.x, nthr = NT ? NT : (int)blockDim.x;
    const int Y = a.Y, X = a.X, N = Y * X, nVy = (Y + 1) * X, nVx = Y * (X + 1), XP = X + 1;
    const int lx = __ffs(X) - 1;                 // X is a power of two: k / X == k >> lx
    const float invXP = 1.f / (float)XP;         // k / XP == (int)((k + 0.5) * invXP), exact for k < 2^22
    const Lds L = carve(smem, Y, X, CPT);
    const bool dirichlet = a.grad_pad == 1;

    SOL_STAMP(0);
    float fdp = 0.f;
    if constexpr (SOLVER == 2) {
        if (Y == FD_Y) fdp = fd_prefetch(a.fd, a.fd_n);
        else fd_small_stage<NT>(a.fd, Y, X, L.fdx);
    }
    // ---- 1: load incoming gradient (+ feature gradient): all global loads in flight first --------
    {
        const float* gy = a.g_vy_out + (size_t)b * nVy;
        const float* gx = a.g_vx_out + (size_t)b * nVx;
        const float* df = a.dfeat ? a.dfeat + (size_t)b * N * 2 : nullptr;
        // v_y gradient, its feature gradient (channel 0 of two float4 = four cells) and the cell mask in 16-byte pieces;
        // the v_x part keeps the per-face form (rows of X+1 faces: its cell index is not the face index)
        constexpr int NV = (CPT + 1 + 3) / 4;
        const int nQy = nVy >> 2;
        const float4* gy4 = reinterpret_cast<const float4*>(gy);
        const float4* df4 = reinterpret_cast<const float4*>(df);
        const float4* gact = reinterpret_cast<const float4*>(a.active);
        float4 ty[NV], ta[CPT / 4], f0[NV], f1[NV];
        float tx[MAXT], fxv[MAXT];
        // EVERY request of the phase goes out before the first value is used (clamped indices; without a feature gradient the
        // feature requests read the velocity gradient and get weight 0).  With the feature loads inside `if (df)` next to their
        // use, the compiler produced one memory round trip per loop iteration: 3.4 us warm, but 15 us in the training pipeline
        // where every operand comes from HBM (tools/step_phases.py 6 64 train).
        const float4* dq4 = df ? df4 : gy4;
        const float* dq = df ? df : gy;
        const float w0 = df ? a.fs0 : 0.f, w1 = df ? a.fs1 : 0.f;
        if (a.feat_tr && df) {      // feature gradient in the transposed cell order: cell (j, i) at i * Y + j (uniform branch around the whole request group)
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int q = min(tid + n * nthr, nQy - 1);
                const int k0 = 4 * min(q, (N >> 2) - 1), j = k0 >> lx, i0 = k0 & (X - 1);       // four cells of one row (X % 4 == 0)
                const float* p = dq + 2 * ((size_t)i0 * Y + j);
                ty[n] = gy4[q];
                f0[n].x = p[0]; f0[n].z = p[2 * Y]; f1[n].x = p[4 * Y]; f1[n].z = p[6 * Y];
                f0[n].y = f0[n].w = f1[n].y = f1[n].w = 0.f;
            }
#pragma unroll
            for (int n = 0; n < MAXT; ++n) {
                const int kx = min(tid + n * nthr, nVx - 1);
                const int j = (int)(((float)kx + 0.5f) * invXP), i = kx - j * XP;
                tx[n] = gx[kx];
                fxv[n] = dq[2 * (min(i, X - 1) * Y + j) + 1];
            }
        } else {
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int q = min(tid + n * nthr, nQy - 1);
                const int qd = df ? min(q, (N >> 2) - 1) : 0;
                ty[n] = gy4[q];
                f0[n] = dq4[2 * qd];
                f1[n] = dq4[2 * qd + 1];
            }
#pragma unroll
            for (int n = 0; n < MAXT; ++n) {
                const int kx = min(tid + n * nthr, nVx - 1);
                const int j = (int)(((float)kx + 0.5f) * invXP), i = kx - j * XP;
                tx[n] = gx[kx];
                fxv[n] = dq[df ? 2 * (j * X + min(i, X - 1)) + 1 : 0];
            }
        }
#pragma unroll
        for (int n = 0; n < CPT / 4; ++n) ta[n] = gact[min(tid + n * nthr, (N >> 2) - 1)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = min(tid + n * nthr, nQy - 1);
            const float w = (q << 2) < N ? w0 : 0.f;                                                  // rows j < Y
            ty[n].x += w * f0[n].x; ty[n].y += w * f0[n].z; ty[n].z += w * f1[n].x; ty[n].w += w * f1[n].z;
        }
#pragma unroll
        for (int n = 0; n < MAXT; ++n) {
            const int kx = min(tid + n * nthr, nVx - 1);
            const int j = (int)(((float)kx + 0.5f) * invXP), i = kx - j * XP;
            tx[n] += i < X ? w1 * fxv[n] : 0.f;
        }
#pragma unroll
        for (int n = 0; n < NV; ++n) { const int q = tid + n * nthr; if (q < nQy) reinterpret_cast<float4*>(L.Avy)[q] = ty[n]; }
#pragma unroll
        for (int n = 0; n < MAXT; ++n) { const int k = tid + n * nthr; if (k < nVx) L.Avx[k] = tx[n]; }
#pragma unroll
        for (int n = 0; n < CPT / 4; ++n) {
            const int q = tid + n * nthr;
            if (q < (N >> 2)) reinterpret_cast<uchar4*>(L.act)[q] = make_uchar4(ta[n].x != 0.f, ta[n].y != 0.f, ta[n].z != 0.f, ta[n].w != 0.f);
        }
    }
    __syncthreads();
    SOL_STAMP(1);

    // ---- 2: projection adjoint:  M z = G^T (m * g) -----------------------------------
    float qys[FD_QYS];
#pragma unroll
    for (int j = 0; j < FD_QYS; ++j) qys[j] = 0.f;
    if constexpr (SOLVER == 2 && CPT == 16) { if (Y == FD_Y) fd_load_qys(a.fd + 16, __builtin_amdgcn_readfirstlane(tid >> 6), qys); }
    const Own o = ownership<CPT>(Y, X);
    float dg[CPT], ac[CPT], r[CPT], z[CPT];
    cell_coeffs<CPT>(o, L.act, Y, X, dg, ac);
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        r[k] = 0.f;
        if (o.owner) {
            const int j = o.j0 + k, i = o.i;
            float s = 0.f;
            if (j >= 1 || dirichlet) s += mask_y(L.act, Y, X, j, i) * L.Avy[j * X + i];
            if (j + 1 <= Y - 1 || dirichlet) s -= mask_y(L.act, Y, X, j + 1, i) * L.Avy[(j + 1) * X + i];
            if (i >= 1 || dirichlet) s += mask_x(L.act, Y, X, j, i) * L.Avx[j * XP + i];
            if (i + 1 <= X - 1 || dirichlet) s -= mask_x(L.act, Y, X, j, i + 1) * L.Avx[j * XP + i + 1];
            r[k] = s;
        }
    }
    int it = 0;
    float* Pfd = nullptr;
    SOL_STAMP(2);
    if constexpr (SOLVER == 2) {
        if constexpr (CPT == 16) Pfd = Y == FD_Y ? fd_solve(a.fd, qys, L.Bvy, r, a.prof) : fd_solve_small<CPT, NT>(a.fd, Y, X, o, L.fdx, r);
        else Pfd = fd_solve_small<CPT, NT>(a.fd, Y, X, o, L.fdx, r);
    } else if constexpr (SOLVER == 1) {
        it = X == 64 ? pcg_solve<CPT, true, 32>(o, Y, X, L.act, dg, ac, r, z, L.E, L.red, L.cs, a.cinv, a.rtol2, a.atol2, a.max_iter)
                     : pcg_solve<CPT, false, 8>(o, Y, X, L.act, dg, ac, r, z, L.E, L.red, L.cs, a.cinv, a.rtol2, a.atol2, a.max_iter);
    } else {
        it = X == 64 ? cg_solve<CPT, true>(o, X, dg, ac, r, z, L.E, L.red, a.rtol2, a.atol2, a.max_iter)
                     : cg_solve<CPT, false>(o, X, dg, ac, r, z, L.E, L.red, a.rtol2, a.atol2, a.max_iter);
    }
    if (a.iters && tid == 0) a.iters[b] = it;
    SOL_STAMP(3);

    // ---- 3: g_adv = m * (g + D^T z), kept in registers ---------------------------------
    float* Z = L.Bvy;
    if constexpr (SOLVER == 2) Z = Pfd;
    else {
        if (o.owner) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) Z[(o.j0 + k) * X + o.i] = z[k];
        }
        __syncthreads();
    }
    float gy[MAXT], gx[MAXT];
#pragma unroll
    for (int n = 0; n < MAXT; ++n) {
        const int k = tid + n * nthr;
        gy[n] = 0.f;
        gx[n] = 0.f;
        if (a.dbg & 16) { gy[n] = 1.f; gx[n] = 1.f; continue; }
        if (k < nVy) {
            const int j = k >> lx, i = k & (X - 1);
            const float zp = j >= 1 ? Z[(j - 1) * X + i] : 0.f;
            const float zc = j <= Y - 1 ? Z[j * X + i] : 0.f;
            gy[n] = mask_y(L.act, Y, X, j, i) * (L.Avy[k] + (zp - zc));
        }
        if (k < nVx) {
            const int j = (int)(((float)k + 0.5f) * invXP), i = k - j * XP;
            const float zp = i >= 1 ? Z[j * X + i - 1] : 0.f;
            const float zc = i <= X - 1 ? Z[j * X + i] : 0.f;
            gx[n] = mask_x(L.act, Y, X, j, i) * (L.Avx[k] + (zp - zc));
        }
    }
    __syncthreads();
    SOL_STAMP(4);

    // ---- 4: stage the saved (post-diffusion) velocity, clear the accumulators ---------
    // The scatter-add runs in int32 FIXED POINT: ds_add_f32 is ~37x slower than ds_add_u32 on gfx950
    // (0.8 vs 30 lane-ops/ns/CU, tools/ubench/lds_atomic.hip; it was 159 of the 237 us of this kernel),
    // and integer accumulation is order independent, i.e. the backward pass is bit-reproducible.
    // scale = 2^31 / (16 * bound), bound = max|g| * max(1, 2*dt/dx*max|v|) >= any single contribution.
    int* Iy = reinterpret_cast<int*>(L.Avy);
    int* Ix = reinterpret_cast<int*>(L.Avx);
    float smax = 0.f, gmax = 0.f;
    {
        constexpr int NV = (CPT + 1 + 3) / 4;       // float4 items of a (Y+1) x X field per thread (flat 16-byte copies)
        const int nQy = nVy >> 2, nQx = nVx >> 2;
        const float4* sy = reinterpret_cast<const float4*>(a.saved_vy + (size_t)b * nVy);
        const float4* sx = reinterpret_cast<const float4*>(a.saved_vx + (size_t)b * nVx);
        float4 ty[NV], tx[NV];
#pragma unroll
        for (int n = 0; n < NV; ++n) { const int q = tid + n * nthr; ty[n] = sy[min(q, nQy - 1)]; tx[n] = sx[min(q, nQx - 1)]; }
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = tid + n * nthr;
            if (q < nQy) {
                reinterpret_cast<float4*>(L.Bvy)[q] = ty[n]; reinterpret_cast<int4*>(Iy)[q] = make_int4(0, 0, 0, 0);
                smax = fmaxf(smax, fmaxf(fmaxf(fabsf(ty[n].x), fabsf(ty[n].y)), fmaxf(fabsf(ty[n].z), fabsf(ty[n].w))));
            }
            if (q < nQx) {
                reinterpret_cast<float4*>(L.Bvx)[q] = tx[n]; reinterpret_cast<int4*>(Ix)[q] = make_int4(0, 0, 0, 0);
                smax = fmaxf(smax, fmaxf(fmaxf(fabsf(tx[n].x), fabsf(tx[n].y)), fmaxf(fabsf(tx[n].z), fabsf(tx[n].w))));
            }
        }
#pragma unroll
        for (int n = 0; n < MAXT; ++n) gmax = fmaxf(gmax, fmaxf(fabsf(gy[n]), fabsf(gx[n])));
    }
    block_max2<NT>(smax, gmax, L.red);          // barrier: S staged, accumulators cleared
    const float bound = 16.f * gmax * fmaxf(1.f, 2.f * a.dtdx * smax);
    const float qs = bound > 0.f ? 2147483648.f / bound : 0.f;      // fixed-point scale
    const float qi = bound > 0.f ? bound / 2147483648.f : 0.f;
    auto fx = [&](float v) { return __float2int_rn(v * qs); };

    SOL_STAMP(5);
    // ---- 5: advection adjoint (integer scatter-add into LDS) ----------------------------
    if (!(a.dbg & 32))
#pragma unroll
    for (int n = 0; n < MAXT; ++n) {
        const int k = tid + n * nthr;
        if (k < nVy && gy[n] != 0.f) {
            const int j = k >> lx, i = k & (X - 1);
            const float g = gy[n];
            const float uy = L.Bvy[k];
            const int ja = max(j - 1, 0), jb = min(j, Y - 1);
            const float ux = 0.25f * (L.Bvx[ja * XP + i] + L.Bvx[ja * XP + i + 1] + L.Bvx[jb * XP + i] + L.Bvx[jb * XP + i + 1]);
            const Bil s = bil_clamp(Y + 1, X, j, -uy * a.dtdx, i, -ux * a.dtdx);
            const float f00 = L.Bvy[s.j0 * X + s.i0], f01 = L.Bvy[s.j0 * X + s.i1];
            const float f10 = L.Bvy[s.j1 * X + s.i0], f11 = L.Bvy[s.j1 * X + s.i1];
            atomicAdd(&Iy[s.j0 * X + s.i0], fx((1.f - s.wy) * (1.f - s.wx) * g));
            atomicAdd(&Iy[s.j0 * X + s.i1], fx((1.f - s.wy) * s.wx * g));
            atomicAdd(&Iy[s.j1 * X + s.i0], fx(s.wy * (1.f - s.wx) * g));
            atomicAdd(&Iy[s.j1 * X + s.i1], fx(s.wy * s.wx * g));
            const float ddy = (1.f - s.wx) * (f10 - f00) + s.wx * (f11 - f01);
            const float ddx = (1.f - s.wy) * (f01 - f00) + s.wy * (f11 - f10);
            const int guy = fx(-a.dtdx * g * ddy), gux = fx(-0.25f * a.dtdx * g * ddx);
            atomicAdd(&Iy[k], guy);
            atomicAdd(&Ix[ja * XP + i], gux);
            atomicAdd(&Ix[ja * XP + i + 1], gux);
            atomicAdd(&Ix[jb * XP + i], gux);
            atomicAdd(&Ix[jb * XP + i + 1], gux);
        }
        if (k < nVx && gx[n] != 0.f) {
            const int j = (int)(((float)k + 0.5f) * invXP), i = k - j * XP;
            const float g = gx[n];
            const float ux = L.Bvx[k];
            const int ia = max(i - 1, 0), ib = min(i, X - 1);
            const float uy = 0.25f * (L.Bvy[j * X + ia] + L.Bvy[j * X + ib] + L.Bvy[(j + 1) * X + ia] + L.Bvy[(j + 1) * X + ib]);
            const Bil s = bil_clamp(Y, XP, j, -uy * a.dtdx, i, -ux * a.dtdx);
            const float f00 = L.Bvx[s.j0 * XP + s.i0], f01 = L.Bvx[s.j0 * XP + s.i1];
            const float f10 = L.Bvx[s.j1 * XP + s.i0], f11 = L.Bvx[s.j1 * XP + s.i1];
            atomicAdd(&Ix[s.j0 * XP + s.i0], fx((1.f - s.wy) * (1.f - s.wx) * g));
            atomicAdd(&Ix[s.j0 * XP + s.i1], fx((1.f - s.wy) * s.wx * g));
            atomicAdd(&Ix[s.j1 * XP + s.i0], fx(s.wy * (1.f - s.wx) * g));
            atomicAdd(&Ix[s.j1 * XP + s.i1], fx(s.wy * s.wx * g));
            const float ddy = (1.f - s.wx) * (f10 - f00) + s.wx * (f11 - f01);
            const float ddx = (1.f - s.wy) * (f01 - f00) + s.wy * (f11 - f10);
            const int gux = fx(-a.dtdx * g * ddx), guy = fx(-0.25f * a.dtdx * g * ddy);
            atomicAdd(&Ix[k], gux);
            atomicAdd(&Iy[j * X + ia], guy);
            atomicAdd(&Iy[j * X + ib], guy);
            atomicAdd(&Iy[(j + 1) * X + ia], guy);
            atomicAdd(&Iy[(j + 1) * X + ib], guy);
        }
    }
    __syncthreads();
    SOL_STAMP(6);
    {   // back to float (in place, element-wise) fused with the BC adjoint; |acc| > 2^30 would mean the 16x headroom was nearly used up
        bool risky = false;
        const float* bcm = a.bcm + (size_t)b * a.bc_stride;
        const float4* bcm4 = reinterpret_cast<const float4*>(bcm);
        #pragma unroll 2
        for (int q = tid; q < (nVy >> 2); q += nthr) {
            const int4 v = reinterpret_cast<const int4*>(Iy)[q];
            const float4 m = bcm4[q];
            risky |= max(max(abs(v.x), abs(v.y)), max(abs(v.z), abs(v.w))) > (1 << 30);
            reinterpret_cast<float4*>(L.Avy)[q] = make_float4((float)v.x * qi * (1.f - m.x), (float)v.y * qi * (1.f - m.y),
                                                              (float)v.z * qi * (1.f - m.z), (float)v.w * qi * (1.f - m.w));
        }
        #pragma unroll 2
        for (int q = tid; q < (nVx >> 2); q += nthr) {
            const int4 v = reinterpret_cast<const int4*>(Ix)[q];
            risky |= max(max(abs(v.x), abs(v.y)), max(abs(v.z), abs(v.w))) > (1 << 30);
            reinterpret_cast<float4*>(L.Avx)[q] = make_float4((float)v.x * qi, (float)v.y * qi, (float)v.z * qi, (float)v.w * qi);
        }
        if (risky && a.iters) a.iters[b] = -1;      // reported by the host wrappers as an error
    }
    __syncthreads();

    SOL_STAMP(7);
    // ---- 6: diffusion adjoint (the replicate Laplacian is symmetric) --
    if (a.dbg & 64) return;
    {
        const float alpha = a.adt / a.re[b];
        float* oy = a.g_vy_in + (size_t)b * nVy;
        float* ox = a.g_vx_in + (size_t)b * nVx;
        #pragma unroll 2
        for (int q = tid; q < (nVy >> 2); q += nthr) {      // four faces of one row; summation order of the scalar form
            const int k = q << 2, j = k >> lx, i = k & (X - 1);
            const float4 c = reinterpret_cast<const float4*>(L.Avy)[q];
            const float4 up = *reinterpret_cast<const float4*>(&L.Avy[min(j + 1, Y) * X + i]);
            const float4 dn = *reinterpret_cast<const float4*>(&L.Avy[max(j - 1, 0) * X + i]);
            const float rt = L.Avy[j * X + min(i + 4, X - 1)], lf = L.Avy[j * X + max(i - 1, 0)];
            reinterpret_cast<float4*>(oy)[q] = make_float4(c.x + alpha * (up.x + dn.x + c.y + lf - 4.f * c.x), c.y + alpha * (up.y + dn.y + c.z + c.x - 4.f * c.y),
                                                           c.z + alpha * (up.z + dn.z + c.w + c.y - 4.f * c.z), c.w + alpha * (up.w + dn.w + rt + c.z - 4.f * c.w));
        }
        #pragma unroll 4
        for (int k = tid; k < nVx; k += nthr) {
            const int j = (int)(((float)k + 0.5f) * invXP), i = k - j * XP;
            const float c = L.Avx[k];
            const float lap = L.Avx[min(j + 1, Y - 1) * XP + i] + L.Avx[max(j - 1, 0) * XP + i] +
                              L.Avx[j * XP + min(i + 1, X)] + L.Avx[j * XP + max(i - 1, 0)] - 4.f * c;
            ox[k] = c + alpha * lap;
        }
    }
    SOL_STAMP(8);
    if (fdp == 1.2345678e-30f && a.iters) a.iters[b] = -2;      // never true: keeps the prefetch loads alive
}

template <int CPT, int SOLVER>
__global__ void __launch_bounds__(CPT == 16 ? 512 : 1024) k_karman_bwd(StepArgs a) {
    extern __shared__ __align__(16) float smem[];
    karman_bwd_body<CPT, SOLVER>(a, smem);
}

// Horizontal fusion: the solver adjoint of an unrolled step occupies ONE CU per simulation for its whole duration
// (6 of 256 at C3) and nothing else of the reverse sweep can run meanwhile (it waits for this kernel's output).  The
// weight gradients of the step, whose dz tensors are complete by then, have no consumer until the end of the sweep:
// their workgroups ride in the SAME launch (blocks >= B), sized to last about as long as the adjoint (32 rows each).
struct BwPack {
    BwArgs a[12];
    int n, wg_per;          // n gradient jobs (layers) of wg_per workgroups each
};
__global__ void __launch_bounds__(512) k_karman_bwd_bww(StepArgs a, BwPack bw) {
    extern __shared__ __align__(16) float smem[];
    // timing experiments (option dbg_skip, results invalid; tools/adjoint_split_experiment.py): 4096 = the gradient workgroups end at once (the
    // adjoint ALONE inside the pipeline, cold operands), 8192 = the adjoint workgroups end at once (the gradient half alone), 2048 = the
    // gradient workgroups start ~3.4 us late (does the adjoint's load phase recover when it does not queue behind their 24 MB prologue burst?)
    if ((int)blockIdx.x < a.B) {
        if (a.dbg & 8192) return;
        karman_bwd_body<16, 2>(a, smem);
    } else {
        if (a.dbg & 4096) return;
        if (a.dbg & 2048) __builtin_amdgcn_s_sleep(127);
        // XCD-aware job order (workgroup u runs on XCD u % 8): the 32-row blocks of image rows 96x .. 96x+95 -- what XCD x's
        // convolution workgroups wrote (xcd_tile) -- are handed to the gradient workgroups of XCD x, layer after layer
        int idx = (int)blockIdx.x - a.B, job = idx / bw.wg_per, sub = idx % bw.wg_per;
#ifndef SOL_NO_XCD_REMAP
        if ((bw.wg_per & 7) == 0) {
            const int u = (int)blockIdx.x, x = u & 7, per = bw.wg_per >> 3;
            const int s = (u - (a.B + ((x - a.B) & 7))) >> 3;          // rank of this workgroup among the gradient workgroups of XCD x
            job = s / per;
            sub = x * per + s % per;
        }
#endif
        const bool stamp = a.prof && threadIdx.x == 0 && (idx == 0 || (int)blockIdx.x == (int)gridDim.x - 1);   // step_prof only
        if (stamp) a.prof[idx == 0 ? 16 : 18] = wall_clock64();
        sbk::bww_sb_body<2>(bw.a[job], sub, reinterpret_cast<unsigned char*>(smem));
        if (stamp) a.prof[idx == 0 ? 17 : 19] = wall_clock64();
    }
}

// The same fusion for the 64 x 32 grid (the reference's own training recipe): its adjoint runs on 256 threads (eight-cell strips,
// LDS-resident direct solver), the weight-gradient body on 512.  The launch has 512-thread workgroups; in a solver workgroup waves
// 4..7 end at once (ended waves do not take part in s_barrier) and the adjoint body runs on the first 256 threads
// (karman_bwd_body<8, 2, 256>: every stride it derives from the workgroup size is the template argument).
// The passive density has no forward launch to ride with at this size (the forward kernel is the 256-thread form): ONE of its msteps
// advections rides in each launch of the REVERSE sweep as q.B more workgroups (all saved velocities exist by then; it was a chain of
// msteps dependent advections in one launch behind the forward unroll: 162 us of the 64x32 recipe's step).
__global__ void __launch_bounds__(512) k_karman_bwd_bww_small(StepArgs a, BwPack bw, DensStep q) {
    extern __shared__ __align__(16) float smem[];
    const int ng = bw.n * bw.wg_per;
    if ((int)blockIdx.x < a.B) {
        if (threadIdx.x >= 256) return;
        karman_bwd_body<8, 2, 256>(a, smem);
    } else if ((int)blockIdx.x < a.B + ng) {
        const int idx = (int)blockIdx.x - a.B;
        sbk::bww_sb_body<2>(bw.a[idx / bw.wg_per], idx % bw.wg_per, reinterpret_cast<unsigned char*>(smem));
    } else density_step_body(q, (int)blockIdx.x - a.B - ng);
}

// ------------------------------------------------------------------------------------
// Passive tracer chain (training path)
// ------------------------------------------------------------------------------------
// The density never feeds the velocity (buoyancy_factor = 0, karman_train.py:363): its advection only needs the
// post-diffusion velocities that the forward unroll keeps anyway (saved_vy / saved_vx).  In the training path the
// msteps density advections therefore leave the critical path: ONE launch, one workgroup per simulation, density
// ping-pong in LDS, on a side stream behind the forward unroll (the fused step kernels are launched with d_out =
// NULL).  Same arithmetic as phase 3 of k_karman_fwd.
struct DensArgs {
    int B, Y, X, ms, inflow_before;
    float dtdx, dt;
    const float *d0, *svy, *svx, *inflow;
    long st_vy, st_vx, st_d;        // strides between consecutive steps (floats)
    float* d_steps;                 // [ms][B][N] or NULL: every intermediate density
    float* d_final;                 // [B][N] or NULL
};

__global__ void __launch_bounds__(1024) k_density_chain(DensArgs a) {
    constexpr int MC = 8;            // cells per thread: N <= 8192 with 1024 threads (host checks)
    extern __shared__ __align__(16) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int Y = a.Y, X = a.X, N = Y * X, XP = X + 1;
    const int lx = __ffs(X) - 1;
    float* D0 = smem;
    float* D1 = smem + N;
    for (int k = tid; k < N; k += nthr) D0[k] = a.d0[(size_t)b * N + k];
    float infl[MC], uy[MC], ux[MC];
#pragma unroll
    for (int n = 0; n < MC; ++n) { const int k = tid + n * nthr; infl[n] = k < N ? a.inflow[k] : 0.f; }
    // cell-centre velocity of step s: global loads, software pipelined one step ahead of the LDS work
    auto load_u = [&](int s, float (&vy)[MC], float (&vx)[MC]) {
        const float* sy = a.svy + (size_t)s * a.st_vy + (size_t)b * (Y + 1) * X;
        const float* sx = a.svx + (size_t)s * a.st_vx + (size_t)b * Y * XP;
#pragma unroll
        for (int n = 0; n < MC; ++n) {
            const int k = min(tid + n * nthr, N - 1), j = k >> lx, i = k & (X - 1);
            vy[n] = 0.5f * (sy[k] + sy[k + X]);
            vx[n] = 0.5f * (sx[j * XP + i] + sx[j * XP + i + 1]);
        }
    };
    load_u(0, uy, ux);
    __syncthreads();
    for (int s = 0; s < a.ms; ++s) {
        float uyn[MC], uxn[MC];
        if (s + 1 < a.ms) load_u(s + 1, uyn, uxn);
        const float* src = (s & 1) ? D1 : D0;
        float* dst = (s & 1) ? D0 : D1;
#pragma unroll
        for (int n = 0; n < MC; ++n) {
            const int k = tid + n * nthr;
            if (k >= N) continue;
            const int j = k >> lx, i = k & (X - 1);
            const float oy = -uy[n] * a.dtdx, ox = -ux[n] * a.dtdx;
            const float fy = floorf(oy), fx = floorf(ox);
            const float wy = oy - fy, wx = ox - fx;
            const int j0 = j + (int)fy, i0 = i + (int)fx;
            float f[2][2];
#pragma unroll
            for (int dj = 0; dj < 2; ++dj)
#pragma unroll
                for (int di = 0; di < 2; ++di) {
                    const int jj = j0 + dj, ii = i0 + di;
                    float v = 0.f;   // extrapolation 'constant': one ring of zero ghost cells
                    if (jj >= 0 && jj < Y && ii >= 0 && ii < X) {
                        v = src[jj * X + ii];
                        if (a.inflow_before) v += a.inflow[jj * X + ii];
                    }
                    f[dj][di] = v;
                }
            float v = (1.f - wy) * ((1.f - wx) * f[0][0] + wx * f[0][1]) + wy * ((1.f - wx) * f[1][0] + wx * f[1][1]);
            if (!a.inflow_before) v += infl[n] * a.dt;
            dst[k] = v;
            if (a.d_steps) a.d_steps[(size_t)s * a.st_d + (size_t)b * N + k] = v;
            if (a.d_final && s == a.ms - 1) a.d_final[(size_t)b * N + k] = v;
        }
        if (s + 1 < a.ms) {
#pragma unroll
            for (int n = 0; n < MC; ++n) { uy[n] = uyn[n]; ux[n] = uxn[n]; }
        }
        __syncthreads();
    }
}

// strip height: 16 cells per thread when the grid allows it (fewer waves -> less per-wave
// reduction overhead in the issue-bound CG loop), 8 otherwise.  SOL_CPT=8|16 overrides.
int pick_cpt(const sol_karman_cfg* c) {
    int cpt = (c->Y % 16 == 0 && c->X >= 16) ? 16 : 8;
    if (c->direct && fd_small_floats(c->Y, c->X) > 0 && (c->Y / 8) * c->X <= 1024) cpt = 8;
    const int v = sol_opt().cpt;
    if (v == 8 || (v == 16 && c->Y % 16 == 0 && c->X >= 16)) cpt = v;
    return cpt;
}

bool precond_ok(int Y, int X) {
    if (Y % 8 || X % 8) return false;
    const int nc = (Y / 8) * (X / 8);
    const int cpt = (Y % 16 == 0 && X >= 16) ? 16 : 8;     // default strip height (SOL_CPT may lower it)
    // kernels are instantiated for nc/4 == 32 columns per thread at X == 64 (128x64) and 8 at X == 32 (64x32)
    if (cpt != 16) return false;
    return (X == 64 && nc / 4 == 32) || (X == 32 && nc / 4 == 8);
}

int check_cfg(const sol_karman_cfg* c) {
    SOL_REQUIRE(c != nullptr, "cfg is NULL");
    SOL_REQUIRE(c->B >= 1, "B must be >= 1 (got %d)", c->B);
    SOL_REQUIRE(c->Y >= 8 && c->Y % 8 == 0, "Y must be a positive multiple of 8 (got %d)", c->Y);
    SOL_REQUIRE(c->X == 8 || c->X == 16 || c->X == 32 || c->X == 64, "X must be 8, 16, 32 or 64 (got %d)", c->X);
    const int cpt = pick_cpt(c);
    SOL_REQUIRE((c->Y / cpt) * c->X <= (cpt == 16 ? 512 : 1024), "grid %dx%d exceeds one workgroup", c->Y, c->X);
    SOL_REQUIRE(lds_bytes(c->Y, c->X, cpt) <= 160 * 1024, "grid %dx%d does not fit the 160 KiB LDS", c->Y, c->X);
    SOL_REQUIRE(c->dx > 0.f && c->cg_max_iter >= 0, "dx must be > 0 and cg_max_iter >= 0");
    if (c->direct) {
        SOL_REQUIRE((c->Y == FD_Y && c->X == FD_X && cpt == 16) || fd_small_floats(c->Y, c->X) > 0,
                    "the direct pressure solver is built for 128x64 and for grids of at most 2048 cells with Y %% 16 == 0, X >= 16 (got %dx%d)", c->Y, c->X);
        SOL_REQUIRE(c->direct_n >= 16 + c->Y * c->Y + c->X * c->X + c->X * c->Y + 64 * 64 + 64 + c->X * FD_WIN,
                    "direct_n = %d is too small for a direct-solver blob", c->direct_n);
        SOL_REQUIRE(c->direct_n <= 8 * 512 * 32, "direct_n = %d: the blob is larger than the kernels' prefetch covers (131072 words)", c->direct_n);
    }
    if (c->coarse_inv && !c->direct) {       // (the direct solver takes precedence: the preconditioner is then unused)
        SOL_REQUIRE(precond_ok(c->Y, c->X) && cpt == 16, "the two-level CG preconditioner is not available for a %dx%d grid", c->Y, c->X);
        SOL_REQUIRE(c->coarse_n == (c->Y / 8) * (c->X / 8), "coarse_n must be (Y/8)*(X/8) = %d (got %d)", (c->Y / 8) * (c->X / 8), c->coarse_n);
    }
    return SOL_OK;
}

thread_local int g_feat_transposed = 0;        // see sol_karman_feat_transposed

void fill_common(StepArgs& a, const sol_karman_cfg* c) {
    a.B = c->B; a.Y = c->Y; a.X = c->X;
    a.dtdx = c->dt / c->dx;
    a.dt = c->dt;
    a.adt = c->dt * c->res * c->res;
    a.rtol2 = c->cg_rtol * c->cg_rtol;
    a.atol2 = c->cg_atol * c->cg_atol;
    a.max_iter = c->cg_max_iter;
    a.grad_pad = c->grad_pad;
    a.inflow_before = c->inflow_before;
    a.cinv = c->coarse_inv;
    a.fd = c->direct;
    a.fd_n = c->direct_n;
    a.dbg = sol_opt().dbg_skip;   // timing experiments only
    a.feat_tr = g_feat_transposed;
}

}  // namespace

// internal (train.hip): while set, the step launches of THIS host thread write `feat_out` and read `dfeat` in the transposed cell
// order [B][X][Y][C] -- the image order of the CNN when it runs on the transposed grid (cnn_transposed: 64x32) --, which folds the
// two k_transpose_cells launches of every unrolled step into the solver kernels.  Returns the previous value.
int sol_karman_feat_transposed(int on) {
    const int old = g_feat_transposed;
    g_feat_transposed = on ? 1 : 0;
    return old;
}

// one-time: allow the full 160 KiB of dynamic LDS (not a stream operation -> done outside graph capture)
int sol_init_karman_kernels() {
    static std::atomic<unsigned long long> optin{0};
    return sol_lds_optin(optin, {SOL_K(k_karman_fwd<8, 0>), SOL_K(k_karman_bwd<8, 0>), SOL_K(k_karman_fwd<8, 2>), SOL_K(k_karman_bwd<8, 2>),
                                 SOL_K(k_karman_fwd<16, 0>), SOL_K(k_karman_bwd<16, 0>), SOL_K(k_karman_fwd<16, 1>), SOL_K(k_karman_bwd<16, 1>),
                                 SOL_K(k_karman_fwd<16, 2>), SOL_K(k_karman_bwd<16, 2>), SOL_K(k_karman_bwd_bww), SOL_K(k_karman_fwd_dens), SOL_K(k_karman_bwd_bww_small), SOL_K(k_karman_fwd_bands)},
                         "karman kernels");
}

namespace {

// step_prof (debugging, synchronous): the kernels' phase stamps (100 MHz wall clock) of workgroup 0, printed per launch
static long long* prof_buffer() {
    static long long* pbuf = nullptr;
    if (!pbuf && hipMalloc(&pbuf, 32 * sizeof(long long)) != hipSuccess) pbuf = nullptr;
    return pbuf;
}
static int prof_print(hipStream_t stream, const StepArgs& a, bool fused, bool bands = false) {
    long long h[32];
    SOL_HIP_CHECK(hipStreamSynchronize(stream));
    SOL_HIP_CHECK(hipMemcpy(h, a.prof, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[SOL_STEP_PROF %s] us per phase:", a.g_vy_in ? "bwd" : "fwd");
    for (int i = 1; i <= 8; ++i) fprintf(stderr, " %.2f", (double)(h[i] - h[i - 1]) * 0.01);
    fprintf(stderr, "  total %.2f\n", (double)(h[8] - h[0]) * 0.01);
    if (a.fd) {
        const int s0 = bands ? 30 : (a.g_vy_in ? 2 : 5);      // stamp taken just before the solve
        fprintf(stderr, "[SOL_STEP_PROF direct] fwd-transform %.2f  u %.2f  x0w %.2f  K' %.2f  scatter+t2w %.2f  spectral add %.2f  x-inverse %.2f\n",
                (double)(h[9] - h[s0]) * 0.01, (double)(h[10] - h[9]) * 0.01, (double)(h[11] - h[10]) * 0.01, (double)(h[12] - h[11]) * 0.01,
                (double)(h[13] - h[12]) * 0.01, (double)(h[14] - h[13]) * 0.01, (double)(h[15] - h[14]) * 0.01);
    }
    if (bands) {   // band-split forward launch: band 1's stamps (h[20..]: 0 start, 3 advected, 4 divergence sent, 6 pressure received, 8 end) and the solver workgroup's, relative to band 0's first
        fprintf(stderr, "[SOL_STEP_PROF bands] band 1: start %.2f  advected %.2f  divergence sent %.2f  pressure received %.2f  end %.2f\n",
                (double)(h[20] - h[0]) * 0.01, (double)(h[23] - h[0]) * 0.01, (double)(h[24] - h[0]) * 0.01, (double)(h[26] - h[0]) * 0.01, (double)(h[28] - h[0]) * 0.01);
        fprintf(stderr, "[SOL_STEP_PROF bands] solver workgroup: start %.2f  divergence gathered %.2f  solved %.2f  pressure sent %.2f\n",
                (double)(h[29] - h[0]) * 0.01, (double)(h[30] - h[0]) * 0.01, (double)(h[31] - h[0]) * 0.01, (double)(h[27] - h[0]) * 0.01);
    }
    if (fused)      // relative to the adjoint workgroup's first stamp: first / last weight-gradient workgroup
        fprintf(stderr, "[SOL_STEP_PROF fused] adjoint 0.00 .. %.2f | first gradient workgroup %.2f .. %.2f | last %.2f .. %.2f\n",
                (double)(h[8] - h[0]) * 0.01, (double)(h[16] - h[0]) * 0.01, (double)(h[17] - h[0]) * 0.01,
                (double)(h[18] - h[0]) * 0.01, (double)(h[19] - h[0]) * 0.01);
    return SOL_OK;
}

template <typename K>
int launch_step(K kernel, int cpt, const sol_karman_cfg* c, void* stream, const StepArgs& a0) {
    const int threads = (int)align_up((size_t)(c->Y / cpt) * c->X, 64);
    const size_t lds = lds_bytes(c->Y, c->X, cpt);
    if (int e = sol_init_karman_kernels()) return e;
    StepArgs a = a0;
    const bool prof = sol_opt().step_prof != 0;     // debugging: synchronous, prints phase times
    if (prof) a.prof = prof_buffer();
    SOL_LAUNCH_NAMED(a.g_vy_in ? "k_karman_bwd" : "k_karman_fwd", kernel, dim3(c->B), dim3(threads), lds, (hipStream_t)stream, a);
    SOL_LAUNCH_CHECK();
    if (prof && a.prof) return prof_print((hipStream_t)stream, a, false);
    return SOL_OK;
}

}  // namespace

// internal (train.hip): the msteps density advections of the unrolled loop as one launch
int sol_density_chain(const sol_karman_cfg* c, void* stream, int ms, const float* d0, const float* svy, const float* svx,
                      long st_vy, long st_vx, const float* inflow, float* d_steps, long st_d, float* d_final) {
    if (int e = check_cfg(c)) return e;
    SOL_REQUIRE(d0 && svy && svx && inflow && ms >= 1, "sol_density_chain: NULL pointer argument");
    const size_t lds = 2 * (size_t)c->Y * c->X * sizeof(float);
    SOL_REQUIRE(lds <= 160 * 1024, "sol_density_chain: grid does not fit the LDS");
    static std::atomic<unsigned long long> optin{0};
    if (int e = sol_lds_optin(optin, {SOL_K(k_density_chain)}, "k_density_chain")) return e;
    DensArgs a{};
    a.B = c->B; a.Y = c->Y; a.X = c->X; a.ms = ms; a.inflow_before = c->inflow_before;
    a.dtdx = c->dt / c->dx; a.dt = c->dt;
    a.d0 = d0; a.svy = svy; a.svx = svx; a.inflow = inflow; a.st_vy = st_vy; a.st_vx = st_vx; a.st_d = st_d;
    a.d_steps = d_steps; a.d_final = d_final;
    const int threads = c->Y * c->X >= 1024 ? 1024 : (int)align_up((size_t)c->Y * c->X, 64);
    SOL_REQUIRE((size_t)c->Y * c->X <= (size_t)8 * threads, "sol_density_chain: grid too large");
    SOL_LAUNCH(k_density_chain, dim3(c->B), dim3(threads), lds, (hipStream_t)stream, a);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

extern "C" int sol_karman_precond_supported(int32_t Y, int32_t X) { return precond_ok(Y, X) ? 1 : 0; }
extern "C" int sol_karman_direct_supported(int32_t Y, int32_t X) {
    return ((Y == FD_Y && X == FD_X) || (fd_small_floats(Y, X) > 0 && Y % 16 == 0 && (X == 16 || X == 32 || X == 64))) ? 1 : 0;
}

static int step_fwd_impl(const sol_karman_cfg* cfg, void* stream,
                                   const float* d_in, const float* vy_in, const float* vx_in,
                                   const float* re, const float* active, const float* inflow,
                                   const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                                   float* d_out, float* vy_out, float* vx_out,
                                   float* saved_vy, float* saved_vx,
                                   float* feat_out, const float* feat_scale, int32_t* iters,
                                   const float* dens_d_in, const float* dens_svy, const float* dens_svx, float* dens_d_out, uint32_t* xchg = nullptr) {
    if (int e = check_cfg(cfg)) return e;
    SOL_REQUIRE(vy_in && vx_in && re && active && velBCy && velBCyMask && vy_out && vx_out,
                "sol_karman_step_fwd: NULL pointer argument");
    SOL_REQUIRE((d_in && inflow) || !d_out, "density output requested without d_in/inflow");
    SOL_REQUIRE(!feat_out || feat_scale, "feat_out requires feat_scale");
    StepArgs a{};
    fill_common(a, cfg);
    a.d_in = d_in; a.vy_in = vy_in; a.vx_in = vx_in; a.re = re; a.active = active; a.inflow = inflow;
    a.bcv = velBCy; a.bcm = velBCyMask; a.bc_stride = bc_batch_stride;
    a.d_out = d_out; a.vy_out = vy_out; a.vx_out = vx_out; a.saved_vy = saved_vy; a.saved_vx = saved_vx;
    a.feat = feat_out;
    if (feat_scale) { a.fs0 = feat_scale[0]; a.fs1 = feat_scale[1]; a.fs2 = feat_scale[2]; }
    a.iters = iters;
    const int cpt = pick_cpt(cfg);
    if (xchg && sol_karman_fwd_bands_usable(cfg) && !d_out && !a.feat_tr) {
        // four workgroups per simulation (k_karman_fwd_bands); the density workgroups of the previous step, if any, ride behind them
        SOL_REQUIRE(!dens_d_out || (dens_d_in && dens_svy && dens_svx && inflow), "fused solver + density launch: unsupported configuration");
        if (int e = sol_init_karman_kernels()) return e;
        DensStep q{};
        if (dens_d_out) q = DensStep{cfg->B, cfg->Y, cfg->X, cfg->inflow_before, cfg->dt / cfg->dx, cfg->dt, dens_d_in, dens_svy, dens_svx, inflow, dens_d_out};
        if (sol_opt().step_prof) a.prof = prof_buffer();
        const int nsol = 8 * (BD_N + 1) * ((cfg->B + 7) / 8);
        SOL_LAUNCH(k_karman_fwd_bands, dim3(nsol + (dens_d_out ? cfg->B : 0)), dim3(512), lds_bytes(cfg->Y, cfg->X, 16), (hipStream_t)stream, a, q, xchg);
        SOL_LAUNCH_CHECK();
        if (a.prof) return prof_print((hipStream_t)stream, a, false, true);
        return SOL_OK;
    }
    if (dens_d_out) {   // density workgroups of the PREVIOUS step ride in this launch (direct-solver kernels only)
        SOL_REQUIRE(cpt == 16 && a.fd && dens_d_in && dens_svy && dens_svx && inflow, "fused solver + density launch: unsupported configuration");
        if (int e = sol_init_karman_kernels()) return e;
        DensStep q{cfg->B, cfg->Y, cfg->X, cfg->inflow_before, cfg->dt / cfg->dx, cfg->dt, dens_d_in, dens_svy, dens_svx, inflow, dens_d_out};
        if (sol_opt().step_prof) a.prof = prof_buffer();
        SOL_LAUNCH(k_karman_fwd_dens, dim3(2 * cfg->B), dim3(512), lds_bytes(cfg->Y, cfg->X, 16), (hipStream_t)stream, a, q);
        SOL_LAUNCH_CHECK();
        if (a.prof) return prof_print((hipStream_t)stream, a, false);
        return SOL_OK;
    }
    if (cpt != 16) return a.fd ? launch_step(k_karman_fwd<8, 2>, 8, cfg, stream, a) : launch_step(k_karman_fwd<8, 0>, 8, cfg, stream, a);
    if (a.fd) return launch_step(k_karman_fwd<16, 2>, 16, cfg, stream, a);
    if (a.cinv) return launch_step(k_karman_fwd<16, 1>, 16, cfg, stream, a);
    return launch_step(k_karman_fwd<16, 0>, 16, cfg, stream, a);
}

extern "C" int sol_karman_step_fwd(const sol_karman_cfg* cfg, void* stream,
                                   const float* d_in, const float* vy_in, const float* vx_in,
                                   const float* re, const float* active, const float* inflow,
                                   const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                                   float* d_out, float* vy_out, float* vx_out,
                                   float* saved_vy, float* saved_vx,
                                   float* feat_out, const float* feat_scale, int32_t* iters) {
    return step_fwd_impl(cfg, stream, d_in, vy_in, vx_in, re, active, inflow, velBCy, velBCyMask, bc_batch_stride, d_out, vy_out, vx_out,
                         saved_vy, saved_vx, feat_out, feat_scale, iters, nullptr, nullptr, nullptr, nullptr);
}

// internal (train.hip): the solver step with the density advection of the previous step in the same launch
int sol_karman_step_fwd_dens(const sol_karman_cfg* cfg, void* stream,
                             const float* vy_in, const float* vx_in, const float* re, const float* active, const float* inflow,
                             const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                             float* vy_out, float* vx_out, float* saved_vy, float* saved_vx,
                             float* feat_out, const float* feat_scale, int32_t* iters,
                             const float* dens_d_in, const float* dens_svy, const float* dens_svx, float* dens_d_out, uint32_t* xchg) {
    return step_fwd_impl(cfg, stream, nullptr, vy_in, vx_in, re, active, inflow, velBCy, velBCyMask, bc_batch_stride, nullptr, vy_out, vx_out,
                         saved_vy, saved_vx, feat_out, feat_scale, iters, dens_d_in, dens_svy, dens_svx, dens_d_out, xchg);
}
// the band-split forward launch (k_karman_fwd_bands): 128 x 64, direct solver, every workgroup of the launch resident at once.
// xchg: sol_karman_fwd_bands_words(B) zeroed 32-bit words that the launches keep zeroed between uses.
int sol_karman_fwd_bands_usable(const sol_karman_cfg* cfg) {
    return cfg && sol_opt().fwd_bands && cfg->direct && cfg->Y == FD_Y && cfg->X == FD_X && pick_cpt(cfg) == 16 && 8 * (BD_N + 1) * ((cfg->B + 7) / 8) + cfg->B <= 256;
}
size_t sol_karman_fwd_bands_words(int B) { return (size_t)B * BD_WORDS; }
// internal: one density step alone (the last step of the unroll)
int sol_density_step(const sol_karman_cfg* c, void* stream, const float* d_in, const float* svy, const float* svx, const float* inflow, float* d_out) {
    SOL_REQUIRE(c && d_in && svy && svx && inflow && d_out, "sol_density_step: NULL pointer argument");
    DensStep q{c->B, c->Y, c->X, c->inflow_before, c->dt / c->dx, c->dt, d_in, svy, svx, inflow, d_out};
    SOL_LAUNCH(k_density_step, dim3(c->B), dim3(512), 0, (hipStream_t)stream, q);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

static int step_bwd_impl(const sol_karman_cfg* cfg, void* stream,
                         const float* saved_vy, const float* saved_vx,
                         const float* re, const float* active,
                         const float* velBCyMask, int64_t bc_batch_stride,
                         const float* g_vy_out, const float* g_vx_out,
                         const float* dfeat, const float* feat_scale,
                         float* g_vy_in, float* g_vx_in, int32_t* iters,
                         const BwArgs* bw, int nbw, int wg_per, const SolDensRide* dens = nullptr) {
    if (int e = check_cfg(cfg)) return e;
    SOL_REQUIRE(saved_vy && saved_vx && re && active && velBCyMask && g_vy_out && g_vx_out && g_vy_in && g_vx_in,
                "sol_karman_step_bwd: NULL pointer argument");
    SOL_REQUIRE(!dfeat || feat_scale, "dfeat requires feat_scale");
    StepArgs a{};
    fill_common(a, cfg);
    a.saved_vy = const_cast<float*>(saved_vy); a.saved_vx = const_cast<float*>(saved_vx);
    a.re = re; a.active = active; a.bcm = velBCyMask; a.bc_stride = bc_batch_stride;
    a.g_vy_out = g_vy_out; a.g_vx_out = g_vx_out; a.dfeat = dfeat;
    if (feat_scale) { a.fs0 = feat_scale[0]; a.fs1 = feat_scale[1]; a.fs2 = feat_scale[2]; }
    a.g_vy_in = g_vy_in; a.g_vx_in = g_vx_in; a.iters = iters;
    const int cpt = pick_cpt(cfg);
    if (nbw > 0) {      // weight-gradient workgroups ride in the adjoint's launch (direct-solver kernels: 128x64 with 16-cell strips, 64x32 with 8-cell strips)
        SOL_REQUIRE((sol_karman_bwd_fusable(cfg) || sol_karman_bwd_fusable_small(cfg)) && nbw <= 12 && wg_per >= 1, "fused adjoint + weight-gradient launch: unsupported configuration");
        if (int e = sol_init_karman_kernels()) return e;
        BwPack pk{};
        for (int k = 0; k < nbw; ++k) pk.a[k] = bw[k];
        pk.n = nbw; pk.wg_per = wg_per;
        size_t lds = lds_bytes(cfg->Y, cfg->X, cpt);
        if (lds < (size_t)sbk::BW_LDS) lds = sbk::BW_LDS;
        if (sol_opt().step_prof) a.prof = prof_buffer();
        SOL_REQUIRE(!dens || cpt == 8, "a density advection rides only in the 64x32 fused launch");
        if (cpt == 8) {
            DensStep q{};
            if (dens) q = DensStep{cfg->B, cfg->Y, cfg->X, cfg->inflow_before, cfg->dt / cfg->dx, cfg->dt, dens->d_in, dens->svy, dens->svx, dens->inflow, dens->d_out};
            SOL_LAUNCH(k_karman_bwd_bww_small, dim3(cfg->B + nbw * wg_per + (dens ? cfg->B : 0)), dim3(512), lds, (hipStream_t)stream, a, pk, q);
            SOL_LAUNCH_CHECK();
            return SOL_OK;
        }
        SOL_LAUNCH(k_karman_bwd_bww, dim3(cfg->B + nbw * wg_per), dim3(512), lds, (hipStream_t)stream, a, pk);
        SOL_LAUNCH_CHECK();
        if (a.prof) return prof_print((hipStream_t)stream, a, true);
        return SOL_OK;
    }
    if (cpt != 16) return a.fd ? launch_step(k_karman_bwd<8, 2>, 8, cfg, stream, a) : launch_step(k_karman_bwd<8, 0>, 8, cfg, stream, a);
    if (a.fd) return launch_step(k_karman_bwd<16, 2>, 16, cfg, stream, a);
    if (a.cinv) return launch_step(k_karman_bwd<16, 1>, 16, cfg, stream, a);
    return launch_step(k_karman_bwd<16, 0>, 16, cfg, stream, a);
}

extern "C" int sol_karman_step_bwd(const sol_karman_cfg* cfg, void* stream,
                                   const float* saved_vy, const float* saved_vx,
                                   const float* re, const float* active,
                                   const float* velBCyMask, int64_t bc_batch_stride,
                                   const float* g_vy_out, const float* g_vx_out,
                                   const float* dfeat, const float* feat_scale,
                                   float* g_vy_in, float* g_vx_in, int32_t* iters) {
    return step_bwd_impl(cfg, stream, saved_vy, saved_vx, re, active, velBCyMask, bc_batch_stride, g_vy_out, g_vx_out, dfeat, feat_scale,
                         g_vy_in, g_vx_in, iters, nullptr, 0, 0);
}

// internal (train.hip): the solver adjoint with `nbw` weight-gradient jobs of `wg_per` workgroups each in the same launch
int sol_karman_step_bwd_fused(const sol_karman_cfg* cfg, void* stream,
                              const float* saved_vy, const float* saved_vx, const float* re, const float* active,
                              const float* velBCyMask, int64_t bc_batch_stride,
                              const float* g_vy_out, const float* g_vx_out, const float* dfeat, const float* feat_scale,
                              float* g_vy_in, float* g_vx_in, int32_t* iters, const BwArgs* bw, int nbw, int wg_per, const SolDensRide* dens) {
    return step_bwd_impl(cfg, stream, saved_vy, saved_vx, re, active, velBCyMask, bc_batch_stride, g_vy_out, g_vx_out, dfeat, feat_scale,
                         g_vy_in, g_vx_in, iters, bw, nbw, wg_per, dens);
}
// weight-gradient jobs alone (the first step of the unroll has no solver adjoint): same kernel, no solver workgroups
int sol_bww_jobs_launch(void* stream, const BwArgs* bw, int nbw, int wg_per, const sol_karman_cfg* cfg, const SolDensRide* dens) {
    SOL_REQUIRE(bw && nbw >= 1 && nbw <= 12 && wg_per >= 1 && (!dens || cfg), "sol_bww_jobs_launch: bad arguments");
    if (int e = sol_init_karman_kernels()) return e;
    StepArgs a{};
    a.B = 0;
    BwPack pk{};
    for (int k = 0; k < nbw; ++k) pk.a[k] = bw[k];
    pk.n = nbw; pk.wg_per = wg_per;
    if (dens) {         // with a density advection riding along (64x32 training path)
        const DensStep q{cfg->B, cfg->Y, cfg->X, cfg->inflow_before, cfg->dt / cfg->dx, cfg->dt, dens->d_in, dens->svy, dens->svx, dens->inflow, dens->d_out};
        SOL_LAUNCH(k_karman_bwd_bww_small, dim3(nbw * wg_per + cfg->B), dim3(512), (size_t)sbk::BW_LDS, (hipStream_t)stream, a, pk, q);
        SOL_LAUNCH_CHECK();
        return SOL_OK;
    }
    SOL_LAUNCH(k_karman_bwd_bww, dim3(nbw * wg_per), dim3(512), (size_t)sbk::BW_LDS, (hipStream_t)stream, a, pk);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}
// 1 if the fused adjoint + weight-gradient launch exists for this configuration
int sol_karman_bwd_fusable(const sol_karman_cfg* cfg) {
    return cfg && cfg->direct && cfg->Y == FD_Y && cfg->X == FD_X && pick_cpt(cfg) == 16;
}
// ... and for the 64 x 32 grid (256 solver threads inside 512-thread workgroups: k_karman_bwd_bww_small)
int sol_karman_bwd_fusable_small(const sol_karman_cfg* cfg) {
    return cfg && cfg->direct && cfg->Y == 64 && cfg->X == 32 && pick_cpt(cfg) == 8;
}
