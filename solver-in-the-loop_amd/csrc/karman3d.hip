// karman-3d solver step and its adjoint (BASELINE.json configs[4]: 128 x 64 x 64, SOL-16) for gfx950.
//
// The reference has no 3-D code (/root/reference/README.md:37-38); the algorithm is the dimension-generic restatement of
// KarmanFlow.step (/root/reference/karman-2d/karman_train.py:173-185) + PhiFlow's IncompressibleFlow.step, exactly as
// oracle/sol_oracle3d.py states it.  Layout: density [B,Y,X,Z], v_y [B,Y+1,X,Z], v_x [B,Y,X+1,Z], v_z [B,Y,X,Z+1]; y = flow
// direction, z contiguous.
//
// A 128 x 64 x 64 component is 2 MB: the state cannot live in one CU's LDS like the 2-D step (karman_step.hip), so the step is
// a sequence of chip-wide launches (the structure of karman_large.hip) over L2 / MALL resident fields:
//   k3_diffuse      explicit 7-point diffusion (replicate padding, dx = 1) + velocity BC on the flow component  -> sv_*
//   k3_advect_tile  semi-Lagrangian advection of the three components (clamped) and the density (zero ghost ring) + inflow +
//                   hard-BC face masks.  One workgroup = an 8 x 8 (y, x) tile holding the FULL z column of the three diffused
//                   components with a 2-cell halo in LDS (13 x 12 x 64 floats per component, 117 KB at Z = 64): every
//                   trilinear gather and every cross-component average of the tile is served from LDS, lanes run along z
//                   (conflict free, coalesced stores); samples that leave the halo (CFL > 1) fall back to global reads.
//                   (k3_advect: the same arithmetic straight from global memory -- option k3d_tile = 0, and any Z > 64.)
//   k3_div          rhs = -div
//   pressure        DIRECT solve x = G (b - U_S E_SS x_S), x_S = (I + G_SS E_SS)^-1 (G b)_S with G = the empty-box Dirichlet
//                   Laplacian diagonalised by three sine transforms (batched fp32 GEMMs along z, x, y) and the capacitance
//                   correction of the obstacle cells (precond3d.direct_solver_blob3d): 12 transform passes + k3_capacitance
//   k3_project      v -= mask * grad p, fused to_feature (4 channels: three components + Re)
// The adjoint of the step (sol_karman3d_step_bwd: k3b_* kernels, second half of this file) reverses these stages; its
// advection scatter runs in 64-bit fixed point and is bit-reproducible.
#include "common.hpp"

namespace {

constexpr int FD3_HEADER = 16;
constexpr int FD3_MAGIC = 0x46443333;     // "FD33"

struct K3Args {
    int B, Y, X, Z;
    float dtdx, dt, adt;
    int grad_pad, inflow_before;
    const float *d_in, *vy_in, *vx_in, *vz_in, *re, *active, *inflow, *bcv, *bcm;
    long bc_stride;
    float *d_out, *vy_out, *vx_out, *vz_out, *svy, *svx, *svz, *rhs, *feat;
    const float* p;
    float fs0, fs1, fs2, fs3;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// 7-point replicate-padded Laplacian of one component array [n0][n1][n2] at (j, i, k)
__device__ __forceinline__ float lap7(const float* f, int n0, int n1, int n2, int j, int i, int k) {
    const size_t s0 = (size_t)n1 * n2, s1 = n2;
    const size_t c = (size_t)j * s0 + (size_t)i * s1 + k;
    const float v = f[c];
    return f[j + 1 < n0 ? c + s0 : c] + f[j > 0 ? c - s0 : c] + f[i + 1 < n1 ? c + s1 : c] + f[i > 0 ? c - s1 : c] +
           f[k + 1 < n2 ? c + 1 : c] + f[k > 0 ? c - 1 : c] - 6.f * v;
}

__global__ void __launch_bounds__(256) k3_diffuse(K3Args a) {
    const int Y = a.Y, X = a.X, Z = a.Z;
    const int nVy = (Y + 1) * X * Z, nVx = Y * (X + 1) * Z, nVz = Y * X * (Z + 1);
    const int b = blockIdx.y;
    const float alpha = a.adt / a.re[b];
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nVy + nVx + nVz; e += gridDim.x * blockDim.x) {
        if (e < nVy) {
            const int k = e % Z, i = (e / Z) % X, j = e / (Z * X);
            const float* f = a.vy_in + (size_t)b * nVy;
            float v = f[e] + alpha * lap7(f, Y + 1, X, Z, j, i, k);
            const size_t m = (size_t)b * a.bc_stride + e;
            v = v * (1.f - a.bcm[m]) + a.bcv[m];
            a.svy[(size_t)b * nVy + e] = v;
        } else if (e < nVy + nVx) {
            const int q = e - nVy, k = q % Z, i = (q / Z) % (X + 1), j = q / (Z * (X + 1));
            const float* f = a.vx_in + (size_t)b * nVx;
            a.svx[(size_t)b * nVx + q] = f[q] + alpha * lap7(f, Y, X + 1, Z, j, i, k);
        } else {
            const int q = e - nVy - nVx, k = q % (Z + 1), i = (q / (Z + 1)) % X, j = q / ((Z + 1) * X);
            const float* f = a.vz_in + (size_t)b * nVz;
            a.svz[(size_t)b * nVz + q] = f[q] + alpha * lap7(f, Y, X, Z + 1, j, i, k);
        }
    }
}

// ---- hard-BC face masks from the `active` cell mask ('boundary' extrapolation: outside the OPEN domain = edge value) ----
__device__ __forceinline__ float acc_at(const float* act, int Y, int X, int Z, int j, int i, int k) {
    return act[((size_t)clampi(j, 0, Y - 1) * X + clampi(i, 0, X - 1)) * Z + clampi(k, 0, Z - 1)] != 0.f ? 1.f : 0.f;
}
template <int AX>
__device__ __forceinline__ float face_mask(const float* act, int Y, int X, int Z, int j, int i, int k) {
    return acc_at(act, Y, X, Z, j - (AX == 0), i - (AX == 1), k - (AX == 2)) * acc_at(act, Y, X, Z, j, i, k);
}

// ---- readers of the diffused components: straight from global memory, or from the workgroup's LDS tile ----------------
struct GlobalReader {
    const float *sy, *sx, *sz;
    const float* act;                 // the scene's `active` cell mask [Y][X][Z]
    int Y, X, Z;
    // accessible mask at a cell index that may lie one cell outside the domain ('boundary' extrapolation: edge value)
    __device__ __forceinline__ float acc(int j, int i, int k) const { return acc_at(act, Y, X, Z, j, i, k); }
    __device__ __forceinline__ float y(int j, int i, int k) const { return sy[((size_t)j * X + i) * Z + k]; }
    __device__ __forceinline__ float x(int j, int i, int k) const { return sx[((size_t)j * (X + 1) + i) * Z + k]; }
    __device__ __forceinline__ float z(int j, int i, int k) const { return sz[((size_t)j * X + i) * (Z + 1) + k]; }
};
constexpr int T3 = 8;          // tile edge (cells) in y and x
constexpr int HL3 = 2;         // halo cells
constexpr int TR = T3 + 2 * HL3;      // cell rows / columns of the tile region (12); face arrays have one more along their axis
struct TileReader {
    GlobalReader g;
    const float *ly, *lx, *lz;        // LDS: [TR+1][TR][Z], [TR][TR+1][Z], [TR][TR][Z+1]
    const unsigned char* la;          // LDS: accessible mask of the region's cells [TR][TR][Z] (bytes; rows / columns beyond the domain hold the edge value)
    int j0, i0;                       // first cell row / column of the region (may be negative: clipped rows are never read)
    __device__ __forceinline__ float acc(int j, int i, int k) const {
        const int rj = j - j0, ri = i - i0, kk = clampi(k, 0, g.Z - 1);
        return ((unsigned)rj < (unsigned)TR && (unsigned)ri < (unsigned)TR) ? (float)la[(rj * TR + ri) * g.Z + kk] : g.acc(j, i, k);
    }
    __device__ __forceinline__ float y(int j, int i, int k) const {
        const int rj = j - j0, ri = i - i0;
        return ((unsigned)rj <= (unsigned)TR && (unsigned)ri < (unsigned)TR) ? ly[(rj * TR + ri) * g.Z + k] : g.y(j, i, k);
    }
    __device__ __forceinline__ float x(int j, int i, int k) const {
        const int rj = j - j0, ri = i - i0;
        return ((unsigned)rj < (unsigned)TR && (unsigned)ri <= (unsigned)TR) ? lx[(rj * (TR + 1) + ri) * g.Z + k] : g.x(j, i, k);
    }
    __device__ __forceinline__ float z(int j, int i, int k) const {
        const int rj = j - j0, ri = i - i0;
        return ((unsigned)rj < (unsigned)TR && (unsigned)ri < (unsigned)TR) ? lz[(rj * TR + ri) * (g.Z + 1) + k] : g.z(j, i, k);
    }
};

// trilinear sample of component C (0: y faces [Y+1][X][Z], 1: x faces, 2: z faces) at index-space position
// (j + oy, i + ox, k + oz) of its own lattice, indices clamped ('boundary' extrapolation).  floor() acts on the OFFSET only:
// the interpolation weights keep their full fp32 precision whatever the base index.
template <int C, class R>
__device__ __forceinline__ float tri_clamp(const R& r, int j, float oy, int i, float ox, int k, float oz) {
    const int n0 = r.g_Y() + (C == 0), n1 = r.g_X() + (C == 1), n2 = r.g_Z() + (C == 2);
    const float fy = floorf(oy), fx = floorf(ox), fz = floorf(oz);
    const float wy = oy - fy, wx = ox - fx, wz = oz - fz;
    const int ja = j + (int)fy, ia = i + (int)fx, ka = k + (int)fz;
    const int j0 = clampi(ja, 0, n0 - 1), j1 = clampi(ja + 1, 0, n0 - 1);
    const int i0 = clampi(ia, 0, n1 - 1), i1 = clampi(ia + 1, 0, n1 - 1);
    const int k0 = clampi(ka, 0, n2 - 1), k1 = clampi(ka + 1, 0, n2 - 1);
    auto rd = [&](int jj, int ii, int kk) { return C == 0 ? r.y(jj, ii, kk) : (C == 1 ? r.x(jj, ii, kk) : r.z(jj, ii, kk)); };
    const float c00 = (1.f - wz) * rd(j0, i0, k0) + wz * rd(j0, i0, k1);
    const float c01 = (1.f - wz) * rd(j0, i1, k0) + wz * rd(j0, i1, k1);
    const float c10 = (1.f - wz) * rd(j1, i0, k0) + wz * rd(j1, i0, k1);
    const float c11 = (1.f - wz) * rd(j1, i1, k0) + wz * rd(j1, i1, k1);
    return (1.f - wy) * ((1.f - wx) * c00 + wx * c01) + wy * ((1.f - wx) * c10 + wx * c11);
}
struct GR : GlobalReader {
    __device__ __forceinline__ int g_Y() const { return Y; }
    __device__ __forceinline__ int g_X() const { return X; }
    __device__ __forceinline__ int g_Z() const { return Z; }
};
struct TRd : TileReader {
    __device__ __forceinline__ int g_Y() const { return g.Y; }
    __device__ __forceinline__ int g_X() const { return g.X; }
    __device__ __forceinline__ int g_Z() const { return g.Z; }
};

// one advected value per call.  kind 0/1/2: face of that component at (j, i, k); kind 3: cell (j, i, k).
// Velocity at the sample point: the own component is the stored value, the others are the bilinear averages of their four
// surrounding faces (index clamped: 'boundary' extrapolation of the component grids)  [oracle: advect_mac]
template <class R>
__device__ __forceinline__ void advect_point(const K3Args& a, const R& r, int b, int kind, int j, int i, int k) {
    const int Y = a.Y, X = a.X, Z = a.Z;
    if (kind == 0) {
        const float uy = r.y(j, i, k);
        const int ja = max(j - 1, 0), jb = min(j, Y - 1);
        const float ux = 0.25f * (r.x(ja, i, k) + r.x(ja, i + 1, k) + r.x(jb, i, k) + r.x(jb, i + 1, k));
        const float uz = 0.25f * (r.z(ja, i, k) + r.z(ja, i, k + 1) + r.z(jb, i, k) + r.z(jb, i, k + 1));
        const float v = tri_clamp<0>(r, j, -uy * a.dtdx, i, -ux * a.dtdx, k, -uz * a.dtdx);
        a.vy_out[(size_t)b * (Y + 1) * X * Z + ((size_t)j * X + i) * Z + k] = v * (r.acc(j - 1, i, k) * r.acc(j, i, k));
    } else if (kind == 1) {
        const float ux = r.x(j, i, k);
        const int ia = max(i - 1, 0), ib = min(i, X - 1);
        const float uy = 0.25f * (r.y(j, ia, k) + r.y(j, ib, k) + r.y(j + 1, ia, k) + r.y(j + 1, ib, k));
        const float uz = 0.25f * (r.z(j, ia, k) + r.z(j, ia, k + 1) + r.z(j, ib, k) + r.z(j, ib, k + 1));
        const float v = tri_clamp<1>(r, j, -uy * a.dtdx, i, -ux * a.dtdx, k, -uz * a.dtdx);
        a.vx_out[(size_t)b * Y * (X + 1) * Z + ((size_t)j * (X + 1) + i) * Z + k] = v * (r.acc(j, i - 1, k) * r.acc(j, i, k));
    } else if (kind == 2) {
        const float uz = r.z(j, i, k);
        const int ka = max(k - 1, 0), kb = min(k, Z - 1);
        const float uy = 0.25f * (r.y(j, i, ka) + r.y(j, i, kb) + r.y(j + 1, i, ka) + r.y(j + 1, i, kb));
        const float ux = 0.25f * (r.x(j, i, ka) + r.x(j, i, kb) + r.x(j, i + 1, ka) + r.x(j, i + 1, kb));
        const float v = tri_clamp<2>(r, j, -uy * a.dtdx, i, -ux * a.dtdx, k, -uz * a.dtdx);
        a.vz_out[(size_t)b * Y * X * (Z + 1) + ((size_t)j * X + i) * (Z + 1) + k] = v * (r.acc(j, i, k - 1) * r.acc(j, i, k));
    } else {
        const size_t N = (size_t)Y * X * Z, c = ((size_t)j * X + i) * Z + k;
        const float uy = 0.5f * (r.y(j, i, k) + r.y(j + 1, i, k));
        const float ux = 0.5f * (r.x(j, i, k) + r.x(j, i + 1, k));
        const float uz = 0.5f * (r.z(j, i, k) + r.z(j, i, k + 1));
        const float oy = -uy * a.dtdx, ox = -ux * a.dtdx, oz = -uz * a.dtdx;
        const float fy = floorf(oy), fx = floorf(ox), fz = floorf(oz);
        const float wy = oy - fy, wx = ox - fx, wz = oz - fz;
        const int j0 = j + (int)fy, i0 = i + (int)fx, k0 = k + (int)fz;
        const float* gd = a.d_in + (size_t)b * N;
        float acc = 0.f;
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int di = 0; di < 2; ++di)
#pragma unroll
                for (int dk = 0; dk < 2; ++dk) {
                    const int jj = j0 + dj, ii = i0 + di, kk = k0 + dk;
                    float v = 0.f;                    // extrapolation 'constant': one ring of zero ghost cells
                    if ((unsigned)jj < (unsigned)Y && (unsigned)ii < (unsigned)X && (unsigned)kk < (unsigned)Z) {
                        const size_t q = ((size_t)jj * X + ii) * Z + kk;
                        v = gd[q];
                        if (a.inflow_before) v += a.inflow[q];
                    }
                    acc += (dj ? wy : 1.f - wy) * (di ? wx : 1.f - wx) * (dk ? wz : 1.f - wz) * v;
                }
        if (!a.inflow_before) acc += a.inflow[c] * a.dt;
        a.d_out[(size_t)b * N + c] = acc;
    }
}

// Work unit of both advection kernels: one z COLUMN (fixed kind, j, i) per wave, lanes along z.  j, i and everything derived
// from them (array bounds, neighbour rows, "is this row in the LDS region") are wave uniform -- scalar registers and scalar
// branches; no per-point index decoding (three integer divisions per point cost more than the interpolation itself).
template <class R>
__device__ __forceinline__ void advect_column(const K3Args& a, const R& r, int b, int kind, int j, int i, int lane) {
    const int nk = a.Z + (kind == 2 ? 1 : 0);
    for (int k = lane; k < nk; k += 64) advect_point(a, r, b, kind, j, i, k);
}

__global__ void __launch_bounds__(256) k3_advect(K3Args a) {
    const int Y = a.Y, X = a.X, Z = a.Z;
    const int nVy = (Y + 1) * X * Z, nVx = Y * (X + 1) * Z, nVz = Y * X * (Z + 1);
    const int b = blockIdx.y;
    GR r;
    r.sy = a.svy + (size_t)b * nVy; r.sx = a.svx + (size_t)b * nVx; r.sz = a.svz + (size_t)b * nVz; r.act = a.active; r.Y = Y; r.X = X; r.Z = Z;
    const int cY = (Y + 1) * X, cX = Y * (X + 1), cC = Y * X;        // columns per kind: y faces, x faces, z faces (= cells), cells
    const int total = cY + cX + cC + (a.d_out ? cC : 0);
    const int lane = threadIdx.x & 63;
    const int wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    for (int c = wave0; c < total; c += nwaves) {
        if (c < cY) advect_column(a, r, b, 0, c / X, c % X, lane);
        else if (c < cY + cX) { const int q = c - cY; advect_column(a, r, b, 1, q / (X + 1), q % (X + 1), lane); }
        else if (c < cY + cX + cC) { const int q = c - cY - cX; advect_column(a, r, b, 2, q / X, q % X, lane); }
        else { const int q = c - cY - cX - cC; advect_column(a, r, b, 3, q / X, q % X, lane); }
    }
}

// One workgroup = tile (ty, tx) of 8 x 8 cells x the full z column of simulation b.  It owns the y faces j0..j0+7 (the last
// tile row also face Y), the x faces i0..i0+7 (the last tile column also face X), all z faces and cells of its columns.
constexpr int ADV_T = 1024;      // 16 waves per workgroup: the tile kernel owns a CU (117 KB of LDS), its gathers are latency bound
__global__ void __launch_bounds__(ADV_T) k3_advect_tile(K3Args a, int tiles_x) {
    extern __shared__ __align__(16) float lds3[];
    const int Y = a.Y, X = a.X, Z = a.Z;
    const int nVy = (Y + 1) * X * Z, nVx = Y * (X + 1) * Z, nVz = Y * X * (Z + 1);
    const int b = blockIdx.y, ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    const int jt = ty * T3, it = tx * T3;             // first owned cell
    TRd r;
    r.g.sy = a.svy + (size_t)b * nVy; r.g.sx = a.svx + (size_t)b * nVx; r.g.sz = a.svz + (size_t)b * nVz;
    r.g.Y = Y; r.g.X = X; r.g.Z = Z;
    r.j0 = jt - HL3; r.i0 = it - HL3;
    float* ly = lds3;
    float* lx = ly + (TR + 1) * TR * Z;
    float* lz = lx + TR * (TR + 1) * Z;
    unsigned char* la = reinterpret_cast<unsigned char*>(lz + TR * TR * (Z + 1));
    r.ly = ly; r.lx = lx; r.lz = lz; r.la = la; r.g.act = a.active;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int NW = ADV_T / 64;
    // stage the region column by column (lanes along z).  The loads are unconditional with clamped indices (a predicated load
    // waits inside its own block: one exposed round trip per column, 28 columns per wave), four columns in flight per wave;
    // positions outside the arrays then hold edge values that no clamped index ever addresses.
    auto stage = [&](float* dst, int ncol_r, int ncol_i, int nj, int ni, int nz, auto&& rd) {
        const int ncol = ncol_r * ncol_i;
        for (int c0 = wave; c0 < ncol; c0 += 4 * NW) {
            float v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = min(c0 + u * NW, ncol - 1), j = clampi(r.j0 + c / ncol_i, 0, nj - 1), i = clampi(r.i0 + c % ncol_i, 0, ni - 1);
                v[u][0] = rd(j, i, min(lane, nz - 1));
                v[u][1] = rd(j, i, min(lane + 64, nz - 1));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * NW;
                if (c < ncol) {
                    if (lane < nz) dst[c * nz + lane] = v[u][0];
                    if (lane + 64 < nz) dst[c * nz + lane + 64] = v[u][1];
                }
            }
        }
    };
    stage(ly, TR + 1, TR, Y + 1, X, Z, [&](int j, int i, int k) { return r.g.y(j, i, k); });
    stage(lx, TR, TR + 1, Y, X + 1, Z, [&](int j, int i, int k) { return r.g.x(j, i, k); });
    stage(lz, TR, TR, Y, X, Z + 1, [&](int j, int i, int k) { return r.g.z(j, i, k); });
    for (int c = wave; c < TR * TR; c += NW) {            // accessible mask of the region's cells: clamped = 'boundary' extrapolation
        const int j = r.j0 + c / TR, i = r.i0 + c % TR;
        for (int k = lane; k < Z; k += 64) la[c * Z + k] = acc_at(a.active, Y, X, Z, j, i, k) != 0.f ? 1 : 0;
    }
    __syncthreads();
    const int ny = min(T3, Y - jt) + (jt + T3 >= Y ? 1 : 0);      // y-face rows owned
    const int nx = min(T3, X - it) + (it + T3 >= X ? 1 : 0);      // x-face columns owned
    const int cy = min(T3, Y - jt), cx = min(T3, X - it);         // cells owned
    const int cY = ny * cx, cX = cy * nx, cC = cy * cx;
    const int total = cY + cX + cC + (a.d_out ? cC : 0);
    for (int c = wave; c < total; c += NW) {
        if (c < cY) advect_column(a, r, b, 0, jt + c / cx, it + c % cx, lane);
        else if (c < cY + cX) { const int q = c - cY; advect_column(a, r, b, 1, jt + q / nx, it + q % nx, lane); }
        else if (c < cY + cX + cC) { const int q = c - cY - cX; advect_column(a, r, b, 2, jt + q / cx, it + q % cx, lane); }
        else { const int q = c - cY - cX - cC; advect_column(a, r, b, 3, jt + q / cx, it + q % cx, lane); }
    }
}

__global__ void __launch_bounds__(256) k3_div(K3Args a) {
    const int Y = a.Y, X = a.X, Z = a.Z, N = Y * X * Z;
    const int b = blockIdx.y;
    const float* vy = a.vy_out + (size_t)b * (Y + 1) * X * Z;
    const float* vx = a.vx_out + (size_t)b * Y * (X + 1) * Z;
    const float* vz = a.vz_out + (size_t)b * Y * X * (Z + 1);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < N; c += gridDim.x * blockDim.x) {
        const int k = c % Z, i = (c / Z) % X, j = c / (Z * X);
        const float div = (vy[((size_t)(j + 1) * X + i) * Z + k] - vy[((size_t)j * X + i) * Z + k]) +
                          (vx[((size_t)j * (X + 1) + i + 1) * Z + k] - vx[((size_t)j * (X + 1) + i) * Z + k]) +
                          (vz[((size_t)j * X + i) * (Z + 1) + k + 1] - vz[((size_t)j * X + i) * (Z + 1) + k]);
        a.rhs[(size_t)b * N + c] = -div;           // M p = -div  <=>  A p = div
    }
}

__global__ void __launch_bounds__(256) k3_project(K3Args a) {
    const int Y = a.Y, X = a.X, Z = a.Z, N = Y * X * Z;
    const int nVy = (Y + 1) * X * Z, nVx = Y * (X + 1) * Z, nVz = Y * X * (Z + 1);
    const int b = blockIdx.y;
    const float* P = a.p + (size_t)b * N;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nVy + nVx + nVz; e += gridDim.x * blockDim.x) {
        if (e < nVy) {
            const int k = e % Z, i = (e / Z) % X, j = e / (Z * X);
            float g = 0.f;
            if (j >= 1 && j <= Y - 1) g = P[((size_t)j * X + i) * Z + k] - P[((size_t)(j - 1) * X + i) * Z + k];
            else if (a.grad_pad == 1) g = j == 0 ? P[((size_t)i) * Z + k] : -P[((size_t)(Y - 1) * X + i) * Z + k];
            float* dst = a.vy_out + (size_t)b * nVy + e;
            const float v = *dst - face_mask<0>(a.active, Y, X, Z, j, i, k) * g;
            *dst = v;
            if (a.feat && j < Y) a.feat[((size_t)b * N + e) * 4 + 0] = v * a.fs0;
        } else if (e < nVy + nVx) {
            const int q = e - nVy, k = q % Z, i = (q / Z) % (X + 1), j = q / (Z * (X + 1));
            float g = 0.f;
            if (i >= 1 && i <= X - 1) g = P[((size_t)j * X + i) * Z + k] - P[((size_t)j * X + i - 1) * Z + k];
            else if (a.grad_pad == 1) g = i == 0 ? P[((size_t)j * X) * Z + k] : -P[((size_t)j * X + X - 1) * Z + k];
            float* dst = a.vx_out + (size_t)b * nVx + q;
            const float v = *dst - face_mask<1>(a.active, Y, X, Z, j, i, k) * g;
            *dst = v;
            if (a.feat && i < X) a.feat[((size_t)b * N + ((size_t)j * X + i) * Z + k) * 4 + 1] = v * a.fs1;
        } else {
            const int q = e - nVy - nVx, k = q % (Z + 1), i = (q / (Z + 1)) % X, j = q / ((Z + 1) * X);
            float g = 0.f;
            if (k >= 1 && k <= Z - 1) g = P[((size_t)j * X + i) * Z + k] - P[((size_t)j * X + i) * Z + k - 1];
            else if (a.grad_pad == 1) g = k == 0 ? P[((size_t)j * X + i) * Z] : -P[((size_t)j * X + i) * Z + Z - 1];
            float* dst = a.vz_out + (size_t)b * nVz + q;
            const float v = *dst - face_mask<2>(a.active, Y, X, Z, j, i, k) * g;
            *dst = v;
            if (a.feat && k < Z) {
                float* f = a.feat + ((size_t)b * N + ((size_t)j * X + i) * Z + k) * 4;
                f[2] = v * a.fs2;
                f[3] = a.re[b] * a.fs3;
            }
        }
    }
}

// T[e] *= il[e]  (1 / eigenvalue of the empty-box Laplacian, natural [m][c][e] order)
__global__ void __launch_bounds__(256) k3_scale(float* __restrict__ T, const float* __restrict__ il, int N) {
    const int b = blockIdx.y;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N; e += gridDim.x * blockDim.x) T[(size_t)b * N + e] *= il[e];
}

// capacitance correction: xs = g[sidx], c = K' xs (KpT = K' transposed, [SP][SP], zero padded), rhs[sidx] -= c.
// Stage 1, grid (SP / 32, CAPQ, B): 32 outputs x 8 slices of this workgroup's chunk of the sum -> part[b][chunk][s]
// (the dense product is a 10 MB stream at 128 x 64 x 64: it wants the whole chip, not SP / 64 workgroups);
// stage 2 folds the CAPQ chunks in a fixed order (deterministic) and applies the correction.
constexpr int CAPQ = 8;
__global__ void __launch_bounds__(256) k3_capacitance(const float* __restrict__ g, const float* __restrict__ KpT, const int* __restrict__ sidx,
                                                       float* __restrict__ part, int SP, int N) {
    __shared__ float xs[1024];                // this chunk of xs (SP / CAPQ <= 1024)
    __shared__ float red[8][32];
    const int b = blockIdx.z, ch = blockIdx.y;
    const int qn = SP / CAPQ, q0 = ch * qn;
    const float* gb = g + (size_t)b * N;
    for (int t = threadIdx.x; t < qn; t += 256) { const int si = sidx[q0 + t]; xs[t] = si >= 0 ? gb[si] : 0.f; }
    __syncthreads();
    const int sl = threadIdx.x >> 5, s = blockIdx.x * 32 + (threadIdx.x & 31);
    float c = 0.f;
    for (int q = sl; q < qn; q += 8) c += KpT[(size_t)(q0 + q) * SP + s] * xs[q];
    red[sl][threadIdx.x & 31] = c;
    __syncthreads();
    if (threadIdx.x < 32) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += red[k][threadIdx.x];
        part[((size_t)b * CAPQ + ch) * SP + s] = v;
    }
}
__global__ void __launch_bounds__(256) k3_cap_apply(const float* __restrict__ part, const int* __restrict__ sidx, float* __restrict__ rhs, int SP, int N) {
    const int b = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    if (s >= SP) return;
    const int si = sidx[s];
    if (si < 0) return;
    float c = 0.f;
#pragma unroll
    for (int k = 0; k < CAPQ; ++k) c += part[((size_t)b * CAPQ + k) * SP + s];
    rhs[(size_t)b * N + si] -= c;
}

// ---- sine transforms with LDS-resident planes / column slabs (X, Z <= 64, Y <= 128) -----------------------------------
// k3_tzx: one workgroup = one (x, z) plane of one simulation: out = Qx (F Qz), both products from LDS (the two transforms
// of a plane cost ONE read and ONE write of the plane instead of two each).  Q symmetric, so the same kernel serves the
// forward and the inverse direction.
__global__ void __launch_bounds__(256) k3_tzx(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ Qx,
                                               const float* __restrict__ Qz, int X, int Z) {
    __shared__ float A[64][65], Q[64][65];
    const size_t plane = (size_t)blockIdx.x * X * Z;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    for (int e = t; e < X * Z; e += 256) A[e / Z][e % Z] = in[plane + e];
    for (int e = t; e < Z * Z; e += 256) Q[e / Z][e % Z] = Qz[e];
    __syncthreads();
    float acc[4][4];
    auto mm = [&](int K, bool left) {        // left: acc = Q[rows ty*4..][k] * A[k][cols tx*4..]; else acc = A[rows][k] * Q[k][cols]
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
        for (int k = 0; k < K; ++k) {
            float av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { av[r] = left ? Q[k][ty * 4 + r] : A[ty * 4 + r][k]; bv[r] = left ? A[k][tx * 4 + r] : Q[k][tx * 4 + r]; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] += av[r] * bv[c];
        }
    };
    mm(Z, false);                            // T = F Qz   (rows: x, cols: z)
    __syncthreads();
    if (ty * 4 < X && tx * 4 < Z)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) A[ty * 4 + r][tx * 4 + c] = acc[r][c];
    for (int e = t; e < X * X; e += 256) Q[e / X][e % X] = Qx[e];
    __syncthreads();
    mm(X, true);                             // out = Qx T  (Qx symmetric: Q[k][m] = Qx[m][k])
    if (ty * 4 < X && tx * 4 < Z)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) out[plane + (size_t)(ty * 4 + r) * Z + tx * 4 + c] = acc[r][c];
}

// k3_ty: one workgroup = a slab of 32 columns (flattened (x, z) index) x all Y rows of one simulation:
// out = Qy diag(il) Qy f  -- forward transform along y, division by the eigenvalues, inverse transform, one read / write.
__global__ void __launch_bounds__(256) k3_ty(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ Qy,
                                              const float* __restrict__ il, int Y, int XZ) {
    extern __shared__ __align__(16) float lty[];          // Qy [Y][Y+4] + F [Y][36]
    const int QS = Y + 4;
    float* Q = lty;
    float* Fm = lty + (size_t)Y * QS;
    const int b = blockIdx.y, c0 = blockIdx.x * 32;
    const size_t base = (size_t)b * Y * XZ + c0;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;       // outputs: rows ty*8 .. +7 (two passes of 64 rows at Y = 128), cols tx*2, +1
    const int Y4 = Y >> 2;                   // 16-byte pieces (Y % 16 == 0; rows of Q / Fm are 16-byte aligned: QS, 36 are multiples of 4)
    for (int e = t; e < Y * Y4; e += 256) *reinterpret_cast<float4*>(&Q[(e / Y4) * QS + (e % Y4) * 4]) = reinterpret_cast<const float4*>(Qy)[e];
    for (int e = t; e < Y * 8; e += 256) *reinterpret_cast<float4*>(&Fm[(e >> 3) * 36 + (e & 7) * 4]) = *reinterpret_cast<const float4*>(&in[base + (size_t)(e >> 3) * XZ + (e & 7) * 4]);
    __syncthreads();
    const int RP = (Y + 127) / 128 * 8;      // rows per thread: 8 for Y <= 128
    float acc[8][2];
    for (int pass = 0; pass < 2; ++pass) {
        // pass 0: T = Qy F, scaled by il;  pass 1: out = Qy T
#pragma unroll
        for (int r = 0; r < 8; ++r) { acc[r][0] = 0.f; acc[r][1] = 0.f; }
        const int m0 = ty * RP;
        if (m0 < Y)
            for (int k = 0; k < Y; ++k) {
                const float f0 = Fm[k * 36 + tx * 2], f1 = Fm[k * 36 + tx * 2 + 1];
#pragma unroll
                for (int r = 0; r < 8; ++r) { const float q = Q[k * QS + m0 + r]; acc[r][0] += q * f0; acc[r][1] += q * f1; }   // Qy symmetric
            }
        __syncthreads();
        if (m0 < Y) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int m = m0 + r;
                if (pass == 0) {
                    const float2 s2 = *reinterpret_cast<const float2*>(&il[(size_t)m * XZ + c0 + tx * 2]);
                    Fm[m * 36 + tx * 2] = acc[r][0] * s2.x;
                    Fm[m * 36 + tx * 2 + 1] = acc[r][1] * s2.y;
                } else {
                    *reinterpret_cast<float2*>(&out[base + (size_t)m * XZ + tx * 2]) = make_float2(acc[r][0], acc[r][1]);
                }
            }
        }
        __syncthreads();
    }
}


// ---- the same transforms on the fp32 matrix cores (option k3d_mfma_tf, default) ------------------------------------------------
// k3_ty / k3_tzx above feed every FMA from LDS (10 reads per 16 FMAs): 35.5 / 10.6 us per launch, a tenth of a CU's fp32 rate, and
// six of them are 113 of the 174 us of a 3-D solver step.  v_mfma_f32_32x32x2_f32 has the same peak as the VALU but takes its
// operands as ONE ds_read_b32 each per 4096 FLOP: A[i][k] in lane (i, k) = (lane & 31, lane >> 5), B[k][j] in lane (j, k), D rows
// (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31 (the layout k_conv5x5_bww32 uses).  LDS strides: arrays read with the lane
// along a ROW index (A operands) have stride = 2 (mod 64) words, arrays read with the lane along the column (B operands) stride = 32
// (mod 64): the two k of a step then hit disjoint bank halves -- conflict free.
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// The transform matrices never go through LDS: they are the same for every workgroup (L2 resident), symmetric, and a wave needs the
// same 32 rows (ty) / one 32 x K block (tzx) of them for all its MFMAs -- one coalesced dword per lane and K step, held in registers
// (staging the 64 KB Q_y into LDS was more than half of the first MFMA version of k3_ty: 18.4 us).

// one workgroup = a slab of 32 columns x all Y rows: out = Qy diag(il) Qy f (Y % 32 == 0, Y <= 128; wave w owns row tile w)
template <int YK>                              // YK = Y / 2 K steps (64 at Y = 128)
__global__ void __launch_bounds__(256) k3_ty_mfma(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ Qy,
                                                   const float* __restrict__ il, int XZ) {
    constexpr int Y = 2 * YK;
    __shared__ __align__(16) float Fm[Y * 32], Tm[Y * 32];
    const int b = blockIdx.y, c0 = blockIdx.x * 32;
    const size_t base = (size_t)b * Y * XZ + c0;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 31, kp = lane >> 5;
    const int m0 = wave * 32;
    // A[m][k] = Qy[k][m] (symmetric): lane (m = m0 + c, k = 2 ks + kp) -- a coalesced row segment per k
    float qa[YK];
    if (m0 < Y) {
#pragma unroll
        for (int ks = 0; ks < YK; ++ks) qa[ks] = Qy[(size_t)(2 * ks + kp) * Y + m0 + c];
    }
    for (int e = t; e < Y * 8; e += 256) *reinterpret_cast<float4*>(&Fm[(e >> 3) * 32 + (e & 7) * 4]) = *reinterpret_cast<const float4*>(&in[base + (size_t)(e >> 3) * XZ + (e & 7) * 4]);
    float ilv[16];
    if (m0 < Y) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ilv[r] = il[(size_t)(m0 + (r & 3) + 8 * (r >> 2) + 4 * kp) * XZ + c0 + c];
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        const float* Bm = pass == 0 ? Fm : Tm;
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (m0 < Y) {
            const float* bp = Bm + kp * 32 + c;
#pragma unroll
            for (int ks = 0; ks < YK; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[ks], bp[ks * 64], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kp;
                if (pass == 0) Tm[m * 32 + c] = acc[r] * ilv[r];
                else out[base + (size_t)m * XZ + c] = acc[r];
            }
        }
        __syncthreads();
    }
}

// one workgroup = one (x, z) plane: out = Qx (F Qz)  (X, Z in {32, 64}); wave = (row tile, column tile)
template <int XK, int ZK>                      // K steps of the two products: X / 2, Z / 2
__global__ void __launch_bounds__(256) k3_tzx_mfma(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ Qx,
                                                    const float* __restrict__ Qz) {
    constexpr int X = 2 * XK, Z = 2 * ZK, SA = 66, SB = 96;      // A-type / B-type LDS strides (words)
    __shared__ __align__(16) float Fa[X * SA], Tb[X * SB];
    const size_t plane = (size_t)blockIdx.x * X * Z;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = lane & 31, kp = lane >> 5;
    const int mt = wave & 1, nt = wave >> 1;   // tile (rows 32 mt.., columns 32 nt..)
    const bool have = mt * 32 < X && nt * 32 < Z;
    float qz[ZK], qx[XK];                      // B[k][z] = Qz[k][32 nt + c];  A[m][k] = Qx[k][32 mt + c] (symmetric): coalesced rows
    if (have) {
#pragma unroll
        for (int ks = 0; ks < ZK; ++ks) qz[ks] = Qz[(2 * ks + kp) * Z + nt * 32 + c];
#pragma unroll
        for (int ks = 0; ks < XK; ++ks) qx[ks] = Qx[(2 * ks + kp) * X + mt * 32 + c];
    }
    for (int e = t; e < X * Z / 2; e += 256) {       // 8-byte pieces (rows of SA words are 8-byte aligned)
        const int row = e / (Z / 2), col = (e % (Z / 2)) * 2;
        *reinterpret_cast<float2*>(&Fa[row * SA + col]) = *reinterpret_cast<const float2*>(&in[plane + (size_t)row * Z + col]);
    }
    __syncthreads();
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (have) {                                // T = F Qz: A = F[x][k] (LDS), B = Qz (registers)
        const float* ap = Fa + (mt * 32 + c) * SA + kp;
#pragma unroll
        for (int ks = 0; ks < ZK; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[ks * 2], qz[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) Tb[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kp) * SB + nt * 32 + c] = acc[r];
    }
    __syncthreads();
    if (have) {                                // out = Qx T: A = Qx (registers), B = T[k][z] (LDS)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* bp = Tb + kp * SB + nt * 32 + c;
#pragma unroll
        for (int ks = 0; ks < XK; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qx[ks], bp[ks * 2 * SB], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) out[plane + (size_t)(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kp) * Z + nt * 32 + c] = acc[r];
    }
}

// velocity += scale_c * correction[..., c]  (to_staggered + add, karman_train.py:88-90, 424-426, three components): the
// correction has no value on the last face of each component's own axis
__global__ void __launch_bounds__(256) k3_correct(const float* __restrict__ out, int CO, float s0, float s1, float s2, float* __restrict__ vy,
                                                   float* __restrict__ vx, float* __restrict__ vz, int B, int Y, int X, int Z) {
    const size_t N = (size_t)Y * X * Z, total = (size_t)B * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / N, c = e - b * N;
        const int k = (int)(c % Z), i = (int)((c / Z) % X), j = (int)(c / ((size_t)Z * X));
        const float* o = out + e * CO;
        vy[b * (size_t)(Y + 1) * X * Z + ((size_t)j * X + i) * Z + k] += s0 * o[0];
        vx[b * (size_t)Y * (X + 1) * Z + ((size_t)j * (X + 1) + i) * Z + k] += s1 * o[1];
        vz[b * (size_t)Y * X * (Z + 1) + ((size_t)j * X + i) * (Z + 1) + k] += s2 * o[2];
    }
}

// The two pieces of glue of an unrolled step's REVERSE sweep (they were seven + three torch elementwise launches per step):
//   k3_correct_bwd  G_c += gin_c on every face (gin: the adjoint of step i + 1 w.r.t. its input; NULL for the last step), and the adjoint of
//                   k3_correct, dO [B][Y][X][Z][4] = (s0 Gy, s1 Gx, s2 Gz, 0) at the low faces of every cell
//   k3_feature_bwd  the adjoint of the solver's scaled feature output: G_c[low faces] += f_c dx[..][c], dx [B][Y][X][Z][4]
__global__ void __launch_bounds__(256) k3_correct_bwd(float* __restrict__ gy, float* __restrict__ gx, float* __restrict__ gz, const float* __restrict__ iy,
                                                       const float* __restrict__ ix, const float* __restrict__ iz, float s0, float s1, float s2,
                                                       float4* __restrict__ dO, int B, int Y, int X, int Z) {
    const size_t N = (size_t)Y * X * Z, nb = (size_t)X * Z + (size_t)Y * Z + (size_t)Y * X, per = N + (iy ? nb : 0), total = (size_t)B * per;
    const size_t nVy = (size_t)(Y + 1) * X * Z, nVx = (size_t)Y * (X + 1) * Z, nVz = (size_t)Y * X * (Z + 1);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / per, c = e - b * per;
        if (c < N) {
            const int k = (int)(c % Z), i = (int)((c / Z) % X), j = (int)(c / ((size_t)Z * X));
            const size_t ey = b * nVy + ((size_t)j * X + i) * Z + k, ex = b * nVx + ((size_t)j * (X + 1) + i) * Z + k, ez = b * nVz + ((size_t)j * X + i) * (Z + 1) + k;
            float a0 = gy[ey], a1 = gx[ex], a2 = gz[ez];
            if (iy) { a0 += iy[ey]; a1 += ix[ex]; a2 += iz[ez]; gy[ey] = a0; gx[ex] = a1; gz[ez] = a2; }
            dO[b * N + c] = make_float4(a0 * s0, a1 * s1, a2 * s2, 0.f);
        } else {                                          // the high boundary faces of the three components (no cell has them as a low face)
            size_t q = c - N;
            if (q < (size_t)X * Z) { const size_t f = b * nVy + (size_t)Y * X * Z + q; gy[f] += iy[f]; }
            else if ((q -= (size_t)X * Z) < (size_t)Y * Z) { const size_t f = b * nVx + ((q / Z) * (X + 1) + X) * Z + q % Z; gx[f] += ix[f]; }
            else { q -= (size_t)Y * Z; const size_t f = b * nVz + q * (Z + 1) + Z; gz[f] += iz[f]; }
        }
    }
}
__global__ void __launch_bounds__(256) k3_feature_bwd(const float4* __restrict__ dx, float f0, float f1, float f2, float* __restrict__ gy, float* __restrict__ gx,
                                                       float* __restrict__ gz, int B, int Y, int X, int Z) {
    const size_t N = (size_t)Y * X * Z, total = (size_t)B * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / N, c = e - b * N;
        const int k = (int)(c % Z), i = (int)((c / Z) % X), j = (int)(c / ((size_t)Z * X));
        const float4 d = dx[e];
        gy[b * (size_t)(Y + 1) * X * Z + ((size_t)j * X + i) * Z + k] += f0 * d.x;
        gx[b * (size_t)Y * (X + 1) * Z + ((size_t)j * (X + 1) + i) * Z + k] += f1 * d.y;
        gz[b * (size_t)Y * X * (Z + 1) + ((size_t)j * X + i) * (Z + 1) + k] += f2 * d.z;
    }
}

// y = src (or 0): small fill / copy without memset / memcpy graph nodes (ROCm 7.2 graph memset defect)
__global__ void __launch_bounds__(256) k3_fill(float* __restrict__ y, const float* __restrict__ src, size_t n) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) y[e] = src ? src[e] : 0.f;
}


int grid_for(size_t n) { const size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }

// ---- direct pressure solve: M x = R (M = -A), in place helpers shared by the forward step and its adjoint.  R is modified
// (the capacitance correction is subtracted from it); *res = the buffer (T1 or T2) that holds x.
int pressure_solve3d(hipStream_t s, const sol_karman3d_cfg* c, const int32_t* hdr, float* R, float* T1, float* T2, float** res_out) {
    const int B = c->B, Y = c->Y, X = c->X, Z = c->Z, N = Y * X * Z;
    const int nS = hdr[4], SP = hdr[5];
    const float* Qy = c->direct + FD3_HEADER;
    const float* Qx = Qy + (size_t)Y * Y;
    const float* Qz = Qx + (size_t)X * X;
    const float* il = Qz + (size_t)Z * Z;
    const float* KpT = il + (size_t)N;
    const int* sidx = reinterpret_cast<const int*>(KpT + (size_t)SP * SP);
    // One sine transform of the whole batch along each axis = one batched GEMM:
    //   z: [(b,j,i)] x Z times Qz;   x: per (b, j): Qx times [X x Z];   y: per b: Qy times [Y x (X Z)]
    auto tz = [&](const float* in, float* out) { return sol_gemm_f32(s, 1, in, Z, 0, Qz, Z, 0, out, Z, 0, B * Y * X, Z, Z, 0); };
    auto tx = [&](const float* in, float* out) { return sol_gemm_f32(s, B * Y, Qx, X, 0, in, Z, (long)X * Z, out, Z, (long)X * Z, X, Z, X, 0); };
    auto ty = [&](const float* in, float* out) { return sol_gemm_f32(s, B, Qy, Y, 0, in, X * Z, (long)N, out, X * Z, (long)N, Y, X * Z, Y, 0); };
    auto G = [&](float* src, float* t1, float* t2) -> int {     // t2 = G src  (src is preserved)
        if (int e = tz(src, t1)) return e;
        if (int e = tx(t1, t2)) return e;
        if (int e = ty(t2, t1)) return e;
        SOL_LAUNCH(k3_scale, dim3(grid_for(N), B), dim3(256), 0, s, t1, il, N);
        if (int e = ty(t1, t2)) return e;
        if (int e = tx(t2, t1)) return e;
        return tz(t1, t2);
    };
    // LDS-resident form: three launches per application of G (plane-wise z+x transforms, column-slab y transform + scaling +
    // inverse y transform, plane-wise inverse) instead of six batched GEMMs + a scaling pass
    const bool fused_tf = sol_opt().k3d_fused_tf && X <= 64 && Z <= 64 && Y <= 128 && X % 4 == 0 && Z % 4 == 0 && Y % 16 == 0 && (X * Z) % 32 == 0;
    const size_t ty_lds = ((size_t)Y * (Y + 4) + (size_t)Y * 36) * sizeof(float);
    // the matrix-core kernels are instantiated for the shapes that occur: Y in {128, 64, 32}, X, Z in {64, 32}
    const bool mfma_tf = sol_opt().k3d_mfma_tf && fused_tf && (X == 64 || X == 32) && (Z == 64 || Z == 32) && (Y == 128 || Y == 64 || Y == 32);
    auto Gf = [&](float* src, float* t1, float* t2) -> int {
        static std::atomic<unsigned long long> optin{0};
        if (int e = sol_lds_optin(optin, {SOL_K(k3_ty)}, "k3_ty")) return e;
        if (mfma_tf) {
            auto tzx = [&](const float* in, float* out) {
                if (X == 64 && Z == 64) SOL_LAUNCH((k3_tzx_mfma<32, 32>), dim3(B * Y), dim3(256), 0, s, in, out, Qx, Qz);
                else if (X == 64) SOL_LAUNCH((k3_tzx_mfma<32, 16>), dim3(B * Y), dim3(256), 0, s, in, out, Qx, Qz);
                else if (Z == 64) SOL_LAUNCH((k3_tzx_mfma<16, 32>), dim3(B * Y), dim3(256), 0, s, in, out, Qx, Qz);
                else SOL_LAUNCH((k3_tzx_mfma<16, 16>), dim3(B * Y), dim3(256), 0, s, in, out, Qx, Qz);
            };
            tzx(src, t1);
            if (Y == 128) SOL_LAUNCH(k3_ty_mfma<64>, dim3(X * Z / 32, B), dim3(256), 0, s, (const float*)t1, t2, Qy, il, X * Z);
            else if (Y == 64) SOL_LAUNCH(k3_ty_mfma<32>, dim3(X * Z / 32, B), dim3(256), 0, s, (const float*)t1, t2, Qy, il, X * Z);
            else SOL_LAUNCH(k3_ty_mfma<16>, dim3(X * Z / 32, B), dim3(256), 0, s, (const float*)t1, t2, Qy, il, X * Z);
            tzx(t2, t1);
        } else {
            SOL_LAUNCH(k3_tzx, dim3(B * Y), dim3(256), 0, s, src, t1, Qx, Qz, X, Z);
            SOL_LAUNCH(k3_ty, dim3(X * Z / 32, B), dim3(256), ty_lds, s, t1, t2, Qy, il, Y, X * Z);
            SOL_LAUNCH(k3_tzx, dim3(B * Y), dim3(256), 0, s, t2, t1, Qx, Qz, X, Z);
        }
        SOL_LAUNCH_CHECK();
        return SOL_OK;
    };
    // result in `res`: T2 for the GEMM path, T1 for the fused path
    float* res = fused_tf ? T1 : T2;
    *res_out = res;
    if (int e = fused_tf ? Gf(R, T1, T2) : G(R, T1, T2)) return e;
    if (nS > 0) {
        SOL_REQUIRE(SP / CAPQ <= 1024 && SP % 64 == 0, "direct-solver blob: SP = %d does not fit the capacitance kernel", SP);
        float* cpart = fused_tf ? T2 : T1;          // scratch: the buffer G does not return its result in (B * CAPQ * SP <= B * N floats)
        SOL_LAUNCH(k3_capacitance, dim3(SP / 32, CAPQ, B), dim3(256), 0, s, res, KpT, sidx, cpart, SP, N);
        SOL_LAUNCH(k3_cap_apply, dim3((SP + 255) / 256, B), dim3(256), 0, s, cpart, sidx, R, SP, N);
        if (int e = fused_tf ? Gf(R, T1, T2) : G(R, T1, T2)) return e;
    }
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

// validates the blob header against the configuration (shared by the forward and the adjoint entry points)
int check_blob3d(const sol_karman3d_cfg* c, const int32_t* hdr) {
    SOL_REQUIRE(c->direct && c->direct_n > 0, "the karman-3d step needs the direct-solver blob (cfg.direct, precond3d.direct_solver_blob3d)");
    SOL_REQUIRE(hdr && hdr[0] == FD3_MAGIC, "direct_header_host must be the first 16 words of the blob (host copy)");
    const int Y = c->Y, X = c->X, Z = c->Z;
    const size_t N = (size_t)Y * X * Z;
    const int nS = hdr[4], SP = hdr[5];
    SOL_REQUIRE(hdr[1] == Y && hdr[2] == X && hdr[3] == Z, "direct-solver blob is for a %dx%dx%d grid, cfg is %dx%dx%d", hdr[1], hdr[2], hdr[3], Y, X, Z);
    SOL_REQUIRE(nS >= 0 && SP >= nS && SP % 64 == 0 && SP <= 8192 && (size_t)8 * SP <= N, "direct-solver blob header is inconsistent (nS %d, SP %d)", nS, SP);
    const size_t want = (size_t)FD3_HEADER + (size_t)Y * Y + (size_t)X * X + (size_t)Z * Z + N + (size_t)SP * SP + SP;
    SOL_REQUIRE((size_t)c->direct_n == want, "direct-solver blob has %d words, expected %zu", c->direct_n, want);
    return SOL_OK;
}

// ========================================================================================================================
// Adjoint of the 3-D step with respect to its input velocity (density is a passive tracer: no adjoint).  Stages in reverse:
//   k3b_rhs        q = G^T (mask . g_out)                  (adjoint of  out = v~ - mask . G p)
//   pressure       g_div = M^-1 q                           (second solve with the same symmetric matrix: PhiFlow's custom gradient)
//   k3b_gva        g_a = mask . (g_out + D^T g_div)          (adjoint of the divergence and of the hard-BC face masks)
//   k3b_advect_adj scatter of g_a through the trilinear gathers of the semi-Lagrangian step, for the field term AND the
//                  back-trace (velocity) term, into g_c.  The scatter runs in 64-bit FIXED POINT (global_atomic_add_x2 on
//                  int64 accumulators, scale = a power of two with max|g_a| * scale in [2^37, 2^38): integer addition is
//                  order independent, so the adjoint is reproducible BIT FOR BIT (run twice, compare) --
//                  the scheme of the 2-D kernels (int32 into LDS) with the headroom global memory affords:
//                  2^25 single-contribution range above max|g_a|, resolution 2^-37 max|g_a|.  Bounds, stated: a FINITE contribution
//                  beyond 2^25 max|g_a| saturates in __float2ll_rn (the back-trace term is g_a times a velocity DIFFERENCE of the
//                  saved field times dt/dx: it would take |dv| dt/dx > 3e7, i.e. a simulation that has already blown up); a
//                  NON-FINITE g_a poisons the whole simulation's input gradient (k3b_gva publishes a NaN maximum, k3b_scale).
//   k3b_diffuse_adj g_in = (I + alpha L^T)(g_c . (1 - bcm))  (gather form of the transposed replicate-padded Laplacian; converts
//                  the fixed-point g_c back to fp32 as it reads it)
// ========================================================================================================================
struct K3BArgs {
    int B, Y, X, Z;
    float dtdx, adt;
    int grad_pad;
    const float *re, *active, *bcm;
    long bc_stride;
    const float *svy, *svx, *svz;           // saved post-diffusion velocity
    const float *goy, *gox, *goz;           // gradient w.r.t. the step's output velocity
    float *gay, *gax, *gaz;                 // g_a
    long long *gcy, *gcx, *gcz;             // g_c: int64 fixed-point accumulators (zeroed by k3b_rhs before the scatter)
    unsigned* gmax;                         // [B][K3B_SLOTS] bits of max|g_a| per simulation (zeroed by k3b_rhs, published by k3b_gva)
    float *giy, *gix, *giz;                 // result: gradient w.r.t. the step's input velocity
    float* rhs;
    const float* gdiv;
};

constexpr int K3B_SLOTS = 64;            // absmax slots per simulation (one per lane of the reading wave; same-address atomics serialise in the L2)
constexpr int K3B_FIXBITS = 37;           // max|g_a| * 2^shift lies in [2^37, 2^38)
// power-of-two fixed-point scale of simulation b's scatter and its inverse, from the published max|g_a| (wave-uniform result)
__device__ __forceinline__ void k3b_scale(const unsigned* gmax_b, float& qs, float& qi) {
    const unsigned m = amax_wave_max(gmax_b[threadIdx.x & (K3B_SLOTS - 1)]);
    if (m >= 0x7f800000u) {                   // k3b_gva met an inf / nan gradient in this simulation: nothing is scattered (qs = 0) and the
        qs = 0.f;                             // conversion back (k3b_diffuse_adj: value * qi) makes EVERY input gradient of the simulation NaN --
        qi = __uint_as_float(0x7fc00000u);    // as the fp32 atomics this scheme replaced did, instead of laundering it into finite numbers
        return;
    }
    int e = (int)(m >> 23) - 127;
    e = m == 0u ? 0 : min(max(e, -80), 120);
    qs = __uint_as_float((unsigned)(K3B_FIXBITS - e + 127) << 23);
    qi = __uint_as_float((unsigned)(e - K3B_FIXBITS + 127) << 23);
}

template <int AX>
__device__ __forceinline__ bool boundary_face(int Y, int X, int Z, int j, int i, int k) {
    return AX == 0 ? (j == 0 || j == Y) : (AX == 1 ? (i == 0 || i == X) : (k == 0 || k == Z));
}

__global__ void __launch_bounds__(256) k3b_rhs(K3BArgs a) {
    const int Y = a.Y, X = a.X, Z = a.Z, N = Y * X * Z;
    const int b = blockIdx.y;
    const float* gy = a.goy + (size_t)b * (Y + 1) * X * Z;
    const float* gx = a.gox + (size_t)b * Y * (X + 1) * Z;
    const float* gz = a.goz + (size_t)b * Y * X * (Z + 1);
    const bool keep = a.grad_pad == 1;           // dirichlet0: the boundary faces' gradient depends on p; replicate: it is zero
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < N; c += gridDim.x * blockDim.x) {
        const int k = c % Z, i = (c / Z) % X, j = c / (Z * X);
        auto wy = [&](int jj) { return (keep || (jj != 0 && jj != Y)) ? face_mask<0>(a.active, Y, X, Z, jj, i, k) * gy[((size_t)jj * X + i) * Z + k] : 0.f; };
        auto wx = [&](int ii) { return (keep || (ii != 0 && ii != X)) ? face_mask<1>(a.active, Y, X, Z, j, ii, k) * gx[((size_t)j * (X + 1) + ii) * Z + k] : 0.f; };
        auto wz = [&](int kk) { return (keep || (kk != 0 && kk != Z)) ? face_mask<2>(a.active, Y, X, Z, j, i, kk) * gz[((size_t)j * X + i) * (Z + 1) + kk] : 0.f; };
        a.rhs[(size_t)b * N + c] = (wy(j) - wy(j + 1)) + (wx(i) - wx(i + 1)) + (wz(k) - wz(k + 1));
    }
    // side job: clear this simulation's fixed-point accumulators (the three components are contiguous) and its absmax slots
    const size_t faces = (size_t)(Y + 1) * X * Z + (size_t)Y * (X + 1) * Z + (size_t)Y * X * (Z + 1);
    uint4* zc = reinterpret_cast<uint4*>(a.gcy) + (size_t)b * (faces / 2);     // faces is even for the grids check_blob3d admits (asserted by the host)
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < faces / 2; e += (size_t)gridDim.x * blockDim.x) zc[e] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x < K3B_SLOTS) a.gmax[b * K3B_SLOTS + threadIdx.x] = 0u;
}

__global__ void __launch_bounds__(256) k3b_gva(K3BArgs a) {
    const int Y = a.Y, X = a.X, Z = a.Z, N = Y * X * Z;
    const int nVy = (Y + 1) * X * Z, nVx = Y * (X + 1) * Z, nVz = Y * X * (Z + 1);
    const int b = blockIdx.y;
    const float* P = a.gdiv + (size_t)b * N;
    auto cell = [&](int j, int i, int k) { return ((unsigned)j < (unsigned)Y && (unsigned)i < (unsigned)X && (unsigned)k < (unsigned)Z) ? P[((size_t)j * X + i) * Z + k] : 0.f; };
    float vmax = 0.f;
    bool bad = false;
    // one wave per z column (round 6; it was one thread per face with three runtime divisions each and 6 200 workgroups queueing on the 64
    // absmax slots: 72 us for 6 MB): (component, j, i) are wave uniform, lanes run along the contiguous k
    const int cY = (Y + 1) * X, cX = Y * (X + 1), cC = Y * X;
    const int lane = threadIdx.x & 63;
    const int wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    for (int c = wave0; c < cY + cX + cC; c += nwaves) {
        if (c < cY) {
            const int j = c / X, i = c % X;
            for (int k = lane; k < Z; k += 64) {
                const size_t e = (size_t)b * nVy + ((size_t)j * X + i) * Z + k;
                const float v = face_mask<0>(a.active, Y, X, Z, j, i, k) * (a.goy[e] + cell(j - 1, i, k) - cell(j, i, k));
                a.gay[e] = v; vmax = fmaxf(vmax, fabsf(v)); bad |= !(fabsf(v) <= 3.402823466e38f);      // inf or nan (fmaxf drops a NaN)
            }
        } else if (c < cY + cX) {
            const int q = c - cY, j = q / (X + 1), i = q % (X + 1);
            for (int k = lane; k < Z; k += 64) {
                const size_t e = (size_t)b * nVx + ((size_t)j * (X + 1) + i) * Z + k;
                const float v = face_mask<1>(a.active, Y, X, Z, j, i, k) * (a.gox[e] + cell(j, i - 1, k) - cell(j, i, k));
                a.gax[e] = v; vmax = fmaxf(vmax, fabsf(v)); bad |= !(fabsf(v) <= 3.402823466e38f);
            }
        } else {
            const int q = c - cY - cX, j = q / X, i = q % X;
            for (int k = lane; k <= Z; k += 64) {
                const size_t e = (size_t)b * nVz + ((size_t)j * X + i) * (Z + 1) + k;
                const float v = face_mask<2>(a.active, Y, X, Z, j, i, k) * (a.goz[e] + cell(j, i, k - 1) - cell(j, i, k));
                a.gaz[e] = v; vmax = fmaxf(vmax, fabsf(v)); bad |= !(fabsf(v) <= 3.402823466e38f);
            }
        }
    }
    // max|g_a| of this simulation -> the scale of the fixed-point scatter (one atomic per workgroup).  A non-finite g_a publishes
    // the bits of a NaN -- the largest value the integer maximum can see -- and k3b_scale turns that into "scatter nothing, convert
    // back to NaN": the simulation's input gradient is NaN, not a finite number made of saturated integer conversions.
    __shared__ unsigned red[16];
    const unsigned wmax = amax_wave_max(bad ? 0x7fc00000u : __float_as_uint(vmax));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = wmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned mb = 0u;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) mb = max(mb, red[w]);
        // (thousands of workgroups share the 64 slots of a simulation and same-address atomics serialise in the L2, ~0.3 us apiece:
        //  33 us for this kernel.  A workgroup whose maximum does not exceed what the slot already holds has nothing to publish;
        //  the maximum is order independent, so the filter changes nothing but the number of atomics.)
        unsigned* slot = &a.gmax[b * K3B_SLOTS + (blockIdx.x & (K3B_SLOTS - 1))];
        if (mb > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, mb);
    }
}

// adjoint of one advected face value of component C at (j, i, k): gs = g_a there
// Where a contribution goes.  GAdd: straight into the int64 accumulators in global memory.  TAdd (k3b_advect_adj_tile): into the workgroup's int64
// LDS window when the target face lies inside it, else into global memory -- integer adds commute, so both give the same bits.
// Order-independent accumulation: round to the fixed-point grid FIRST (each contribution on its own), then integer add.
struct GAdd {
    long long *gy, *gx, *gz;
    float qs;
    int X, Z;
    __device__ __forceinline__ void operator()(int comp, int jj, int ii, int kk, float v) const {
        long long* p = comp == 0 ? gy + ((size_t)jj * X + ii) * Z + kk : (comp == 1 ? gx + ((size_t)jj * (X + 1) + ii) * Z + kk : gz + ((size_t)jj * X + ii) * (Z + 1) + kk);
        ::atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__float2ll_rn(v * qs));
    }
};
constexpr int K3B_TJ = 4, K3B_TH = 2, K3B_TW = K3B_TJ + 2 * K3B_TH;      // tile of 4 x 4 columns, window of 8 x 8 columns (halo 2: CFL < 2)
struct TAdd {
    GAdd g;
    unsigned long long* L;         // [3][K3B_TW][K3B_TW][ZP]
    int jw0, iw0, ZP;
    __device__ __forceinline__ void operator()(int comp, int jj, int ii, int kk, float v) const {
        const int lj = jj - jw0, li = ii - iw0;
        if ((unsigned)lj < (unsigned)K3B_TW && (unsigned)li < (unsigned)K3B_TW)
            ::atomicAdd(&L[((comp * K3B_TW + lj) * K3B_TW + li) * ZP + kk], (unsigned long long)__float2ll_rn(v * g.qs));
        else g(comp, jj, ii, kk, v);
    }
};

template <int C, class Add>
__device__ __forceinline__ void advect_adj_point(const K3BArgs& a, const GR& r, const Add& add, int j, int i, int k, float gs) {
    // no fused multiply-adds here: which products the compiler contracts depends on the kernel this is inlined into, and the two scatter
    // kernels (global atomics / LDS window) are held to the SAME BITS by the test suite (the integer accumulation is exact, so the only
    // freedom is in the fp32 contributions themselves)
#pragma clang fp contract(off)
    const int Y = a.Y, X = a.X, Z = a.Z;
    // velocity at the sample point, exactly as the forward pass forms it
    float uy, ux, uz;
    int ja, jb, ia, ib, ka, kb;
    if (C == 0) {
        ja = max(j - 1, 0); jb = min(j, Y - 1);
        uy = r.y(j, i, k);
        ux = 0.25f * (r.x(ja, i, k) + r.x(ja, i + 1, k) + r.x(jb, i, k) + r.x(jb, i + 1, k));
        uz = 0.25f * (r.z(ja, i, k) + r.z(ja, i, k + 1) + r.z(jb, i, k) + r.z(jb, i, k + 1));
    } else if (C == 1) {
        ia = max(i - 1, 0); ib = min(i, X - 1);
        ux = r.x(j, i, k);
        uy = 0.25f * (r.y(j, ia, k) + r.y(j, ib, k) + r.y(j + 1, ia, k) + r.y(j + 1, ib, k));
        uz = 0.25f * (r.z(j, ia, k) + r.z(j, ia, k + 1) + r.z(j, ib, k) + r.z(j, ib, k + 1));
    } else {
        ka = max(k - 1, 0); kb = min(k, Z - 1);
        uz = r.z(j, i, k);
        uy = 0.25f * (r.y(j, i, ka) + r.y(j, i, kb) + r.y(j + 1, i, ka) + r.y(j + 1, i, kb));
        ux = 0.25f * (r.x(j, i, ka) + r.x(j, i, kb) + r.x(j, i + 1, ka) + r.x(j, i + 1, kb));
    }
    const float oy = -uy * a.dtdx, ox = -ux * a.dtdx, oz = -uz * a.dtdx;
    const int n0 = Y + (C == 0), n1 = X + (C == 1), n2 = Z + (C == 2);
    const float fy = floorf(oy), fx = floorf(ox), fz = floorf(oz);
    const float wy = oy - fy, wx = ox - fx, wz = oz - fz;
    const int j0 = clampi(j + (int)fy, 0, n0 - 1), j1 = clampi(j + (int)fy + 1, 0, n0 - 1);
    const int i0 = clampi(i + (int)fx, 0, n1 - 1), i1 = clampi(i + (int)fx + 1, 0, n1 - 1);
    const int k0 = clampi(k + (int)fz, 0, n2 - 1), k1 = clampi(k + (int)fz + 1, 0, n2 - 1);
    auto val = [&](int jj, int ii, int kk) { return C == 0 ? r.y(jj, ii, kk) : (C == 1 ? r.x(jj, ii, kk) : r.z(jj, ii, kk)); };
    float dy = 0.f, dx = 0.f, dz = 0.f;          // d(sample) / d(offset) along each axis
#pragma unroll
    for (int cj = 0; cj < 2; ++cj)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int ck = 0; ck < 2; ++ck) {
                const int jj = cj ? j1 : j0, ii = ci ? i1 : i0, kk = ck ? k1 : k0;
                const float by = cj ? wy : 1.f - wy, bx = ci ? wx : 1.f - wx, bz = ck ? wz : 1.f - wz;
                const float v = val(jj, ii, kk);
                add(C, jj, ii, kk, by * bx * bz * gs);          // field term
                dy += (cj ? 1.f : -1.f) * bx * bz * v;
                dx += by * (ci ? 1.f : -1.f) * bz * v;
                dz += by * bx * (ck ? 1.f : -1.f) * v;
            }
    // back-trace term: offset_a = -dtdx * u_a(x0)
    const float guy = -a.dtdx * gs * dy, gux = -a.dtdx * gs * dx, guz = -a.dtdx * gs * dz;
    if (C == 0) {
        add(0, j, i, k, guy);
        const float qx = 0.25f * gux, qz = 0.25f * guz;
        add(1, ja, i, k, qx); add(1, ja, i + 1, k, qx); add(1, jb, i, k, qx); add(1, jb, i + 1, k, qx);
        add(2, ja, i, k, qz); add(2, ja, i, k + 1, qz); add(2, jb, i, k, qz); add(2, jb, i, k + 1, qz);
    } else if (C == 1) {
        add(1, j, i, k, gux);
        const float qy = 0.25f * guy, qz = 0.25f * guz;
        add(0, j, ia, k, qy); add(0, j, ib, k, qy); add(0, j + 1, ia, k, qy); add(0, j + 1, ib, k, qy);
        add(2, j, ia, k, qz); add(2, j, ia, k + 1, qz); add(2, j, ib, k, qz); add(2, j, ib, k + 1, qz);
    } else {
        add(2, j, i, k, guz);
        const float qy = 0.25f * guy, qx = 0.25f * gux;
        add(0, j, i, ka, qy); add(0, j, i, kb, qy); add(0, j + 1, i, ka, qy); add(0, j + 1, i, kb, qy);
        add(1, j, i, ka, qx); add(1, j, i, kb, qx); add(1, j, i + 1, ka, qx); add(1, j, i + 1, kb, qx);
    }
}

__global__ void __launch_bounds__(256) k3b_advect_adj(K3BArgs a) {
    const int Y = a.Y, X = a.X, Z = a.Z;
    const int nVy = (Y + 1) * X * Z, nVx = Y * (X + 1) * Z, nVz = Y * X * (Z + 1);
    const int b = blockIdx.y;
    GR r;
    r.sy = a.svy + (size_t)b * nVy; r.sx = a.svx + (size_t)b * nVx; r.sz = a.svz + (size_t)b * nVz; r.Y = Y; r.X = X; r.Z = Z;
    long long* gy = a.gcy + (size_t)b * nVy;
    long long* gx = a.gcx + (size_t)b * nVx;
    long long* gz = a.gcz + (size_t)b * nVz;
    float qs, qi;
    k3b_scale(a.gmax + b * K3B_SLOTS, qs, qi);
    const GAdd add{gy, gx, gz, qs, X, Z};
    const int cY = (Y + 1) * X, cX = Y * (X + 1), cC = Y * X;
    const int lane = threadIdx.x & 63;
    const int wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    for (int c = wave0; c < cY + cX + cC; c += nwaves) {
        if (c < cY) {
            const int j = c / X, i = c % X;
            for (int k = lane; k < Z; k += 64) { const float g = a.gay[(size_t)b * nVy + ((size_t)j * X + i) * Z + k]; if (g != 0.f) advect_adj_point<0>(a, r, add, j, i, k, g); }
        } else if (c < cY + cX) {
            const int q = c - cY, j = q / (X + 1), i = q % (X + 1);
            for (int k = lane; k < Z; k += 64) { const float g = a.gax[(size_t)b * nVx + ((size_t)j * (X + 1) + i) * Z + k]; if (g != 0.f) advect_adj_point<1>(a, r, add, j, i, k, g); }
        } else {
            const int q = c - cY - cX, j = q / X, i = q % X;
            for (int k = lane; k <= Z; k += 64) { const float g = a.gaz[(size_t)b * nVz + ((size_t)j * X + i) * (Z + 1) + k]; if (g != 0.f) advect_adj_point<2>(a, r, add, j, i, k, g); }
        }
    }
}

// The same scatter with an LDS window per workgroup (round 6, option k3d_adj_tile): the 17 contributions of a face value -- eight corners of
// its trilinear gather, nine back-trace terms -- land within two cells of it (CFL < 2), so a workgroup that owns the faces of 4 x 4 columns
// (all k) accumulates into an 8 x 8-column int64 window in LDS (3 x 8 x 8 x (Z + 1) x 8 B = 99,840 B at Z = 64) and flushes the non-zero
// cells with ONE global atomic each: 52 K contributions per tile become <= 12.5 K global atomics (the int64 device-scope atomics were the
// whole cost of the kernel: 26.7 M of them in 250 us at 128 x 64 x 64).  Targets beyond the window go to global memory directly.  Integer
// adds commute: the result equals k3b_advect_adj's BIT FOR BIT.
__global__ void __launch_bounds__(256) k3b_advect_adj_tile(K3BArgs a, int nti) {
    extern __shared__ __align__(16) unsigned long long smem_adj[];
    const int Y = a.Y, X = a.X, Z = a.Z, ZP = Z + 1;
    const int nVy = (Y + 1) * X * Z, nVx = Y * (X + 1) * Z, nVz = Y * X * (Z + 1);
    const int b = blockIdx.y;
    const int tj = (int)blockIdx.x / nti, ti = (int)blockIdx.x % nti, j0 = tj * K3B_TJ, i0 = ti * K3B_TJ;
    GR r;
    r.sy = a.svy + (size_t)b * nVy; r.sx = a.svx + (size_t)b * nVx; r.sz = a.svz + (size_t)b * nVz; r.Y = Y; r.X = X; r.Z = Z;
    float qs, qi;
    k3b_scale(a.gmax + b * K3B_SLOTS, qs, qi);
    const GAdd gadd{a.gcy + (size_t)b * nVy, a.gcx + (size_t)b * nVx, a.gcz + (size_t)b * nVz, qs, X, Z};
    const TAdd add{gadd, smem_adj, j0 - K3B_TH, i0 - K3B_TH, ZP};
    const int ncell = 3 * K3B_TW * K3B_TW * ZP;
    for (int e = threadIdx.x; e < ncell; e += 256) smem_adj[e] = 0ull;
    __syncthreads();
    // column tasks: y faces j0 .. j1y-1 (the last tile row also owns face row Y), x faces i0 .. i1x-1 (the last tile column: face column X), cells
    const int j1 = min(j0 + K3B_TJ, Y), i1 = min(i0 + K3B_TJ, X);
    const int j1y = j0 + K3B_TJ >= Y ? Y + 1 : j1, i1x = i0 + K3B_TJ >= X ? X + 1 : i1;
    const int nJ = j1 - j0, nI = i1 - i0, nY = (j1y - j0) * nI, nX = nJ * (i1x - i0), nC = nJ * nI;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int t = wave; t < nY + nX + nC; t += 4) {
        if (t < nY) {
            const int j = j0 + t / nI, i = i0 + t % nI;
            for (int k = lane; k < Z; k += 64) { const float g = a.gay[(size_t)b * nVy + ((size_t)j * X + i) * Z + k]; if (g != 0.f) advect_adj_point<0>(a, r, add, j, i, k, g); }
        } else if (t < nY + nX) {
            const int q = t - nY, w = i1x - i0, j = j0 + q / w, i = i0 + q % w;
            for (int k = lane; k < Z; k += 64) { const float g = a.gax[(size_t)b * nVx + ((size_t)j * (X + 1) + i) * Z + k]; if (g != 0.f) advect_adj_point<1>(a, r, add, j, i, k, g); }
        } else {
            const int q = t - nY - nX, j = j0 + q / nI, i = i0 + q % nI;
            for (int k = lane; k <= Z; k += 64) { const float g = a.gaz[(size_t)b * nVz + ((size_t)j * X + i) * (Z + 1) + k]; if (g != 0.f) advect_adj_point<2>(a, r, add, j, i, k, g); }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < ncell; e += 256) {
        const unsigned long long v = smem_adj[e];
        if (v == 0ull) continue;                                   // (cells outside the arrays never receive a contribution)
        const int kk = e % ZP, li = (e / ZP) % K3B_TW, lj = (e / (ZP * K3B_TW)) % K3B_TW, comp = e / (ZP * K3B_TW * K3B_TW);
        const int jj = j0 - K3B_TH + lj, ii = i0 - K3B_TH + li;
        long long* p = comp == 0 ? gadd.gy + ((size_t)jj * X + ii) * Z + kk : (comp == 1 ? gadd.gx + ((size_t)jj * (X + 1) + ii) * Z + kk : gadd.gz + ((size_t)jj * X + ii) * (Z + 1) + kk);
        ::atomicAdd(reinterpret_cast<unsigned long long*>(p), v);
    }
}

// (I + alpha L^T) g at (j, i, k) of a component array [n0][n1][n2]: the transposed replicate-padded 7-point Laplacian in gather
// form -- a neighbour q - delta inside the array contributes g there, a direction that leaves the array contributes g[q] itself
__device__ __forceinline__ float lapT7(const long long* g, float qi, float sc_here, const float* scm, int n0, int n1, int n2, int j, int i, int k) {
    const size_t s0 = (size_t)n1 * n2, s1 = n2;
    const size_t c = (size_t)j * s0 + (size_t)i * s1 + k;
    auto at = [&](size_t q) { const float v = __ll2float_rn(g[q]) * qi; return scm ? v * (1.f - scm[q]) : v; };
    const float v = __ll2float_rn(g[c]) * qi * sc_here;
    float acc = -6.f * v;
    acc += j + 1 < n0 ? at(c + s0) : v;
    acc += j > 0 ? at(c - s0) : v;
    acc += i + 1 < n1 ? at(c + s1) : v;
    acc += i > 0 ? at(c - s1) : v;
    acc += k + 1 < n2 ? at(c + 1) : v;
    acc += k > 0 ? at(c - 1) : v;
    return acc;
}

__global__ void __launch_bounds__(256) k3b_diffuse_adj(K3BArgs a) {
    const int Y = a.Y, X = a.X, Z = a.Z;
    const int nVy = (Y + 1) * X * Z, nVx = Y * (X + 1) * Z, nVz = Y * X * (Z + 1);
    const int b = blockIdx.y;
    const float alpha = a.adt / a.re[b];
    float qs, qi;
    k3b_scale(a.gmax + b * K3B_SLOTS, qs, qi);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nVy + nVx + nVz; e += gridDim.x * blockDim.x) {
        if (e < nVy) {
            const int k = e % Z, i = (e / Z) % X, j = e / (Z * X);
            const long long* g = a.gcy + (size_t)b * nVy;
            const float* m = a.bcm + (size_t)b * a.bc_stride;          // g' = g . (1 - bcm): the BC blend's adjoint
            const float sc = 1.f - m[e];
            a.giy[(size_t)b * nVy + e] = __ll2float_rn(g[e]) * qi * sc + alpha * lapT7(g, qi, sc, m, Y + 1, X, Z, j, i, k);
        } else if (e < nVy + nVx) {
            const int q = e - nVy, k = q % Z, i = (q / Z) % (X + 1), j = q / (Z * (X + 1));
            const long long* g = a.gcx + (size_t)b * nVx;
            a.gix[(size_t)b * nVx + q] = __ll2float_rn(g[q]) * qi + alpha * lapT7(g, qi, 1.f, nullptr, Y, X + 1, Z, j, i, k);
        } else {
            const int q = e - nVy - nVx, k = q % (Z + 1), i = (q / (Z + 1)) % X, j = q / ((Z + 1) * X);
            const long long* g = a.gcz + (size_t)b * nVz;
            a.giz[(size_t)b * nVz + q] = __ll2float_rn(g[q]) * qi + alpha * lapT7(g, qi, 1.f, nullptr, Y, X, Z + 1, j, i, k);
        }
    }
}

}  // namespace

extern "C" int32_t sol_abi_size_karman3d(void) { return (int32_t)sizeof(sol_karman3d_cfg); }

extern "C" size_t sol_karman3d_step_workspace_bytes(const sol_karman3d_cfg* c) {
    if (!c) return 0;
    const size_t B = c->B, Y = c->Y, X = c->X, Z = c->Z;
    // three diffused components + rhs + two transform buffers
    const size_t floats = B * ((Y + 1) * X * Z + Y * (X + 1) * Z + Y * X * (Z + 1) + 3 * Y * X * Z) + 256;
    return floats * sizeof(float);
}

extern "C" int sol_karman3d_step_fwd(const sol_karman3d_cfg* c, void* stream,
                                     const float* d_in, const float* vy_in, const float* vx_in, const float* vz_in,
                                     const float* re, const float* active, const float* inflow,
                                     const float* velBCy, const float* velBCyMask, int64_t bc_batch_stride,
                                     float* d_out, float* vy_out, float* vx_out, float* vz_out,
                                     float* saved_vy, float* saved_vx, float* saved_vz,
                                     float* feat_out, const float* feat_scale, const int32_t* direct_header_host,
                                     void* workspace, size_t workspace_bytes) {
    SOL_REQUIRE(c != nullptr, "cfg is NULL");
    SOL_REQUIRE((saved_vy && saved_vx && saved_vz) || (!saved_vy && !saved_vx && !saved_vz), "saved_vy / saved_vx / saved_vz: all three or none");
    SOL_REQUIRE(c->B >= 1 && c->Y >= 8 && c->X >= 8 && c->Z >= 8 && c->B <= 65535, "sol_karman3d_step_fwd: B >= 1, Y, X, Z >= 8 (got %d, %d, %d, %d)", c->B, c->Y, c->X, c->Z);
    SOL_REQUIRE((size_t)(c->Y + 1) * (c->X + 1) * (c->Z + 1) * 3 < ((size_t)1 << 31), "sol_karman3d_step_fwd: grid too large for 32-bit face indices");
    SOL_REQUIRE(vy_in && vx_in && vz_in && re && active && velBCy && velBCyMask && vy_out && vx_out && vz_out && workspace, "sol_karman3d_step_fwd: NULL pointer argument");
    SOL_REQUIRE((d_in && inflow) || !d_out, "density output requested without d_in / inflow");
    SOL_REQUIRE(!feat_out || feat_scale, "feat_out requires feat_scale");
    if (int e = check_blob3d(c, direct_header_host)) return e;
    SOL_REQUIRE(workspace_bytes >= sol_karman3d_step_workspace_bytes(c), "workspace too small");
    SOL_REQUIRE(vy_in != vy_out && vx_in != vx_out && vz_in != vz_out && (d_in != d_out || !d_out), "sol_karman3d_step_fwd: outputs must not alias the inputs");
    const int B = c->B, Y = c->Y, X = c->X, Z = c->Z, N = Y * X * Z;
    hipStream_t s = (hipStream_t)stream;
    float* w = static_cast<float*>(workspace);
    const size_t nVy = (size_t)(Y + 1) * X * Z, nVx = (size_t)Y * (X + 1) * Z, nVz = (size_t)Y * X * (Z + 1);
    float* svy = w; w += B * nVy;
    float* svx = w; w += B * nVx;
    float* svz = w; w += B * nVz;
    float* R = w; w += (size_t)B * N;
    float* T1 = w; w += (size_t)B * N;
    float* T2 = w; w += (size_t)B * N;
    if (saved_vy) { svy = saved_vy; svx = saved_vx; svz = saved_vz; }     // training: the post-diffusion velocity is the only state the adjoint needs

    K3Args a{};
    a.B = B; a.Y = Y; a.X = X; a.Z = Z; a.dtdx = c->dt / c->dx; a.dt = c->dt; a.adt = c->dt * c->res * c->res;
    a.grad_pad = c->grad_pad; a.inflow_before = c->inflow_before;
    a.d_in = d_in; a.vy_in = vy_in; a.vx_in = vx_in; a.vz_in = vz_in; a.re = re; a.active = active; a.inflow = inflow;
    a.bcv = velBCy; a.bcm = velBCyMask; a.bc_stride = bc_batch_stride;
    a.d_out = d_out; a.vy_out = vy_out; a.vx_out = vx_out; a.vz_out = vz_out; a.svy = svy; a.svx = svx; a.svz = svz; a.rhs = R; a.feat = feat_out; a.p = T2;
    if (feat_scale) { a.fs0 = feat_scale[0]; a.fs1 = feat_scale[1]; a.fs2 = feat_scale[2]; a.fs3 = feat_scale[3]; }
    const size_t faces = nVy + nVx + nVz;
    SOL_LAUNCH(k3_diffuse, dim3(grid_for(faces), B), dim3(256), 0, s, a);
    const size_t tile_lds = ((size_t)(TR + 1) * TR * Z + (size_t)TR * (TR + 1) * Z + (size_t)TR * TR * (Z + 1)) * sizeof(float) + (size_t)TR * TR * Z;
    if (sol_opt().k3d_tile && tile_lds <= 160 * 1024) {
        static std::atomic<unsigned long long> optin{0};
        if (int e = sol_lds_optin(optin, {SOL_K(k3_advect_tile)}, "k3_advect_tile")) return e;
        const int tiles_y = (Y + T3 - 1) / T3, tiles_x = (X + T3 - 1) / T3;
        SOL_LAUNCH(k3_advect_tile, dim3(tiles_y * tiles_x, B), dim3(ADV_T), tile_lds, s, a, tiles_x);
    } else {
        SOL_LAUNCH(k3_advect, dim3(grid_for((size_t)((Y + 1) * X + Y * (X + 1) + 2 * Y * X) * 64), B), dim3(256), 0, s, a);
    }
    SOL_LAUNCH(k3_div, dim3(grid_for(N), B), dim3(256), 0, s, a);
    SOL_LAUNCH_CHECK();

    float* res = nullptr;
    if (int e = pressure_solve3d(s, c, direct_header_host, R, T1, T2, &res)) return e;
    a.p = res;
    SOL_LAUNCH(k3_project, dim3(grid_for(faces), B), dim3(256), 0, s, a);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

extern "C" size_t sol_karman3d_step_bwd_workspace_bytes(const sol_karman3d_cfg* c) {
    if (!c) return 0;
    const size_t B = c->B, Y = c->Y, X = c->X, Z = c->Z;
    // g_a (fp32) and g_c (int64 fixed point), three components each + rhs + two transform buffers + the absmax slots
    const size_t floats = B * (3 * ((Y + 1) * X * Z + Y * (X + 1) * Z + Y * X * (Z + 1)) + 3 * Y * X * Z + K3B_SLOTS) + 256;
    return floats * sizeof(float);
}

extern "C" int sol_karman3d_step_bwd(const sol_karman3d_cfg* c, void* stream,
                                     const float* saved_vy, const float* saved_vx, const float* saved_vz,
                                     const float* re, const float* active, const float* velBCyMask, int64_t bc_batch_stride,
                                     const float* g_vy_out, const float* g_vx_out, const float* g_vz_out,
                                     float* g_vy_in, float* g_vx_in, float* g_vz_in,
                                     const int32_t* direct_header_host, void* workspace, size_t workspace_bytes) {
    SOL_REQUIRE(c != nullptr, "cfg is NULL");
    SOL_REQUIRE(c->B >= 1 && c->Y >= 8 && c->X >= 8 && c->Z >= 8 && c->B <= 65535, "sol_karman3d_step_bwd: B >= 1, Y, X, Z >= 8");
    SOL_REQUIRE(saved_vy && saved_vx && saved_vz && re && active && velBCyMask && g_vy_out && g_vx_out && g_vz_out && g_vy_in && g_vx_in && g_vz_in && workspace,
                "sol_karman3d_step_bwd: NULL pointer argument");
    if (int e = check_blob3d(c, direct_header_host)) return e;
    SOL_REQUIRE(workspace_bytes >= sol_karman3d_step_bwd_workspace_bytes(c), "workspace too small");
    const int B = c->B, Y = c->Y, X = c->X, Z = c->Z, N = Y * X * Z;
    hipStream_t s = (hipStream_t)stream;
    const size_t nVy = (size_t)(Y + 1) * X * Z, nVx = (size_t)Y * (X + 1) * Z, nVz = (size_t)Y * X * (Z + 1), faces = nVy + nVx + nVz;
    float* w = static_cast<float*>(workspace);
    K3BArgs a{};
    a.B = B; a.Y = Y; a.X = X; a.Z = Z; a.dtdx = c->dt / c->dx; a.adt = c->dt * c->res * c->res; a.grad_pad = c->grad_pad;
    a.re = re; a.active = active; a.bcm = velBCyMask; a.bc_stride = bc_batch_stride;
    a.svy = saved_vy; a.svx = saved_vx; a.svz = saved_vz; a.goy = g_vy_out; a.gox = g_vx_out; a.goz = g_vz_out;
    SOL_REQUIRE(faces % 2 == 0, "sol_karman3d_step_bwd: odd face count (%zu): even X, Z expected", faces);
    // per-simulation layout of g_c: [b][y faces | x faces | z faces] int64 -- k3b_rhs clears simulation b's block
    SOL_REQUIRE(reinterpret_cast<uintptr_t>(workspace) % 16 == 0, "sol_karman3d_step_bwd: workspace must be 16-byte aligned");
    long long* gc = reinterpret_cast<long long*>(w); w += 2 * B * faces;
    a.gcy = gc; a.gcx = gc + B * nVy; a.gcz = a.gcx + B * nVx;
    a.gay = w; w += B * nVy; a.gax = w; w += B * nVx; a.gaz = w; w += B * nVz;
    a.gmax = reinterpret_cast<unsigned*>(w); w += (size_t)B * K3B_SLOTS;
    float* R = w; w += (size_t)B * N;
    float* T1 = w; w += (size_t)B * N;
    float* T2 = w; w += (size_t)B * N;
    a.rhs = R; a.giy = g_vy_in; a.gix = g_vx_in; a.giz = g_vz_in;
    SOL_LAUNCH(k3b_rhs, dim3(grid_for(N), B), dim3(256), 0, s, a);
    SOL_LAUNCH_CHECK();
    float* res = nullptr;
    if (int e = pressure_solve3d(s, c, direct_header_host, R, T1, T2, &res)) return e;
    a.gdiv = res;
    {   // wave per column, a few columns per wave: 1 024 workgroups at most publish into the 64 absmax slots
        const size_t cols = (size_t)(Y + 1) * X + (size_t)Y * (X + 1) + (size_t)Y * X;
        const unsigned g_gva = (unsigned)std::min<size_t>(1024, (cols + 3) / 4);
        SOL_LAUNCH(k3b_gva, dim3(g_gva, B), dim3(256), 0, s, a);
    }
    if (sol_opt().k3d_adj_tile && Z <= 64) {
        static std::atomic<unsigned long long> optin_adj{0};
        if (int e = sol_lds_optin(optin_adj, {SOL_K(k3b_advect_adj_tile)}, "k3b_advect_adj_tile")) return e;
        const int ntj = (Y + K3B_TJ - 1) / K3B_TJ, nti = (X + K3B_TJ - 1) / K3B_TJ;
        SOL_LAUNCH(k3b_advect_adj_tile, dim3(ntj * nti, B), dim3(256), (size_t)3 * K3B_TW * K3B_TW * (Z + 1) * 8, s, a, nti);
    } else
    SOL_LAUNCH(k3b_advect_adj, dim3(grid_for((size_t)((Y + 1) * X + Y * (X + 1) + Y * X) * 64), B), dim3(256), 0, s, a);
    SOL_LAUNCH(k3b_diffuse_adj, dim3(grid_for(faces), B), dim3(256), 0, s, a);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

extern "C" int sol_karman3d_correct(void* stream, const float* out, int32_t cout, float s0, float s1, float s2,
                                    float* vy, float* vx, float* vz, int32_t B, int32_t Y, int32_t X, int32_t Z) {
    SOL_REQUIRE(out && vy && vx && vz && cout >= 3 && B >= 1 && Y >= 1 && X >= 1 && Z >= 1, "sol_karman3d_correct: bad arguments");
    SOL_LAUNCH(k3_correct, dim3(grid_for((size_t)B * Y * X * Z)), dim3(256), 0, (hipStream_t)stream, out, cout, s0, s1, s2, vy, vx, vz, B, Y, X, Z);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

extern "C" int sol_karman3d_correct_bwd(void* stream, float* g_vy, float* g_vx, float* g_vz, const float* gin_vy, const float* gin_vx, const float* gin_vz,
                                        float s0, float s1, float s2, float* d_out4, int32_t B, int32_t Y, int32_t X, int32_t Z) {
    SOL_REQUIRE(g_vy && g_vx && g_vz && d_out4 && B >= 1 && Y >= 1 && X >= 1 && Z >= 1, "sol_karman3d_correct_bwd: bad arguments");
    SOL_REQUIRE((gin_vy != nullptr) == (gin_vx != nullptr) && (gin_vy != nullptr) == (gin_vz != nullptr), "sol_karman3d_correct_bwd: the three gin pointers are all NULL or all set");
    SOL_REQUIRE(reinterpret_cast<uintptr_t>(d_out4) % 16 == 0, "sol_karman3d_correct_bwd: d_out4 must be 16-byte aligned");
    SOL_LAUNCH(k3_correct_bwd, dim3(grid_for((size_t)B * ((size_t)Y * X * Z + (gin_vy ? (size_t)X * Z + (size_t)Y * Z + (size_t)Y * X : 0)))), dim3(256), 0, (hipStream_t)stream,
               g_vy, g_vx, g_vz, gin_vy, gin_vx, gin_vz, s0, s1, s2, reinterpret_cast<float4*>(d_out4), B, Y, X, Z);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

extern "C" int sol_karman3d_feature_bwd(void* stream, const float* dx4, float f0, float f1, float f2, float* g_vy, float* g_vx, float* g_vz,
                                        int32_t B, int32_t Y, int32_t X, int32_t Z) {
    SOL_REQUIRE(dx4 && g_vy && g_vx && g_vz && B >= 1 && Y >= 1 && X >= 1 && Z >= 1, "sol_karman3d_feature_bwd: bad arguments");
    SOL_REQUIRE(reinterpret_cast<uintptr_t>(dx4) % 16 == 0, "sol_karman3d_feature_bwd: dx4 must be 16-byte aligned");
    SOL_LAUNCH(k3_feature_bwd, dim3(grid_for((size_t)B * Y * X * Z)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(dx4), f0, f1, f2, g_vy, g_vx, g_vz, B, Y, X, Z);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// 5 x 5 x 5 SAME convolution, NDHWC (D = y, H = x, W = z), as FIVE passes of the 2-D kernels of conv5x5(_sb).hip over the
// (x, z) planes: y[d] = sum_kd conv2d(x[d + kd - 2], w[kd]).  The running sum lives in y (fp32, read back as the 2-D
// kernels' residual operand); the centre slice runs LAST over all B*D planes and carries bias, activation and the absmax
// publish.  Same arithmetic per product as the 2-D layers (split fp16 / bf16 MFMA or fp32 MFMA, option conv_precision).
// ------------------------------------------------------------------------------------------------------------------------
static bool conv3d_fusable_shape(int cin, int cout) { return cin == 32 && (cout == 32 || cout <= 16); }
// five 2-D packed buffers (one per depth slice) + for 32 -> 32 and 32 -> (<= 16) layers the one-launch kernel's weight section (conv3d_sb.hip)
extern "C" size_t sol_conv3d_packed_floats(int32_t cin, int32_t cout) {
    return 5 * align_up(sol_conv5x5_packed_floats(cin, cout, SOL_CONV_FWD), 64) + (conv3d_fusable_shape(cin, cout) ? sol_conv3d_sh_packed_floats(cout) : 0);
}

extern "C" int sol_conv3d_pack(void* stream, const float* w_dhwio, int32_t cin, int32_t cout, int32_t mode, float* packed) {
    SOL_REQUIRE(w_dhwio && packed, "sol_conv3d_pack: NULL pointer");
    SOL_REQUIRE(mode == SOL_CONV_FWD || mode == SOL_CONV_BWD_DATA, "sol_conv3d_pack: bad mode %d", mode);
    const size_t per = align_up(sol_conv5x5_packed_floats(cin, cout, SOL_CONV_FWD), 64);
    // backward-data: the flipped kernel -- slice kd of the run convolution is the forward slice 4 - kd, its 2-D taps flipped
    // and its channel axes swapped by the 2-D packer's SOL_CONV_BWD_DATA mode.  The five slices are ONE launch (k_pack_jobs: the same
    // sections as sol_conv5x5_pack writes, which took three launches per slice -- 330 launches of a karman-3d training step)
    const float* ws[5]; float* outs[5]; int32_t ci[5], co[5], md[5];
    for (int kd = 0; kd < 5; ++kd) {
        const int src = mode == SOL_CONV_FWD ? kd : 4 - kd;
        ws[kd] = w_dhwio + (size_t)src * 25 * cin * cout; outs[kd] = packed + kd * per; ci[kd] = cin; co[kd] = cout; md[kd] = mode;
    }
    if (int e = sol_conv5x5_pack_jobs(stream, 5, ws, ci, co, md, outs)) return e;
    if (conv3d_fusable_shape(cin, cout)) return sol_conv3d_sh_pack((hipStream_t)stream, w_dhwio, mode, cout, packed + 5 * per);
    return SOL_OK;
}

extern "C" int sol_conv3d(void* stream, const float* x, const float* packed, const float* bias, const float* residual, const float* act_ref, float* y,
                          int32_t B, int32_t D, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t epilogue, float slope,
                          const uint32_t* x_absmax, uint32_t* y_absmax) {
    SOL_REQUIRE(x && packed && y && x != y, "sol_conv3d: NULL pointer / in-place call");
    SOL_REQUIRE(B >= 1 && D >= 3, "sol_conv3d: B >= 1, D >= 3 (got %d, %d)", B, D);
    SOL_REQUIRE(epilogue == SOL_EPI_NONE || epilogue == SOL_EPI_LRELU || epilogue == SOL_EPI_DLRELU, "sol_conv3d: epilogue must be SOL_EPI_NONE, SOL_EPI_LRELU or SOL_EPI_DLRELU");
    SOL_REQUIRE((epilogue == SOL_EPI_DLRELU) == (act_ref != nullptr) && act_ref != y, "sol_conv3d: SOL_EPI_DLRELU needs an activation reference (and only it does), distinct from y");
    hipStream_t s = (hipStream_t)stream;
    const int cin_k = cin <= 4 ? 4 : 32;
    SOL_REQUIRE(cin == cin_k, "sol_conv3d: input channels must be 4 (zero padded) or 32 (got %d)", cin);
    const size_t per = align_up(sol_conv5x5_packed_floats(cin, cout, SOL_CONV_FWD), 64);
    if (conv3d_fusable_shape(cin, cout) && W == 64 && x_absmax && sol_opt().conv_precision == 0 && sol_opt().k3d_conv_fused && residual != y)
        return sol_conv3d_sb_launch(s, x, packed + 5 * per, bias, residual, act_ref, y, B, D, H, cout, epilogue, slope, x_absmax, y_absmax);
    const size_t pin = (size_t)H * W * cin, pout = (size_t)H * W * cout;      // floats per plane
    for (int b = 0; b < B; ++b) {
        const float* xb = x + (size_t)b * D * pin;
        float* yb = y + (size_t)b * D * pout;
        const float* rb = residual ? residual + (size_t)b * D * pout : nullptr;
        // planes 0, 1 receive nothing from slice kd = 0: they start from the residual (or zero)
        SOL_LAUNCH(k3_fill, dim3(grid_for(2 * pout)), dim3(256), 0, s, yb, rb, 2 * pout);
        SOL_LAUNCH_CHECK();
        // kd = 0: y[2:D] = conv(x[0:D-2], w0) (+ residual[2:D])
        if (int e = sol_conv5x5_scaled(stream, xb, packed, nullptr, rb ? rb + 2 * pout : nullptr, nullptr, yb + 2 * pout, D - 2, H, W, cin, cout, SOL_EPI_NONE, slope, x_absmax, nullptr)) return e;
        // kd = 1: y[1:D] += conv(x[0:D-1], w1)
        if (int e = sol_conv5x5_scaled(stream, xb, packed + per, nullptr, yb + pout, nullptr, yb + pout, D - 1, H, W, cin, cout, SOL_EPI_NONE, slope, x_absmax, nullptr)) return e;
        // kd = 3: y[0:D-1] += conv(x[1:D], w3)
        if (int e = sol_conv5x5_scaled(stream, xb + pin, packed + 3 * per, nullptr, yb, nullptr, yb, D - 1, H, W, cin, cout, SOL_EPI_NONE, slope, x_absmax, nullptr)) return e;
        // kd = 4: y[0:D-2] += conv(x[2:D], w4)
        if (int e = sol_conv5x5_scaled(stream, xb + 2 * pin, packed + 4 * per, nullptr, yb, nullptr, yb, D - 2, H, W, cin, cout, SOL_EPI_NONE, slope, x_absmax, nullptr)) return e;
    }
    // kd = 2 (centre): every plane of the batch, with bias / activation / absmax publish
    return sol_conv5x5_scaled(stream, x, packed + 2 * per, bias, y, act_ref, y, B * D, H, W, cin, cout, epilogue, slope, x_absmax, y_absmax);
}

// ------------------------------------------------------------------------------------------------------------------------
// Thin-INPUT Conv3D (<= 4 -> 32 channels: the network's first layer 4 -> 32 and the output layer's data gradient 3 -> 32) with the
// DEPTH taps packed into the channel axis.  The five-pass form above runs these layers on the fp32 matrix pipe (k_conv5x5<4, 2>: K = 4
// channels per tap, 5 x 43 us at 128 x 64 x 64 = 78 TF, the rate that pipe sustains) and moves the 67 MB result through HBM five times.
// Here x[d][h][w][0:4] is first gathered into x'[d][h][w][4 s + c] = x[d + s - 2][h][w][c] (s = 0..4; 5..7 and planes outside the volume
// are zero: the SAME padding of the depth axis) -- "im2col along depth", 32 channels -- and the layer becomes ONE 2-D 5 x 5 convolution
// 32 -> 32 over the (H, W) planes with w'[dy][dx][4 s + c][co] = w[s][dy][dx][c][co]: the dx-major split-fp16 kernel of the 32 -> 32
// layers (conv5x5_dx.hip), 25 taps of K = 32 (20 of them in use) instead of 125 taps of K = 4, bias / activation / absmax publish in
// its epilogue, the result written once.  Same per-product arithmetic as every other 32 -> 32 layer of the network (option conv_precision).
// ------------------------------------------------------------------------------------------------------------------------
namespace {
// dir = +1: out[d][..][4 s + c] = x[d + s - 2][..][c] (an INPUT gathered for the taps that read it); dir = -1: out[d][..][4 s + c] = x[d - s + 2][..][c]
// (an OUTPUT gradient gathered for the weight gradient of a thin-OUTPUT layer: slot s pairs plane d of the wide input with dz of plane d - s + 2)
__global__ void __launch_bounds__(256) k3_kpack_depth(const float4* __restrict__ x, float4* __restrict__ out, int D, int HW, size_t total, int dir) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int s = (int)(e & 7);
        const size_t q = e >> 3, plane = q / HW;
        const int p = (int)(q - plane * HW), off = dir * (s - 2), ds = (int)(plane % D) + off;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < 5 && ds >= 0 && ds < D) v = x[(size_t)((long)plane + off) * HW + p];
        out[e] = v;
    }
}
// w' [5][5][32][32] (HWIO of the 2-D convolution that is RUN) from the FORWARD kernel w:
//   mode SOL_CONV_FWD:      w [5][5][5][cin][32]:  w'[dy][dx][4 s + c][co] = w[s][dy][dx][c][co]
//   mode SOL_CONV_BWD_DATA: w [5][5][5][32][cr], the run convolution reads the cr channels of dy and writes 32: the flipped kernel with
//                           its channel axes swapped, w'[dy][dx][4 s + c][co] = w[4 - s][4 - dy][4 - dx][co][c]
__global__ void __launch_bounds__(256) k3_kpack_weights(const float* __restrict__ w, float* __restrict__ w2, int cin, int mode) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 25 * 32 * 32) return;
    const int co = e & 31, k = (e >> 5) & 31, tap = e >> 10, dy = tap / 5, dx = tap % 5, s = k >> 2, c = k & 3;
    float v = 0.f;
    if (s < 5 && c < cin)
        v = mode == SOL_CONV_FWD ? w[((((size_t)s * 5 + dy) * 5 + dx) * cin + c) * 32 + co]
                                 : w[((((size_t)(4 - s) * 5 + (4 - dy)) * 5 + (4 - dx)) * 32 + co) * cin + c];
    w2[e] = v;
}
}  // namespace

extern "C" size_t sol_conv3d_thin_packed_floats(void) { return align_up(sol_conv5x5_packed_floats(32, 32, SOL_CONV_FWD), 64) + 25 * 32 * 32; }
extern "C" size_t sol_conv3d_thin_ws_floats(int32_t B, int32_t D, int32_t H, int32_t W) { return (size_t)B * D * H * W * 32 + 512; }      // gathered tensor + absmax slots of x (and of dz: weight gradient)

extern "C" int sol_conv3d_thin_pack(void* stream, const float* w_dhwio, int32_t cin, int32_t mode, float* packed) {
    SOL_REQUIRE(w_dhwio && packed, "sol_conv3d_thin_pack: NULL pointer");
    SOL_REQUIRE(mode == SOL_CONV_FWD || mode == SOL_CONV_BWD_DATA, "sol_conv3d_thin_pack: bad mode %d", mode);
    SOL_REQUIRE(cin >= 1 && cin <= 4, "sol_conv3d_thin_pack: 1..4 input channels of the convolution that is run (got %d)", cin);
    float* w2 = packed + align_up(sol_conv5x5_packed_floats(32, 32, SOL_CONV_FWD), 64);
    SOL_LAUNCH(k3_kpack_weights, dim3(100), dim3(256), 0, (hipStream_t)stream, w_dhwio, w2, cin, mode);
    SOL_LAUNCH_CHECK();
    return sol_conv5x5_pack(stream, w2, 32, 32, SOL_CONV_FWD, packed);
}

extern "C" int sol_conv3d_thin(void* stream, const float* x, const float* packed, const float* bias, const float* act_ref, float* y, float* ws,
                               int32_t B, int32_t D, int32_t H, int32_t W, int32_t epilogue, float slope, uint32_t* y_absmax) {
    SOL_REQUIRE(x && packed && y && ws && x != y, "sol_conv3d_thin: NULL pointer / in-place call");
    SOL_REQUIRE(B >= 1 && D >= 1 && H >= 1 && W >= 4 && W % 4 == 0, "sol_conv3d_thin: bad shape (B %d, D %d, H %d, W %d)", B, D, H, W);
    SOL_REQUIRE(epilogue == SOL_EPI_NONE || epilogue == SOL_EPI_LRELU || epilogue == SOL_EPI_DLRELU, "sol_conv3d_thin: epilogue must be SOL_EPI_NONE, SOL_EPI_LRELU or SOL_EPI_DLRELU");
    SOL_REQUIRE((epilogue == SOL_EPI_DLRELU) == (act_ref != nullptr) && act_ref != y, "sol_conv3d_thin: SOL_EPI_DLRELU needs an activation reference (and only it does), distinct from y");
    const size_t npx = (size_t)B * D * H * W;
    uint32_t* slots = reinterpret_cast<uint32_t*>(ws + npx * 32);
    if (int e = sol_absmax(stream, x, (int64_t)(npx * 4), slots)) return e;
    SOL_LAUNCH(k3_kpack_depth, dim3(grid_for(npx * 8)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(ws), D, H * W, npx * 8, 1);
    SOL_LAUNCH_CHECK();
    return sol_conv5x5_scaled(stream, ws, packed, bias, nullptr, act_ref, y, B * D, H, W, 32, 32, epilogue, slope, slots, y_absmax);
}

// ------------------------------------------------------------------------------------------------------------------------
// Thin-OUTPUT layers (32 -> cr <= 4 channels: the network's output layer, and the first layer's data gradient) with the depth taps packed into
// the OUTPUT channel axis: y'[q][..][4 s + c] = sum_(dy, dx, ci) x[q][.. + (dy, dx)][ci] w_run[s][dy][dx][ci][c] is ONE 2-D 32 -> 32 convolution
// over the (H, W) planes (20 of its 32 output channels in use), and y[d][..][c] = bias[c] + sum_s y'[d + s - 2][..][4 s + c] a gather over five
// planes.  Replaces k_conv3d_sb8<1, 0> for these layers (180 us at 128 x 64 x 64: the eight-row kernel's staging skeleton for half a
// channel tile of which 3 / 4 of 16 columns are real).
// ------------------------------------------------------------------------------------------------------------------------
namespace {
//   mode SOL_CONV_FWD:      w [5][5][5][32][cr]:  w'[dy][dx][ci][4 s + c] = w[s][dy][dx][ci][c]
//   mode SOL_CONV_BWD_DATA: w [5][5][5][cr][32] (the FORWARD kernel of a cr -> 32 layer); the run convolution reads the 32 channels of dz and
//                           writes cr: the flipped kernel with its channel axes swapped, w'[dy][dx][ci][4 s + c] = w[4 - s][4 - dy][4 - dx][c][ci]
__global__ void __launch_bounds__(256) k3_kpack_weights_out(const float* __restrict__ w, float* __restrict__ w2, int cr, int mode) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 25 * 32 * 32) return;
    const int k = e & 31, ci = (e >> 5) & 31, tap = e >> 10, dy = tap / 5, dx = tap % 5, s = k >> 2, c = k & 3;
    float v = 0.f;
    if (s < 5 && c < cr)
        v = mode == SOL_CONV_FWD ? w[((((size_t)s * 5 + dy) * 5 + dx) * 32 + ci) * cr + c]
                                 : w[((((size_t)(4 - s) * 5 + (4 - dy)) * 5 + (4 - dx)) * cr + c) * 32 + ci];
    w2[e] = v;
}
// y [planes][HW][cout] from y' [planes][HW][32]: one thread per output pixel (float4 per depth offset; the five planes of a pixel are 512 KB
// apart at 64 x 64 -- L2 / MALL resident, the tensor was just written)
__global__ void __launch_bounds__(256) k3_gather_depth(const float4* __restrict__ y2, const float* __restrict__ bias, float* __restrict__ y, int D, int HW, int cout, size_t npx) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < npx; q += (size_t)gridDim.x * 256) {
        const size_t plane = q / HW;
        const int d = (int)(plane % D);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int ds = d + s - 2;
            if (ds < 0 || ds >= D) continue;
            const float4 v = y2[((size_t)((long)q + (long)(s - 2) * HW)) * 8 + s];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const float r[4] = {acc.x, acc.y, acc.z, acc.w};
        if (cout == 4) {
            float4 o = acc;
            if (bias) { o.x += bias[0]; o.y += bias[1]; o.z += bias[2]; o.w += bias[3]; }
            reinterpret_cast<float4*>(y)[q] = o;
        } else {
            for (int c = 0; c < cout; ++c) y[q * cout + c] = r[c] + (bias ? bias[c] : 0.f);
        }
    }
}
}  // namespace

extern "C" int sol_conv3d_thin_out_pack(void* stream, const float* w_dhwio, int32_t cr, int32_t mode, float* packed) {
    SOL_REQUIRE(w_dhwio && packed, "sol_conv3d_thin_out_pack: NULL pointer");
    SOL_REQUIRE(mode == SOL_CONV_FWD || mode == SOL_CONV_BWD_DATA, "sol_conv3d_thin_out_pack: bad mode %d", mode);
    SOL_REQUIRE(cr >= 1 && cr <= 4, "sol_conv3d_thin_out_pack: 1..4 output channels of the convolution that is run (got %d)", cr);
    float* w2 = packed + align_up(sol_conv5x5_packed_floats(32, 32, SOL_CONV_FWD), 64);
    SOL_LAUNCH(k3_kpack_weights_out, dim3(100), dim3(256), 0, (hipStream_t)stream, w_dhwio, w2, cr, mode);
    SOL_LAUNCH_CHECK();
    return sol_conv5x5_pack(stream, w2, 32, 32, SOL_CONV_FWD, packed);
}

extern "C" int sol_conv3d_thin_out(void* stream, const float* x, const float* packed, const float* bias, float* y, float* ws,
                                   int32_t B, int32_t D, int32_t H, int32_t W, int32_t cout, const uint32_t* x_absmax) {
    SOL_REQUIRE(x && packed && y && ws && x != y, "sol_conv3d_thin_out: NULL pointer / in-place call");
    SOL_REQUIRE(B >= 1 && D >= 1 && H >= 1 && W >= 4 && W % 4 == 0 && cout >= 1 && cout <= 4, "sol_conv3d_thin_out: bad shape (B %d, D %d, H %d, W %d, cout %d)", B, D, H, W, cout);
    const size_t npx = (size_t)B * D * H * W;
    if (!x_absmax) {
        uint32_t* slots = reinterpret_cast<uint32_t*>(ws + npx * 32);
        if (int e = sol_absmax(stream, x, (int64_t)(npx * 32), slots)) return e;
        x_absmax = slots;
    }
    if (int e = sol_conv5x5_scaled(stream, x, packed, nullptr, nullptr, nullptr, ws, B * D, H, W, 32, 32, SOL_EPI_NONE, 0.f, x_absmax, nullptr)) return e;
    SOL_LAUNCH(k3_gather_depth, dim3(grid_for(npx)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(ws), bias, y, D, H * W, cout, npx);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

// Weight gradient of a thin-INPUT layer in the same packing: dW'[dy][dx][4 s + c][co] = sum_px x'[px + (dy, dx)][4 s + c] dz[px][co] is the 2-D
// 32 -> 32 weight gradient of the gathered tensor -- ONE pass of the fp16 three-product body (bww_sb_body) instead of five passes of the
// thin fp32-MFMA kernel (k_conv5x5_bww_thin<0, 2>: 5 x 79 us at 128 x 64 x 64) -- and dW[s][dy][dx][c][co] = dW'[dy][dx][4 s + c][co].
namespace {
__global__ void __launch_bounds__(256) k3_kunpack_dw(const float* __restrict__ dw2, float* __restrict__ dw, int cin) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 125 * cin * 32) return;
    const int co = e & 31, c = (e >> 5) % cin, t = e / (32 * cin), s = t / 25, tap = t % 25;
    dw[e] = dw2[(tap * 32 + 4 * s + c) * 32 + co];
}
// thin-OUTPUT layer (32 -> cout <= 4): dW[s][dy][dx][ci][c] = dW'[dy][dx][ci][4 s + c], db[c] = db'[8 + c] (the centre slot holds every plane of dz)
__global__ void __launch_bounds__(256) k3_kunpack_dw_out(const float* __restrict__ dw2, const float* __restrict__ db2, float* __restrict__ dw, float* __restrict__ db, int cout) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < cout) db[e] = db2[8 + e];
    if (e >= 125 * 32 * cout) return;
    const int c = e % cout, ci = (e / cout) & 31, t = e / (32 * cout), s = t / 25, tap = t % 25;
    dw[e] = dw2[(tap * 32 + ci) * 32 + 4 * s + c];
}
}  // namespace

extern "C" size_t sol_conv3d_thin_bwd_weight_ws_floats(int32_t B, int32_t D, int32_t H, int32_t /*W*/) {
    return align_up(sol_bww_batched_ws_floats(1, B * D, H, 32, 32), 64) + 25 * 32 * 32 + 32;
}

// The thin-OUTPUT layer (32 -> cout <= 4: the network's last layer) the same way: its output gradient dz [B,D,H,W,4] (zero padded) is gathered
// with the OPPOSITE depth offsets, dz'[d][..][4 s + c] = dz[d - s + 2][..][c], and dW'[dy][dx][ci][4 s + c] = sum x[px + (dy, dx)][ci] dz'[px][4 s + c]
// is one pass of the 32 -> 32 kernel (the five-pass form pads dz to 32 channels and runs FIVE full 32 -> 32 passes for three useful columns).
extern "C" int sol_conv3d_thin_out_bwd_weight_acc(void* stream, const float* x, const uint32_t* x_absmax, const float* dz4, float* ws, float* partial,
                                                  float* dw_dhwio, float* db, int32_t B, int32_t D, int32_t H, int32_t W, int32_t cout_real,
                                                  int32_t accumulate_partial, int32_t do_reduce) {
    SOL_REQUIRE(x && dz4 && ws && partial && dw_dhwio && db, "sol_conv3d_thin_out_bwd_weight: NULL pointer");
    SOL_REQUIRE(B >= 1 && D >= 1 && H >= 1 && W == 64 && cout_real >= 1 && cout_real <= 4, "sol_conv3d_thin_out_bwd_weight: needs W == 64 and 1..4 output channels (B %d, D %d, H %d, W %d, cout %d)", B, D, H, W, cout_real);
    const size_t npx = (size_t)B * D * H * W;
    uint32_t* slots_z = reinterpret_cast<uint32_t*>(ws + npx * 32);
    uint32_t* slots_x = slots_z + 256;
    if (int e = sol_absmax(stream, dz4, (int64_t)(npx * 4), slots_z)) return e;
    if (!x_absmax) {
        if (int e = sol_absmax(stream, x, (int64_t)(npx * 32), slots_x)) return e;
        x_absmax = slots_x;
    }
    SOL_LAUNCH(k3_kpack_depth, dim3(grid_for(npx * 8)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(dz4), reinterpret_cast<float4*>(ws), D, H * W, npx * 8, -1);
    SOL_LAUNCH_CHECK();
    if (int e = sol_bww_batched(stream, x, ws, partial, 1, 1, accumulate_partial ? 0 : 1, 0, 0, B * D, H, W, 32, 32, x_absmax, slots_z, 0, 0)) return e;
    if (!do_reduce) return SOL_OK;
    float* dw2 = partial + align_up(sol_bww_batched_ws_floats(1, B * D, H, 32, 32), 64);
    float* db2 = dw2 + 25 * 32 * 32;
    float* parts[1] = {partial}; float* dws[1] = {dw2}; float* dbs[1] = {db2};
    const int rows[1] = {B * D * H}, rbs[1] = {0}, ci[1] = {32}, co[1] = {32};
    if (int e = sol_bww_reduce_layers(stream, 1, parts, dws, dbs, rows, rbs, ci, co, 0, 0)) return e;
    SOL_LAUNCH(k3_kunpack_dw_out, dim3((125 * 32 * cout_real + 255) / 256), dim3(256), 0, (hipStream_t)stream, dw2, db2, dw_dhwio, db, cout_real);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

extern "C" int sol_conv3d_thin_bwd_weight_acc(void* stream, const float* x, const float* dz, const uint32_t* dz_absmax, float* ws, float* partial,
                                              float* dw_dhwio, float* db, int32_t B, int32_t D, int32_t H, int32_t W, int32_t cin_real,
                                              int32_t accumulate_partial, int32_t do_reduce) {
    SOL_REQUIRE(x && dz && ws && partial && dw_dhwio && db, "sol_conv3d_thin_bwd_weight: NULL pointer");
    SOL_REQUIRE(B >= 1 && D >= 1 && H >= 1 && W == 64 && cin_real >= 1 && cin_real <= 4, "sol_conv3d_thin_bwd_weight: needs W == 64 and 1..4 input channels (B %d, D %d, H %d, W %d, cin %d)", B, D, H, W, cin_real);
    const size_t npx = (size_t)B * D * H * W;
    uint32_t* slots_x = reinterpret_cast<uint32_t*>(ws + npx * 32);
    uint32_t* slots_z = slots_x + 256;
    if (int e = sol_absmax(stream, x, (int64_t)(npx * 4), slots_x)) return e;
    if (!dz_absmax) {
        if (int e = sol_absmax(stream, dz, (int64_t)(npx * 32), slots_z)) return e;
        dz_absmax = slots_z;
    }
    SOL_LAUNCH(k3_kpack_depth, dim3(grid_for(npx * 8)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(ws), D, H * W, npx * 8, 1);
    SOL_LAUNCH_CHECK();
    if (int e = sol_bww_batched(stream, ws, dz, partial, 1, 1, accumulate_partial ? 0 : 1, 0, 0, B * D, H, W, 32, 32, slots_x, dz_absmax, 0, 0)) return e;
    if (!do_reduce) return SOL_OK;
    float* dw2 = partial + align_up(sol_bww_batched_ws_floats(1, B * D, H, 32, 32), 64);
    float* parts[1] = {partial}; float* dws[1] = {dw2}; float* dbs[1] = {db};
    const int rows[1] = {B * D * H}, rbs[1] = {0}, ci[1] = {32}, co[1] = {32};
    if (int e = sol_bww_reduce_layers(stream, 1, parts, dws, dbs, rows, rbs, ci, co, 0, 0)) return e;
    SOL_LAUNCH(k3_kunpack_dw, dim3((125 * cin_real * 32 + 255) / 256), dim3(256), 0, (hipStream_t)stream, dw2, dw_dhwio, cin_real);
    SOL_LAUNCH_CHECK();
    return SOL_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Conv3D weight gradient: dW[kd][dy][dx][ci][co] = sum over planes d of the 2-D weight gradient of (x[d + kd - 2], dz[d]) --
// five passes of the batched 2-D weight-gradient kernels (conv5x5_sb.hip: one workgroup per CU owning many rows, fp16
// three-product operands when the absmax of both tensors is given) over the shifted plane ranges, each into `partial` and
// reduced into its depth slice of dw; db from the centre pass (all planes).
// ------------------------------------------------------------------------------------------------------------------------
static size_t conv3d_bww_part_floats(int D, int H, int cin, int cout) {    // the block layout depends on the plane count of a pass
    size_t m = 0;
    for (int np = D - 2; np <= D; ++np) m = std::max(m, sol_bww_batched_ws_floats(1, np, H, cin, cout));
    return align_up(m, 64);
}

extern "C" size_t sol_conv3d_bwd_weight_ws_floats(int32_t B, int32_t D, int32_t H, int32_t /*W*/, int32_t cin, int32_t cout) {
    (void)B;
    return 5 * conv3d_bww_part_floats(D, H, cin, cout);                           // one partial buffer per depth slice
}

// accumulate_partial: the passes add onto what `partial` holds (the previous unrolled steps of the same layer) instead of overwriting it;
// do_reduce: fold the partials into dw / db now.  A trainer that unrolls n steps calls this n times per layer on ONE partial buffer
// (first call: accumulate_partial = 0) and reduces once, with the last call: n - 1 reduce launch pairs per layer less (3.5 % of the
// kernel time of a SOL-16 step went into them) and no n-fold sum of weight-sized tensors on the host side.
extern "C" int sol_conv3d_bwd_weight_acc(void* stream, const float* x, const float* dz, const uint32_t* x_absmax, const uint32_t* dz_absmax,
                                         float* partial, float* dw_dhwio, float* db, float* db_scratch,
                                         int32_t B, int32_t D, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t cin_real, int32_t cout_real,
                                         int32_t accumulate_partial, int32_t do_reduce) {
    SOL_REQUIRE(x && dz && partial && dw_dhwio && db && db_scratch, "sol_conv3d_bwd_weight: NULL pointer");
    SOL_REQUIRE(B >= 1 && D >= 3 && (cin == 4 || cin == 32) && (cout == 2 || cout == 32) && cin_real >= 1 && cin_real <= cin && cout_real >= 1 && cout_real <= cout,
                "sol_conv3d_bwd_weight: bad shape (B %d, D %d, channels %d -> %d)", B, D, cin, cout);
    SOL_REQUIRE(cout_real == cout, "sol_conv3d_bwd_weight: the reduce writes dw with the kernel's output-channel count (pad dz and slice the result)");
    const size_t pin = (size_t)H * W * cin, pout = (size_t)H * W * cout;
    const size_t slice = (size_t)25 * cin_real * cout_real;
    const size_t per = conv3d_bww_part_floats(D, H, cin, cout);
    // Five passes (depth slices) x B simulations into five partial buffers -- a simulation's pass accumulates onto the previous
    // one's blocks (same plane range, same block layout) --, then ONE two-stage reduce launch pair for all five slices
    // (it was a pair per slice and simulation: 10 B launches of a few microseconds of work each per layer).
    float* parts[5]; float* dws[5]; float* dbs[5]; int rows[5], rbs[5], cins[5], couts[5];
    // 32 -> 32 on the split kernels: the five slices of a simulation are ONE launch (five rounds of workgroups back to back instead of five
    // launches of one round each; option k3d_bww_jobs)
    const bool jobs = cin == 32 && cout == 32 && W == 64 && sol_opt().conv_precision != 2 && sol_opt().k3d_bww_jobs;
    for (int kd = 0; kd < 5; ++kd) {
        const int lo = kd < 2 ? 2 - kd : 0, hi = kd > 2 ? D + 2 - kd : D;       // output planes that see input plane d + kd - 2
        parts[kd] = partial + kd * per; dws[kd] = dw_dhwio + kd * slice; dbs[kd] = kd == 2 ? db : db_scratch + (size_t)kd * cout;
        rows[kd] = (hi - lo) * H; rbs[kd] = 0; cins[kd] = cin_real; couts[kd] = cout_real;
    }
    // k3d_bww_jobs == 2: ONE round -- 51 workgroups per slice (255 of 256 CUs), each with a fifth of the rows a 32-row block form gives a CU
    // over five rounds: one prologue (three rows of look-ahead from HBM) and one read-modify-write of the 100 KB partial instead of five
    int rb_jobs = 0;
    if (jobs && sol_opt().k3d_bww_jobs == 2) {
        const int rbw = (D * H + 50) / 51;
        if (rbw > sol_bww_pick_rb((D - 2) * H) && rbw > sol_bww_pick_rb(D * H)) { rb_jobs = rbw; for (int kd = 0; kd < 5; ++kd) rbs[kd] = rbw; }
    }
    for (int b = 0; jobs && b < B; ++b) {
        const float* xs[5]; const float* zs[5]; int np[5];
        for (int kd = 0; kd < 5; ++kd) {
            const int lo = kd < 2 ? 2 - kd : 0, hi = kd > 2 ? D + 2 - kd : D;
            xs[kd] = x + ((size_t)b * D + lo + kd - 2) * pin; zs[kd] = dz + ((size_t)b * D + lo) * pout; np[kd] = hi - lo;
        }
        if (int e = sol_bww_batched_jobs(stream, 5, xs, zs, parts, np, b == 0 && !accumulate_partial, H, W, x_absmax, dz_absmax, rb_jobs)) return e;
    }
    for (int kd = 0; !jobs && kd < 5; ++kd) {
        const int lo = kd < 2 ? 2 - kd : 0, hi = kd > 2 ? D + 2 - kd : D;
        for (int b = 0; b < B; ++b) {
            const float* xb = x + ((size_t)b * D + lo + kd - 2) * pin;
            const float* zb = dz + ((size_t)b * D + lo) * pout;
            if (int e = sol_bww_batched(stream, xb, zb, parts[kd], 1, 1, b == 0 && !accumulate_partial, 0, 0, hi - lo, H, W, cin, cout, x_absmax, dz_absmax, 0, 0)) return e;
        }
    }
    if (!do_reduce) return SOL_OK;
    return sol_bww_reduce_layers(stream, 5, parts, dws, dbs, rows, rbs, cins, couts, 0, 0);
}

extern "C" int sol_conv3d_bwd_weight(void* stream, const float* x, const float* dz, const uint32_t* x_absmax, const uint32_t* dz_absmax,
                                     float* partial, float* dw_dhwio, float* db, float* db_scratch,
                                     int32_t B, int32_t D, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t cin_real, int32_t cout_real) {
    return sol_conv3d_bwd_weight_acc(stream, x, dz, x_absmax, dz_absmax, partial, dw_dhwio, db, db_scratch, B, D, H, W, cin, cout, cin_real, cout_real, 0, 1);
}
