"""Host-side setup of the DIRECT pressure solver of the karman-3d step (csrc/karman3d.hip).

Same construction as precond.direct_solver_blob, one axis more.  On the OPEN box without obstacle the pressure matrix
M = -A (PhiFlow sparse_pressure_matrix semantics: diagonal = number of accessible neighbours, -1 between two active cells,
p = 0 outside) is the 7-point Dirichlet Laplacian M_r = T_Y (x) I (x) I + I (x) T_X (x) I + I (x) I (x) T_Z, diagonalised by
the orthonormal sine transforms Q_Y, Q_X, Q_Z.  The obstacle changes M only on the set S of obstacle cells and their
neighbours, M = M_r + U_S E_SS U_S^T, hence

    x = G (b - U_S E_SS x_S),    x_S = (I + G_SS E_SS)^-1 (G b)_S,    G = M_r^-1

i.e. two applications of G (three sine transforms each way = batched fp32 GEMMs on the device) and one dense |S| x |S|
product: NO iteration, the solution equals the converged CG solution up to fp32 round-off.  Everything here is float64
numpy, computed once per scene geometry (about a second at 128 x 64 x 64 with the sphere: |S| = 1 744).
"""
import numpy as np

from .precond import dst_matrix

FD3_MAGIC = 0x46443333          # "FD33"
FD3_HEADER = 16                 # int32 words
FD3_MAX_S = 8192                # largest padded perturbation set the device kernels take


def _accessible_diag(act):
    acc = np.pad(act, 1, mode="edge")
    n = np.zeros_like(act)
    for ax in range(3):
        lo = [slice(1, -1)] * 3
        hi = [slice(1, -1)] * 3
        lo[ax] = slice(0, -2)
        hi[ax] = slice(2, None)
        n += acc[tuple(lo)] + acc[tuple(hi)]
    return np.maximum(n, 1.0)


def _perturbation3d(active):
    """Sparse difference E = M - M_r as (rows, cols, vals) over global cell indices."""
    act = (np.asarray(active, dtype=np.float64) != 0).astype(np.float64)
    Y, X, Z = act.shape
    idx = np.arange(Y * X * Z).reshape(Y, X, Z)
    diag = _accessible_diag(act)
    sel = diag != 6.0
    rows, cols, vals = [idx[sel]], [idx[sel]], [diag[sel] - 6.0]
    for ax in range(3):
        lo = [slice(None)] * 3
        hi = [slice(None)] * 3
        lo[ax] = slice(0, -1)
        hi[ax] = slice(1, None)
        a = act[tuple(lo)] * act[tuple(hi)]
        sel = a != 1.0
        r, c = idx[tuple(lo)][sel], idx[tuple(hi)][sel]
        rows += [r, c]; cols += [c, r]; vals += [1.0 - a[sel], 1.0 - a[sel]]        # M has -a, M_r has -1
    return np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)


def rect_eigenvalues(Y, X, Z):
    ev = lambda n: 2.0 - 2.0 * np.cos(np.pi * np.arange(1, n + 1) / (n + 1))
    return ev(Y)[:, None, None] + ev(X)[None, :, None] + ev(Z)[None, None, :]


def _green_on_window(Qy, Qx, Qz, lam, wy, wx, wz):
    """G = M_r^-1 restricted to the box window wy x wx x wz (index arrays), as a 6-index array
    Gw[a, a', b, b', c, c'] = G[(wy[a], wx[b], wz[c]), (wy[a'], wx[b'], wz[c'])], contracted one axis at a time."""
    Wy, Wx, Wz = Qy[wy, :], Qx[wx, :], Qz[wz, :]
    H = np.einsum("am,jm,mce->ajce", Wy, Wy, 1.0 / lam, optimize=True)            # y axis contracted
    H = np.einsum("bc,ic,ajce->ajbie", Wx, Wx, H, optimize=True)                   # x axis
    return np.einsum("ke,le,ajbie->ajbikl", Wz, Wz, H, optimize=True)              # z axis


def direct_solver_blob3d(active):
    """float32 blob consumed by sol_karman3d_cfg.direct, or None when the scene does not qualify (no obstacle at all is
    fine: nS = 0; too many perturbed cells, or an ill-conditioned capacitance system, are not).

    layout (32-bit words): header[16] = {magic, Y, X, Z, nS, SP, ...};  Qy[Y*Y];  Qx[X*X];  Qz[Z*Z];  invlam[Y*X*Z];
    KpT[SP*SP] (K' = E_SS (I + G_SS E_SS)^-1, stored transposed, zero padded);  sidx[SP] int32 (global cell index
    (j*X + i)*Z + k, -1 = padding)."""
    act = (np.asarray(active, dtype=np.float64) != 0).astype(np.float64)
    Y, X, Z = act.shape
    rows, cols, vals = _perturbation3d(act)
    S = np.unique(rows)
    nS = len(S)
    SP = (nS + 63) // 64 * 64
    if SP > FD3_MAX_S:
        return None
    Qy, Qx, Qz = dst_matrix(Y), dst_matrix(X), dst_matrix(Z)
    lam = rect_eigenvalues(Y, X, Z)
    KpT = np.zeros((SP, SP))
    if nS:
        js, is_, ks = S // (X * Z), (S // Z) % X, S % Z
        wy, wx, wz = (np.arange(v.min(), v.max() + 1) for v in (js, is_, ks))
        if (len(wy) * len(wx) * len(wz)) ** 2 > 6e7:
            return None                                   # (an extruded obstacle spanning the domain: not a window-sized perturbation)
        Gw = _green_on_window(Qy, Qx, Qz, lam, wy, wx, wz)
        a, b, c = js - wy[0], is_ - wx[0], ks - wz[0]
        GSS = Gw[a[:, None], a[None, :], b[:, None], b[None, :], c[:, None], c[None, :]]
        pos = np.full(Y * X * Z, -1, dtype=np.int64)
        pos[S] = np.arange(nS)
        ESS = np.zeros((nS, nS))
        np.add.at(ESS, (pos[rows], pos[cols]), vals)
        cap = np.eye(nS) + GSS @ ESS
        if np.linalg.cond(cap) > 1e6:
            return None
        KpT[:nS, :nS] = (ESS @ np.linalg.inv(cap)).T
    sidx = np.full(SP, -1, dtype=np.int32)
    sidx[:nS] = S.astype(np.int32)
    header = np.zeros(FD3_HEADER, dtype=np.int32)
    header[:6] = [FD3_MAGIC, Y, X, Z, nS, SP]
    parts = [header.view(np.float32), Qy.astype(np.float32).ravel(), Qx.astype(np.float32).ravel(), Qz.astype(np.float32).ravel(),
             (1.0 / lam).astype(np.float32).ravel(), KpT.astype(np.float32).ravel(), sidx.view(np.float32)]
    return np.concatenate(parts)


def direct_solve_reference3d(blob, b):
    """float64 numpy restatement of the device algorithm on the blob: b [Y,X,Z] -> x with M x = b (CPU tests)."""
    hdr = blob[:FD3_HEADER].view(np.int32)
    assert hdr[0] == FD3_MAGIC
    Y, X, Z, nS, SP = (int(v) for v in hdr[1:6])
    o = FD3_HEADER
    Qy = blob[o:o + Y * Y].astype(np.float64).reshape(Y, Y); o += Y * Y
    Qx = blob[o:o + X * X].astype(np.float64).reshape(X, X); o += X * X
    Qz = blob[o:o + Z * Z].astype(np.float64).reshape(Z, Z); o += Z * Z
    il = blob[o:o + Y * X * Z].astype(np.float64).reshape(Y, X, Z); o += Y * X * Z
    KpT = blob[o:o + SP * SP].astype(np.float64).reshape(SP, SP); o += SP * SP
    sidx = blob[o:o + SP].view(np.int32)
    q3 = lambda t: np.einsum("am,bc,ke,mce->abk", Qy, Qx, Qz, t, optimize=True)
    G = lambda t: q3(q3(t) * il)
    g = G(b)
    if nS:
        xs = np.where(sidx >= 0, g.ravel()[np.maximum(sidx, 0)], 0.0)
        c = KpT.T @ xs
        b = b.copy().ravel()
        b[sidx[sidx >= 0]] -= c[sidx >= 0]
        g = G(b.reshape(Y, X, Z))
    return g
