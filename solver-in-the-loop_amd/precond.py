"""Host-side setup of the two-level CG preconditioner of the fused solver step.

M^-1 = D^-1 + P (P^T M P)^-1 P^T for M = -A, the SPD pressure matrix of the scene
(PhiFlow sparse_pressure_matrix semantics: diagonal = number of accessible neighbours (>= 1),
off-diagonal -1 between two active cells, p = 0 outside the OPEN domain), with P the
piecewise-constant prolongation of 8x8-cell aggregates restricted to active cells.  Only the dense
coarse inverse (nc x nc, nc = Y*X/64) is computed here, once per scene geometry, in float64; the
device kernels apply it inside their CG loop (csrc/karman_step.hip: pcg_solve).  It changes the
iteration count (about 235 -> 56 at 128x64), not the converged solution.
"""
import numpy as np


def pressure_matrix_dense_coarse(active, block=8):
    """P^T M P as a dense [nc,nc] float64 array, assembled directly from the stencil."""
    act = (np.asarray(active, dtype=np.float64) != 0).astype(np.float64)
    Y, X = act.shape
    assert Y % block == 0 and X % block == 0
    nby, nbx = Y // block, X // block
    nc = nby * nbx
    acc = np.pad(act, 1, mode="edge")            # accessible: 'boundary' extrapolation (OPEN)
    diag = np.maximum(acc[0:Y, 1:X + 1] + acc[2:Y + 2, 1:X + 1] + acc[1:Y + 1, 0:X] + acc[1:Y + 1, 2:X + 2], 1.0)
    jj, ii = np.meshgrid(np.arange(Y), np.arange(X), indexing="ij")
    blk = (jj // block) * nbx + (ii // block)
    Ac = np.zeros((nc, nc))
    # diagonal contributions (P has the active mask as entries)
    np.add.at(Ac, (blk.ravel(), blk.ravel()), (act * diag * act).ravel())
    # off-diagonal contributions: -act[c]*act[n] for the 4 neighbours inside the domain
    for sj, si in ((1, 0), (0, 1)):
        a = act[0:Y - sj, 0:X - si] * act[sj:Y, si:X]
        b0 = blk[0:Y - sj, 0:X - si].ravel()
        b1 = blk[sj:Y, si:X].ravel()
        np.add.at(Ac, (b0, b1), -a.ravel())
        np.add.at(Ac, (b1, b0), -a.ravel())
    return Ac


def coarse_inverse(active, block=8):
    """float32 [nc,nc] inverse of the coarse matrix (aggregates without active cells get 1)."""
    Ac = pressure_matrix_dense_coarse(active, block)
    empty = np.diag(Ac) == 0
    Ac[empty, empty] = 1.0
    return np.linalg.inv(Ac).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# Direct pressure solver: fast diagonalisation of the rectangle + capacitance correction
# ---------------------------------------------------------------------------------------------
# On the OPEN rectangle without obstacle M is the 5-point Dirichlet Laplacian M_r = T_Y (x) I + I (x) T_X,
# diagonalised by the orthonormal sine transforms Q_Y, Q_X:  M_r^-1 = (Q_Y (x) Q_X) diag(1/lam) (Q_Y (x) Q_X).
# The obstacle changes M only on the set S of obstacle cells and their neighbours (164 cells at
# 128x64, inside one 16x16 window):  M = M_r + U_S E_SS U_S^T.  Then
#     x = G (b - U_S E_SS x_S),   x_S = (I + G_SS E_SS)^-1 (G b)_S,   G = M_r^-1
# i.e. one forward transform, a window-sized dense correction, one inverse transform: NO iteration,
# same solution as the converged CG up to fp32 round-off (csrc/karman_step.hip: fd_solve).
FD_MAGIC = 0x46443032          # "FD02"
FD_HEADER = 16                 # int32 words
FD_WIN = 16                    # window edge (cells) of the one-workgroup kernels; the large-grid path takes 16/32/64


def dst_matrix(n):
    k = np.arange(1, n + 1)
    return np.sqrt(2.0 / (n + 1)) * np.sin(np.pi * np.outer(k, k) / (n + 1))


def scene_matrix(active):
    """Dense float64 M = -A of the scene (small grids / setup only)."""
    act = (np.asarray(active, dtype=np.float64) != 0).astype(np.float64)
    Y, X = act.shape
    N = Y * X
    acc = np.pad(act, 1, mode="edge")
    diag = np.maximum(acc[0:Y, 1:X + 1] + acc[2:Y + 2, 1:X + 1] + acc[1:Y + 1, 0:X] + acc[1:Y + 1, 2:X + 2], 1.0)
    idx = np.arange(N).reshape(Y, X)
    M = np.zeros((N, N))
    M[idx.ravel(), idx.ravel()] = diag.ravel()
    for sj, si in ((1, 0), (0, 1)):
        a = (act[0:Y - sj, 0:X - si] * act[sj:Y, si:X]).ravel()
        r = idx[0:Y - sj, 0:X - si].ravel()
        c = idx[sj:Y, si:X].ravel()
        M[r, c] -= a
        M[c, r] -= a
    return M


def _perturbation(active):
    """Sparse difference E = M - M_r as {(row, col): value} restricted to its support set S."""
    act = (np.asarray(active, dtype=np.float64) != 0).astype(np.float64)
    Y, X = act.shape
    acc = np.pad(act, 1, mode="edge")
    diag = np.maximum(acc[0:Y, 1:X + 1] + acc[2:Y + 2, 1:X + 1] + acc[1:Y + 1, 0:X] + acc[1:Y + 1, 2:X + 2], 1.0)
    ent = {}
    for j, i in zip(*np.nonzero(diag != 4.0)):
        ent[(j * X + i, j * X + i)] = diag[j, i] - 4.0
    for sj, si in ((1, 0), (0, 1)):
        a = act[0:Y - sj, 0:X - si] * act[sj:Y, si:X]
        for j, i in zip(*np.nonzero(a != 1.0)):
            r, c = j * X + i, (j + sj) * X + (i + si)
            ent[(r, c)] = ent[(c, r)] = 1.0 - a[j, i]      # M has -a, M_r has -1
    return ent


def direct_solver_blob(active, max_window=FD_WIN):
    """float32 blob consumed by sol_karman_cfg.direct, or None when the scene does not qualify
    (perturbed cells do not fit one window, or the capacitance system is ill conditioned).  The window edge is the
    smallest of 16, 32, 64 (<= max_window) that holds the perturbed cells; it is stored in header[7].

    layout (32-bit words): header[16] = {magic, Y, X, wy0, wx0, nS, SP, win, ...};  Qy[Y*Y];  Qx[X*X];
    invlamT[X*Y] (= 1/lam[m][c] stored [c][m]);  KpT[SP*SP] (K' = E_SS (I + G_SS E_SS)^-1, stored
    transposed, zero padded);  sidx[SP] int32 (window-local index j'*win + i', -1 = padding);
    QxW[X*win] (= Qx[c][wx0 + i'], the window columns of Qx as one compact slab)."""
    act = (np.asarray(active, dtype=np.float64) != 0).astype(np.float64)
    Y, X = act.shape
    ent = _perturbation(act)
    if not ent:
        return None
    S = np.array(sorted({r for r, _ in ent}), dtype=np.int64)
    js, is_ = S // X, S % X
    span = int(max(js.max() - js.min(), is_.max() - is_.min())) + 1
    win = next((w for w in (16, 32, 64) if w >= span and w <= max_window), None)
    if win is None or Y < win or X < win:
        return None
    wy0 = int(min(js.min(), Y - win))
    wx0 = int(min(is_.min(), X - win))
    nS = len(S)
    SP = (nS + 63) // 64 * 64
    pos = {int(s): n for n, s in enumerate(S)}
    ESS = np.zeros((nS, nS))
    for (r, c), v in ent.items():
        ESS[pos[r], pos[c]] = v
    Qy, Qx = dst_matrix(Y), dst_matrix(X)
    lam = (2 - 2 * np.cos(np.pi * np.arange(1, Y + 1) / (Y + 1)))[:, None] + (2 - 2 * np.cos(np.pi * np.arange(1, X + 1) / (X + 1)))[None, :]
    # G_SS[s, t] = sum_{m,c} Qy[js,m] Qx[is,c] / lam[m,c] * Qy[jt,m] Qx[it,c]
    Fy, Fx = Qy[js, :], Qx[is_, :]                                # [nS, Y], [nS, X]
    F = (Fy[:, :, None] * Fx[:, None, :]).reshape(nS, Y * X)
    GSS = (F / lam.reshape(1, Y * X)) @ F.T
    cap = np.eye(nS) + GSS @ ESS
    if np.linalg.cond(cap) > 1e6:
        return None
    Kp = ESS @ np.linalg.inv(cap)
    KpT = np.zeros((SP, SP))
    KpT[:nS, :nS] = Kp.T
    sidx = np.full(SP, -1, dtype=np.int32)
    sidx[:nS] = ((js - wy0) * win + (is_ - wx0)).astype(np.int32)
    header = np.zeros(FD_HEADER, dtype=np.int32)
    header[:8] = [FD_MAGIC, Y, X, wy0, wx0, nS, SP, win]
    parts = [header.view(np.float32), Qy.astype(np.float32).ravel(), Qx.astype(np.float32).ravel(),
             (1.0 / lam).T.astype(np.float32).ravel(), KpT.astype(np.float32).ravel(), sidx.view(np.float32),
             np.ascontiguousarray(Qx[:, wx0:wx0 + win]).astype(np.float32).ravel()]
    return np.concatenate(parts)


def direct_solve_reference(blob, b):
    """float64 numpy restatement of the device algorithm on the blob (used by the CPU tests)."""
    hdr = blob[:FD_HEADER].view(np.int32)
    assert hdr[0] == FD_MAGIC
    Y, X, wy0, wx0, nS, SP = (int(v) for v in hdr[1:7])
    FD_WIN = int(hdr[7])
    o = FD_HEADER
    Qy = blob[o:o + Y * Y].astype(np.float64).reshape(Y, Y); o += Y * Y
    Qx = blob[o:o + X * X].astype(np.float64).reshape(X, X); o += X * X
    il = blob[o:o + X * Y].astype(np.float64).reshape(X, Y).T; o += X * Y
    KpT = blob[o:o + SP * SP].astype(np.float64).reshape(SP, SP); o += SP * SP
    sidx = blob[o:o + SP].view(np.int32)
    T2 = (Qy @ b @ Qx) * il
    x0w = Qy[wy0:wy0 + FD_WIN, :] @ T2 @ Qx[:, wx0:wx0 + FD_WIN]
    xs = np.where(sidx >= 0, x0w.ravel()[np.maximum(sidx, 0)], 0.0)
    c = KpT.T @ xs
    w2 = np.zeros(FD_WIN * FD_WIN)
    w2[sidx[sidx >= 0]] = -c[sidx >= 0]
    w2 = w2.reshape(FD_WIN, FD_WIN)
    T2 = T2 + il * (Qy[:, wy0:wy0 + FD_WIN] @ w2 @ Qx[wx0:wx0 + FD_WIN, :])
    return Qy @ T2 @ Qx
