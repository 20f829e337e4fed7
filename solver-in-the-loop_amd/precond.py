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
