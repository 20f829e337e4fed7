"""Synthetic karman-2d batches of the shape BASELINE.json's configs name.

The reference ships no dataset (`*-set/` is git-ignored); benchmarks and the GPU tests use
seeded smooth random fields: v_y = 1 + 0.2*G1, v_x = 0.2*G2, density = U(0,1), G = Gaussian
noise low-passed by 4 Jacobi sweeps.  Generated on the CPU with torch, deterministic per seed.
"""
import numpy as np
import torch
import torch.nn.functional as F

RE_TRAIN = [160000.0, 320000.0, 640000.0, 1280000.0, 2560000.0, 5120000.0]   # karman-2d/Makefile:22
STD_RE = float(np.std(RE_TRAIN))


def _smooth(g, sweeps=4):
    for _ in range(sweeps):
        gp = F.pad(g.unsqueeze(1), (1, 1, 1, 1), mode="replicate").squeeze(1)
        g = 0.2 * (g + gp[:, 2:, 1:-1] + gp[:, :-2, 1:-1] + gp[:, 1:-1, 2:] + gp[:, 1:-1, :-2])
    return g


def state(B, Y, X, seed):
    """(d [B,Y,X], vy [B,Y+1,X], vx [B,Y,X+1]) float64 CPU tensors."""
    gen = torch.Generator().manual_seed(seed)
    vy = 1.0 + 0.2 * _smooth(torch.randn(B, Y + 1, X, generator=gen, dtype=torch.float64))
    vx = 0.2 * _smooth(torch.randn(B, Y, X + 1, generator=gen, dtype=torch.float64))
    d = torch.rand(B, Y, X, generator=gen, dtype=torch.float64)
    return d, vy, vx


def frames(msteps, B, Y, X, seed):
    """ground-truth velocity frames gt_vy [msteps,B,Y+1,X], gt_vx [msteps,B,Y,X+1]."""
    gy, gx = [], []
    for i in range(msteps):
        _, vy, vx = state(B, Y, X, seed + 7919 * (i + 1))
        gy.append(vy)
        gx.append(vx)
    return torch.stack(gy), torch.stack(gx)


def reynolds(B):
    return torch.tensor([RE_TRAIN[i % len(RE_TRAIN)] for i in range(B)], dtype=torch.float64)
