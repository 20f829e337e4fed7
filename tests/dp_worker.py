"""Rank process of tests/test_gpu_parity.py::test_data_parallel_two_ranks_one_gpu (and of the CPU gloo test's GPU twin):
SolTrainer under torch.distributed with world_size 2, both ranks on cuda:0 (RCCL refuses two ranks on one device, so the
transport is gloo; the trainer, the shard arithmetic, the SUM all-reduce semantics and the replicated Adam update are
the ones the RCCL path uses).  Usage: RANK/WORLD_SIZE/MASTER_* in the env, argv[1] = output .npz prefix."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import sol_amd                      # noqa: E402
import sol_oracle as o              # noqa: E402
from sol_amd import ops             # noqa: E402


def problem(Bg, Y, X, ms):
    """global batch of Bg simulations (float64 CPU tensors), identical on every rank"""
    g = o.geometry(Y, X)
    d, vy, vx = o.synthetic_state(Bg, Y, X, 1234)
    re = torch.tensor([o.RE_TRAIN[i % 6] for i in range(Bg)], dtype=torch.float64)
    gts = [o.synthetic_state(Bg, Y, X, 4321 + i, project_it=False) for i in range(ms)]
    return g, d, vy, vx, re, torch.stack([s[1] for s in gts]), torch.stack([s[2] for s in gts])


def run(Bg, Y, X, ms, rank, world, group=None, steps=2, lr=1e-4):
    dev = "cuda:0"
    g, d, vy, vx, re, gty, gtx = problem(Bg, Y, X, ms)
    lo, hi = sol_amd.dist.shard_range(Bg, rank, world)
    f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
    mk = ops.SceneMasks(g.active, g.inflow, g.bc_mask, g.bc_mask, dev)
    net = sol_amd.model_mars_moon(cin=3, cout=2, seed=0, device=dev)
    tr = sol_amd.SolTrainer(net, mk, hi - lo, Y, X, ms, g.dx, (0.2, 0.25), o.STD_RE, group=group)
    args = (f(d[lo:hi]), f(vy[lo:hi]), f(vx[lo:hi]), f(re[lo:hi]), f(gty[:, lo:hi]), f(gtx[:, lo:hi]))
    losses, grads0 = [], None
    for s in range(steps):
        losses.append(float(tr.train_step(*args, lr=lr)))
        if s == 0:
            grads0 = tr.grads.detach().cpu().numpy().copy()       # after the all-reduce: the GLOBAL gradient
    torch.cuda.synchronize()
    return np.array(losses), grads0, net.params.detach().cpu().numpy()


def run3d(Bg, Y, X, Z, ms, rank, world, group=None, steps=2, lr=1e-4):
    """the same for Karman3DTrainer (karman-3d, BASELINE configs[4] shards simulations over the GPUs exactly like the 2-D scene)"""
    import sol_oracle3d as o3
    from sol_amd import karman3d as k3
    dev = "cuda:0"
    d, v = o3.synthetic_state(Bg, Y, X, Z, 77)
    re = torch.tensor([o3.RE_TRAIN[i % len(o3.RE_TRAIN)] for i in range(Bg)], dtype=torch.float64)
    gts = [o3.synthetic_state(Bg, Y, X, Z, 500 + i)[1] for i in range(ms)]
    lo, hi = sol_amd.dist.shard_range(Bg, rank, world)
    f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
    sc = k3.Scene3D(Y, X, Z, device=dev)
    net = k3.MarsMoon3D(seed=3, device=dev)
    tr = k3.Karman3DTrainer(net, sc, hi - lo, ms, (0.2, 0.25, 0.3), o3.STD_RE, group=group)
    args = (f(d[lo:hi]), f(v[0][lo:hi]), f(v[1][lo:hi]), f(v[2][lo:hi]), f(re[lo:hi]), [tuple(f(c[lo:hi]) for c in g) for g in gts])
    losses, grads0 = [], None
    for s in range(steps):
        losses.append(float(tr.train_step(*args, lr=lr)))
        if s == 0:
            grads0 = tr.grads.detach().cpu().numpy().copy()
    torch.cuda.synchronize()
    return np.array(losses), grads0, net.params.detach().cpu().numpy()


if __name__ == "__main__":
    rank, world, _ = sol_amd.dist.init_from_env("gloo")
    if sys.argv[2] == "k3d":
        Bg, Y, X, Z, ms = (int(v) for v in sys.argv[3:8])
        losses, grads0, params = run3d(Bg, Y, X, Z, ms, rank, world)
    else:
        Bg, Y, X, ms = (int(v) for v in sys.argv[2:6])
        losses, grads0, params = run(Bg, Y, X, ms, rank, world)
    np.savez(sys.argv[1] + "_rank%d.npz" % rank, losses=losses, grads0=grads0, params=params)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
