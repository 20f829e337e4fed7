#!/usr/bin/env python
"""Micro-driver for profiling: runs the fused solver step (fwd, and bwd with --bwd) N times on a
synthetic C3 batch.  Usage: python tools/step_micro.py [--n 20] [--rtol 1e-6] [--bwd] [--res 64] [--batch 6]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sol_amd
from sol_amd import ops, synthetic

p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=20)
p.add_argument("--rtol", type=float, default=1e-6)
p.add_argument("--res", type=int, default=64)
p.add_argument("--batch", type=int, default=6)
p.add_argument("--bwd", action="store_true")
p.add_argument("--maxiter", type=int, default=2000)
a = p.parse_args()
X, Y, B = a.res, 2 * a.res, a.batch
dom = sol_amd.Domain([Y, X], box=sol_amd.box[0:200, 0:100])
flow = sol_amd.KarmanFlow()
active, inflow = flow.scene_arrays(dom)
bcv, bcm = sol_amd.velocity_bc_masks(Y, X)
masks = ops.SceneMasks(active, inflow, bcv.reshape(Y + 1, X), bcm.reshape(Y + 1, X))
f = lambda t: t.to(device="cuda", dtype=torch.float32).contiguous()
d0, vy0, vx0 = (f(t) for t in synthetic.state(B, Y, X, 1234))
re = f(synthetic.reynolds(B))
cfg = ops.karman_cfg(B, Y, X, dom.dx[1], cg_rtol=a.rtol, cg_max_iter=a.maxiter, masks=masks)
d0, vy0, vx0 = (t.detach() for t in ops.karman_step(d0, vy0, vx0, re, cfg, masks))
info = {}
if a.bwd:
    vy0.requires_grad_(True); vx0.requires_grad_(True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.n):
    out = ops.karman_step(d0, vy0, vx0, re, cfg, masks, info)
    if a.bwd:
        (out[1].sum() + out[2].sum()).backward()
e1.record()
torch.cuda.synchronize()
print("rtol %g: %.1f us per call, iters %s bwd %s" % (a.rtol, e0.elapsed_time(e1) / a.n * 1e3, info["iterations"].tolist(),
      info.get("iterations_bwd", torch.zeros(0)).tolist()))
