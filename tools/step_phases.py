#!/usr/bin/env python
"""Phase times of the fused solver step (forward and adjoint) from the kernels' own stamps (option step_prof: synchronous,
prints to stderr).  Usage: python tools/step_phases.py [B] [X]   (grid 2X x X).  The per-op step is launched three times
forward and backward; read the last pair (warm instruction cache, blob in the L2)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sol_amd
from sol_amd import ops, synthetic, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
X = int(sys.argv[2]) if len(sys.argv) > 2 else 64
Y = 2 * X
dev = torch.device("cuda", 0)
dom = sol_amd.Domain([Y, X], box=sol_amd.box[0:200, 0:100])
active, inflow = sol_amd.KarmanFlow().scene_arrays(dom)
bcv, bcm = sol_amd.velocity_bc_masks(Y, X)
masks = ops.SceneMasks(active, inflow, bcv.reshape(Y + 1, X), bcm.reshape(Y + 1, X), dev)
f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
d0, vy0, vx0 = (f(t) for t in synthetic.state(B, Y, X, 1234))
re = f(synthetic.reynolds(B))
cfg = ops.karman_cfg(B, Y, X, dom.dx[1], masks=masks)
_lib.set_option("step_prof", 1)
for it in range(3):
    vy, vx = vy0.clone().requires_grad_(True), vx0.clone().requires_grad_(True)
    sys.stderr.write("--- launch pair %d\n" % it)
    _, py, px = ops.karman_step(d0, vy, vx, re, cfg, masks)
    (py.sum() + px.sum()).backward()
torch.cuda.synchronize()

# ---- the same stamps inside ONE eager training step (cold operands, the weight-gradient workgroups riding in the adjoint
#      launches): `python tools/step_phases.py B X train` prints the fused launches of the reverse sweep (last three shown by tail)
if len(sys.argv) > 3 and sys.argv[3] == "train":
    import bench
    _lib.set_option("step_prof", 0)
    wl = bench.Workload(sol_amd, dev, B, Y, X, 32, 0)
    for _ in range(2):
        wl.step(1e-6)
    torch.cuda.synchronize()
    _lib.set_option("step_prof", 1)
    sys.stderr.write("--- eager training step\n")
    wl.trainer.fwd_bwd(wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx, want_final=True, eager=True)
    torch.cuda.synchronize()
    _lib.set_option("step_prof", 0)
