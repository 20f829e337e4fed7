#!/usr/bin/env python
"""Timing experiment: the persistent chain with a library variant (argv[1] = lib name under lib/), eager profile of the chain."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, sol_amd
from sol_amd import _lib, _build
if len(sys.argv) > 1 and sys.argv[1] != "product":
    _build.LIB = os.path.join(ROOT, "solver-in-the-loop_amd", "lib", sys.argv[1])
    _build._stale = lambda: False
import bench
_lib.set_option("cnn_persistent", 1)
wl = bench.Workload(sol_amd, torch.device("cuda", 0), 6, 128, 64, 4, 0)
for _ in range(3):
    wl.trainer.fwd_bwd(wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx, eager=True)
with _lib.profile() as p:
    wl.trainer.fwd_bwd(wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx, eager=True)
c, t = p.kernels["k_cnn_chain"]
print("%s: k_cnn_chain %d calls, %.2f us per chain" % (sys.argv[1] if len(sys.argv) > 1 else "product", c, t / c))
