#!/usr/bin/env python
"""karman-3d SOL-16 training steps at 128x64x64 ALONE (no roll-out legs), for a kernel trace of exactly the training step:
    rocprofv3 --kernel-trace --stats -d OUT -o trace -- python tools/k3d_train_prof.py [--steps 3] [--eager]
    python tools/rocpd_stats.py OUT/.../*.db        (divide the call counts by steps + 1: one warm-up / capture step)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, sol_amd
from sol_amd import karman3d as k3, synthetic
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 3
dev = torch.device("cuda", 0)
B, Y, X, Z, ms3 = 1, 128, 64, 64, 16
sc = k3.Scene3D(Y, X, Z, device=dev)
net = k3.MarsMoon3D(device=dev)
w = net.get_weights(); w[22] = w[22] * 0.01; net.set_weights(w)
gen = torch.Generator().manual_seed(4)
r = lambda *s: torch.randn(*s, generator=gen)
st = (torch.rand(B, Y, X, Z, generator=gen).to(dev), (1.0 + 0.1 * r(B, Y + 1, X, Z)).to(dev), (0.1 * r(B, Y, X + 1, Z)).to(dev), (0.1 * r(B, Y, X, Z + 1)).to(dev))
re = synthetic.reynolds(B).float().to(dev)
tr = k3.Karman3DTrainer(net, sc, B, ms3, (0.2, 0.2, 0.2), synthetic.STD_RE, use_graph="--eager" not in sys.argv)
gts, gs = [], st
with torch.no_grad():
    for _ in range(ms3):
        gs = tr.sim.step(gs[0], gs[1], gs[2], gs[3], re)
        gts.append(tuple(t + 0.01 * torch.randn_like(t) for t in gs[1:]))
l0 = float(tr.train_step(*st, re, gts, lr=1e-7))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l1 = float(tr.train_step(*st, re, gts, lr=1e-7))
torch.cuda.synchronize()
print(json.dumps({"ms_per_step": (time.perf_counter() - t0) / steps * 1e3, "steps": steps, "loss": l1, "loss_first": l0}))
