#!/usr/bin/env python
"""Why does the replayed 3-D training graph go wrong after a few replays?  (Found by the SOL-16 full-size test: the loss of the
5th..7th replay jumps by 30 %, per-step losses from step 1 on differ.)  Variants of what happens BETWEEN replays."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sol_amd
from sol_amd import karman3d as k3, synthetic
DEV = "cuda"
B, Y, X, Z = 1, 128, 64, 64
ms = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mode = sys.argv[2] if len(sys.argv) > 2 else "none"
sc = k3.Scene3D(Y, X, Z, device=DEV)
gen = torch.Generator().manual_seed(11)
rn = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float32)
st = (torch.rand(B, Y, X, Z, generator=gen).to(DEV), (1.0 + 0.1 * rn(B, Y + 1, X, Z)).to(DEV), (0.1 * rn(B, Y, X + 1, Z)).to(DEV), (0.1 * rn(B, Y, X, Z + 1)).to(DEV))
re = synthetic.reynolds(B).float().to(DEV)
net = k3.MarsMoon3D(seed=3, device=DEV)
w = net.get_weights(); w[22] = w[22] * 0.01; net.set_weights(w)
tr = k3.Karman3DTrainer(net, sc, B, ms, (0.2, 0.2, 0.2), synthetic.STD_RE, use_graph=True)
gts = []
with torch.no_grad():
    st = tr.sim.step(*st, re)
    gs = (st[0], st[1] + 0.02, st[2], st[3])
    for _ in range(ms):
        gs = tr.sim.step(*gs, re)
        gts.append(tuple(t.clone() for t in gs[1:]))
keep = []
p0 = net.params.detach().clone()
big = torch.randn(1 << 20, device=DEV)
for k in range(12):
    l = tr.fwd_bwd(*st, re, gts)
    keep.append((l.clone(), tr.loss_steps.clone()))
    if mode == "item":
        l.item()
    elif mode == "tolist":
        tr.loss_steps.tolist()
    elif mode == "params":
        with torch.no_grad(): net.params.copy_((p0.double() * 1.0).float())
    elif mode == "d2h_big":
        big.cpu()
    elif mode == "sync":
        torch.cuda.synchronize()
torch.cuda.synchronize()
print("ms %d mode %-8s losses %s" % (ms, mode, ["%.3f" % float(a) for a, _ in keep]))
print("   step-1 losses %s" % ["%.2f" % float(b[1]) for _, b in keep])
