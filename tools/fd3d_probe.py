import sys, os
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, numpy as np
import sol_amd
from sol_amd import karman3d as k3, synthetic
DEV="cuda"
B, Y, X, Z = 1, 128, 64, 64
ms = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sc = k3.Scene3D(Y, X, Z, device=DEV)
gen = torch.Generator().manual_seed(11)
rn = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float32)
st = (torch.rand(B, Y, X, Z, generator=gen).to(DEV), (1.0 + 0.1 * rn(B, Y + 1, X, Z)).to(DEV), (0.1 * rn(B, Y, X + 1, Z)).to(DEV), (0.1 * rn(B, Y, X, Z + 1)).to(DEV))
re = synthetic.reynolds(B).float().to(DEV)
def make(use_graph):
    net = k3.MarsMoon3D(seed=3, device=DEV)
    w = net.get_weights(); w[22] = w[22] * 0.01; net.set_weights(w)
    return net, k3.Karman3DTrainer(net, sc, B, ms, (0.2, 0.2, 0.2), synthetic.STD_RE, use_graph=use_graph)
net, tr = make(True)
gts = []
with torch.no_grad():
    st = tr.sim.step(*st, re)
    gs = (st[0], st[1] + 0.02, st[2], st[3])
    for _ in range(ms):
        gs = tr.sim.step(*gs, re)
        gts.append(tuple(t.clone() for t in gs[1:]))
l0 = float(tr.fwd_bwd(*st, re, gts)); g = tr.grads.double().clone()
u = torch.zeros(net.n_params, dtype=torch.float64)
for k in range(len(net.shapes)):
    sl = slice(int(net.offsets[k]), int(net.offsets[k + 1]))
    wk = net.params.detach()[sl].double().cpu()
    u[sl] = torch.randn(wk.numel(), generator=gen, dtype=torch.float64) * (float(wk.abs().mean()) + 1e-3)
u = u.to(DEV)
print("ms", ms, "loss", l0, "<g,u>", float((g*u).sum()), "per-step losses", tr.loss_steps.tolist())
p0 = net.params.detach().clone()
print("gts checksums at start", [[float(t.double().sum()) for t in g_] for g_ in gts][:2])
for k in range(-4, 5):
    eps = k * 2.5e-4
    with torch.no_grad(): net.params.copy_((p0.double() + eps * u).float())
    l = float(tr.fwd_bwd(*st, re, gts))
    print("eps %+.2e loss %.4f  dl %.4f  lin %.4f   steps[0,1,-1] %s" % (eps, l, l - l0, eps * float((g*u).sum()), [tr.loss_steps[0].item(), tr.loss_steps[1].item(), tr.loss_steps[-1].item()]))
# ---- which persistent buffer changed? ----
cs = lambda ts: [float(t.double().sum()) for t in ts]
print("gts checksums now", [cs(g_) for g_ in gts][:2], "...")
with torch.no_grad(): net.params.copy_(p0)
l = float(tr.fwd_bwd(*st, re, gts)); print("p0 again (graph):", l, tr.loss_steps.tolist()[:3])
net2 = k3.MarsMoon3D(seed=3, device=DEV); w = net2.get_weights(); w[22] = w[22] * 0.01; net2.set_weights(w)
tr2 = k3.Karman3DTrainer(net2, sc, B, ms, (0.2, 0.2, 0.2), synthetic.STD_RE, use_graph=False)
l = float(tr2.fwd_bwd(*st, re, gts)); print("fresh eager trainer, same scene object:", l, tr2.loss_steps.tolist()[:3])
sc2 = k3.Scene3D(Y, X, Z, device=DEV)
tr3 = k3.Karman3DTrainer(net2, sc2, B, ms, (0.2, 0.2, 0.2), synthetic.STD_RE, use_graph=False)
l = float(tr3.fwd_bwd(*st, re, gts)); print("fresh eager trainer, fresh scene:", l, tr3.loss_steps.tolist()[:3])
for i in range(ms):
    print("step", i, "tr._gt == gts:", [bool(torch.equal(tr._gt[c][i], gts[i][c])) for c in range(3)], " max|diff|", [float((tr._gt[c][i] - gts[i][c]).abs().max()) for c in range(3)])
print("tr._in == st:", [bool(torch.equal(a_, b_)) for a_, b_ in zip(tr._in[:4], st)])
print("data_ptrs gt:", [hex(t.data_ptr()) for t in tr._gt], "in:", [hex(t.data_ptr()) for t in tr._in], "flat", hex(tr._flat.data_ptr()), "loss_steps", hex(tr.loss_steps.data_ptr()), "params", hex(net.params.data_ptr()))
