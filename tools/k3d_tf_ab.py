#!/usr/bin/env python
"""karman-3d 128x64x64 solver step with the sine transforms on the fp32 matrix cores (option k3d_mfma_tf = 1) against the LDS-fed
VALU kernels (0): fields of one step against each other, per-kernel table, microseconds per solver step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sol_amd
from sol_amd import karman3d as k3, synthetic, _lib
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Y, X, Z = 128, 64, 64
sc = k3.Scene3D(Y, X, Z, device=dev)
gen = torch.Generator().manual_seed(1)
f = lambda *s: torch.randn(*s, generator=gen)
st = (torch.rand(B, Y, X, Z, generator=gen).to(dev), (1.0 + 0.1 * f(B, Y + 1, X, Z)).to(dev), (0.1 * f(B, Y, X + 1, Z)).to(dev), (0.1 * f(B, Y, X, Z + 1)).to(dev))
re = synthetic.reynolds(B).float().to(dev)
outs = {}
for m in (0, 1):
    _lib.set_option("k3d_mfma_tf", m)
    sim = k3.Karman3DFlow(sc, B)
    o = sim.step(*st, re)
    torch.cuda.synchronize()
    outs[m] = tuple(t.clone() for t in o)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        sim.step(*st, re)
    b.record(); torch.cuda.synchronize()
    with _lib.profile() as p:
        sim.step(*st, re)
    print("k3d_mfma_tf=%d: %.1f us per solver step (eager, 20 steps); kernels: %s" % (
        m, a.elapsed_time(b) / 20 * 1e3, {k.strip("()"): round(t / c, 1) for k, (c, t) in sorted(p.kernels.items(), key=lambda kv: -kv[1][1])}))
rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
print("fields mfma vs valu (rel L2):", [rel(x, y) for x, y in zip(outs[1], outs[0])])
_lib.set_option("k3d_mfma_tf", 1)
