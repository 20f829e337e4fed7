#!/usr/bin/env python
"""A/B of the persistent CNN chain (option cnn_persistent) against the per-layer launches on the bench workload:
loss / gradient / final-state agreement and ms per training step.  Usage: python tools/ab_chain.py [msteps] [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, sol_amd
from sol_amd import _lib
dev = torch.device("cuda", 0)
ms = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))
res = {}
for mode in (0, 1):
    _lib.set_option("cnn_persistent", mode)
    wl = bench.Workload(sol_amd, dev, B, 128, 64, ms, 0)
    tr = wl.trainer
    loss = tr.fwd_bwd(wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx, want_final=True)
    torch.cuda.synchronize()
    res[mode] = (float(loss), tr.loss_steps.clone(), tr.grads.clone(), [t.clone() for t in tr.final])
    for _ in range(3):
        wl.step(1e-6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step(1e-6)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    print("cnn_persistent=%d: loss %.6f  %.3f ms/step  finite grads %s" % (mode, res[mode][0], dt, bool(torch.isfinite(tr.grads).all())), flush=True)
    with _lib.profile() as p:
        tr.fwd_bwd(wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx, want_final=True, eager=True)
    for k, (c, t) in sorted(p.kernels.items(), key=lambda kv: -kv[1][1])[:6]:
        print("    %-28s calls %4d  avg %8.2f us  total %8.3f ms" % (k, c, t / c, t * 1e-3))
    del wl, tr
a, b = res[0], res[1]
print("loss rel diff %.3e | loss_steps %.3e | grads %.3e | final vy %.3e vx %.3e d %.3e" % (
    abs(a[0] - b[0]) / abs(a[0]), rel(b[1], a[1]), rel(b[2], a[2]), rel(b[3][1], a[3][1]), rel(b[3][2], a[3][2]), rel(b[3][0], a[3][0])))
