#!/usr/bin/env python
"""Burgers training step (BASELINE configs[0]: 32x32, batch 5, dt 0.1): replayed hipGraph (sol_amd.BurgersTrainer) against the
eager composition of the same ops.  Usage: python tools/burgers_step_time.py [msteps ...]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sol_amd
B, Y, X, dt = 5, 32, 32, 0.1
dom = sol_amd.Domain([Y, X], box=sol_amd.box([32, 32]), boundaries=sol_amd.PERIODIC)
rng = np.random.default_rng(0)
for ms in ([int(a) for a in sys.argv[1:]] or [1, 2, 4]):
    velo = 0.3 * rng.standard_normal((ms + 1, B, Y + 1, X + 1, 2)).astype(np.float32)
    forc = 0.1 * rng.standard_normal((ms, B, Y + 1, X + 1, 2)).astype(np.float32)
    out = []
    for graph in (True, False):
        net = sol_amd.model_mars_moon(cin=4, cout=2, seed=0)
        tr = sol_amd.BurgersTrainer(net, dom, B, ms, dt, (0.2, 0.2), (0.1, 0.1), use_graph=graph)
        v, f = torch.as_tensor(velo, device="cuda"), torch.as_tensor(forc, device="cuda")
        for _ in range(3):
            tr.train_step(v, f, 1e-4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            loss = tr.train_step(v, f, 1e-4)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e3)
    print("msteps %d: hipGraph %.3f ms/step, eager %.3f ms/step (x%.1f), loss %.4f" % (ms, out[0], out[1], out[1] / out[0], float(loss)))
