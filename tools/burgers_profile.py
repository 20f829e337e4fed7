#!/usr/bin/env python
"""Per-kernel table of the Burgers training step (32x32, batch 5; NON m=1 / SOL-04 m=4, burgers/Makefile:69-77) as replayed by
sol_amd.BurgersTrainer: run under `rocprofv3 --kernel-trace --stats -- python tools/burgers_profile.py [msteps] [schedule]`."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sol_amd
ms = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sched = sys.argv[2] if len(sys.argv) > 2 else "manual"
B, Y, X, dt = 5, 32, 32, 0.1
dom = sol_amd.Domain([Y, X], box=sol_amd.box([32, 32]), boundaries=sol_amd.PERIODIC)
rng = np.random.default_rng(0)
velo = torch.as_tensor(0.3 * rng.standard_normal((ms + 1, B, Y + 1, X + 1, 2)).astype(np.float32), device="cuda")
forc = torch.as_tensor(0.1 * rng.standard_normal((ms, B, Y + 1, X + 1, 2)).astype(np.float32), device="cuda")
net = sol_amd.model_mars_moon(cin=4, cout=2, seed=0)
tr = sol_amd.BurgersTrainer(net, dom, B, ms, dt, (0.2, 0.2), (0.1, 0.1), use_graph=True, schedule=sched)
for _ in range(3):
    tr.train_step(velo, forc, 1e-4)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(50):
    tr.train_step(velo, forc, 1e-4)
torch.cuda.synchronize()
print("burgers m=%d %s: %.3f ms per training step" % (ms, sched, (time.perf_counter() - t0) / 50 * 1e3))
