#!/usr/bin/env python
"""Loss trajectories of the bench workload (bench.Workload) for several Adam learning rates: which lr keeps the
synthetic SOL-32 workload finite over the driver's 5 + 20 steps.  Usage: python tools/lr_sweep.py [steps] [lr ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench            # noqa: E402
import sol_amd          # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lrs = [float(v) for v in sys.argv[2:]] or [1e-4, 3e-5, 1e-5, 3e-6, 1e-6]
dev = torch.device("cuda", 0)
for lr in lrs:
    wl = bench.Workload(sol_amd, dev, 6, 128, 64, 32, 0)
    tr = [float(wl.step(lr)) for _ in range(steps)]
    print("lr %g: %s" % (lr, " ".join("%.5g" % v for v in tr)), flush=True)
    del wl
