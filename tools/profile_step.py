#!/usr/bin/env python
"""Per-kernel table of ONE eager training step (per-launch HIP events, sol_prof_*) + graph-replay ms/step.
Usage: python tools/profile_step.py [B] [res] [msteps] [cnn_persistent]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, sol_amd
from sol_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
X = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ms = int(sys.argv[3]) if len(sys.argv) > 3 else 32
if len(sys.argv) > 4:
    _lib.set_option("cnn_persistent", int(sys.argv[4]))
dev = torch.device("cuda", 0)
wl = bench.Workload(sol_amd, dev, B, 2 * X, X, ms, 0)
for _ in range(3):
    wl.step(1e-6)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    wl.step(1e-6)
torch.cuda.synchronize()
print("B=%d %dx%d SOL-%d: %.3f ms/step (graph replay), CG iters fwd mean %.1f" % (B, 2 * X, X, ms, (time.perf_counter() - t0) * 100, wl.trainer.solver_algorithmic_bytes()[2]))
prof = bench.profile_kernels(wl, 1e-6)
tot = sum(v["total_us"] for v in prof.values())
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_us"]):
    print("  %-34s calls %5d  avg %9.2f us  total %8.3f ms  %5.1f%%" % (k, v["calls"], v["avg_us"], v["total_us"] * 1e-3, 100 * v["total_us"] / tot))
print("  sum of kernel durations %.3f ms" % (tot * 1e-3))
