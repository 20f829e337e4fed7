#!/usr/bin/env python
"""Per-kernel table of one training step of the reference's own recipe (64x32, B = 3, SOL-32; karman-2d/Makefile:78-80) or any
other shape: one EAGER forward + reverse sweep with HIP events around every launch (bench.profile_kernels), next to the replayed
graph's ms per step.
    python tools/recipe_profile.py [B Y X msteps]"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = [int(a) for a in sys.argv[1:5]] or [3, 64, 32, 32]
sys.argv = ["bench.py"]
import torch  # noqa: E402
import sol_amd  # noqa: E402
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
B, Y, X, ms = args
if os.environ.get("SOL_AB_CONV_DX"):
    from sol_amd import _lib
    _lib.set_option("conv_dx", int(os.environ["SOL_AB_CONV_DX"]))
wl = bench.Workload(sol_amd, torch.device("cuda", 0), B, Y, X, ms, 0)
sec, loss, _ = bench.timed_steps(wl, 1e-6, 20, 3, torch.cuda.synchronize)
print("B %d  %dx%d  SOL-%d: %.3f ms per step (replayed graph), loss %.5f" % (B, Y, X, ms, sec / 20 * 1e3, loss))
tab = bench.profile_kernels(wl, 1e-6)
tot = sum(v["total_us"] for v in tab.values())
print("eager sweep: %.3f ms of kernel time in %d launches" % (tot * 1e-3, sum(v["calls"] for v in tab.values())))
for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["total_us"]):
    print("  %-44s %5d x %8.2f us = %8.1f us  %5.1f %%" % (k[:44], v["calls"], v["avg_us"], v["total_us"], 100 * v["total_us"] / tot))
