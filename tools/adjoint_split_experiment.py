#!/usr/bin/env python
"""Round-6 pricing of the fused adjoint + weight-gradient launch (k_karman_bwd_bww, 32 launches per C3 step): how long is each half
alone INSIDE the replayed training step, what would accumulators that stay resident across the steps save at most, and does the
adjoint's load phase recover when the gradient workgroups start late?  Library option dbg_skip (timing experiments, results invalid):
  1024  every step overwrites its weight-gradient partial slice (no read of the old partials)
  2048  gradient workgroups sleep ~3.4 us before their prologue
  4096  gradient workgroups end at once      (adjoint alone, cold operands, in the pipeline)
  8192  adjoint workgroups end at once       (gradient half alone)
Prints ms per step (median of per-step HIP events, alternating variants) and the fused launch's average duration from an eager sweep.
GPU tool: gpurun -- 'python tools/adjoint_split_experiment.py [out.json]'."""
import json
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
import sol_amd                                  # noqa: E402
from sol_amd import _lib                        # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    variants = [("product", 0), ("overwrite partials every step", 1024), ("gradient WGs start 3.4 us late", 2048), ("adjoint alone", 4096),
                ("gradient half alone", 8192), ("gradient alone + overwrite", 8192 | 1024)]
    wls = {}
    for name, v in variants:
        _lib.set_option("dbg_skip", v)          # read when the graph is captured (the kernel arguments carry it)
        wl = bench.Workload(sol_amd, dev, 6, 128, 64, 32, 0)
        for _ in range(3):
            wl.step(1e-6)
        torch.cuda.synchronize()
        with _lib.profile() as p:
            wl.trainer.fwd_bwd(wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx, want_final=True, eager=True)
        k = {n.strip("()"): (c, t / max(c, 1)) for n, (c, t) in p.kernels.items()}
        wls[name] = (wl, k.get("k_karman_bwd_bww"))
    _lib.set_option("dbg_skip", 0)
    times = {name: [] for name, _ in variants}
    for rnd in range(4):
        for name, _ in variants:
            wl = wls[name][0]
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
            wl.step(1e-6)
            for i in range(8):
                evs[i].record()
                wl.step(1e-6)
            evs[8].record()
            torch.cuda.synchronize()
            times[name] += [evs[i].elapsed_time(evs[i + 1]) for i in range(8)]
    out = {}
    base = None
    for name, _ in variants:
        t = sorted(times[name])
        med = t[len(t) // 2]
        base = med if base is None else base
        ker = wls[name][1]
        out[name] = {"ms_per_step_median": med, "delta_ms": med - base, "fused_launch_us_eager": ker[1] if ker else None}
        print("%-36s %.3f ms per step (%+.3f)   k_karman_bwd_bww %.1f us (eager sweep, per-launch events)" % (name, med, med - base, ker[1] if ker else float("nan")))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
