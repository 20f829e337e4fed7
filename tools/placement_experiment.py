#!/usr/bin/env python
"""Does the C3 step time depend on WHERE the trainer's buffers lie?  One process, one box: the same workload is built several times with
dummy allocations of different sizes in front of it (the caching allocator then hands out other addresses); ms per step (median of per-step
HIP events) of every instance, alternating.  If the instances differ by more than their own p10..p90, buffer placement (HBM channel / bank
interleave, TLB reach) is a component of the run-to-run spread that bench.py's clock, dispatch and bandwidth probes cannot see.
GPU tool: gpurun -- 'python tools/placement_experiment.py'."""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
import sol_amd                                  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    pads = [0, 1 << 20, 37 << 20, 256 << 20, (1 << 30) + (13 << 12)]
    keep, wls = [], []
    for p in pads:
        if p:
            keep.append(torch.empty(p, dtype=torch.uint8, device=dev))
        wl = bench.Workload(sol_amd, dev, 6, 128, 64, 32, 0)
        for _ in range(3):
            wl.step(1e-6)
        wls.append(wl)
        print("instance pad %11d B: workspace at 0x%x (%.2f GB)" % (p, wl.trainer.workspace.data_ptr(), wl.trainer.workspace.numel() * 4 / 2 ** 30))
    times = [[] for _ in wls]
    for rnd in range(4):
        for k, wl in enumerate(wls):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
            wl.step(1e-6)
            for i in range(8):
                evs[i].record()
                wl.step(1e-6)
            evs[8].record()
            torch.cuda.synchronize()
            times[k] += [evs[i].elapsed_time(evs[i + 1]) for i in range(8)]
    for k, t in enumerate(times):
        t = sorted(t)
        print("instance %d (pad %11d B): median %.3f ms per step, p10 %.3f, p90 %.3f" % (k, pads[k], t[len(t) // 2], t[len(t) // 10], t[9 * len(t) // 10]))


if __name__ == "__main__":
    main()
