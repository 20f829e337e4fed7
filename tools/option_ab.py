#!/usr/bin/env python
"""Same-box A/B of a library option on the C3 training step (karman-2d 128x64 SOL-32, B = 6): one Workload per option value (the
option is read when the graph is captured), alternating blocks of timed replays, median ms per step per value and the per-kernel
averages of an eager sweep (per-launch HIP events) for the kernels whose name contains one of the given substrings.
GPU tool: gpurun -- 'python tools/option_ab.py fwd_bands 0,1 [kernel-substring ...] [--out f.json] [--b B] [--ms M]'."""
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
    args = sys.argv[1:]
    out_path, B, ms = None, 6, 32
    for flag in ("--out", "--b", "--ms"):
        if flag in args:
            i = args.index(flag)
            val = args[i + 1]
            del args[i:i + 2]
            if flag == "--out":
                out_path = val
            elif flag == "--b":
                B = int(val)
            else:
                ms = int(val)
    opt, values, subs = args[0], [int(v) for v in args[1].split(",")], args[2:]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    old = _lib.get_option(opt)
    wls, kern = {}, {}
    for v in values:
        _lib.set_option(opt, v)
        wl = bench.Workload(sol_amd, dev, B, 128, 64, ms, 0)
        for _ in range(3):
            wl.step(1e-6)
        torch.cuda.synchronize()
        with _lib.profile() as p:
            wl.trainer.fwd_bwd(wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx, want_final=True, eager=True)
        kern[v] = {n.strip("()"): (c, t / max(c, 1)) for n, (c, t) in p.kernels.items() if not subs or any(s in n for s in subs)}
        kern[v]["sum_all_us"] = (sum(c for c, _ in p.kernels.values()), sum(t for _, t in p.kernels.values()))
        wls[v] = wl
    _lib.set_option(opt, old)
    times = {v: [] for v in values}
    for rnd in range(5):
        for v in values:
            wl = wls[v]
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
            wl.step(1e-6)
            for i in range(8):
                evs[i].record()
                wl.step(1e-6)
            evs[8].record()
            torch.cuda.synchronize()
            times[v] += [evs[i].elapsed_time(evs[i + 1]) for i in range(8)]
    out = {}
    for v in values:
        t = sorted(times[v])
        med = t[len(t) // 2]
        out[str(v)] = {"ms_per_step_median": med, "p10": t[len(t) // 10], "p90": t[9 * len(t) // 10], "kernels_eager_us": {k: {"calls": c, "avg_us": a} for k, (c, a) in kern[v].items()}}
        print("%s = %d: median %.3f ms per step (p10 %.3f, p90 %.3f)" % (opt, v, med, t[len(t) // 10], t[9 * len(t) // 10]))
        for k, (c, a) in sorted(kern[v].items()):
            print("      %-60s %5d x %9.2f us" % (k[:60], c, a))
    if out_path:
        with open(out_path, "w") as f:
            json.dump({"option": opt, "B": B, "msteps": ms, "values": out}, f, indent=1)


if __name__ == "__main__":
    main()
