#!/usr/bin/env python
"""Debug: are back-to-back training steps (graph replay -> loss sum -> Adam -> graph replay, no host sync) equal to the
same steps with a host sync in between?  The engine is deterministic, so the weights must be BIT identical."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, sol_amd
from sol_amd import _lib
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4

def run(tag, sync, stream=None, graph_stream=0, use_graph=True):
    _lib.set_option("graph_stream", graph_stream)
    wl = bench.Workload(sol_amd, dev, 6, 128, 64, 32, 0, use_graph=use_graph)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    losses = []
    with ctx:
        for _ in range(N):
            losses.append(wl.step(lr))
            if sync:
                torch.cuda.synchronize()
    torch.cuda.synchronize()
    p = wl.net.params.detach().clone()
    print("%-34s last loss %.6g  finite %s" % (tag, float(losses[-1]), bool(torch.isfinite(p).all())), flush=True)
    return p, [float(l) for l in losses]

ref, lref = run("synced, null stream", True)
for tag, kw in [("unsynced, null stream", dict(sync=False)),
                ("unsynced, null stream (again)", dict(sync=False)),
                ("unsynced, side torch stream", dict(sync=False, stream=torch.cuda.Stream())),
                ("unsynced, internal graph stream", dict(sync=False, graph_stream=1)),
                ("unsynced, eager (no graph)", dict(sync=False, use_graph=False))]:
    p, l = run(tag, **kw)
    bad = [i for i, (a, b) in enumerate(zip(l, lref)) if a != b]
    print("    weights bit-identical to synced run: %s   first differing step: %s" % (bool(torch.equal(p, ref)), bad[:1]), flush=True)
