#!/usr/bin/env python
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, sol_amd
from sol_amd import _lib
dev = torch.device("cuda", 0)
ms = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for mode, graph in ((1, True), (1, False), (0, True), (1, True)):
    _lib.set_option("cnn_persistent", mode)
    wl = bench.Workload(sol_amd, dev, 6, 128, 64, ms, 0, use_graph=graph)
    ts = []
    for k in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        wl.step(1e-6)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("persistent=%d graph=%s per-step ms (synced): %s  captures %d" % (mode, graph, " ".join("%.2f" % t for t in ts), wl.trainer._captures), flush=True)
    del wl
