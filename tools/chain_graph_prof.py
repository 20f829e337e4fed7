#!/usr/bin/env python
"""graph-replay training steps with the persistent chain on (for rocprofv3 --kernel-trace: in-graph kernel durations)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, sol_amd, bench
from sol_amd import _lib
_lib.set_option("cnn_persistent", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
wl = bench.Workload(sol_amd, torch.device("cuda", 0), 6, 128, 64, 32, 0)
for _ in range(6):
    wl.step(1e-6)
torch.cuda.synchronize()
