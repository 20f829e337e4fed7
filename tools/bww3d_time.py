#!/usr/bin/env python
"""Time of one 32 -> 32 Conv3D weight gradient (sol_conv3d_bwd_weight: five passes + reduce) at 128 x 64 x 64.
Usage: python tools/bww3d_time.py   (SOL_HIP_LIB selects a tools/ab_lib.py variant, e.g. the BWW_DBG removal experiments)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sol_amd
from sol_amd import karman3d as k3
dev = "cuda"
gen = torch.Generator().manual_seed(0)
x = torch.randn(1, 128, 64, 64, 32, generator=gen, dtype=torch.float32).to(dev)
dz = (torch.randn(1, 128, 64, 64, 32, generator=gen, dtype=torch.float32) * 1e-3).to(dev)
xm, zm = k3._absmax(x), k3._absmax(dz)
for _ in range(2):
    k3.conv3d_bwd_weight(x, dz, 32, 32, xmax=xm, zmax=zm)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    k3.conv3d_bwd_weight(x, dz, 32, 32, xmax=xm, zmax=zm)
b.record()
torch.cuda.synchronize()
print(json.dumps({"lib": os.environ.get("SOL_HIP_LIB", "product"), "us_per_layer": a.elapsed_time(b) * 100.0}))
