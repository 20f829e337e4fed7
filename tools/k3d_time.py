#!/usr/bin/env python
"""karman-3d (128 x 64 x 64) timing: solver step and CNN forward per simulation step, per-kernel table (sol_prof_* events).
Usage: python tools/k3d_time.py [B] [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch          # noqa: E402
import sol_amd        # noqa: E402
from sol_amd import karman3d as k3, synthetic   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
Y, X, Z = 128, 64, 64
dev = "cuda"
t0 = time.time()
sc = k3.Scene3D(Y, X, Z, device=dev)
print("scene setup %.1f s" % (time.time() - t0), flush=True)
net = k3.MarsMoon3D(device=dev)
w = net.get_weights()
w[22] = w[22] * 0.01
net.set_weights(w)
ro = k3.Karman3DRollout(net, sc, B, (0.2, 0.2, 0.2), synthetic.STD_RE)
gen = torch.Generator().manual_seed(1)
f = lambda *s: torch.randn(*s, generator=gen)
d = torch.rand(B, Y, X, Z, generator=gen).to(dev)
vy = (1.0 + 0.1 * f(B, Y + 1, X, Z)).to(dev)
vx = (0.1 * f(B, Y, X + 1, Z)).to(dev)
vz = (0.1 * f(B, Y, X, Z + 1)).to(dev)
re = synthetic.reynolds(B).float().to(dev)
st = ro.step(d, vy, vx, vz, re)          # warm-up (packs the weights)
torch.cuda.synchronize()
out = {"B": B}
for tile, ftf in ((1, 1), (0, 1), (0, 0)):
    sol_amd._lib.set_option("k3d_tile", tile)
    sol_amd._lib.set_option("k3d_fused_tf", ftf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s2 = st
    e0.record()
    for _ in range(steps):
        s2 = ro.sim.step(*s2, re)
    e1.record()
    torch.cuda.synchronize()
    out["solver_ms_tile%d_fusedtf%d" % (tile, ftf)] = e0.elapsed_time(e1) / steps
sol_amd._lib.set_option("k3d_tile", 0)
sol_amd._lib.set_option("k3d_fused_tf", 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rows in (3, 6, 8):                     # rows per workgroup of the one-launch Conv3D kernel; the default (8) is timed last and stays on
    sol_amd._lib.set_option("k3d_conv_rows", rows)
    ro.correction()
    e0.record()
    for _ in range(steps):
        ro.correction()
    e1.record()
    torch.cuda.synchronize()
    out["cnn_ms_rows%d" % rows] = e0.elapsed_time(e1) / steps
out["cnn_ms"] = out["cnn_ms_rows8"]
for dbg in (1, 8, 32, 64, 104):          # timing experiments (results invalid): 1 no taps, 8 no barrier, 32 no weights, 64 no rows
    sol_amd._lib.set_option("dbg_skip", dbg)
    ro.correction()
    e0.record()
    for _ in range(steps):
        ro.correction()
    e1.record()
    torch.cuda.synchronize()
    out["cnn_ms_dbg%d" % dbg] = e0.elapsed_time(e1) / steps
sol_amd._lib.set_option("dbg_skip", 0)
e0.record()
s2 = st
for _ in range(steps):
    s2 = ro.step(*s2, re)
e1.record()
torch.cuda.synchronize()
out["step_ms"] = e0.elapsed_time(e1) / steps
out["finite"] = bool(torch.isfinite(s2[1]).all())
with sol_amd._lib.profile() as p:
    ro.step(*st, re)
out["kernels"] = {k: {"calls": c, "us": round(t, 1)} for k, (c, t) in sorted(p.kernels.items(), key=lambda kv: -kv[1][1])}
flop = 2 * 125 * (4 * 32 + 10 * 32 * 32 + 32 * 3) * B * Y * X * Z
out["cnn_tflops_fp32_equiv"] = flop / (out["cnn_ms"] * 1e-3) / 1e12
print(json.dumps(out, indent=1))
