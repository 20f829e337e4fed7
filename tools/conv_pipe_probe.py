#!/usr/bin/env python
"""Phase timeline of the 32 -> 32 convolution launches INSIDE a training step (the operands' real cache state: inputs just written
by the previous launch, weights of 20 different layers, residual / activation operands from HBM), for the dx-major kernel and for
k_conv5x5_sb<2, 2>.  Needs the stamped library:
    python tools/ab_lib.py --build dxprof conv5x5_dx.hip:-DSOL_CONV_PROF conv5x5_sb.hip:-DSOL_CONV_PROF
    python tools/conv_pipe_probe.py            (GPU box)
One EAGER SOL-32 step (B = 6, 128x64) per kernel: 320 forward + 320 backward-data launches, every launch's 256 workgroups x 16
stamps kept (the kernels index a [launch][workgroup][16] buffer by a device-side launch counter)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SOL_HIP_LIB"] = os.path.join(ROOT, "solver-in-the-loop_amd", "lib", "libsol_dxprof.so")
import ctypes as C
import importlib.util
import numpy as np
import torch
import sol_amd
from sol_amd import _lib

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
sys.argv = ["bench.py"]
spec.loader.exec_module(bench)
lib = _lib.load()
for f in (lib.sol_conv_dx_prof_set, lib.sol_conv_prof_set):
    f.argtypes = [C.c_void_p, C.c_uint]
dev = torch.device("cuda", 0)
NL, NWG = 640, 256
NAMES = {1: [(0, "start"), (1, "requests out"), (2, "scale known"), (12, "rows0-2+w written"), (3, "prologue barrier"), (4, "dx0"), (5, "dx1"), (6, "dx2"), (7, "dx3"), (8, "dx4"),
             (9, "stores issued"), (10, "absmax"), (11, "drained")],
         0: [(0, "start"), (10, "scale known"), (1, "prologue"), (2, "dy0"), (3, "dy1"), (4, "dy2"), (5, "dy3"), (6, "dy4"), (7, "stores issued"), (8, "absmax"), (9, "drained")]}
for dx in (0, 1):
    _lib.set_option("conv_dx", dx)
    wl = bench.Workload(sol_amd, dev, 6, 128, 64, 32, 0, use_graph=False)
    tr = wl.trainer
    for _ in range(2):
        tr.fwd_bwd(wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx, want_final=True, eager=True)
    torch.cuda.synchronize()
    st = torch.zeros(NL * NWG * 16, dtype=torch.int64, device=dev)
    setter = lib.sol_conv_dx_prof_set if dx else lib.sol_conv_prof_set
    assert setter(C.c_void_p(st.data_ptr()), NL) == 0
    torch.cuda.synchronize()
    tr.fwd_bwd(wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx, want_final=True, eager=True)
    torch.cuda.synchronize()
    assert setter(None, 1) == 0
    s = st.cpu().numpy().reshape(NL, NWG, 16).astype(np.int64)
    last = NAMES[dx][-1][0]
    for name, sl in (("forward (launches 0..319)", slice(0, 320)), ("backward-data (launches 320..639)", slice(320, 640))):
        part = s[sl]
        t0 = part[:, :, 0].min(axis=1)[:, None]
        print("conv_dx=%d %s: us after the launch's first workgroup start -- median over launches of [min / median / max over workgroups]" % (dx, name))
        for k, n in NAMES[dx]:
            v = (part[:, :, k] - t0) * 0.01
            print("  %-20s %6.2f %6.2f %6.2f" % (n, np.median(v.min(axis=1)), np.median(np.median(v, axis=1)), np.median(v.max(axis=1))))
        span = (part[:, :, last].max(axis=1) - part[:, :, 0].min(axis=1)) * 0.01
        print("  launch span (first start -> last drained): median %.2f  p10 %.2f  p90 %.2f us" % (np.median(span), np.percentile(span, 10), np.percentile(span, 90)))
    del wl, tr
