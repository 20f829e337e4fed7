#!/usr/bin/env python
"""Phase timeline of the dx-major 32 -> 32 convolution launch (csrc/conv5x5_dx.hip) next to k_conv5x5_sb<2, 2>, C3 shape.
    python tools/ab_lib.py --build dxprof conv5x5_dx.hip:-DSOL_CONV_PROF conv5x5_sb.hip:-DSOL_CONV_PROF     (needs hipcc; no GPU)
    python tools/conv_dx_probe.py [B H W]                                                                      (on the GPU box)
The stamps perturb the kernels (a scalar load, s_memrealtime and a store by thread 0 each): read DIFFERENCES between phases."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SOL_HIP_LIB"] = os.path.join(ROOT, "solver-in-the-loop_amd", "lib", "libsol_dxprof.so")
import ctypes as C
import numpy as np
import torch
import sol_amd
from sol_amd import ops, _lib
from sol_amd._lib import ptr, stream, check

lib = _lib.load()
for f in (lib.sol_conv_dx_prof_set, lib.sol_conv_prof_set):
    f.argtypes = [C.c_void_p, C.c_uint]
B, Y, X = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (6, 128, 64)      # e.g. 3 32 64: the 64x32 recipe's CNN shape (one row per workgroup)
dev = "cuda"
x = torch.randn(B, Y, X, 32, device=dev)
res = torch.randn(B, Y, X, 32, device=dev)
w = torch.randn(5, 5, 32, 32, device=dev) * 0.05
packed = ops._pack(w, 32, 32, ops.CONV_FWD)
bias = torch.randn(32, device=dev)
y = torch.empty_like(x)
xam = ops.absmax_slots(x)
yam = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device=dev)
rows = B * Y
nwg = (rows + 2) // 3 if (rows + 2) // 3 >= 128 else rows      # rows per workgroup as the launchers choose them
nwg = (nwg + 7) // 8 * 8 if nwg > 64 else nwg
st = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
junk = torch.empty(64 << 20, dtype=torch.float32, device=dev)      # 256 MB: flushes L2 / MALL between launches ("cold" = the training pipeline's state)
call = lambda: check(lib.sol_conv5x5_scaled(stream(), ptr(x), ptr(packed), ptr(bias), ptr(res), None, ptr(y), B, Y, X, 32, 32,
                                             ops.EPI_LRELU, 0.3, ptr(xam), ptr(yam)))
NAMES = {1: [(0, "start"), (1, "requests out"), (2, "scale known"), (12, "rows+w written"), (3, "prologue barrier"), (4, "dx0"), (5, "dx1"), (6, "dx2"), (7, "dx3"), (8, "dx4"),
             (9, "stores issued"), (10, "absmax"), (11, "drained")],
         0: [(0, "start"), (10, "scale known"), (1, "prologue"), (2, "dy0"), (3, "dy1"), (4, "dy2"), (5, "dy3"), (6, "dy4"), (7, "stores issued"), (8, "absmax"), (9, "drained")]}
for dx in (0, 1):
    _lib.set_option("conv_dx", dx)
    setter = lib.sol_conv_dx_prof_set if dx else lib.sol_conv_prof_set
    for cold in (False, True):
        for _ in range(3):
            call()
        assert setter(C.c_void_p(st.data_ptr()), 1) == 0
        torch.cuda.synchronize()
        acc = []
        for rep in range(5):
            if cold:
                junk.fill_(1.0)
            call(); torch.cuda.synchronize()
            s = st.cpu().numpy().reshape(nwg, 16).astype(np.int64)
            acc.append(s - s[:, 0].min())
        s = np.median(np.stack(acc[1:]), axis=0)
        print("conv_dx=%d %s: us after the first workgroup start, min / median / max over %d workgroups" % (dx, "COLD (caches flushed)" if cold else "warm", nwg))
        for k, n in NAMES[dx]:
            v = s[:, k] * 0.01
            print("  %-18s %6.2f %6.2f %6.2f" % (n, v.min(), np.median(v), v.max()))
        assert setter(None, 1) == 0
        torch.cuda.synchronize()
