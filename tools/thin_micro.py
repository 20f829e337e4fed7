#!/usr/bin/env python
"""Times the four thin convolution launches of a C3 training step exactly as train.hip issues them
(3->32 first layer, 2->32 last backward-data layer, 32->2 last layer, 32->2 first backward-data layer)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sol_amd
from sol_amd import ops, _lib
from sol_amd._lib import ptr, stream, check

B, Y, X = 6, 128, 64
lib = _lib.load()
dev = "cuda"
x4 = torch.randn(B, Y, X, 4, device=dev); x4[..., 3] = 0
x32 = torch.randn(B, Y, X, 32, device=dev)
act = torch.randn(B, Y, X, 32, device=dev)
w_first = torch.randn(5, 5, 4, 32, device=dev) * 0.1
w_last = torch.randn(5, 5, 32, 2, device=dev) * 0.05
p_first = ops._pack(w_first, 4, 32, ops.CONV_FWD)
p_lastT = ops._pack(torch.randn(5, 5, 32, 4, device=dev) * 0.05, 4, 32, ops.CONV_BWD_DATA)     # 4 -> 32 data gradient
p_last = ops._pack(w_last, 32, 2, ops.CONV_FWD)
p_firstT = ops._pack(torch.randn(5, 5, 2, 32, device=dev) * 0.1, 32, 2, ops.CONV_BWD_DATA)       # 32 -> 2 data gradient
b32 = torch.randn(32, device=dev); b2 = torch.randn(2, device=dev)
y32 = torch.empty(B, Y, X, 32, device=dev); y2 = torch.empty(B, Y, X, 2, device=dev)
xam = ops.absmax_slots(x32)
yam = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device=dev)


def timeit(name, fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    print("%-42s %.2f us" % (name, e0.elapsed_time(e1) / n * 1e3))


s = stream()
timeit("first layer 4->32 lrelu (+ymax)", lambda: check(lib.sol_conv5x5_scaled(s, ptr(x4), ptr(p_first), ptr(b32), None, None, ptr(y32), B, Y, X, 4, 32, ops.EPI_LRELU, 0.3, None, ptr(yam))))
timeit("last bwd-data 4->32 dlrelu (+ymax)", lambda: check(lib.sol_conv5x5_scaled(s, ptr(x4), ptr(p_lastT), None, None, ptr(act), ptr(y32), B, Y, X, 4, 32, ops.EPI_DLRELU, 0.3, None, ptr(yam))))
timeit("last layer 32->2 (xmax)", lambda: check(lib.sol_conv5x5_scaled(s, ptr(x32), ptr(p_last), ptr(b2), None, None, ptr(y2), B, Y, X, 32, 2, ops.EPI_NONE, 0.3, ptr(xam), None)))
timeit("first bwd-data 32->2 (xmax)", lambda: check(lib.sol_conv5x5_scaled(s, ptr(x32), ptr(p_firstT), None, None, None, ptr(y2), B, Y, X, 32, 2, ops.EPI_NONE, 0.3, ptr(xam), None)))
P42 = ops._pack(torch.randn(5, 5, 4, 2, device=dev), 4, 2, ops.CONV_FWD)
P32 = ops._pack(torch.randn(5, 5, 32, 32, device=dev) * .05, 32, 32, ops.CONV_FWD)
timeit("32->32 lrelu (xmax, ymax)", lambda: check(lib.sol_conv5x5_scaled(s, ptr(x32), ptr(P32), ptr(b32), None, None, ptr(y32), B, Y, X, 32, 32, ops.EPI_LRELU, 0.3, ptr(xam), ptr(yam))))
timeit("first layer 4->32 lrelu, no ymax", lambda: check(lib.sol_conv5x5_scaled(s, ptr(x4), ptr(p_first), ptr(b32), None, None, ptr(y32), B, Y, X, 4, 32, ops.EPI_LRELU, 0.3, None, None)))
timeit("first layer 4->32 none, no bias/ymax", lambda: check(lib.sol_conv5x5_scaled(s, ptr(x4), ptr(p_first), None, None, None, ptr(y32), B, Y, X, 4, 32, ops.EPI_NONE, 0.3, None, None)))
timeit("4->2 none", lambda: check(lib.sol_conv5x5_scaled(s, ptr(x4), ptr(P42), None, None, None, ptr(y2), B, Y, X, 4, 2, ops.EPI_NONE, 0.3, None, None)))
