#!/usr/bin/env python
"""Micro-driver for profiling the conv kernels: fwd 32->32 (+lrelu), bwd-weight 32x32 on a C3-sized
activation.  Usage: python tools/conv_micro.py [--n 20]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
import sol_amd
from sol_amd import ops, _lib
from sol_amd._lib import ptr, stream, check

p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=20)
p.add_argument("--batch", type=int, default=6)
p.add_argument("--res", type=int, default=64)
a = p.parse_args()
B, Y, X = a.batch, 2 * a.res, a.res
lib = _lib.load()
x = torch.randn(B, Y, X, 32, device="cuda")
dz = torch.randn(B, Y, X, 32, device="cuda")
w = torch.randn(5, 5, 32, 32, device="cuda") * 0.05
packed = ops._pack(w, 32, 32, ops.CONV_FWD)
bias = torch.zeros(32, device="cuda")
nws = lib.sol_conv5x5_bwd_weight_ws_floats(B, Y, X, 32, 32)
part = torch.zeros(nws, device="cuda")


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


flop = 2.0 * 25 * 32 * 32 * B * Y * X
t = timeit(lambda: ops.conv5x5_raw(x, packed, bias, None, None, 32, ops.EPI_LRELU, 0.3), a.n)
print("conv fwd 32->32: %.1f us  %.1f TF" % (t, flop / t / 1e6))
t = timeit(lambda: ops.conv5x5_raw(x, packed, None, dz, x, 32, ops.EPI_DLRELU, 0.3), a.n)
print("conv bwd-data-like (res+dlrelu): %.1f us  %.1f TF" % (t, flop / t / 1e6))
t = timeit(lambda: check(lib.sol_conv5x5_bwd_weight(stream(), ptr(x), ptr(dz), ptr(part), B, Y, X, 32, 32)), a.n)
print("conv bwd-weight 32x32: %.1f us  %.1f TF" % (t, flop / t / 1e6))
