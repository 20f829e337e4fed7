#!/usr/bin/env python
"""Accuracy + time of the 32->32 conv forward against a float64 torch convolution on the GPU.
Run under SOL_CONV_NO_SB=1 (fp32 MFMA), default (split-bf16 x6) and SOL_CONV_SPLIT=3 to compare."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sol_amd
from sol_amd import ops

torch.manual_seed(0)
for (B, Y, X, cout) in [(6, 128, 64, 32), (2, 128, 64, 2), (1, 16, 64, 32)]:
    x = torch.randn(B, Y, X, 32, device="cuda")
    w = torch.randn(5, 5, 32, cout, device="cuda") * 0.05
    bias = torch.randn(cout, device="cuda") * 0.1
    packed = ops._pack(w, 32, cout, ops.CONV_FWD)
    y = ops.conv5x5_raw(x, packed, bias, None, None, cout, ops.EPI_NONE, 0.3)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), bias.double(), padding=2).permute(0, 2, 3, 1)
    err = ((y.double() - ref).norm() / ref.norm()).item()
    mx = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn = lambda: ops.conv5x5_raw(x, packed, bias, None, None, cout, ops.EPI_LRELU, 0.3)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("B%d %dx%d 32->%d: rel L2 %.3e  max/max %.3e  %.1f us  %.1f TF (env NO_SB=%s SPLIT=%s)" % (
        B, Y, X, cout, err, mx, us, 2.0 * 25 * 32 * cout * B * Y * X / us / 1e6,
        os.environ.get("SOL_CONV_NO_SB"), os.environ.get("SOL_CONV_SPLIT")))

# ---- backward-weight 32x32 against float64 -------------------------------------------------
import ctypes as C
from sol_amd import _lib
from sol_amd._lib import ptr, stream, check
lib = _lib.load()
for (B, Y, X) in [(2, 128, 64), (6, 128, 64), (24, 128, 64)]:
    x = torch.randn(B, Y, X, 32, device="cuda")
    dz = torch.randn(B, Y, X, 32, device="cuda")
    nws = lib.sol_conv5x5_bwd_weight_ws_floats(B, Y, X, 32, 32)
    part = torch.zeros(nws, device="cuda")
    dw = torch.zeros(5, 5, 32, 32, device="cuda")
    db = torch.zeros(32, device="cuda")
    check(lib.sol_conv5x5_bwd_weight(stream(), ptr(x), ptr(dz), ptr(part), B, Y, X, 32, 32))
    check(lib.sol_conv5x5_bwd_weight_reduce(stream(), ptr(part), ptr(dw), ptr(db), B, Y, X, 32, 32, 0))
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(False)
    wd = torch.zeros(32, 32, 5, 5, dtype=torch.float64, device="cuda", requires_grad=True)
    bd = torch.zeros(32, dtype=torch.float64, device="cuda", requires_grad=True)
    yy = torch.nn.functional.conv2d(xd, wd, bd, padding=2)
    (yy * dz.double().permute(0, 3, 1, 2)).sum().backward()
    ref = wd.grad.permute(2, 3, 1, 0)
    e_w = ((dw.double() - ref).norm() / ref.norm()).item()
    e_b = ((db.double() - bd.grad).norm() / bd.grad.norm()).item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn = lambda: check(lib.sol_conv5x5_bwd_weight(stream(), ptr(x), ptr(dz), ptr(part), B, Y, X, 32, 32))
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("bww B%d %dx%d: dw rel L2 %.3e  db rel %.3e  %.1f us  %.1f TF" % (B, Y, X, e_w, e_b, us, 2.0 * 25 * 1024 * B * Y * X / us / 1e6))

# ---- fp16 three-product path (sol_conv5x5_scaled with the absmax of x) ----------------------------
print("fp16 x3 path (sol_conv5x5_scaled):")
for (B, Y, X, cout, dist) in [(6, 128, 64, 32, "normal"), (6, 128, 64, 32, "wide"), (2, 128, 64, 2, "normal"), (1, 16, 64, 32, "tiny")]:
    x = torch.randn(B, Y, X, 32, device="cuda")
    if dist == "wide":
        x = x * torch.exp(3 * torch.randn_like(x))
    if dist == "tiny":
        x = x * 1e-7
    w = torch.randn(5, 5, 32, cout, device="cuda") * 0.05
    bias = torch.randn(cout, device="cuda") * (1e-9 if dist == "tiny" else 0.1)
    packed = ops._pack(w, 32, cout, ops.CONV_FWD)
    xmax = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device="cuda")
    xmax[0] = int(x.abs().max().view(torch.int32).item())
    ymax = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device="cuda")
    y = torch.empty(B, Y, X, cout, device="cuda")
    call = lambda: check(lib.sol_conv5x5_scaled(stream(), ptr(x), ptr(packed), ptr(bias), None, None, ptr(y), B, Y, X, 32, cout,
                                                 ops.EPI_NONE, 0.3, ptr(xmax), ptr(ymax)))
    call()
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), bias.double(), padding=2).permute(0, 2, 3, 1)
    err = ((y.double() - ref).norm() / ref.norm()).item()
    got_max = float(ymax.max().view(torch.float32).item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    call(); torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("  B%d %dx%d 32->%d %-6s: rel L2 %.3e  ymax %.6g (true %.6g)  %.1f us  %.1f TF" % (
        B, Y, X, cout, dist, err, got_max, float(y.abs().max()), us, 2.0 * 25 * 32 * cout * B * Y * X / us / 1e6))
