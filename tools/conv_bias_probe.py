#!/usr/bin/env python
"""Is the error of the three 32 -> 32 convolution arithmetics (fp16x3 split, bf16x6, fp32 MFMA) against a float64 convolution
COHERENT with the result (a systematic gain epsilon: y = (1 + eps) y_ref + noise) or incoherent noise?  Only the coherent part
accumulates linearly through the 11 layers x 32 unrolled steps of a SOL-32 training step (the per-step losses of every arithmetic
drift by a few 1e-7 with one sign, tools/precision_at_c3.py).  Prints per case: relative L2 error, eps = <y - ref, ref> / <ref, ref>,
and the mean signed error / mean |ref|, for random-normal and for post-LeakyReLU (one-signed-heavy) inputs.
GPU tool: gpurun -- 'python tools/conv_bias_probe.py'."""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import sol_amd                                   # noqa: E402,F401
from sol_amd import ops, _lib                    # noqa: E402

DEV = "cuda"


def outputs(x, w, bias):
    packed = ops._pack(w, 32, 32, ops.CONV_FWD)
    saved = _lib.get_option("conv_precision")
    try:
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), bias.double(), padding=2).permute(0, 2, 3, 1)
        _lib.set_option("conv_precision", 0)
        y_h = ops.conv5x5_scaled_raw(x, packed, bias, None, None, 32, ops.EPI_NONE, 0.3, ops.absmax_slots(x))
        y_b = ops.conv5x5_raw(x, packed, bias, None, None, 32, ops.EPI_NONE, 0.3)
        _lib.set_option("conv_precision", 2)
        y_f = ops.conv5x5_raw(x, packed, bias, None, None, 32, ops.EPI_NONE, 0.3)
    finally:
        _lib.set_option("conv_precision", saved)
    return ref, {"fp16x3": y_h, "bf16x6": y_b, "fp32": y_f}


def main():
    gen = torch.Generator().manual_seed(11)
    B, Y, X = 6, 128, 64
    w = (torch.randn(5, 5, 32, 32, generator=gen) * 0.035).to(DEV)
    bias = (torch.randn(32, generator=gen) * 0.01).to(DEV)
    xn = torch.randn(B, Y, X, 32, generator=gen).to(DEV)
    cases = {"normal": xn, "leaky_relu(normal)": torch.nn.functional.leaky_relu(xn, 0.3), "abs(normal)": xn.abs(),
             "smooth positive": (1.0 + 0.1 * xn)}
    for wname, ww in (("w normal", w), ("w positive", w.abs())):
        for name, x in cases.items():
            ref, ys = outputs(x.contiguous(), ww.contiguous(), bias)
            print("== %s, %s" % (name, wname))
            for k, y in ys.items():
                e = y.double() - ref
                rel = float(e.norm() / ref.norm())
                eps = float((e * ref).sum() / (ref * ref).sum())
                msg = float(e.mean() / ref.abs().mean())
                print("   %-7s rel L2 %.2e   coherent gain eps %+.2e   mean signed error / mean|ref| %+.2e   incoherent rest %.2e"
                      % (k, rel, eps, msg, float((e - eps * ref).norm() / ref.norm())))


if __name__ == "__main__":
    main()
