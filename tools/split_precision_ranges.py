#!/usr/bin/env python
"""Error of the split-precision 32->32 convolution kernels (fp16x3 with per-tensor power-of-two scale, bf16x6) relative to
the strict fp32-MFMA kernel, against a float64 convolution, for inputs with a prescribed WITHIN-TENSOR dynamic range:
the left half of every image row is scaled by `ratio` (1e-3, 1e-5, 2^-18 .. 2^-22: around the documented cliff of the fp16
lo plane, 2^-19 of the tensor maximum).  Per decile of |reference|: rms and max error ratios.
Output: JSON (the bounds of tests/test_gpu_parity.py::test_split_conv_error_per_decile_stays_in_the_measured_envelope come from here)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch            # noqa: E402
import sol_amd          # noqa: E402
from sol_amd import ops, _lib   # noqa: E402

DEV = "cuda"
gen = torch.Generator().manual_seed(5)
B, Y, X = 2, 64, 64
cases = {"normal": torch.randn(B, Y, X, 32, generator=gen, dtype=torch.float32),
         "heavy": torch.randn(B, Y, X, 32, generator=gen, dtype=torch.float32) * torch.exp(2.0 * torch.randn(B, Y, X, 32, generator=gen, dtype=torch.float32))}
for name, ratio in (("mixed_1e-3", 1e-3), ("mixed_1e-5", 1e-5), ("mixed_2^-18", 2.0 ** -18), ("mixed_2^-19", 2.0 ** -19),
                    ("mixed_2^-20", 2.0 ** -20), ("mixed_2^-22", 2.0 ** -22)):
    m = torch.randn(B, Y, X, 32, generator=gen, dtype=torch.float32)
    m[:, :, :32] *= ratio
    cases[name] = m
w = (torch.randn(5, 5, 32, 32, generator=gen, dtype=torch.float32) * 0.05).to(DEV)
bias = torch.zeros(32, dtype=torch.float32, device=DEV)
packed = ops._pack(w, 32, 32, ops.CONV_FWD)
out = {}
for name, x in cases.items():
    x = x.float().to(DEV)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, padding=2).permute(0, 2, 3, 1)
    _lib.set_option("conv_precision", 0)
    y_h = ops.conv5x5_scaled_raw(x, packed, bias, None, None, 32, ops.EPI_NONE, 0.3, ops.absmax_slots(x))
    y_b = ops.conv5x5_raw(x, packed, bias, None, None, 32, ops.EPI_NONE, 0.3)
    _lib.set_option("conv_precision", 2)
    y_f = ops.conv5x5_raw(x, packed, bias, None, None, 32, ops.EPI_NONE, 0.3)
    _lib.set_option("conv_precision", 0)
    order = ref.abs().reshape(-1).argsort()
    n = order.numel()
    rel = lambda a: float((a.double() - ref).norm() / ref.norm())
    rec = {"rel_l2": {"fp16x3": rel(y_h), "bf16x6": rel(y_b), "fp32": rel(y_f)}, "deciles": []}
    # also the small half alone (pixels 2..29 of each row only see scaled inputs)
    small = (slice(None), slice(None), slice(2, 30))
    rs = lambda a: float((a.double()[small] - ref[small]).norm() / ref[small].norm())
    rec["rel_l2_small_half"] = {"fp16x3": rs(y_h), "bf16x6": rs(y_b), "fp32": rs(y_f)}
    for q in range(10):
        idx = order[q * n // 10:(q + 1) * n // 10]
        r = ref.reshape(-1)[idx]
        e = {k: (v.double().reshape(-1)[idx] - r) for k, v in (("fp16x3", y_h), ("bf16x6", y_b), ("fp32", y_f))}
        rms = {k: float(v.pow(2).mean().sqrt()) for k, v in e.items()}
        mx = {k: float(v.abs().max()) for k, v in e.items()}
        rec["deciles"].append({"rms_ratio_fp16x3": rms["fp16x3"] / rms["fp32"], "rms_ratio_bf16x6": rms["bf16x6"] / rms["fp32"],
                               "max_ratio_fp16x3": mx["fp16x3"] / mx["fp32"], "max_ratio_bf16x6": mx["bf16x6"] / mx["fp32"]})
    rec["worst"] = {k: max(d[k] for d in rec["deciles"]) for k in rec["deciles"][0]}
    out[name] = rec
print(json.dumps(out, indent=1))
