#!/usr/bin/env python
"""A/B of the thin-layer form of the dx-major convolution kernel (k_conv5x5_dx<R, 1>: <= 16 output channels; option conv_dx bit 2 = everywhere (15); the default 11 uses it in one-row-per-workgroup launches only)
against k_conv5x5_sb<1, 2> in ONE process: error against a float64 convolution for 2 / 3 / 16 output channels and every epilogue,
then ms per SOL-32 training step (correction-mode output layer + its data gradient) with either kernel, alternating.
    python tools/conv_thin_ab.py [--no-step]"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sol_amd  # noqa: E402
from sol_amd import ops, _lib  # noqa: E402

DEV = "cuda"
out = {"errors": []}
gen = torch.Generator(device="cpu").manual_seed(0)
for (B, H, W) in [(6, 128, 64), (2, 5, 64), (1, 7, 128), (1, 1, 64), (3, 64, 64)]:
    for cout in (2, 3, 16):
        x = torch.randn(B, H, W, 32, generator=gen).to(DEV)
        w = (torch.randn(5, 5, 32, cout, generator=gen) * 0.05).to(DEV)
        b = torch.randn(cout, generator=gen).to(DEV)
        res = torch.randn(B, H, W, cout, generator=gen).to(DEV)
        act = torch.randn(B, H, W, cout, generator=gen).to(DEV)
        packed = ops._pack(w, 32, cout, ops.CONV_FWD)
        xm = ops.absmax_slots(x)
        conv = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, padding=2).permute(0, 2, 3, 1)
        for name, (bb, rr, aa, epi) in {"bias+lrelu": (b, None, None, ops.EPI_LRELU), "res+dlrelu": (None, res, act, ops.EPI_DLRELU),
                                        "plain": (None, None, None, ops.EPI_NONE)}.items():
            ref = conv + (bb.double() if bb is not None else 0.0) + (rr.double() if rr is not None else 0.0)
            if epi == ops.EPI_LRELU:
                ref = torch.where(ref > 0, ref, 0.3 * ref)
            elif epi == ops.EPI_DLRELU:
                ref = ref * torch.where(aa.double() > 0, 1.0, 0.3)
            e = {}
            for dx in (11, 15):
                _lib.set_option("conv_dx", dx)
                ym = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)
                y = ops.conv5x5_scaled_raw(x, packed, bb, rr, aa, cout, epi, 0.3, xm, ym)
                torch.cuda.synchronize()
                e[dx] = float((y.double() - ref).norm() / ref.norm())
                assert float(ym.max().view(torch.float32).item()) == float(y.abs().max()), ("absmax", dx)
            with _lib.profile() as p:
                ops.conv5x5_scaled_raw(x, packed, bb, rr, aa, cout, epi, 0.3, xm, None)
            assert any("k_conv5x5_dx" in k and ", 1>" in k for k in p.kernels), p.kernels
            out["errors"].append({"shape": [B, H, W], "cout": cout, "epilogue": name, "err_sb": e[11], "err_dx": e[15]})
            print("%-14s cout %2d %-11s err vs float64: sb %.2e  dx %.2e" % ((B, H, W), cout, name, e[11], e[15]), flush=True)
            assert e[15] < 6e-7 and e[15] < 1.5 * e[11] + 1e-8

if "--no-step" not in sys.argv:
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    sys.argv = ["bench.py"]
    spec.loader.exec_module(bench)
    dev = torch.device("cuda", 0)
    for name, (Bq, Yq, Xq, n) in {"sol32_c3": (6, 128, 64, 20), "recipe_64x32_b3": (3, 64, 32, 20)}.items():
        wls = {}
        for dx in (11, 15):
            _lib.set_option("conv_dx", dx)              # read when the graph is captured (first step)
            wls[dx] = bench.Workload(sol_amd, dev, Bq, Yq, Xq, 32, 0)
            wls[dx].step(1e-6)
        res_ms = {11: [], 15: []}
        losses = {}
        for rep in range(8):
            for dx in ((11, 15) if rep % 2 == 0 else (15, 11)):
                _lib.set_option("conv_dx", dx)
                sec, loss, _ = bench.timed_steps(wls[dx], 1e-6, n, 2, torch.cuda.synchronize)
                res_ms[dx].append(sec / n * 1e3)
                losses[dx] = loss
        out[name] = res_ms
        print("%s medians: thin layers on sb %.3f ms, on dx %.3f ms   (losses %r)" % (name, statistics.median(res_ms[11]), statistics.median(res_ms[15]), losses), flush=True)
_lib.set_option("conv_dx", 11)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "conv_thin_ab.json"), "w") as f:
    json.dump(out, f, indent=1)
