#!/usr/bin/env python
"""A/B of the dx-major 32 -> 32 convolution kernel (csrc/conv5x5_dx.hip, option conv_dx = 1) against k_conv5x5_sb<2, 2> (conv_dx = 0)
in ONE process on one box: (1) error of both against a float64 convolution for every epilogue form and several shapes (image-
straddling workgroups, odd heights, two column blocks, the transposed 64x32 recipe), (2) microseconds per launch under graph replay
of 50 back-to-back launches, (3) ms per SOL-32 training step with either kernel (two trainers, alternating).
    python tools/conv_dx_ab.py [--no-step]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sol_amd  # noqa: E402
from sol_amd import ops, _lib  # noqa: E402

DEV = "cuda"
out = {}
DXV = int(os.environ.get("SOL_AB_CONV_DX", "11"))      # the option value the "dx" leg runs with (11: default; 3: without the split one-row form)


def ref64(x, w, b, res, act, epi, slope=0.3):
    y = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None if b is None else b.double(), padding=2).permute(0, 2, 3, 1)
    if res is not None:
        y = y + res.double()
    if epi == ops.EPI_LRELU:
        y = torch.where(y > 0, y, slope * y)
    elif epi == ops.EPI_DLRELU:
        y = y * torch.where(act.double() > 0, 1.0, slope)
    return y


def run(dx, x, packed, b, res, act, epi, xm):
    _lib.set_option("conv_dx", DXV if dx else 0)
    ym = torch.zeros(256, dtype=torch.int32, device=DEV)
    y = ops.conv5x5_scaled_raw(x, packed, b, res, act, 32, epi, 0.3, xm, ym)
    torch.cuda.synchronize()
    return y, ym


errs = []
gen = torch.Generator(device="cpu").manual_seed(0)
for (B, H, W) in [(6, 128, 64), (3, 32, 64), (2, 5, 64), (1, 7, 128), (4, 3, 64), (1, 1, 64), (5, 2, 64)]:
    x = torch.randn(B, H, W, 32, generator=gen).to(DEV)
    w = (torch.randn(5, 5, 32, 32, generator=gen) * 0.05).to(DEV)
    b = torch.randn(32, generator=gen).to(DEV)
    res = torch.randn(B, H, W, 32, generator=gen).to(DEV)
    act = torch.randn(B, H, W, 32, generator=gen).to(DEV)
    packed = ops._pack(w, 32, 32, ops.CONV_FWD)
    xm = ops.absmax_slots(x)
    for name, (bb, rr, aa, epi) in {"bias+lrelu": (b, None, None, ops.EPI_LRELU), "res+lrelu": (b, res, None, ops.EPI_LRELU),
                                    "res+dlrelu": (None, res, act, ops.EPI_DLRELU), "plain": (None, None, None, ops.EPI_NONE)}.items():
        r = ref64(x, w, bb, rr, aa, epi)
        e = {}
        for dx in (0, 1):
            y, ym = run(dx, x, packed, bb, rr, aa, epi, xm)
            e[dx] = float((y.double() - r).norm() / r.norm())
            amax_pub = float(ym.view(torch.float32).max())
            assert abs(amax_pub - float(y.abs().max())) <= 1e-6 * amax_pub, ("absmax", dx, amax_pub, float(y.abs().max()))
        y0, _ = run(0, x, packed, bb, rr, aa, epi, xm)
        y1, _ = run(1, x, packed, bb, rr, aa, epi, xm)
        d01 = float((y1.double() - y0.double()).norm() / r.norm())
        errs.append({"shape": [B, H, W], "epilogue": name, "err_sb": e[0], "err_dx": e[1], "dx_vs_sb": d01})
        print("%-14s %-11s err vs float64: sb %.2e  dx %.2e   dx vs sb %.2e" % ((B, H, W), name, e[0], e[1], d01), flush=True)
        assert e[1] < 6e-7 and e[1] < 1.5 * e[0] + 1e-8, "dx kernel error"
out["errors"] = errs

# ---- microseconds per launch, 50 launches per graph replay ----
B, H, W = 6, 128, 64
x = torch.randn(B, H, W, 32, device=DEV)
y = torch.empty_like(x)
res = torch.randn(B, H, W, 32, device=DEV)
w = torch.randn(5, 5, 32, 32, device=DEV) * 0.02          # gain < 1 per layer: the 1000-launch ping-pong chain stays bounded
packed = ops._pack(w, 32, 32, ops.CONV_FWD)
b = torch.zeros(32, device=DEV)
xm = ops.absmax_slots(x)
ym = torch.zeros(256, dtype=torch.int32, device=DEV)
lib = _lib.load()
times = {}
for dx in (0, 1, 0, 1):
    _lib.set_option("conv_dx", DXV if dx else 0)
    bufs = [x, y]

    def chain():
        for k in range(50):
            src, dst = bufs[k & 1], bufs[(k + 1) & 1]
            _lib.check(lib.sol_conv5x5_scaled(_lib.stream(), _lib.ptr(src), _lib.ptr(packed), _lib.ptr(b), _lib.ptr(res), None, _lib.ptr(dst),
                                              B, H, W, 32, 32, ops.EPI_LRELU, 0.3, _lib.ptr(xm), _lib.ptr(ym)))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        chain()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with _lib.no_gc_during_capture(), torch.cuda.graph(g):
        chain()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 / 50 * 1e3
    times.setdefault(dx, []).append(us)
    print("conv_dx=%d: %.2f us per launch (graph replay, 50 launches, res + lrelu + absmax)" % (dx, us), flush=True)
out["us_per_launch"] = times

# ---- the same chain as the training step sees it: ten different layers' weights in turn and a residual operand that is HBM-cold
#      (50 different tensors = 315 MB, more than the 256 MB memory-side cache) ----
packs = [ops._pack(torch.randn(5, 5, 32, 32, device=DEV) * 0.02, 32, 32, ops.CONV_FWD) for _ in range(10)]
ress = [torch.randn(B, H, W, 32, device=DEV) for _ in range(50)]
times_cold = {}
for dx in (0, 1, 0, 1):
    _lib.set_option("conv_dx", DXV if dx else 0)
    bufs = [x, y]

    def chain2():
        for k in range(50):
            src, dst = bufs[k & 1], bufs[(k + 1) & 1]
            _lib.check(lib.sol_conv5x5_scaled(_lib.stream(), _lib.ptr(src), _lib.ptr(packs[k % 10]), _lib.ptr(b), _lib.ptr(ress[k]), None, _lib.ptr(dst),
                                              B, H, W, 32, 32, ops.EPI_LRELU, 0.3, _lib.ptr(xm), _lib.ptr(ym)))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        chain2()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with _lib.no_gc_during_capture(), torch.cuda.graph(g):
        chain2()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 / 50 * 1e3
    times_cold.setdefault(dx, []).append(us)
    print("conv_dx=%d: %.2f us per launch (graph replay, 50 launches, TEN layers' weights in turn, 50 different residual tensors)" % (dx, us), flush=True)
out["us_per_launch_cold_operands"] = times_cold
del ress

if "--no-step" not in sys.argv:
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    sys.argv = ["bench.py"]
    spec.loader.exec_module(bench)
    dev = torch.device("cuda", 0)
    wls = {}
    for dx in (0, 1):
        _lib.set_option("conv_dx", DXV if dx else 0)              # read when the graph is captured (first step)
        wls[dx] = bench.Workload(sol_amd, dev, 6, 128, 64, 32, 0)
        wls[dx].step(1e-6)
    res_ms = {0: [], 1: []}
    for rep in range(8):
        for dx in ((0, 1) if rep % 2 == 0 else (1, 0)):
            _lib.set_option("conv_dx", DXV if dx else 0)
            sec, loss, _ = bench.timed_steps(wls[dx], 1e-6, 20, 2, torch.cuda.synchronize)
            res_ms[dx].append(sec / 20 * 1e3)
            print("rep %d conv_dx=%d: %.3f ms per SOL-32 step (loss %.4f)" % (rep, dx, sec / 20 * 1e3, loss), flush=True)
    out["ms_per_step"] = res_ms
    import statistics
    print("medians: sb %.3f ms  dx %.3f ms" % (statistics.median(res_ms[0]), statistics.median(res_ms[1])), flush=True)
    for name, (Bq, Yq, Xq) in {"recipe_64x32_b3": (3, 64, 32)}.items():
        r2 = {}
        for dx in (0, 1):
            _lib.set_option("conv_dx", DXV if dx else 0)
            wl = bench.Workload(sol_amd, dev, Bq, Yq, Xq, 32, 0)
            sec, loss, _ = bench.timed_steps(wl, 1e-6, 10, 3, torch.cuda.synchronize)
            r2[dx] = sec / 10 * 1e3
            print("%s conv_dx=%d: %.3f ms per step" % (name, dx, r2[dx]), flush=True)
        out[name] = r2
_lib.set_option("conv_dx", 11)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "conv_dx_ab.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "errors"}))
