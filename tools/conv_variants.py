#!/usr/bin/env python
"""A/B timing of library variants on the 32 -> 32 convolution launch (C3 shape), GPU-bound (50 launches per graph replay).
    python tools/conv_variants.py --build NAME "-DSOL_CONV_TRUNC=1"   -> lib/libvar_NAME.so (conv5x5_sb.hip recompiled with the flags; no GPU)
    python tools/conv_variants.py libA.so libB.so ...                 (on the GPU box; each library is timed in its own process)
Round-1 result (us per launch, lrelu + absmax): return at entry 1.74 | + prologue 3.3 | + tap-row loop 10.0 | full 12.5."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "solver-in-the-loop_amd")

if len(sys.argv) >= 3 and sys.argv[1] == "--build":
    import importlib.util
    spec = importlib.util.spec_from_file_location("_b", os.path.join(PKG, "_build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    b.build()
    name, flags = sys.argv[2], (sys.argv[3].split() if len(sys.argv) > 3 else [])
    objdir = os.path.join(PKG, "build")
    obj = os.path.join(objdir, "conv5x5_sb_%s.o" % name)
    subprocess.check_call([b._hipcc()] + b.FLAGS + flags + ["-c", os.path.join(PKG, "csrc", "conv5x5_sb.hip"), "-o", obj])
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in b.SOURCES if s != "conv5x5_sb.hip"] + [obj]
    out = os.path.join(PKG, "lib", "libvar_%s.so" % name)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    print(out)
    sys.exit(0)

if len(sys.argv) > 2 or (len(sys.argv) == 2 and not os.environ.get("_CONV_VARIANT_CHILD")):
    for lib in sys.argv[1:]:
        env = dict(os.environ, _CONV_VARIANT_CHILD="1")
        subprocess.call([sys.executable, os.path.abspath(__file__), lib], env=env)
    sys.exit(0)

import torch
import sol_amd
from sol_amd import ops, _lib, _build
lib_path = os.path.abspath(sys.argv[1])
_build.LIB = lib_path
_build._stale = lambda: False
from sol_amd._lib import ptr, stream, check
lib = _lib.load()
B, Y, X = 6, 128, 64
dev = "cuda"
x = torch.randn(B, Y, X, 32, device=dev)
res = torch.randn(B, Y, X, 32, device=dev)
w_raw = torch.randn(5, 5, 32, 32, device=dev) * 0.05
packed = ops._pack(w_raw, 32, 32, ops.CONV_FWD)
bias = torch.randn(32, device=dev)
y = torch.empty_like(x)
nslots = lib.sol_absmax_slots() if hasattr(lib, "sol_absmax_slots") else 64
xam = torch.zeros(256, dtype=torch.int32, device=dev); xam[0] = x.abs().max().view(torch.int32)
yam = torch.zeros(256, dtype=torch.int32, device=dev)


def timed(fn):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(50):
                fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(side)
            for _ in range(4):
                gr.replay()
            e1.record(side); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 5)
    return best


def conv(epi, r, act, ym):
    return lambda: check(lib.sol_conv5x5_scaled(stream(), ptr(x), ptr(packed), ptr(bias), r, act, ptr(y), B, Y, X, 32, 32, epi, 0.3, ptr(xam), ym))


t1 = timed(conv(ops.EPI_LRELU, None, None, ptr(yam)))
t2 = timed(conv(ops.EPI_LRELU, None, None, None))
t3 = timed(conv(ops.EPI_LRELU, ptr(res), None, ptr(yam)))
t4 = timed(conv(ops.EPI_DLRELU, ptr(res), ptr(x), ptr(yam)))
print("%-44s lrelu+ymax %.2f | lrelu %.2f | res+lrelu+ymax %.2f | res+dlrelu+ymax %.2f us" % (os.path.basename(lib_path), t1, t2, t3, t4))

# correctness of the variant (residual + LeakyReLU) against float64 torch
conv(ops.EPI_LRELU, ptr(res), None, ptr(yam))(); torch.cuda.synchronize()
ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w_raw.double().permute(3, 2, 0, 1), bias.double(), padding=2).permute(0, 2, 3, 1) + res.double()
ref = torch.where(ref > 0, ref, 0.3 * ref)
print("    rel-L2 error vs float64: %.2e" % ((y.double() - ref).norm() / ref.norm()).item())
