#!/usr/bin/env python
"""Phase timeline of the 32 -> 32 split-precision convolution launch (C3 shape).  Builds a second library with
-DSOL_CONV_PROF (stamps compiled into k_conv5x5_sb) next to the product one:
    python tools/conv_phase_probe.py --build      (needs hipcc; no GPU)
    python tools/conv_phase_probe.py              (on the GPU box)
The stamps perturb the kernel (each one is a scalar load, s_memrealtime and a store by wave 0: ~0.4 us per phase, and the
"start" stamp waits for a scalar load), so read DIFFERENCES between phases; tools/conv_variants.py gives unperturbed totals."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "solver-in-the-loop_amd")
PROF_LIB = os.path.join(PKG, "lib", "libsol_prof.so")

if "--build" in sys.argv:
    import importlib.util
    spec = importlib.util.spec_from_file_location("_b", os.path.join(PKG, "_build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    b.build()
    objdir = os.path.join(PKG, "build")
    hipcc = b._hipcc()
    obj = os.path.join(objdir, "conv5x5_sb_prof.o")
    subprocess.check_call([hipcc] + b.FLAGS + ["-DSOL_CONV_PROF", "-c", os.path.join(PKG, "csrc", "conv5x5_sb.hip"), "-o", obj])
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in b.SOURCES if s != "conv5x5_sb.hip"] + [obj]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", PROF_LIB])
    print(PROF_LIB)
    sys.exit(0)

import ctypes as C
import numpy as np
import torch
import sol_amd
from sol_amd import ops, _lib, _build
_build.LIB = PROF_LIB                       # load the instrumented library instead of the product one
from sol_amd._lib import ptr, stream, check

lib = _lib.load()
lib.sol_conv_prof_set.argtypes = [C.c_void_p, C.c_uint]
B, Y, X = 6, 128, 64
dev = "cuda"
x = torch.randn(B, Y, X, 32, device=dev)
w = torch.randn(5, 5, 32, 32, device=dev) * 0.05
packed = ops._pack(w, 32, 32, ops.CONV_FWD)
bias = torch.randn(32, device=dev)
y = torch.empty_like(x)
xam = ops.absmax_slots(x)
yam = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device=dev)
nwg = B * Y // 3
st = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
call = lambda: check(lib.sol_conv5x5_scaled(stream(), ptr(x), ptr(packed), ptr(bias), None, None, ptr(y), B, Y, X, 32, 32,
                                             ops.EPI_LRELU, 0.3, ptr(xam), ptr(yam)))
for _ in range(5):
    call()
assert lib.sol_conv_prof_set(C.c_void_p(st.data_ptr()), 1) == 0
torch.cuda.synchronize()
for rep in range(3):
    call(); torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(nwg, 16).astype(np.int64)
    t0 = s[:, 0].min()
    names = [(0, "start"), (10, "scale known"), (1, "prologue"), (2, "dy0"), (3, "dy1"), (4, "dy2"), (5, "dy3"), (6, "dy4"), (7, "stores issued"), (8, "absmax"), (9, "drained")]
    if rep < 2:
        continue
    print("us after the first workgroup start: min / median / max over %d workgroups" % nwg)
    for k, n in names:
        v = (s[:, k] - t0) * 0.01
        print("  %-14s %6.2f %6.2f %6.2f" % (n, v.min(), np.median(v), v.max()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    call()
e1.record(); torch.cuda.synchronize()
print("launch: %.2f us (with stamps)" % (e0.elapsed_time(e1) * 10))
assert lib.sol_conv_prof_set(None, 1) == 0
torch.cuda.synchronize()
# GPU-bound timing: 50 launches captured in one graph (a ctypes call costs ~8 us of host time)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    call(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        for _ in range(50):
            call()
    gr.replay(); torch.cuda.synchronize()
    e0.record(side)
    for _ in range(4):
        gr.replay()
    e1.record(side); torch.cuda.synchronize()
print("launch: %.2f us (stamps off, graph replay)" % (e0.elapsed_time(e1) * 5))
