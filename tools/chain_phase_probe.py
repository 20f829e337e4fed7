#!/usr/bin/env python
"""Phase timeline of the persistent CNN chain launch (csrc/cnn_chain.hip) at the C3 shape.  Builds a second library with
-DSOL_CHAIN_PROF next to the product one:
    python tools/chain_phase_probe.py --build      (needs hipcc; no GPU)
    python tools/chain_phase_probe.py              (on the GPU box)
Stamps: thread 0 of every workgroup, 8 per layer (layer start, after tap steps 0..4, output stores issued, layer end)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "solver-in-the-loop_amd")
PROF_LIB = os.path.join(PKG, "lib", "libsol_chainprof.so")

if "--build" in sys.argv:
    import importlib.util
    spec = importlib.util.spec_from_file_location("_b", os.path.join(PKG, "_build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    b.build()
    objdir = os.path.join(PKG, "build")
    hipcc = b._hipcc()
    obj = os.path.join(objdir, "cnn_chain_prof.o")
    subprocess.check_call([hipcc] + b.FLAGS + ["-DSOL_CHAIN_PROF", "-c", os.path.join(PKG, "csrc", "cnn_chain.hip"), "-o", obj])
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in b.SOURCES if s != "cnn_chain.hip"] + [obj]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", PROF_LIB])
    print(PROF_LIB)
    sys.exit(0)

import ctypes as C
import numpy as np
import torch
import sol_amd
from sol_amd import _lib, _build
_build.LIB = PROF_LIB
_build._stale = lambda: False
import bench
lib = _lib.load()
lib.sol_chain_prof_set.argtypes = [C.c_void_p]
dev = torch.device("cuda", 0)
_lib.set_option("cnn_persistent", 1)
wl = bench.Workload(sol_amd, dev, 6, 128, 64, 1, 0)
tr = wl.trainer
args = (wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx)
for _ in range(3):
    tr.fwd_bwd(*args, eager=True)
nwg, NL = 256, 12
st = torch.zeros(nwg * NL * 8, dtype=torch.int64, device=dev)
assert lib.sol_chain_prof_set(_lib.ptr(st)) == 0
torch.cuda.synchronize()
names = ["step0", "step1", "step2", "step3", "step4", "stores", "end"]
for rep in range(2):
    st.zero_()
    tr.fwd_bwd(*args, eager=True)       # the LAST chain launch (backward-data pass) leaves its stamps
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(nwg, NL, 8).astype(np.int64)[:, :10]
    t0 = s[:, 0, 0].min()
    print("rep %d: chain total %.2f us (first start -> last end)" % (rep, (s[:, 9, 7].max() - t0) * 0.01))
    print("  layer | start spread | mean us per phase: " + " ".join("%7s" % n for n in names) + " |  layer total (mean / max)")
    for l in range(10):
        d = np.diff(s[:, l, :], axis=1) * 0.01
        tot = (s[:, l, 7] - s[:, l, 0]) * 0.01
        print("  %5d | %12.2f | %s | %6.2f / %6.2f" % (l, (s[:, l, 0].max() - s[:, l, 0].min()) * 0.01,
                                                      " ".join("%7.2f" % v for v in d.mean(axis=0)), tot.mean(), tot.max()))
    if rep == 1:
        l = 5
        d = np.diff(s[:, l, :], axis=1) * 0.01
        print("  layer 5 per-phase max over workgroups: " + " ".join("%7.2f" % v for v in d.max(axis=0)))
        print("  layer 5 per-phase min over workgroups: " + " ".join("%7.2f" % v for v in d.min(axis=0)))
