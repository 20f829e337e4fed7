#!/usr/bin/env python
"""Per-wave phase timeline of the six-row Conv3D kernel (csrc/conv3d_sb.hip, k_conv3d_sb6) at 128 x 64 x 64.  Builds a second
library with -DSOL_C6_PROF next to the product one:
    python tools/c6_phase_probe.py --build      (needs hipcc; no GPU)
    python tools/c6_phase_probe.py              (on the GPU box)
Stamps (s_memtime, per wave, 8 per tap row): head, taps 0-1 done, row staged, taps 2-3 done, weight DMA issued, tap 4 done,
DMA wait done, barrier passed."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "solver-in-the-loop_amd")
PROF_LIB = os.path.join(PKG, "lib", "libsol_c6prof.so")

if "--build" in sys.argv:
    import importlib.util
    spec = importlib.util.spec_from_file_location("_b", os.path.join(PKG, "_build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    b.build()
    objdir = os.path.join(PKG, "build")
    hipcc = b._hipcc()
    obj = os.path.join(objdir, "conv3d_sb_prof.o")
    subprocess.check_call([hipcc] + b.FLAGS + b.EXTRA.get("conv3d_sb.hip", []) + ["-DSOL_C6_PROF", "-c", os.path.join(PKG, "csrc", "conv3d_sb.hip"), "-o", obj])
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in b.SOURCES if s != "conv3d_sb.hip"] + [obj]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", PROF_LIB])
    print(PROF_LIB)
    sys.exit(0)

import ctypes as C
import numpy as np
import torch
import sol_amd
from sol_amd import _lib, _build, karman3d as k3
_build.LIB = PROF_LIB
_build._stale = lambda: False
lib = _lib.load()
lib.sol_c6_prof_set.argtypes = [C.c_void_p]
dev = "cuda"
B, D, H, W = 1, 128, 64, 64
gen = torch.Generator().manual_seed(0)
x = torch.randn(B, D, H, W, 32, generator=gen, dtype=torch.float32).to(dev)
w = (torch.randn(5, 5, 5, 32, 32, generator=gen, dtype=torch.float32) / 63.0).to(dev)
packed = torch.empty(lib.sol_conv3d_packed_floats(32, 32), dtype=torch.float32, device=dev)
_lib.check(lib.sol_conv3d_pack(_lib.stream(), _lib.ptr(w), 32, 32, 0, _lib.ptr(packed)))
bias = torch.zeros(32, dtype=torch.float32, device=dev)
amax = sol_amd.ops.absmax_slots(x)
NS, nwg = 216, (B * D * H + 5) // 6
nwg = (nwg + 7) // 8 * 8
st = torch.zeros(nwg * 12 * NS, dtype=torch.int32, device=dev)
for _ in range(2):
    k3.conv3d(x, packed, bias, None, 32, True, 0.3, amax, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
assert lib.sol_c6_prof_set(st.data_ptr()) == 0
e0.record()
k3.conv3d(x, packed, bias, None, 32, True, 0.3, amax, None)
e1.record()
torch.cuda.synchronize()
print("launch %.1f us (with stamps)" % (e0.elapsed_time(e1) * 1e3))
s = st.cpu().numpy().astype(np.uint32).reshape(nwg, 12, NS)
real = (s[:, :, 205] - s[:, :, 204]).astype(np.int64)            # 100 MHz ticks
core = (s[:, :, 203] - s[:, :, 200]).astype(np.int64)
valid = real[:, 0] > 0
mhz = core[valid].sum() / real[valid].sum() * 100.0
print("workgroups with stamps: %d of %d; s_memtime rate %.0f MHz; workgroup duration mean %.1f us, min %.1f, max %.1f"
      % (valid.sum(), nwg, mhz, real[valid, 0].mean() * 0.01, real[valid, 0].min() * 0.01, real[valid, 0].max() * 0.01))
t = s[valid][:, :, :200].reshape(-1, 12, 25, 8).astype(np.int64)
d = np.diff(t, axis=3) & 0xffffffff                               # [wg, wave, tap row, 7 phases]
names = ["taps01", "row->LDS", "taps23", "dma", "tap4", "wait", "barrier"]
us = lambda c: c / mhz
print("mean cycles per phase over all workgroups / waves / tap rows:")
print("  " + " ".join("%9s" % n for n in names) + " |   tap row")
print("  " + " ".join("%9.0f" % v for v in d.mean(axis=(0, 1, 2))) + " | %9.0f" % d.sum(axis=3).mean())
for lbl, ws in (("waves 0-7", slice(0, 8)), ("wave 8", slice(8, 9)), ("waves 9-11", slice(9, 12))):
    print("  %-10s " % lbl + " ".join("%9.0f" % v for v in d[:, ws].mean(axis=(0, 1, 2))))
print("by tap row dy (mean over kd):")
dd = d.reshape(d.shape[0], 12, 5, 5, 7)
for dy in range(5):
    print("  dy=%d " % dy + " ".join("%9.0f" % v for v in dd[:, :, :, dy].mean(axis=(0, 1, 2))))
head = (t[:, :, 1:, 0] - t[:, :, :-1, 7]) & 0xffffffff           # barrier passed -> next head stamp (slice boundary work at dy 4 -> 0)
print("between tap rows (next head - barrier passed): mean %.0f cycles; at slice boundaries %.0f" %
      (head.mean(), head.reshape(head.shape[0], 12, 24)[:, :, 4::5].mean()))
pro = (s[valid][:, :, 201] - s[valid][:, :, 200]).astype(np.int64) & 0xffffffff
epi = (s[valid][:, :, 203] - s[valid][:, :, 202]).astype(np.int64) & 0xffffffff
print("prologue %.0f cycles, epilogue %.0f cycles, tap rows total %.0f" % (pro.mean(), epi.mean(), d.sum(axis=(2, 3)).mean()))
# skew at the barrier: per tap row, arrival spread over the twelve waves
arr = t[:, :, :, 6]
print("arrival spread at the barrier (max - min over waves): mean %.0f cycles" % ((arr.max(axis=1) - arr.min(axis=1)) & 0xffffffff).mean())
wg = np.flatnonzero(valid)[len(np.flatnonzero(valid)) // 2]
print("one workgroup (%d), kd = 2, cycles per phase by wave:" % wg)
for wv in range(12):
    print("  wave %2d " % wv + " | ".join(" ".join("%5d" % v for v in d[np.flatnonzero(valid).tolist().index(wg), wv, 10 + dy]) for dy in range(5)))
