#!/usr/bin/env python
"""Per-WAVE timeline of one image row of the 32 -> 32 weight-gradient body (bww_sb_body<2>, split_kernels.hpp): where do the 2.2 us of a
row go when its 150 MFMAs per SIMD take 1.4?
    python tools/ab_lib.py --build bwwprof conv5x5_sb.hip:-DBWW_PROF          (no GPU)
    SOL_HIP_LIB=solver-in-the-loop_amd/lib/libsol_bwwprof.so python tools/bww_row_probe.py      (GPU box)
Stamps (s_memtime, shader clocks, low 32 bits; each stamp waits for lgkmcnt(0) = the wave's outstanding LDS operations):
 0 row start | 1 after the dz role's early staging | 2 operand fragments in registers | 3 last MFMA issued | 4 late staging written | 5 behind the barrier."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import sol_amd
from sol_amd import _lib, karman3d as k3

lib = _lib.load()
dev = "cuda"
gen = torch.Generator().manual_seed(0)
x = torch.randn(1, 128, 64, 64, 32, generator=gen, dtype=torch.float32).to(dev)
dz = (torch.randn(1, 128, 64, 64, 32, generator=gen, dtype=torch.float32) * 1e-3).to(dev)
xm, zm = k3._absmax(x), k3._absmax(dz)
for _ in range(2):
    k3.conv3d_bwd_weight(x, dz, 32, 32, xmax=xm, zmax=zm)
torch.cuda.synchronize()
NWG = 512
st = torch.zeros(NWG * 8 * 320, dtype=torch.int32, device=dev)
lib.sol_bww_prof_set.argtypes = [C.c_void_p]
assert lib.sol_bww_prof_set(st.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
k3.conv3d_bwd_weight(x, dz, 32, 32, xmax=xm, zmax=zm)
e1.record()
torch.cuda.synchronize()
print("five passes + reduce: %.1f us (with stamps)" % (e0.elapsed_time(e1) * 1e3))
s = st.cpu().numpy().astype(np.uint32).reshape(NWG, 8, 40, 8)
used = [b for b in range(NWG) if s[b, 0, 0, 0] != 0 and s[b, 0, 1, 0] != 0]
print("workgroups with stamps:", len(used))
s = s[used]
rows = [r for r in range(32) if (s[:, :, r, 5] != 0).all() and (s[:, :, r, 0] != 0).all()]
print("rows per workgroup:", len(rows))
d = lambda a, b: (a.astype(np.int64) - b.astype(np.int64)) & 0xffffffff
mid = rows[2:-2]
row_t = float(np.median(d(s[:, :, mid[-1], 0], s[:, :, mid[0], 0]))) / (len(mid) - 1)
print("clocks per row (rows %d..%d): %.0f" % (mid[0], mid[-1], row_t))
rt = d(s[:, 0, mid[-1], 6], s[:, 0, mid[0], 6]).astype(np.float64)          # 100 MHz ticks over the same rows (wave 0)
ck = d(s[:, 0, mid[-1], 0], s[:, 0, mid[0], 0]).astype(np.float64)
mhz = 100.0 * ck / np.maximum(rt, 1.0)
print("shader clock held by the CUs during rows %d..%d: median %.0f MHz (p10 %.0f, p90 %.0f); row = %.0f clocks = %.2f us" % (
    mid[0], mid[-1], np.median(mhz), np.percentile(mhz, 10), np.percentile(mhz, 90), np.median(ck) / (len(mid) - 1), np.median(rt) / (len(mid) - 1) / 100.0))
names = ["request + early stage", "operand reads (LDS)", "MFMA block (issue)", "late stage (+ MFMA drain)", "barrier wait"]
out = {"clocks_per_row": float(row_t)}
for role, ws in (("x role (waves 0-3: stage late)", range(0, 4)), ("dz role (waves 4-7: stage early)", range(4, 8))):
    print(role)
    for k in range(5):
        v = d(s[:, list(ws)][:, :, mid, k + 1], s[:, list(ws)][:, :, mid, k])
        print("   %-28s median %6.0f   p10 %6.0f   p90 %6.0f clocks" % (names[k], np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
        out["%s/%s" % (role[:2].strip(), names[k])] = float(np.median(v))
# one workgroup, one row, all waves: offsets from the earliest row start
b, r = len(used) // 2, mid[len(mid) // 2]
t0 = s[b, :, r, 0].astype(np.int64).min()
print("workgroup %d row %d, stamps relative to the first wave's row start:" % (used[b], r))
for w in range(8):
    print("   wave %d: %s" % (w, " ".join("%6d" % ((int(s[b, w, r, k]) - t0) & 0xffffffff) for k in range(6))))
print(json.dumps(out))
