#!/usr/bin/env python
"""Does a library variant (tools/ab_lib.py --build NAME ...) compute the SAME BITS as the product library?  One C3 training step (loss,
gradient, final state) and one small karman-3d training step per library in fresh processes (SOL_HIP_LIB), SHA-1 of every result.
    python tools/lib_bitcompare.py NAME [NAME2 ...]        (on the GPU box)"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" in sys.argv:
    sys.path.insert(0, ROOT)
    import torch
    import sol_amd
    import bench
    dev = torch.device("cuda", 0)
    sha = lambda t: hashlib.sha1(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:16]
    out = {}
    for (B, Y, X, ms) in ((6, 128, 64, 32), (3, 64, 32, 4)):
        wl = bench.Workload(sol_amd, dev, B, Y, X, ms, 0, use_graph=False)
        tr = wl.trainer
        loss = tr.fwd_bwd(wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx, want_final=True)
        torch.cuda.synchronize()
        g = tr.grads
        g = torch.cat([t.flatten() for t in g]) if isinstance(g, (list, tuple)) else g
        out["%dx%dx%d_ms%d" % (B, Y, X, ms)] = {"loss": float(loss).hex(), "grad": sha(g), "final": [sha(t) for t in tr.final]}
    print(json.dumps(out))
    sys.exit(0)
names = ["product"] + [a for a in sys.argv[1:] if not a.startswith("--")]
res = {}
for n in names:
    env = dict(os.environ)
    if n != "product":
        env["SOL_HIP_LIB"] = os.path.join(ROOT, "solver-in-the-loop_amd", "lib", "libsol_%s.so" % n)
    o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        res[n] = json.loads(o.stdout.strip().splitlines()[-1])
    except Exception:
        print(o.stdout[-1500:], o.stderr[-1500:])
        raise
    print(n, json.dumps(res[n]), flush=True)
for n in names[1:]:
    print("%-10s %s" % (n, "BIT-IDENTICAL to product" if res[n] == res["product"] else "DIFFERS from product"))
