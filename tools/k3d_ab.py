#!/usr/bin/env python
"""karman-3d A/B of a library variant (tools/ab_lib.py --build NAME ...) against the product library on ONE box: the karman3d leg of
bench.py (SOL-16 training step at 128x64x64 through the replayed graph, CNN pass back to back, solver step) alternating between the
libraries in fresh processes.   python tools/k3d_ab.py NAME [NAME2 ...] [--reps 2]"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" in sys.argv:
    sys.path.insert(0, ROOT)
    import torch
    import sol_amd
    import bench
    r = bench.karman3d_leg(sol_amd, torch.device("cuda", 0))
    print(json.dumps({"sol16_ms": r["train_sol16"]["ms_per_step"], "cnn_ms": r["cnn_ms_back_to_back"], "solver_us": r["solver_us"]}))
    sys.exit(0)
names = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
res = {n: [] for n in ["product"] + names}
for r in range(reps):
    for n in res:
        env = dict(os.environ)
        if n != "product":
            env["SOL_HIP_LIB"] = os.path.join(ROOT, "solver-in-the-loop_amd", "lib", "libsol_%s.so" % n)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(out.stdout[-1500:], out.stderr[-1500:])
            raise
        res[n].append(d)
        print("rep %d %-10s %s" % (r, n, d), flush=True)
print(json.dumps({n: {k: statistics.median(x[k] for x in v) for k in v[0]} for n, v in res.items()}))
