#!/usr/bin/env python
"""karman-3d SOL-16 training step with the fused reverse-sweep glue (sol_karman3d_correct_bwd / _feature_bwd) against the torch elementwise
composition, same box, fresh processes.   python tools/k3d_glue_ab.py [--reps 2]"""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" in sys.argv:
    sys.path.insert(0, ROOT)
    import torch, sol_amd, bench
    from sol_amd import karman3d as k3
    glue = sys.argv[sys.argv.index("--child") + 1]
    init = k3.Karman3DTrainer.__init__
    def patched(self, *a, **kw):
        kw.setdefault("glue", glue)
        init(self, *a, **kw)
    k3.Karman3DTrainer.__init__ = patched
    r = bench.karman3d_leg(sol_amd, torch.device("cuda", 0))
    print(json.dumps({"sol16_ms": r["train_sol16"]["ms_per_step"]}))
    sys.exit(0)
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
res = {"fused": [], "torch": []}
for r in range(reps):
    for n in res:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(out.stdout[-1500:], out.stderr[-1500:]); raise
        res[n].append(d)
        print("rep %d %-6s %s" % (r, n, d), flush=True)
print(json.dumps({n: {k: statistics.median(x[k] for x in v) for k in v[0]} for n, v in res.items()}))
