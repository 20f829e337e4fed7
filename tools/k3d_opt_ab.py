#!/usr/bin/env python
"""karman-3d A/B of a library OPTION (sol_set_option) on one box, fresh processes: the karman3d leg of bench.py (SOL-16 training step at
128x64x64 through the replayed graph, CNN pass back to back, solver step).
    python tools/k3d_opt_ab.py OPTION [VALUE_A VALUE_B] [--reps 2]          (default values 1 0)"""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" in sys.argv:
    sys.path.insert(0, ROOT)
    import torch, sol_amd, bench
    i = sys.argv.index("--child")
    sol_amd._lib.set_option(sys.argv[i + 1], int(sys.argv[i + 2]))
    r = bench.karman3d_leg(sol_amd, torch.device("cuda", 0))
    print(json.dumps({"sol16_ms": r["train_sol16"]["ms_per_step"], "cnn_ms": r["cnn_ms_back_to_back"], "solver_us": r["solver_us"]}))
    sys.exit(0)
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
reps = 2
if "--reps" in sys.argv:
    reps = int(sys.argv[sys.argv.index("--reps") + 1]); pos.remove(str(reps))
opt, vals = pos[0], (pos[1:3] if len(pos) >= 3 else ["1", "0"])
res = {v: [] for v in vals}
for r in range(reps):
    for v in vals:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", opt, v], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(out.stdout[-1500:], out.stderr[-1500:]); raise
        res[v].append(d)
        print("rep %d %s=%s %s" % (r, opt, v, d), flush=True)
print(json.dumps({"%s=%s" % (opt, v): {k: statistics.median(x[k] for x in l) for k in l[0]} for v, l in res.items()}))
