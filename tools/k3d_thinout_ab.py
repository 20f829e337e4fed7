#!/usr/bin/env python
"""karman-3d: the thin-OUTPUT layers as one depth-packed 2-D launch + gather (sol_conv3d_thin_out, MarsMoon3D.thin_out_kpack = True) against the
eight-row Conv3D kernel with one channel tile, same box, fresh processes (bench.karman3d_leg).   python tools/k3d_thinout_ab.py [--reps 2]"""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" in sys.argv:
    sys.path.insert(0, ROOT)
    import torch, sol_amd, bench
    from sol_amd import karman3d as k3
    k3.MarsMoon3D.thin_out_kpack = sys.argv[sys.argv.index("--child") + 1] == "1"
    r = bench.karman3d_leg(sol_amd, torch.device("cuda", 0))
    print(json.dumps({"sol16_ms": r["train_sol16"]["ms_per_step"], "cnn_ms": r["cnn_ms_back_to_back"], "solver_us": r["solver_us"]}))
    sys.exit(0)
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
res = {"kpack_out": [], "sb8_nt1": []}
for r in range(reps):
    for n, flag in (("kpack_out", "1"), ("sb8_nt1", "0")):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", flag], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(out.stdout[-1500:], out.stderr[-1500:]); raise
        res[n].append(d)
        print("rep %d %-10s %s" % (r, n, d), flush=True)
print(json.dumps({n: {k: statistics.median(x[k] for x in v) for k in v[0]} for n, v in res.items()}))
