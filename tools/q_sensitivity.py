#!/usr/bin/env python
"""How much does each RECALLED PhiFlow-1.5.1 choice (the oracle's named switches) move the numbers the
parity tests pin?  CPU only, float64 oracle.  For every non-default setting: relative L2 change of ONE solver step of the bench
workload (BASELINE configs[2]: 128x64, B = 6) per output field, and the relative change of the SOL-<msteps> loss (forward unroll
with the bench weights).  Q7 only exists for Burgers (periodic faces): reported on the 32x32 Burgers fixture step instead.

    python tools/q_sensitivity.py [--msteps 32] [--json profiles/r04_q_sensitivity.json]

Reading: a switch whose change is far above the 1e-5 parity tolerance is one a PhiFlow fixture (tests/golden/make_phiflow_fixtures.py)
decides unambiguously; one far below it cannot be told apart by any test and does not matter for parity."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sol_oracle as o  # noqa: E402

torch.set_default_dtype(torch.float64)

SETTINGS = [
    ("Q2 inflow_order=before", dict(inflow_order="before"), {}),
    ("Q3 inflow_antialias=True", {}, dict(inflow_antialias=True)),
    ("Q4 den_mode=zero_box", dict(den_mode="zero_box"), {}),
    ("Q5 grad_pad=dirichlet0", dict(grad_pad="dirichlet0"), {}),
    ("Q6 solver=cg (accuracy 1e-5, batch-global stop)", dict(solver="cg"), {}),
]


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--msteps", type=int, default=32)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    B, Y, X, ms = 6, 128, 64, args.msteps
    w = o.bench_workload(B, Y, X, ms)
    rows = []
    with torch.no_grad():
        t0 = time.time()
        base_step = o.karman_step(w["d0"], w["vy0"], w["vx0"], w["re"], w["geom"])
        base_loss = float(o.unrolled_loss(w["params"], w["d0"], w["vy0"], w["vx0"], w["re"], w["gt_vy"], w["gt_vx"], w["geom"], w["std_v"], w["std_re"]))
        print("default: SOL-%d loss %.6f (%.0f s)" % (ms, base_loss, time.time() - t0), flush=True)
        for name, step_kw, geom_kw in SETTINGS:
            t0 = time.time()
            g = o.geometry(Y, X, **geom_kw) if geom_kw else w["geom"]
            st = o.karman_step(w["d0"], w["vy0"], w["vx0"], w["re"], g, **step_kw)
            loss = float(o.unrolled_loss(w["params"], w["d0"], w["vy0"], w["vx0"], w["re"], w["gt_vy"], w["gt_vx"], g, w["std_v"], w["std_re"], **step_kw))
            row = {"setting": name, "step_density": rel(st[0], base_step[0]), "step_vy": rel(st[1], base_step[1]), "step_vx": rel(st[2], base_step[2]),
                   "loss": loss, "loss_rel_change": abs(loss - base_loss) / abs(base_loss)}
            rows.append(row)
            print("%-50s step d %.2e vy %.2e vx %.2e | SOL-%d loss %.6f (rel %.2e)  [%.0f s]" % (
                name, row["step_density"], row["step_vy"], row["step_vx"], ms, loss, row["loss_rel_change"], time.time() - t0), flush=True)
        # Q7: Burgers periodic faces (configs[0]) on the committed fixture's inputs
        z = np.load(os.path.join(ROOT, "tests", "golden", "burgers_step_32x32.npz"))
        t = lambda k: torch.as_tensor(z[k], dtype=torch.float64)
        a = o.burgers_step(t("vy"), t("vx"), float(z["dt"]), float(z["nu"]), t("fy"), t("fx"))
        b = o.burgers_step(t("vy"), t("vx"), float(z["dt"]), float(z["nu"]), t("fy"), t("fx"), periodic_faces="domain")
        row = {"setting": "Q7 burgers periodic_faces=domain (32x32 Burgers step, configs[0])", "step_vy": rel(b[0], a[0]), "step_vx": rel(b[1], a[1])}
        rows.append(row)
        print("%-50s step vy %.2e vx %.2e" % (row["setting"], row["step_vy"], row["step_vx"]), flush=True)
    out = {"workload": "oracle.bench_workload(B=%d, %dx%d, msteps=%d), float64" % (B, Y, X, ms), "default_loss": base_loss, "rows": rows}
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)
    return out


if __name__ == "__main__":
    main()
