#!/usr/bin/env python
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sol_amd, sol_oracle as o
from sol_amd import ops
import test_gpu_parity as T
gd = os.path.join(ROOT, "tests", "golden")
try:
    T.test_sol32_bench_workload_against_golden(gd)
    print("test function outside pytest: PASS", flush=True)
except AssertionError as e:
    print("test function outside pytest: FAIL", str(e)[:300], flush=True)

z = np.load(os.path.join(gd, "train_128x64_sol32.npz"))
B, Y, X, ms = 6, 128, 64, 32
w = o.bench_workload(B, Y, X, ms)
f32, rel = T.f32, T.rel

def variant(tag, blocks):
    net, tr = T._trainer_from(w["params"], w["geom"], B, Y, X, ms, w["std_v"])
    args = (f32(w["d0"]), f32(w["vy0"]), f32(w["vx0"]), f32(w["re"]), f32(torch.stack(w["gt_vy"])), f32(torch.stack(w["gt_vx"])))
    traj = []
    for t in range(3):
        loss = tr.fwd_bwd(*args, want_final=True)
        traj.append(float(loss))
        if t == 0:
            if "ls" in blocks: np.allclose(tr.loss_steps.cpu().numpy(), z["loss_steps"], rtol=2e-5)
            if "sub" in blocks: rel(tr.grads[::16], z["grads_sub16"])
            if "l2" in blocks: float(tr.grads.double().norm())
            if "norms" in blocks: np.array([float(tr.grads[net.offsets[k]:net.offsets[k + 1]].double().norm()) for k in range(24)])
            if "final" in blocks: rel(tr.final[1], z["vy_final"]); rel(tr.final[2], z["vx_final"]); rel(tr.final[0], z["d_final"])
        tr.apply_gradients(float(z["lr"]))
    print("%-28s %s" % (tag, traj), flush=True)

variant("none", ())
variant("all", ("ls", "sub", "l2", "norms", "final"))
for b in ("ls", "sub", "l2", "norms", "final"):
    variant(b, (b,))
variant("none again", ())
