#!/usr/bin/env python
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sol_amd, sol_oracle as o
from sol_amd import ops
dev = "cuda"
f32 = lambda t: torch.as_tensor(np.asarray(t), dtype=torch.float32).to(dev).contiguous()
B, Y, X, ms = 6, 128, 64, 32
w = o.bench_workload(B, Y, X, ms)
oin = (f32(w["d0"]), f32(w["vy0"]), f32(w["vx0"]), f32(w["re"]), f32(torch.stack(w["gt_vy"])), f32(torch.stack(w["gt_vx"])))

def run(tag, sync_after_apply, sync_after_fwd, want_final=True, nsteps=4):
    g = w["geom"]
    mk = ops.SceneMasks(g.active, g.inflow, g.bc_mask, g.bc_mask)
    net = sol_amd.model_mars_moon(cin=3, cout=2, seed=0)
    net.set_weights([p.detach().numpy() for p in w["params"]])
    tr = sol_amd.SolTrainer(net, mk, B, Y, X, ms, g.dx, w["std_v"], o.STD_RE)
    losses, gn, wn = [], [], []
    for t in range(nsteps):
        loss = tr.fwd_bwd(*oin, want_final=want_final)
        if sync_after_fwd:
            torch.cuda.synchronize()
        losses.append(loss)
        gn.append(tr.grads.double().norm())
        tr.apply_gradients(1e-4)
        wn.append(net.params.detach().double().norm())
        if sync_after_apply:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("%-40s loss %s\n%40s |g|  %s\n%40s |w|  %s" % (tag, " ".join("%.6g" % float(l) for l in losses), "", " ".join("%.5g" % float(v) for v in gn),
                                                     "", " ".join("%.8g" % float(v) for v in wn)), flush=True)

run("sync after apply + fwd", True, True)
run("sync after fwd only (the test)", False, True)
run("sync after fwd only (again)", False, True)
run("no sync at all", False, False)
run("sync after apply only", True, False)
run("sync after fwd only, no final", False, True, want_final=False)
