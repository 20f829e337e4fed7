#!/usr/bin/env python
"""Debug: SOL-32 bench workload, oracle-generated vs HIP-generated inputs, graph vs eager, 3 Adam steps at lr 1e-4."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench, sol_amd, sol_oracle as o
from sol_amd import ops
dev = torch.device("cuda", 0)
f32 = lambda t: torch.as_tensor(np.asarray(t), dtype=torch.float32).to(dev).contiguous()
B, Y, X, ms = 6, 128, 64, 32
w = o.bench_workload(B, Y, X, ms)
wl = bench.Workload(sol_amd, dev, B, Y, X, ms, 0)
oin = (f32(w["d0"]), f32(w["vy0"]), f32(w["vx0"]), f32(w["re"]), f32(torch.stack(w["gt_vy"])), f32(torch.stack(w["gt_vx"])))
gin = (wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx)
for n, a, b in zip("d0 vy0 vx0 re gt_vy gt_vx".split(), oin, gin):
    print("input %-6s max|oracle-hip| %.3e  max|.| %.3e" % (n, float((a - b).abs().max()), float(a.abs().max())))
pw = torch.cat([p.reshape(-1) for p in w["params"]]).float().to(dev)
print("weights max diff", float((pw - wl.net.params.detach()).abs().max()))

def run(tag, inputs, use_graph, eager, via_step):
    g = w["geom"]
    mk = ops.SceneMasks(g.active, g.inflow, g.bc_mask, g.bc_mask)
    net = sol_amd.model_mars_moon(cin=3, cout=2, seed=0)
    net.set_weights([p.detach().numpy() for p in w["params"]])
    tr = sol_amd.SolTrainer(net, mk, B, Y, X, ms, g.dx, w["std_v"], o.STD_RE, use_graph=use_graph)
    out = []
    for t in range(3):
        if via_step:
            loss = tr.train_step(*inputs, lr=1e-4, want_final=True, eager=eager)
        else:
            loss = tr.fwd_bwd(*inputs, want_final=True, eager=eager)
            gn = float(tr.grads.double().norm()); gmax = float(tr.grads.abs().max())
            tr.apply_gradients(1e-4)
        out.append(float(loss))
        torch.cuda.synchronize()
        print("  %s step %d loss %.6g  |w|max %.4g  grad finite %s" % (tag, t, out[-1], float(net.params.detach().abs().max()), bool(torch.isfinite(tr.grads).all())), flush=True)
    return out, net.params.detach().clone()

r1, p1 = run("oracle-in graph fwd_bwd+apply", oin, True, False, False)
r2, p2 = run("hip-in    graph fwd_bwd+apply", gin, True, False, False)
r3, p3 = run("oracle-in graph train_step   ", oin, True, False, True)
r4, p4 = run("oracle-in eager fwd_bwd+apply", oin, False, False, False)
r5, p5 = run("hip-in    graph train_step   ", gin, True, False, True)
