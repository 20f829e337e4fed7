import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import sol_amd
from sol_amd import ops
import test_gpu_parity as T
z = np.load("/root/repo/tests/golden/karman_step_64x32.npz")
B, Y, X = z["d"].shape
for solver in ("auto", "cg"):
    g, mk = T.masks_for(Y, X) if solver == "auto" else T.masks_for(Y, X, True, "cg")
    cfg = ops.karman_cfg(B, Y, X, g.dx, masks=mk)
    for rep in range(2):
        d2, py, px = ops.karman_step(T.f32(z["d"]), T.f32(z["vy"]), T.f32(z["vx"]), T.f32(z["re"]), cfg, mk)
        e = (py.cpu().double() - torch.as_tensor(z["vy_out"]).double())
        print(solver, rep, "rel", T.rel(py, z["vy_out"]), T.rel(px, z["vx_out"]), "max abs err", float(e.abs().max()), "rows with err > 1e-4:", (e.abs().amax(dim=(0, 2)) > 1e-4).nonzero().flatten().tolist()[:20],
              "cols:", (e.abs().amax(dim=(0, 1)) > 1e-4).nonzero().flatten().tolist()[:20], "sims", (e.abs().amax(dim=(1, 2)) > 1e-4).nonzero().flatten().tolist())

        def div(vy, vx):
            return (vy[:, 1:, :] - vy[:, :-1, :]) + (vx[:, :, 1:] - vx[:, :, :-1])
        act = torch.as_tensor(g.active).bool()
        dg = div(torch.as_tensor(z["vy_out"]), torch.as_tensor(z["vx_out"]))
        dh = div(py.cpu(), px.cpu())
        print("   max |div| on active cells: golden %.3e  hip %.3e ;  |hip - golden| vx by column:" % (float(dg[:, act].abs().max()), float(dh[:, act].abs().max())),
              [round(float(v), 5) for v in (px.cpu() - torch.as_tensor(z["vx_out"])).abs().amax(dim=(0, 1))[:33]])
