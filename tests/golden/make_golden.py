#!/usr/bin/env python
"""Generates the golden vectors in tests/golden/ from the float64 CPU oracle.

The reference has no tests / golden vectors and its PhiFlow/TF path cannot be imported here
(SURVEY.md section 8c), so these fixtures pin the ORACLE (regression) and give the GPU tests
committed input/output pairs.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import sol_oracle as o  # noqa: E402
import sol_oracle3d as o3  # noqa: E402

torch.set_default_dtype(torch.float64)


def r32(t):
    """round to fp32-representable values (the GPU path consumes fp32 inputs)"""
    return t.detach().float().double()


def step_case(B, Y, X, seed, **kw):
    g = o.geometry(Y, X)
    d, vy, vx = (r32(t) for t in o.synthetic_state(B, Y, X, seed))
    re = torch.tensor([o.RE_TRAIN[i % 6] for i in range(B)])
    vy = vy.clone().requires_grad_(True)
    vx = vx.clone().requires_grad_(True)
    d2, py, px = o.karman_step(d, vy, vx, re, g, **kw)
    gen = torch.Generator().manual_seed(seed + 1)
    wy = r32(torch.randn(py.shape, generator=gen))
    wx = r32(torch.randn(px.shape, generator=gen))
    ((py * wy).sum() + (px * wx).sum()).backward()
    n = lambda t: t.detach().numpy().astype(np.float32)
    return dict(d=n(d), vy=n(vy), vx=n(vx), re=n(re), d_out=n(d2), vy_out=n(py), vx_out=n(px),
                wy=n(wy), wx=n(wx), g_vy=n(vy.grad), g_vx=n(vx.grad))


def burgers_case():
    B, Y, X = 5, 32, 32
    gen = torch.Generator().manual_seed(3)
    vy = r32(0.3 * o._smooth(torch.randn(B, Y + 1, X, generator=gen))).requires_grad_(True)
    vx = r32(0.3 * o._smooth(torch.randn(B, Y, X + 1, generator=gen))).requires_grad_(True)
    fy = r32(0.15 * o._smooth(torch.randn(B, Y + 1, X, generator=gen)))
    fx = r32(0.15 * o._smooth(torch.randn(B, Y, X + 1, generator=gen)))
    ay, ax = o.burgers_step(vy, vx, 0.1, 0.1, fy, fx)
    wy = r32(torch.randn(ay.shape, generator=gen))
    wx = r32(torch.randn(ax.shape, generator=gen))
    ((ay * wy).sum() + (ax * wx).sum()).backward()
    n = lambda t: t.detach().numpy().astype(np.float32)
    return dict(vy=n(vy), vx=n(vx), fy=n(fy), fx=n(fx), vy_out=n(ay), vx_out=n(ax), wy=n(wy), wx=n(wx),
                g_vy=n(vy.grad), g_vx=n(vx.grad), dt=0.1, nu=0.1)


def train_case(B=2, Y=16, X=8, ms=2):
    g = o.geometry(Y, X)
    d, vy, vx = (r32(t) for t in o.synthetic_state(B, Y, X, 1234))
    re = torch.tensor(o.RE_TRAIN[:B])
    gts = [tuple(r32(t) for t in o.synthetic_state(B, Y, X, 4321 + i, project_it=False)) for i in range(ms)]
    params = [r32(p).requires_grad_(True) for p in o.init_params(0)]
    gen = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for p in params:
            if p.dim() == 1:
                p.copy_(r32(0.01 * torch.randn(p.shape, generator=gen)))
    std_v = (0.2, 0.25)
    loss, losses, states = o.unrolled_loss(params, d, vy, vx, re, [s[1] for s in gts], [s[2] for s in gts], g,
                                           std_v, o.STD_RE, return_states=True)
    loss.backward()
    n = lambda t: t.detach().numpy().astype(np.float32)
    grads = np.concatenate([p.grad.numpy().ravel() for p in params])
    # weights are reproducible from seed 0 (fp32 rounded) + the stored biases; the 260k-element
    # gradient is stored as per-tensor L2 norms plus every 16th element
    return dict(d=n(d), vy=n(vy), vx=n(vx), re=n(re), gt_vy=np.stack([n(s[1]) for s in gts]),
                gt_vx=np.stack([n(s[2]) for s in gts]), std_v=np.array(std_v), std_re=o.STD_RE,
                biases=np.concatenate([n(p).ravel() for p in params if p.dim() == 1]),
                loss=float(loss), loss_steps=np.array([float(l) for l in losses]),
                grad_norms=np.array([float(p.grad.norm()) for p in params]), grads_sub16=grads[::16].astype(np.float32),
                vy_final=n(states[-1][1]), vx_final=n(states[-1][2]), d_final=n(states[-1][0]))


def sol32_case(B=6, Y=128, X=64, ms=32, adam_steps=3, lr=1e-4):
    """BASELINE.json configs[2] at its real depth: the workload bench.py times (oracle.bench_workload), one float64
    forward + autograd backward per Adam step.  ~2 min per step on 8 cores.  Stored: the loss after 0..adam_steps-1
    updates, and for the FIRST step the 32 per-step losses, per-tensor gradient norms, every 16th gradient element and
    the final state (fp32).  Inputs are NOT stored (13 MB of ground-truth frames): the test regenerates them with the same
    oracle call, which is deterministic to float64 rounding."""
    import time
    w = o.bench_workload(B, Y, X, ms)
    params = [p.clone().requires_grad_(True) for p in w["params"]]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    out = {}
    traj = []
    for t in range(1, adam_steps + 1):
        t0 = time.time()
        loss, losses, states = o.unrolled_loss(params, w["d0"], w["vy0"], w["vx0"], w["re"], w["gt_vy"], w["gt_vx"], w["geom"],
                                               w["std_v"], w["std_re"], return_states=True)
        grads = torch.autograd.grad(loss, params)
        traj.append(float(loss))
        print("adam step %d: loss %.6f  (%.0f s)" % (t, float(loss), time.time() - t0), flush=True)
        if t == 1:
            n = lambda a: a.detach().numpy().astype(np.float32)
            flat = np.concatenate([g.numpy().ravel() for g in grads])
            out.update(loss_steps=np.array([float(l) for l in losses]), grad_norms=np.array([float(g.norm()) for g in grads]),
                       grads_sub16=flat[::16].astype(np.float32), grad_l2=float(np.sqrt((flat * flat).sum())),
                       vy_final=n(states[-1][1]), vx_final=n(states[-1][2]), d_final=n(states[-1][0]))
        with torch.no_grad():
            ps, m, v = o.adam_tf([p.detach() for p in params], list(grads), m, v, t, lr)
        params = [p.clone().requires_grad_(True) for p in ps]
        del loss, losses, states, grads
    out.update(loss_traj=np.array(traj), lr=lr, B=B, Y=Y, X=X, msteps=ms)
    return out


def k3d_case(B=2, Y=32, X=16, Z=16, nroll=2):
    """karman-3d (BASELINE configs[4]) at 32 x 16 x 16: one solver step, the network on that step's features, and a
    2-step roll-out (solver + correction), all from oracle/sol_oracle3d.py.  Weights = init_params(0) with seeded biases
    and the output layer scaled by 0.05 (an untrained Glorot corrector fed back blows the roll-out up); they are NOT
    stored (4 MB): the tests regenerate them with `k3d_params()` below.  The roll-out state is stored subsampled."""
    g = o3.geometry(Y, X, Z)
    d, v = o3.synthetic_state(B, Y, X, Z, 77)
    d, v = r32(d), tuple(r32(c) for c in v)
    re = torch.tensor(o3.RE_TRAIN[:B])
    params = k3d_params()
    std_v = (0.2, 0.25, 0.3)
    with torch.no_grad():
        d1, v1 = o3.karman3d_step(d, v, re, g)
        feat = o3.to_feature(v1, re) / torch.tensor(list(std_v) + [o3.STD_RE])
        net_out = o3.mars_moon3d(params, feat)
        states = o3.rollout(params, d, v, re, g, std_v, o3.STD_RE, nroll)
    n = lambda t: t.detach().numpy().astype(np.float32)
    dr, vr = states[-1]
    sub = lambda t: n(t).ravel()[::4]
    return dict(d=n(d), vy=n(v[0]), vx=n(v[1]), vz=n(v[2]), re=n(re), std_v=np.array(std_v), std_re=o3.STD_RE,
                d_out=n(d1), vy_out=n(v1[0]), vx_out=n(v1[1]), vz_out=n(v1[2]), net_out=n(net_out), nroll=nroll,
                roll_norms=np.array([float(t.norm()) for t in (dr,) + tuple(vr)]),
                roll_d_sub4=sub(dr), roll_vy_sub4=sub(vr[0]), roll_vx_sub4=sub(vr[1]), roll_vz_sub4=sub(vr[2]))


def k3d_params(last_layer_scale=0.05):
    ps = [r32(p) for p in o3.init_params(0)]
    gen = torch.Generator().manual_seed(99)
    for k, p in enumerate(ps):
        if p.dim() == 1:
            ps[k] = r32(0.01 * torch.randn(p.shape, generator=gen))
    ps[22] = r32(ps[22] * last_layer_scale)
    return ps


if __name__ == "__main__":
    if "--k3d" in sys.argv:
        np.savez_compressed(os.path.join(HERE, "karman3d_32x16x16.npz"), **k3d_case())
        print("karman3d_32x16x16.npz", os.path.getsize(os.path.join(HERE, "karman3d_32x16x16.npz")) // 1024, "KB")
        sys.exit(0)
    if "--sol32" in sys.argv:          # the 6-minute fixture is generated on request only
        np.savez_compressed(os.path.join(HERE, "train_128x64_sol32.npz"), **sol32_case())
        sys.exit(0)
    np.savez_compressed(os.path.join(HERE, "karman_step_16x8.npz"), **step_case(2, 16, 8, 1234))
    np.savez_compressed(os.path.join(HERE, "karman_step_64x32.npz"), **step_case(3, 64, 32, 1234))
    np.savez_compressed(os.path.join(HERE, "karman_step_16x8_dirichlet_before.npz"),
                        **step_case(2, 16, 8, 77, grad_pad="dirichlet0", inflow_order="before"))
    np.savez_compressed(os.path.join(HERE, "burgers_step_32x32.npz"), **burgers_case())
    np.savez_compressed(os.path.join(HERE, "train_16x8_sol2.npz"), **train_case())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KB")
