"""Diagnostic (not a pytest file): prints HIP-vs-oracle errors for every op on the GPU box.
Usage: python tests/gpu_probe.py [section ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np
import torch

import sol_amd
import sol_oracle as o
from sol_amd import ops

torch.set_default_dtype(torch.float64)
dev = "cuda"


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def masks_for(Y, X):
    g = o.geometry(Y, X)
    return g, ops.SceneMasks(g.active, g.inflow, g.bc_mask, g.bc_mask)


def probe_step():
    for (B, Y, X) in [(2, 16, 8), (3, 64, 32), (2, 128, 64)]:
        g, mk = masks_for(Y, X)
        d, vy, vx = o.synthetic_state(B, Y, X, 1234)
        re = torch.tensor(o.RE_TRAIN[:B])
        vy = vy.clone().requires_grad_(True)
        vx = vx.clone().requires_grad_(True)
        d2, py, px = o.karman_step(d, vy, vx, re, g)
        gen = torch.Generator().manual_seed(5)
        wy = torch.randn(py.shape, generator=gen)
        wx = torch.randn(px.shape, generator=gen)
        ((py * wy).sum() + (px * wx).sum()).backward()
        cfg = ops.karman_cfg(B, Y, X, g.dx, masks=mk)
        hvy = vy.detach().float().to(dev).requires_grad_(True)
        hvx = vx.detach().float().to(dev).requires_grad_(True)
        info = {}
        t0 = time.time()
        hd, hpy, hpx = ops.karman_step(d.float().to(dev), hvy, hvx, re.float().to(dev), cfg, mk, info)
        torch.cuda.synchronize()
        t1 = time.time()
        ((hpy * wy.float().to(dev)).sum() + (hpx * wx.float().to(dev)).sum()).backward()
        torch.cuda.synchronize()
        print("step %dx%d B=%d: d %.2e vy %.2e vx %.2e | gvy %.2e gvx %.2e | iters %s bwd %s (%.1f ms first call)" % (
            Y, X, B, rel(hd, d2), rel(hpy, py), rel(hpx, px), rel(hvy.grad, vy.grad), rel(hvx.grad, vx.grad),
            info["iterations"].tolist(), info["iterations_bwd"].tolist(), (t1 - t0) * 1e3))


def probe_conv():
    gen = torch.Generator().manual_seed(0)
    for (B, H, W, cin, cout, lrelu, res) in [(2, 16, 8, 3, 32, True, False), (2, 16, 8, 32, 32, True, True),
                                             (2, 16, 8, 32, 2, False, False), (1, 64, 32, 32, 32, True, True),
                                             (2, 128, 64, 32, 32, True, False), (2, 128, 64, 3, 32, True, False),
                                             (2, 128, 64, 32, 2, False, False)]:
        x = torch.randn(B, H, W, cin, generator=gen, requires_grad=True)
        w = (torch.randn(5, 5, cin, cout, generator=gen) * 0.05).requires_grad_(True)
        b = (torch.randn(cout, generator=gen) * 0.1).requires_grad_(True)
        r = torch.randn(B, H, W, cout, generator=gen, requires_grad=True) if res else None
        y = o._conv(x, w, b)
        if res:
            y = y + r
        if lrelu:
            y = torch.nn.functional.leaky_relu(y, 0.3)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        hx = x.detach().float().to(dev).requires_grad_(True)
        hw = w.detach().float().to(dev).requires_grad_(True)
        hb = b.detach().float().to(dev).requires_grad_(True)
        hr = r.detach().float().to(dev).requires_grad_(True) if res else None
        hy = ops.conv5x5(hx, hw, hb, hr, lrelu, 0.3)
        (hy * gy.float().to(dev)).sum().backward()
        torch.cuda.synchronize()
        print("conv B%d %dx%d %d->%d lrelu=%d res=%d: y %.2e dx %.2e dw %.2e db %.2e%s" % (
            B, H, W, cin, cout, lrelu, res, rel(hy, y), rel(hx.grad, x.grad), rel(hw.grad, w.grad), rel(hb.grad, b.grad),
            (" dres %.2e" % rel(hr.grad, r.grad)) if res else ""))


def oracle_train(B, Y, X, ms, seed=0):
    g = o.geometry(Y, X)
    d, vy, vx = o.synthetic_state(B, Y, X, 1234)
    re = torch.tensor([o.RE_TRAIN[i % 6] for i in range(B)])
    gts = [o.synthetic_state(B, Y, X, 4321 + i, project_it=False) for i in range(ms)]
    gt_vy = [s[1] for s in gts]
    gt_vx = [s[2] for s in gts]
    params = [p.clone().requires_grad_(True) for p in o.init_params(seed)]
    # non-zero biases so that their gradients are exercised
    gen = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for p in params:
            if p.dim() == 1:
                p.copy_(0.01 * torch.randn(p.shape, generator=gen))
    std_v = (0.2, 0.25)
    loss, losses, states = o.unrolled_loss(params, d, vy, vx, re, gt_vy, gt_vx, g, std_v, o.STD_RE, return_states=True)
    loss.backward()
    return g, d, vy, vx, re, gt_vy, gt_vx, params, std_v, loss, losses, states


def probe_train():
    for (B, Y, X, ms) in [(2, 16, 8, 2), (3, 64, 32, 4)]:
        t0 = time.time()
        g, d, vy, vx, re, gt_vy, gt_vx, params, std_v, loss, losses, states = oracle_train(B, Y, X, ms)
        t1 = time.time()
        mk = ops.SceneMasks(g.active, g.inflow, g.bc_mask, g.bc_mask)
        net = sol_amd.model_mars_moon(cin=3, cout=2, seed=0)
        net.set_weights([p.detach().numpy() for p in params])
        tr = sol_amd.SolTrainer(net, mk, B, Y, X, ms, g.dx, std_v, o.STD_RE)
        f = lambda t: t.float().to(dev).contiguous()
        hl = tr.fwd_bwd(f(d), f(vy), f(vx), f(re), f(torch.stack(gt_vy)), f(torch.stack(gt_vx)), want_final=True)
        torch.cuda.synchronize()
        gflat = torch.cat([p.grad.reshape(-1) for p in params])
        print("train %dx%d B=%d ms=%d: loss %.8g vs %.8g (rel %.2e) grad rel %.2e | final vy %.2e vx %.2e d %.2e | oracle %.1fs" % (
            Y, X, B, ms, float(hl), float(loss), abs(float(hl) - float(loss)) / abs(float(loss)), rel(tr.grads, gflat),
            rel(tr.final[1], states[-1][1]), rel(tr.final[2], states[-1][2]), rel(tr.final[0], states[-1][0]), t1 - t0))
        print("   per-step loss rel:", [abs(float(a) - float(b)) / abs(float(b)) for a, b in zip(tr.loss_steps.tolist(), losses)])
        off = net.offsets
        pg = [rel(tr.grads[off[k]:off[k + 1]], params[k].grad.reshape(-1)) for k in range(len(params))]
        print("   per-tensor grad rel:", ["%.1e" % v for v in pg])
        print("   iters fwd", tr.iters_fwd.tolist(), "bwd", tr.iters_bwd.tolist())
        # Adam
        m0 = [torch.zeros_like(p) for p in params]
        v0 = [torch.zeros_like(p) for p in params]
        p2, _, _ = o.adam_tf([p.detach() for p in params], [p.grad for p in params], m0, v0, 1, 1e-4)
        tr.apply_gradients(1e-4)
        print("   adam rel %.2e" % rel(net.params, torch.cat([p.reshape(-1) for p in p2])))
        # per-op autograd composition must agree with the fused path
        state = sol_amd.Fluid(sol_amd.Domain([Y, X], box=sol_amd.box[0:200, 0:100]), batch_size=B)


def probe_burgers():
    B, Y, X = 5, 32, 32
    gen = torch.Generator().manual_seed(3)
    vy = (0.3 * o._smooth(torch.randn(B, Y + 1, X, generator=gen))).requires_grad_(True)
    vx = (0.3 * o._smooth(torch.randn(B, Y, X + 1, generator=gen))).requires_grad_(True)
    fy = 0.15 * o._smooth(torch.randn(B, Y + 1, X, generator=gen))
    fx = 0.15 * o._smooth(torch.randn(B, Y, X + 1, generator=gen))
    dt, nu = 0.1, 0.1
    ay, ax = o.burgers_step(vy, vx, dt, nu, fy, fx)
    wy = torch.randn(ay.shape, generator=gen)
    wx = torch.randn(ax.shape, generator=gen)
    ((ay * wy).sum() + (ax * wx).sum()).backward()
    cfg = sol_amd._lib.BurgersCfg(B, Y, X, 1.0, dt)
    circ = ops.burgers_circ(Y, X, dt * nu)
    hvy = vy.detach().float().to(dev).requires_grad_(True)
    hvx = vx.detach().float().to(dev).requires_grad_(True)
    hy, hx = ops.burgers_step(hvy, hvx, fy.float().to(dev), fx.float().to(dev), cfg, circ)
    ((hy * wy.float().to(dev)).sum() + (hx * wx.float().to(dev)).sum()).backward()
    torch.cuda.synchronize()
    print("burgers 32x32: vy %.2e vx %.2e | gvy %.2e gvx %.2e" % (rel(hy, ay), rel(hx, ax), rel(hvy.grad, vy.grad), rel(hvx.grad, vx.grad)))


if __name__ == "__main__":
    which = sys.argv[1:] or ["step", "conv", "burgers", "train"]
    print("device:", torch.cuda.get_device_name(0), "lib:", sol_amd.lib_path())
    for w in which:
        try:
            globals()["probe_" + w]()
        except Exception as e:   # keep going: one call to the GPU box is expensive
            import traceback
            traceback.print_exc()
            print("PROBE %s FAILED: %s" % (w, e))
