"""The committed golden vectors (tests/golden/*.npz, made by make_golden.py) must be reproduced
by the oracle: a regression pin for the CPU restatement (the reference ships no vectors)."""
import os

import numpy as np
import pytest
import torch

import sol_oracle as o

torch.set_default_dtype(torch.float64)


def rel(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-300))


@pytest.mark.parametrize("name,kw", [("karman_step_16x8", {}), ("karman_step_64x32", {}),
                                     ("karman_step_16x8_dirichlet_before", dict(grad_pad="dirichlet0", inflow_order="before"))])
def test_karman_step_golden(golden_dir, name, kw):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    B, Y, X = z["d"].shape
    t = lambda k: torch.as_tensor(z[k].astype(np.float64))
    vy = t("vy").requires_grad_(True)
    vx = t("vx").requires_grad_(True)
    d2, py, px = o.karman_step(t("d"), vy, vx, t("re"), o.geometry(Y, X), **kw)
    ((py * t("wy")).sum() + (px * t("wx")).sum()).backward()
    # fixtures are stored in fp32 -> 6e-8 relative rounding
    assert rel(d2.detach(), z["d_out"]) < 2e-7 and rel(py.detach(), z["vy_out"]) < 2e-7 and rel(px.detach(), z["vx_out"]) < 2e-7
    assert rel(vy.grad, z["g_vy"]) < 2e-7 and rel(vx.grad, z["g_vx"]) < 2e-7


def test_burgers_step_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "burgers_step_32x32.npz"))
    t = lambda k: torch.as_tensor(z[k].astype(np.float64))
    vy = t("vy").requires_grad_(True)
    vx = t("vx").requires_grad_(True)
    ay, ax = o.burgers_step(vy, vx, float(z["dt"]), float(z["nu"]), t("fy"), t("fx"))
    ((ay * t("wy")).sum() + (ax * t("wx")).sum()).backward()
    assert rel(ay.detach(), z["vy_out"]) < 2e-7 and rel(ax.detach(), z["vx_out"]) < 2e-7
    assert rel(vy.grad, z["g_vy"]) < 2e-7 and rel(vx.grad, z["g_vx"]) < 2e-7


def golden_train_params(z):
    """seed-0 glorot weights (fp32 rounded) with the stored biases"""
    params = [p.float().double() for p in o.init_params(0)]
    off = 0
    for p in params:
        if p.dim() == 1:
            p.copy_(torch.as_tensor(z["biases"][off:off + p.numel()].astype(np.float64)))
            off += p.numel()
    return params


def test_train_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "train_16x8_sol2.npz"))
    t = lambda k: torch.as_tensor(z[k].astype(np.float64))
    B, Y, X = z["d"].shape
    params = [p.requires_grad_(True) for p in golden_train_params(z)]
    gt_vy, gt_vx = t("gt_vy"), t("gt_vx")
    loss, losses, states = o.unrolled_loss(params, t("d"), t("vy"), t("vx"), t("re"), list(gt_vy), list(gt_vx),
                                           o.geometry(Y, X), tuple(z["std_v"]), float(z["std_re"]), return_states=True)
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) < 1e-9 * abs(float(z["loss"]))
    grads = torch.cat([p.grad.reshape(-1) for p in params])
    assert rel(grads[::16], z["grads_sub16"]) < 2e-7
    norms = np.array([float(p.grad.norm()) for p in params])
    assert np.allclose(norms, z["grad_norms"], rtol=1e-9)
    assert rel(states[-1][1].detach(), z["vy_final"]) < 2e-7
