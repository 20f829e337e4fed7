"""Known-answer tests that pin the 3-D CPU oracle (oracle/sol_oracle3d.py).

The reference has no 3-D code at all (README.md:37-38); the oracle is the dimension-generic restatement of the 2-D path.
It is pinned (a) by reducing every local operator to the 2-D oracle on z-invariant inputs, (b) by analytic facts about
each operator, (c) by solver cross-checks (sparse LU vs the DST-preconditioned CG, residual of A p = div), (d) autograd
vs finite differences.
"""
import math

import numpy as np
import pytest
import torch

import sol_oracle as o2
import sol_oracle3d as o

torch.set_default_dtype(torch.float64)


def test_geometry_facts():
    g = o.geometry(32, 16, 16)
    # dx = 6.25: inflow row j = 1 (centre 9.375), i, k = 4..11 -> 64 cells; sphere r = 10 around (50,50,50): the 2x2x2 block
    assert g.dx == 6.25 and g.inflow.sum() == 64 and g.obstacle.sum() == 8
    assert np.array_equal(np.argwhere(g.obstacle)[0], [7, 7, 7]) and np.array_equal(np.argwhere(g.obstacle)[-1], [8, 8, 8])
    my, mx, mz = g.masks
    assert my.shape == (33, 16, 16) and mx.shape == (32, 17, 16) and mz.shape == (32, 16, 17)
    for j, i, k in np.argwhere(g.obstacle):
        assert my[j, i, k] == 0 and my[j + 1, i, k] == 0 and mx[j, i, k] == 0 and mx[j, i + 1, k] == 0
        assert mz[j, i, k] == 0 and mz[j, i, k + 1] == 0
    # open-boundary faces stay free, corner cell has six accessible neighbours
    assert my[0].min() == 1 and my[-1].min() == 1 and mz[:, :, 0].min() == 1
    assert g.diag[0, 0, 0] == -6 and g.diag.max() <= -1 and g.diag[7, 7, 7] == -3      # obstacle cell: 3 fluid neighbours
    # the flow-component BC mask: two inflow-side planes + the four lateral walls
    assert g.bc_mask.sum() == 2 * 16 * 16 + 31 * (16 * 16 - 14 * 14)
    # the z = const mid-plane of the 3-D masks equals the 2-D scene, and the cylinder is the 2-D disc extruded
    g2 = o2.geometry(32, 16)
    gc = o.geometry(32, 16, 16, obstacle="cylinder")
    assert np.array_equal(gc.obstacle[:, :, 3], g2.obstacle) and np.array_equal(gc.obstacle[:, :, 0], gc.obstacle[:, :, 15])
    assert np.array_equal(g.inflow[:, :, 8], g2.inflow) and np.array_equal(g.bc_mask[:, :, 5], np.maximum(g2.bc_mask, 0))


def test_full_size_geometry_counts():
    g = o.geometry(128, 64, 64)
    assert g.inflow.sum() == 3 * 32 * 32                 # rows j = 3..5, i, k = 16..47 (2-D: 96 = 3 x 32)
    n_ob = int(g.obstacle.sum())
    assert abs(n_ob - 4.0 / 3.0 * math.pi * 6.4 ** 3) < 0.03 * n_ob      # sphere of radius 6.4 cells
    idx = np.argwhere(g.obstacle)
    assert idx.min() == 26 and idx.max() == 37           # same extent as the 2-D disc


def test_local_operators_reduce_to_2d_on_z_invariant_fields():
    B, Y, X, Z = 2, 16, 8, 8
    gen = torch.Generator().manual_seed(5)
    vy2 = 1.0 + 0.3 * torch.randn(B, Y + 1, X, generator=gen)
    vx2 = 0.3 * torch.randn(B, Y, X + 1, generator=gen)
    d2 = torch.rand(B, Y, X, generator=gen)
    ex = lambda t, n: t.unsqueeze(-1).expand(*t.shape, n).contiguous()
    v3 = (ex(vy2, Z), ex(vx2, Z), torch.zeros(B, Y, X, Z + 1))
    # Laplacian
    assert torch.allclose(o.laplace_replicate(v3[0])[..., 3], o2.laplace_replicate(vy2), atol=1e-13)
    # advection (dx of the 2-D grid), including the zero-extrapolated density: interior z planes only see z-invariant data
    dx = 100.0 / X
    da, (ay, ax, az) = o.advect_mac(ex(d2, Z), v3, 1.0, dx)
    d2a, ay2, ax2 = o2.advect_mac(d2, vy2, vx2, 1.0, dx)
    assert torch.allclose(ay[..., 4], ay2, atol=1e-12) and torch.allclose(ax[..., 2], ax2, atol=1e-12)
    assert torch.allclose(da[..., 3], d2a, atol=1e-12) and float(az.abs().max()) == 0.0
    # divergence / gradient
    assert torch.allclose(o.divergence(v3)[..., 1], o2.divergence(vy2, vx2), atol=1e-13)
    p2 = torch.randn(B, Y, X, generator=gen)
    gy, gx, gz = o.grad_p(ex(p2, Z))
    gy2, gx2 = o2.grad_p(p2)
    assert torch.allclose(gy[..., 0], gy2) and torch.allclose(gx[..., 7], gx2) and float(gz.abs().max()) == 0.0


def test_diffusion_neumann_mode():
    n = (9, 6, 7)
    k = 2
    z = torch.arange(n[2]) + 0.5
    kz = math.pi * k / n[2]
    f = torch.cos(kz * z)[None, None, None, :].expand(1, *n).clone()
    assert torch.allclose(o.laplace_replicate(f), (2 * math.cos(kz) - 2) * f, atol=1e-12)


def test_integer_shift_advection():
    B, Y, X, Z = 1, 16, 8, 8
    g = o.geometry(Y, X, Z)
    gen = torch.Generator().manual_seed(1)
    d = torch.rand(B, Y, X, Z, generator=gen)
    v = (torch.zeros(B, Y + 1, X, Z), torch.zeros(B, Y, X + 1, Z), torch.full((B, Y, X, Z + 1), g.dx))   # one cell per step along +z
    da, va = o.advect_mac(d, v, 1.0, g.dx)
    assert torch.allclose(da[..., 1:], d[..., :-1], atol=1e-13) and float(da[..., 0].abs().max()) < 1e-13   # zero ghost ring
    assert torch.allclose(va[2], v[2])


def test_projection_is_divergence_free_and_solvers_agree():
    B, Y, X, Z = 2, 16, 8, 8
    g = o.geometry(Y, X, Z)
    _, v = o.synthetic_state(B, Y, X, Z, 11)
    pv, info = o.project(v, g, return_info=True)
    # A p = div, and the projected field is divergence free on active cells that touch no domain-boundary face
    # (grad_pad 'replicate' leaves the outer faces uncorrected)
    assert float((o.apply_A(info["pressure"], g) - info["divergence"]).abs().max()) < 1e-10
    div = o.divergence(pv)
    inner = torch.zeros(Y, X, Z)
    inner[1:-1, 1:-1, 1:-1] = 1.0
    assert float((div * inner * torch.as_tensor(g.active)).abs().max()) < 1e-10
    pd = o.project(v, g, grad_pad="dirichlet0")
    assert float((o.divergence(pd) * torch.as_tensor(g.active)).abs().max()) < 1e-10
    # the DST-preconditioned CG (used where sparse LU does not fit) solves the same system
    p2 = o.solve_pcg(info["divergence"].numpy(), g)
    assert np.abs(p2 - info["pressure"].numpy()).max() < 1e-10 * np.abs(p2).max()


def test_rect_solve_inverts_the_empty_box_laplacian():
    n = (8, 6, 5)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1,) + n)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (1, 1)))
    Mx = 6 * x - (xp[:, 2:, 1:-1, 1:-1] + xp[:, :-2, 1:-1, 1:-1] + xp[:, 1:-1, 2:, 1:-1] + xp[:, 1:-1, :-2, 1:-1] +
                  xp[:, 1:-1, 1:-1, 2:] + xp[:, 1:-1, 1:-1, :-2])
    assert np.allclose(o.rect_solve(Mx, o._rect_eigenvalues(n)), x, atol=1e-12)


def test_step_autograd_matches_finite_differences():
    B, Y, X, Z = 1, 16, 8, 8
    g = o.geometry(Y, X, Z)
    d, v = o.synthetic_state(B, Y, X, Z, 7)
    re = torch.tensor([1.6e5])
    gen = torch.Generator().manual_seed(2)
    w = [torch.randn(c.shape, generator=gen) for c in v]
    f = lambda vv: sum((a * b).sum() for a, b in zip(o.karman3d_step(d, vv, re, g)[1], w))
    vr = tuple(c.clone().requires_grad_(True) for c in v)
    f(vr).backward()
    for comp, (j, i, k) in ((0, (5, 3, 4)), (1, (9, 8, 2)), (2, (7, 4, 8)), (0, (0, 0, 0))):
        e = 1e-5
        vp = [c.clone() for c in v]
        vm = [c.clone() for c in v]
        vp[comp][0, j, i, k] += e
        vm[comp][0, j, i, k] -= e
        fd = float(f(tuple(vp)) - f(tuple(vm))) / (2 * e)
        assert abs(fd - float(vr[comp].grad[0, j, i, k])) < 1e-6 * max(1.0, abs(fd))


def test_conv3d_against_direct_sum():
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(1, 6, 5, 7, 3, generator=gen)
    w = torch.randn(5, 5, 5, 3, 2, generator=gen)
    b = torch.randn(2, generator=gen)
    y = o.conv3d_same(x, w, b)
    xp = F_pad5(x)
    for (j, i, k) in ((0, 0, 0), (3, 2, 5), (5, 4, 6)):
        ref = b.clone()
        for dy in range(5):
            for dx in range(5):
                for dz in range(5):
                    ref = ref + xp[0, j + dy, i + dx, k + dz] @ w[dy, dx, dz]
        assert torch.allclose(y[0, j, i, k], ref, atol=1e-12)


def F_pad5(x):
    return torch.nn.functional.pad(x, (0, 0, 2, 2, 2, 2, 2, 2))


def test_feature_and_pad_glue():
    B, Y, X, Z = 2, 4, 2, 2
    v = (torch.arange(B * (Y + 1) * X * Z, dtype=torch.float64).reshape(B, Y + 1, X, Z),
         torch.ones(B, Y, X + 1, Z), 2 * torch.ones(B, Y, X, Z + 1))
    re = torch.tensor([3.0, 5.0])
    f = o.to_feature(v, re)
    assert f.shape == (B, Y, X, Z, 4) and torch.equal(f[..., 0], v[0][:, :Y]) and float(f[1, 0, 0, 0, 3]) == 5.0
    c = o.to_staggered(torch.ones(B, Y, X, Z, 3))
    assert c[0].shape == (B, Y + 1, X, Z) and float(c[0][:, Y].abs().max()) == 0 and float(c[1][:, :, X].abs().max()) == 0
    assert float(c[2][..., Z].abs().max()) == 0 and float(c[2][..., :Z].min()) == 1
    assert sum(int(np.prod(s)) for s in o.mars_moon3d_param_shapes()) == 125 * (4 * 32 + 10 * 32 * 32 + 32 * 3) + 11 * 32 + 3
