"""Known-answer tests that pin the CPU oracle (oracle/sol_oracle.py).

The reference has no tests or golden vectors, so the oracle is pinned
by analytic facts about each operator plus the geometry facts measured in SURVEY appendix B.
"""
import math

import numpy as np
import pytest
import scipy.sparse.linalg as spla
import torch

import sol_oracle as o

torch.set_default_dtype(torch.float64)


@pytest.mark.parametrize("Y,X,n_in,n_ob,n_bc", [(64, 32, 16, 32, 190), (128, 64, 96, 124, 382)])
def test_geometry_facts(Y, X, n_in, n_ob, n_bc):
    g = o.geometry(Y, X)
    assert g.inflow.sum() == n_in
    assert g.obstacle.sum() == n_ob
    assert g.bc_mask.sum() == n_bc
    # faces touching an obstacle cell are closed, open-boundary faces stay free
    assert g.my[0].min() == 1.0 and g.my[-1].min() == 1.0
    assert g.mx[:, 0].min() == 1.0 and g.mx[:, -1].min() == 1.0
    jj, ii = np.nonzero(g.obstacle)
    for j, i in zip(jj, ii):
        assert g.my[j, i] == 0 and g.my[j + 1, i] == 0 and g.mx[j, i] == 0 and g.mx[j, i + 1] == 0
    # interior fluid cell far from the sphere: diag -4; corner cell still -4 (outside accessible)
    assert g.diag[0, 0] == -4 and g.diag[1, 1] == -4
    assert g.diag.max() <= -1


def test_diffusion_fourier_mode_neumann():
    # cos modes are eigenvectors of the replicate-padded (Neumann) 5-point Laplacian
    H, W = 33, 16
    j = torch.arange(H) + 0.5
    ky = math.pi * 3 / H
    f = torch.cos(ky * j)[None, :, None].expand(1, H, W).clone()
    lam = 2 * math.cos(ky) - 2
    assert torch.allclose(o.laplace_replicate(f), lam * f, atol=1e-12)


def test_laplace_replicate_is_symmetric():
    H, W = 7, 5
    n = H * W
    L = torch.stack([o.laplace_replicate(torch.eye(n)[k].reshape(1, H, W)).reshape(-1) for k in range(n)])
    assert torch.allclose(L, L.T)


def test_advection_of_constant_field_is_identity():
    B, Y, X = 2, 16, 8
    g = o.geometry(Y, X)
    gen = torch.Generator().manual_seed(3)
    vy = torch.full((B, Y + 1, X), 0.7)
    vx = torch.full((B, Y, X + 1), -0.3)
    d = torch.rand(B, Y, X, generator=gen)
    _, ay, ax = o.advect_mac(d, vy, vx, 1.0, g.dx)
    assert torch.allclose(ay, vy) and torch.allclose(ax, vx)


def test_advection_integer_shift():
    # uniform velocity of exactly one cell per step shifts the field by one cell (interior)
    B, Y, X = 1, 16, 8
    g = o.geometry(Y, X)
    gen = torch.Generator().manual_seed(4)
    d = torch.rand(B, Y, X, generator=gen)
    vy = torch.full((B, Y + 1, X), g.dx)
    vx = torch.zeros(B, Y, X + 1)
    d2, _, _ = o.advect_mac(d, vy, vx, 1.0, g.dx)
    assert torch.allclose(d2[:, 1:], d[:, :-1])
    assert torch.allclose(d2[:, 0], torch.zeros(X))      # zero extrapolation upstream of the inlet


def test_density_zero_extrapolation_blend():
    # half a cell of inflow from outside: value = 0.5*edge (one ghost ring of zeros)
    B, Y, X = 1, 16, 8
    g = o.geometry(Y, X)
    d = torch.ones(B, Y, X)
    vy = torch.full((B, Y + 1, X), 0.5 * g.dx)
    vx = torch.zeros(B, Y, X + 1)
    d2, _, _ = o.advect_mac(d, vy, vx, 1.0, g.dx)
    assert torch.allclose(d2[:, 0], torch.full((X,), 0.5))
    assert torch.allclose(d2[:, 1:], torch.ones(Y - 1, X))


@pytest.mark.parametrize("Y,X", [(16, 8), (64, 32)])
def test_pressure_matrix_matches_matrix_free(Y, X):
    g = o.geometry(Y, X)
    A = g.pressure_matrix()
    assert abs(A - A.T).max() == 0
    gen = torch.Generator().manual_seed(0)
    p = torch.randn(2, Y, X, generator=gen)
    ref = torch.as_tensor((A @ p.reshape(2, -1).numpy().T).T).reshape(2, Y, X)
    assert torch.allclose(o.apply_A(p, g), ref, atol=1e-12)


def test_cg_matches_direct_solve_and_iteration_counts():
    Y, X = 64, 32
    g = o.geometry(Y, X)
    gen = torch.Generator().manual_seed(1)
    rhs = 0.01 * torch.randn(2, Y, X, generator=gen) * torch.as_tensor(g.active)
    p_cg, its = o.cg_reference(rhs, g, accuracy=1e-5)
    p_lu = torch.as_tensor(spla.spsolve(g.pressure_matrix(), rhs.reshape(2, -1).numpy().T).T).reshape(2, Y, X)
    assert (p_cg - p_lu).abs().max() < 1e-3 * p_lu.abs().max()
    assert 60 <= int(its.max()) <= 130          # 84/106 at 64x32
    p_tight, _ = o.cg_reference(rhs, g, accuracy=1e-12)
    assert (p_tight - p_lu).abs().max() < 1e-9


def test_projection_is_divergence_free_inside_and_idempotent():
    Y, X = 64, 32
    g = o.geometry(Y, X)
    d, vy, vx = o.synthetic_state(2, Y, X, 7, project_it=False)
    py, px = o.project(vy, vx, g)
    div = o.divergence(py, px)
    act = torch.as_tensor(g.active)
    assert (div * act)[:, 1:-1, 1:-1].abs().max() < 1e-12
    # documented PhiFlow-1.x quirk (Q5): boundary faces get no pressure gradient
    assert torch.allclose(py[:, 0], vy[:, 0]) and torch.allclose(px[:, :, -1], vx[:, :, -1])
    # with Dirichlet-consistent padding the whole field is divergence free
    qy, qx = o.project(vy, vx, g, grad_pad="dirichlet0")
    assert (o.divergence(qy, qx) * act).abs().max() < 1e-12
    q2y, q2x = o.project(qy, qx, g, grad_pad="dirichlet0")
    assert torch.allclose(q2y, qy, atol=1e-12) and torch.allclose(q2x, qx, atol=1e-12)


def test_staggered_glue_roundtrip():
    B, Y, X = 2, 8, 4
    vy = torch.randn(B, Y + 1, X)
    vx = torch.randn(B, Y, X + 1)
    st = o.staggered_tensor(vy, vx)
    assert st.shape == (B, Y + 1, X + 1, 2)
    uy, ux = o.unstack_staggered(st)
    assert torch.equal(uy, vy) and torch.equal(ux, vx)
    re = torch.tensor([2.0, 3.0])
    f = o.to_feature(vy, vx, re)
    assert f.shape == (B, Y, X, 3)
    assert torch.equal(f[..., 0], vy[:, :Y]) and torch.equal(f[..., 1], vx[:, :, :X])
    assert torch.equal(f[1, ..., 2], torch.full((Y, X), 3.0))
    cy, cx = o.to_staggered(f[..., :2])
    assert cy.shape == vy.shape and cx.shape == vx.shape
    assert torch.equal(cy[:, Y], torch.zeros(B, X)) and torch.equal(cx[:, :, X], torch.zeros(B, Y))


def test_mars_moon_param_count_and_flops():
    ps = o.init_params(0)
    assert sum(p.numel() for p in ps) == 260354
    x = torch.randn(1, 8, 4, 3)
    assert o.mars_moon(ps, x).shape == (1, 8, 4, 2)


def test_adam_tf_first_step_is_lr_sign():
    p = [torch.tensor([1.0, -2.0])]
    gr = [torch.tensor([0.5, -0.25])]
    m = [torch.zeros(2)]
    v = [torch.zeros(2)]
    p2, _, _ = o.adam_tf(p, gr, m, v, 1, 1e-3)
    # t=1: lr_t*m/(sqrt(v)+eps) = lr * g/(|g| + eps*sqrt(1-b2)^-1...) ~ lr*sign(g)
    assert torch.allclose(p2[0], p[0] - 1e-3 * torch.sign(gr[0]), atol=1e-9)


def test_burgers_fft_diffusion_equals_circulant_matrices():
    H, W = 33, 32
    gen = torch.Generator().manual_seed(5)
    f = torch.randn(2, H, W, generator=gen)
    Cy, Cx = o.burgers_diffusion_matrices(H, W, 0.01)
    ref = o.diffuse_periodic_fft(f, 0.01)
    assert torch.allclose(Cy @ f @ Cx.T, ref, atol=1e-12)
    assert torch.allclose(Cy, Cy.T) and torch.allclose(Cx, Cx.T)


def test_burgers_constant_velocity_is_fixed_point():
    vy = torch.full((1, 33, 32), 0.3)
    vx = torch.full((1, 32, 33), -0.2)
    ay, ax = o.burgers_step(vy, vx, 0.1)
    assert torch.allclose(ay, vy) and torch.allclose(ax, vx)


def test_step_autograd_matches_finite_differences():
    Y, X = 16, 8
    g = o.geometry(Y, X)
    d, vy, vx = o.synthetic_state(1, Y, X, 11)
    re = torch.tensor([1.0e3])
    gen = torch.Generator().manual_seed(12)
    wy = torch.randn(1, Y + 1, X, generator=gen)
    wx = torch.randn(1, Y, X + 1, generator=gen)

    def fn(a, b):
        _, py, px = o.karman_step(d, a, b, re, g)
        return (py * wy).sum() + (px * wx).sum()

    vy = vy.clone().requires_grad_(True)
    vx = vx.clone().requires_grad_(True)
    fn(vy, vx).backward()
    eps = 1e-6
    for (t, gr) in ((vy, vy.grad), (vx, vx.grad)):
        for idx in [(0, 3, 2), (0, 9, 5), (0, 15, 7)]:
            tp = t.detach().clone(); tp[idx] += eps
            tm = t.detach().clone(); tm[idx] -= eps
            a = (tp, vx.detach()) if t is vy else (vy.detach(), tp)
            b = (tm, vx.detach()) if t is vy else (vy.detach(), tm)
            fd = (fn(*a) - fn(*b)) / (2 * eps)
            assert abs(fd - gr[idx]) < 1e-5 * max(1.0, abs(fd)), (idx, fd, gr[idx])


# ---------------------------------------------------------------------------------------------
# the recalled PhiFlow choices are switchable
# ---------------------------------------------------------------------------------------------
def test_q3_inflow_antialias_option():
    g0, g1 = o.geometry(64, 32), o.geometry(64, 32, inflow_antialias=True)
    assert g0.inflow.sum() == 16 and set(np.unique(g0.inflow)) == {0.0, 1.0}
    # anti-aliased: a one-cell linear ramp -- fractional values appear, cells well inside stay 1, the integral is the box area
    assert 0.0 < g1.inflow[g1.inflow > 0].min() < 1.0 and g1.inflow.max() == 1.0
    assert abs(g1.inflow.sum() * g1.dx ** 2 - 5.0 * 50.0) < 0.25 * 5.0 * 50.0
    assert np.array_equal(g0.active, g1.active) and np.array_equal(g0.my, g1.my)        # the obstacle / pressure matrix are untouched


def test_q4_density_extrapolation_variants():
    f = torch.ones(1, 4, 4)
    ly = torch.tensor([[[-0.5, -0.25, -0.75, 1.5]]])
    lx = torch.full_like(ly, 1.0)
    ring = o._sample(f, ly, lx, "zero")          # blends to the zero ghost cell centre at -1
    box = o._sample(f, ly, lx, "zero_box")       # 1 inside the box (edge at -0.5), 0 outside
    assert torch.allclose(ring, torch.tensor([[[0.5, 0.75, 0.25, 1.0]]]))
    assert torch.allclose(box, torch.tensor([[[1.0, 1.0, 0.0, 1.0]]]))
    # in a step the two differ only next to the open boundary
    g = o.geometry(16, 8)
    d, vy, vx = o.synthetic_state(1, 16, 8, 3)
    re = torch.tensor([1e5])
    a = o.karman_step(d, vy, vx, re, g)[0]
    b = o.karman_step(d, vy, vx, re, g, den_mode="zero_box")[0]
    assert torch.allclose(a[:, 2:-2, 2:-2], b[:, 2:-2, 2:-2]) and not torch.allclose(a, b)


def test_q6_sparse_cg_restatement_vs_converged_solve():
    g = o.geometry(32, 16)
    d, vy, vx = o.synthetic_state(2, 32, 16, 9, project_it=False)
    re = torch.tensor([1e5, 2e5])
    a = o.karman_step(d, vy, vx, re, g, solver="direct")
    b = o.karman_step(d, vy, vx, re, g, solver="cg")            # accuracy 1e-5 on max|r|, batch-global stop
    err = float((a[1] - b[1]).abs().max())
    assert 0.0 < err < 1e-4


def test_q7_periodic_duplicated_face_variants():
    B, Y, X = 1, 8, 8
    gen = torch.Generator().manual_seed(0)
    # a field that IS periodic with a consistent duplicated face
    cy = torch.randn(B, Y, X, generator=gen) * 0.3
    cx = torch.randn(B, Y, X, generator=gen) * 0.3
    vy = torch.cat([cy, cy[:, :1]], dim=1)
    vx = torch.cat([cx, cx[:, :, :1]], dim=2)
    ay, ax = o.burgers_step(vy, vx, 0.1, periodic_faces="domain")
    assert torch.equal(ay[:, -1], ay[:, 0]) and torch.equal(ax[:, :, -1], ax[:, :, 0])      # the copy stays a copy
    by, bx = o.burgers_step(vy, vx, 0.1, periodic_faces="array")
    assert by.shape == ay.shape and not torch.allclose(ay, by)                              # the recalled default wraps modulo Y+1
    # zero velocity, zero viscosity: both are the identity
    z = torch.zeros_like(vy), torch.zeros_like(vx)
    for mode in ("array", "domain"):
        ry, rx = o.burgers_step(*z, 0.1, nu=0.0, fy=vy, fx=vx, periodic_faces=mode)
        assert torch.allclose(ry, 0.1 * vy) and torch.allclose(rx, 0.1 * vx)
    with pytest.raises(ValueError):
        o.burgers_step(vy, vx, 0.1, periodic_faces="nope")
