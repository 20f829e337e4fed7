"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against the CPU
oracle and the committed golden vectors.  Tolerances: the north star asks for <= 1e-5 relative
L2 on velocity/density fields (fp32 kernels vs the float64 oracle with an exactly solved
pressure system); gradients pass through two fp32 CG solves per step and are held to 1e-4.
"""
import os

import numpy as np
import pytest
import torch

import sol_amd
import sol_oracle as o
from sol_amd import ops
from test_golden_oracle import golden_train_params

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL_FIELD = 1e-5     # north star: relative L2 on velocity / density
TOL_GRAD = 1e-4


def rel(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b), dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-300))


def f32(t):
    return torch.as_tensor(np.asarray(t), dtype=torch.float32).to(DEV).contiguous()


def masks_for(Y, X, precondition=True, pressure_solver="auto"):
    g = o.geometry(Y, X)
    return g, ops.SceneMasks(g.active, g.inflow, g.bc_mask, g.bc_mask, precondition=precondition, pressure_solver=pressure_solver)


@pytest.fixture(scope="module", autouse=True)
def _loaded_native_library():
    lib = sol_amd.load()
    assert os.path.exists(sol_amd.lib_path())
    assert torch.cuda.is_available()
    yield lib


# ---------------------------------------------------------------------------------------------
# solver step
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precond", [True, False])
@pytest.mark.parametrize("name,kw", [("karman_step_16x8", {}), ("karman_step_64x32", {}),
                                     ("karman_step_16x8_dirichlet_before", dict(grad_pad="dirichlet0", inflow_order="before"))])
def test_karman_step_against_golden(golden_dir, name, kw, precond):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    B, Y, X = z["d"].shape
    g, mk = masks_for(Y, X, precond, "cg")                          # the iterative solvers (the direct one: next test)
    assert (mk.coarse_inv is not None) == (precond and Y >= 64)     # two-level CG only where the grid allows it
    assert mk.direct is None
    cfg = ops.karman_cfg(B, Y, X, g.dx, masks=mk, **kw)
    vy = f32(z["vy"]).requires_grad_(True)
    vx = f32(z["vx"]).requires_grad_(True)
    info = {}
    d2, py, px = ops.karman_step(f32(z["d"]), vy, vx, f32(z["re"]), cfg, mk, info)
    ((py * f32(z["wy"])).sum() + (px * f32(z["wx"])).sum()).backward()
    assert rel(d2, z["d_out"]) < TOL_FIELD and rel(py, z["vy_out"]) < TOL_FIELD and rel(px, z["vx_out"]) < TOL_FIELD
    assert rel(vy.grad, z["g_vy"]) < TOL_GRAD and rel(vx.grad, z["g_vx"]) < TOL_GRAD
    assert int(info["iterations"].min()) > 5 and int(info["iterations"].max()) < 2000


def test_direct_solver_at_the_reference_training_size(golden_dir):
    """64x32 (the reference's own recipe, karman-2d/Makefile:78-80; BASELINE configs[1]): the direct pressure solver's
    small-grid form (LDS-resident sine transforms + capacitance correction) -- no CG iteration at all -- against the golden
    step (fields and input gradients) and against the preconditioned CG on the same inputs."""
    z = np.load(os.path.join(golden_dir, "karman_step_64x32.npz"))
    B, Y, X = z["d"].shape
    g, mk = masks_for(Y, X)                                          # "auto": direct where it is built
    assert mk.direct is not None and sol_amd.load().sol_karman_direct_supported(Y, X) == 1
    assert sol_amd.load().sol_karman_direct_supported(16, 8) == 0
    cfg = ops.karman_cfg(B, Y, X, g.dx, masks=mk)
    vy = f32(z["vy"]).requires_grad_(True)
    vx = f32(z["vx"]).requires_grad_(True)
    info = {}
    d2, py, px = ops.karman_step(f32(z["d"]), vy, vx, f32(z["re"]), cfg, mk, info)
    ((py * f32(z["wy"])).sum() + (px * f32(z["wx"])).sum()).backward()
    assert int(info["iterations"].max()) == 0 and int(info["iterations_bwd"].max()) == 0
    assert rel(d2, z["d_out"]) < TOL_FIELD and rel(py, z["vy_out"]) < TOL_FIELD and rel(px, z["vx_out"]) < TOL_FIELD
    assert rel(vy.grad, z["g_vy"]) < TOL_GRAD and rel(vx.grad, z["g_vx"]) < TOL_GRAD
    g2, mk2 = masks_for(Y, X, True, "cg")
    cfg2 = ops.karman_cfg(B, Y, X, g.dx, masks=mk2)
    _, qy, qx = ops.karman_step(f32(z["d"]), f32(z["vy"]), f32(z["vx"]), f32(z["re"]), cfg2, mk2)
    assert rel(py, qy) < TOL_FIELD and rel(px, qx) < TOL_FIELD


@pytest.mark.parametrize("Y,X", [(32, 16), (64, 32)])
def test_small_grid_direct_solver_shapes_against_oracle(Y, X):
    """fd_solve_small on the small grids of the reference domain (Y = 2 X; it accepts Y % 16 == 0, X in {16, 32, 64}, at most
    2048 cells): step and input gradients against the exactly solved float64 oracle, zero iterations."""
    B = 2
    g, mk = masks_for(Y, X)
    assert mk.direct is not None
    d, vy, vx = o.synthetic_state(B, Y, X, 77)
    re = torch.tensor(o.RE_TRAIN[:B], dtype=torch.float64)
    vy = vy.clone().requires_grad_(True)
    vx = vx.clone().requires_grad_(True)
    d2, py, px = o.karman_step(d, vy, vx, re, g)
    gen = torch.Generator().manual_seed(3)
    wy, wx = torch.randn(py.shape, generator=gen, dtype=torch.float64), torch.randn(px.shape, generator=gen, dtype=torch.float64)
    ((py * wy).sum() + (px * wx).sum()).backward()
    cfg = ops.karman_cfg(B, Y, X, g.dx, masks=mk)
    hy, hx = f32(vy.detach()).requires_grad_(True), f32(vx.detach()).requires_grad_(True)
    info = {}
    e2, qy, qx = ops.karman_step(f32(d), hy, hx, f32(re), cfg, mk, info)
    ((qy * f32(wy)).sum() + (qx * f32(wx)).sum()).backward()
    assert int(info["iterations"].max()) == 0 and int(info["iterations_bwd"].max()) == 0
    assert rel(e2, d2) < TOL_FIELD and rel(qy, py) < TOL_FIELD and rel(qx, px) < TOL_FIELD
    assert rel(hy.grad, vy.grad) < TOL_GRAD and rel(hx.grad, vx.grad) < TOL_GRAD


@pytest.mark.parametrize("B,solver", [(1, "direct"), (2, "direct"), (1, "pcg"), (2, "pcg"), (2, "cg")])
def test_karman_step_full_size_against_oracle(B, solver):
    """The three PressureSolver implementations (direct = fast diagonalisation + capacitance correction,
    two-level preconditioned CG, plain CG) against the exactly solved oracle."""
    Y, X = 128, 64
    precond = solver != "cg"
    g, mk = masks_for(Y, X, precond, "direct" if solver == "direct" else "cg")
    assert (mk.direct is not None) == (solver == "direct")
    d, vy, vx = o.synthetic_state(B, Y, X, 1234)
    re = torch.tensor(o.RE_TRAIN[:B], dtype=torch.float64)
    vy = vy.clone().requires_grad_(True)
    vx = vx.clone().requires_grad_(True)
    d2, py, px = o.karman_step(d, vy, vx, re, g)
    gen = torch.Generator().manual_seed(5)
    wy = torch.randn(py.shape, generator=gen, dtype=torch.float64)
    wx = torch.randn(px.shape, generator=gen, dtype=torch.float64)
    ((py * wy).sum() + (px * wx).sum()).backward()
    hvy = f32(vy.detach()).requires_grad_(True)
    hvx = f32(vx.detach()).requires_grad_(True)
    info = {}
    hd, hpy, hpx = ops.karman_step(f32(d), hvy, hvx, f32(re), ops.karman_cfg(B, Y, X, g.dx, masks=mk), mk, info)
    ((hpy * f32(wy)).sum() + (hpx * f32(wx)).sum()).backward()
    assert rel(hd, d2) < TOL_FIELD and rel(hpy, py) < TOL_FIELD and rel(hpx, px) < TOL_FIELD
    assert rel(hvy.grad, vy.grad) < TOL_GRAD and rel(hvx.grad, vx.grad) < TOL_GRAD
    its = int(info["iterations"].max())
    if solver == "direct":
        assert its == 0                                       # no iteration at all
    else:
        assert (its < 100) if precond else (150 < its < 400)  # the coarse space cuts the iteration count ~4x


def test_full_size_properties():
    """BASELINE.json's full per-GPU batch (6 x 128x64): size independent properties."""
    B, Y, X = 6, 128, 64
    g, mk = masks_for(Y, X)
    d, vy, vx = (f32(t) for t in sol_amd.synthetic.state(B, Y, X, 1))
    re = f32(sol_amd.synthetic.reynolds(B))
    cfg = ops.karman_cfg(B, Y, X, g.dx, masks=mk)
    d2, py, px = ops.karman_step(d, vy, vx, re, cfg, mk)
    # (a) divergence free on interior fluid cells (boundary cells keep PhiFlow's replicate-pad quirk)
    div = (py[:, 1:, :] - py[:, :-1, :]) + (px[:, :, 1:] - px[:, :, :-1])
    act = f32(g.active)
    scale = float(py.abs().max())
    assert float((div * act)[:, 1:-1, 1:-1].abs().max()) < 2e-5 * scale
    # (b) faces touching the obstacle are closed
    assert float((py * (1 - f32(g.my))).abs().max()) == 0.0 and float((px * (1 - f32(g.mx))).abs().max()) == 0.0
    # (c) passive tracer: zero density only grows where the inflow is (density >= 0, += dt inside the box)
    z, _, _ = ops.karman_step(torch.zeros_like(d), vy, vx, re, cfg, mk)
    assert torch.equal(z, f32(g.inflow).expand(B, -1, -1))
    # (d) deterministic forward
    d3, py3, px3 = ops.karman_step(d, vy, vx, re, cfg, mk)
    assert torch.equal(py, py3) and torch.equal(px, px3) and torch.equal(d2, d3)
    # (e) projection is idempotent up to solver tolerance: a second step from a quiescent
    #     divergence-free uniform flow changes nothing but the BC rows
    # (f) adjoint consistency <J u, w> == <u, J^T w> by central differences
    vy = vy.clone().requires_grad_(True)
    vx = vx.clone().requires_grad_(True)
    _, qy, qx = ops.karman_step(d, vy, vx, re, cfg, mk)
    gen = torch.Generator().manual_seed(3)
    wy, wx = f32(torch.randn(qy.shape, generator=gen)), f32(torch.randn(qx.shape, generator=gen))
    ((qy * wy).sum() + (qx * wx).sum()).backward()
    uy = f32(sol_amd.synthetic._smooth(torch.randn(vy.shape, generator=gen, dtype=torch.float64)))
    ux = f32(sol_amd.synthetic._smooth(torch.randn(vx.shape, generator=gen, dtype=torch.float64)))
    eps = 1e-2
    with torch.no_grad():
        _, ay, ax = ops.karman_step(d, vy + eps * uy, vx + eps * ux, re, cfg, mk)
        _, by, bx = ops.karman_step(d, vy - eps * uy, vx - eps * ux, re, cfg, mk)
    lhs = float((((ay - by) * wy).sum() + ((ax - bx) * wx).sum()).double() / (2 * eps))
    rhs = float(((vy.grad * uy).sum() + (vx.grad * ux).sum()).double())
    assert abs(lhs - rhs) < 2e-2 * max(abs(lhs), abs(rhs))


def test_unsupported_shapes_fail_loudly():
    z = torch.zeros(1, 16, 12, device=DEV)
    with pytest.raises(sol_amd.SolError):
        ops.karman_step(z, torch.zeros(1, 17, 12, device=DEV), torch.zeros(1, 16, 13, device=DEV),
                        torch.ones(1, device=DEV), ops.karman_cfg(1, 16, 12, 1.0), masks_for(16, 8)[1])


# ---------------------------------------------------------------------------------------------
# conv 5x5 (fp32 MFMA) forward / backward
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,cin,cout,lrelu,res", [
    (2, 16, 8, 3, 32, True, False), (2, 16, 8, 32, 32, True, True), (2, 16, 8, 32, 2, False, False),
    (1, 64, 32, 32, 32, True, True), (2, 128, 64, 32, 32, True, False), (1, 128, 64, 3, 32, True, False),
    (1, 128, 64, 32, 2, False, False)])
def test_conv5x5_against_oracle(B, H, W, cin, cout, lrelu, res):
    gen = torch.Generator().manual_seed(0)
    dt = torch.float64
    x = torch.randn(B, H, W, cin, generator=gen, dtype=dt, requires_grad=True)
    w = (torch.randn(5, 5, cin, cout, generator=gen, dtype=dt) * 0.05).requires_grad_(True)
    b = (torch.randn(cout, generator=gen, dtype=dt) * 0.1).requires_grad_(True)
    r = torch.randn(B, H, W, cout, generator=gen, dtype=dt, requires_grad=True) if res else None
    y = o._conv(x, w, b)
    if res:
        y = y + r
    if lrelu:
        y = torch.nn.functional.leaky_relu(y, 0.3)
    gy = torch.randn(y.shape, generator=gen, dtype=dt)
    (y * gy).sum().backward()
    hx, hw, hb = (f32(t.detach()).requires_grad_(True) for t in (x, w, b))
    hr = f32(r.detach()).requires_grad_(True) if res else None
    hy = ops.conv5x5(hx, hw, hb, hr, lrelu, 0.3)
    (hy * f32(gy)).sum().backward()
    assert rel(hy, y) < 2e-6
    assert rel(hx.grad, x.grad) < 2e-6 and rel(hw.grad, w.grad) < 2e-6 and rel(hb.grad, b.grad) < 2e-6
    if res:
        assert rel(hr.grad, r.grad) < 2e-6


# ---------------------------------------------------------------------------------------------
# fused training step, Adam, roll-out
# ---------------------------------------------------------------------------------------------
def _trainer_from(params, g, B, Y, X, ms, std_v, **kw):
    mk = ops.SceneMasks(g.active, g.inflow, g.bc_mask, g.bc_mask)
    net = sol_amd.model_mars_moon(cin=3, cout=2, seed=0)
    net.set_weights([p.detach().numpy() for p in params])
    return net, sol_amd.SolTrainer(net, mk, B, Y, X, ms, g.dx, std_v, o.STD_RE, **kw)


def test_train_step_against_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "train_16x8_sol2.npz"))
    B, Y, X = z["d"].shape
    ms = z["gt_vy"].shape[0]
    params = golden_train_params(z)
    net, tr = _trainer_from(params, o.geometry(Y, X), B, Y, X, ms, tuple(z["std_v"]))
    loss = tr.fwd_bwd(f32(z["d"]), f32(z["vy"]), f32(z["vx"]), f32(z["re"]), f32(z["gt_vy"]), f32(z["gt_vx"]), want_final=True)
    assert abs(float(loss) - float(z["loss"])) < 1e-5 * abs(float(z["loss"]))
    assert np.allclose(tr.loss_steps.cpu().numpy(), z["loss_steps"], rtol=1e-5)
    assert rel(tr.grads[::16], z["grads_sub16"]) < TOL_GRAD
    norms = np.array([float(tr.grads[net.offsets[k]:net.offsets[k + 1]].double().norm()) for k in range(24)])
    assert np.allclose(norms, z["grad_norms"], rtol=1e-4)
    assert rel(tr.final[1], z["vy_final"]) < TOL_FIELD and rel(tr.final[2], z["vx_final"]) < TOL_FIELD
    assert rel(tr.final[0], z["d_final"]) < TOL_FIELD


def _oracle_problem(B, Y, X, ms, clip=None):
    g = o.geometry(Y, X)
    d, vy, vx = o.synthetic_state(B, Y, X, 1234)
    re = torch.tensor([o.RE_TRAIN[i % 6] for i in range(B)], dtype=torch.float64)
    gts = [o.synthetic_state(B, Y, X, 4321 + i, project_it=False) for i in range(ms)]
    params = [p.clone().requires_grad_(True) for p in o.init_params(0)]
    std_v = (0.2, 0.25)
    loss = o.unrolled_loss(params, d, vy, vx, re, [s[1] for s in gts], [s[2] for s in gts], g, std_v, o.STD_RE)
    loss.backward()
    return g, d, vy, vx, re, gts, params, std_v, loss


@pytest.mark.parametrize("clip", [False, True])
def test_train_step_c2_config_and_adam(clip):
    """BASELINE configs[1]: karman-2d 64x32, msteps=4, batch=3 -- fwd/bwd + TF-Adam (+ clip_by_norm)."""
    B, Y, X, ms = 3, 64, 32, 4
    g, d, vy, vx, re, gts, params, std_v, loss = _oracle_problem(B, Y, X, ms)
    net, tr = _trainer_from(params, g, B, Y, X, ms, std_v, clip_grad=clip)
    hl = tr.fwd_bwd(f32(d), f32(vy), f32(vx), f32(re), f32(torch.stack([s[1] for s in gts])), f32(torch.stack([s[2] for s in gts])))
    gref = torch.cat([p.grad.reshape(-1) for p in params])
    assert abs(float(hl) - float(loss)) < 1e-5 * abs(float(loss))
    assert rel(tr.grads, gref) < TOL_GRAD
    m0 = [torch.zeros_like(p) for p in params]
    v0 = [torch.zeros_like(p) for p in params]
    p_ref = [p.detach() for p in params]
    for t in (1, 2):
        p_ref, m0, v0 = o.adam_tf(p_ref, [p.grad for p in params], m0, v0, t, 1e-4, clip_norm=1e-3 if clip else None)
        tr.grads.copy_(f32(gref))         # same (oracle) gradient both steps: isolates the optimizer
        tr.apply_gradients(1e-4)
    assert rel(net.params, torch.cat([p.reshape(-1) for p in p_ref])) < 1e-6
    # the update itself (not just the weights) must match
    upd_ref = torch.cat([(a - b.detach()).reshape(-1) for a, b in zip(p_ref, params)])
    upd = net.params.detach().double().cpu() - torch.cat([p.detach().reshape(-1) for p in params])
    assert rel(upd, upd_ref) < 1e-3


@pytest.mark.parametrize("B", [3, 12])
def test_transposed_cnn_recipe_paths_against_oracle(B):
    """64x32 (the reference's own recipe: the CNN runs on the transposed images) through everything round 4 folded into its launches:
    features / feature gradients in the CNN's cell order straight from the solver kernels, the correction + loss epilogue in transposed
    cell order (B = 3: the one-row thin form of the dx kernel; B = 12: a chip-filling launch, k_conv5x5_sb<1, 2>), weight gradients and
    the passive density riding in the solver-adjoint launches (k_karman_bwd_bww_small) -- loss, per-step losses, gradient and the final
    state (density included) against the float64 oracle, and the kernel set that ran."""
    from sol_amd import _lib
    Y, X, ms = 64, 32, 2
    g, d, vy, vx, re, gts, params, std_v, loss = _oracle_problem(B, Y, X, ms)
    net, tr = _trainer_from(params, g, B, Y, X, ms, std_v)
    args = (f32(d), f32(vy), f32(vx), f32(re), f32(torch.stack([s[1] for s in gts])), f32(torch.stack([s[2] for s in gts])))
    hl = tr.fwd_bwd(*args, want_final=True)
    gref = torch.cat([p.grad.reshape(-1) for p in params])
    assert abs(float(hl) - float(loss)) < 1e-5 * abs(float(loss))
    assert rel(tr.grads, gref) < TOL_GRAD
    with torch.no_grad():
        rd, ry, rx = d, vy, vx
        for _ in range(ms):
            rd, ry, rx = o.karman_step(rd, ry, rx, re, g)
            cy, cx = o.correction([p.detach() for p in params], ry, rx, re, std_v, o.STD_RE)
            ry, rx = ry + cy, rx + cx
    assert rel(tr.final[1], ry) < TOL_FIELD and rel(tr.final[2], rx) < TOL_FIELD and rel(tr.final[0], rd) < TOL_FIELD
    with _lib.profile() as p:
        tr.fwd_bwd(*args, want_final=True, eager=True)
    names = {k.strip("()"): v[0] for k, v in p.kernels.items()}
    assert names.get("k_karman_bwd_bww_small") == ms and "k_density_chain" not in names and "k_correct_loss" not in names, names
    assert not any(k.startswith("k_transpose_cells") for k in names), names
    thin_fwd = "k_conv5x5_thin32<2, 16>"                            # (round 5: the exact-fp32 VALU kernel, every batch size; k_conv5x5_dx<1, 1> / k_conv5x5_sb<1, 2> before)
    assert names.get(thin_fwd) == ms + (ms - 1), names              # correction-mode output layer + the 32 -> 3 data gradient


_ORACLE_CACHE = {}


def _cached(key, fn):
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = fn()
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize("precision", ["split", "bf16x6", "fp32"])
def test_train_step_full_size_against_oracle(precision):
    """BASELINE configs[2] grid (128x64) through the fused trainer, in each of the three convolution arithmetics: direct pressure solver, fp16/bf16 split-MFMA convolutions
    scaled by the absmax the producer kernels publish, weight gradients batched over the unrolled steps, density chain --
    loss, per-step losses, the full gradient and the final state against the float64 oracle (B=2, msteps=2)."""
    B, Y, X, ms = 2, 128, 64, 2
    g, d, vy, vx, re, gts, params, std_v, loss = _cached(("full", B, Y, X, ms), lambda: _oracle_problem(B, Y, X, ms))
    net, tr = _trainer_from(params, g, B, Y, X, ms, std_v, conv_precision=precision)
    assert tr.masks.direct is not None
    hl = tr.fwd_bwd(f32(d), f32(vy), f32(vx), f32(re), f32(torch.stack([s[1] for s in gts])), f32(torch.stack([s[2] for s in gts])),
                    want_final=True)
    gref = torch.cat([p.grad.reshape(-1) for p in params])
    assert abs(float(hl) - float(loss)) < 1e-5 * abs(float(loss))
    assert rel(tr.grads, gref) < TOL_GRAD
    off = net.offsets
    per_tensor = [rel(tr.grads[off[k]:off[k + 1]], params[k].grad.reshape(-1)) for k in range(len(params))]
    assert max(per_tensor) < 3 * TOL_GRAD, per_tensor
    assert int(tr.iters_fwd.max()) == 0 and int(tr.iters_bwd.max()) == 0          # direct solver: no CG iterations
    # final state: roll the oracle forward with the same weights
    with torch.no_grad():
        rd, ry, rx = d, vy, vx
        for _ in range(ms):
            rd, ry, rx = o.karman_step(rd, ry, rx, re, g)
            cy, cx = o.correction([p.detach() for p in params], ry, rx, re, std_v, o.STD_RE)
            ry, rx = ry + cy, rx + cx
    assert rel(tr.final[1], ry) < TOL_FIELD and rel(tr.final[2], rx) < TOL_FIELD and rel(tr.final[0], rd) < TOL_FIELD


@pytest.fixture
def conv_dx(request):
    """option conv_dx for the duration of a test: 1 = dx-major kernel (k_conv5x5_dx, the default), 0 = k_conv5x5_sb<2, 2>"""
    from sol_amd import _lib
    saved = _lib.get_option("conv_dx")
    _lib.set_option("conv_dx", request.param)
    yield request.param
    _lib.set_option("conv_dx", saved)


@pytest.mark.parametrize("conv_dx", [1, 0], indirect=True)
def test_conv5x5_scaled_fp16_path_against_float64(conv_dx):
    """sol_conv5x5_scaled: three fp16 MFMA products with the power-of-two scale from the absmax slots, for well and
    badly conditioned dynamic ranges; also checks the absmax the kernel publishes for its own output.  Both 32 -> 32 kernels."""
    import ctypes as C
    from sol_amd._lib import ptr, stream, check
    lib = sol_amd.load()
    gen = torch.Generator().manual_seed(11)
    for scale, heavy in [(1.0, False), (1e-6, False), (3e4, False), (1.0, True)]:
        B, Y, X, cout = 2, 32, 64, 32
        x = torch.randn(B, Y, X, 32, generator=gen)
        if heavy:
            x = x * torch.exp(2.0 * torch.randn(B, Y, X, 32, generator=gen))
        x = (x * scale).float().to(DEV)
        w = (torch.randn(5, 5, 32, cout, generator=gen) * 0.05).float().to(DEV)
        bias = torch.zeros(cout, dtype=torch.float32, device=DEV)
        packed = ops._pack(w, 32, cout, ops.CONV_FWD)
        xmax = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)
        xmax[5] = int(x.abs().max().view(torch.int32).item())              # any slot: the consumer takes the max over all 64
        ymax = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)
        y = torch.empty(B, Y, X, cout, dtype=torch.float32, device=DEV)
        check(lib.sol_conv5x5_scaled(stream(), ptr(x), ptr(packed), ptr(bias), None, None, ptr(y), B, Y, X, 32, cout,
                                     ops.EPI_LRELU, 0.3, ptr(xmax), ptr(ymax)))
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, padding=2).permute(0, 2, 3, 1)
        ref = torch.nn.functional.leaky_relu(ref, 0.3)
        assert rel(y, ref) < 1e-6, (scale, heavy, rel(y, ref))
        if not os.environ.get("SOL_CONV_NO_SB"):        # the fp32 fallback kernels neither consume nor publish the absmax
            assert float(ymax.max().view(torch.float32).item()) == float(y.abs().max())
    # degenerate tensor: all-zero input (absmax 0 -> scale of 1) gives exactly act(bias)
    B, Y, X, cout = 1, 16, 64, 32
    x = torch.zeros(B, Y, X, 32, dtype=torch.float32, device=DEV)
    w = (torch.randn(5, 5, 32, cout, generator=gen) * 0.05).float().to(DEV)
    bias = torch.linspace(-1, 1, cout, dtype=torch.float32).to(DEV)
    y = torch.empty(B, Y, X, cout, dtype=torch.float32, device=DEV)
    check(lib.sol_conv5x5_scaled(stream(), ptr(x), ptr(ops._pack(w, 32, cout, ops.CONV_FWD)), ptr(bias), None, None, ptr(y), B, Y, X, 32, cout,
                                 ops.EPI_LRELU, 0.3, ptr(torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)), None))
    assert torch.equal(y, torch.nn.functional.leaky_relu(bias, 0.3).expand(B, Y, X, cout))


@pytest.mark.parametrize("B,H,W", [(6, 128, 64), (3, 32, 64), (2, 5, 64), (1, 7, 128), (4, 3, 64), (1, 1, 64), (5, 2, 64), (3, 4, 64)])
def test_conv5x5_dx_kernel_every_epilogue_against_float64_and_the_row_per_wave_kernel(B, H, W):
    """The dx-major 32 -> 32 kernel (csrc/conv5x5_dx.hip: a wave owns a pixel segment of ALL output rows of its workgroup, input
    rows shared across tap rows) against a float64 convolution and against k_conv5x5_sb<2, 2> for every epilogue form: workgroups
    that straddle two images (H % 3 != 0), images shorter than the tap window, one-row images, two column blocks (W = 128), the
    transposed 64x32 recipe's shape, and the small-launch form (one output row per workgroup when the launch has few rows).  Same
    operand splits and products as the other kernel, another summation order: equal to it to fp32 round-off, not bit for bit."""
    from sol_amd import _lib
    gen = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, H, W, 32, generator=gen, dtype=torch.float32).to(DEV)
    w = (torch.randn(5, 5, 32, 32, generator=gen, dtype=torch.float32) * 0.05).to(DEV)
    b = torch.randn(32, generator=gen, dtype=torch.float32).to(DEV)
    res = torch.randn(B, H, W, 32, generator=gen, dtype=torch.float32).to(DEV)
    act = torch.randn(B, H, W, 32, generator=gen, dtype=torch.float32).to(DEV)
    packed = ops._pack(w, 32, 32, ops.CONV_FWD)
    xm = ops.absmax_slots(x)
    conv = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, padding=2).permute(0, 2, 3, 1)
    saved = _lib.get_option("conv_dx")
    try:
        for name, (bb, rr, aa, epi) in {"bias+lrelu": (b, None, None, ops.EPI_LRELU), "res+lrelu": (b, res, None, ops.EPI_LRELU),
                                        "res+dlrelu": (None, res, act, ops.EPI_DLRELU), "plain": (None, None, None, ops.EPI_NONE)}.items():
            ref = conv + (bb.double() if bb is not None else 0.0) + (rr.double() if rr is not None else 0.0)
            if epi == ops.EPI_LRELU:
                ref = torch.where(ref > 0, ref, 0.3 * ref)
            elif epi == ops.EPI_DLRELU:
                ref = ref * torch.where(aa.double() > 0, 1.0, 0.3)
            ys = {}
            for dx in (1, 0):
                _lib.set_option("conv_dx", dx)
                ym = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)
                ys[dx] = ops.conv5x5_scaled_raw(x, packed, bb, rr, aa, 32, epi, 0.3, xm, ym)
                assert float(ym.max().view(torch.float32).item()) == float(ys[dx].abs().max()), (name, dx)
            assert rel(ys[1], ref) < 6e-7 and rel(ys[1], ref) < 1.5 * rel(ys[0], ref) + 1e-8, (name, rel(ys[1], ref), rel(ys[0], ref))
            assert rel(ys[1], ys[0]) < 6e-7
            _lib.set_option("conv_dx", 1)
            with _lib.profile() as p:
                ops.conv5x5_scaled_raw(x, packed, bb, rr, aa, 32, epi, 0.3, xm, None)
            assert any(k.strip("()").startswith("k_conv5x5_dx") for k in p.kernels), p.kernels       # the dx kernel did run
    finally:
        _lib.set_option("conv_dx", saved)


@pytest.mark.parametrize("B,H,W", [(6, 128, 64), (2, 5, 64), (1, 7, 128), (1, 1, 64)])
@pytest.mark.parametrize("cout", [2, 3, 16])
def test_conv5x5_dx_thin_layers_against_float64_and_the_row_per_wave_kernel(B, H, W, cout):
    """The thin-layer form of the dx-major kernel (k_conv5x5_dx<R, 1>: <= 16 output channels, waves 4..7 stage only; option conv_dx
    bit 1 = in one-row-per-workgroup launches (default), bit 2 = everywhere) against a float64 convolution and k_conv5x5_sb<1, 2>
    for the strided-store epilogues (the correction-mode epilogue is covered by the trainer and roll-out tests at B = 1)."""
    from sol_amd import _lib
    gen = torch.Generator().manual_seed(B * 100 + H + cout)
    x = torch.randn(B, H, W, 32, generator=gen, dtype=torch.float32).to(DEV)
    w = (torch.randn(5, 5, 32, cout, generator=gen, dtype=torch.float32) * 0.05).to(DEV)
    b = torch.randn(cout, generator=gen, dtype=torch.float32).to(DEV)
    res = torch.randn(B, H, W, cout, generator=gen, dtype=torch.float32).to(DEV)
    act = torch.randn(B, H, W, cout, generator=gen, dtype=torch.float32).to(DEV)
    packed = ops._pack(w, 32, cout, ops.CONV_FWD)
    xm = ops.absmax_slots(x)
    conv = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, padding=2).permute(0, 2, 3, 1)
    saved, saved_valu = _lib.get_option("conv_dx"), _lib.get_option("conv_thin_valu")
    _lib.set_option("conv_thin_valu", 0)            # (the exact-fp32 VALU form of the plain thin layers has its own test below)
    try:
        for name, (bb, rr, aa, epi) in {"bias+lrelu": (b, None, None, ops.EPI_LRELU), "res+dlrelu": (None, res, act, ops.EPI_DLRELU),
                                        "plain": (None, None, None, ops.EPI_NONE)}.items():
            ref = conv + (bb.double() if bb is not None else 0.0) + (rr.double() if rr is not None else 0.0)
            if epi == ops.EPI_LRELU:
                ref = torch.where(ref > 0, ref, 0.3 * ref)
            elif epi == ops.EPI_DLRELU:
                ref = ref * torch.where(aa.double() > 0, 1.0, 0.3)
            ys = {}
            for dx in (7, 1):
                _lib.set_option("conv_dx", dx)
                ym = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)
                with _lib.profile() as p:
                    ys[dx] = ops.conv5x5_scaled_raw(x, packed, bb, rr, aa, cout, epi, 0.3, xm, ym)
                assert float(ym.max().view(torch.float32).item()) == float(ys[dx].abs().max()), (name, dx)
                assert any(("k_conv5x5_dx" in k) == (dx == 7) for k in p.kernels), (dx, p.kernels)
            assert rel(ys[7], ref) < 6e-7 and rel(ys[7], ref) < 1.5 * rel(ys[1], ref) + 1e-8, (name, rel(ys[7], ref), rel(ys[1], ref))
    finally:
        _lib.set_option("conv_dx", saved)
        _lib.set_option("conv_thin_valu", saved_valu)


@pytest.mark.parametrize("B,H", [(6, 128), (2, 5), (1, 1), (3, 64), (1, 7)])
@pytest.mark.parametrize("cout", [1, 2, 3, 4])
def test_conv5x5_thin_valu_kernel_against_float64_and_the_split_kernel(B, H, cout):
    """k_conv5x5_thin32 (conv5x5_thin.hip, option conv_thin_valu, default on): the thin 32 -> (<= 4) layers of 64-pixel images in EXACT
    fp32 on the vector ALU -- no absmax dependency, no operand split.  Against a float64 convolution (with and without bias; image
    heights that are not multiples of the three rows of a workgroup, single rows, images straddling workgroups) and against the
    split-precision kernel it replaces; the published absmax equals the output's.  Launches it does not take (a residual, an
    activation, 128-pixel rows, 16 output channels) must keep their kernels.  The correction-mode epilogue (velocity update + l2 loss)
    runs in every trainer / roll-out test at 128x64 and in test_sol32_bench_workload_against_golden."""
    from sol_amd import _lib
    W = 64
    gen = torch.Generator().manual_seed(B * 1000 + H * 10 + cout)
    x = torch.randn(B, H, W, 32, generator=gen, dtype=torch.float32).to(DEV)
    w = (torch.randn(5, 5, 32, cout, generator=gen, dtype=torch.float32) * 0.05).to(DEV)
    b = torch.randn(cout, generator=gen, dtype=torch.float32).to(DEV)
    packed = ops._pack(w, 32, cout, ops.CONV_FWD)
    xm = ops.absmax_slots(x)
    conv = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, padding=2).permute(0, 2, 3, 1)
    saved = _lib.get_option("conv_thin_valu")
    try:
        for bb in (None, b):
            ref = conv + (bb.double() if bb is not None else 0.0)
            ys = {}
            for valu in (1, 0):
                _lib.set_option("conv_thin_valu", valu)
                ym = torch.zeros(ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)
                with _lib.profile() as p:
                    ys[valu] = ops.conv5x5_scaled_raw(x, packed, bb, None, None, cout, ops.EPI_NONE, 0.3, xm, ym)
                assert float(ym.max().view(torch.float32).item()) == float(ys[valu].abs().max())
                assert any("k_conv5x5_thin32" in k for k in p.kernels) == (valu == 1), (valu, p.kernels)
            assert rel(ys[1], ref) < 4e-7, rel(ys[1], ref)
            assert rel(ys[1], ref) <= 1.5 * rel(ys[0], ref) + 1e-8, (rel(ys[1], ref), rel(ys[0], ref))
        _lib.set_option("conv_thin_valu", 1)
        res = torch.randn(B, H, W, cout, generator=gen, dtype=torch.float32).to(DEV)
        for kw in (dict(residual=res, epi=ops.EPI_NONE), dict(residual=None, epi=ops.EPI_LRELU)):
            with _lib.profile() as p:
                ops.conv5x5_scaled_raw(x, packed, None, kw["residual"], None, cout, kw["epi"], 0.3, xm, None)
            assert not any("k_conv5x5_thin32" in k for k in p.kernels), p.kernels
    finally:
        _lib.set_option("conv_thin_valu", saved)


def test_mars_moon_network_full_size_against_torch_float64_autograd():
    """model_mars_moon at the bench shape (B = 6, 128 x 64): forward, input gradient and all 24 parameter gradients of the HIP
    convolution path (per-op autograd surface: the same forward / backward-data / weight-gradient kernels the fused trainer
    schedules) against plain PyTorch float64 F.conv2d + autograd on the device -- an implementation that shares nothing with
    oracle/ or the product.  The reference applies LeakyReLU with the HIP forward's sign masks: ONE pre-activation that rounds
    across zero changes a layer's gradient by 0.7 / sqrt(1.6 M) = 6e-4 relative (see the 3-D twin of this test).  Parameter-gradient
    bound: sums of 49 152 zero-mean products per weight (condition number ~ sqrt(N) = 220) in fp32: measured 3e-6 ... 1.2e-5 on
    kernels, 2-3e-5 on biases (a plain fp32 sum of dz)."""
    import torch.nn.functional as F
    from sol_amd import ops
    B, Y, X = 6, 128, 64
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(B, Y, X, 3, generator=gen, dtype=torch.float32).to(DEV)
    gy = (torch.randn(B, Y, X, 2, generator=gen, dtype=torch.float32) * 1e-3).to(DEV)
    net = sol_amd.model_mars_moon(cin=3, cout=2, seed=5, device=DEV)
    p, sl = net.tensors(), net.slope
    xi = x.clone().requires_grad_(True)
    acts = [ops.conv5x5(xi, p[0], p[1], None, True, sl)]
    for k in range(5):
        a = ops.conv5x5(acts[-1], p[2 + 4 * k], p[3 + 4 * k], None, True, sl)
        acts += [a, ops.conv5x5(a, p[4 + 4 * k], p[5 + 4 * k], acts[-1], True, sl)]
    out = ops.conv5x5(acts[-1], p[22], p[23], None, False, sl)
    (out * gy).sum().backward()
    masks = [(t.detach() > 0).permute(0, 3, 1, 2) for t in acts]
    torch.cuda.synchronize()
    tp = [t.detach().double().clone().requires_grad_(True) for t in p]
    conv = lambda t, k: F.conv2d(t, tp[2 * k].permute(3, 2, 0, 1), tp[2 * k + 1], padding=2)
    lrelu = lambda z, m: z * torch.where(m, 1.0, sl).to(z.dtype)
    xt = x.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    h = lrelu(conv(xt, 0), masks[0])
    for k in range(5):
        a = lrelu(conv(h, 1 + 2 * k), masks[1 + 2 * k])
        h = lrelu(conv(a, 2 + 2 * k) + h, masks[2 + 2 * k])
    ref = conv(h, 11)
    (ref * gy.double().permute(0, 3, 1, 2)).sum().backward()
    torch.cuda.synchronize()
    off = net.offsets
    per = [rel(net.params.grad[off[k]:off[k + 1]], tp[k].grad.reshape(-1)) for k in range(24)]
    e_out, e_x = rel(out.detach(), ref.detach().permute(0, 2, 3, 1)), rel(xi.grad, xt.grad.permute(0, 2, 3, 1))
    print("torch float64 reference (2-D): out %.2e, dx %.2e, params max %.2e" % (e_out, e_x, max(per)))
    assert e_out < 5e-6 and e_x < 1e-5 and max(per[0::2]) < 3e-5 and max(per[1::2]) < 1e-4, (e_out, e_x, per)


def test_karman_step_adjoint_identity_by_finite_differences():
    """<J u, w> = <u, J^T w> for the fused step at 128 x 64 with the HIP forward on both sides: J u from central differences of
    the forward kernel, J^T w from the adjoint kernel.  No oracle involved -- the adjoint must be the adjoint of THIS forward
    (semi-Lagrangian gathers, diffusion, projection).  fp32 differences limit the agreement to ~1e-3."""
    from sol_amd import ops, synthetic
    B, Y, X = 2, 128, 64
    dom = sol_amd.Domain([Y, X], box=sol_amd.box[0:200, 0:100])
    active, inflow = sol_amd.KarmanFlow().scene_arrays(dom)
    bcv, bcm = sol_amd.velocity_bc_masks(Y, X)
    masks = ops.SceneMasks(active, inflow, bcv.reshape(Y + 1, X), bcm.reshape(Y + 1, X), DEV)
    cfg = ops.karman_cfg(B, Y, X, dom.dx[1], masks=masks)
    d0, vy, vx = (f32(t) for t in synthetic.state(B, Y, X, 99))
    re = f32(synthetic.reynolds(B))
    with torch.no_grad():
        for _ in range(2):                            # spin-up: smooth, divergence free, consistent with the BCs
            d0, vy, vx = ops.karman_step(d0, vy, vx, re, cfg, masks)
    gen = torch.Generator().manual_seed(17)
    smooth = lambda t: torch.nn.functional.avg_pool2d(t[:, None], 5, 1, 2)[:, 0]
    uy, ux = f32(smooth(torch.randn(vy.shape, generator=gen))), f32(smooth(torch.randn(vx.shape, generator=gen)))
    wy, wx = f32(torch.randn(vy.shape, generator=gen)), f32(torch.randn(vx.shape, generator=gen))
    ay, ax = vy.clone().requires_grad_(True), vx.clone().requires_grad_(True)
    _, py, px = ops.karman_step(d0, ay, ax, re, cfg, masks)
    ((py * wy).sum() + (px * wx).sum()).backward()
    rhs = float((ay.grad.double() * uy.double()).sum() + (ax.grad.double() * ux.double()).sum())
    res = {}
    for eps in (2e-2, 1e-2, 5e-3):
        with torch.no_grad():
            _, p1y, p1x = ops.karman_step(d0, vy + eps * uy, vx + eps * ux, re, cfg, masks)
            _, m1y, m1x = ops.karman_step(d0, vy - eps * uy, vx - eps * ux, re, cfg, masks)
        lhs = float((((p1y.double() - m1y.double()) * wy.double()).sum() + ((p1x.double() - m1x.double()) * wx.double()).sum()) / (2 * eps))
        res[eps] = lhs
    torch.cuda.synchronize()
    scale = float((ay.grad.double().norm() ** 2 + ax.grad.double().norm() ** 2).sqrt() * (uy.double().norm() ** 2 + ux.double().norm() ** 2).sqrt())
    print("adjoint identity: <u, J^T w> = %.6e, <J u, w> by central differences %s, |u||J^T w| = %.3e" % (rhs, res, scale))
    assert min(abs(v - rhs) for v in res.values()) < 2e-3 * abs(rhs) + 2e-4 * scale, (rhs, res, scale)


def test_training_step_gradient_by_finite_differences():
    """d loss / d weights of the fused SOL-4 training step (64 x 32, B = 3: forward unroll, reverse sweep through solver adjoints and
    network) along a random direction in parameter space against central differences of the SAME engine's loss -- no oracle.
    The direction is scaled per tensor to the weights' magnitude; fp32 losses limit the agreement to ~1e-3."""
    import bench
    dev = torch.device(DEV)
    wl = bench.Workload(sol_amd, dev, 3, 64, 32, 4, 0, use_graph=False)      # the bench workload's construction at the reference's recipe size
    tr, net = wl.trainer, wl.net
    args = (wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx)
    loss = float(tr.fwd_bwd(*args))
    g = tr.grads.detach().double().clone()
    gen = torch.Generator().manual_seed(3)
    u = torch.zeros(net.n_params, dtype=torch.float64)
    for k in range(len(net.shapes)):
        sl = slice(int(net.offsets[k]), int(net.offsets[k + 1]))
        wk = net.params.detach()[sl].double().cpu()
        u[sl] = torch.randn(wk.numel(), generator=gen, dtype=torch.float64) * (float(wk.abs().mean()) + 1e-3)
    u = u.to(dev)
    rhs = float((g * u).sum())
    p0 = net.params.detach().clone()
    res = {}
    for eps in (3e-2, 1e-2, 3e-3):
        with torch.no_grad():
            net.params.copy_((p0.double() + eps * u).float())
        lp = float(tr.fwd_bwd(*args))
        with torch.no_grad():
            net.params.copy_((p0.double() - eps * u).float())
        lm = float(tr.fwd_bwd(*args))
        res[eps] = (lp - lm) / (2 * eps)
    with torch.no_grad():
        net.params.copy_(p0)
    print("training-step gradient: loss %.6e, <grad, u> = %.6e, central differences %s" % (loss, rhs, res))
    assert min(abs(v - rhs) for v in res.values()) < 3e-3 * abs(rhs), (rhs, res)


def test_per_op_autograd_path_equals_fused_trainer():
    """The reference-shaped Python surface (KarmanFlow.step, to_feature, model, to_staggered) composed
    with torch autograd must give the same loss and gradient as the fused C++ training step."""
    B, Y, X, ms = 2, 16, 8, 2
    g, d, vy, vx, re, gts, params, std_v, loss = _oracle_problem(B, Y, X, ms)
    net, tr = _trainer_from(params, g, B, Y, X, ms, std_v)
    tr.fwd_bwd(f32(d), f32(vy), f32(vx), f32(re), f32(torch.stack([s[1] for s in gts])), f32(torch.stack([s[2] for s in gts])))
    dom = sol_amd.Domain([Y, X], box=sol_amd.box[0:200, 0:100])
    sim = sol_amd.KarmanFlow()
    bcv, bcm = sol_amd.velocity_bc_masks(Y, X, batch_size=B)
    st = sol_amd.Fluid(dom, density=f32(d).reshape(B, Y, X, 1), velocity=f32(o.staggered_tensor(vy, vx)), batch_size=B)
    scale_in = torch.tensor([std_v[0], std_v[1], o.STD_RE], device=DEV)
    scale_out = torch.tensor([std_v[0], std_v[1]], device=DEV)
    losses = []
    for i in range(ms):
        st = sim.step(st, re=f32(re), res=X, velBCy=bcv, velBCyMask=bcm)
        corr = sol_amd.to_staggered(net(sol_amd.to_feature(st, f32(re)) / scale_in) * scale_out, dom.box)
        st = st.copied_with(velocity=st.velocity + corr)
        gt = f32(o.staggered_tensor(gts[i][1], gts[i][2]))
        losses.append(0.5 * (((gt - st.velocity.staggered_tensor()) / scale_out) ** 2).sum())
    total = torch.stack(losses).sum() / ms
    total.backward()
    assert abs(float(total) - float(loss)) < 1e-5 * abs(float(loss))
    assert rel(net.params.grad, tr.grads) < 1e-5
    assert rel(net.params.grad, torch.cat([p.grad.reshape(-1) for p in params])) < TOL_GRAD


def test_graph_trainer_equals_fused_trainer_on_mars_moon():
    """GraphTrainer (autograd composition of the HIP ops captured into a hipGraph; the path for networks without a C++ schedule)
    on model_mars_moon must reproduce SolTrainer: loss, per-step losses, gradient, final state, and the weights after Adam."""
    B, Y, X, ms = 2, 16, 8, 2
    g, d, vy, vx, re, gts, params, std_v, loss = _oracle_problem(B, Y, X, ms)
    net, tr = _trainer_from(params, g, B, Y, X, ms, std_v)
    net2 = net.clone()
    gt = sol_amd.make_trainer(net2, None, B, Y, X, ms, g.dx, std_v, o.STD_RE, use_graph=False)
    assert isinstance(gt, sol_amd.SolTrainer) is False or True      # (mars_moon -> SolTrainer by the factory; built directly below)
    gt = sol_amd.GraphTrainer(net2, B, Y, X, ms, std_v, o.STD_RE)
    args = (f32(d), f32(vy), f32(vx), f32(re), f32(torch.stack([s[1] for s in gts])), f32(torch.stack([s[2] for s in gts])))
    for it in range(2):                       # second call = graph replay
        la = tr.train_step(*args, 1e-4, want_final=True)
        lb = gt.train_step(*args, 1e-4, want_final=True)
        assert gt._graph is not None
        assert abs(float(la) - float(lb)) < 1e-5 * abs(float(la))
        assert rel(gt.loss_steps, tr.loss_steps) < 1e-5 and rel(gt.grads, tr.grads) < 1e-4
        for a, b in zip(gt.final, tr.final):
            assert rel(a, b) < 1e-5
        assert rel(net2.params.detach(), net.params.detach()) < 1e-6
    assert abs(float(loss) - float(tr.loss_steps.sum() / ms)) > 0      # (weights moved: not the initial loss any more)


@pytest.mark.parametrize("Y,X,ms", [(16, 8, 2), (64, 32, 2)])
def test_graph_trainer_model_mercury_against_oracle(Y, X, ms):
    """`--model mercury` (karman_train.py:92-99, :394) through the training step: GraphTrainer's replayed hipGraph against the
    eager composition (two different batches through one captured graph) and against the float64 oracle with the mercury
    network (loss 1e-5, full weight gradient 1e-4), then the factory and one Adam step."""
    B = 2
    g = o.geometry(Y, X)
    params = [p.clone().requires_grad_(True) for p in o.init_params_mercury(1)]
    std_v = (0.2, 0.25)
    net = sol_amd.model_mercury(cin=3, cout=2, seed=0)
    net.set_weights([p.detach().numpy() for p in params])
    tg = sol_amd.make_trainer(net, None, B, Y, X, ms, g.dx, std_v, o.STD_RE)
    assert isinstance(tg, sol_amd.GraphTrainer)
    te = sol_amd.GraphTrainer(net.clone(), B, Y, X, ms, std_v, o.STD_RE, use_graph=False)
    for it in range(2):
        d, vy, vx = o.synthetic_state(B, Y, X, 1234 + it)
        re = torch.tensor([o.RE_TRAIN[(i + it) % 6] for i in range(B)], dtype=torch.float64)
        gts = [o.synthetic_state(B, Y, X, 4321 + 7 * it + i, project_it=False) for i in range(ms)]
        args = (f32(d), f32(vy), f32(vx), f32(re), f32(torch.stack([s[1] for s in gts])), f32(torch.stack([s[2] for s in gts])))
        lg, le = float(tg.fwd_bwd(*args)), float(te.fwd_bwd(*args))
        assert tg._graph is not None and abs(lg - le) <= 1e-6 * abs(le) and rel(tg.grads, te.grads) < 1e-6
    loss = o.unrolled_loss(params, d, vy, vx, re, [s[1] for s in gts], [s[2] for s in gts], g, std_v, o.STD_RE)
    loss.backward()
    assert abs(lg - float(loss)) < 1e-5 * abs(float(loss))
    assert rel(tg.grads, torch.cat([p.grad.reshape(-1) for p in params])) < TOL_GRAD
    before = net.params.detach().clone()
    tg.train_step(*args, 1e-4)
    assert tg.t == 1 and float((net.params.detach() - before).abs().max()) > 0


def test_captured_trainers_stay_correct_over_many_replays_with_changing_weights():
    """GraphTrainer (model_mercury, B = 6, 128x64, SOL-2) and BurgersTrainer: twelve replays of the captured graph with the weights
    moved between replays; every replay must report the per-step losses and the gradient of the EAGER composition on the same
    weights.  Regression for the memset-node defect of replayed hipGraphs on ROCm 7.2: the loss was a torch reduction whose
    semaphores are cleared by cudaMemsetAsync -- a memset node in the captured graph -- and after a few replays the reduction
    folded early (per-step losses 0.5x / 2x the true values; found on the 3-D trainer at full size).  The loss is now one
    deterministic kernel (ops.l2_loss / sol_l2_loss_fwd_bwd)."""
    B, Y, X, ms = 6, 128, 64, 2
    g = o.geometry(Y, X)
    mk = ops.SceneMasks(g.active, g.inflow, g.bc_mask, g.bc_mask)
    gen = torch.Generator().manual_seed(3)
    d, vy, vx = (f32(t) for t in o.synthetic_state(B, Y, X, 5, project_it=False))
    re = f32(torch.tensor([o.RE_TRAIN[i % 6] for i in range(B)]))
    gts = [o.synthetic_state(B, Y, X, 900 + i, project_it=False) for i in range(ms)]
    gy, gx = f32(torch.stack([s[1] for s in gts])), f32(torch.stack([s[2] for s in gts]))
    nets = [sol_amd.model_mercury(cin=3, cout=2, seed=1, device=DEV) for _ in range(2)]
    trs = [sol_amd.GraphTrainer(nets[k], B, Y, X, ms, (0.2, 0.25), o.STD_RE, dx=g.dx, masks=mk, use_graph=(k == 0)) for k in range(2)]
    p0 = nets[0].params.detach().clone()
    for it in range(12):
        with torch.no_grad():
            pert = p0 * (1.0 + 0.02 * torch.randn(p0.shape, generator=gen, dtype=torch.float32).to(DEV)) if it % 3 else p0
            for n in nets:
                n.params.copy_(pert)
        outs = []
        for tr in trs:
            tr.grads.zero_()
            tr.fwd_bwd(d, vy, vx, re, gy, gx)
            torch.cuda.synchronize()
            outs.append((tr.loss_steps.clone(), tr.grads.clone()))
        assert rel(outs[0][0], outs[1][0]) < 1e-6, (it, outs[0][0].tolist(), outs[1][0].tolist())
        assert rel(outs[0][1], outs[1][1]) < 1e-5, (it, rel(outs[0][1], outs[1][1]))


@pytest.mark.parametrize("name,Y,X,pretf", [("mercury", 32, 16, False), ("mercury", 128, 64, True), ("mars_moon", 128, 64, False), ("mars_moon", 64, 32, True)])
def test_graph_trainer_manual_schedule_equals_autograd_composition(name, Y, X, pretf):
    """GraphTrainer's hand-written schedule over the C ABI (round 6: trainer.GraphTrainer._unrolled_schedule + schedule2d.NetSchedule2D --
    karman_train.py:92-99 / 101-138 and :397-457 differentiated by hand) against the torch-autograd composition of the differentiable HIP ops
    it replaces as the default (schedule="autograd"): per-step losses, the flat gradient, the final state; eager and through the captured
    graph (kernel nodes only -- the capture guard runs on both); with separate input / output scales (--pretf).  128x64: the 32-channel
    layers run the fp16 three-product kernels with the absmax slots handed from layer to layer (the autograd form: bf16 six-product)."""
    from sol_amd import _lib
    B, ms = 2, 3
    g = o.geometry(Y, X)
    mk = ops.SceneMasks(g.active, g.inflow, g.bc_mask, g.bc_mask)
    d, vy, vx = (f32(t) for t in o.synthetic_state(B, Y, X, 5))
    re = f32(torch.tensor([o.RE_TRAIN[i % 6] for i in range(B)]))
    gts = [o.synthetic_state(B, Y, X, 900 + i, project_it=False) for i in range(ms)]
    gy, gx = f32(torch.stack([s[1] for s in gts])), f32(torch.stack([s[2] for s in gts]))
    mk_net = sol_amd.model_mercury if name == "mercury" else sol_amd.model_mars_moon
    kw = dict(in_std_v=(0.3, 0.35), out_std_v=(0.15, 0.1)) if pretf else {}
    res = {}
    for sched in ("manual", "autograd"):
        for graph in (False, True):
            net = mk_net(cin=3, cout=2, seed=4, device=DEV)
            with torch.no_grad():
                net.params.add_(0.01 * torch.randn(net.params.shape, generator=torch.Generator().manual_seed(9)).to(DEV))      # non-zero biases
            tr = sol_amd.GraphTrainer(net, B, Y, X, ms, (0.2, 0.25), o.STD_RE, dx=g.dx, masks=mk, use_graph=graph, schedule=sched, **kw)
            for _ in range(2):                  # second call: the replay (or the second eager sweep: partial buffers re-zeroed)
                loss = tr.fwd_bwd(d, vy, vx, re, gy, gx, want_final=True)
            torch.cuda.synchronize()
            res[(sched, graph)] = (float(loss), tr.loss_steps.clone(), tr.grads.clone(), [t.clone() for t in tr.final])
            if graph:
                census = _lib.graph_census(tr._graph.raw_cuda_graph())
                assert set(census) <= {"kernel", "empty"}, census
                res[(sched, "nodes")] = census["kernel"]
    ref = res[("autograd", False)]
    for key in (("manual", False), ("manual", True), ("autograd", True)):
        got = res[key]
        assert abs(got[0] - ref[0]) < 2e-6 * abs(ref[0]), (key, got[0], ref[0])
        assert rel(got[1], ref[1]) < 2e-6, (key, rel(got[1], ref[1]))
        assert rel(got[2], ref[2]) < 3e-5, (key, rel(got[2], ref[2]))
        assert all(rel(a, b) < 2e-6 for a, b in zip(got[3], ref[3])), key
    assert torch.equal(res[("manual", False)][2], res[("manual", True)][2])          # the replayed graph IS the eager schedule: bit for bit
    assert res[("manual", "nodes")] < 0.75 * res[("autograd", "nodes")], (res[("manual", "nodes")], res[("autograd", "nodes")])


def test_shard_gradients_sum_to_the_large_batch_gradient():
    """What the RCCL all-reduce(SUM) relies on: grad(batch) == grad(shard 0) + grad(shard 1)."""
    B, Y, X, ms = 4, 16, 8, 2
    g = o.geometry(Y, X)
    d, vy, vx = (f32(t) for t in o.synthetic_state(B, Y, X, 1234))
    re = f32(sol_amd.synthetic.reynolds(B))
    gy, gx = (f32(t) for t in sol_amd.synthetic.frames(ms, B, Y, X, 4321))
    params = o.init_params(0)
    _, full = _trainer_from(params, g, B, Y, X, ms, (0.2, 0.2))
    lf = full.fwd_bwd(d, vy, vx, re, gy, gx)
    _, half = _trainer_from(params, g, B // 2, Y, X, ms, (0.2, 0.2))
    acc = torch.zeros_like(full.grads)
    lsum = 0.0
    for lo in (0, 2):
        sl = slice(lo, lo + 2)
        lsum += float(half.fwd_bwd(d[sl].contiguous(), vy[sl].contiguous(), vx[sl].contiguous(), re[sl].contiguous(),
                                   gy[:, sl].contiguous(), gx[:, sl].contiguous()))
        acc += half.grads
    assert abs(lsum - float(lf)) < 1e-5 * abs(float(lf))
    assert rel(acc, full.grads) < 1e-5


@pytest.mark.parametrize("Y,X,n", [(64, 32, 3), (64, 32, 10), (128, 64, 2), (128, 64, 3), (128, 64, 1)])
def test_rollout_against_oracle(Y, X, n):
    """odd / even step counts (the state ping-pongs between the caller's buffers and the workspace), more steps than absmax
    slot sets, and the 128x64 path whose last CNN layer applies the correction itself and whose passive density rides one step
    behind in the solver launches (even, odd and a single step: the last advection is its own launch)"""
    B = 1
    g = o.geometry(Y, X)
    d, vy, vx = o.synthetic_state(B, Y, X, 21)
    re = torch.tensor([o.RE_TRAIN[2]], dtype=torch.float64)
    params = o.init_params(3)
    std_v = (0.2, 0.2)
    rd, ry, rx = d, vy, vx
    for _ in range(n):
        rd, ry, rx = o.karman_step(rd, ry, rx, re, g)
        cy, cx = o.correction(params, ry, rx, re, std_v, o.STD_RE)
        ry, rx = ry + cy, rx + cx
    mk = ops.SceneMasks(g.active, g.inflow, g.bc_mask, g.bc_mask)
    net = sol_amd.model_mars_moon(cin=3, cout=2, seed=3)
    ro = sol_amd.SolRollout(net, mk, B, Y, X, g.dx, std_v, o.STD_RE)
    hd, hy, hx = f32(d), f32(vy), f32(vx)
    its = ro.run(hd, hy, hx, f32(re), n)
    assert its.shape == (n, B)
    assert int(its.min()) > 5 or mk.direct is not None      # the direct solver reports 0 iterations
    assert rel(hy, ry) < TOL_FIELD and rel(hx, rx) < TOL_FIELD and rel(hd, rd) < TOL_FIELD


# ---------------------------------------------------------------------------------------------
# Burgers (BASELINE configs[0]: 32x32, msteps=1)
# ---------------------------------------------------------------------------------------------
def test_burgers_step_against_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "burgers_step_32x32.npz"))
    B, Yp1, X = z["vy"].shape
    Y = Yp1 - 1
    dt, nu = float(z["dt"]), float(z["nu"])
    cfg = sol_amd._lib.BurgersCfg(B, Y, X, 1.0, dt)
    circ = ops.burgers_circ(Y, X, dt * nu)
    vy = f32(z["vy"]).requires_grad_(True)
    vx = f32(z["vx"]).requires_grad_(True)
    fy = f32(z["fy"]).requires_grad_(True)
    hy, hx = ops.burgers_step(vy, vx, fy, f32(z["fx"]), cfg, circ)
    ((hy * f32(z["wy"])).sum() + (hx * f32(z["wx"])).sum()).backward()
    assert rel(hy, z["vy_out"]) < TOL_FIELD and rel(hx, z["vx_out"]) < TOL_FIELD
    assert rel(vy.grad, z["g_vy"]) < TOL_GRAD and rel(vx.grad, z["g_vx"]) < TOL_GRAD
    assert rel(fy.grad, dt * z["wy"]) < 1e-6
    # BurgersTest.step (no force) == step_with_f with zero force
    ny, nx = ops.burgers_step(vy.detach(), vx.detach(), None, None, cfg, circ)
    zy, zx = ops.burgers_step(vy.detach(), vx.detach(), torch.zeros_like(vy), torch.zeros_like(vx), cfg, circ)
    assert torch.equal(ny, zy) and torch.equal(nx, zx)


@pytest.mark.parametrize("Y,X", [(128, 128), (96, 160)])
def test_burgers_large_grid_forward_step_against_oracle(Y, X):
    """The reference generates its Burgers training data at 128 x 128 (burgers/Makefile:19-29, `-r 128 --dt 0.1`): beyond the
    one-workgroup LDS kernels.  The forward-only multi-workgroup step (sol_burgers_step_fwd_large) against the float64 oracle
    (FFT diffusion), with and without force, through ops and through the reference-shaped BurgersTest surface; at 64 x 64 it
    must agree with the LDS kernel."""
    B, dt, nu = 2, 0.1, 0.1
    gen = torch.Generator().manual_seed(3)
    sm = lambda *shape: o._smooth(torch.randn(*shape, generator=gen, dtype=torch.float64))
    vy, vx = 0.8 * sm(B, Y + 1, X), 0.8 * sm(B, Y, X + 1)
    fy, fx = 0.2 * sm(B, Y + 1, X), 0.2 * sm(B, Y, X + 1)
    dx = 32.0 / Y                                                  # -l 32
    cfg = sol_amd._lib.BurgersCfg(B, Y, X, dx, dt)
    circ = ops.burgers_circ(Y, X, dt * nu)
    for force in (True, False):
        ry, rx = o.burgers_step(vy, vx, dt, nu, fy if force else None, fx if force else None, dx=dx)
        hy, hx = ops.burgers_step_large(f32(vy), f32(vx), f32(fy) if force else None, f32(fx) if force else None, cfg, circ)
        assert rel(hy, ry) < TOL_FIELD and rel(hx, rx) < TOL_FIELD
    dom = sol_amd.Domain([Y, X], box=sol_amd.box([32.0, 32.0 * X / Y]), boundaries=sol_amd.PERIODIC)
    st = sol_amd.BurgersVelocitySMAC(dom, velocity=f32(o.staggered_tensor(vy, vx)), batch_size=B)
    fr = sol_amd.BurgersVelocitySMAC(dom, velocity=f32(o.staggered_tensor(fy, fx)), batch_size=B)
    out = sol_amd.BurgersTest().step_with_f(st, fr, dt=dt).velocity.staggered_tensor()
    ry, rx = o.burgers_step(vy, vx, dt, nu, fy, fx, dx=dx)
    assert rel(out, o.staggered_tensor(ry, rx)) < TOL_FIELD
    with pytest.raises(NotImplementedError):
        st2 = sol_amd.BurgersVelocitySMAC(dom, velocity=f32(o.staggered_tensor(vy, vx)).requires_grad_(True), batch_size=B)
        sol_amd.BurgersTest().step(st2, dt=dt)
    # same operator as the one-workgroup kernel where both apply
    c64 = sol_amd._lib.BurgersCfg(B, 64, 64, 0.5, dt)
    k64 = ops.burgers_circ(64, 64, dt * nu)
    a, b = f32(0.8 * sm(B, 65, 64)), f32(0.8 * sm(B, 64, 65))
    ly, lx = ops.burgers_step(a, b, None, None, c64, k64)
    gy, gx = ops.burgers_step_large(a, b, None, None, c64, k64)
    assert rel(gy, ly) < 1e-6 and rel(gx, lx) < 1e-6


# ---------------------------------------------------------------------------------------------
# scripts: data generation -> training -> roll-out on a tiny scene (flags of the reference scripts)
# ---------------------------------------------------------------------------------------------
def test_scripts_end_to_end(tmp_path):
    import importlib.util
    import pickle
    sdir = os.path.join(os.path.dirname(os.path.abspath(sol_amd.__file__)), "scripts")
    import sys
    sys.path.insert(0, sdir)

    def load(name):
        spec = importlib.util.spec_from_file_location("sol_script_" + name, os.path.join(sdir, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    data = str(tmp_path / "set")
    for re_nr in (1.6e5, 3.2e5):
        p = load("karman").main(["-o", data, "-r", "32", "-t", "14", "-s", "1", "--re", str(re_nr)])
        assert len([f for f in os.listdir(p) if f.startswith("velo_")]) == 12
    tf = str(tmp_path / "tf")
    # (a) ONE training step (-t 3 -m 2: a single (batch, step) pair) against the oracle on the very batch the script drew:
    #     same dataset object, same `random` seed -> same shuffled (sim, frame) pairs; weights = the seed-0 initialisation
    import random
    from sol_amd import scene as sc
    tf1 = str(tmp_path / "tf1")
    loss1 = load("karman_train").main(["--train", data, "-s", "1", "-n", "2", "-b", "2", "-t", "3", "-m", "2", "-e", "1",
                                       "--lr", "1e-4", "--tf", tf1, "--seed", "0"])
    random.seed(0)
    ds = sc.PhifDataset(data, 3, 2, 2, print_fn=lambda *a: None, scale=1)
    ds.newEpoch(exclude_tail=2)
    dens, velo, ext = ds.getData(consecutive_frames=2, with_skip=1)
    f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64)
    vys, vxs = zip(*[sc.split_staggered(v) for v in velo])
    ref = o.unrolled_loss(o.init_params(0), f64(dens[0][..., 0]), f64(vys[0]), f64(vxs[0]), f64(np.asarray(ext, dtype=np.float64)),
                          [f64(v) for v in vys[1:]], [f64(v) for v in vxs[1:]], o.geometry(64, 32),
                          tuple(float(v) for v in ds.dataStats["std"][1]), float(ds.dataStats["ext.std"][0]))
    assert loss1 is not None and abs(loss1 - float(ref)) < 2e-5 * abs(float(ref)), (loss1, float(ref))
    # (b) a longer run for the artefacts
    loss = load("karman_train").main(["--train", data, "-s", "1", "-n", "2", "-b", "2", "-t", "12", "-m", "2", "-e", "1",
                                      "--lr", "1e-4", "--tf", tf, "--seed", "0"])
    assert loss is not None and np.isfinite(loss)
    assert os.path.isfile(tf + "/model.pt") and os.path.isfile(tf + "/dataStats.pickle")
    # (c) --model mercury (karman_train.py:394 `eval('model_'+params['model'])`): first-step loss against the oracle's mercury
    tfm = str(tmp_path / "tf_mercury")
    lossm = load("karman_train").main(["--train", data, "-s", "1", "-n", "2", "-b", "2", "-t", "3", "-m", "2", "-e", "1", "--model", "mercury",
                                       "--lr", "1e-4", "--tf", tfm, "--seed", "0"])
    refm = o.unrolled_loss(o.init_params_mercury(0), f64(dens[0][..., 0]), f64(vys[0]), f64(vxs[0]), f64(np.asarray(ext, dtype=np.float64)),
                           [f64(v) for v in vys[1:]], [f64(v) for v in vxs[1:]], o.geometry(64, 32),
                           tuple(float(v) for v in ds.dataStats["std"][1]), float(ds.dataStats["ext.std"][0]))
    assert lossm is not None and abs(lossm - float(refm)) < 2e-5 * abs(float(refm)), (lossm, float(refm))
    assert sol_amd.ConvNet.load(tfm + "/model.pt").name == "mercury"
    with open(tf + "/dataStats.pickle", "rb") as f:
        st = pickle.load(f)
    assert len(st["std"][1]) == 2 and st["ext.std"][0] > 0
    out = load("karman_apply").main(["-r", "32", "-t", "4", "-o", str(tmp_path / "run"), "--stats", tf + "/dataStats.pickle",
                                     "--model", tf + "/model.pt", "--re", "2.4e5"])
    from sol_amd import scene
    v = scene.read_zipped_array(out + "/velTf_000003.npz")
    c = scene.read_zipped_array(out + "/corTf_000003.npz")
    assert v.shape == (1, 65, 33, 2) and c.shape == (1, 65, 33, 2) and np.isfinite(v).all() and np.abs(c).max() > 0
    # the reference's own data-generation chain (Makefile:19-44): hi-res run at -r 128 (256 x 128, forward-only large-grid
    # path), then a 4x down-sampled lo-res run started from one of its frames
    hi = load("karman").main(["-o", str(tmp_path / "hires"), "-r", "128", "-t", "6", "-s", "2", "--re", "1.6e5"])
    vh = scene.read_zipped_array(hi + "/velo_000005.npz")
    assert vh.shape == (1, 257, 129, 2) and np.isfinite(vh).all()
    lo = load("karman").main(["-o", str(tmp_path / "lores"), "-r", "32", "-t", "4", "-s", "0", "-d", "4", "--re", "1.6e5",
                              "--initdH", hi + "/dens_000005.npz", "--initvH", hi + "/velo_000005.npz"])
    vl = scene.read_zipped_array(lo + "/velo_000003.npz")
    assert vl.shape == (1, 65, 33, 2) and np.isfinite(vl).all()


def test_burgers_training_script(tmp_path):
    """BASELINE configs[0]: burgers 32x32, msteps=1 -- the reference's script shape on the HIP ops."""
    import importlib.util
    import pickle
    from sol_amd import scene
    sdir = os.path.join(os.path.dirname(os.path.abspath(sol_amd.__file__)), "scripts")
    import sys
    sys.path.insert(0, sdir)
    spec = importlib.util.spec_from_file_location("sol_script_burgers_train", os.path.join(sdir, "burgers_train.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # tiny synthetic set: forced Burgers roll-outs by the HIP step itself (data generation proper is out of scope)
    B, Y, X, frames, dt = 1, 32, 32, 8, 0.1
    dom = sol_amd.Domain([Y, X], box=sol_amd.box([32, 32]), boundaries=sol_amd.PERIODIC)
    sim = sol_amd.BurgersTest()
    gen = torch.Generator().manual_seed(0)
    for s in range(2):
        path = scene.scene_create(str(tmp_path / "set"))
        with open(path + "/params.pickle", "wb") as f:
            pickle.dump({"seed": s}, f)
        v = f32(0.3 * sol_amd.synthetic._smooth(torch.randn(B, Y + 1, X + 1, generator=gen, dtype=torch.float64))).unsqueeze(-1).repeat(1, 1, 1, 2)
        st = sol_amd.BurgersVelocitySMAC(dom, velocity=v, batch_size=B)
        for i in range(frames):
            fr_t = f32(0.15 * sol_amd.synthetic._smooth(torch.randn(B, Y + 1, X + 1, generator=gen, dtype=torch.float64))).unsqueeze(-1).repeat(1, 1, 1, 2)
            fr = sol_amd.BurgersVelocitySMAC(dom, velocity=fr_t, batch_size=B)
            scene.scene_write(path, [st.velocity.staggered_tensor().cpu().numpy(), fr.velocity.staggered_tensor().cpu().numpy()], ["velo", "forc"], i)
            with torch.no_grad():
                st = sim.step_with_f(st, fr, dt=dt)
    tf = str(tmp_path / "tf")
    loss = mod.main(["--train", str(tmp_path / "set"), "-s", "1", "-n", "2", "-b", "2", "-t", str(frames), "-m", "1", "-e", "1",
                     "--lr", "1e-4", "--dt", str(dt), "--tf", tf])
    assert loss is not None and np.isfinite(loss)
    assert os.path.isfile(tf + "/model.pt") and os.path.isfile(tf + "/model_epoch0001.pt")


def test_model_mercury_against_torch_reference():
    """model_mercury (karman_train.py:92-99) on the split 32-channel kernels vs plain fp64 torch convs."""
    net = sol_amd.model_mercury(cin=3, cout=2, seed=5)
    assert net.n_params == 25 * 3 * 32 + 32 + 25 * 32 * 64 + 64 + 25 * 64 * 2 + 2
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(2, 16, 8, 3, generator=gen, dtype=torch.float64)
    ws = [torch.as_tensor(w, dtype=torch.float64).requires_grad_(True) for w in net.get_weights()]
    ws[1].data += 0.1
    ws[3].data -= 0.05
    net.set_weights([w.detach().numpy() for w in ws])
    h = torch.relu(o._conv(x, ws[0], ws[1]))
    h = torch.relu(o._conv(h, ws[2], ws[3]))
    y = o._conv(h, ws[4], ws[5])
    gy = torch.randn(y.shape, generator=gen, dtype=torch.float64)
    (y * gy).sum().backward()
    hy = net(f32(x))
    (hy * f32(gy)).sum().backward()
    assert rel(hy, y) < 2e-6
    gref = torch.cat([w.grad.reshape(-1) for w in ws])
    assert rel(net.params.grad, gref) < 5e-6


def test_large_grid_forward_step_against_oracle():
    """the reference generates its data at 256 x 128 (karman.py -r 128).  Forward-only multi-launch step with
    the direct pressure solver (window 32) against the float64 oracle, two consecutive steps, and through KarmanFlow.step."""
    B, Y, X = 2, 256, 128
    g = o.geometry(Y, X)
    mk = ops.SceneMasks(g.active, g.inflow, g.bc_mask, g.bc_mask)
    assert mk.large and mk.direct is not None and mk.coarse_inv is None
    d, vy, vx = o.synthetic_state(B, Y, X, 77)
    re = torch.tensor(o.RE_TRAIN[:B], dtype=torch.float64)
    cfg = ops.karman_cfg(B, Y, X, g.dx, masks=mk)
    hd, hy, hx = f32(d), f32(vy), f32(vx)
    rd, ry, rx = d, vy, vx
    for _ in range(2):
        rd, ry, rx = o.karman_step(rd, ry, rx, re, g)
        hd, hy, hx = ops.karman_step_large(hd, hy, hx, f32(re), cfg, mk)
        assert rel(hy, ry) < TOL_FIELD and rel(hx, rx) < TOL_FIELD and rel(hd, rd) < TOL_FIELD
    # the reference-shaped surface dispatches to the same path
    dom = sol_amd.Domain([Y, X], box=sol_amd.box[0:200, 0:100])
    sim = sol_amd.KarmanFlow()
    bcv, bcm = sol_amd.velocity_bc_masks(Y, X, batch_size=B)
    st = sol_amd.Fluid(dom, density=f32(d).reshape(B, Y, X, 1), velocity=f32(o.staggered_tensor(vy, vx)), batch_size=B)
    with torch.no_grad():
        st = sim.step(st, re=f32(re), res=X, velBCy=bcv, velBCyMask=bcm)
    r1 = o.karman_step(d, vy, vx, re, g)
    assert rel(st.velocity.data[0].data.reshape(B, Y + 1, X), r1[1]) < TOL_FIELD
    assert rel(st.velocity.data[1].data.reshape(B, Y, X + 1), r1[2]) < TOL_FIELD


# ---------------------------------------------------------------------------------------------
# BASELINE configs[2] at its real depth, through the hipGraph
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["split", "bf16x6", "fp32"])
def test_sol32_bench_workload_against_golden(golden_dir, precision):
    """karman-2d 128x64, B=6, msteps=32 (the workload bench.py times) through the REPLAYED hipGraph, in each of the three
    convolution arithmetics (fp16x3 split = the headline, bf16x6 = true 24-bit operands, strict fp32 MFMA): 32-step unroll, absmax
    slot rotation, weight gradients batched over / fused into the 32 adjoint launches, density riding in the forward
    launches -- against the float64 oracle fixture (tests/golden/make_golden.py --sol32): loss, the 32 per-step losses,
    per-tensor gradient norms, every 16th gradient element, the final state, and the loss after each of 3 Adam steps at
    lr 1e-4 (2386.489 -> 118279.49 -> 11700.69: the divergent trajectory the round-1 driver bench ran into)."""
    z = np.load(os.path.join(golden_dir, "train_128x64_sol32.npz"))
    B, Y, X, ms = int(z["B"]), int(z["Y"]), int(z["X"]), int(z["msteps"])
    w = _cached(("sol32", B, Y, X, ms), lambda: o.bench_workload(B, Y, X, ms))   # inputs: regenerated (deterministic float64), not stored
    net, tr = _trainer_from(w["params"], w["geom"], B, Y, X, ms, w["std_v"], conv_precision=precision)
    assert tr.use_graph and tr.masks.direct is not None
    args = (f32(w["d0"]), f32(w["vy0"]), f32(w["vx0"]), f32(w["re"]), f32(torch.stack(w["gt_vy"])), f32(torch.stack(w["gt_vx"])))
    traj = []
    for t in range(len(z["loss_traj"])):
        loss = tr.fwd_bwd(*args, want_final=True)
        traj.append(float(loss))
        if t == 0:
            assert tr._graphs.get(True) is not None
            assert np.allclose(tr.loss_steps.cpu().numpy(), z["loss_steps"], rtol=2e-5)
            assert rel(tr.grads[::16], z["grads_sub16"]) < TOL_GRAD
            assert abs(float(tr.grads.double().norm()) - float(z["grad_l2"])) < 1e-4 * float(z["grad_l2"])
            norms = np.array([float(tr.grads[net.offsets[k]:net.offsets[k + 1]].double().norm()) for k in range(24)])
            assert np.allclose(norms, z["grad_norms"], rtol=3e-4), np.abs(norms / z["grad_norms"] - 1).max()
            assert rel(tr.final[1], z["vy_final"]) < TOL_FIELD and rel(tr.final[2], z["vx_final"]) < TOL_FIELD
            assert rel(tr.final[0], z["d_final"]) < TOL_FIELD
            # the first TF-Adam update in closed form: m = 0.1 g, v = 0.001 g^2, lr_t = lr sqrt(1 - 0.999) / (1 - 0.9)
            g64, p64 = tr.grads.double(), net.params.detach().double()
            expect = p64 - float(z["lr"]) * (1 - 0.999) ** 0.5 / (1 - 0.9) * 0.1 * g64 / ((0.001 * g64 * g64).sqrt() + 1e-8)
        tr.apply_gradients(float(z["lr"]))
        if t == 0:
            assert rel(net.params.detach(), expect) < 1e-6
    assert len(tr._graphs) == 1 and tr._captures == 0         # one graph, captured once, replayed
    assert np.allclose(traj, z["loss_traj"], rtol=5e-5), (traj, z["loss_traj"])


C3_FIELD_ALLOWANCE = 1.25          # final velocity / density: err(split) <= 1.25 err(fp32), err(bf16x6) <= 1.25 err(fp32) (round-5 verdict, "next round" 2)
C3_LOSS_GRAD_ENVELOPE = 3.0        # per-step losses / gradient: measured 1.86x / 2.26x (fp16x3), 1.30x / 1.53x (bf16x6) of the strict trainer's 2.7e-7 / 1.7e-7
C3_LOSS_GRAD_ABS = 1e-6            # ... and in absolute terms 10x / 100x inside the tolerances (1e-5 fields, 1e-4 gradients)


def round_to_two_fp16_planes(t):
    """w -> (g1 + g2 / 2048) / 2^s with g1 = fp16(w 2^s), g2 = fp16((w 2^s - g1) 2048): what the split kernels' weight packing keeps of an fp32
    weight (relative error <= 2^-23, i.e. at most the distance to the NEIGHBOURING fp32 number; the power-of-two scale does not change it)."""
    t = t.detach().double()
    s = 2.0 ** (14 - int(np.floor(np.log2(float(t.abs().max())))))
    ws = (t * s).float()
    g1 = ws.half().float()
    g2 = ((ws - g1) * 2048.0).half().float()
    return (g1.double() + g2.double() / 2048.0) / s


def _c3_step(golden_dir, precision, round_weights=False):
    """ONE first training step of the C3 workload (B=6, 128x64, SOL-32, replayed hipGraph) in the given convolution arithmetic: the trainer
    (loss_steps, grads, final) and the fixture.  round_weights: the ten 32 -> 32 kernels replaced by their two-fp16-plane roundings."""
    z = np.load(os.path.join(golden_dir, "train_128x64_sol32.npz"))
    B, Y, X, ms = int(z["B"]), int(z["Y"]), int(z["X"]), int(z["msteps"])
    w = _cached(("sol32", B, Y, X, ms), lambda: o.bench_workload(B, Y, X, ms))
    params = [p.detach() for p in w["params"]]
    if round_weights:
        params = [round_to_two_fp16_planes(p) if (p.dim() == 4 and tuple(p.shape[2:]) == (32, 32)) else p for p in params]
    net, tr = _trainer_from(params, w["geom"], B, Y, X, ms, w["std_v"], conv_precision=precision)
    loss = tr.fwd_bwd(f32(w["d0"]), f32(w["vy0"]), f32(w["vx0"]), f32(w["re"]), f32(torch.stack(w["gt_vy"])), f32(torch.stack(w["gt_vx"])),
                      want_final=True)
    torch.cuda.synchronize()
    return z, tr, float(loss)


def _c3_errors(golden_dir, precision):
    """relative-L2 errors of that step against the float64 fixture: final state, the 32 per-step losses, the gradient (every 16th element)"""
    z, tr, loss = _c3_step(golden_dir, precision)
    return {"vy_final": rel(tr.final[1], z["vy_final"]), "vx_final": rel(tr.final[2], z["vx_final"]), "d_final": rel(tr.final[0], z["d_final"]),
            "loss_steps": rel(tr.loss_steps, z["loss_steps"]), "gradient_every_16th": rel(tr.grads[::16], z["grads_sub16"]),
            "loss": abs(loss - float(z["loss_traj"][0])) / abs(float(z["loss_traj"][0]))}


def test_sol32_workload_error_of_each_conv_arithmetic_against_float64(golden_dir):
    """The WORKLOAD-level form of the precision claim behind the bench line's dtype (reference arithmetic: tf.float32,
    karman-2d/karman_train.py:376-377): the error against the float64 fixture of the fp16x3 `split` trainer (the headline), the `bf16x6`
    trainer and the strict fp32-MFMA trainer at the configuration the metric is quoted on (C3: B=6, 128x64, SOL-32, through the replayed
    graph: 32 solver steps, 384 convolutions, the reverse sweep).  What holds, and is asserted:
      * final velocity / density: err(split), err(bf16x6) <= 1.25 err(fp32)        (measured 0.94 .. 1.08: the solver's fp32 error dominates)
      * per-step losses and gradient: every arithmetic <= 1e-6 (tolerances: 1e-5 / 1e-4), but NOT <= 1.25 err(fp32): measured 5.0e-7 /
        3.8e-7 (fp16x3) and 3.5e-7 / 2.5e-7 (bf16x6) against 2.7e-7 / 1.7e-7 (fp32), i.e. 1.9x / 2.3x and 1.3x / 1.5x.  Bounded here by a
        regression envelope (3x); WHY is the next test (the weights' two-plane representation), profiles/r06_precision_at_c3.json.
    (The fixture's fields are stored as fp32: 3e-8 of quantisation under errors of order 1e-7.)  The triples are written to
    gpurun_out/precision_at_c3.json -> profiles/r06_precision_at_c3.json, which bench.py quotes in its dtype_note."""
    errs = {p: _c3_errors(golden_dir, p) for p in ("split", "bf16x6", "fp32")}
    ratios = {p: {k: errs[p][k] / max(errs["fp32"][k], 1e-300) for k in errs[p]} for p in ("split", "bf16x6")}
    try:
        import json
        root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
        os.makedirs(root, exist_ok=True)
        with open(os.path.join(root, "precision_at_c3.json"), "w") as f:
            json.dump({"workload": "karman-2d 128x64 SOL-32, B=6, first training step through the replayed hipGraph, vs tests/golden/train_128x64_sol32.npz (float64 oracle)",
                       "relative_l2_error": errs, "ratio_to_strict_fp32": ratios,
                       "asserted": {"fields_ratio_max": C3_FIELD_ALLOWANCE, "loss_gradient_ratio_envelope": C3_LOSS_GRAD_ENVELOPE, "loss_gradient_abs_max": C3_LOSS_GRAD_ABS},
                       "device": torch.cuda.get_device_name(0)}, f, indent=1)
    except OSError:
        pass
    for p in ("split", "bf16x6"):
        for k in ("vy_final", "vx_final", "d_final"):
            assert errs[p][k] <= C3_FIELD_ALLOWANCE * errs["fp32"][k], (p, k, errs[p][k], errs["fp32"][k])
        for k in ("loss_steps", "gradient_every_16th"):
            assert errs[p][k] <= C3_LOSS_GRAD_ENVELOPE * errs["fp32"][k], (p, k, errs[p][k], errs["fp32"][k])
    for p in errs:
        assert max(errs[p]["vy_final"], errs[p]["vx_final"], errs[p]["d_final"]) < TOL_FIELD and errs[p]["gradient_every_16th"] < TOL_GRAD, (p, errs[p])
        assert errs[p]["loss_steps"] <= C3_LOSS_GRAD_ABS and errs[p]["gradient_every_16th"] <= C3_LOSS_GRAD_ABS, (p, errs[p])


def test_sol32_split_trainer_equals_strict_fp32_on_weights_representable_in_two_fp16_planes(golden_dir):
    """WHY the split trainer's loss / gradient error at C3 is ~2x the strict trainer's although its single convolution is MORE accurate
    (test_split_conv_relative_l2_not_worse_than_fp32_mfma; tools/conv_bias_probe.py: no coherent gain, |eps| < 4e-9): the split kernels keep
    a WEIGHT as two fp16 planes, w' = g1 + g2 / 2048 -- a rounding of at most 2^-23 relative, i.e. to a neighbouring fp32 number at worst.
    Activation roundings differ from pixel to pixel and average out; the weight rounding is ONE fixed perturbation dw of the network, the same
    at every pixel of all 32 steps, and this workload's loss is very sensitive to the weights (|grad| = 1.06e7 at a loss of 2386; one Adam
    step at lr 1e-4 takes it to 118279).  Measured (tools/precision_at_c3.py, profiles/r06_precision_diag.txt): rounding the ten 32 -> 32
    kernels that way moves the STRICT fp32 trainer's per-step losses by +1.0e-7 .. +3.3e-7 -- the whole split-minus-fp32 offset (+1.9e-7 ..
    +3.3e-7) -- and on such weights the two trainers agree to 1.4e-7 (losses) / 9.7e-8 (gradient) instead of 3.3e-7 / 3.2e-7.
    Asserted: (a) on representable weights split == strict fp32 within 2.5e-7 per step loss and 2e-7 gradient; (b) with the original
    weights their distance is at least 1.5x that of (a) -- the representation, not the arithmetic, is what separates them; (c) the strict
    trainer itself moves by the same amount when only its weights are rounded."""
    _, tr_s, _ = _c3_step(golden_dir, "split")
    ls_s, g_s = tr_s.loss_steps.double().cpu().numpy(), tr_s.grads[::16].double().cpu()
    _, tr_f, _ = _c3_step(golden_dir, "fp32")
    ls_f, g_f = tr_f.loss_steps.double().cpu().numpy(), tr_f.grads[::16].double().cpu()
    _, tr_sr, _ = _c3_step(golden_dir, "split", round_weights=True)
    ls_sr, g_sr = tr_sr.loss_steps.double().cpu().numpy(), tr_sr.grads[::16].double().cpu()
    _, tr_fr, _ = _c3_step(golden_dir, "fp32", round_weights=True)
    ls_fr, g_fr = tr_fr.loss_steps.double().cpu().numpy(), tr_fr.grads[::16].double().cpu()
    d_repr, d_orig, d_round = np.abs(ls_sr / ls_fr - 1), np.abs(ls_s / ls_f - 1), np.abs(ls_fr / ls_f - 1)
    assert d_repr.max() <= 2.5e-7 and rel(g_sr, g_fr) <= 2e-7, (d_repr.max(), rel(g_sr, g_fr))
    assert d_orig.mean() >= 1.5 * d_repr.mean() and rel(g_s, g_f) >= 1.5 * rel(g_sr, g_fr), (d_orig.mean(), d_repr.mean(), rel(g_s, g_f), rel(g_sr, g_fr))
    assert d_round.mean() >= 0.5 * d_orig.mean(), (d_round.mean(), d_orig.mean())
    try:
        import json
        root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")
        os.makedirs(root, exist_ok=True)
        with open(os.path.join(root, "precision_at_c3_weights.json"), "w") as f:
            json.dump({"per_step_loss_rel_distance_split_vs_strict_fp32": {"original_weights_mean": float(d_orig.mean()), "original_weights_max": float(d_orig.max()),
                                                                          "two_plane_representable_weights_mean": float(d_repr.mean()), "two_plane_representable_weights_max": float(d_repr.max())},
                       "strict_fp32_rounded_vs_original_weights_mean": float(d_round.mean()),
                       "gradient_rel_l2_split_vs_strict_fp32": {"original_weights": rel(g_s, g_f), "two_plane_representable_weights": rel(g_sr, g_fr)}}, f, indent=1)
    except OSError:
        pass


# ---------------------------------------------------------------------------------------------
# precision equivalence of the split-operand convolution
# ---------------------------------------------------------------------------------------------
# REGRESSION ENVELOPE (test_split_conv_error_per_decile_stays_in_the_measured_envelope; the accuracy claim itself -- relative L2 ratio <= 1.0,
# no allowance -- is test_split_conv_relative_l2_not_worse_than_fp32_mfma):
# worst ratio over the ten |reference| deciles of (rms error of the split kernel) / (rms error of the strict fp32-MFMA kernel), and
# of the max errors, MEASURED on MI355X with tools/split_precision_ranges.py (profiles/r03_split_precision_ranges.json); the
# test allows these + 10 %.  Columns: rms fp16x3, rms bf16x6, max fp16x3, max bf16x6.  "mixed_r": the left half of every
# image row scaled by r -- the within-tensor dynamic range; 2^-19 of the tensor maximum is where the fp16 lo plane starts to
# underflow: no cliff shows in the forward convolution, the worst case of both split kernels is the
# 1e-5 range (1.4x / 1.6x the fp32-MFMA kernel's error in ONE decile, relative L2 still 0.4x / 0.8x).
SPLIT_RATIOS = {     # per case: worst decile of (rms fp16x3, rms bf16x6, max fp16x3, max bf16x6) / the fp32-MFMA kernel's; the larger of the values measured
                    # for k_conv5x5_sb<2, 2> (profiles/r03_split_precision_ranges.json) and for k_conv5x5_dx (profiles/r04_...: another summation order)
    "normal":      (0.46, 0.86, 0.52, 1.02),
    "heavy":       (0.47, 0.83, 0.89, 1.08),
    "mixed_1e-3":  (0.69, 1.00, 0.65, 1.47),
    "mixed_1e-5":  (1.38, 1.60, 1.39, 1.38),
    "mixed_2^-18": (0.44, 0.85, 0.49, 1.22),
    "mixed_2^-19": (0.45, 0.85, 0.44, 1.22),
    "mixed_2^-20": (0.46, 0.86, 0.49, 1.14),
    "mixed_2^-22": (0.45, 0.86, 0.61, 1.03),
}


def _split_conv_cases():
    """(name, x) inputs: normal, heavy-tailed, and within-tensor dynamic ranges from 1e-3 down to 2^-22 (the left half of every image row
    scaled by r; 2^-19 of the tensor maximum is where the fp16 lo plane starts to underflow)"""
    gen = torch.Generator().manual_seed(5)
    B, Y, X = 2, 64, 64
    cases = {"normal": torch.randn(B, Y, X, 32, generator=gen, dtype=torch.float32),
             "heavy": torch.randn(B, Y, X, 32, generator=gen, dtype=torch.float32) * torch.exp(2.0 * torch.randn(B, Y, X, 32, generator=gen, dtype=torch.float32))}
    for name, ratio in (("mixed_1e-3", 1e-3), ("mixed_1e-5", 1e-5), ("mixed_2^-18", 2.0 ** -18), ("mixed_2^-19", 2.0 ** -19),
                        ("mixed_2^-20", 2.0 ** -20), ("mixed_2^-22", 2.0 ** -22)):
        m = torch.randn(B, Y, X, 32, generator=gen, dtype=torch.float32)
        m[:, :, :32] *= ratio
        cases[name] = m
    w = (torch.randn(5, 5, 32, 32, generator=gen, dtype=torch.float32) * 0.05).to(DEV)
    return cases, w


def _split_conv_outputs(x, w):
    """(float64 reference, fp16x3 split kernel, bf16x6 split kernel, strict fp32-MFMA kernel) on the same input"""
    from sol_amd import _lib
    bias = torch.zeros(32, dtype=torch.float32, device=DEV)
    packed = ops._pack(w, 32, 32, ops.CONV_FWD)
    saved = _lib.get_option("conv_precision")
    try:
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), None, padding=2).permute(0, 2, 3, 1)
        _lib.set_option("conv_precision", 0)
        y_h = ops.conv5x5_scaled_raw(x, packed, bias, None, None, 32, ops.EPI_NONE, 0.3, ops.absmax_slots(x))   # fp16 x3
        y_b = ops.conv5x5_raw(x, packed, bias, None, None, 32, ops.EPI_NONE, 0.3)                               # bf16 x6
        _lib.set_option("conv_precision", 2)
        y_f = ops.conv5x5_raw(x, packed, bias, None, None, 32, ops.EPI_NONE, 0.3)                               # fp32 MFMA
    finally:
        _lib.set_option("conv_precision", saved)
    assert not torch.equal(y_f, y_h) and not torch.equal(y_f, y_b)          # three different kernels did run
    return ref, y_h, y_b, y_f


def test_split_conv_relative_l2_not_worse_than_fp32_mfma():
    """THE claim behind the bench line's dtype: in relative L2 against a float64 convolution -- the parity criterion of the north star --
    the fp16x3 split kernel (22-bit operand splits) and the bf16x6 one (24-bit) are AT LEAST as accurate as the strict fp32-MFMA kernel, ratio
    <= 1.0 with no allowance, in every case: normal and heavy-tailed inputs, within-tensor dynamic ranges down to 2^-22 of the tensor
    maximum, on the whole tensor AND on the small half alone (pixels that only see the scaled-down inputs)."""
    cases, w = _split_conv_cases()
    for name, x in cases.items():
        ref, y_h, y_b, y_f = _split_conv_outputs(x.float().to(DEV), w)
        small = (slice(None), slice(None), slice(2, 30))
        for what, sl in (("whole tensor", (slice(None),)), ("small half", small)):
            rf = rel(y_f[sl], ref[sl])
            assert rel(y_h[sl], ref[sl]) <= rf, (name, what, "fp16x3", rel(y_h[sl], ref[sl]) / rf)
            assert rel(y_b[sl], ref[sl]) <= rf, (name, what, "bf16x6", rel(y_b[sl], ref[sl]) / rf)


def test_split_conv_error_per_decile_stays_in_the_measured_envelope():
    """A finer-grained REGRESSION envelope, not an accuracy claim: rms and max error per decile of |reference| of both split kernels divided
    by the strict fp32-MFMA kernel's, bounded by what was measured on MI355X (SPLIT_RATIOS, + 10 %).  In one case (mixed_1e-5) one decile of
    the split kernels carries 1.4x / 1.6x the fp32-MFMA kernel's error -- the bf16x6 kernel, whose operands are EXACT 24-bit splits, more
    than the 22-bit fp16x3 one: that is fp32 accumulation order, not operand width (relative L2 of the same case: 0.4x / 0.8x, asserted
    with no allowance by test_split_conv_relative_l2_not_worse_than_fp32_mfma)."""
    cases, w = _split_conv_cases()
    for name, x in cases.items():
        ref, y_h, y_b, y_f = _split_conv_outputs(x.float().to(DEV), w)
        order = ref.abs().reshape(-1).argsort()
        n = order.numel()
        worst = {"rms_fp16x3": 0.0, "rms_bf16x6": 0.0, "max_fp16x3": 0.0, "max_bf16x6": 0.0}
        for q in range(10):
            idx = order[q * n // 10:(q + 1) * n // 10]
            r = ref.reshape(-1)[idx]
            e = {k: (v.double().reshape(-1)[idx] - r) for k, v in (("fp16x3", y_h), ("bf16x6", y_b), ("fp32", y_f))}
            rms = {k: float(v.pow(2).mean().sqrt()) for k, v in e.items()}
            mx = {k: float(v.abs().max()) for k, v in e.items()}
            for k in ("fp16x3", "bf16x6"):
                worst["rms_" + k] = max(worst["rms_" + k], rms[k] / rms["fp32"])
                worst["max_" + k] = max(worst["max_" + k], mx[k] / mx["fp32"])
        bound = SPLIT_RATIOS[name]
        got = (worst["rms_fp16x3"], worst["rms_bf16x6"], worst["max_fp16x3"], worst["max_bf16x6"])
        assert all(g <= 1.10 * b + 0.005 for g, b in zip(got, bound)), (name, got, bound)
        # the 22-bit kernel is never the worse of the two split kernels by more than the envelope's own spread: operand width is not what shows here
        assert worst["rms_fp16x3"] <= max(1.0, 1.10 * worst["rms_bf16x6"] + 0.005), (name, got)


def test_options_and_launch_profiler():
    """sol_set_option / sol_get_option reject unknown names and out-of-range values; the launch profiler sees the kernels of
    an eager training step with per-launch HIP events."""
    from sol_amd import _lib
    lib = sol_amd.load()
    assert _lib.get_option("conv_precision") in (0, 1, 2)
    with pytest.raises(sol_amd.SolError):
        _lib.set_option("no_such_option", 1)
    with pytest.raises(sol_amd.SolError):
        _lib.set_option("conv_precision", 7)
    B, Y, X, ms = 2, 128, 64, 2
    g, d, vy, vx, re, gts, params, std_v, loss = _oracle_problem(B, Y, X, ms)
    net, tr = _trainer_from(params, g, B, Y, X, ms, std_v)
    args = (f32(d), f32(vy), f32(vx), f32(re), f32(torch.stack([s[1] for s in gts])), f32(torch.stack([s[2] for s in gts])))
    with _lib.profile() as p:
        hl = tr.fwd_bwd(*args, eager=True)
    assert abs(float(hl) - float(loss)) < 1e-5 * abs(float(loss))
    names = {k.strip("()"): v for k, v in p.kernels.items()}
    # ten 32->32 layers, forward + backward-data, per unrolled step (B * Y = 256 rows: the one-row-per-workgroup form of the dx kernel)
    thin = ("k_conv5x5_dx<1, 1>", "k_conv5x5_dx<3, 1>")
    assert sum(v[0] for k, v in names.items() if k.startswith("k_conv5x5_dx") and k not in thin) == ms * (10 + 10)
    # ... and the exact-fp32 VALU kernel for the 32 -> 2 output layer (correction mode) and the 32 -> 3 data gradient of the first layer
    assert sum(v[0] for k, v in names.items() if k.startswith("k_conv5x5_thin32")) == ms + (ms - 1), names
    assert not any(k in thin for k in names), names
    assert all(c > 0 and t > 0 for c, t in names.values())
    assert lib.sol_version() == _lib.ABI_VERSION


def test_persistent_cnn_chain_equals_per_layer_launches():
    """Option cnn_persistent: the ten 32->32 layers of every CNN pass (forward and backward-data) as ONE persistent launch
    with neighbour-flag halo exchange and per-row fp16 scales, against the per-layer launches and the float64 oracle:
    training step (loss, gradient, final state) at 128x64 and the no-grad roll-out."""
    from sol_amd import _lib
    B, Y, X, ms = 2, 128, 64, 2
    g, d, vy, vx, re, gts, params, std_v, loss = _oracle_problem(B, Y, X, ms)
    gref = torch.cat([p.grad.reshape(-1) for p in params])
    args = (f32(d), f32(vy), f32(vx), f32(re), f32(torch.stack([s[1] for s in gts])), f32(torch.stack([s[2] for s in gts])))
    saved = _lib.get_option("cnn_persistent")
    out = {}
    try:
        for mode in (0, 1):
            _lib.set_option("cnn_persistent", mode)
            net, tr = _trainer_from(params, g, B, Y, X, ms, std_v)
            hl = tr.fwd_bwd(*args, want_final=True)
            with _lib.profile() as p:
                tr.fwd_bwd(*args, want_final=True, eager=True)
            names = {k.strip("()") for k in p.kernels}
            thin = ("k_conv5x5_dx<1, 1>", "k_conv5x5_dx<3, 1>")         # the 32 -> 2 / 32 -> 3 layers are per-layer launches in both modes
            assert ("k_cnn_chain" in names) == (mode == 1) and any(n.startswith("k_conv5x5_dx") and n not in thin for n in names) == (mode == 0)
            assert abs(float(hl) - float(loss)) < 1e-5 * abs(float(loss))
            assert rel(tr.grads, gref) < TOL_GRAD
            ro = sol_amd.SolRollout(net, tr.masks, B, Y, X, g.dx, std_v, o.STD_RE)
            rd, ry, rx = (t.clone() for t in args[:3])
            ro.run(rd, ry, rx, args[3], 5)
            out[mode] = (tr.grads.clone(), [t.clone() for t in tr.final], ry, rx)
    finally:
        _lib.set_option("cnn_persistent", saved)
    assert rel(out[1][0], out[0][0]) < 2e-6
    for a, b in zip(out[1][1], out[0][1]):
        assert rel(a, b) < 2e-6
    assert rel(out[1][2], out[0][2]) < 2e-6 and rel(out[1][3], out[0][3]) < 2e-6


# ---------------------------------------------------------------------------------------------
# data parallel: two ranks, one device
# ---------------------------------------------------------------------------------------------
def test_data_parallel_two_ranks_one_gpu(tmp_path):
    """SolTrainer with world_size 2 (both ranks on cuda:0, gloo transport: RCCL refuses two ranks per device) against
    the single-process large batch: the all-reduced gradient equals the global-batch gradient, the loss is the global loss
    and both replicas hold bit-identical weights after two Adam steps."""
    import subprocess
    import sys
    import socket
    import dp_worker
    Bg, Y, X, ms = 4, 64, 32, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    prefix = str(tmp_path / "dp")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "dp_worker.py"), prefix, str(Bg), str(Y), str(X), str(ms)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    r0, r1 = (np.load(prefix + "_rank%d.npz" % r) for r in range(2))
    assert np.array_equal(r0["params"], r1["params"])               # replicas stay bit-identical
    assert np.array_equal(r0["grads0"], r1["grads0"]) and np.array_equal(r0["losses"], r1["losses"])
    losses, grads0, params = dp_worker.run(Bg, Y, X, ms, 0, 1)      # single process, global batch
    assert np.allclose(r0["losses"], losses, rtol=2e-5)
    assert rel(torch.as_tensor(r0["grads0"]), torch.as_tensor(grads0)) < 2e-5
    assert rel(torch.as_tensor(r0["params"]), torch.as_tensor(params)) < 1e-5


def test_library_rccl_communicator_single_rank():
    """sol_comm_unique_id / sol_comm_init / sol_allreduce_grads / sol_comm_destroy (the library's own RCCL communicator)
    with one rank: the plumbing the N-GPU path uses, as far as a 1-GPU box can exercise it (SUM over one rank = identity)."""
    from sol_amd.dist import SolComm
    comm = SolComm()
    assert comm.world == 1 and comm.rank == 0
    g = torch.randn(sol_amd.model_mars_moon(cin=3, cout=2, seed=0).n_params, device=DEV, dtype=torch.float32)
    ref = g.clone()
    comm.allreduce_sum_(g)
    torch.cuda.synchronize()
    assert torch.equal(g, ref)
    comm.close()


# ---------------------------------------------------------------------------------------------
# Burgers unrolled loss + gradient (BASELINE configs[0]); --pretf scales in the fused trainer
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("noforce,ms", [(False, 1), (False, 2), (True, 2), (False, 4)])       # (False, 4): the reference's SOL-04 recipe, burgers/Makefile:75-77 `-m 4 -b 5`
def test_burgers_unrolled_loss_and_gradient_against_oracle(noforce, ms):
    """burgers/ 32x32, batch 5, dt 0.1 (karman... burgers/Makefile:71 `-l 32 --dt 0.1 -b 5 -m 1`): the unrolled graph of
    burgers_train.py:379-437 composed from the reference-shaped surface (BurgersTest.step_with_f, to_feature, model,
    to_staggered) on the HIP ops with torch autograd, against the float64 oracle: loss and the full weight gradient."""
    from sol_amd.burgers import to_feature, to_feature_noforce
    B, Y, X, dt = 5, 32, 32, 0.1
    gen = torch.Generator().manual_seed(7)
    sm = lambda *shape: o._smooth(torch.randn(*shape, generator=gen, dtype=torch.float64))
    vy, vx = 0.3 * sm(B, Y + 1, X), 0.3 * sm(B, Y, X + 1)
    fy = [0.15 * sm(B, Y + 1, X) for _ in range(ms)]
    fx = [0.15 * sm(B, Y, X + 1) for _ in range(ms)]
    gy = [0.3 * sm(B, Y + 1, X) for _ in range(ms)]
    gx = [0.3 * sm(B, Y, X + 1) for _ in range(ms)]
    std_v, std_f = (0.21, 0.19), (0.09, 0.11)
    cin = 2 if noforce else 4
    params = [p.clone().requires_grad_(True) for p in o.init_params(0, cin=cin)]
    loss = o.burgers_unrolled_loss(params, vy, vx, fy, fx, gy, gx, std_v, std_f, dt, noforce=noforce)
    loss.backward()
    gref = torch.cat([p.grad.reshape(-1) for p in params])

    dom = sol_amd.Domain([Y, X], box=sol_amd.box([32, 32]), boundaries=sol_amd.PERIODIC)
    sim = sol_amd.BurgersTest()
    net = sol_amd.model_mars_moon(cin=cin, cout=2, seed=0)
    net.set_weights([p.detach().numpy() for p in params])
    st = sol_amd.BurgersVelocitySMAC(dom, velocity=f32(o.staggered_tensor(vy, vx)), batch_size=B)
    sv = torch.tensor(std_v, device=DEV, dtype=torch.float32)
    sin = sv if noforce else torch.cat([sv, torch.tensor(std_f, device=DEV, dtype=torch.float32)])
    losses = []
    for k in range(ms):
        fr = sol_amd.BurgersVelocitySMAC(dom, velocity=f32(o.staggered_tensor(fy[k], fx[k])), batch_size=B)
        st = sim.step(st, dt=dt) if noforce else sim.step_with_f(st, fr, dt=dt)
        feat = to_feature_noforce([st]) if noforce else to_feature([st], [fr])
        corr = sol_amd.to_staggered(net(feat / sin) * sv, dom.box)
        st = st.copied_with(velocity=st.velocity + corr)
        diff = (f32(o.staggered_tensor(gy[k], gx[k])) - st.velocity.staggered_tensor()) / sv
        losses.append(0.5 * (diff * diff).sum())
    total = torch.stack(losses).sum() / ms
    total.backward()
    assert abs(float(total) - float(loss)) < 1e-5 * abs(float(loss))
    assert rel(net.params.grad, gref) < TOL_GRAD


def test_burgers_data_generation_scripts(tmp_path):
    """scripts/burgers.py (flags of /root/reference/burgers/burgers.py:35-49): a 64x64 run with the SinForces model, then the
    reference's hires -> lores chain (Makefile:35-41: --initvH / --loadfH, -d 2) at 32x32, then burgers_train.py on it."""
    import importlib.util
    import sys
    from sol_amd import scene
    sdir = os.path.join(os.path.dirname(os.path.abspath(sol_amd.__file__)), "scripts")
    sys.path.insert(0, sdir)

    def load(name):
        spec = importlib.util.spec_from_file_location("sol_script_" + name, os.path.join(sdir, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    # the reference's own hi-res setting (burgers/Makefile:19-23: -r 128 -l 32 --dt 0.1): forward-only large-grid step
    h128 = load("burgers").main(["-o", str(tmp_path / "hires128"), "-r", "128", "-l", "32", "--dt", "0.1", "--skipsteps", "2", "-t", "3", "--seed", "0"])
    v128 = scene.read_zipped_array(h128 + "/velo_000002.npz")
    assert v128.shape == (1, 129, 129, 2) and np.isfinite(v128).all() and np.abs(v128).max() > 0
    hi = load("burgers").main(["-o", str(tmp_path / "hires"), "-r", "64", "-l", "32", "--dt", "0.1", "--skipsteps", "3", "-t", "10", "--seed", "1"])
    v0 = scene.read_zipped_array(hi + "/velo_000000.npz")
    f5 = scene.read_zipped_array(hi + "/forc_000005.npz")
    assert v0.shape == (1, 65, 65, 2) and f5.shape == (1, 65, 65, 2) and np.isfinite(v0).all() and np.abs(f5).max() > 0
    assert len(glob_files(hi, "velo_")) == 10 and len(glob_files(hi, "forc_")) == 10
    for s in (0, 1):
        lo = load("burgers").main(["-o", str(tmp_path / "lores"), "-r", "32", "-l", "32", "--dt", "0.1", "--skipsteps", "0", "-t", "10", "-d", "2", "--seed", str(s),
                                   "--initvH", hi + "/velo_000000.npz", "--loadfH", hi + "/forc_0*.npz"])
    vl = scene.read_zipped_array(lo + "/velo_000009.npz")
    assert vl.shape == (1, 33, 33, 2) and np.isfinite(vl).all()
    with pytest.raises(SystemExit):
        load("burgers").main(["-r", "2048"])
    loss = load("burgers_train").main(["--train", str(tmp_path / "lores"), "-s", "1", "-n", "2", "-b", "2", "-t", "8", "-m", "2", "-e", "1", "--dt", "0.1",
                                       "--lr", "1e-4", "--tf", str(tmp_path / "tf"), "--seed", "0"])
    assert loss is not None and np.isfinite(loss)


def glob_files(path, prefix):
    return [f for f in os.listdir(path) if f.startswith(prefix)]


def test_fused_trainer_with_pretf_scales_against_oracle():
    """--pretf (karman_train.py:351-355,416-421): input / output normalisation of a pre-trained model ('in.std', 'out.std')
    differ from the loss scale ('std'): fused trainer vs oracle at 64x32, and the roll-out with the same scales."""
    B, Y, X, ms = 2, 64, 32, 2
    g = o.geometry(Y, X)
    d, vy, vx = o.synthetic_state(B, Y, X, 1234)
    re = torch.tensor(o.RE_TRAIN[:B], dtype=torch.float64)
    gts = [o.synthetic_state(B, Y, X, 4321 + i, project_it=False) for i in range(ms)]
    params = [p.clone().requires_grad_(True) for p in o.init_params(0)]
    std_v, in_std, out_std = (0.2, 0.25), (0.31, 0.17), (0.05, 0.08)
    loss = o.unrolled_loss(params, d, vy, vx, re, [s[1] for s in gts], [s[2] for s in gts], g, std_v, o.STD_RE,
                           in_std_v=in_std, out_std_v=out_std)
    loss.backward()
    gref = torch.cat([p.grad.reshape(-1) for p in params])
    net, tr = _trainer_from(params, g, B, Y, X, ms, std_v, in_std_v=in_std, out_std_v=out_std)
    hl = tr.fwd_bwd(f32(d), f32(vy), f32(vx), f32(re), f32(torch.stack([s[1] for s in gts])), f32(torch.stack([s[2] for s in gts])))
    assert abs(float(hl) - float(loss)) < 1e-5 * abs(float(loss))
    assert rel(tr.grads, gref) < TOL_GRAD
    ro = sol_amd.SolRollout(net, tr.masks, B, Y, X, g.dx, std_v, o.STD_RE, in_std_v=in_std, out_std_v=out_std)
    hd, hy, hx = f32(d), f32(vy), f32(vx)
    ro.run(hd, hy, hx, f32(re), 3)
    with torch.no_grad():
        rd, ry, rx = d, vy, vx
        for _ in range(3):
            rd, ry, rx = o.karman_step(rd, ry, rx, re, g)
            cy, cx = o.correction([p.detach() for p in params], ry, rx, re, std_v, o.STD_RE, in_std, out_std)
            ry, rx = ry + cy, rx + cx
    assert rel(hy, ry) < TOL_FIELD and rel(hx, rx) < TOL_FIELD


def test_torch_library_ops_are_registered_and_differentiable():
    """torch.ops.sol.* (SURVEY 8b2 / north star "PyTorch-ROCm custom ops with a hand-written backward"): the registered
    ops run the same C-ABI entry points as the ctypes path and carry the hand-written adjoints."""
    import sol_amd.torch_ops  # noqa: F401  (registers the library)
    B, Y, X = 2, 16, 8
    g, mk = masks_for(Y, X)
    d, vy, vx = (f32(t) for t in o.synthetic_state(B, Y, X, 5))
    re = f32(torch.tensor(o.RE_TRAIN[:B]))
    cfg = ops.karman_cfg(B, Y, X, g.dx, masks=mk)
    h = sol_amd.torch_ops.register_scene(cfg, mk)
    vy1 = vy.clone().requires_grad_(True)
    vx1 = vx.clone().requires_grad_(True)
    d2, py, px = torch.ops.sol.karman_step(d, vy1, vx1, re, h)
    (py.sum() + 2 * px.sum()).backward()
    vy2 = vy.clone().requires_grad_(True)
    vx2 = vx.clone().requires_grad_(True)
    e2, qy, qx = ops.karman_step(d, vy2, vx2, re, cfg, mk)
    (qy.sum() + 2 * qx.sum()).backward()
    assert torch.equal(py, qy) and torch.equal(px, qx) and torch.equal(d2, e2)
    assert torch.equal(vy1.grad, vy2.grad) and torch.equal(vx1.grad, vx2.grad)
    x = torch.randn(2, 16, 8, 32, device=DEV, dtype=torch.float32, requires_grad=True)
    w = (torch.randn(5, 5, 32, 32, device=DEV, dtype=torch.float32) * 0.05).requires_grad_(True)
    b = torch.zeros(32, device=DEV, dtype=torch.float32, requires_grad=True)
    y1 = torch.ops.sol.conv5x5(x, w, b, None, True, 0.3)
    y1.square().sum().backward()
    gx1, gw1 = x.grad.clone(), w.grad.clone()
    x.grad = w.grad = b.grad = None
    y2 = ops.conv5x5(x, w, b, None, True, 0.3)
    y2.square().sum().backward()
    assert torch.equal(y1, y2) and torch.equal(gx1, x.grad) and torch.equal(gw1, w.grad)
    assert torch.library  # the ops live in the dispatcher: torch.ops.sol.karman_step / conv5x5 / burgers_step / adam_tf_step
    for name in ("karman_step", "conv5x5", "burgers_step", "adam_tf_step"):
        assert hasattr(torch.ops.sol, name)


@pytest.mark.parametrize("noforce,ms", [(False, 2), (True, 3)])
def test_burgers_fused_trainer_graph_equals_eager_and_oracle(noforce, ms):
    """BurgersTrainer: the unrolled Burgers training step (burgers_train.py:379-437) captured into ONE hipGraph over static
    buffers.  Replays must reproduce the eager composition (same kernels: loss and gradient to 1e-6), follow NEW batch data
    copied into the static buffers, agree with the float64 oracle, and a TF-Adam step through the trainer must equal the
    eager one bit for bit in the weights."""
    B, Y, X, dt = 5, 32, 32, 0.1
    gen = torch.Generator().manual_seed(11)
    sm = lambda *shape: o._smooth(torch.randn(*shape, generator=gen, dtype=torch.float64))
    std_v, std_f = (0.21, 0.19), (0.09, 0.11)
    cin = 2 if noforce else 4
    dom = sol_amd.Domain([Y, X], box=sol_amd.box([32, 32]), boundaries=sol_amd.PERIODIC)

    def batch():
        vy = [0.3 * sm(B, Y + 1, X) for _ in range(ms + 1)]
        vx = [0.3 * sm(B, Y, X + 1) for _ in range(ms + 1)]
        fy = [0.15 * sm(B, Y + 1, X) for _ in range(ms)]
        fx = [0.15 * sm(B, Y, X + 1) for _ in range(ms)]
        velo = torch.stack([o.staggered_tensor(a, b) for a, b in zip(vy, vx)])
        forc = torch.stack([o.staggered_tensor(a, b) for a, b in zip(fy, fx)])
        return vy, vx, fy, fx, velo, forc

    params = [p.clone().requires_grad_(True) for p in o.init_params(3, cin=cin)]
    net_g = sol_amd.model_mars_moon(cin=cin, cout=2, seed=0)
    net_g.set_weights([p.detach().numpy() for p in params])
    net_e = net_g.clone()
    tg = sol_amd.BurgersTrainer(net_g, dom, B, ms, dt, std_v, std_f, noforce=noforce, use_graph=True)
    te = sol_amd.BurgersTrainer(net_e, dom, B, ms, dt, std_v, std_f, noforce=noforce, use_graph=False)
    for it in range(3):                       # three different batches through the SAME captured graph
        vy, vx, fy, fx, velo, forc = batch()
        lg = float(tg.fwd_bwd(velo, forc))
        le = float(te.fwd_bwd(velo, forc))
        assert tg._graph is not None and te._graph is None
        assert abs(lg - le) <= 1e-6 * abs(le)
        assert rel(net_g.params.grad, net_e.params.grad) < 1e-6
        if it == 2:                           # float64 oracle on the last batch
            loss = o.burgers_unrolled_loss(params, vy[0], vx[0], fy, fx, vy[1:], vx[1:], std_v, std_f, dt, noforce=noforce)
            loss.backward()
            gref = torch.cat([p.grad.reshape(-1) for p in params])
            assert abs(lg - float(loss)) < 1e-5 * abs(float(loss))
            assert rel(net_g.params.grad, gref) < TOL_GRAD
    vy, vx, fy, fx, velo, forc = batch()
    tg.train_step(velo, forc, lr=1e-3)
    te.train_step(velo, forc, lr=1e-3)
    assert tg.opt.t == 1 and rel(net_g.params.detach(), net_e.params.detach()) < 1e-7
    assert float((net_g.params.detach() - f32(torch.cat([p.detach().reshape(-1) for p in params]))).abs().max()) > 0





# ---------------------------------------------------------------------------------------------
# Burgers roll-out (burgers_apply.py:129-151)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("noforce,ms,model", [(False, 1, "mars_moon"), (False, 4, "mars_moon"), (True, 2, "mars_moon"), (False, 2, "mercury")])
def test_burgers_trainer_manual_schedule_equals_autograd_composition(noforce, ms, model):
    """BurgersTrainer's hand-written schedule over the C ABI (round 6: burgers.BurgersTrainer._schedule_step -- burgers_train.py:379-437
    differentiated by hand; NON `-m 1` and SOL-04 `-m 4` of burgers/Makefile:69-77) against the torch-autograd composition
    (schedule="autograd"): loss and flat gradient, eager and replayed, after new batch data and after a TF-Adam step; far fewer kernel
    nodes in the captured graph."""
    from sol_amd import _lib
    B, Y, X, dt = 5, 32, 32, 0.1
    gen = torch.Generator().manual_seed(13)
    dom = sol_amd.Domain([Y, X], box=sol_amd.box([32, 32]), boundaries=sol_amd.PERIODIC)
    mk_net = sol_amd.model_mercury if model == "mercury" else sol_amd.model_mars_moon
    cin = 2 if noforce else 4

    def batch():
        velo = (0.3 * torch.randn(ms + 1, B, Y + 1, X + 1, 2, generator=gen, dtype=torch.float32)).to(DEV)
        forc = (0.1 * torch.randn(ms, B, Y + 1, X + 1, 2, generator=gen, dtype=torch.float32)).to(DEV)
        return velo, forc

    trs = {}
    for sched in ("manual", "autograd"):
        for graph in (False, True):
            net = mk_net(cin=cin, cout=2, seed=2, device=DEV)
            with torch.no_grad():
                net.params.add_(0.01 * torch.randn(net.params.shape, generator=torch.Generator().manual_seed(9)).to(DEV))
            trs[(sched, graph)] = sol_amd.BurgersTrainer(net, dom, B, ms, dt, (0.21, 0.19), (0.09, 0.11), noforce=noforce, use_graph=graph, schedule=sched)
    for it in range(3):
        velo, forc = batch()
        out = {}
        for key, tr in trs.items():
            loss = tr.train_step(velo, forc, 1e-4) if it == 2 else tr.fwd_bwd(velo, forc)       # last batch: one TF-Adam step in every trainer
            torch.cuda.synchronize()
            out[key] = (float(loss), tr.net.params.grad.detach().clone(), tr.net.params.detach().clone())
        ref = out[("autograd", False)]
        for key, got in out.items():
            assert abs(got[0] - ref[0]) < 2e-6 * abs(ref[0]), (it, key, got[0], ref[0])
            # the forward passes of the two forms are bit-identical (same launches on the same operands), so every LeakyReLU mask is too;
            # what differs is the summation order of the weight gradients (accumulated over the steps vs summed by autograd)
            assert rel(got[1], ref[1]) < 1e-5, (it, key, rel(got[1], ref[1]))
            # the first Adam update is lr * sign(g) per weight: gradient elements at rounding level move by 2 lr between the two summation orders
            assert rel(got[2], ref[2]) < 1e-4, (it, key)
        assert torch.equal(out[("manual", False)][1], out[("manual", True)][1]) and torch.equal(out[("manual", False)][2], out[("manual", True)][2])
    nodes = {s_: _lib.graph_census(trs[(s_, True)]._graph.raw_cuda_graph()) for s_ in ("manual", "autograd")}
    assert all(set(c) <= {"kernel", "empty"} for c in nodes.values()), nodes
    assert nodes["manual"]["kernel"] < 0.75 * nodes["autograd"]["kernel"], nodes


@pytest.mark.parametrize("noforce,use_graph", [(False, True), (False, False), (True, True)])
def test_burgers_rollout_against_oracle(noforce, use_graph):
    """BurgersRollout: 10 corrected steps (solver step with the PREVIOUS force frame, network input with the CURRENT one, as
    burgers_apply.py:131-134 does) against the float64 oracle; the replayed hipGraph equals the eager composition."""
    B, Y, X, dt, nsteps = 2, 32, 32, 0.1, 10
    gen = torch.Generator().manual_seed(21)
    sm = lambda *shape: o._smooth(torch.randn(*shape, generator=gen, dtype=torch.float64))
    std_v, std_f = (0.21, 0.19), (0.09, 0.11)
    cin = 2 if noforce else 4
    dom = sol_amd.Domain([Y, X], box=sol_amd.box([32, 32]), boundaries=sol_amd.PERIODIC)
    vy, vx = 0.3 * sm(B, Y + 1, X), 0.3 * sm(B, Y, X + 1)
    fy = [0.15 * sm(B, Y + 1, X) for _ in range(nsteps + 1)]
    fx = [0.15 * sm(B, Y, X + 1) for _ in range(nsteps + 1)]
    params = [p.clone() for p in o.init_params(3, cin=cin)]
    params[22] = params[22] * 0.1
    net = sol_amd.model_mars_moon(cin=cin, cout=2, seed=0)
    net.set_weights([p.numpy() for p in params])
    ro = sol_amd.BurgersRollout(net, dom, B, dt, std_v, std_f, noforce=noforce, use_graph=use_graph)
    ro.reset(o.staggered_tensor(vy, vx))
    sv = torch.tensor(std_v)
    ry, rx = vy, vx
    for i in range(1, nsteps + 1):
        ro.step(None if noforce else o.staggered_tensor(fy[i - 1], fx[i - 1]), None if noforce else o.staggered_tensor(fy[i], fx[i]))
        with torch.no_grad():
            ry, rx = o.burgers_step(ry, rx, dt, 0.1, None if noforce else fy[i - 1], None if noforce else fx[i - 1])
            feat = o.staggered_tensor(ry, rx)[:, :-1, :-1, :] / sv
            if not noforce:
                feat = torch.cat([feat, o.staggered_tensor(fy[i], fx[i])[:, :-1, :-1, :] / torch.tensor(std_f)], dim=-1)
            cy, cx = o.to_staggered(o.mars_moon(params, feat) * sv)
            ry, rx = ry + cy, rx + cx
    torch.cuda.synchronize()
    assert (ro._graph is not None) == use_graph
    assert rel(ro.vel, o.staggered_tensor(ry, rx)) < TOL_FIELD, rel(ro.vel, o.staggered_tensor(ry, rx))
    assert rel(ro.corr, o.staggered_tensor(cy, cx)) < 1e-4 and float(ro.corr.abs().max()) > 0


def test_burgers_apply_script(tmp_path):
    """scripts/burgers_apply.py (flags of burgers_apply.py:22-33): velTf / corTf frames of a forced roll-out from force files."""
    import importlib.util
    import pickle
    import sys
    from sol_amd import scene
    sdir = os.path.join(os.path.dirname(os.path.abspath(sol_amd.__file__)), "scripts")
    sys.path.insert(0, sdir)
    spec = importlib.util.spec_from_file_location("sol_script_burgers_apply", os.path.join(sdir, "burgers_apply.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res, steps = 32, 6
    gen = torch.Generator().manual_seed(5)
    fdir = scene.scene_create(str(tmp_path / "forces"))
    for i in range(steps):
        fr = 0.15 * sol_amd.synthetic._smooth(torch.randn(1, res + 1, res + 1, generator=gen, dtype=torch.float64)).unsqueeze(-1).repeat(1, 1, 1, 2)
        scene.scene_write(fdir, [fr.numpy().astype(np.float32)], ["forc"], i)
    net = sol_amd.model_mars_moon(cin=4, cout=2, seed=1)
    with torch.no_grad():
        net.tensors()[22].mul_(0.1)
    net.save(str(tmp_path / "model.pt"))
    with open(tmp_path / "dataStats.pickle", "wb") as f:
        pickle.dump({"std": [(0.3, 0.3), (0.1, 0.1)]}, f)
    out = mod.main(["-r", str(res), "-l", "32", "-t", str(steps), "--dt", "0.1", "-s", "1", "--loadfH", fdir + "/forc_0*.npz",
                    "-o", str(tmp_path / "run"), "--stats", str(tmp_path / "dataStats.pickle"), "--model", str(tmp_path / "model.pt")])
    v = scene.read_zipped_array(out + "/velTf_%06d.npz" % (steps - 1))
    c = scene.read_zipped_array(out + "/corTf_%06d.npz" % (steps - 1))
    assert v.shape == (1, res + 1, res + 1, 2) and c.shape == v.shape and np.isfinite(v).all() and np.abs(c).max() > 0
    v0 = scene.read_zipped_array(out + "/velTf_000000.npz")
    assert np.abs(v - v0).max() > 0
    with pytest.raises(SystemExit):
        mod.main(["-r", str(res), "-t", "3", "-o", str(tmp_path / "run2"), "--stats", str(tmp_path / "dataStats.pickle"),
                  "--model", str(tmp_path / "model.pt")])          # forces required unless --noforce


# ---------------------------------------------------------------------------------------------
# real-data training loop at graph speed
# ---------------------------------------------------------------------------------------------
@pytest.mark.timeout(900)
def test_real_data_training_loop_runs_at_graph_speed(tmp_path):
    """scripts/karman_train.py on a generated 128 x 64 set (6 simulations, SOL-32, the BASELINE configs[2] shape): with the
    set resident on the device and the loss read back one step late, a training step of the script costs what a replay of
    the captured graph on persistent synthetic buffers costs (bench.py's number).  --host-feed (the reference's per-step
    host assembly + pageable copy + float(loss)) is timed next to it."""
    import importlib.util
    import sys
    sdir = os.path.join(os.path.dirname(os.path.abspath(sol_amd.__file__)), "scripts")
    sys.path.insert(0, sdir)

    def load(name):
        spec = importlib.util.spec_from_file_location("sol_script_" + name, os.path.join(sdir, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    B, Y, X, ms, frames = 6, 128, 64, 32, 47
    data = str(tmp_path / "set")
    for k in range(B):
        load("karman").main(["-o", data, "-r", str(X), "-t", str(frames + 2), "-s", "1", "--re", str(o.RE_TRAIN[k])])
    kt = load("karman_train")
    args = ["--train", data, "-s", "1", "-n", str(B), "-b", str(B), "-t", str(frames), "-m", str(ms), "-e", "1", "--lr", "1e-6", "--seed", "0"]
    loss = kt.main(args + ["--tf", str(tmp_path / "tf")])
    t_res = kt.main.last_ms_per_step
    assert loss is not None and np.isfinite(loss)
    loss_h = kt.main(args + ["--tf", str(tmp_path / "tf_h"), "--host-feed"])
    t_host = kt.main.last_ms_per_step
    assert abs(loss_h - loss) <= 1e-5 * abs(loss)               # same batches, same arithmetic: the feed path changes nothing
    # the synthetic replay of the same shape
    g, mk = masks_for(Y, X)
    net = sol_amd.model_mars_moon(cin=3, cout=2, seed=0)
    tr = sol_amd.SolTrainer(net, mk, B, Y, X, ms, g.dx, (0.2, 0.2), o.STD_RE)
    d, vy, vx = (f32(t) for t in sol_amd.synthetic.state(B, Y, X, 3))
    gy, gx = (f32(t) for t in sol_amd.synthetic.frames(ms, B, Y, X, 9))
    re = f32(sol_amd.synthetic.reynolds(B))
    for _ in range(4):
        tr.train_step(d, vy, vx, re, gy, gx, lr=1e-9)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    n = frames - ms - 4
    for _ in range(n):
        tr.train_step(d, vy, vx, re, gy, gx, lr=1e-9)
    torch.cuda.synchronize()
    t_syn = (time.perf_counter() - t0) / n * 1e3
    print("real-data loop: resident %.3f ms/step, host-feed %.3f ms/step, synthetic replay %.3f ms/step" % (t_res, t_host, t_syn))
    assert t_res <= 1.10 * t_syn, (t_res, t_syn)


def test_q3_antialiased_inflow_mask_through_the_hip_step():
    """the inflow rate mask is an INPUT array of the HIP step, so the anti-aliased variant of the
    recalled GeometryMask semantics runs unchanged (fractional rates)."""
    B, Y, X = 2, 64, 32
    g = o.geometry(Y, X, inflow_antialias=True)
    mk = ops.SceneMasks(g.active, g.inflow, g.bc_mask, g.bc_mask)
    d, vy, vx = o.synthetic_state(B, Y, X, 8)
    re = torch.tensor(o.RE_TRAIN[:B], dtype=torch.float64)
    rd, ry, rx = o.karman_step(d, vy, vx, re, g)
    hd, hy, hx = ops.karman_step(f32(d), f32(vy), f32(vx), f32(re), ops.karman_cfg(B, Y, X, g.dx, masks=mk), mk)
    assert rel(hd, rd) < TOL_FIELD and rel(hy, ry) < TOL_FIELD and rel(hx, rx) < TOL_FIELD
    assert float((rd - o.karman_step(d, vy, vx, re, o.geometry(Y, X))[0]).abs().max()) > 1e-3        # the option does change the density
