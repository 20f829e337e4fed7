"""GPU parity tests of the karman-3d forward path (pytest -m gpu): csrc/karman3d.hip through the C ABI against
oracle/sol_oracle3d.py (float64) and the committed fixture tests/golden/karman3d_32x16x16.npz.
Tolerance: <= 1e-5 relative L2 on velocity / density fields (north star), fp32 kernels vs the float64 oracle."""
import os
import sys

import numpy as np
import pytest
import torch

import sol_amd
import sol_oracle3d as o
from sol_amd import karman3d as k3

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL_FIELD = 1e-5


def rel(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b), dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-300))


def f32(t):
    return torch.as_tensor(np.asarray(t), dtype=torch.float32).to(DEV).contiguous()


@pytest.fixture(scope="module")
def fixture3d(golden_dir):
    return np.load(os.path.join(golden_dir, "karman3d_32x16x16.npz"))


@pytest.fixture(scope="module")
def scene_small():
    return k3.Scene3D(32, 16, 16, device=DEV)


@pytest.mark.parametrize("tile,fused_tf", [(1, 1), (0, 1), (1, 0)])
def test_karman3d_step_against_golden(fixture3d, scene_small, tile, fused_tf):
    """tile: advection from LDS tiles / from global memory; fused_tf: sine transforms as LDS-resident plane / slab kernels /
    as batched GEMMs."""
    z = fixture3d
    B = z["d"].shape[0]
    sol_amd._lib.set_option("k3d_tile", tile)
    sol_amd._lib.set_option("k3d_fused_tf", fused_tf)
    try:
        sim = k3.Karman3DFlow(scene_small, B)
        feat = torch.zeros(B, 32, 16, 16, 4, dtype=torch.float32, device=DEV)
        fs = [1 / 0.2, 1 / 0.25, 1 / 0.3, 1 / float(z["std_re"])]
        d, vy, vx, vz = sim.step(f32(z["d"]), f32(z["vy"]), f32(z["vx"]), f32(z["vz"]), f32(z["re"]), feat_out=feat, feat_scale=fs)
        torch.cuda.synchronize()
    finally:
        sol_amd._lib.set_option("k3d_tile", 0)
        sol_amd._lib.set_option("k3d_fused_tf", 1)
    errs = [rel(a, z[k]) for a, k in ((d, "d_out"), (vy, "vy_out"), (vx, "vx_out"), (vz, "vz_out"))]
    assert max(errs) < TOL_FIELD, errs
    # fused to_feature: the three components at the low faces + Re, scaled
    ref = torch.stack([torch.as_tensor(z["vy_out"])[:, :32] * fs[0], torch.as_tensor(z["vx_out"])[:, :, :16] * fs[1],
                       torch.as_tensor(z["vz_out"])[..., :16] * fs[2],
                       torch.as_tensor(z["re"]).reshape(B, 1, 1, 1).expand(B, 32, 16, 16) * fs[3]], dim=-1)
    ferr = [rel(feat[..., c], ref[..., c]) for c in range(4)]
    assert max(ferr) < TOL_FIELD, (ferr, errs)


def test_cabi_wrappers_refuse_non_32bit_buffers(scene_small):
    sim = k3.Karman3DFlow(scene_small, 2)
    z = torch.zeros
    with pytest.raises(sol_amd.SolError):
        sim.step(z(2, 32, 16, 16), z(2, 33, 16, 16), z(2, 32, 17, 16), z(2, 32, 16, 17), torch.ones(2),
                 feat_out=z(2, 32, 16, 16, 4, dtype=torch.float64, device=DEV), feat_scale=[1, 1, 1, 1])


def test_karman3d_tile_and_global_advection_agree(fixture3d, scene_small):
    z = fixture3d
    outs = []
    for tile in (1, 0):
        sol_amd._lib.set_option("k3d_tile", tile)
        sim = k3.Karman3DFlow(scene_small, 2)
        outs.append(sim.step(f32(z["d"]), f32(z["vy"]), f32(z["vx"]), f32(z["vz"]), f32(z["re"])))
    sol_amd._lib.set_option("k3d_tile", 0)
    for a, b in zip(*outs):
        assert rel(a, b) < 1e-6           # same arithmetic, LDS vs global operands (the compiler contracts the two kernels differently)


def test_karman3d_step_variants_and_large_cfl_against_oracle():
    """dirichlet0 pressure-gradient padding, inflow before advection, dt = 2 with |v| ~ 1.3 cells per step (samples leave
    the LDS halo and take the global fallback)."""
    B, Y, X, Z = 2, 16, 8, 8
    g = o.geometry(Y, X, Z)
    d, v = o.synthetic_state(B, Y, X, Z, 5)
    v = (v[0] * 4.0, v[1] * 6.0, v[2] * 6.0)            # dx = 12.5: ~0.4 cells per unit time along y, up to ~1 across
    re = torch.tensor(o.RE_TRAIN[:B])
    sc = k3.Scene3D(Y, X, Z, device=DEV)
    for kw in (dict(grad_pad="dirichlet0", inflow_order="before"), dict()):
        with torch.no_grad():
            dr, vr = o.karman3d_step(d, v, re, g, dt=2.0, **kw)
        sim = k3.Karman3DFlow(sc, B, dt=2.0, **kw)
        out = sim.step(f32(d), f32(v[0]), f32(v[1]), f32(v[2]), f32(re))
        errs = [rel(a, b) for a, b in zip(out, (dr,) + tuple(vr))]
        assert max(errs) < TOL_FIELD, (kw, errs)


def _oracle_conv(x, w, b, res, lrelu, slope=0.3):
    y = o.conv3d_same(x, w, b)
    if res is not None:
        y = y + res
    return torch.nn.functional.leaky_relu(y, slope) if lrelu else y


@pytest.mark.parametrize("shape,cin,cout,res,lrelu", [
    ((1, 6, 16, 16), 4, 32, False, True),          # first layer (fp32 MFMA thin kernel), W | 64
    ((2, 5, 16, 16), 32, 32, True, True),          # residual block tail, two simulations
    ((1, 4, 16, 16), 32, 3, False, False),         # output layer
    ((1, 4, 64, 64), 32, 32, True, True),          # W = 64: the split fp16 / bf16 MFMA kernels (with absmax: the one-launch kernel)
    ((2, 6, 64, 64), 32, 32, False, True),         # two simulations, six planes: slice validity at both ends and across samples
    ((2, 3, 20, 64), 32, 32, True, True),          # H not a multiple of the rows per workgroup: tiles span planes and samples
    ((1, 7, 5, 64), 32, 32, False, False),         # H smaller than a workgroup's rows
    ((1, 3, 64, 64), 4, 32, False, True),
    ((1, 3, 64, 64), 32, 3, False, False),
    ((2, 5, 24, 64), 32, 3, False, False),         # thin output layer in one launch, tiles spanning planes and samples
    ((1, 4, 64, 64), 32, 4, True, True),           # 32 -> 4 (the first layer's data gradient shape) with residual and activation
])
def test_conv3d_against_oracle(shape, cin, cout, res, lrelu):
    B, D, H, W = shape
    gen = torch.Generator().manual_seed(B * 1000 + D * 100 + cin)
    x = torch.randn(B, D, H, W, cin, generator=gen, dtype=torch.float64).float().double()
    w = (torch.randn(5, 5, 5, cin, cout, generator=gen, dtype=torch.float64) / np.sqrt(125 * cin)).float().double()
    b = torch.randn(cout, generator=gen, dtype=torch.float64).float().double()
    r = torch.randn(B, D, H, W, cout, generator=gen, dtype=torch.float64).float().double() if res else None
    ref = _oracle_conv(x, w, b, r, lrelu)
    lib = sol_amd.load()
    wd = f32(w)
    packed = torch.empty(lib.sol_conv3d_packed_floats(cin, cout), dtype=torch.float32, device=DEV)
    sol_amd._lib.check(lib.sol_conv3d_pack(sol_amd._lib.stream(), sol_amd._lib.ptr(wd), cin, cout, 0, sol_amd._lib.ptr(packed)))
    xd = f32(x)
    for use_amax in ((False, True) if cin == 32 else (False,)):
        amax = sol_amd.ops.absmax_slots(xd) if use_amax else None
        ymax = torch.zeros(256, dtype=torch.int32, device=DEV)
        y = k3.conv3d(xd, packed, f32(b), f32(r) if res else None, cout, lrelu, 0.3, amax, ymax)
        torch.cuda.synchronize()
        assert rel(y, ref) < 2e-6, (use_amax, rel(y, ref))
        pub = float(ymax.max().view(torch.float32))
        assert abs(pub - float(y.abs().max())) <= 1e-6 * pub            # the centre pass publishes max|y| of the finished tensor
        if use_amax and cin == 32 and W == 64:
            # this call ran the one-launch 5x5x5 kernel (conv3d_sb.hip; cout = 3: its one-channel-tile form); the five-pass composition must agree with it
            sol_amd._lib.set_option("k3d_conv_fused", 0)
            try:
                y5 = k3.conv3d(xd, packed, f32(b), f32(r) if res else None, cout, lrelu, 0.3, amax, None)
            finally:
                sol_amd._lib.set_option("k3d_conv_fused", 1)
            assert not torch.equal(y, y5) and rel(y, y5) < 1e-6
            # ... and so must the three- and six-rows-per-workgroup forms of the one-launch kernel (default: eight rows)
            for rows in (3, 6):
                sol_amd._lib.set_option("k3d_conv_rows", rows)
                try:
                    y3 = k3.conv3d(xd, packed, f32(b), f32(r) if res else None, cout, lrelu, 0.3, amax, None)
                finally:
                    sol_amd._lib.set_option("k3d_conv_rows", 8)
                assert rel(y, y3) < 1e-6, (rows, rel(y, y3))


def test_network_and_rollout_against_golden(fixture3d, scene_small):
    import make_golden as mg
    z = fixture3d
    B = z["d"].shape[0]
    net = k3.MarsMoon3D(device=DEV)
    assert net.n_params == 1308355
    net.set_weights([p.numpy() for p in mg.k3d_params()])
    std_v = tuple(float(s) for s in z["std_v"])
    ro = k3.Karman3DRollout(net, scene_small, B, std_v, float(z["std_re"]))
    # the network alone, on the features of the fixture's solver step
    fs = ro.feat_scale
    ref_feat = torch.stack([torch.as_tensor(z["vy_out"])[:, :32] * fs[0], torch.as_tensor(z["vx_out"])[:, :, :16] * fs[1],
                            torch.as_tensor(z["vz_out"])[..., :16] * fs[2],
                            torch.as_tensor(z["re"]).reshape(B, 1, 1, 1).expand(B, 32, 16, 16) * fs[3]], dim=-1)
    ro.feat.copy_(ref_feat.to(DEV))
    out = ro.correction()
    assert rel(out, z["net_out"]) < 5e-6, rel(out, z["net_out"])
    # two roll-out steps (solver + correction)
    d, vy, vx, vz = ro.run(f32(z["d"]), f32(z["vy"]), f32(z["vx"]), f32(z["vz"]), f32(z["re"]), int(z["nroll"]))
    torch.cuda.synchronize()
    for a, k in ((d, "roll_d_sub4"), (vy, "roll_vy_sub4"), (vx, "roll_vx_sub4"), (vz, "roll_vz_sub4")):
        assert rel(a.reshape(-1)[::4], z[k]) < TOL_FIELD, (k, rel(a.reshape(-1)[::4], z[k]))
    assert np.allclose([float(a.double().norm()) for a in (d, vy, vx, vz)], z["roll_norms"], rtol=1e-5)


@pytest.mark.timeout(900)
def test_karman3d_full_size_step_against_oracle():
    """BASELINE configs[4] grid: one step at 128 x 64 x 64 against the float64 oracle (DST-preconditioned CG pressure solve),
    plus the size-independent property: the projected field is divergence free on interior active cells."""
    B, Y, X, Z = 1, 128, 64, 64
    g = o.geometry(Y, X, Z)
    d, v = o.synthetic_state(B, Y, X, Z, 1234)
    re = torch.tensor([o.RE_TRAIN[2]])
    with torch.no_grad():
        d1, v1 = o.karman3d_step(d, v, re, g)                     # spin-up: divergence free, consistent with the BCs
        d1, v1 = d1.float().double(), tuple(c.float().double() for c in v1)
        dr, vr = o.karman3d_step(d1, v1, re, g)
    sc = k3.Scene3D(Y, X, Z, device=DEV)
    sim = k3.Karman3DFlow(sc, B)
    out = sim.step(f32(d1), f32(v1[0]), f32(v1[1]), f32(v1[2]), f32(re))
    torch.cuda.synchronize()
    errs = [rel(a, b) for a, b in zip(out, (dr,) + tuple(vr))]
    assert max(errs) < TOL_FIELD, errs
    vy, vx, vz = (t.double().cpu() for t in out[1:])
    div = o.divergence((vy, vx, vz))
    inner = torch.zeros(Y, X, Z, dtype=torch.float64)
    inner[1:-1, 1:-1, 1:-1] = 1.0
    resid = float((div * inner * torch.as_tensor(g.active)).abs().max())
    assert resid < 2e-5, resid                                    # |v| ~ 1: fp32 round-off of the direct solve


# ---------------------------------------------------------------------------------------------
# training path: adjoint of the 3-D step, Conv3D gradients, SOL-n trainer
# ---------------------------------------------------------------------------------------------
TOL_GRAD = 1e-4          # as in 2-D: gradients pass through two fp32 direct solves per step


@pytest.mark.parametrize("shape,kw", [((2, 16, 8, 8), {}), ((2, 16, 8, 8), dict(grad_pad="dirichlet0")), ((1, 32, 16, 16), {})])
def test_karman3d_step_adjoint_against_oracle_autograd(shape, kw):
    B, Y, X, Z = shape
    g = o.geometry(Y, X, Z)
    d, v = o.synthetic_state(B, Y, X, Z, 13)
    v = tuple(c.float().double() for c in v)
    re = torch.tensor(o.RE_TRAIN[:B])
    gen = torch.Generator().manual_seed(3)
    w = [torch.randn(c.shape, generator=gen, dtype=torch.float64).float().double() for c in v]
    vr = tuple(c.clone().requires_grad_(True) for c in v)
    _, out = o.karman3d_step(d, vr, re, g, **kw)
    sum((a * b).sum() for a, b in zip(out, w)).backward()
    sc = k3.Scene3D(Y, X, Z, device=DEV)
    sim = k3.Karman3DFlow(sc, B, **kw)
    hv = [f32(c).requires_grad_(True) for c in v]
    hout = sim.step(f32(d), hv[0], hv[1], hv[2], f32(re))
    assert max(rel(a, b) for a, b in zip(hout[1:], out)) < TOL_FIELD
    sum((a * f32(b)).sum() for a, b in zip(hout[1:], w)).backward()
    errs = [rel(a.grad, b.grad) for a, b in zip(hv, vr)]
    assert max(errs) < TOL_GRAD, errs
    assert not hout[0].requires_grad                      # the density is a passive tracer


def test_karman3d_step_adjoint_identity_by_finite_differences():
    """<J u, w> = <u, J^T w> for the 3-D step at 128 x 64 x 64 with the HIP forward on both sides (central differences of the forward
    kernels against the adjoint kernels; no oracle involved).  fp32 differences limit the agreement to ~1e-3."""
    from sol_amd import synthetic
    B, Y, X, Z = 1, 128, 64, 64
    sc = k3.Scene3D(Y, X, Z, device=DEV)
    sim = k3.Karman3DFlow(sc, B)
    gen = torch.Generator().manual_seed(23)
    rn = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float32)
    st = (torch.rand(B, Y, X, Z, generator=gen).to(DEV), (1.0 + 0.1 * rn(B, Y + 1, X, Z)).to(DEV), (0.1 * rn(B, Y, X + 1, Z)).to(DEV), (0.1 * rn(B, Y, X, Z + 1)).to(DEV))
    re = synthetic.reynolds(B).float().to(DEV)
    with torch.no_grad():
        for _ in range(2):
            st = sim.step(*st, re)
    d0, v = st[0], [t.clone() for t in st[1:]]
    smooth = lambda t: torch.nn.functional.avg_pool3d(t[:, None], 5, 1, 2)[:, 0]
    u = [smooth(rn(*t.shape)).to(DEV) for t in v]
    w = [rn(*t.shape).to(DEV) for t in v]
    a = [t.clone().requires_grad_(True) for t in v]
    out = sim.step(d0, a[0], a[1], a[2], re)
    sum((o_ * w_).sum() for o_, w_ in zip(out[1:], w)).backward()
    dot = lambda xs, ys: float(sum((x.double() * y.double()).sum() for x, y in zip(xs, ys)))
    rhs = dot([t.grad for t in a], u)
    res = {}
    for eps in (2e-2, 1e-2, 5e-3):
        with torch.no_grad():
            p = sim.step(d0, *[t + eps * du for t, du in zip(v, u)], re)[1:]
            m = sim.step(d0, *[t - eps * du for t, du in zip(v, u)], re)[1:]
        res[eps] = dot([x.double() - y.double() for x, y in zip(p, m)], w) / (2 * eps)
    torch.cuda.synchronize()
    scale = dot([t.grad for t in a], [t.grad for t in a]) ** 0.5 * dot(u, u) ** 0.5
    print("adjoint identity 3-D: <u, J^T w> = %.6e, <J u, w> by central differences %s, |u||J^T w| = %.3e" % (rhs, res, scale))
    assert min(abs(x - rhs) for x in res.values()) < 2e-3 * abs(rhs) + 2e-4 * scale, (rhs, res, scale)


@pytest.mark.parametrize("shape,vscale", [((2, 16, 8, 8), 1.0), ((1, 20, 10, 10), 1.0), ((1, 32, 16, 16), 3.5), ((1, 128, 64, 64), 1.0)])
def test_karman3d_advection_adjoint_lds_window_equals_global_atomics_bit_for_bit(shape, vscale):
    """Option k3d_adj_tile: the advection adjoint's fixed-point scatter through an int64 LDS window per workgroup (k3b_advect_adj_tile: 4 x 4
    columns + halo 2, one global atomic per non-zero cell) against the all-global-atomics kernel -- integer adds commute, so the step's input
    gradient must be equal BIT FOR BIT: tiles that end inside the grid (20 x 10), the extra face row / column of the last tiles, targets
    beyond the window (velocities scaled to CFL > 2: those go to global memory directly), two simulations, the BASELINE configs[4] grid."""
    from sol_amd import synthetic
    B, Y, X, Z = shape
    sc = k3.Scene3D(Y, X, Z, device=DEV)
    sim = k3.Karman3DFlow(sc, B)
    gen = torch.Generator().manual_seed(41 + Y)
    rn = lambda *s_: torch.randn(*s_, generator=gen, dtype=torch.float32)
    st = (torch.rand(B, Y, X, Z, generator=gen).to(DEV), (vscale * (1.0 + 0.3 * rn(B, Y + 1, X, Z))).to(DEV), (vscale * 0.5 * rn(B, Y, X + 1, Z)).to(DEV),
          (vscale * 0.5 * rn(B, Y, X, Z + 1)).to(DEV))
    re = synthetic.reynolds(B).float().to(DEV)
    w = [rn(*t.shape).to(DEV) for t in st[1:]]
    grads = {}
    for tile in (1, 0):
        sol_amd._lib.set_option("k3d_adj_tile", tile)
        try:
            a = [t.clone().requires_grad_(True) for t in st[1:]]
            out = sim.step(st[0], a[0], a[1], a[2], re)
            sum((o_ * w_).sum() for o_, w_ in zip(out[1:], w)).backward()
            torch.cuda.synchronize()
            grads[tile] = [t.grad.clone() for t in a]
        finally:
            sol_amd._lib.set_option("k3d_adj_tile", 1)
    for x, y in zip(grads[1], grads[0]):
        assert torch.isfinite(x).all() and float(x.abs().max()) > 0
        assert torch.equal(x, y), float((x - y).abs().max())


@pytest.mark.timeout(1500)
def test_karman3d_full_size_step_adjoint_against_oracle():
    """BASELINE configs[4] grid: the ADJOINT of one step at 128 x 64 x 64 (diffusion^T, the scatter form of the advection's
    trilinear gathers with fp32 atomics, the second direct solve of the projection) against autograd through the float64 oracle
    (DST-preconditioned CG solves forward and backward)."""
    B, Y, X, Z = 1, 128, 64, 64
    g = o.geometry(Y, X, Z)
    d, v = o.synthetic_state(B, Y, X, Z, 4321)
    re = torch.tensor([o.RE_TRAIN[1]])
    with torch.no_grad():
        d, v = o.karman3d_step(d, v, re, g)                       # spin-up: a divergence-free state consistent with the BCs
    d, v = d.float().double(), tuple(c.float().double() for c in v)
    gen = torch.Generator().manual_seed(9)
    w = [torch.randn(c.shape, generator=gen, dtype=torch.float64).float().double() for c in v]
    vr = tuple(c.clone().requires_grad_(True) for c in v)
    _, out = o.karman3d_step(d, vr, re, g)
    sum((a * b).sum() for a, b in zip(out, w)).backward()
    sc = k3.Scene3D(Y, X, Z, device=DEV)
    sim = k3.Karman3DFlow(sc, B)
    hv = [f32(c).requires_grad_(True) for c in v]
    hout = sim.step(f32(d), hv[0], hv[1], hv[2], f32(re))
    assert max(rel(a, b) for a, b in zip(hout[1:], out)) < TOL_FIELD
    sum((a * f32(b)).sum() for a, b in zip(hout[1:], w)).backward()
    torch.cuda.synchronize()
    errs = [rel(a.grad, b.grad) for a, b in zip(hv, vr)]
    assert max(errs) < TOL_GRAD, errs


@pytest.mark.parametrize("shape,cin,cout,res,lrelu", [
    ((1, 5, 16, 16), 4, 32, False, True), ((2, 4, 16, 16), 32, 32, True, True), ((1, 4, 16, 16), 32, 3, False, False),
    ((1, 3, 64, 64), 32, 32, True, True), ((1, 3, 64, 64), 32, 3, False, False), ((1, 3, 64, 64), 4, 32, False, True)])
def test_conv3d_gradients_against_oracle(shape, cin, cout, res, lrelu):
    B, D, H, W = shape
    gen = torch.Generator().manual_seed(7 + cin + cout)
    r64 = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64).float().double()
    x, w, b = r64(B, D, H, W, cin).requires_grad_(True), (r64(5, 5, 5, cin, cout) / np.sqrt(125 * cin)).float().double().requires_grad_(True), r64(cout).requires_grad_(True)
    r = r64(B, D, H, W, cout).requires_grad_(True) if res else None
    gy = r64(B, D, H, W, cout) * 1e-3             # the upstream gradient lives on another scale than the activations (operand scales must not mix)
    (_oracle_conv(x, w, b, r, lrelu) * gy).sum().backward()
    hx, hw, hb = (f32(t.detach()).requires_grad_(True) for t in (x, w, b))
    hr = f32(r.detach()).requires_grad_(True) if res else None
    y = k3.conv3d_fn(hx, hw, hb, hr, lrelu, 0.3)
    (y * f32(gy)).sum().backward()
    errs = (rel(hx.grad, x.grad), rel(hw.grad, w.grad), rel(hb.grad, b.grad))
    per_plane = [rel(hx.grad[:, dd], x.grad[:, dd]) for dd in range(D)]
    per_slice = [rel(hw.grad[kd], w.grad[kd]) for kd in range(5)]
    assert max(errs) < 5e-6, (errs, per_plane, per_slice)
    if res:
        assert rel(hr.grad, r.grad) < 5e-6


def test_conv3d_weight_gradient_accumulation_state_is_checked():
    """conv3d_bwd_weight(acc=(state, first, last)) keeps a layer's partial sums in the caller's state dict across the calls of a reverse
    sweep; the C entry point cannot see the size of `partial`, so the Python wrapper refuses a state reused with another shape, a fresh
    state that does not start with first=True, and a continuation of a sequence that was closed -- and the legal sequence
    (first .. last) sums exactly like two single calls."""
    gen = torch.Generator().manual_seed(3)
    B, D, H, W = 1, 6, 8, 64
    x = torch.randn(B, D, H, W, 32, generator=gen, dtype=torch.float32).to(DEV)
    dz = [(torch.randn(B, D, H, W, 32, generator=gen, dtype=torch.float32) * 1e-2).to(DEV) for _ in range(2)]
    st = {}
    with pytest.raises(sol_amd._lib.SolError, match="first=True"):
        k3.conv3d_bwd_weight(x, dz[0], 32, 32, acc=(st, False, False))
    assert k3.conv3d_bwd_weight(x, dz[0], 32, 32, acc=(st, True, False)) == (None, None)
    with pytest.raises(sol_amd._lib.SolError, match="allocated for"):
        k3.conv3d_bwd_weight(x[:, :4].contiguous(), dz[0][:, :4].contiguous(), 32, 32, acc=(st, False, True))
    dW, db = k3.conv3d_bwd_weight(x, dz[1], 32, 32, acc=(st, False, True))
    with pytest.raises(sol_amd._lib.SolError, match="not open"):
        k3.conv3d_bwd_weight(x, dz[1], 32, 32, acc=(st, False, True))
    ref = [k3.conv3d_bwd_weight(x, d, 32, 32) for d in dz]
    assert rel(dW, ref[0][0].double() + ref[1][0].double()) < 2e-6 and rel(db, ref[0][1].double() + ref[1][1].double()) < 2e-6


@pytest.mark.parametrize("shape", [(1, 6, 8, 64), (2, 5, 16, 64), (1, 128, 64, 64)])
def test_conv3d_weight_gradient_one_launch_for_the_five_depth_slices_is_bit_identical(shape):
    """Option k3d_bww_jobs: the five depth slices of a 32 -> 32 Conv3D weight gradient as ONE launch.  1 (k_conv5x5_bww_sb_jobs with the
    five-launch form's blocks, partial layout and arithmetic) against 0 (five launches): dW and db bit for bit, single call and accumulated
    over two calls.  2 (the default: ONE round of workgroups, 51 per slice -- another block partition, so another summation order): equal to
    the five-launch form to round-off and not further from float64 than it."""
    B, D, H, W = shape
    gen = torch.Generator().manual_seed(77 + D)
    x = torch.randn(B, D, H, W, 32, generator=gen, dtype=torch.float32).to(DEV)
    dz = [(torch.randn(B, D, H, W, 32, generator=gen, dtype=torch.float32) * 1e-2).to(DEV) for _ in range(2)]
    out = {}
    for jobs in (2, 1, 0):
        sol_amd._lib.set_option("k3d_bww_jobs", jobs)
        try:
            one = k3.conv3d_bwd_weight(x, dz[0], 32, 32)
            st = {}
            k3.conv3d_bwd_weight(x, dz[0], 32, 32, acc=(st, True, False))
            two = k3.conv3d_bwd_weight(x, dz[1], 32, 32, acc=(st, False, True))
            torch.cuda.synchronize()
            out[jobs] = [t.clone() for t in one + two]
        finally:
            sol_amd._lib.set_option("k3d_bww_jobs", 2)
    for a, b in zip(out[1], out[0]):
        assert torch.equal(a, b), float((a - b).abs().max())
    for a, b in zip(out[2], out[0]):
        assert rel(a, b) < 2e-6, rel(a, b)
    if D <= 6:      # float64 reference of the single call (small case only: the direct sum over taps on the host)
        import torch.nn.functional as F
        w = torch.zeros(32, 32, 5, 5, 5, dtype=torch.float64, device=DEV, requires_grad=True)
        (F.conv3d(x.double().permute(0, 4, 1, 2, 3), w, padding=2) * dz[0].double().permute(0, 4, 1, 2, 3)).sum().backward()
        ref = w.grad.permute(2, 3, 4, 1, 0)
        e2, e0 = rel(out[2][0], ref), rel(out[0][0], ref)
        print("one-round form vs float64 %.2e, five-launch form %.2e" % (e2, e0))
        assert e2 < 3e-6 and e2 < 1.5 * e0 + 1e-7


@pytest.mark.parametrize("cout,res,mode", [(32, True, "lrelu"), (32, True, "dlrelu"), (3, False, "none"), (4, False, "none")])
def test_conv3d_full_size_one_launch_against_five_pass_and_shift_property(cout, res, mode):
    """BASELINE configs[4] size (128 x 64 x 64, all 1 024 workgroups and every XCD tile mapping of the one-launch kernel): (i) the
    one-launch kernel against the five-pass composition of the 2-D kernels (an independent code path held to the float64 oracle
    at small sizes), (ii) a size-independent property: shifting the input by one (y) plane shifts the output by one plane --
    bit for bit on the interior planes, since the per-tensor scales are unchanged and every output sees the same products in the
    same order."""
    B, D, H, W = 1, 128, 64, 64
    gen = torch.Generator().manual_seed(5 + cout)
    x = torch.randn(B, D, H, W, 32, generator=gen, dtype=torch.float32)
    x[:, :3] = 0
    x[:, -3:] = 0                                                    # zero planes at both ends: a cyclic roll is a shift
    x = x.to(DEV)
    w = (torch.randn(5, 5, 5, 32, cout, generator=gen, dtype=torch.float32) / np.sqrt(125 * 32)).to(DEV)
    b = torch.randn(cout, generator=gen, dtype=torch.float32).to(DEV)
    r = torch.randn(B, D, H, W, cout, generator=gen, dtype=torch.float32).to(DEV) if res else None
    act = torch.randn(B, D, H, W, cout, generator=gen, dtype=torch.float32).to(DEV) if mode == "dlrelu" else None
    lib = sol_amd.load()
    packed = torch.empty(lib.sol_conv3d_packed_floats(32, cout), dtype=torch.float32, device=DEV)
    sol_amd._lib.check(lib.sol_conv3d_pack(sol_amd._lib.stream(), sol_amd._lib.ptr(w), 32, cout, 0, sol_amd._lib.ptr(packed)))
    amax = sol_amd.ops.absmax_slots(x)
    run = lambda xx, rr, aa: k3.conv3d(xx, packed, None if mode == "dlrelu" else b, rr, cout, mode == "lrelu", 0.3, amax, None, act_ref=aa)
    y = run(x, r, act)
    sol_amd._lib.set_option("k3d_conv_fused", 0)
    try:
        y5 = run(x, r, act)
    finally:
        sol_amd._lib.set_option("k3d_conv_fused", 1)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all() and rel(y, y5) < 1e-6, rel(y, y5)
    # option k3d_conv_persist (256 workgroups of four consecutive tiles; default off, measured slower): bit-identical to one tile per workgroup
    sol_amd._lib.set_option("k3d_conv_persist", 1)
    try:
        yp = run(x, r, act)
        torch.cuda.synchronize()
    finally:
        sol_amd._lib.set_option("k3d_conv_persist", 0)
    assert torch.equal(yp, y)
    roll = lambda t: None if t is None else torch.roll(t, 1, dims=1).contiguous()
    ys = run(roll(x), roll(r), roll(act))
    torch.cuda.synchronize()
    assert torch.equal(ys[:, 4:-4], torch.roll(y, 1, dims=1)[:, 4:-4])


@pytest.mark.parametrize("shape,cin,mode", [((1, 6, 8, 64), 4, "fwd"), ((2, 5, 16, 16), 3, "fwd"), ((1, 4, 64, 64), 3, "bwd"), ((1, 128, 64, 64), 4, "fwd"),
                                            ((1, 128, 64, 64), 3, "bwd")])
def test_conv3d_thin_input_depth_packed_launch_against_float64_and_the_five_pass_form(shape, cin, mode):
    """sol_conv3d_thin (the <= 4 -> 32 layers as ONE 2-D 32 -> 32 launch over depth-packed channels): the first layer (forward packing,
    bias + LeakyReLU) and the output layer's data gradient (backward-data packing of the FORWARD kernel [5,5,5,32,cin], times
    LeakyReLU'(act_ref)) against torch float64 F.conv3d on the same device and against the five-pass composition of the 2-D fp32-MFMA
    kernel (sol_conv3d); small shapes, W = 16 (not the dx kernel's width) and the BASELINE configs[4] volume 128 x 64 x 64."""
    import torch.nn.functional as F
    B, D, H, W = shape
    gen = torch.Generator().manual_seed(31 + D + cin)
    x = torch.randn(B, D, H, W, cin, generator=gen, dtype=torch.float32).to(DEV)
    x4 = k3._pad_ch(x, 4)
    if mode == "fwd":
        w = (torch.randn(5, 5, 5, cin, 32, generator=gen, dtype=torch.float32) / np.sqrt(125 * cin)).to(DEV)
        b = torch.randn(32, generator=gen, dtype=torch.float32).to(DEV)
        y = k3.conv3d_thin(x4, k3._pack3d_thin(w, cin, 0), b, True, 0.3)
        z = F.conv3d(x.double().permute(0, 4, 1, 2, 3), w.double().permute(4, 3, 0, 1, 2), b.double(), padding=2).permute(0, 2, 3, 4, 1)
        ref = torch.where(z > 0, z, 0.3 * z)
        wf = torch.nn.functional.pad(w, (0, 0, 0, 4 - cin))
        y5 = k3.conv3d(x4, k3._pack3d(wf, 4, 32, 0), b, None, 32, True, 0.3)
    else:
        w = (torch.randn(5, 5, 5, 32, cin, generator=gen, dtype=torch.float32) / np.sqrt(125 * 32)).to(DEV)      # the forward kernel of a 32 -> cin layer
        act = torch.randn(B, D, H, W, 32, generator=gen, dtype=torch.float32).to(DEV)
        y = k3.conv3d_thin(x4, k3._pack3d_thin(w, cin, 1), None, False, 0.3, act_ref=act)
        # dx = conv3d(dy, flip(w)^T) = the transposed convolution of the forward layer
        z = F.conv_transpose3d(x.double().permute(0, 4, 1, 2, 3), w.double().permute(4, 3, 0, 1, 2), padding=2).permute(0, 2, 3, 4, 1)
        ref = z * torch.where(act > 0, 1.0, 0.3).double()
        y5 = k3.conv3d(x4, k3._pack3d(w, cin, 32, 1), None, None, 32, False, 0.3, act_ref=act)
    torch.cuda.synchronize()
    e, e5 = rel(y, ref), rel(y5, ref)
    print("depth-packed launch vs float64 %.2e (five-pass fp32-MFMA form %.2e)" % (e, e5))
    assert torch.isfinite(y).all() and e < 2e-6 and rel(y, y5) < 3e-6, (e, e5, rel(y, y5))


@pytest.mark.parametrize("shape,cr,mode", [((1, 6, 8, 64), 3, "fwd"), ((2, 5, 16, 16), 4, "bwd"), ((1, 3, 8, 64), 4, "fwd"), ((1, 128, 64, 64), 3, "fwd"),
                                           ((1, 128, 64, 64), 4, "bwd")])
def test_conv3d_thin_output_depth_packed_launch_against_float64_and_the_eight_row_kernel(shape, cr, mode):
    """sol_conv3d_thin_out (the 32 -> (<= 4) layers as ONE 2-D 32 -> 32 launch whose output channels are the depth taps, then a gather over five
    planes): the output layer (forward packing, bias) and the first layer's data gradient (backward-data packing of the FORWARD kernel
    [5,5,5,cr,32]) against torch float64 F.conv3d on the same device and against sol_conv3d (k_conv3d_sb8<1, 0> where W == 64); small shapes,
    fewer planes than taps (D = 3), W = 16 and the BASELINE configs[4] volume."""
    import torch.nn.functional as F
    B, D, H, W = shape
    gen = torch.Generator().manual_seed(61 + D + cr)
    x = torch.randn(B, D, H, W, 32, generator=gen, dtype=torch.float32).to(DEV)
    if mode == "fwd":
        w = (torch.randn(5, 5, 5, 32, cr, generator=gen, dtype=torch.float32) / np.sqrt(125 * 32)).to(DEV)
        b = torch.randn(cr, generator=gen, dtype=torch.float32).to(DEV)
        ref = F.conv3d(x.double().permute(0, 4, 1, 2, 3), w.double().permute(4, 3, 0, 1, 2), b.double(), padding=2).permute(0, 2, 3, 4, 1)
        y = k3.conv3d_thin_out(x, k3._pack3d_thin_out(w, cr, 0), b, cr, sol_amd.ops.absmax_slots(x))
        y8 = k3.conv3d(x, k3._pack3d(w, 32, cr, 0), b, None, cr, False, 0.3, sol_amd.ops.absmax_slots(x), None)
    else:
        w = (torch.randn(5, 5, 5, cr, 32, generator=gen, dtype=torch.float32) / np.sqrt(125 * 32)).to(DEV)      # forward kernel of a cr -> 32 layer
        xin = torch.zeros(B, cr, D, H, W, dtype=torch.float64, device=DEV, requires_grad=True)
        (F.conv3d(xin, w.double().permute(4, 3, 0, 1, 2), padding=2) * x.double().permute(0, 4, 1, 2, 3)).sum().backward()
        ref = xin.grad.permute(0, 2, 3, 4, 1)
        y = k3.conv3d_thin_out(x, k3._pack3d_thin_out(w, cr, 1), None, cr)                   # (absmax computed by the call)
        y8 = k3.conv3d(x, k3._pack3d(w, 32, cr, 1), None, None, cr, False, 0.3, sol_amd.ops.absmax_slots(x), None)
    torch.cuda.synchronize()
    e, e8 = rel(y, ref), rel(y8, ref)
    print("depth-packed thin-output launch vs float64 %.2e (sol_conv3d %.2e)" % (e, e8))
    assert y.shape == (B, D, H, W, cr) and e < 3e-6 and e < 2.0 * e8 + 2e-7, (e, e8)


@pytest.mark.parametrize("shape,cin", [((1, 6, 8, 64), 4), ((2, 5, 16, 64), 3), ((1, 128, 64, 64), 4)])
def test_conv3d_thin_input_depth_packed_weight_gradient_against_float64(shape, cin):
    """sol_conv3d_thin_bwd_weight_acc (the thin-input layer's weight gradient as ONE pass of the 2-D 32 -> 32 fp16 three-product kernel over the
    depth-packed tensor) against torch float64 autograd of F.conv3d and against the five-pass thin kernel; single call, and two calls
    accumulated in the caller's state (the unrolled trainer's form)."""
    import torch.nn.functional as F
    B, D, H, W = shape
    gen = torch.Generator().manual_seed(41 + D + cin)
    xs = [torch.randn(B, D, H, W, cin, generator=gen, dtype=torch.float32).to(DEV) for _ in range(2)]
    dzs = [(torch.randn(B, D, H, W, 32, generator=gen, dtype=torch.float32) * 1e-3).to(DEV) for _ in range(2)]
    refs = []
    for x, dz in zip(xs, dzs):
        w = torch.zeros(32, cin, 5, 5, 5, dtype=torch.float64, device=DEV, requires_grad=True)
        (F.conv3d(x.double().permute(0, 4, 1, 2, 3), w, padding=2) * dz.double().permute(0, 4, 1, 2, 3)).sum().backward()
        refs.append((w.grad.permute(2, 3, 4, 1, 0), dz.double().sum(dim=(0, 1, 2, 3))))
    dW, db = k3.conv3d_thin_bwd_weight(k3._pad_ch(xs[0], 4), dzs[0], cin)
    dW5, db5 = k3.conv3d_bwd_weight(k3._pad_ch(xs[0], 4), dzs[0], cin, 32)
    torch.cuda.synchronize()
    e, e5 = rel(dW, refs[0][0]), rel(dW5, refs[0][0])
    print("depth-packed weight gradient vs float64 %.2e (five-pass thin kernel %.2e), db %.2e" % (e, e5, rel(db, refs[0][1])))
    assert dW.shape == (5, 5, 5, cin, 32) and e < 3e-6 and rel(db, refs[0][1]) < 3e-6, (e, e5)
    st = {}
    assert k3.conv3d_thin_bwd_weight(k3._pad_ch(xs[0], 4), dzs[0], cin, acc=(st, True, False)) == (None, None)
    dW2, db2 = k3.conv3d_thin_bwd_weight(k3._pad_ch(xs[1], 4), dzs[1], cin, acc=(st, False, True))
    torch.cuda.synchronize()
    assert rel(dW2, refs[0][0] + refs[1][0]) < 3e-6 and rel(db2, refs[0][1] + refs[1][1]) < 3e-6


@pytest.mark.parametrize("shape,cout", [((1, 6, 8, 64), 3), ((2, 5, 16, 64), 4), ((1, 128, 64, 64), 3)])
def test_conv3d_thin_output_depth_packed_weight_gradient_against_float64(shape, cout):
    """sol_conv3d_thin_out_bwd_weight_acc: the weight gradient of the 32 -> cout (<= 4) output layer with the output gradient gathered over the
    depth offsets (one 32 -> 32 pass instead of five on a dz padded to 32 channels) against torch float64 autograd and the five-pass form."""
    import torch.nn.functional as F
    B, D, H, W = shape
    gen = torch.Generator().manual_seed(51 + D + cout)
    x = torch.randn(B, D, H, W, 32, generator=gen, dtype=torch.float32).to(DEV)
    dz = (torch.randn(B, D, H, W, cout, generator=gen, dtype=torch.float32) * 1e-3).to(DEV)
    w = torch.zeros(cout, 32, 5, 5, 5, dtype=torch.float64, device=DEV, requires_grad=True)
    (F.conv3d(x.double().permute(0, 4, 1, 2, 3), w, padding=2) * dz.double().permute(0, 4, 1, 2, 3)).sum().backward()
    ref_w, ref_b = w.grad.permute(2, 3, 4, 1, 0), dz.double().sum(dim=(0, 1, 2, 3))
    dW, db = k3.conv3d_thin_bwd_weight(k3._pad_ch(dz, 4), x, cout, thin_out=True)
    dW5, db5 = k3.conv3d_bwd_weight(x, dz, 32, cout)
    torch.cuda.synchronize()
    e, e5 = rel(dW, ref_w), rel(dW5, ref_w)
    print("depth-packed output-layer weight gradient vs float64 %.2e (five-pass form %.2e), db %.2e" % (e, e5, rel(db, ref_b)))
    assert dW.shape == (5, 5, 5, 32, cout) and db.shape == (cout,) and e < 3e-6 and rel(db, ref_b) < 3e-6, (e, e5)


@pytest.mark.parametrize("D", [16, 128])
def test_network_w64_against_torch_float64_autograd(D):
    """model_mars_moon3d on 16 x 64 x 64 and on the BASELINE configs[4] grid 128 x 64 x 64 (the one-launch Conv3D kernels): forward, input gradient
    and the 1.3 M parameter gradients of the fused HIP network against plain PyTorch float64 on the same device (F.conv3d and its
    autograd) -- an independent implementation at a size the CPU oracle cannot reach in test time.
    The reference applies LeakyReLU with the HIP forward's sign masks: with its own, ONE activation whose fp32 pre-activation
    rounds across zero changes a layer's gradient by 0.7 / sqrt(2.1 M) = 5e-4 relative (D = 16) -- measured: 5e-4 per layer, 1.2e-3 on the
    first layer, with a forward that agrees to 8e-7 -- a property of LeakyReLU' under round-off, not of either implementation.
    (The fp32 F.conv3d of this ROCm build shows the same through its own 2e-6 forward differences.)"""
    import make_golden as mg
    import torch.nn.functional as F
    B, H, W = 1, 64, 64
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(B, D, H, W, 4, generator=gen, dtype=torch.float32).to(DEV)
    gy = (torch.randn(B, D, H, W, 3, generator=gen, dtype=torch.float32) * 1e-3).to(DEV)
    params = [p.float() for p in mg.k3d_params()]
    net = k3.MarsMoon3D(device=DEV)
    net.set_weights([p.numpy() for p in params])
    net.params.requires_grad_(True)
    xi = x.clone().requires_grad_(True)
    out = net(xi)
    (out * gy).sum().backward()
    masks = [(t > 0).permute(0, 4, 1, 2, 3) for t in net.activations(x)]
    torch.cuda.synchronize()
    # the same network in torch float64: NCDHW, kernels DHWIO -> OIDHW
    tp = [p.to(DEV).double().requires_grad_(True) for p in params]
    conv = lambda t, k: F.conv3d(t, tp[2 * k].permute(4, 3, 0, 1, 2), tp[2 * k + 1], padding=2)
    lrelu = lambda z, m: z * torch.where(m, 1.0, 0.3).to(z.dtype)
    xt = x.double().permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    h = lrelu(conv(xt, 0), masks[0])
    flips = 0
    for k in range(5):
        a = lrelu(conv(h, 1 + 2 * k), masks[1 + 2 * k])
        z = conv(a, 2 + 2 * k) + h
        flips += int(((z > 0) != masks[2 + 2 * k]).sum())
        h = lrelu(z, masks[2 + 2 * k])
    ref = conv(h, 11)
    (ref * gy.double().permute(0, 4, 1, 2, 3)).sum().backward()
    torch.cuda.synchronize()
    off = net.offsets
    per = [rel(net.params.grad[off[k]:off[k + 1]], tp[k].grad.reshape(-1)) for k in range(24)]
    e_out, e_x = rel(out.detach(), ref.detach().permute(0, 2, 3, 4, 1)), rel(xi.grad, xt.grad.permute(0, 2, 3, 4, 1))
    print("torch float64 reference: out %.2e, dx %.2e, params max %.2e, %d block-output signs differ" % (e_out, e_x, max(per), flips))
    assert e_out < 5e-6 and e_x < 1e-5 and max(per) < 1e-5, (e_out, e_x, per)


@pytest.mark.parametrize("shape", [(1, 4, 64, 64), (2, 5, 16, 16)])
def test_network_fused_reverse_sweep_equals_per_layer_autograd(shape):
    """MarsMoon3D as one autograd node (data gradient + skip gradient + LeakyReLU' in the conv epilogues, absmax slots handed
    from producer to consumer) against the per-layer composition with torch glue: outputs bit-equal, input and parameter
    gradients to round-off.  W = 64 runs the one-launch Conv3D kernels with the SOL_EPI_DLRELU epilogue, W = 16 the five-pass
    form; the per-layer path itself is held to the float64 oracle by test_conv3d_gradients_against_oracle and the trainer test."""
    import make_golden as mg
    B, D, H, W = shape
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B, D, H, W, 4, generator=gen, dtype=torch.float32).to(DEV)
    gy = (torch.randn(B, D, H, W, 3, generator=gen, dtype=torch.float32) * 1e-3).to(DEV)
    res = {}
    for fused in (True, False):
        net = k3.MarsMoon3D(device=DEV)
        net.set_weights([p.numpy() for p in mg.k3d_params()])
        net.fused_backward = fused
        net.params.requires_grad_(True)
        xi = x.clone().requires_grad_(True)
        out = net(xi)
        (out * gy).sum().backward()
        torch.cuda.synchronize()
        res[fused] = (out.detach(), xi.grad, net.params.grad)
    assert torch.equal(res[True][0], res[False][0])
    assert rel(res[True][1], res[False][1]) < 2e-6, rel(res[True][1], res[False][1])
    assert rel(res[True][2], res[False][2]) < 2e-6, rel(res[True][2], res[False][2])
    off = k3.MarsMoon3D(device="cpu").offsets
    per = [rel(res[True][2][off[k]:off[k + 1]], res[False][2][off[k]:off[k + 1]]) for k in range(24)]
    assert max(per) < 1e-5, per


def test_karman3d_training_step_gradient_by_finite_differences():
    """d loss / d weights of the SOL-2 3-D training step (32 x 16 x 16, B = 2) along a random direction in parameter space against
    central differences of the same engine's loss -- no oracle.  fp32 losses limit the agreement to ~1e-3."""
    from sol_amd import synthetic
    B, Y, X, Z, ms = 2, 32, 16, 16, 2
    sc = k3.Scene3D(Y, X, Z, device=DEV)
    net = k3.MarsMoon3D(seed=4, device=DEV)
    w = net.get_weights()
    w[22] = w[22] * 0.1
    net.set_weights(w)
    tr = k3.Karman3DTrainer(net, sc, B, ms, (0.2, 0.2, 0.2), synthetic.STD_RE)
    gen = torch.Generator().manual_seed(8)
    rn = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float32)
    st = (torch.rand(B, Y, X, Z, generator=gen).to(DEV), (1.0 + 0.1 * rn(B, Y + 1, X, Z)).to(DEV), (0.1 * rn(B, Y, X + 1, Z)).to(DEV), (0.1 * rn(B, Y, X, Z + 1)).to(DEV))
    re = synthetic.reynolds(B).float().to(DEV)
    gts = []
    with torch.no_grad():
        st = tr.sim.step(*st, re)
        gs = (st[0], st[1] + 0.02, st[2], st[3])
        for _ in range(ms):
            gs = tr.sim.step(*gs, re)
            gts.append(tuple(t.clone() for t in gs[1:]))
    loss = float(tr.fwd_bwd(*st, re, gts))
    g = tr.grads.detach().double().clone()
    u = torch.zeros(net.n_params, dtype=torch.float64)
    for k in range(len(net.shapes)):
        sl = slice(int(net.offsets[k]), int(net.offsets[k + 1]))
        wk = net.params.detach()[sl].double().cpu()
        u[sl] = torch.randn(wk.numel(), generator=gen, dtype=torch.float64) * (float(wk.abs().mean()) + 1e-3)
    u = u.to(DEV)
    rhs = float((g * u).sum())
    p0 = net.params.detach().clone()
    res = {}
    for eps in (3e-2, 1e-2, 3e-3):
        vals = []
        for sgn in (1.0, -1.0):
            with torch.no_grad():
                net.params.copy_((p0.double() + sgn * eps * u).float())
            net._tpacks = None
            vals.append(float(tr.fwd_bwd(*st, re, gts)))
        res[eps] = (vals[0] - vals[1]) / (2 * eps)
    with torch.no_grad():
        net.params.copy_(p0)
    print("3-D training-step gradient: loss %.6e, <grad, u> = %.6e, central differences %s" % (loss, rhs, res))
    assert min(abs(v - rhs) for v in res.values()) < 3e-3 * abs(rhs), (rhs, res)


@pytest.mark.timeout(1500)
def test_karman3d_sol16_full_size_replay_equals_eager_and_directional_derivative():
    """BASELINE configs[4] at its REAL depth: SOL-16 at 128 x 64 x 64 (B = 1, the per-GPU share of the 8-GPU config), the unroll of
    karman_train.py:397-457 with three components.  The float64 oracle cannot afford this size (16 unrolled steps with autograd
    through DST-preconditioned CG solves), so the step is held to properties instead:
      (1) the replayed hipGraph equals the eager composition BIT FOR BIT (loss, the 1.3 M-element gradient, final state) -- every
          kernel of the step is deterministic since the advection adjoint scatters in fixed point;
      (2) a second replay reproduces the first bit for bit, the loss is finite and non-trivial;
      (3) <grad, u> along a random direction in parameter space equals the central difference of the engine's own loss
          (the reverse sweep through 16 solver adjoints and 16 x 24 convolutions is the derivative of the forward unroll);
      (4) one TF-Adam step with the gradient changes every parameter tensor and lowers the loss at a small learning rate."""
    from sol_amd import synthetic
    B, Y, X, Z, ms = 1, 128, 64, 64, 16
    sc = k3.Scene3D(Y, X, Z, device=DEV)
    gen = torch.Generator().manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float32)
    st = (torch.rand(B, Y, X, Z, generator=gen).to(DEV), (1.0 + 0.1 * rn(B, Y + 1, X, Z)).to(DEV), (0.1 * rn(B, Y, X + 1, Z)).to(DEV), (0.1 * rn(B, Y, X, Z + 1)).to(DEV))
    re = synthetic.reynolds(B).float().to(DEV)

    def make(use_graph):
        net = k3.MarsMoon3D(seed=3, device=DEV)
        w = net.get_weights()
        w[22] = w[22] * 0.01
        net.set_weights(w)
        return net, k3.Karman3DTrainer(net, sc, B, ms, (0.2, 0.2, 0.2), synthetic.STD_RE, use_graph=use_graph)

    net_g, tr_g = make(True)
    gts = []
    with torch.no_grad():
        st = tr_g.sim.step(*st, re)                     # spin-up: consistent with the boundary conditions
        gs = (st[0], st[1] + 0.02, st[2], st[3])
        for _ in range(ms):
            gs = tr_g.sim.step(*gs, re)
            gts.append(tuple(t.clone() for t in gs[1:]))
    bits = lambda t: t.detach().contiguous().view(torch.int32)
    l1 = tr_g.fwd_bwd(*st, re, gts).clone()
    assert tr_g._graph is not None
    g1, f1 = tr_g.grads.clone(), [t.clone() for t in tr_g.final]
    tr_g._grads.zero_()
    l2 = tr_g.fwd_bwd(*st, re, gts).clone()             # a pure replay
    assert bool((bits(l1) == bits(l2)).all()) and bool((bits(g1) == bits(tr_g.grads)).all()), "two replays of the SOL-16 step differ"
    assert np.isfinite(float(l1)) and float(l1) > 0 and float(g1.abs().max()) > 0 and bool(torch.isfinite(g1).all())
    # (1) eager composition
    net_e, tr_e = make(False)
    le = tr_e.fwd_bwd(*st, re, gts)
    assert bool((bits(le) == bits(l1)).all()), (float(le), float(l1))
    assert bool((bits(tr_e.grads) == bits(g1)).all()), "replayed graph and eager composition give different gradients"
    for a, b in zip(tr_e.final, f1):
        assert bool((bits(a) == bits(b)).all())
    del tr_e, net_e
    # (3) directional derivative, with the replayed graph (the weights are read from net.params at every replay)
    g = g1.double()
    u = torch.zeros(net_g.n_params, dtype=torch.float64)
    for k in range(len(net_g.shapes)):
        sl = slice(int(net_g.offsets[k]), int(net_g.offsets[k + 1]))
        wk = net_g.params.detach()[sl].double().cpu()
        u[sl] = torch.randn(wk.numel(), generator=gen, dtype=torch.float64) * (float(wk.abs().mean()) + 1e-3)
    u = u.to(DEV)
    rhs = float((g * u).sum())
    p0 = net_g.params.detach().clone()
    res = {}
    for eps in (1e-3, 3e-4, 1e-4):                      # (16 unrolled steps: the loss leaves its linear range beyond ~1e-3 of the weights' scale)
        vals = []
        for sgn in (1.0, -1.0):
            with torch.no_grad():
                net_g.params.copy_((p0.double() + sgn * eps * u).float())
            vals.append(float(tr_g.fwd_bwd(*st, re, gts)))
        res[eps] = (vals[0] - vals[1]) / (2 * eps)
    with torch.no_grad():
        net_g.params.copy_(p0)
    print("3-D SOL-16 full size: loss %.6e, <grad, u> = %.6e, central differences %s" % (float(l1), rhs, res))
    assert min(abs(v - rhs) for v in res.values()) < 1e-2 * abs(rhs), (rhs, res)
    # the graph is still the function it was: the original weights reproduce the first replay bit for bit after eight replays on
    # other weights.  (Regression: with the loss as a torch reduction the captured graph held memset nodes -- the reduction's
    # semaphores -- and after a few replays reported 0.5x / 2x the true per-step losses)
    l4 = tr_g.fwd_bwd(*st, re, gts).clone()
    assert bool((bits(l4) == bits(l1)).all()) and bool((bits(tr_g.grads) == bits(g1)).all()), (float(l4), float(l1))
    # (4) one optimizer step
    tr_g.fwd_bwd(*st, re, gts)
    before = [t.clone() for t in net_g.tensors()]
    tr_g.apply_gradients(1e-6)
    assert all(float((a - b).abs().max()) > 0 for a, b in zip(net_g.tensors(), before) if a.dim() > 1)
    l3 = float(tr_g.fwd_bwd(*st, re, gts))
    assert np.isfinite(l3) and l3 < float(l1), (l3, float(l1))


@pytest.mark.parametrize("use_graph", [False, True])
def test_karman3d_trainer_sol2_against_oracle(use_graph):
    """SOL-2 at 32 x 16 x 16, B = 2: loss, the full 1.3 M-element gradient and one TF-Adam update against the float64 oracle
    (autograd through the unrolled 3-D graph); eager composition and the replayed hipGraph (second replay checked)."""
    import make_golden as mg
    import sol_oracle as o2
    B, Y, X, Z, ms = 2, 32, 16, 16, 2
    g = o.geometry(Y, X, Z)
    d, v = o.synthetic_state(B, Y, X, Z, 77)
    d, v = d.float().double(), tuple(c.float().double() for c in v)
    re = torch.tensor(o.RE_TRAIN[:B])
    gts = []
    for i in range(ms):
        _, gv = o.synthetic_state(B, Y, X, Z, 500 + i)
        gts.append(tuple(c.float().double() for c in gv))
    std_v = (0.2, 0.25, 0.3)
    params = [p.clone().requires_grad_(True) for p in mg.k3d_params()]
    loss = o.unrolled_loss(params, d, v, re, gts, g, std_v, o.STD_RE)
    loss.backward()
    gref = torch.cat([p.grad.reshape(-1) for p in params])
    sc = k3.Scene3D(Y, X, Z, device=DEV)
    net = k3.MarsMoon3D(device=DEV)
    net.set_weights([p.detach().numpy() for p in params])
    tr = k3.Karman3DTrainer(net, sc, B, ms, std_v, o.STD_RE, use_graph=use_graph)
    hl = tr.fwd_bwd(d, v[0], v[1], v[2], re, gts)
    if use_graph:
        assert tr._graph is not None
        tr._grads.zero_()
        hl = tr.fwd_bwd(d, v[0], v[1], v[2], re, gts)          # a pure replay
    assert abs(float(hl) - float(loss)) < 1e-5 * abs(float(loss)), (float(hl), float(loss))
    assert rel(tr.grads, gref) < TOL_GRAD, rel(tr.grads, gref)
    off = net.offsets
    per = [rel(tr.grads[off[k]:off[k + 1]], params[k].grad.reshape(-1)) for k in range(len(params))]
    assert max(per) < 3 * TOL_GRAD, per
    # one TF-Adam update in closed form from the gradient the trainer holds (epsilon-hat form: m = 0.1 g, v = 0.001 g^2,
    # lr_t = lr sqrt(1 - 0.999) / (1 - 0.9)); the float64 oracle's update differs where fp32 round-off flips the sign of a
    # near-zero gradient element, so the comparison is made on the same gradient
    g64, p64 = tr.grads.double(), net.params.detach().double()
    expect = p64 - 1e-4 * (1 - 0.999) ** 0.5 / (1 - 0.9) * 0.1 * g64 / ((0.001 * g64 * g64).sqrt() + 1e-8)
    tr.apply_gradients(1e-4)
    assert rel(net.params.detach(), expect) < 1e-6
    p2, _, _ = o2.adam_tf([p.detach() for p in params], [p.grad for p in params], [torch.zeros_like(p) for p in params],
                          [torch.zeros_like(p) for p in params], 1, 1e-4)
    assert rel(net.params.detach(), torch.cat([p.reshape(-1) for p in p2])) < 1e-4


def test_karman3d_data_parallel_two_ranks_one_gpu(tmp_path):
    """Karman3DTrainer with world_size 2 (both ranks on cuda:0, gloo transport, one simulation each) against the single process with
    both simulations: ONE all-reduce of [gradient | loss] per step gives the global-batch gradient and loss, the replicas hold
    bit-identical weights after two Adam steps (BASELINE configs[4] shards the simulations over 8 GPUs this way)."""
    import subprocess
    import socket
    import dp_worker
    Bg, Y, X, Z, ms = 2, 32, 16, 16, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    prefix = str(tmp_path / "dp3")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "dp_worker.py"), prefix, "k3d"] +
                                      [str(v) for v in (Bg, Y, X, Z, ms)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    r0, r1 = (np.load(prefix + "_rank%d.npz" % r) for r in range(2))
    assert np.array_equal(r0["params"], r1["params"])
    assert np.array_equal(r0["grads0"], r1["grads0"]) and np.array_equal(r0["losses"], r1["losses"])
    losses, grads0, params = dp_worker.run3d(Bg, Y, X, Z, ms, 0, 1)
    assert np.allclose(r0["losses"], losses, rtol=2e-5)
    assert rel(torch.as_tensor(r0["grads0"]), torch.as_tensor(grads0)) < 2e-5
    assert rel(torch.as_tensor(r0["params"]), torch.as_tensor(params)) < 1e-5


def test_torch_library_ops_3d_are_registered_and_differentiable(scene_small):
    """torch.ops.sol.karman3d_step / conv3d: the dispatcher entries run the same C-ABI entry points and hand-written adjoints as the
    module-level functions (bit-equal outputs and gradients)."""
    import sol_amd.torch_ops as tops
    B, Y, X, Z = 2, 32, 16, 16
    d, v = o.synthetic_state(B, Y, X, Z, 3)
    re = f32(torch.tensor(o.RE_TRAIN[:B]))
    sim = k3.Karman3DFlow(scene_small, B)
    h = tops.register_scene3d(sim)
    res = []
    for use_op in (True, False):
        hv = [f32(c).requires_grad_(True) for c in v]
        out = torch.ops.sol.karman3d_step(f32(d), hv[0], hv[1], hv[2], re, h) if use_op else sim.step(f32(d), hv[0], hv[1], hv[2], re)
        (out[1].sum() + 2 * out[2].sum() - out[3].sum()).backward()
        res.append(([t.detach() for t in out], [t.grad for t in hv]))
    assert all(torch.equal(a, b) for a, b in zip(res[0][0], res[1][0]))
    assert all(rel(a, b) < 1e-6 for a, b in zip(res[0][1], res[1][1]))            # (the advection adjoint scatters with fp32 atomics)
    x = torch.randn(1, 4, 16, 16, 32, device=DEV, dtype=torch.float32, requires_grad=True)
    w = (torch.randn(5, 5, 5, 32, 32, device=DEV, dtype=torch.float32) * 0.02).requires_grad_(True)
    b = torch.zeros(32, device=DEV, dtype=torch.float32, requires_grad=True)
    y1 = torch.ops.sol.conv3d(x, w, b, None, True, 0.3)
    y1.square().sum().backward()
    gx1, gw1 = x.grad.clone(), w.grad.clone()
    x.grad = w.grad = b.grad = None
    y2 = k3.conv3d_fn(x, w, b, None, True, 0.3)
    y2.square().sum().backward()
    assert torch.equal(y1, y2) and torch.equal(gx1, x.grad) and torch.equal(gw1, w.grad)


def test_karman3d_script(tmp_path):
    """scripts/karman3d.py: data generation at 16 x 8 x 8, a SOL-2 training demo on those frames, and a corrected roll-out
    with the model it wrote."""
    import importlib.util
    from sol_amd import scene
    sdir = os.path.join(os.path.dirname(os.path.abspath(sol_amd.__file__)), "scripts")
    sys.path.insert(0, sdir)
    spec = importlib.util.spec_from_file_location("sol_script_karman3d", os.path.join(sdir, "karman3d.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(["-r", "8", "-t", "8", "--re", "1.6e5", "-o", str(tmp_path / "run"), "--train-sol", "2", "--train-steps", "2"])
    v = scene.read_zipped_array(out + "/velo_000007.npz")
    d = scene.read_zipped_array(out + "/dens_000007.npz")
    assert v.shape == (1, 17, 9, 9, 3) and d.shape == (1, 16, 8, 8, 1) and np.isfinite(v).all() and d.max() > 0
    assert abs(float(v[0, 0, 0, 0, 0]) - 1.0) < 1e-6 and float(v[0, :, :, 8, 0].max()) == 0.0      # component 0 = flow component, zero padded beyond Z
    assert os.path.isfile(out + "/model3d.pt")
    out2 = mod.main(["-r", "8", "-t", "4", "--re", "1.6e5", "-o", str(tmp_path / "run2"), "--model", out + "/model3d.pt"])
    v2 = scene.read_zipped_array(out2 + "/velo_000003.npz")
    assert np.isfinite(v2).all() and np.abs(v2 - scene.read_zipped_array(out + "/velo_000003.npz")).max() > 0       # the corrector acts


@pytest.mark.parametrize("shape,use_graph", [((2, 32, 16, 16), False), ((1, 128, 64, 64), True)])
def test_karman3d_manual_schedule_equals_autograd_composition(shape, use_graph):
    """Karman3DTrainer's hand-written schedule (forward unroll + reverse sweep over the C ABI, no autograd graph: the default) against the
    torch-autograd composition of the same differentiable ops (schedule="autograd", the form of rounds 3-4): per-step losses, the 1.3 M
    parameter gradient and the final state.  Same kernels on the same operands except for the order of a few elementwise additions:
    agreement to fp32 round-off.  At the BASELINE configs[4] grid through the replayed graph (captured through the kernel-nodes-only guard)."""
    B, Y, X, Z = shape
    ms = 3
    sc = k3.Scene3D(Y, X, Z, device=DEV)
    gen = torch.Generator().manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=gen)
    st = (torch.rand(B, Y, X, Z, generator=gen), 1.0 + 0.2 * r(B, Y + 1, X, Z), 0.2 * r(B, Y, X + 1, Z), 0.2 * r(B, Y, X, Z + 1))
    re = torch.tensor(o.RE_TRAIN[:B])
    gts = [(1.0 + 0.2 * r(B, Y + 1, X, Z), 0.2 * r(B, Y, X + 1, Z), 0.2 * r(B, Y, X, Z + 1)) for _ in range(ms)]
    res = {}
    for schedule in ("manual", "autograd"):
        net = k3.MarsMoon3D(device=DEV)
        w = net.get_weights()
        w[22] = w[22] * 0.05
        net.set_weights(w)
        tr = k3.Karman3DTrainer(net, sc, B, ms, (0.2, 0.25, 0.3), o.STD_RE, use_graph=use_graph, schedule=schedule)
        assert tr.schedule == schedule
        for _ in range(2 if use_graph else 1):                  # (graph: the second call is a pure replay)
            tr._grads.zero_()
            loss = tr.fwd_bwd(*st, re, gts)
        torch.cuda.synchronize()
        res[schedule] = (float(loss), tr.loss_steps.clone(), tr.grads.clone(), [t.clone() for t in tr.final])
        del tr, net
    lm, la = res["manual"], res["autograd"]
    assert np.isfinite(lm[0]) and abs(lm[0] - la[0]) < 2e-6 * abs(la[0]), (lm[0], la[0])
    assert rel(lm[1], la[1]) < 2e-6
    assert rel(lm[2], la[2]) < 2e-5, rel(lm[2], la[2])
    for a, b in zip(lm[3], la[3]):
        assert rel(a, b) < 2e-6
    assert float(lm[2].abs().max()) > 0
    with pytest.raises(ValueError):
        k3.Karman3DTrainer(k3.MarsMoon3D(device=DEV), sc, B, ms, (0.2, 0.25, 0.3), o.STD_RE, schedule="tf1")


@pytest.mark.parametrize("shape", [(2, 32, 16, 16), (1, 128, 64, 64)])
def test_karman3d_fused_reverse_sweep_glue_equals_the_torch_composition(shape):
    """glue="fused" (sol_karman3d_correct_bwd: G += gin and the 4-channel output gradient in one launch; sol_karman3d_feature_bwd: the feature map's
    adjoint in one launch) against glue="torch" (the seven + three elementwise torch kernels per unrolled step they replace): same per-step losses
    bit for bit (the forward is untouched), the parameter gradient to round-off (one add differs: a + f b is a fused multiply-add in the kernel).
    16 x 16 planes run the five-pass weight gradients (the 4-channel gradient is sliced back to cout there), 64 x 64 the depth-packed launches."""
    B, Y, X, Z = shape
    ms = 3
    sc = k3.Scene3D(Y, X, Z, device=DEV)
    gen = torch.Generator().manual_seed(19)
    r = lambda *s: torch.randn(*s, generator=gen)
    st = (torch.rand(B, Y, X, Z, generator=gen), 1.0 + 0.2 * r(B, Y + 1, X, Z), 0.2 * r(B, Y, X + 1, Z), 0.2 * r(B, Y, X, Z + 1))
    re = torch.tensor(o.RE_TRAIN[:B])
    gts = [(1.0 + 0.2 * r(B, Y + 1, X, Z), 0.2 * r(B, Y, X + 1, Z), 0.2 * r(B, Y, X, Z + 1)) for _ in range(ms)]
    res = {}
    for glue in ("fused", "torch"):
        net = k3.MarsMoon3D(device=DEV)
        w = net.get_weights()
        w[22] = w[22] * 0.05
        net.set_weights(w)
        tr = k3.Karman3DTrainer(net, sc, B, ms, (0.2, 0.25, 0.3), o.STD_RE, glue=glue)
        tr._grads.zero_()
        loss = tr.fwd_bwd(*st, re, gts)
        torch.cuda.synchronize()
        res[glue] = (float(loss), tr.loss_steps.clone(), tr.grads.clone())
        del tr, net
    f, t = res["fused"], res["torch"]
    assert np.isfinite(f[0]) and torch.equal(f[1], t[1])
    assert float(f[2].abs().max()) > 0 and rel(f[2], t[2]) < 2e-6, rel(f[2], t[2])
