"""Host-side mirror of the reference's Python surface (no GPU needed: CPU tensors)."""
import numpy as np
import pytest
import torch

import sol_amd
import sol_oracle as o
from sol_amd import ops, synthetic


def test_scene_masks_match_the_oracle_geometry():
    for (Y, X) in [(16, 8), (64, 32), (128, 64)]:
        dom = sol_amd.Domain([Y, X], box=sol_amd.box[0:200, 0:100])
        flow = sol_amd.KarmanFlow()
        active, inflow = flow.scene_arrays(dom)
        g = o.geometry(Y, X)
        assert np.array_equal(active, g.active) and np.array_equal(inflow, g.inflow)
        bcv, bcm = sol_amd.velocity_bc_masks(Y, X, batch_size=3)
        assert bcv.shape == (3, Y + 1, X, 1)
        assert np.array_equal(bcv[0, ..., 0], g.bc_mask) and np.array_equal(bcm[2, ..., 0], g.bc_mask)
        assert dom.dx == (100.0 / X, 100.0 / X)


def test_box_syntax_and_geometry_objects():
    b = sol_amd.box[5:10, 25:75]
    assert b.lower == (5.0, 25.0) and b.upper == (10.0, 75.0)
    assert sol_amd.box([32, 32]).size == (32.0, 32.0)
    flow = sol_amd.KarmanFlow()
    assert flow.infl.geometry.lower == (5.0, 25.0) and flow.obst.geometry.radius == 10.0
    with pytest.raises(NotImplementedError):
        sol_amd.KarmanFlow(make_input_divfree=True)


def test_fluid_state_layouts_and_glue():
    B, Y, X = 2, 8, 4
    dom = sol_amd.Domain([Y, X], box=sol_amd.box[0:200, 0:100])
    gen = torch.Generator().manual_seed(0)
    vy = torch.randn(B, Y + 1, X, generator=gen, dtype=torch.float32)
    vx = torch.randn(B, Y, X + 1, generator=gen, dtype=torch.float32)
    st_ref = o.staggered_tensor(vy.double(), vx.double()).float()
    st = sol_amd.Fluid(dom, density=torch.zeros(B, Y, X, 1, dtype=torch.float32), velocity=st_ref, batch_size=B, device="cpu")
    assert st.density.data.shape == (B, Y, X, 1)
    assert torch.equal(st.velocity.data[0].data[..., 0], vy) and torch.equal(st.velocity.data[1].data[..., 0], vx)
    assert torch.equal(st.velocity.staggered_tensor(), st_ref)
    re = torch.tensor([2.0, 3.0], dtype=torch.float32)
    feat = sol_amd.to_feature(st, re)
    assert torch.allclose(feat, o.to_feature(vy.double(), vx.double(), re.double()).float())
    corr = sol_amd.to_staggered(feat[..., 0:2], dom.box)
    cy, cx = o.to_staggered(feat[..., 0:2].double())
    assert torch.equal(corr.data[0].data[..., 0], cy.float()) and torch.equal(corr.data[1].data[..., 0], cx.float())
    st2 = st.copied_with(velocity=st.velocity + corr)
    assert torch.allclose(st2.velocity.data[0].data[..., 0], vy + cy.float())
    assert st2.density is st.density and st2._batch_size == B


def test_model_matches_oracle_init_and_keras_weight_order():
    net = sol_amd.model_mars_moon(cin=3, cout=2, seed=0, device="cpu")
    ref = o.init_params(0)
    ws = net.get_weights()
    assert len(ws) == 24 and [w.shape for w in ws] == [tuple(p.shape) for p in ref]
    for w, p in zip(ws, ref):
        assert np.allclose(w, p.numpy().astype(np.float32))
    ws[3][:] = 0.5
    net.set_weights(ws)
    assert float(net.tensors()[3].detach().min()) == 0.5
    assert net.losses == []


def test_burgers_circulant_matches_oracle():
    for n in (32, 33):
        assert np.allclose(ops.circulant_diffusion_matrix(n, 0.01), o.burgers_diffusion_matrices(n, n, 0.01)[0].numpy(), atol=1e-14)


def test_synthetic_inputs_are_deterministic_and_shaped():
    d, vy, vx = synthetic.state(3, 16, 8, 7)
    d2, vy2, vx2 = synthetic.state(3, 16, 8, 7)
    assert torch.equal(vy, vy2) and d.shape == (3, 16, 8) and vy.shape == (3, 17, 8) and vx.shape == (3, 16, 9)
    gy, gx = synthetic.frames(2, 3, 16, 8, 9)
    assert gy.shape == (2, 3, 17, 8) and gx.shape == (2, 3, 16, 9)
    assert abs(synthetic.STD_RE - 1732512.626) < 1e-2
    assert synthetic.reynolds(8).tolist()[6] == 160000.0


def test_lr_schedule_and_sharding():
    lr = 1e-4
    seq = []
    for ep in range(25):
        lr = sol_amd.lr_schedule(ep, lr)
        seq.append(lr)
    assert np.isclose(seq[10], 1e-4) and np.isclose(seq[11], 1e-5) and np.isclose(seq[16], 1e-6)
    assert np.isclose(seq[21], 1e-7) and np.isclose(seq[23], 5e-8)
    assert sol_amd.dist.shard_range(48, 3, 8) == (18, 24)
    with pytest.raises(ValueError):
        sol_amd.dist.shard_range(10, 0, 4)


def test_coarse_inverse_matches_galerkin_projection_of_the_oracle_matrix():
    import scipy.sparse as sp
    from sol_amd.precond import pressure_matrix_dense_coarse, coarse_inverse
    Y, X = 64, 32
    g = o.geometry(Y, X)
    M = (-g.pressure_matrix()).tocsr()
    N = Y * X
    jj, ii = np.divmod(np.arange(N), X)
    blk = (jj // 8) * (X // 8) + ii // 8
    P = sp.csr_matrix((g.active.ravel(), (np.arange(N), blk)), shape=(N, (Y // 8) * (X // 8)))
    ref = (P.T @ M @ P).toarray()
    assert np.allclose(pressure_matrix_dense_coarse(g.active), ref, atol=1e-12)
    ci = coarse_inverse(g.active)
    assert ci.dtype == np.float32 and np.allclose(ci @ ref, np.eye(ref.shape[0]), atol=1e-4)
    lib = sol_amd.load()
    assert lib.sol_karman_precond_supported(128, 64) == 1 and lib.sol_karman_precond_supported(64, 32) == 1
    assert lib.sol_karman_precond_supported(16, 8) == 0


def test_direct_solver_blob_reproduces_the_exact_pressure_solve():
    """precond.direct_solver_blob: sine-transform diagonalisation of the rectangle + capacitance
    correction for the obstacle == the oracle's sparse LU solve of the same matrix."""
    import scipy.sparse.linalg as spl
    from sol_amd import precond
    for (Y, X) in [(128, 64), (32, 16)]:
        g = o.geometry(Y, X)
        blob = precond.direct_solver_blob(g.active)
        hdr = blob[:precond.FD_HEADER].view(np.int32)
        assert hdr[0] == precond.FD_MAGIC and (hdr[1], hdr[2]) == (Y, X) and hdr[5] <= hdr[6] <= 256
        M = (-g.pressure_matrix()).tocsc()
        rng = np.random.default_rng(0)
        b = rng.standard_normal((Y, X)) * np.asarray(g.active).reshape(Y, X)
        exact = spl.splu(M).solve(b.ravel()).reshape(Y, X)
        x = precond.direct_solve_reference(blob, b)
        assert np.linalg.norm(x - exact) < 5e-6 * np.linalg.norm(exact)      # blob is stored in fp32
    assert np.array_equal(precond.scene_matrix(o.geometry(32, 16).active), (-o.geometry(32, 16).pressure_matrix()).toarray())
    # a scene whose modified cells do not fit one 16x16 window is refused (-> CG)
    act = np.ones((128, 64)); act[10:12, 10:12] = 0; act[100:102, 40:42] = 0
    assert precond.direct_solver_blob(act) is None


def test_library_staleness_is_decided_by_source_content(tmp_path, monkeypatch):
    """_build._stale(): a shipped library is current iff the recorded hash equals the hash of the sources, whatever the
    modification times say (a tree copied without time stamps must not rebuild on every rank of a node)."""
    import os
    from sol_amd import _build
    lib = tmp_path / "libsol_hip.so"
    stamp = tmp_path / "libsol_hip.sources.sha1"
    monkeypatch.setattr(_build, "LIB", str(lib))
    monkeypatch.setattr(_build, "STAMP", str(stamp))
    assert _build._stale()                              # no library
    lib.write_bytes(b"x")
    assert _build._stale()                              # library without a stamp
    stamp.write_text(_build._source_hash() + "\n")
    assert not _build._stale()
    os.utime(lib, (0, 0))                               # older than every source: still current
    assert not _build._stale()
    stamp.write_text("0" * 40 + "\n")
    assert _build._stale()                              # sources changed since the build


def test_burgers_forcing_model_and_randfreq():
    """burgers.py:99-122 forcing model as restated in sol_amd.burgers (recalled PhiFlow semantics behind named variants):
    closed-form value of a single wave at the staggered sample points, phase advance, both variants, randfreq statistics."""
    import numpy as np
    from sol_amd.burgers import SinForces, randfreq
    rng = np.random.default_rng(3)
    f = SinForces(rng, num_forces=1)
    Y = X = 8
    dx = 0.5
    a = f.staggered(Y, X, dx)
    assert a.shape == (1, Y + 1, X + 1, 2) and np.all(a[0, :, X, 0] == 0) and np.all(a[0, Y, :, 1] == 0)
    k, amp, ph = f.k[0], f.amp[0], f.phase[0]
    j, i = 3, 5
    assert np.isclose(a[0, j, i, 0], amp[0] * np.sin(k[0] * j * dx + k[1] * (i + 0.5) * dx + ph))
    assert np.isclose(a[0, j, i, 1], amp[1] * np.sin(k[0] * (j + 0.5) * dx + k[1] * i * dx + ph))
    assert 0.8 <= np.linalg.norm(k) <= 1.6 and np.all(np.abs(amp) <= 0.15) and -0.4 <= f.omega[0] <= 0.4
    f.step(0.1)
    assert np.isclose(f.phase[0], ph + 0.1 * f.omega[0])
    g = SinForces(np.random.default_rng(3), num_forces=1, variant="gradient")
    b = g.staggered(Y, X, dx)
    assert np.isclose(b[0, j, i, 0], amp[0] * np.cos(k[0] * j * dx + k[1] * (i + 0.5) * dx + ph) * k[0])
    r = randfreq((33, 32), np.random.default_rng(0))
    assert r.shape == (33, 32) and abs(r.std() - 1) < 1e-12
    spec = np.abs(np.fft.fft2(r))
    assert spec[0:3, 0:3].sum() > 0.9 * spec.sum()              # power 8: essentially the lowest wave numbers


def test_capture_safe_torch_helpers_equal_the_torch_ops_they_replace():
    """The helpers that keep captured graphs kernel-only must be drop-in: same values AND same gradients as
    the torch forms they replace -- `_lib.pad_high` vs `F.pad` (constant_pad_nd, whose backward clone()s a narrow of the gradient),
    `ops.split_flat` vs 1-D slicing (SliceBackward = zeros + copy_), `_lib.stack0` vs `torch.stack`.  On the CPU: the same Python runs
    there (the GPU-only part is the kernel copy inside SplitFlatFn.backward / dcopy_)."""
    from sol_amd import _lib
    F = torch.nn.functional
    gen = torch.Generator().manual_seed(3)
    # pad_high: every (dims) combination the package uses
    for shape, dims, pad in (((2, 5, 4, 1), (2,), (0, 0, 0, 1)), ((2, 5, 4, 1), (1,), (0, 0, 0, 0, 0, 1)), ((3, 4, 5, 2), (1, 2), (0, 0, 0, 1, 0, 1)),
                             ((1, 6, 3, 4), (1,), (0, 0, 0, 0, 0, 1)), ((2, 4, 3, 5), (3,), (0, 1)), ((2, 5, 4), (2,), (0, 1)), ((2, 5, 4), (1,), (0, 0, 0, 1))):
        x = torch.randn(*shape, generator=gen, dtype=torch.float64)
        a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = _lib.pad_high(a, *dims), F.pad(b, pad)
        assert ya.shape == yb.shape and torch.equal(ya, yb), (shape, dims)
        w = torch.randn(ya.shape, generator=gen, dtype=torch.float64)
        (ya * w).sum().backward()
        (yb * w).sum().backward()
        assert torch.equal(a.grad, b.grad), (shape, dims)
    # split_flat: pieces are views of the flat buffer, gradient = the pieces' gradients in place (zeros where a piece is unused)
    flat = torch.randn(37, generator=gen, dtype=torch.float64)
    bounds = (0, 5, 5, 17, 30, 37)                                   # (an empty piece and a full cover)
    a, b = flat.clone().requires_grad_(True), flat.clone().requires_grad_(True)
    pa = ops.split_flat(a, bounds)
    pb = [b[bounds[k]:bounds[k + 1]] for k in range(len(bounds) - 1)]
    assert all(torch.equal(u, v) for u, v in zip(pa, pb))
    ws = [torch.randn(p.shape, generator=gen, dtype=torch.float64) for p in pb]
    (sum((u * w).sum() for u, w in zip(pa, ws) if u.numel()) - 2.0 * pa[3].sum()).backward()
    (sum((u * w).sum() for u, w in zip(pb, ws) if u.numel()) - 2.0 * pb[3].sum()).backward()
    assert torch.equal(a.grad, b.grad)
    a2 = flat.clone().requires_grad_(True)
    pa2 = ops.split_flat(a2, (3, 10, 20))                             # partial cover, one piece without a gradient
    (pa2[1] * 3.0).sum().backward()
    exp = torch.zeros(37, dtype=torch.float64)
    exp[10:20] = 3.0
    assert torch.equal(a2.grad, exp)
    with torch.no_grad():                                             # no-grad / non-differentiable inputs: plain views
        v = ops.split_flat(a2, (0, 4, 37))
    assert v[0].data_ptr() == a2.data_ptr() and v[0].grad_fn is None
    # the networks' tensors() go through it: in-place edits under no_grad still reach the flat buffer
    net = sol_amd.model_mercury(seed=1, device="cpu")
    with torch.no_grad():
        net.tensors()[1].fill_(0.25)
    assert float(net.params[net.offsets[1]:net.offsets[2]].mean()) == 0.25
    # stack0
    one = [torch.tensor(2.5, dtype=torch.float64, requires_grad=True)]
    s1 = _lib.stack0(one)
    assert s1.shape == (1,) and torch.equal(s1.detach(), torch.stack(one).detach())
    s1.sum().backward()
    assert float(one[0].grad) == 1.0
    many = [torch.tensor(float(k)) for k in range(4)]
    assert torch.equal(_lib.stack0(many), torch.stack(many))


def test_cpu_affinity_binding_follows_the_gpus_numa_node(tmp_path, monkeypatch):
    """dist.bind_cpu_affinity (first-run hardening for a multi-GPU node, SURVEY.md section 8e: host jitter between graph replays): the
    process is pinned to the CPUs of its GPU's NUMA node as sysfs reports them, ranks that share a node split it, and every way the
    platform can fail to answer leaves the process unbound WITH a reason -- on a fake sysfs tree (no GPU here)."""
    import os
    from sol_amd import dist as sd
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity")
    mine = sorted(os.sched_getaffinity(0))
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("%d-%d\n" % (mine[0], mine[-1]))
    calls = []
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: calls.append(sorted(cpus)))
    monkeypatch.setattr(sd, "gpu_numa_node", lambda idx, sysfs="/sys": 1)
    rec = sd.bind_cpu_affinity(0, sysfs=str(tmp_path))
    assert rec["bound"] and rec["numa_node"] == 1 and calls[-1] == mine and rec["inherited"] == mine and sd.AFFINITY is rec
    if len(mine) >= 4:                                  # two local ranks on one node: disjoint halves
        a = sd.bind_cpu_affinity(0, 0, 2, sysfs=str(tmp_path)); ca = calls[-1]
        b = sd.bind_cpu_affinity(0, 1, 2, sysfs=str(tmp_path)); cb = calls[-1]
        assert a["bound"] and b["bound"] and not set(ca) & set(cb) and len(ca) == len(cb) == len(mine) // 2
    n = len(calls)
    monkeypatch.setattr(sd, "gpu_numa_node", lambda idx, sysfs="/sys": None)
    rec = sd.bind_cpu_affinity(0, sysfs=str(tmp_path))
    assert not rec["bound"] and "NUMA" in rec["why"] and len(calls) == n
    monkeypatch.setattr(sd, "gpu_numa_node", lambda idx, sysfs="/sys": 7)          # a node sysfs does not list
    rec = sd.bind_cpu_affinity(0, sysfs=str(tmp_path))
    assert not rec["bound"] and rec["why"].startswith("failed") and len(calls) == n
    monkeypatch.setenv("SOL_NO_AFFINITY", "1")
    assert sd.bind_cpu_affinity(0, sysfs=str(tmp_path))["why"] == "SOL_NO_AFFINITY set" and len(calls) == n
    assert sd._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    sd.AFFINITY = None


def test_dcopy_validates_its_operands_before_touching_the_library():
    """_lib.dcopy_ (the kernel copy used inside stream captures) refuses what the C entry point cannot take BEFORE any library call -- a
    dtype that is not 32-bit, host tensors, unequal sizes -- instead of failing inside ptr() in the middle of a capture."""
    from sol_amd import _lib
    a, b = torch.zeros(6), torch.zeros(2, 3)
    with pytest.raises(_lib.SolError, match="CUDA"):
        _lib.dcopy_(a, b)                                # host tensors
    with pytest.raises(_lib.SolError):
        _lib.dcopy_(torch.zeros(6), torch.zeros(5))      # sizes
    with pytest.raises(_lib.SolError):
        _lib.dcopy_(torch.zeros(4, dtype=torch.float64), torch.zeros(4, dtype=torch.float64))
