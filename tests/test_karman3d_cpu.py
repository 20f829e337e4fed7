"""karman-3d, CPU side: the oracle reproduces its committed fixture, the product's host logic (scene masks, direct-solver
blob of precond3d) agrees with the oracle, and the C ABI exports the 3-D entry points (no compute calls without a GPU)."""
import os
import sys

import numpy as np
import pytest
import torch

import sol_oracle3d as o
import sol_amd
from sol_amd import karman3d as k3, precond3d as p3

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

torch.set_default_dtype(torch.float64)


def rel(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-300))


def test_oracle_reproduces_the_3d_fixture(golden_dir):
    import make_golden as mg
    z = np.load(os.path.join(golden_dir, "karman3d_32x16x16.npz"))
    t = lambda k: torch.as_tensor(z[k].astype(np.float64))
    B, Y, X, Z = z["d"].shape
    g = o.geometry(Y, X, Z)
    v = (t("vy"), t("vx"), t("vz"))
    # the stored inputs are the seeded generator's output (regenerable)
    d0, v0 = o.synthetic_state(B, Y, X, Z, 77)
    assert rel(d0, z["d"]) < 1e-7 and rel(v0[1], z["vx"]) < 1e-7
    d1, v1 = o.karman3d_step(t("d"), v, t("re"), g)
    for a, k in ((d1, "d_out"), (v1[0], "vy_out"), (v1[1], "vx_out"), (v1[2], "vz_out")):
        assert rel(a, z[k]) < 2e-7
    params = mg.k3d_params()
    std_v = tuple(float(s) for s in z["std_v"])
    with torch.no_grad():
        feat = o.to_feature(v1, t("re")) / torch.tensor(list(std_v) + [float(z["std_re"])])
        assert rel(o.mars_moon3d(params, feat), z["net_out"]) < 2e-7
        dr, vr = o.rollout(params, t("d"), v, t("re"), g, std_v, float(z["std_re"]), int(z["nroll"]))[-1]
    assert np.allclose([float(a.norm()) for a in (dr,) + tuple(vr)], z["roll_norms"], rtol=1e-9)
    assert rel(vr[0].reshape(-1)[::4], z["roll_vy_sub4"]) < 2e-7 and rel(dr.reshape(-1)[::4], z["roll_d_sub4"]) < 2e-7


def test_product_scene_equals_oracle_geometry():
    for (Y, X, Z) in ((32, 16, 16), (128, 64, 64)):
        g = o.geometry(Y, X, Z)
        active, inflow = k3.scene_arrays3d(Y, X, Z)
        assert np.array_equal(active, g.active) and np.array_equal(inflow, g.inflow)
        bcv, bcm = k3.velocity_bc_masks3d(Y, X, Z)
        assert np.array_equal(bcv, g.bc_mask) and np.array_equal(bcm, g.bc_mask)
    with pytest.raises(NotImplementedError):
        k3.scene_arrays3d(32, 16, 16, obstacle="cylinder")
    with pytest.raises(ValueError):
        k3.scene_arrays3d(32, 16, 8)


def test_direct_solver_blob_solves_the_oracle_system():
    g = o.geometry(32, 16, 16)
    blob = p3.direct_solver_blob3d(g.active)
    hdr = blob[:16].view(np.int32)
    assert hdr[0] == p3.FD3_MAGIC and tuple(hdr[1:4]) == (32, 16, 16) and hdr[4] == 32 and hdr[5] == 64   # 8 obstacle cells + 24 neighbours
    assert blob.size == 16 + 32 * 32 + 2 * 16 * 16 + 32 * 16 * 16 + 64 * 64 + 64
    rng = np.random.default_rng(1)
    b = rng.standard_normal((32, 16, 16))
    x = p3.direct_solve_reference3d(blob, b)
    p = o._PressureSolve.apply(torch.as_tensor(-b)[None], g)[0].numpy()        # A p = -b  <=>  M p = b
    assert np.abs(x - p).max() < 1e-6 * np.abs(p).max()                        # fp32 storage of the blob
    # no obstacle at all: nS = 0, the plain sine-transform solve
    blob0 = p3.direct_solver_blob3d(np.ones((16, 8, 8)))
    assert blob0[:16].view(np.int32)[4] == 0
    x0 = p3.direct_solve_reference3d(blob0, b[:16, :8, :8])
    assert np.allclose(x0, o.rect_solve(b[None, :16, :8, :8], o._rect_eigenvalues((16, 8, 8)))[0], atol=1e-5)


def test_direct_solver_blob_at_full_size_has_small_residual():
    g = o.geometry(128, 64, 64)
    blob = p3.direct_solver_blob3d(g.active)
    hdr = blob[:16].view(np.int32)
    assert hdr[4] == 1568 and hdr[5] == 1600
    rng = np.random.default_rng(2)
    b = rng.standard_normal((128, 64, 64))
    x = p3.direct_solve_reference3d(blob, b)
    r = -o.apply_A(torch.as_tensor(x)[None], g)[0].numpy() - b
    assert np.abs(r).max() < 2e-6 * np.abs(b).max()


def test_cabi_exports_the_3d_entry_points():
    lib = sol_amd.load()
    for name in ("sol_karman3d_step_workspace_bytes", "sol_karman3d_step_fwd", "sol_karman3d_correct", "sol_conv3d_packed_floats",
                 "sol_conv3d_pack", "sol_conv3d", "sol_abi_size_karman3d"):
        assert name in sol_amd.declared_symbols() and hasattr(lib, name)
    import ctypes as C
    cfg = sol_amd._lib.Karman3DCfg(2, 128, 64, 64, 1.5625, 1.0, 64.0, 0, 0, 0, None)
    assert lib.sol_abi_size_karman3d() == C.sizeof(cfg)
    nb = lib.sol_karman3d_step_workspace_bytes(C.byref(cfg))
    assert 6 * 2 * 128 * 64 * 64 * 4 < nb < 6.3 * 2 * 128 * 64 * 64 * 4          # three components + three cell buffers
    assert lib.sol_conv3d_packed_floats(32, 32) >= 5 * lib.sol_conv5x5_packed_floats(32, 32, 0)
    assert sol_amd._lib.get_option("k3d_tile") == 0
