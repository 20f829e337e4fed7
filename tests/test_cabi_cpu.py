"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/sol_hip.h
declares, and rejects bad arguments with an error code + message (no compute, no GPU)."""
import ctypes as C
import os

import numpy as np
import pytest

import sol_amd
from sol_amd import _lib, ops


@pytest.fixture(scope="module")
def lib():
    return sol_amd.load()


def test_library_is_in_tree_and_exports_all_declared_symbols(lib):
    path = os.path.realpath(sol_amd.lib_path())
    assert path.startswith(os.path.realpath(os.path.dirname(sol_amd.__file__)))
    decl = sol_amd.declared_symbols()
    assert len(decl) >= 18
    for s in decl:
        assert hasattr(lib, s), "libsol_hip.so does not export %s" % s
    assert set(decl) == set(_lib._SIGS.keys())      # every declared function has a typed binding
    assert lib.sol_version() >= 100


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sol_amd.SolError):
        _lib.require_gpu()
    with pytest.raises(sol_amd.SolError):
        cfg = ops.karman_cfg(1, 16, 8, 12.5)
        z = torch.zeros(1, 16, 8)
        ops.karman_step(z, torch.zeros(1, 17, 8), torch.zeros(1, 16, 9), torch.ones(1), cfg, None)


def test_bad_arguments_are_rejected_with_a_message(lib):
    cfg = ops.karman_cfg(1, 16, 8, 12.5)
    # NULL pointers
    rc = lib.sol_karman_step_fwd(C.byref(cfg), None, *([None] * 8), 0, *([None] * 6), None, None)
    assert rc == -1 and b"NULL" in lib.sol_last_error()
    # unsupported grid (X must be 8/16/32/64, Y a multiple of 8)
    bad = ops.karman_cfg(1, 16, 12, 1.0)
    rc = lib.sol_karman_step_fwd(C.byref(bad), None, *([None] * 8), 0, *([None] * 6), None, None)
    assert rc == -1 and b"X must be" in lib.sol_last_error()
    bad = ops.karman_cfg(1, 20, 8, 1.0)
    rc = lib.sol_karman_step_bwd(C.byref(bad), None, *([None] * 5), 0, *([None] * 3), None, *([None] * 3))
    assert rc == -1 and b"multiple of 8" in lib.sol_last_error()
    # grid larger than one workgroup / the LDS
    big = ops.karman_cfg(1, 256, 64, 1.0)
    rc = lib.sol_karman_step_fwd(C.byref(big), None, *([None] * 8), 0, *([None] * 6), None, None)
    assert rc == -1
    # conv: bad channel count
    rc = lib.sol_conv5x5(None, None, None, None, None, None, None, 1, 16, 8, 5, 32, 0, 0.3)
    assert rc == -1 and b"input channels" in lib.sol_last_error()
    # std == 0 (the reference's `-n 1` division by zero, karman-2d/Makefile:73)
    tc = _lib.TrainCfg(ops.karman_cfg(1, 16, 8, 12.5), 2, 0.2, 0.2, 0.0, 0.3)
    one = C.c_void_p(1)
    rc = lib.sol_train_fwd_bwd(C.byref(tc), None, *([one] * 9), 0, one, one, one, 0, *([one] * 2), *([None] * 5))
    assert rc == -1 and b"std" in lib.sol_last_error()
    # workspace too small
    tc = _lib.TrainCfg(ops.karman_cfg(1, 16, 8, 12.5), 2, 0.2, 0.2, 1.0, 0.3)
    rc = lib.sol_train_fwd_bwd(C.byref(tc), None, *([one] * 9), 0, one, one, one, 16, *([one] * 2), *([None] * 5))
    assert rc == -3 and b"workspace" in lib.sol_last_error()


def test_workspace_sizes_and_layer_table(lib):
    tc = _lib.TrainCfg(ops.karman_cfg(6, 128, 64, 1.5625), 32, 0.2, 0.2, 1.0, 0.3)
    nb = lib.sol_train_workspace_bytes(C.byref(tc))
    # dominated by 11 x 32-channel activations per sim-step: 32*11*6*8192*32*4 B = 2.2 GB
    # + the kept pre-activation gradients (same size) for the batched weight gradient
    # + 0.8 GB of hand-off regions of the persistent CNN launches (64 x 12.6 MB; carved whatever the option says)
    assert 4.4e9 < nb < 5.2e9
    assert lib.sol_rollout_workspace_bytes(C.byref(tc)) < nb / 20
    import torch
    net = sol_amd.model_mars_moon(cin=3, cout=2, device="cpu")
    assert net.n_params == 260354
    for l in range(12):
        k, b = C.c_int64(), C.c_int64()
        ci, co = C.c_int32(), C.c_int32()
        assert lib.sol_mars_moon_layer(l, C.byref(k), C.byref(b), C.byref(ci), C.byref(co)) == 0
        assert k.value == net.offsets[2 * l] and b.value == net.offsets[2 * l + 1]
        assert net.shapes[2 * l] == (5, 5, ci.value, co.value)
    assert lib.sol_mars_moon_layer(12, None, None, None, None) == -1
    assert lib.sol_conv5x5_packed_floats(3, 32, 0) == 25 * 4 * 32
    assert lib.sol_conv5x5_packed_floats(2, 3, 1) == 25 * 4 * 16


def test_conv3d_sizes_and_options(lib):
    """Host-side arithmetic of the Conv3D entry points (no GPU): packed-buffer sizes incl. the one-launch kernels' weight sections
    (32 -> 32: 32 rows per plane, 32 -> <= 16: 16 rows; other shapes: five 2-D sections only), the weight-gradient workspace (five
    partial buffers, sized for the largest of the D-2 / D-1 / D plane passes) and the 3-D option defaults."""
    al = lambda v: (v + 63) // 64 * 64
    per = lambda ci, co: al(lib.sol_conv5x5_packed_floats(ci, co, 0))
    sect = lambda op: 4 + 125 * 2 * op * 16
    assert lib.sol_conv3d_packed_floats(32, 32) == 5 * per(32, 32) + sect(32)
    assert lib.sol_conv3d_packed_floats(32, 3) == 5 * per(32, 3) + sect(16)
    assert lib.sol_conv3d_packed_floats(32, 4) == 5 * per(32, 4) + sect(16)
    assert lib.sol_conv3d_packed_floats(4, 32) == 5 * per(4, 32)
    ws = lib.sol_conv3d_bwd_weight_ws_floats(1, 128, 64, 64, 32, 32)
    assert ws % 5 == 0 and ws // 5 >= 256 * (25 * 1024 + 32) > 0             # >= one [25][32][32] + bias partial per CU
    assert lib.sol_conv3d_bwd_weight_ws_floats(3, 128, 64, 64, 32, 32) == ws    # simulations accumulate onto the same partial blocks
    assert lib.sol_conv3d_bwd_weight_ws_floats(1, 16, 16, 16, 32, 32) <= ws
    for k, v in {"k3d_conv_fused": 1, "k3d_conv_rows": 8, "k3d_tile": 0, "k3d_fused_tf": 1}.items():
        assert _lib.get_option(k) == v, k
    with pytest.raises(sol_amd.SolError, match="must be in"):
        _lib.set_option("k3d_conv_rows", 9)


def test_options_table_and_abi_checks(lib):
    """sol_set_option / sol_get_option (no GPU needed): defaults, range and name checks; the library never reads the
    environment, the Python loader forwards SOL_* debugging overrides once."""
    assert lib.sol_version() == _lib.ABI_VERSION
    kc, bc, tc = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.sol_abi_sizes(C.byref(kc), C.byref(bc), C.byref(tc)) == 0
    assert (kc.value, bc.value, tc.value) == (C.sizeof(_lib.KarmanCfg), C.sizeof(_lib.BurgersCfg), C.sizeof(_lib.TrainCfg))
    defaults = {"conv_precision": 0, "cnn_persistent": 0, "bww_fuse": 1, "correct_fuse": 1, "density_mode": 0, "streams": 1, "bww_chunk": 0}
    for k, v in defaults.items():
        if not any(os.environ.get(e) for e, (o_, _) in _lib._ENV_OPTIONS.items() if o_ == k):
            assert _lib.get_option(k) == v, k
    _lib.set_option("conv_precision", 2)
    assert _lib.get_option("conv_precision") == 2
    _lib.set_option("conv_precision", 0)
    with pytest.raises(sol_amd.SolError, match="unknown option"):
        _lib.set_option("bogus", 1)
    with pytest.raises(sol_amd.SolError, match="must be in"):
        _lib.set_option("streams", 99)
    # no getenv left in the library sources
    csrc = os.path.join(os.path.dirname(sol_amd.__file__), "csrc")
    for f in os.listdir(csrc):
        with open(os.path.join(csrc, f)) as fh:
            assert "getenv" not in fh.read(), f


def test_sol32_fixture_is_the_bench_workload(golden_dir):
    """The committed SOL-32 fixture was generated from oracle.bench_workload at BASELINE configs[2]; its three-step Adam
    trajectory at lr 1e-4 is the one an independent float64 run of the same workload gave (round-1 verdict)."""
    z = np.load(os.path.join(golden_dir, "train_128x64_sol32.npz"))
    assert (int(z["B"]), int(z["Y"]), int(z["X"]), int(z["msteps"])) == (6, 128, 64, 32) and float(z["lr"]) == 1e-4
    assert np.allclose(z["loss_traj"], [2386.489, 118279.49, 11700.69], rtol=1e-6)
    assert abs(z["loss_steps"].sum() / 32 - z["loss_traj"][0]) < 1e-9 * z["loss_traj"][0]
    assert z["grads_sub16"].shape == ((260354 + 15) // 16,) and z["grad_norms"].shape == (24,)


def test_round5_entry_points_reject_bad_arguments_and_options_have_their_defaults(lib):
    """sol_graph_check / sol_graph_census / sol_copy_words / sol_conv3d_bwd_weight_acc validate their arguments before they touch the HIP
    runtime (no GPU needed), SOL_ERR_GRAPH is a code of its own, the node-type names are those of hipGraphNodeType, and the options
    added in round 5 have their documented defaults and ranges."""
    assert lib.sol_graph_check(None, b"x") == -1 and b"NULL graph" in lib.sol_last_error()
    counts = (C.c_int32 * 8)()
    assert lib.sol_graph_census(None, counts, 8) == -1
    assert lib.sol_graph_census(C.c_void_p(1), counts, 0) == -1 and lib.sol_graph_census(C.c_void_p(1), counts, 65) == -1
    assert lib.sol_graph_node_type_name(0) == b"kernel" and lib.sol_graph_node_type_name(1) == b"memcpy" and lib.sol_graph_node_type_name(2) == b"memset"
    one = C.c_void_p(4)
    assert lib.sol_copy_words(None, None, one, 4) == -1                     # no destination
    assert lib.sol_copy_words(None, one, one, -1) == -1                     # negative count
    assert lib.sol_copy_words(None, C.c_void_p(6), one, 4) == -1            # destination not 4-byte aligned
    assert lib.sol_copy_words(None, one, one, 0) == 0 and lib.sol_copy_words(None, one, one, 16) == 0      # nothing to do / dst == src: no launch
    rc = lib.sol_conv3d_bwd_weight_acc(None, None, one, None, None, one, one, one, one, 1, 8, 8, 64, 32, 32, 32, 32, 0, 1)
    assert rc == -1 and b"NULL pointer" in lib.sol_last_error()
    rc = lib.sol_conv3d_bwd_weight_acc(None, one, one, None, None, one, one, one, one, 1, 2, 8, 64, 32, 32, 32, 32, 1, 0)
    assert rc == -1 and b"bad shape" in lib.sol_last_error()                # D < 3
    assert _lib.get_option("conv_thin_valu") == 1
    with pytest.raises(sol_amd.SolError):
        _lib.set_option("conv_thin_valu", 3)
    for v in (0, 2, 1):
        _lib.set_option("conv_thin_valu", v)
        assert _lib.get_option("conv_thin_valu") == v
    # the header documents every option name the table knows that a host may want to flip
    hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "sol_hip.h")).read()
    for name in ("conv_precision", "conv_dx", "conv_thin_valu", "cnn_persistent", "k3d_conv_rows"):
        assert name in hdr, name
    assert "SOL_ERR_GRAPH (-4)" in hdr


def test_round6_karman3d_entry_points_reject_bad_arguments_and_options_have_their_defaults(lib):
    """The depth-packed thin-layer entry points (sol_conv3d_thin_bwd_weight_acc, sol_conv3d_thin_out_bwd_weight_acc) validate their arguments
    before any launch (no GPU needed); the karman-3d options added in round 6 have their documented defaults and ranges and are in the header."""
    one = C.c_void_p(4)
    rc = lib.sol_conv3d_thin_out_bwd_weight_acc(None, None, None, one, one, one, one, one, 1, 8, 8, 64, 3, 0, 1)
    assert rc == -1 and b"NULL pointer" in lib.sol_last_error()
    rc = lib.sol_conv3d_thin_out_bwd_weight_acc(None, one, None, one, one, one, one, one, 1, 8, 8, 32, 3, 0, 1)
    assert rc == -1 and b"W == 64" in lib.sol_last_error()                  # the depth-packed form runs the 64-pixel-row kernels only
    rc = lib.sol_conv3d_thin_out_bwd_weight_acc(None, one, None, one, one, one, one, one, 1, 8, 8, 64, 5, 0, 1)
    assert rc == -1 and b"output channels" in lib.sol_last_error()          # more than four output channels is not a thin layer
    rc = lib.sol_conv3d_thin_bwd_weight_acc(None, None, one, None, one, one, one, one, 1, 8, 8, 64, 3, 0, 1)
    assert rc == -1
    assert lib.sol_conv3d_thin_bwd_weight_ws_floats(1, 8, 8, 64) > 25 * 32 * 32
    assert _lib.get_option("k3d_bww_jobs") == 2 and _lib.get_option("k3d_conv_persist") == 0
    with pytest.raises(sol_amd.SolError):
        _lib.set_option("k3d_bww_jobs", 3)
    for name, vals in (("k3d_bww_jobs", (0, 1, 2)), ("k3d_conv_persist", (1, 0))):
        for v in vals:
            _lib.set_option(name, v)
            assert _lib.get_option(name) == v
    hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "sol_hip.h")).read()
    for name in ("k3d_bww_jobs", "k3d_conv_persist", "sol_conv3d_thin_out_bwd_weight_acc"):
        assert name in hdr, name
