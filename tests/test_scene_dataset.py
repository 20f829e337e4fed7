"""Scene / dataset I/O on the input edge of the hot path (host side, CPU)."""
import os
import pickle

import numpy as np

from sol_amd import scene


def _fake_set(tmp, nsims=2, frames=5, Y=32, X=16):
    rng = np.random.default_rng(0)
    for s in range(nsims):
        p = scene.scene_create(str(tmp))
        with open(p + "/params.pickle", "wb") as f:
            pickle.dump({"re": 1e5 * (s + 1)}, f)
        for i in range(frames):
            scene.scene_write(p, [rng.random((1, Y, X, 1)), rng.random((1, Y + 1, X + 1, 2))], ["dens", "velo"], i)


def test_zipped_array_roundtrip_and_disk_layout(tmp_path):
    rng = np.random.default_rng(1)
    v = rng.random((1, 9, 5, 2))
    scene.write_zipped_array(str(tmp_path / "v.npz"), v)
    assert np.array_equal(scene.read_zipped_array(str(tmp_path / "v.npz")), v)
    raw = np.load(tmp_path / "v.npz")["arr_0"]
    assert raw.shape == (9, 5, 2) and np.array_equal(raw[..., ::-1], v[0])     # batch dropped, components reversed
    d = rng.random((1, 8, 4, 1))
    scene.write_zipped_array(str(tmp_path / "d.npz"), d)
    assert np.load(tmp_path / "d.npz")["arr_0"].shape == (8, 4, 1)
    assert np.array_equal(scene.read_zipped_array(str(tmp_path / "d.npz")), d)


def test_downsampling_centered_and_staggered():
    d = np.arange(1 * 4 * 4 * 1, dtype=np.float64).reshape(1, 4, 4, 1)
    assert np.allclose(scene.downsample(d, 2)[0, :, :, 0], [[2.5, 4.5], [10.5, 12.5]])
    # a uniform staggered field stays uniform on the kept faces, padding stays zero
    st = np.zeros((1, 9, 5, 2))
    st[:, :, :4, 0] = 3.0
    st[:, :8, :, 1] = -2.0
    ds = scene.downsample_staggered(st, 4)
    assert ds.shape == (1, 3, 2, 2)
    assert np.allclose(ds[:, :, :1, 0], 3.0) and np.allclose(ds[:, :2, :, 1], -2.0)
    assert np.all(ds[:, :, 1, 0] == 0) and np.all(ds[:, 2, :, 1] == 0)
    # divergence of a linear field is preserved in the mean: v_y = y  ->  faces at 2x spacing
    st = np.zeros((1, 9, 5, 2))
    st[:, :, :4, 0] = np.arange(9)[None, :, None]
    ds = scene.downsample2x_staggered(st)
    assert np.allclose(ds[0, :, 0, 0], [0, 2, 4, 6, 8])


def test_phif_dataset_stats_and_batches(tmp_path):
    _fake_set(tmp_path)
    logs = []
    ds = scene.PhifDataset(str(tmp_path), 5, 2, 2, print_fn=logs.append, scale=4)
    assert ds.resolution == (8, 4) and ds.numOfBatchs == 1
    assert os.path.isfile(ds.filenameToDownscaled(ds.pathsDen[0][0]))
    # dataStats: std of ABSOLUTE values (karman_train.py:234-255); ext.std = std of the Reynolds numbers
    alld = np.concatenate([np.abs(ds.dataPreloaded[s][i][0]).ravel() for s in ds.dataSims for i in range(5)])
    assert np.isclose(ds.dataStats["std"][0], np.std(alld))
    assert np.isclose(ds.dataStats["ext.std"][0], 5e4)
    ds.newEpoch(exclude_tail=2)
    assert ds.numOfSteps == 3 and len(ds.epoch) == 2 and all(len(e) == 3 for e in ds.epoch)
    dens, velo, ext = ds.getData(consecutive_frames=2)
    assert len(dens) == 3 and dens[0].shape == (2, 8, 4, 1) and velo[2].shape == (2, 9, 5, 2) and len(ext) == 2
    # consecutive frames of the SAME simulation
    s, f = ds.epoch[0][0]
    assert np.array_equal(dens[1][0], ds.dataPreloaded[ds.dataSims[s]][f + 1][0][0])
    vy, vx = scene.split_staggered(velo[0])
    assert vy.shape == (2, 9, 4) and vx.shape == (2, 8, 5)
    # skip_preprocessing reuses the cached ds_ files
    ds2 = scene.PhifDataset(str(tmp_path), 5, 2, 2, print_fn=logs.append, skip_preprocessing=True, scale=4)
    assert np.array_equal(ds2.dataPreloaded[ds2.dataSims[0]][0][1], ds.dataPreloaded[ds.dataSims[0]][0][1])


def test_resident_frames_gather_equals_host_assembly(tmp_path):
    """ResidentFrames (the set kept in device memory): the device-side gather of (sim, frame) windows fills the
    trainer's buffers exactly as the reference-shaped host path (getData + split + stack) does.  (device = cpu here.)"""
    import torch
    _fake_set(tmp_path, nsims=3, frames=7)
    ds = scene.PhifDataset(str(tmp_path), 7, 3, 3, print_fn=lambda *a: None, scale=1)
    ms = 3
    ds.newEpoch(exclude_tail=ms)
    ds.nextStep()
    res = scene.ResidentFrames(ds, "cpu")
    Y, X = ds.resolution
    B = 2                                         # this "rank" owns simulations 1..2 of the batch of 3
    f32 = lambda *s: torch.zeros(*s, dtype=torch.float32)
    d0, vy0, vx0, re = f32(B, Y, X), f32(B, Y + 1, X), f32(B, Y, X + 1), f32(B)
    gvy, gvx = f32(ms, B, Y + 1, X), f32(ms, B, Y, X + 1)
    res.gather(ds.selection()[1:3], ms, d0, vy0, vx0, re, gvy, gvx)
    dens, velo, ext = ds.getData(consecutive_frames=ms)
    vy, vx = zip(*[scene.split_staggered(v[1:3]) for v in velo])
    assert np.array_equal(d0.numpy(), dens[0][1:3, ..., 0]) and np.array_equal(vy0.numpy(), vy[0]) and np.array_equal(vx0.numpy(), vx[0])
    assert np.array_equal(gvy.numpy(), np.stack(vy[1:])) and np.array_equal(gvx.numpy(), np.stack(vx[1:]))
    assert np.allclose(re.numpy(), np.asarray(ext[1:3], dtype=np.float32))
    assert res.bytes == 3 * 7 * 4 * (Y * X + (Y + 1) * X + Y * (X + 1))
