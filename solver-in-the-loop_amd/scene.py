"""Scene / dataset I/O of the training scripts (host side, numpy): the input edge of the hot path.

Mirrors what the reference scripts use of PhiFlow's Scene API and their own PhifDataset:
  read_zipped_array / write_zipped_array  (`<name>_%06d.npz`, key arr_0, batch dim dropped when 1,
      component axis reversed on disk -- evidence: karman-2d/karman.py:104 "read_zipped_array reverts
      indices", karman_train_pre.py:88-104,170)                                  [EXT-RECALL A.12]
  downsample4x / downsample4xSMAC  (/root/reference/karman-2d/karman_train.py:140-144) [EXT-RECALL A.11]
  PhifDataset  (karman_train.py:187-337): preload, 4x down-scaling cache (`ds_` files), dataStats =
      std of ABSOLUTE values, shuffled (sim, frame) pairs, msteps+1 consecutive frames per sample.
Layouts: density [1,Y,X,1], velocity = staggered tensor [1,Y+1,X+1,2] (component 0 = y).
"""
import glob
import os
import pickle
import random

import numpy as np


def read_zipped_array(filename):
    f = np.load(filename)
    array = f[f.files[-1]]
    if array.shape[0] != 1 or array.ndim == 1:
        array = np.expand_dims(array, axis=0)
    if array.shape[-1] != 1:
        array = array[..., ::-1]                     # component order is reversed on disk
    return np.ascontiguousarray(array)


def write_zipped_array(filename, array):
    array = np.asarray(array)
    if array.shape[0] == 1 and array.ndim > 1:
        array = array[0, ...]
    if array.shape[-1] != 1:
        array = array[..., ::-1]
    # atomic: several processes (ranks of a data-parallel job, a second run on the same directory) may look at the
    # same ds_*.npz path; a reader must never see a half-written archive
    if not filename.endswith(".npz"):
        filename += ".npz"
    tmp = os.path.join(os.path.dirname(filename) or ".", ".tmp%d_%s" % (os.getpid(), os.path.basename(filename)))
    np.savez_compressed(tmp, array)
    os.replace(tmp, filename)


def _dist_rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def preprocess_on_rank0(work):
    """Runs `work()` (the down-scaling pass that writes ds_*.npz next to the raw frames) on rank 0 only and makes
    every other rank of a data-parallel job wait for it: all ranks share one dataset directory."""
    rank, world = _dist_rank_world()
    if rank == 0:
        work()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def scene_dir(base, index):
    return os.path.join(base, "sim_%06d" % index)


def scene_create(base):
    """Scene.create(directory): next free `sim_%06d` under base."""
    os.makedirs(base, exist_ok=True)
    idx = 0
    while os.path.exists(scene_dir(base, idx)):
        idx += 1
    path = scene_dir(base, idx)
    os.makedirs(path)
    return path


def scene_write(path, arrays, names, frame):
    for a, n in zip(arrays, names):
        write_zipped_array(os.path.join(path, "%s_%06d.npz" % (n, frame)), a)


# ---- down-sampling ------------------------------------------------------------------------------
def downsample2x(t):
    """math.downsample2x: mean over 2x2 blocks, [B,Y,X,C] with even Y, X."""
    return 0.25 * (t[:, 0::2, 0::2] + t[:, 1::2, 0::2] + t[:, 0::2, 1::2] + t[:, 1::2, 1::2])


def downsample2x_staggered(st):
    """StaggeredGrid(t).downsample2x().staggered_tensor() for a staggered tensor [B,Y+1,X+1,2]:
    every second face along the component's own axis, pairs averaged along the other axis."""
    B, Yp, Xp, _ = st.shape
    Y, X = Yp - 1, Xp - 1
    vy = st[:, :, :X, 0]                    # [B,Y+1,X]
    vx = st[:, :Y, :, 1]                    # [B,Y,X+1]
    vy2 = 0.5 * (vy[:, 0::2, 0::2] + vy[:, 0::2, 1::2])          # [B,Y/2+1,X/2]
    vx2 = 0.5 * (vx[:, 0::2, 0::2] + vx[:, 1::2, 0::2])          # [B,Y/2,X/2+1]
    out = np.zeros((B, Y // 2 + 1, X // 2 + 1, 2), dtype=st.dtype)
    out[:, :, :X // 2, 0] = vy2
    out[:, :Y // 2, :, 1] = vx2
    return out


def downsample(t, scale):
    while scale > 1:
        t = downsample2x(t)
        scale //= 2
    return t


def downsample_staggered(t, scale):
    while scale > 1:
        t = downsample2x_staggered(t)
        scale //= 2
    return t


def split_staggered(v):
    """[B,Y+1,X+1,2] -> (v_y [B,Y+1,X], v_x [B,Y,X+1])  (unstack_staggered_tensor)"""
    return np.ascontiguousarray(v[:, :, :-1, 0]), np.ascontiguousarray(v[:, :-1, :, 1])


# ---- dataset ------------------------------------------------------------------------------------
class PhifDataset:
    """karman_train.py:187-337 (same attribute / method names)."""

    def __init__(self, dirpath, num_frames, num_sims=None, batch_size=1, print_fn=print, skip_preprocessing=False, scale=4):
        self.dataSims = sorted(glob.glob(dirpath + "/sim_0*"))[0:num_sims]
        self.pathsDen = [sorted(glob.glob(s + "/dens_0*.npz")) for s in self.dataSims]
        self.pathsVel = [sorted(glob.glob(s + "/velo_0*.npz")) for s in self.dataSims]
        self.dataFrms = [np.arange(num_frames) for _ in self.dataSims]
        self.batchSize = batch_size
        self.epoch, self.epochIdx, self.batch, self.batchIdx, self.step, self.stepIdx = None, 0, None, 0, None, 0
        self.printFn = print_fn
        self.numOfSims = len(self.dataSims) if num_sims is None else num_sims
        self.numOfBatchs = self.numOfSims // self.batchSize
        self.numOfFrames = num_frames
        self.numOfSteps = num_frames
        if not skip_preprocessing and scale > 1:
            def work():
                self.printFn("Pre-processing: Loading data from {} = {} and save down-scaled data".format(dirpath, self.dataSims))
                for j in range(len(self.dataSims)):
                    for i in range(num_frames):
                        for paths, fn in ((self.pathsDen, downsample), (self.pathsVel, downsample_staggered)):
                            dst = self.filenameToDownscaled(paths[j][i])
                            if not os.path.isfile(dst):
                                write_zipped_array(dst, fn(read_zipped_array(paths[j][i]), scale))
            preprocess_on_rank0(work)
        name = self.filenameToDownscaled if scale > 1 else (lambda p: p)
        self.printFn("Preload: Loading data from {} = {}".format(dirpath, self.dataSims))
        self.dataPreloaded = {
            s: [(read_zipped_array(name(self.pathsDen[j][i])).astype(np.float32),
                 read_zipped_array(name(self.pathsVel[j][i])).astype(np.float32)) for i in range(num_frames)]
            for j, s in enumerate(self.dataSims)}
        self.resolution = self.dataPreloaded[self.dataSims[0]][0][0].shape[1:3]
        cat = lambda k, sl: np.concatenate([np.absolute(self.dataPreloaded[s][i][k][sl].reshape(-1))
                                            for s in self.dataSims for i in range(num_frames)])
        self.dataStats = {"std": (np.std(cat(0, Ellipsis)), (np.std(cat(1, (Ellipsis, 0))), np.std(cat(1, (Ellipsis, 1)))))}
        self.extConstChannelPerSim = {}
        for s in self.dataSims:
            with open(s + "/params.pickle", "rb") as f:
                self.extConstChannelPerSim[s] = [pickle.load(f)["re"]]
        self.dataStats["ext.std"] = [np.std([np.absolute(self.extConstChannelPerSim[s][0]) for s in self.dataSims])]
        self.printFn(self.dataStats)

    def filenameToDownscaled(self, fname):
        return os.path.dirname(fname) + "/ds_" + os.path.basename(fname)

    def newEpoch(self, exclude_tail=0, shuffle_data=True):
        self.numOfSteps = self.numOfFrames - exclude_tail
        pairs = []
        for i in range(len(self.dataSims)):
            pairs += [(i, int(st)) for st in self.dataFrms[i][0:len(self.dataFrms[i]) - exclude_tail]]
        if shuffle_data:
            random.shuffle(pairs)
        self.epoch = [list(pairs[i * self.numOfSteps:(i + 1) * self.numOfSteps]) for i in range(self.batchSize * self.numOfBatchs)]
        self.epochIdx += 1
        self.batchIdx = 0
        self.stepIdx = 0

    def nextBatch(self):
        self.batchIdx += self.batchSize
        self.stepIdx = 0

    def nextStep(self):
        self.stepIdx += 1

    def selection(self):
        """[(sim index, first frame)] of the current batch / step (what getData assembles)."""
        return [self.epoch[self.batchIdx + i][self.stepIdx] for i in range(self.batchSize)]

    def getData(self, consecutive_frames, with_skip=1):
        sel = self.selection()
        frames = lambda k: [np.concatenate([self.dataPreloaded[self.dataSims[s]][f + j * with_skip][k] for (s, f) in sel], axis=0)
                            for j in range(consecutive_frames + 1)]
        ext = [self.extConstChannelPerSim[self.dataSims[s]][0] for (s, _) in sel]
        return [frames(0), frames(1), ext]


class ResidentFrames:
    """The pre-loaded (down-sampled) frames of a PhifDataset resident in device memory (the whole
    6 x 500-frame set of the reference's recipe is ~300 MB), split into the solver's layouts once:
    dens [S,F,Y,X], vy [S,F,Y+1,X], vx [S,F,Y,X+1].  A training batch is then a device-side gather of (sim, frame) windows
    into the trainer's persistent buffers -- no per-step host concatenation, no pageable 20 MB host-to-device copy."""

    def __init__(self, dataset, device):
        import torch
        sims = dataset.dataSims
        pre = dataset.dataPreloaded
        F = len(pre[sims[0]])
        dens = np.stack([np.stack([pre[s][i][0][0, ..., 0] for i in range(F)]) for s in sims])
        velo = [[split_staggered(pre[s][i][1]) for i in range(F)] for s in sims]
        vy = np.stack([np.stack([velo[j][i][0][0] for i in range(F)]) for j in range(len(sims))])
        vx = np.stack([np.stack([velo[j][i][1][0] for i in range(F)]) for j in range(len(sims))])
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        self.dens, self.vy, self.vx = up(dens), up(vy), up(vx)
        self.re = torch.tensor([float(dataset.extConstChannelPerSim[s][0]) for s in sims], dtype=torch.float32, device=device)
        self.device = device
        self.bytes = sum(t.numel() * 4 for t in (self.dens, self.vy, self.vx))

    def gather(self, sel, msteps, d0, vy0, vx0, re, gt_vy, gt_vx, with_skip=1):
        """sel: [(sim index, first frame)] of this rank's simulations (what PhifDataset.getData selects).  Fills the start
        state (frame f), the msteps ground-truth frames (f+1 .. f+msteps) and Re, all on the device."""
        import torch
        s = torch.tensor([a for a, _ in sel], dtype=torch.long, device=self.device)
        f = torch.tensor([b for _, b in sel], dtype=torch.long, device=self.device)
        fr = f[None, :] + (torch.arange(1, msteps + 1, device=self.device) * with_skip)[:, None]      # [msteps, B]
        torch.index_select(self.re, 0, s, out=re)
        d0.copy_(self.dens[s, f])
        vy0.copy_(self.vy[s, f])
        vx0.copy_(self.vx[s, f])
        gt_vy.copy_(self.vy[s[None, :], fr])
        gt_vx.copy_(self.vx[s[None, :], fr])


class BurgersDataset(PhifDataset):
    """burgers/burgers_train.py:189-324: frames hold (velocity, force) staggered tensors [1,Y+1,X+1,2]."""

    def __init__(self, dirpath, num_frames, num_sims=None, batch_size=1, print_fn=print, skip_preprocessing=False, scale=4):
        self.dataSims = sorted(glob.glob(dirpath + "/sim_0*"))[0:num_sims]
        self.pathsVel = [sorted(glob.glob(s + "/velo_0*.npz")) for s in self.dataSims]
        self.pathsFrc = [sorted(glob.glob(s + "/forc_0*.npz")) for s in self.dataSims]
        self.dataFrms = [np.arange(num_frames) for _ in self.dataSims]
        self.batchSize = batch_size
        self.epoch, self.epochIdx, self.batch, self.batchIdx, self.step, self.stepIdx = None, 0, None, 0, None, 0
        self.printFn = print_fn
        self.numOfSims = len(self.dataSims) if num_sims is None else num_sims
        self.numOfBatchs = self.numOfSims // self.batchSize
        self.numOfFrames = num_frames
        self.numOfSteps = num_frames
        if not skip_preprocessing and scale > 1:
            def work():
                for j in range(len(self.dataSims)):
                    for i in range(num_frames):
                        for paths in (self.pathsVel, self.pathsFrc):
                            write_zipped_array(self.filenameToDownscaled(paths[j][i]), downsample_staggered(read_zipped_array(paths[j][i]), scale))
            preprocess_on_rank0(work)
        name = self.filenameToDownscaled if scale > 1 else (lambda p: p)
        self.dataPreloaded = {
            s: [(read_zipped_array(name(self.pathsVel[j][i])).astype(np.float32),
                 read_zipped_array(name(self.pathsFrc[j][i])).astype(np.float32)) for i in range(num_frames)]
            for j, s in enumerate(self.dataSims)}
        self.resolution = [v - 1 for v in self.dataPreloaded[self.dataSims[0]][0][0].shape[1:3]]     # SMAC grid -> cells
        std = lambda k, c: np.std(np.concatenate([np.absolute(self.dataPreloaded[s][i][k][..., c].reshape(-1))
                                                  for s in self.dataSims for i in range(num_frames)]))
        self.dataStats = {"std": ((std(0, 0), std(0, 1)), (std(1, 0), std(1, 1)))}
        self.printFn("Loaded {} samples".format(self.numOfSims * self.numOfFrames))
        self.printFn(self.dataStats)

    def getData(self, consecutive_frames, with_skip=1):
        sel = [self.epoch[self.batchIdx + i][self.stepIdx] for i in range(self.batchSize)]
        frames = lambda k: [np.concatenate([self.dataPreloaded[self.dataSims[s]][f + j * with_skip][k] for (s, f) in sel], axis=0)
                            for j in range(consecutive_frames + 1)]
        return [frames(0), frames(1)]
