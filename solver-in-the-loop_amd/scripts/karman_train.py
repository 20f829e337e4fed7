#!/usr/bin/env python
"""Solver-in-the-loop training of the karman-2d corrector -- same flags, loop structure, log lines and
output files as /root/reference/karman-2d/karman_train.py (flags :20-47, dataset :344-349, masks
:366-373, train loop :483-517).  The msteps graph + sess.run of :397-457,502 is ONE replayed hipGraph
(SolTrainer.train_step).  Differences, all at the edges: eager PyTorch buffers instead of TF
placeholders; model files are `model_epochNNNN.pt` / `model.pt` (Keras get_weights() order, torch.save)
instead of .h5; TensorBoard summaries are replaced by the log lines; one process per GPU under
torch.distributed.run shards the `-b` simulations of a batch over the ranks (new capability); the down-sampled set stays
resident in device memory and a batch is a device-side gather (--host-feed restores the reference's per-step host
assembly + copy); the loss of step i is read back after step i+1 has been enqueued, so logging never idles the GPU."""
import argparse
import os
import pickle
import random
import time

import numpy as np
import torch

from _common import logger, select_gpu
import sol_amd
from sol_amd import ops, scene


def main(argv=None):
    p = argparse.ArgumentParser(description="Parameter Parser", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--gpu", default="0", help="visible GPUs")
    p.add_argument("--cuda", action="store_true", help="(accepted for compatibility)")
    p.add_argument("--train", default=None, help="training; will load data from this simulation folder (set) and save down-sampled files")
    p.add_argument("--skip-ds", action="store_true", help="skip down-scaling; assume you have already saved")
    p.add_argument("--only-ds", action="store_true", help="exit after down-scaling and saving")
    p.add_argument("--log", default=None, help="path to a log file")
    p.add_argument("-s", "--scale", default=4, type=int, help="simulation scale for high-res")
    p.add_argument("-n", "--nsims", default=1, type=int, help="number of simulations")
    p.add_argument("-b", "--sbatch", default=1, type=int, help="size of a batch")
    p.add_argument("-t", "--simsteps", default=1500, type=int, help="frames per simulation")
    p.add_argument("-m", "--msteps", default=2, type=int, help="multi steps in training loss")
    p.add_argument("-e", "--epochs", default=10, type=int, help="training epochs")
    p.add_argument("--seed", default=None, type=int, help="seed for random number generator")
    p.add_argument("-l", "--len", default=100, type=int, help="length of the reference axis")
    p.add_argument("--model", default="mars_moon", help="(predefined) network model")
    p.add_argument("--reg-loss", action="store_true", help="turn on regularization loss (the models define none)")
    p.add_argument("--lr", default=1e-3, type=float, help="start learning rate")
    p.add_argument("--adplr", action="store_true", help="turn on adaptive learning rate")
    p.add_argument("--clip-grad", action="store_true", help="turn on clip gradients")
    p.add_argument("--resume", default=-1, type=int, help="resume training epochs")
    p.add_argument("--inittf", default=None, help="load initial model weights (warm start)")
    p.add_argument("--pretf", default=None, help="load pre-trained weights (only for testing pre-trained supervised model; do not use for a warm start!)")
    p.add_argument("--tf", default="/tmp/phiflow/tf", help="path to an output dir (model, logs, etc.)")
    p.add_argument("--host-feed", action="store_true", help="assemble every batch on the host and copy it (the reference's feed_dict path) "
                                                            "instead of gathering from the device-resident set")
    params = vars(p.parse_args(argv))
    select_gpu(params["gpu"])
    rank, world, local = sol_amd.dist.init_from_env()
    if params["resume"] > 0 and params["log"]:
        root, ext = os.path.splitext(params["log"])
        params["log"] = root + "_resume{:04d}".format(params["resume"]) + ext
    log = logger(params["log"] if rank == 0 else None)
    if params["nsims"] % params["sbatch"] != 0:
        params["nsims"] = (params["nsims"] // params["sbatch"]) * params["sbatch"]
        log.info("Number of simulations is not divided by the batch size thus adjusted to {}".format(params["nsims"]))
    log.info(params)
    seed = 0 if params["seed"] is None else params["seed"]
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    if params["train"] is None:
        log.info("No pre-loadable training data path is given.")
        return None

    dataset = scene.PhifDataset(params["train"], params["simsteps"], params["nsims"], params["sbatch"], print_fn=log.info,
                                skip_preprocessing=params["skip_ds"], scale=params["scale"])
    if params["only_ds"]:
        return None
    if params["pretf"]:
        # karman_train.py:351-355: the supervised model's own input / output normalisation travels in stats.pickle next to it
        with open(os.path.dirname(params["pretf"]) + "/stats.pickle", "rb") as f:
            ld_stats = pickle.load(f)
        dataset.dataStats["in.std"] = (ld_stats["in.std"][0], (ld_stats["in.std"][1], ld_stats["in.std"][2]))
        dataset.dataStats["out.std"] = ld_stats["out.std"]
        log.info(dataset.dataStats)
    if params["resume"] > 0:
        with open(params["tf"] + "/dataStats.pickle", "rb") as f:
            dataset.dataStats = pickle.load(f)
    Y, X = dataset.resolution
    B, ms = params["sbatch"], params["msteps"]
    lo, hi = sol_amd.dist.shard_range(B, rank, world)            # simulations of a batch owned by this rank
    Bl = hi - lo
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dom = sol_amd.Domain([Y, X], box=sol_amd.box[0:params["len"] * 2, 0:params["len"]])
    simulator_lo = sol_amd.KarmanFlow()
    active, inflow = simulator_lo.scene_arrays(dom)
    velBCy, velBCyMask = sol_amd.velocity_bc_masks(Y, X)
    masks = ops.SceneMasks(active, inflow, velBCy.reshape(Y + 1, X), velBCyMask.reshape(Y + 1, X), dev)
    # eval('model_'+params['model']) (karman_train.py:394): mars_moon runs the C++ schedule (SolTrainer), mercury the
    # autograd composition of the same ops captured into a hipGraph (GraphTrainer)
    assert params["model"] in sol_amd.model.MODELS, "unknown model %r (have: %s)" % (params["model"], ", ".join(sol_amd.model.MODELS))
    model = sol_amd.model.MODELS[params["model"]](3, 2, seed, dev)
    model.summary(print_fn=log.info)
    if params["pretf"]:
        log.info("load a pre-trained model: {}".format(params["pretf"]))
        model.set_weights(sol_amd.ConvNet.load(params["pretf"], device="cpu").get_weights())
    if params["inittf"]:
        log.info("load an initial model (warm start): {}".format(params["inittf"]))
        model.set_weights(sol_amd.ConvNet.load(params["inittf"], device="cpu").get_weights())
    os.makedirs(params["tf"], exist_ok=True)
    if params["resume"] < 1:
        if rank == 0:
            with open(params["tf"] + "/dataStats.pickle", "wb") as f:
                pickle.dump(dataset.dataStats, f)
    else:
        model.set_weights(sol_amd.ConvNet.load(params["tf"] + "/model_epoch{:04d}.pt".format(params["resume"]), device="cpu").get_weights())
    std_v = dataset.dataStats["std"][1]
    # karman_train.py:416-421: 'in.std' / 'out.std' (only present with --pretf) scale the network's input / output, the loss keeps 'std'
    trainer = sol_amd.make_trainer(model, masks, Bl, Y, X, ms, dom.dx[1], std_v, dataset.dataStats["ext.std"][0],
                                 clip_grad=params["clip_grad"],
                                 in_std_v=dataset.dataStats["in.std"][1] if "in.std" in dataset.dataStats else None,
                                 out_std_v=dataset.dataStats["out.std"] if "out.std" in dataset.dataStats else None)
    # persistent device buffers: the captured hipGraph keeps their addresses, new data is copied in
    f32 = lambda shape: torch.empty(shape, dtype=torch.float32, device=dev)
    d0, vy0, vx0, re = f32((Bl, Y, X)), f32((Bl, Y + 1, X)), f32((Bl, Y, X + 1)), f32((Bl,))
    gt_vy, gt_vx = f32((ms, Bl, Y + 1, X)), f32((ms, Bl, Y, X + 1))

    resident = None if params["host_feed"] else scene.ResidentFrames(dataset, dev)
    if resident is not None:
        log.info("training set resident on {}: {:.1f} MB".format(dev, resident.bytes / 2 ** 20))
    # loss read-back: step i's scalar goes to pinned host memory asynchronously and is logged once step i+1 is enqueued
    host_loss = [torch.zeros(1, dtype=torch.float32).pin_memory() for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    pending = None                      # (slot, log prefix)
    stats = {"steps": 0, "t0": None}

    def flush(p):
        copied[p[0]].synchronize()
        log.info(p[1].format(float(host_loss[p[0]][0])))
        return float(host_loss[p[0]][0])

    current_lr = params["lr"]
    loss = None
    nstep = 0
    for j in range(params["epochs"]):
        dataset.newEpoch(exclude_tail=ms)
        if j < params["resume"]:
            log.info("resume: skipping {} epoch".format(j + 1))       # replays the shuffling RNG like the reference
            continue
        current_lr = sol_amd.lr_schedule(j, current_lr) if params["adplr"] else params["lr"]
        for ib in range(dataset.numOfBatchs):
            for i in range(dataset.numOfSteps):
                if resident is not None:
                    resident.gather(dataset.selection()[lo:hi], ms, d0, vy0, vx0, re, gt_vy, gt_vx)
                else:
                    dens, velo, ext = dataset.getData(consecutive_frames=ms, with_skip=1)
                    vy, vx = zip(*[scene.split_staggered(v[lo:hi]) for v in velo])
                    d0.copy_(torch.from_numpy(dens[0][lo:hi, ..., 0]))
                    vy0.copy_(torch.from_numpy(vy[0])); vx0.copy_(torch.from_numpy(vx[0]))
                    gt_vy.copy_(torch.from_numpy(np.stack(vy[1:]))); gt_vx.copy_(torch.from_numpy(np.stack(vx[1:])))
                    re.copy_(torch.as_tensor(ext[lo:hi], dtype=torch.float32))
                loss_t = trainer.train_step(d0, vy0, vx0, re, gt_vy, gt_vx, current_lr)
                slot = nstep & 1
                host_loss[slot].copy_(loss_t.detach().reshape(1), non_blocking=True)
                copied[slot].record()
                if pending is not None:
                    loss = flush(pending)
                pending = (slot, "epoch {:03d}/{:03d}, batch {:03d}/{:03d}, step {:04d}/{:04d}: loss={{}}".format(
                    j + 1, params["epochs"], ib + 1, dataset.numOfBatchs, i + 1, dataset.numOfSteps))
                nstep += 1
                if nstep == 4:                      # steady state: graph captured, allocator warm
                    torch.cuda.synchronize()
                    stats["t0"] = time.perf_counter()
                dataset.nextStep()
            dataset.nextBatch()
        if j % 10 == 9 and rank == 0:
            model.save(params["tf"] + "/model_epoch{:04d}.pt".format(j + 1))
    if pending is not None:
        loss = flush(pending)
    if stats["t0"] is not None and nstep > 4:
        torch.cuda.synchronize()
        main.last_ms_per_step = (time.perf_counter() - stats["t0"]) / (nstep - 4) * 1e3
        log.info("steady state: {:.3f} ms per training step over {} steps".format(main.last_ms_per_step, nstep - 4))
    if rank == 0:
        model.save(params["tf"] + "/model.pt")
    return None if loss is None else float(loss)


if __name__ == "__main__":
    main()
