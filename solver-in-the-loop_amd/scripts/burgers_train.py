#!/usr/bin/env python
"""Solver-in-the-loop training for forced Burgers -- flags / loop / outputs of
/root/reference/burgers/burgers_train.py (flags :22-44, unroll :379-417, loss :419-437, loop :465-500).
The unrolled graph is composed from the differentiable HIP ops (BurgersTest.step_with_f = fused periodic advection + spectral
diffusion kernel, 5x5 convs on the matrix cores) by torch autograd and captured ONCE into a hipGraph over static buffers
(sol_amd.BurgersTrainer); --no-graph steps the same composition eagerly.  The optimizer is the TF1-Adam kernel."""
import argparse
import os
import pickle
import random

import numpy as np
import torch

from _common import logger, select_gpu
import sol_amd
from sol_amd import ops, scene, _lib


def main(argv=None):
    p = argparse.ArgumentParser(description="Parameter Parser", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--gpu", default="0")
    p.add_argument("--cuda", action="store_true")
    p.add_argument("--train", default=None, help="training; will load data from this folder (set)")
    p.add_argument("--skip-ds", action="store_true")
    p.add_argument("--only-ds", action="store_true")
    p.add_argument("--log", default=None)
    p.add_argument("-s", "--scale", default=4, type=int)
    p.add_argument("-n", "--nsims", default=10, type=int)
    p.add_argument("-b", "--sbatch", default=2, type=int)
    p.add_argument("-t", "--simsteps", default=200, type=int)
    p.add_argument("-m", "--msteps", default=2, type=int)
    p.add_argument("-e", "--epochs", default=10, type=int)
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--noforce", action="store_true")
    p.add_argument("-l", "--len", default=32, type=int)
    p.add_argument("--dt", default=1.0, type=float)
    p.add_argument("--model", default="mars_moon")
    p.add_argument("--lr", default=1e-3, type=float)
    p.add_argument("--adplr", action="store_true")
    p.add_argument("--resume", default=-1, type=int)
    p.add_argument("--inittf", default=None)
    p.add_argument("--tf", default="/tmp/phiflow/tf")
    p.add_argument("--no-graph", action="store_true", help="step the unrolled graph eagerly instead of replaying a captured hipGraph")
    params = vars(p.parse_args(argv))
    select_gpu(params["gpu"])
    log = logger(params["log"])
    log.info(params)
    random.seed(params["seed"]); np.random.seed(params["seed"]); torch.manual_seed(params["seed"])
    if params["train"] is None:
        log.info("No pre-loadable training data path is given.")
        return None
    dataset = scene.BurgersDataset(params["train"], params["simsteps"], params["nsims"], params["sbatch"], print_fn=log.info,
                                   skip_preprocessing=params["skip_ds"], scale=params["scale"])
    if params["only_ds"]:
        return None
    if params["resume"] > 0:
        with open(params["tf"] + "/dataStats.pickle", "rb") as f:
            dataset.dataStats = pickle.load(f)
    Y, X = dataset.resolution
    B, ms, dt = params["sbatch"], params["msteps"], params["dt"]
    dom = sol_amd.Domain([Y, X], box=sol_amd.box([params["len"]] * 2), boundaries=sol_amd.PERIODIC)
    cin = 2 if params["noforce"] else 4
    model = sol_amd.model_mars_moon(cin=cin, cout=2, seed=params["seed"])
    model.summary(print_fn=log.info)
    if params["inittf"]:
        model.set_weights(sol_amd.ConvNet.load(params["inittf"], device="cpu").get_weights())
    os.makedirs(params["tf"], exist_ok=True)
    if params["resume"] < 1:
        with open(params["tf"] + "/dataStats.pickle", "wb") as f:
            pickle.dump(dataset.dataStats, f)
    else:
        model.set_weights(sol_amd.ConvNet.load(params["tf"] + "/model_epoch{:04d}.pt".format(params["resume"]), device="cpu").get_weights())
    trainer = sol_amd.BurgersTrainer(model, dom, B, ms, dt, dataset.dataStats["std"][0], dataset.dataStats["std"][1],
                                     noforce=params["noforce"], use_graph=not params["no_graph"])
    current_lr = params["lr"]
    l2 = None
    for j in range(params["epochs"]):
        dataset.newEpoch(exclude_tail=ms)
        if j < params["resume"]:
            log.info("resume: skipping {} epoch".format(j + 1))
            continue
        current_lr = sol_amd.lr_schedule(j, current_lr) if params["adplr"] else params["lr"]
        for ib in range(dataset.numOfBatchs):
            for i in range(dataset.numOfSteps):
                velo, forc = dataset.getData(consecutive_frames=ms, with_skip=1)
                l2 = float(trainer.train_step(np.stack(velo[:ms + 1]), np.stack(forc[:ms]), current_lr))
                log.info("epoch {:03d}/{:03d}, batch {:03d}/{:03d}, step {:04d}/{:04d}: loss={}".format(
                    j + 1, params["epochs"], ib + 1, dataset.numOfBatchs, i + 1, dataset.numOfSteps, l2))
                dataset.nextStep()
            dataset.nextBatch()
        if j % 10 == 9 or j == 0:
            model.save(params["tf"] + "/model_epoch{:04d}.pt".format(j + 1))
    model.save(params["tf"] + "/model.pt")
    return l2


if __name__ == "__main__":
    main()
