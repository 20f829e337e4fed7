#!/usr/bin/env python
"""Solver-in-the-loop training for forced Burgers -- flags / loop / outputs of
/root/reference/burgers/burgers_train.py (flags :22-44, unroll :379-417, loss :419-437, loop :465-500).
The unrolled graph is composed from the differentiable HIP ops (BurgersTest.step_with_f = fused
periodic advection + spectral diffusion kernel, 5x5 convs on fp32 MFMA) with torch autograd; the
optimizer is the TF1-Adam kernel.  BASELINE configs[0] (32x32, msteps 1) is a plumbing/correctness
config, so this path is not graph-captured."""
import argparse
import os
import pickle
import random

import numpy as np
import torch

from _common import logger, select_gpu
import sol_amd
from sol_amd import ops, scene, _lib
from sol_amd.burgers import BurgersTest, TFAdam, to_feature, to_feature_noforce


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
    simulator_lo = BurgersTest()
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
    opt = TFAdam(model)
    std_v = torch.tensor(dataset.dataStats["std"][0], dtype=torch.float32, device="cuda")
    std_f = torch.tensor(dataset.dataStats["std"][1], dtype=torch.float32, device="cuda")
    std_in = std_v if params["noforce"] else torch.cat([std_v, std_f])
    f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
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
                st = sol_amd.BurgersVelocitySMAC(dom, velocity=f32(velo[0]), batch_size=B)
                losses = []
                for k in range(ms):
                    fr = sol_amd.BurgersVelocitySMAC(dom, velocity=f32(forc[k]), batch_size=B)
                    st = simulator_lo.step(st, dt=dt) if params["noforce"] else simulator_lo.step_with_f(st, fr, dt=dt)
                    feat = to_feature_noforce([st]) if params["noforce"] else to_feature([st], [fr])
                    corr = sol_amd.to_staggered(model(feat / std_in) * std_v, dom.box)
                    st = st.copied_with(velocity=st.velocity + corr)
                    diff = (f32(velo[k + 1]) - st.velocity.staggered_tensor()) / std_v
                    losses.append(0.5 * (diff * diff).sum())
                loss = torch.stack(losses).sum() / ms
                model.params.grad = None
                loss.backward()
                opt.step(current_lr)
                l2 = float(loss)
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
