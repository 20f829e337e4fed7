#!/usr/bin/env python
"""karman-3d (BASELINE.json configs[4]) from the command line: data generation / roll-out of the 3-D wake flow, optionally
with a trained 3-D corrector, and a SOL-n training demo on frames generated on the fly.

The reference has no 3-D script (/root/reference/README.md:37-38); the flags follow karman-2d/karman.py:33-47 and
karman_apply.py:20-31 (-r res of the reference axis -> grid 2r x r x r, --re, -t steps, -s skip, -o output, --model) so that
the 2-D workflow carries over.  Frames are written like the 2-D scenes: dens_%06d.npz [1,Y,X,Z,1], velo_%06d.npz
[1,Y+1,X+1,Z+1,3] (staggered tensor, zero padded at the high ends, components reversed on disk as PhiFlow does)."""
import argparse
import pickle

import numpy as np
import torch

from _common import logger, select_gpu
import sol_amd
from sol_amd import karman3d as k3, scene, synthetic


def staggered3d(vy, vx, vz):
    B, Y1, X, Z = vy.shape
    t = np.zeros((B, Y1, X + 1, Z + 1, 3), dtype=np.float32)
    t[:, :, :X, :Z, 0] = vy
    t[:, :Y1 - 1, :, :Z, 1] = vx
    t[:, :Y1 - 1, :X, :, 2] = vz
    return t


def main(argv=None):
    p = argparse.ArgumentParser(description="Parameter Parser", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--gpu", default="0", help="visible GPUs")
    p.add_argument("-o", "--output", default=None, help="path to an output directory")
    p.add_argument("-t", "--simsteps", default=100, type=int, help="simulation steps")
    p.add_argument("-s", "--skipsteps", default=0, type=int, help="skip first steps when writing")
    p.add_argument("-r", "--res", default=32, type=int, help="resolution of the reference axis (grid 2r x r x r)")
    p.add_argument("--re", default=1e6, type=float, help="Effective Reynolds number")
    p.add_argument("-l", "--len", default=100, type=int, help="length of the reference axis")
    p.add_argument("--model", default=None, help="trained 3-D corrector (.pt written by --train-sol): applied after every solver step")
    p.add_argument("--train-sol", default=0, type=int, help="SOL-n training demo: n unrolled steps per training step on frames generated first")
    p.add_argument("--train-steps", default=4, type=int, help="training steps of the demo")
    p.add_argument("--lr", default=1e-5, type=float)
    p.add_argument("--seed", default=0, type=int)
    params = vars(p.parse_args(argv))
    select_gpu(params["gpu"])
    log = logger()
    res = params["res"]
    Y, X, Z = 2 * res, res, res
    dev = "cuda"
    sc = k3.Scene3D(Y, X, Z, length=float(params["len"]), device=dev)
    sim = k3.Karman3DFlow(sc, 1)
    f = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev).contiguous()
    # karman.py:106-110 in 3-D: uniform flow along y + a sideways poke behind the obstacle
    vy = np.ones((1, Y + 1, X, Z), dtype=np.float32)
    vx = np.zeros((1, Y, X + 1, Z), dtype=np.float32)
    vz = np.zeros((1, Y, X, Z + 1), dtype=np.float32)
    vx[:, Y // 2 + Y // 12:Y // 2 + Y // 6, X // 2 - 2:X // 2 + 2, Z // 2 - 2:Z // 2 + 2] = 1.0
    st = (f(np.zeros((1, Y, X, Z))), f(vy), f(vx), f(vz))
    re = f([params["re"]])
    path = None
    if params["output"]:
        path = scene.scene_create(params["output"])
        logger(path + "/run.log")
        with open(path + "/params.pickle", "wb") as fh:
            pickle.dump(params, fh)
    log.info(params)
    net = ro = None
    if params["model"]:
        blob = torch.load(params["model"], map_location="cpu")
        net = k3.MarsMoon3D(device=dev)
        net.set_weights([w.numpy() for w in blob["weights"]])
        ro = k3.Karman3DRollout(net, sc, 1, blob["std_v"], blob["std_re"])
    frames = []
    with torch.no_grad():
        for i in range(1, params["simsteps"]):
            st = ro.step(*st, re) if ro is not None else sim.step(*st, re)
            if params["train_sol"]:
                frames.append(tuple(t.clone() for t in st))
            if i % 50 == 0:
                log.info("Step {:06d}".format(i))
            if path and i > params["skipsteps"]:
                scene.scene_write(path, [st[0].reshape(1, Y, X, Z, 1).cpu().numpy(),
                                         staggered3d(st[1].cpu().numpy(), st[2].cpu().numpy(), st[3].cpu().numpy())], ["dens", "velo"], i)
    loss = None
    if params["train_sol"]:
        ms = params["train_sol"]
        if len(frames) < ms + 1:
            raise SystemExit("karman3d.py: --train-sol %d needs at least %d simulation steps" % (ms, ms + 2))
        std_v = tuple(float(torch.stack([fr[c] for fr in frames]).abs().std()) + 1e-6 for c in (1, 2, 3))
        net = k3.MarsMoon3D(seed=params["seed"], device=dev)
        w = net.get_weights()
        w[22] = w[22] * 0.01
        net.set_weights(w)
        tr = k3.Karman3DTrainer(net, sc, 1, ms, std_v, max(params["re"], 1.0), use_graph=True)
        rng = np.random.default_rng(params["seed"])
        for it in range(params["train_steps"]):
            k = int(rng.integers(0, len(frames) - ms))
            gts = [frames[k + 1 + j][1:] for j in range(ms)]
            loss = float(tr.train_step(*frames[k], re, gts, lr=params["lr"]))
            log.info("train step {:04d}: loss={}".format(it + 1, loss))
        if path:
            torch.save({"name": net.name, "weights": [torch.as_tensor(a) for a in net.get_weights()], "std_v": std_v,
                        "std_re": max(params["re"], 1.0)}, path + "/model3d.pt")
    return path if path else loss


if __name__ == "__main__":
    main()
