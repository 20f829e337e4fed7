#!/usr/bin/env python
"""Inference roll-out with the trained corrector -- flags of /root/reference/karman-2d/karman_apply.py:
20-31, loop :138-158 (simulator.step -> model.predict correction -> write denTf/velTf/corTf frames).
Solver step + CNN run on the GPU (SolRollout, one frame per call so that every frame can be written)."""
import argparse
import pickle

import numpy as np
import torch

from _common import logger, select_gpu
import sol_amd
from sol_amd import ops, scene


def main(argv=None):
    p = argparse.ArgumentParser(description="Parameter Parser", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--gpu", default="0", help="visible GPUs")
    p.add_argument("-s", "--scale", default=4, type=int, help="simulation scale for high-res")
    p.add_argument("-r", "--res", default=32, type=int, help="resolution of the reference axis")
    p.add_argument("-l", "--len", default=100, type=int, help="length of the reference axis")
    p.add_argument("--re", default=1e6, type=float, help="Reynolds number")
    p.add_argument("--initdH", default=None, help="load hires (will be downsampled) density")
    p.add_argument("--initvH", default=None, help="load hires (will be downsampled) velocity")
    p.add_argument("-t", "--simsteps", default=500, type=int, help="simulation steps")
    p.add_argument("-o", "--output", default="/tmp/phiflow/run", help="path to an output directory")
    p.add_argument("--stats", default="/tmp/phiflow/data/dataStats.pickle", help="path to datastats")
    p.add_argument("--model", default="/tmp/phiflow/tf/model.pt", help="path to a trained model")
    params = vars(p.parse_args(argv))
    select_gpu(params["gpu"])
    log = logger()
    res = params["res"]
    Y, X = 2 * res, res
    dom = sol_amd.Domain([Y, X], box=sol_amd.box[0:params["len"] * 2, 0:params["len"]])
    d0 = scene.downsample(scene.read_zipped_array(params["initdH"]), params["scale"]) if params["initdH"] else np.zeros((1, Y, X, 1))
    if params["initvH"]:
        vn = scene.downsample_staggered(scene.read_zipped_array(params["initvH"]), params["scale"])
    else:
        vn = np.zeros((1, Y + 1, X + 1, 2))
        vn[..., 0] = 1.0
        vn[..., vn.shape[1] // 2 + 10:vn.shape[1] // 2 + 20, vn.shape[2] // 2 - 2:vn.shape[2] // 2 + 2, 1] = 1.0
    path = scene.scene_create(params["output"])
    logger(path + "/run.log")
    log.info(params)
    with open(path + "/params.pickle", "wb") as f:
        pickle.dump(params, f)
    with open(params["stats"], "rb") as f:
        data_stats = pickle.load(f)
    log.info(data_stats)
    model = sol_amd.ConvNet.load(params["model"])
    model.summary(print_fn=log.info)
    sim = sol_amd.KarmanFlow()
    active, inflow = sim.scene_arrays(dom)
    velBCy, velBCyMask = sol_amd.velocity_bc_masks(Y, X)
    masks = ops.SceneMasks(active, inflow, velBCy.reshape(Y + 1, X), velBCyMask.reshape(Y + 1, X))
    ro = sol_amd.SolRollout(model, masks, 1, Y, X, dom.dx[1], data_stats["std"][1], data_stats["ext.std"][0])
    f = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda").contiguous()
    vy0, vx0 = scene.split_staggered(np.asarray(vn, dtype=np.float32))
    d, vy, vx = f(np.asarray(d0)[..., 0]), f(vy0), f(vx0)
    re = f([params["re"]])

    def stag(a, b):
        return sol_amd.StaggeredGrid([a.reshape(1, Y + 1, X, 1), b.reshape(1, Y, X + 1, 1)]).staggered_tensor().cpu().numpy()

    zero = np.zeros((1, Y + 1, X + 1, 2), dtype=np.float32)
    scene.scene_write(path, [d.reshape(1, Y, X, 1).cpu().numpy(), stag(vy, vx), zero], ["denTf", "velTf", "corTf"], 0)
    for i in range(1, params["simsteps"]):
        py, px = vy.clone(), vx.clone()
        # uncorrected step (for the written correction field) and corrected step
        cfg = ops.karman_cfg(1, Y, X, dom.dx[1], res=res, masks=masks)
        with torch.no_grad():
            _, sy, sx = ops.karman_step(d, py, px, re, cfg, masks)
        ro.run(d, vy, vx, re, 1)
        scene.scene_write(path, [d.reshape(1, Y, X, 1).cpu().numpy(), stag(vy, vx), stag(vy - sy, vx - sx)], ["denTf", "velTf", "corTf"], i)
        if i % 100 == 0:
            log.info("step {:06d}".format(i))
    return path


if __name__ == "__main__":
    main()
