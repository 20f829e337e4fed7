#!/usr/bin/env python
"""karman-2d data generation -- same flags as /root/reference/karman-2d/karman.py:33-47, the loop of
:138-159 on the fused HIP solver step (one kernel launch per frame, state stays on the GPU).
-r <= 64 runs the fused one-workgroup-per-simulation kernel; larger grids (the reference's 256x128 `-r 128`
reference solutions, Makefile:19-28) run the forward-only multi-launch path with the direct pressure solver."""
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
    p.add_argument("--cuda", action="store_true", help="(accepted for compatibility; the solver is always the HIP kernel)")
    p.add_argument("-o", "--output", default=None, help="path to an output directory")
    p.add_argument("--thumb", action="store_true", help="(ignored: no PNG thumbnails)")
    p.add_argument("-t", "--simsteps", default=1500, type=int, help="simulation steps: an epoch")
    p.add_argument("-s", "--skipsteps", default=999, type=int, help="skip first steps; (vortices may not form)")
    p.add_argument("-r", "--res", default=32, type=int, help="resolution of the reference axis")
    p.add_argument("--re", default=1e6, type=float, help="Effective Reynolds number")
    p.add_argument("--initdH", default=None, help="load hires (will be downsampled) density")
    p.add_argument("--initvH", default=None, help="load hires (will be downsampled) velocity")
    p.add_argument("-d", "--scale", default=4, type=int, help="down-sampling scale of hires")
    p.add_argument("-l", "--len", default=100, type=int, help="length of the reference axis")
    p.add_argument("--seed", default=0, type=int, help="seed for random number generator")
    params = vars(p.parse_args(argv))
    select_gpu(params["gpu"])
    log = logger()
    res = params["res"]
    Y, X = 2 * res, res
    dom = sol_amd.Domain([Y, X], box=sol_amd.box[0:params["len"] * 2, 0:params["len"]])
    sim = sol_amd.KarmanFlow()
    d0 = scene.downsample(scene.read_zipped_array(params["initdH"]), params["scale"]) if params["initdH"] else np.zeros((1, Y, X, 1))
    if params["initvH"]:
        vn = scene.downsample_staggered(scene.read_zipped_array(params["initvH"]), params["scale"])
    else:                                   # karman.py:106-110: warm start + sideways poke
        vn = np.zeros((1, Y + 1, X + 1, 2))
        vn[..., 0] = 1.0
        vn[..., vn.shape[1] // 2 + 10:vn.shape[1] // 2 + 20, vn.shape[2] // 2 - 2:vn.shape[2] // 2 + 2, 1] = 1.0
    st = sol_amd.Fluid(dom, density=d0, velocity=vn, batch_size=1)
    velBCy, velBCyMask = sol_amd.velocity_bc_masks(Y, X, batch_size=1)
    path = None
    if params["output"]:
        path = scene.scene_create(params["output"])
        logger(path + "/run.log")
        with open(path + "/params.pickle", "wb") as f:
            pickle.dump(params, f)
    log.info(params)

    def write(state, i):
        scene.scene_write(path, [state.density.data.cpu().numpy(), state.velocity.staggered_tensor().cpu().numpy()], ["dens", "velo"], i)

    if params["skipsteps"] == 0 and path:
        write(st, 0)
    with torch.no_grad():
        for i in range(1, params["simsteps"]):
            st = sim.step(st, re=[params["re"]], res=res, velBCy=velBCy, velBCyMask=velBCyMask)
            if i % 100 == 0:
                log.info("Step {:06d}".format(i))
            if params["skipsteps"] < i and path:
                write(st, i)
    return path


if __name__ == "__main__":
    main()
