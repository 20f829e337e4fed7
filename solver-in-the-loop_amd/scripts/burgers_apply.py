#!/usr/bin/env python
"""Inference roll-out of the Burgers corrector -- flags of /root/reference/burgers/burgers_apply.py:22-33, loop :129-151
(simulator.step_with_f -> to_feature -> model.predict correction -> to_staggered, add -> write velTf / corTf frames).
The step (fused periodic advection + spectral diffusion kernel, twelve 5x5 convolutions, pad + add) is one replayed
hipGraph (sol_amd.BurgersRollout); --no-graph steps the same composition eagerly."""
import argparse
import glob
import pickle

import numpy as np
import torch

from _common import logger, select_gpu
import sol_amd
from sol_amd import burgers, scene


def main(argv=None):
    p = argparse.ArgumentParser(description="Parameter Parser", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--gpu", default="0", help="visible GPUs")
    p.add_argument("-t", "--simsteps", default=200, type=int, help="simulation steps")
    p.add_argument("-r", "--res", default=32, type=int, help="resolution of the reference axis")
    p.add_argument("-l", "--len", default=96, type=int, help="length of the reference axis")
    p.add_argument("--dt", default=1.0, type=float, help="simulation time step size")
    p.add_argument("--noforce", action="store_true", help="no randomized external forces")
    p.add_argument("--initvH", default=None, help="load hires (will be downsampled) velocity (e.g., velo_0000.npz)")
    p.add_argument("--loadfH", default=None, help='load hires (will be downsampled) force files (will be passed to glob) (e.g., "sim_000000/forc_0*.npz")')
    p.add_argument("-s", "--scale", default=4, type=int, help="simulation scale for high-res")
    p.add_argument("-o", "--output", default="/tmp/phiflow/run", help="path to an output directory")
    p.add_argument("--stats", default="/tmp/phiflow/data/dataStats.pickle", help="path to datastats")
    p.add_argument("--model", default="/tmp/phiflow/tf/model.pt", help="path to a trained model")
    p.add_argument("--seed", default=0, type=int, help="seed of the random initial velocity (without --initvH)")
    p.add_argument("--no-graph", action="store_true", help="step eagerly instead of replaying a captured hipGraph")
    params = vars(p.parse_args(argv))
    select_gpu(params["gpu"])
    log = logger()
    res = params["res"]
    dom = sol_amd.Domain([res, res], box=sol_amd.box([params["len"]] * 2), boundaries=sol_amd.PERIODIC)
    down = lambda a: scene.downsample_staggered(a, params["scale"])
    rng = np.random.default_rng(params["seed"])
    if params["initvH"]:
        v0 = np.asarray(down(scene.read_zipped_array(params["initvH"])), dtype=np.float32)
    else:   # velocity=lambda s: math.randfreq(s) * 2  (burgers_apply.py:87)
        v0 = np.zeros((1, res + 1, res + 1, 2), dtype=np.float32)
        v0[0, :, :res, 0] = burgers.randfreq((res + 1, res), rng) * 2
        v0[0, :res, :, 1] = burgers.randfreq((res, res + 1), rng) * 2
    fc_files = None
    if not params["noforce"]:
        if not params["loadfH"]:
            raise SystemExit("burgers_apply.py: --loadfH is required unless --noforce is given (burgers_apply.py:90-92)")
        fc_files = sorted(glob.glob(params["loadfH"]))
        if len(fc_files) < params["simsteps"]:
            raise SystemExit("burgers_apply.py: %d force files for %d steps" % (len(fc_files), params["simsteps"]))
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
    std_f = None if params["noforce"] else data_stats["std"][1]
    ro = sol_amd.BurgersRollout(model, dom, 1, params["dt"], data_stats["std"][0], std_f, noforce=params["noforce"],
                                use_graph=not params["no_graph"])
    ro.reset(v0)
    fc = None if params["noforce"] else np.asarray(down(scene.read_zipped_array(fc_files[0])), dtype=np.float32)
    scene.scene_write(path, [ro.vel.cpu().numpy(), ro.corr.cpu().numpy()], ["velTf", "corTf"], 0)
    for i in range(1, params["simsteps"]):
        if params["noforce"]:
            ro.step()
        else:
            fc_next = np.asarray(down(scene.read_zipped_array(fc_files[i])), dtype=np.float32)
            ro.step(fc, fc_next)          # solver step with the previous frame's force, network input with this frame's (:131-134)
            fc = fc_next
        scene.scene_write(path, [ro.vel.cpu().numpy(), ro.corr.cpu().numpy()], ["velTf", "corTf"], i)
        log.info("step {:06d}".format(i))
    return path


if __name__ == "__main__":
    main()
