#!/usr/bin/env python
"""burgers data generation -- same flags as /root/reference/burgers/burgers.py:35-49, the loop of :160-177 on the fused
HIP Burgers step (state on the GPU, one launch per frame).  Forcing: sol_amd.burgers.SinForces (the reference's 20
SinPotential forces + ForcingPhysics, recalled semantics behind --force-variant), or hi-res force / velocity files
down-sampled by -d (--initvH / --loadfH, the reference's hires -> lores chain, Makefile:35-49).
Grids up to 64 x 64 run the one-workgroup (differentiable) kernel, larger ones -- the reference's 128 x 128 hi-res set,
burgers/Makefile:19-29 -- the forward-only multi-workgroup step."""
import argparse
import glob
import pickle

import numpy as np
import torch

from _common import logger, select_gpu
import sol_amd
from sol_amd import scene
from sol_amd.burgers import SinForces, randfreq


def main(argv=None):
    p = argparse.ArgumentParser(description="Parameter Parser", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--gpu", default="0", help="visible GPUs")
    p.add_argument("--cuda", action="store_true", help="(accepted for compatibility)")
    p.add_argument("-o", "--output", default=None, help="output directory")
    p.add_argument("--thumb", action="store_true", help="(ignored: no PNG thumbnails)")
    p.add_argument("--noforce", action="store_true", help="no randomized external forces")
    p.add_argument("-s", "--skipsteps", default=0, type=int, help="skip first steps")
    p.add_argument("-t", "--simsteps", default=200, type=int, help="simulation steps after skipsteps")
    p.add_argument("-r", "--res", default=32, type=int, help="resolution of the reference axis")
    p.add_argument("-l", "--len", default=32, type=int, help="length of the reference axis")
    p.add_argument("--dt", default=0.1, type=float, help="simulation time step size")
    p.add_argument("--initvH", default=None, help="load hires (will be downsampled) velocity (e.g., velo_0000.npz)")
    p.add_argument("--loadfH", default=None, help="load hires (will be downsampled) force files (will be passed to glob)")
    p.add_argument("-d", "--scale", default=4, type=int, help="down-sampling scale of hires (only valid when initvH given)")
    p.add_argument("--seed", default=0, type=int, help="seed for random number generator")
    p.add_argument("--force-variant", default="sin", choices=["sin", "gradient"], help="recalled SinPotential evaluation (see sol_amd.burgers.SinForces)")
    params = vars(p.parse_args(argv))
    select_gpu(params["gpu"])
    log = logger()
    res = params["res"]
    if res > 1024:
        raise SystemExit("burgers.py: -r %d exceeds the 1024 x 1024 limit of the large-grid Burgers step" % res)
    rng = np.random.default_rng(params["seed"])
    dx = params["len"] / res
    dom = sol_amd.Domain([res, res], box=sol_amd.box([params["len"]] * 2), boundaries=sol_amd.PERIODIC)
    forces = SinForces(rng, 20, params["force_variant"])
    fc_files = sorted(glob.glob(params["loadfH"])) if params["loadfH"] else None
    if params["initvH"]:
        vel = scene.downsample_staggered(scene.read_zipped_array(params["initvH"]), params["scale"])
    else:                                    # st = BurgersVelocitySMAC(dm, velocity=lambda s: math.randfreq(s) * 2)
        vel = np.zeros((1, res + 1, res + 1, 2))
        vel[0, :, :res, 0] = randfreq((res + 1, res), rng) * 2
        vel[0, :res, :, 1] = randfreq((res, res + 1), rng) * 2
    frc = scene.downsample_staggered(scene.read_zipped_array(fc_files[0]), params["scale"]) if fc_files else forces.staggered(res, res, dx)
    f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    st = sol_amd.BurgersVelocitySMAC(dom, velocity=f32(vel), batch_size=1)
    fc = sol_amd.BurgersVelocitySMAC(dom, velocity=f32(frc), batch_size=1)
    path = None
    if params["output"]:
        path = scene.scene_create(params["output"])
        logger(path + "/run.log")
        with open(path + "/params.pickle", "wb") as f:
            pickle.dump(params, f)
    log.info(params)
    sim = sol_amd.BurgersTest()

    def write(i):
        scene.scene_write(path, [st.velocity.staggered_tensor().cpu().numpy(), fc.velocity.staggered_tensor().cpu().numpy()], ["velo", "forc"], i)

    if params["skipsteps"] == 0 and path:
        write(0)
    with torch.no_grad():
        for i in range(1, max(params["simsteps"] + params["skipsteps"], 1)):
            st = sim.step(st, dt=params["dt"]) if params["noforce"] else sim.step_with_f(st, fc, dt=params["dt"])
            if fc_files is None:
                forces.step(params["dt"])
                fc = fc.copied_with(velocity=sol_amd.StaggeredGrid(f32(forces.staggered(res, res, dx)), dom.box))
            else:
                fc = fc.copied_with(velocity=sol_amd.StaggeredGrid(f32(scene.downsample_staggered(scene.read_zipped_array(fc_files[i]), params["scale"])), dom.box))
            if params["skipsteps"] <= i and path:
                write(max(i - params["skipsteps"], 0))
    return path


if __name__ == "__main__":
    main()
