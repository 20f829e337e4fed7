"""Reference-shaped Python surface of the Burgers path (/root/reference/burgers/burgers_train.py):
to_feature / to_feature_noforce (l.75-92), BurgersTest.step / step_with_f (l.178-187), and the TF1
AdamOptimizer used at l.437 as a small optimizer object over the flat parameter buffer."""

import torch

from . import _lib, ops
from ._lib import BurgersCfg, check, ptr, stream
from .fluid import StaggeredGrid


def to_feature(smokestates, forcestates):
    return torch.cat([s.velocity.staggered_tensor()[:, :-1, :-1, 0:2] for s in smokestates] +
                     [f.velocity.staggered_tensor()[:, :-1, :-1, 0:2] for f in forcestates], dim=-1)


def to_feature_noforce(smokestates):
    return torch.cat([s.velocity.staggered_tensor()[:, :-1, :-1, 0:2] for s in smokestates], dim=-1)


class BurgersTest:
    """BurgersTest(Burgers): step(v, dt) / step_with_f(v, f, dt); viscosity = default_viscosity = 0.1."""

    def __init__(self, default_viscosity=0.1, viscosity=None, diffusion_substeps=1):
        if diffusion_substeps != 1:
            raise NotImplementedError("the periodic (spectral) diffusion has no substeps")
        self.viscosity = default_viscosity if viscosity is None else viscosity
        self._circ = {}

    def _run(self, v, f, dt):
        Y, X = v.domain.resolution
        B = v._batch_size
        dx = v.domain.dx[1]
        key = (Y, X, float(dt), str(v.velocity.data[0].data.device))
        if key not in self._circ:
            self._circ[key] = ops.burgers_circ(Y, X, dt * self.viscosity, v.velocity.data[0].data.device)
        cfg = BurgersCfg(B, Y, X, float(dx), float(dt))
        vy = v.velocity.data[0].data.reshape(B, Y + 1, X)
        vx = v.velocity.data[1].data.reshape(B, Y, X + 1)
        fy = fx = None
        if f is not None:
            fy = f.velocity.data[0].data.reshape(B, Y + 1, X)
            fx = f.velocity.data[1].data.reshape(B, Y, X + 1)
        oy, ox = ops.burgers_step(vy, vx, fy, fx, cfg, self._circ[key])
        return v.copied_with(velocity=StaggeredGrid([oy.reshape(B, Y + 1, X, 1), ox.reshape(B, Y, X + 1, 1)], v.velocity.box))

    def step(self, v, dt=1.0, effects=()):
        assert not effects, "effects are unused on the reference path"
        return self._run(v, None, dt)

    def step_with_f(self, v, f, dt=1.0):
        return self._run(v, f, dt)


class TFAdam:
    """tf.compat.v1.train.AdamOptimizer(lr).minimize over a ConvNet's flat parameter buffer."""

    def __init__(self, net, beta1=0.9, beta2=0.999, eps=1e-8):
        self.net, self.t = net, 0
        self.beta1, self.beta2, self.eps = beta1, beta2, eps
        self.m = torch.zeros_like(net.params.detach())
        self.v = torch.zeros_like(net.params.detach())

    def step(self, lr):
        self.t += 1
        g = self.net.params.grad.contiguous()
        check(_lib.load().sol_adam_tf_step(stream(), ptr(self.net.params.detach()), ptr(g), ptr(self.m), ptr(self.v),
                                           self.net.n_params, self.t, float(lr), self.beta1, self.beta2, self.eps, 0.0,
                                           None, 0, None))
