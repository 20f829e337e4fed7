"""Reference-shaped Python surface of the Burgers path (/root/reference/burgers/burgers_train.py):
to_feature / to_feature_noforce (l.75-92), BurgersTest.step / step_with_f (l.178-187), and the TF1
AdamOptimizer used at l.437 as a small optimizer object over the flat parameter buffer."""

import ctypes as C

import numpy as np
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
        if max(Y, X) > ops.BURGERS_LDS_MAX:
            # beyond the one-workgroup kernels (data generation at the reference's 128 x 128): forward-only multi-workgroup path
            if torch.is_grad_enabled() and (vy.requires_grad or vx.requires_grad):
                raise NotImplementedError("the large-grid Burgers step (%dx%d) is forward only" % (Y, X))
            oy, ox = ops.burgers_step_large(vy, vx, fy, fx, cfg, self._circ[key])
        else:
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


# ---- data generation: forcing model and initial state (host side, numpy) --------------------------------------------
# /root/reference/burgers/burgers.py:89-114 builds 20 travelling sine forces from PhiFlow's SinPotential / FieldEffect
# and advances their phases with ForcingPhysics; :120 draws the initial velocity with math.randfreq.  PhiFlow 1.5.1 is
# not installable here, so the field evaluation is RECALLED [EXT-RECALL]; every recalled choice is a named option.
class SinForces:
    """num_forces plane waves f_i(x) = amplitude_i * sin(k_i . x + phase_i) (per velocity component), k_i = (1 + u) 0.8
    (sin a, cos a) with a = pi u, amplitude = (u - 0.5) 0.3 per component, phase = 2 pi u, omega = 0.8 u - 0.4
    (burgers.py:99-108; u = independent uniform draws in that order).  step(dt): phase += dt * omega (ForcingPhysics.step,
    :89-97).  variant "sin" (default): the field VALUE is data * sin(.) as recalled from SinPotential.sample_at;
    variant "gradient": data * cos(.) * k (a potential's gradient), the alternative reading of the class name."""

    def __init__(self, rng, num_forces=20, variant="sin"):
        if variant not in ("sin", "gradient"):
            raise ValueError("variant must be 'sin' or 'gradient'")
        self.variant = variant
        self.k, self.amp, self.phase, self.omega = [], [], [], []
        for _ in range(num_forces):
            angle = rng.uniform() * np.pi
            direction = np.array([np.sin(angle), np.cos(angle)])
            self.k.append((rng.uniform() + 1) * 0.8 * direction)
            self.amp.append((rng.uniform(size=2) - 0.5) * 0.3)
            self.phase.append(rng.uniform() * 2 * np.pi)
            self.omega.append(rng.uniform() * 0.8 - 0.4)
        self.k, self.amp = np.array(self.k), np.array(self.amp)
        self.phase, self.omega = np.array(self.phase), np.array(self.omega)

    def step(self, dt):
        self.phase = self.phase + dt * self.omega

    def _at(self, py, px, comp):
        ph = self.k[:, 0, None, None] * py[None] + self.k[:, 1, None, None] * px[None] + self.phase[:, None, None]
        if self.variant == "sin":
            return (self.amp[:, comp, None, None] * np.sin(ph)).sum(0)
        return (self.amp[:, comp, None, None] * np.cos(ph) * self.k[:, comp, None, None]).sum(0)

    def staggered(self, Y, X, dx):
        """sum of the force fields sampled on the staggered grid (`.at(dm.staggered_grid(0))`, :122): [1,Y+1,X+1,2], component 0
        (y) at the y-faces (j dx, (i+0.5) dx), component 1 at the x-faces ((j+0.5) dx, i dx), physical coordinates."""
        out = np.zeros((1, Y + 1, X + 1, 2))
        jy, ix = np.meshgrid(np.arange(Y + 1) * dx, (np.arange(X) + 0.5) * dx, indexing="ij")
        out[0, :, :X, 0] = self._at(jy, ix, 0)
        jy, ix = np.meshgrid((np.arange(Y) + 0.5) * dx, np.arange(X + 1) * dx, indexing="ij")
        out[0, :Y, :, 1] = self._at(jy, ix, 1)
        return out


def randfreq(shape, rng, power=8):
    """math.randfreq (burgers.py:120, [EXT-RECALL]): complex white noise in Fourier space damped by (1 + |k|)^-power
    (k = integer wave numbers), real part of the inverse FFT, normalised to unit standard deviation."""
    Y, X = shape
    noise = rng.standard_normal((Y, X)) + 1j * rng.standard_normal((Y, X))
    ky, kx = np.meshgrid(np.fft.fftfreq(Y) * Y, np.fft.fftfreq(X) * X, indexing="ij")
    k = np.sqrt(ky * ky + kx * kx)
    f = np.fft.ifft2(noise * (1.0 / (1.0 + k)) ** power).real
    return f / (f.std() + 1e-30)


# ---- fused trainer: the whole unrolled step (burgers_train.py:379-437) as ONE hipGraph ----------------------------------
class BurgersTrainer:
    """Unrolled solver-in-the-loop training step of the Burgers path (/root/reference/burgers/burgers_train.py:379-437):
    msteps x [BurgersTest.step(_with_f) -> to_feature -> model -> to_staggered correction -> l2 loss], loss = sum / msteps,
    its gradient with respect to the flat parameter buffer, TF1-Adam.  The graph is COMPOSED from the differentiable HIP ops
    of this package by torch autograd (the same composition scripts/burgers_train.py steps eagerly) and captured ONCE into a
    hipGraph over static input buffers; every further step copies the batch into those buffers and replays
    (forward + backward ~ 60 launches per unrolled step: the path is launch bound at 32x32, which is what the capture removes).
    Adam runs outside the graph (its bias correction takes the host-side step counter).

    velo: [msteps+1, B, Y+1, X+1, 2] staggered frames (frame 0 = start state, frames 1.. = targets), forc: [msteps, B, Y+1, X+1, 2]
    (ignored with noforce) -- what BurgersDataset.getData(consecutive_frames=msteps) returns, stacked."""

    def __init__(self, net, domain, batch_size, msteps, dt, std_v, std_f=None, noforce=False, use_graph=True, viscosity=0.1, schedule="manual"):
        """schedule: "manual" (default since round 6) = the unrolled step as a HAND-WRITTEN schedule over the C ABI (_schedule_step: forward
        unroll keeping the step inputs and the network's activations, reverse sweep with the weight gradients accumulated over the steps;
        no autograd graph), "autograd" = the torch-autograd composition of the differentiable HIP ops (rounds 2-5; the cross-check)."""
        from . import fluid
        if schedule not in ("manual", "autograd"):
            raise ValueError("schedule must be 'manual' or 'autograd'")
        self.schedule, self._sched = schedule, None
        self.net, self.dom, self.B, self.ms, self.dt, self.noforce = net, domain, int(batch_size), int(msteps), float(dt), bool(noforce)
        dev = net.params.device
        Y, X = domain.resolution
        self.sim = BurgersTest(default_viscosity=viscosity)
        self.std_v = torch.as_tensor(std_v, dtype=torch.float32, device=dev).reshape(2)
        self._std_v_host = tuple(float(v) for v in np.asarray(std_v, dtype=np.float64).reshape(2))
        if noforce:
            self.std_in = self.std_v
            self._std_in_host = self._std_v_host
        else:
            self.std_in = torch.cat([self.std_v, torch.as_tensor(std_f, dtype=torch.float32, device=dev).reshape(2)])
            self._std_in_host = self._std_v_host + tuple(float(v) for v in np.asarray(std_f, dtype=np.float64).reshape(2))
        self.grads = torch.zeros(net.n_params, dtype=torch.float32, device=dev)        # manual schedule: the flat gradient (also net.params.grad)
        self.velo = torch.zeros(self.ms + 1, self.B, Y + 1, X + 1, 2, dtype=torch.float32, device=dev)
        self.forc = torch.zeros(self.ms, self.B, Y + 1, X + 1, 2, dtype=torch.float32, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.opt = TFAdam(net)
        self.use_graph, self._graph, self._fluid = bool(use_graph), None, fluid

    # the unrolled loss on the static buffers (reference-shaped surface; scripts/burgers_train.py holds the same lines)
    def _unrolled_loss(self):
        from .karman import to_staggered
        F = self._fluid
        st = F.BurgersVelocitySMAC(self.dom, velocity=self.velo[0], batch_size=self.B)
        losses = []
        for k in range(self.ms):
            fr = None if self.noforce else F.BurgersVelocitySMAC(self.dom, velocity=self.forc[k], batch_size=self.B)
            st = self.sim.step(st, dt=self.dt) if self.noforce else self.sim.step_with_f(st, fr, dt=self.dt)
            feat = to_feature_noforce([st]) if self.noforce else to_feature([st], [fr])
            corr = to_staggered(self.net(feat / self.std_in) * self.std_v, self.dom.box)
            st = st.copied_with(velocity=st.velocity + corr)
            # l2_loss((gt.staggered - prd.staggered) / std_v), burgers_train.py:421-428, channel by channel: one kernel per step, no torch
            # reduction (a multi-workgroup torch .sum() puts a memset node into the captured graph: ops.L2LossFn)
            vt, gt_t = st.velocity.staggered_tensor(), self.velo[k + 1]
            losses.append(ops.l2_loss((vt[..., 0].contiguous(), vt[..., 1].contiguous()), (gt_t[..., 0].contiguous(), gt_t[..., 1].contiguous()), self._std_v_host))
        return _lib.stack0(losses).sum() / self.ms

    def _schedule_step(self):
        """burgers_train.py:379-437 differentiated by hand.  Forward, per unrolled step k: sol_burgers_step_fwd on (v, f_k) (the step's INPUT
        velocity is what its adjoint needs) -> features = (v, f_k at the low faces) / std_in -> the network's forward launches
        (schedule2d.NetSchedule2D) -> v += std_v * to_staggered(out) -> loss_k and d loss_k / d v_k in one pass over the padded staggered
        tensors (sol_l2_loss_fwd_bwd; the padding only adds the constant the reference's l2_loss sees there).  Reverse, k = n-1 .. 0:
        G = d loss_k / d v_k + (adjoint of step k+1) -> d out = std_v * G at the corrected faces -> network reverse sweep (weight gradients
        accumulated over the steps) -> G += d features / std_in at the low faces -> sol_burgers_step_bwd.  One reduce per layer at the end."""
        from .schedule2d import NetSchedule2D
        lib = _lib.load()
        Y, X = self.dom.resolution
        B, ms, dev = self.B, self.ms, self.velo.device
        if self._sched is None:
            self._sched = NetSchedule2D(self.net, B, Y, X)
            self._circ = ops.burgers_circ(Y, X, self.dt * self.sim.viscosity, dev)
            self._bcfg = BurgersCfg(B, Y, X, float(self.dom.dx[1]), float(self.dt))
        sch, circ, cfg = self._sched, self._circ, self._bcfg
        sv, sin = self._std_v_host, self._std_in_host
        sch.begin_step()
        vy, vx = self.velo[0][:, :, :X, 0].contiguous(), self.velo[0][:, :Y, :, 1].contiguous()
        keep, losses = [], []
        for k in range(ms):
            fy = fx = None
            if not self.noforce:
                fy, fx = self.forc[k][:, :, :X, 0].contiguous(), self.forc[k][:, :Y, :, 1].contiguous()
            oy, ox = torch.empty_like(vy), torch.empty_like(vx)
            check(lib.sol_burgers_step_fwd(C.byref(cfg), stream(), ptr(vy), ptr(vx), ptr(fy), ptr(fx), ptr(circ[0]), ptr(circ[1]), ptr(circ[2]), ptr(circ[3]),
                                           ptr(oy), ptr(ox)))
            chans = [oy[:, :Y], ox[:, :, :X]]
            if not self.noforce:
                chans += [fy[:, :Y], fx[:, :, :X]]
            # (the division by the std TENSOR, as to_feature(...) / std_in of the autograd composition does it: a scalar division is a
            #  multiplication by the reciprocal in torch -- one ulp apart, enough to flip the sign of an activation that sits at a
            #  LeakyReLU kink and with it a gradient element: tools/dbg/burgers_sched_dbg.py)
            out, state = sch.forward(torch.stack(chans, dim=-1) / self.std_in)
            oy[:, :Y].add_(out[..., 0], alpha=sv[0])                 # to_staggered + add: the last row / column receives no correction
            ox[:, :, :X].add_(out[..., 1], alpha=sv[1])
            gt = self.velo[k + 1]
            li, gi = ops.l2_loss_fwd_bwd((_lib.pad_high(oy, 2), _lib.pad_high(ox, 1)), (gt[..., 0].contiguous(), gt[..., 1].contiguous()), sv, gscale=1.0 / ms)
            losses.append(li.reshape(()))
            keep.append((vy, vx, state, (gi[0][:, :, :X].contiguous(), gi[1][:, :Y, :].contiguous())))
            vy, vx = oy, ox
        gin = None
        for k in range(ms - 1, -1, -1):
            iy, ix, state, G = keep[k]
            if gin is not None:
                G[0].add_(gin[0])
                G[1].add_(gin[1])
            dO = torch.stack([G[0][:, :Y] * sv[0], G[1][:, :, :X] * sv[1]], dim=-1)
            dx = sch.backward(state, dO)
            G[0][:, :Y].add_(dx[..., 0], alpha=1.0 / sin[0])
            G[1][:, :, :X].add_(dx[..., 1], alpha=1.0 / sin[1])
            oy, ox = torch.empty_like(iy), torch.empty_like(ix)
            check(lib.sol_burgers_step_bwd(C.byref(cfg), stream(), ptr(iy), ptr(ix), ptr(circ[0]), ptr(circ[1]), ptr(circ[2]), ptr(circ[3]),
                                           ptr(G[0]), ptr(G[1]), ptr(oy), ptr(ox)))
            gin = (oy, ox)
            keep[k] = None
        _lib.dcopy_(self.loss, _lib.stack0(losses).sum() / ms)
        _lib.dcopy_(self.grads, sch.end_step())

    def _eager(self):
        if self.schedule == "manual":
            with torch.no_grad():
                self._schedule_step()
            self.net.params.grad = self.grads
            return
        self.net.params.grad = None
        loss = self._unrolled_loss()
        loss.backward()
        _lib.dcopy_(self.loss, loss)

    def _capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # warm-up off the capture: library initialisation, allocator pools
            for _ in range(2):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.schedule == "manual":
            def body():
                with torch.no_grad():
                    self._schedule_step()
            self._graph = _lib.capture_graph(body, "BurgersTrainer")
            self.net.params.grad = self.grads    # static buffer, rewritten by every replay
            return
        self.net.params.grad = None             # the captured backward allocates .grad from the graph's pool and rewrites it per replay
        def body():
            loss = self._unrolled_loss()
            loss.backward()
            _lib.dcopy_(self.loss, loss)          # (a kernel: a contiguous copy_ would be a memcpy node, refused by the capture guard)
        self._graph = _lib.capture_graph(body, "BurgersTrainer")      # kernel nodes only (sol_graph_check), then instantiated

    def fwd_bwd(self, velo, forc=None, eager=False):
        """Copies the batch into the static buffers, runs forward + backward; returns the loss tensor (device scalar); the
        gradient is net.params.grad."""
        self.velo.copy_(torch.as_tensor(velo, dtype=torch.float32).reshape(self.velo.shape), non_blocking=True)
        if not self.noforce:
            self.forc.copy_(torch.as_tensor(forc, dtype=torch.float32).reshape(self.forc.shape), non_blocking=True)
        if eager or not self.use_graph:
            self._eager()
        else:
            if self._graph is None:
                self._capture()
            self._graph.replay()
        return self.loss

    def train_step(self, velo, forc=None, lr=1e-3, eager=False):
        loss = self.fwd_bwd(velo, forc, eager)
        self.opt.step(lr)
        return loss


# ---- no-grad roll-out (burgers_apply.py:129-151) ------------------------------------------------------------------------
class BurgersRollout:
    """Inference loop of /root/reference/burgers/burgers_apply.py:129-151 without autograd state:
        v = step_with_f(v, f_prev, dt)  (or step(v, dt))  ->  features = to_feature([v], [f_next]) / std  ->
        correction = to_staggered(model(features) * std_v)  ->  v += correction
    (the reference advances with the force frame it loaded on the PREVIOUS iteration and feeds the network the frame it
    loads on THIS iteration, :131-134).  One step = the HIP solver step + twelve conv launches + the pad/add, captured ONCE
    into a hipGraph over static buffers (the path is launch bound at 32 x 32); step() copies the two force frames in and
    replays.  `vel` / `corr` are the static staggered tensors [B,Y+1,X+1,2] holding the state after the step and its
    correction (what the script writes as velTf / corTf)."""

    def __init__(self, net, domain, batch_size, dt, std_v, std_f=None, noforce=False, use_graph=True, viscosity=0.1):
        from . import fluid
        _lib.require_gpu()
        self.net, self.dom, self.B, self.dt, self.noforce = net, domain, int(batch_size), float(dt), bool(noforce)
        dev = net.params.device
        Y, X = domain.resolution
        self.sim = BurgersTest(default_viscosity=viscosity)
        self.std_v = torch.as_tensor(std_v, dtype=torch.float32, device=dev).reshape(2)
        self.std_in = self.std_v if noforce else torch.cat([self.std_v, torch.as_tensor(std_f, dtype=torch.float32, device=dev).reshape(2)])
        z = lambda: torch.zeros(self.B, Y + 1, X + 1, 2, dtype=torch.float32, device=dev)
        self.vel, self.corr, self.f_step, self.f_feat = z(), z(), z(), z()
        self.use_graph, self._graph, self._fluid = bool(use_graph), None, fluid

    def _one(self):
        from .karman import to_staggered
        F = self._fluid
        with torch.no_grad():
            st = F.BurgersVelocitySMAC(self.dom, velocity=self.vel, batch_size=self.B)
            if self.noforce:
                st = self.sim.step(st, dt=self.dt)
                feat = to_feature_noforce([st])
            else:
                st = self.sim.step_with_f(st, F.BurgersVelocitySMAC(self.dom, velocity=self.f_step, batch_size=self.B), dt=self.dt)
                feat = to_feature([st], [F.BurgersVelocitySMAC(self.dom, velocity=self.f_feat, batch_size=self.B)])
            cv = to_staggered(self.net.predict(feat / self.std_in) * self.std_v, self.dom.box)
            _lib.dcopy_(self.corr, cv.staggered_tensor())      # (kernel copies: no memcpy nodes in the captured graph)
            _lib.dcopy_(self.vel, (st.velocity + cv).staggered_tensor())

    def reset(self, velocity):
        self.vel.copy_(torch.as_tensor(velocity, dtype=torch.float32).reshape(self.vel.shape))
        self.corr.zero_()

    def step(self, f_step=None, f_feat=None):
        """Advances self.vel by one corrected step.  f_step: force frame of the solver step, f_feat: force frame of the
        network's input ([B,Y+1,X+1,2] staggered; ignored with noforce)."""
        if not self.noforce:
            self.f_step.copy_(torch.as_tensor(f_step, dtype=torch.float32).reshape(self.f_step.shape), non_blocking=True)
            self.f_feat.copy_(torch.as_tensor(f_feat, dtype=torch.float32).reshape(self.f_feat.shape), non_blocking=True)
        if not self.use_graph:
            self._one()
            return self.vel
        if self._graph is None:
            keep = self.vel.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):           # warm-up off the capture (library initialisation, allocator pools)
                self._one()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.vel.copy_(keep)
            self._graph = _lib.capture_graph(self._one, "BurgersRollout")
            self.vel.copy_(keep)
        self._graph.replay()
        return self.vel
