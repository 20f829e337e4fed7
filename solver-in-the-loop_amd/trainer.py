"""Fused solver-in-the-loop training step (one C call = the reference's one sess.run).

Replaces the graph built at /root/reference/karman-2d/karman_train.py:397-457 and executed
at :502: msteps x [simulator_lo.step -> CNN correction -> add], the l2 loss against the
ground-truth frames, the reverse sweep and TF-Adam.  Python only owns the device buffers;
the whole unroll is driven from C++ (sol_train_fwd_bwd) so that no per-op Python overhead
sits between the ~1000 kernel launches of a SOL-32 step.

Data parallel (new capability): one process per GPU, each rank runs the
full unroll on its shard of simulations, then ONE all-reduce(SUM) of the flat 260,354-float
gradient over RCCL (the loss is a batch SUM, karman_train.py:430, so summed shard gradients
equal the large-batch gradient) and an identical Adam update on every rank.
"""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import TrainCfg, check, ptr, stream
from .dist import DPStep


_CONV_PRECISION = {"split": 0, "bf16x6": 1, "fp32": 2}


def _conv_precision_code(name):
    if name is None:
        return None
    if name not in _CONV_PRECISION:
        raise ValueError("conv_precision must be one of %s or None (got %r)" % (sorted(_CONV_PRECISION), name))
    return _CONV_PRECISION[name]


def _apply_conv_precision(code):
    if code is not None:
        _lib.set_option("conv_precision", code)


class _conv_precision_scope:
    """The library's `conv_precision` option is process wide; a trainer / roll-out sets ITS value for the duration of a call
    and puts the previous one back, so objects of different precision can alternate and nothing leaks into later calls of
    the per-op API (None: leave the option alone)."""

    def __init__(self, code):
        self.code = code

    def __enter__(self):
        if self.code is not None:
            self.prev = _lib.get_option("conv_precision")
            _lib.set_option("conv_precision", self.code)

    def __exit__(self, *exc):
        if self.code is not None:
            _lib.set_option("conv_precision", self.prev)
        return False


class SolTrainer:
    def __init__(self, net, masks, B, Y, X, msteps, dx, std_v, std_re, dt=1.0, res=None,
                 clip_grad=False, beta1=0.9, beta2=0.999, eps=1e-8, group=None, use_graph=True,
                 cg_rtol=1e-6, cg_atol=1e-9, cg_max_iter=2000, grad_pad="replicate", inflow_order="after",
                 conv_precision="split", comm=None, in_std_v=None, out_std_v=None):
        """conv_precision: arithmetic of the 32-channel convolutions (library option `conv_precision`):
        "split" (default) fp32-equivalent fp16x3 / bf16x6 operand splits on the 16-bit matrix pipe, "bf16x6",
        or "fp32" = strict fp32 MFMA.  The option is process wide in the library; every call of this trainer sets
        it for the duration of the call (and restores the previous value), so trainers of different precision can alternate in
        one process.  None keeps whatever the option table holds
        (e.g. a SOL_CONV_NO_SB / SOL_CONV_NO_FP16 debugging override applied when the library was loaded)."""
        _lib.require_gpu()
        self.lib = _lib.load()
        self.conv_precision = _conv_precision_code(conv_precision)
        assert net.name == "mars_moon", "the fused trainer implements model_mars_moon (the reference default)"
        self.net, self.masks = net, masks
        self.B, self.Y, self.X, self.msteps = B, Y, X, msteps
        kc = ops.karman_cfg(B, Y, X, dx, dt=dt, res=res, cg_rtol=cg_rtol, cg_atol=cg_atol,
                            cg_max_iter=cg_max_iter, grad_pad=grad_pad, inflow_order=inflow_order, masks=masks)
        # --pretf (karman_train.py:351-355,416-421): separate input / output normalisation of a pre-trained supervised model
        i0, i1 = (float(in_std_v[0]), float(in_std_v[1])) if in_std_v is not None else (0.0, 0.0)
        o0, o1 = (float(out_std_v[0]), float(out_std_v[1])) if out_std_v is not None else (0.0, 0.0)
        self.cfg = TrainCfg(kc, msteps, float(std_v[0]), float(std_v[1]), float(std_re), float(net.slope), i0, i1, o0, o1)
        dev = net.params.device
        self.device = dev
        nbytes = self.lib.sol_train_workspace_bytes(C.byref(self.cfg))
        self.workspace = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
        self.workspace_bytes = nbytes
        n = net.n_params
        # gradient + one slot for the loss: the data-parallel exchange is ONE all-reduce of this buffer (dist.DPStep)
        self._flat = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        self.grads = self._flat[:n]
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.loss_steps = torch.zeros(msteps, dtype=torch.float32, device=dev)
        self.iters_fwd = torch.zeros(msteps * B, dtype=torch.int32, device=dev)
        self.iters_bwd = torch.zeros(msteps * B, dtype=torch.int32, device=dev)
        self.scratch = torch.zeros(64, dtype=torch.float32, device=dev)
        self.t = 0
        self.clip_norm = 1e-3 if clip_grad else 0.0      # karman_train.py:453
        self.beta1, self.beta2, self.eps = beta1, beta2, eps
        self._offsets = (C.c_int64 * len(net.offsets))(*[int(o) for o in net.offsets])
        self.final = None
        self.use_graph = use_graph
        self._graphs = {}           # want_final -> (key, handle): replayable hipGraphs of the whole fwd+bwd
        self._captures = 0          # re-captures caused by MOVED input buffers so far
        self._stage = None          # internal input buffers, used once the caller's buffers turn out not to be persistent
        self._fin = None
        self._want_final = False
        self._eager = False
        self._dp = DPStep(self._fwd_bwd_flat, self._apply_flat, group=group, comm=comm, flat=self._flat)

    def __del__(self):
        try:
            for _, h in self._graphs.values():
                self.lib.sol_train_graph_destroy(h)
        except Exception:
            pass

    # ---- the two halves of a step -------------------------------------------------------
    def fwd_bwd(self, d0, vy0, vx0, re, gt_vy, gt_vx, want_final=False, eager=False):
        """Inputs: d0 [B,Y,X], vy0 [B,Y+1,X], vx0 [B,Y,X+1], re [B], gt_vy [msteps,B,Y+1,X],
        gt_vx [msteps,B,Y,X+1] (fp32 CUDA, contiguous).  Fills self.grads / self.loss_steps and
        returns the scalar loss tensor (sum of per-step l2 losses / msteps, karman_train.py:436).
        The hipGraph bakes the input pointers in: pass PERSISTENT buffers (copy new data into them) for zero-copy
        replays.  Buffers that move are detected; after the third re-capture the trainer copies the inputs into
        internal staging buffers instead (one D2D copy of the batch per step, ~20 MB at C3) and keeps one graph.
        eager=True launches the kernels one by one (profiling, debugging) whatever use_graph says."""
        B, Y, X, ms = self.B, self.Y, self.X, self.msteps
        assert vy0.shape == (B, Y + 1, X) and vx0.shape == (B, Y, X + 1) and d0.shape == (B, Y, X)
        assert gt_vy.shape == (ms, B, Y + 1, X) and gt_vx.shape == (ms, B, Y, X + 1) and re.shape == (B,)
        with _conv_precision_scope(self.conv_precision):
            return self._fwd_bwd(d0, vy0, vx0, re, gt_vy, gt_vx, want_final, eager)

    def _fwd_bwd(self, d0, vy0, vx0, re, gt_vy, gt_vx, want_final, eager):
        B, Y, X, ms = self.B, self.Y, self.X, self.msteps
        if self._stage is not None and not eager:
            for dst, src in zip(self._stage, (d0, vy0, vx0, re, gt_vy, gt_vx)):
                dst.copy_(src)
            d0, vy0, vx0, re, gt_vy, gt_vx = self._stage
        fin = [None, None, None]
        if want_final:
            if self._fin is None:
                self._fin = [torch.empty_like(d0), torch.empty_like(vy0), torch.empty_like(vx0)]
            fin = self._fin
        mk = self.masks
        args = [ptr(self.net.params.detach()), ptr(d0), ptr(vy0), ptr(vx0), ptr(re), ptr(mk.active), ptr(mk.inflow),
                ptr(mk.velBCy), ptr(mk.velBCyMask), mk.bc_stride, ptr(gt_vy), ptr(gt_vx),
                ptr(self.workspace), self.workspace_bytes, ptr(self.grads), ptr(self.loss_steps),
                ptr(fin[0]), ptr(fin[1]), ptr(fin[2]), ptr(self.iters_fwd), ptr(self.iters_bwd)]
        if self.use_graph and not eager:
            # all pointers are baked into the graph: re-capture only when a buffer moved
            key = tuple(a.value if isinstance(a, C.c_void_p) else a for a in args)
            slot = bool(want_final)                      # one graph per output set: toggling want_final re-captures nothing
            cur = self._graphs.get(slot)
            if cur is None or cur[0] != key:
                if cur is not None:
                    # the caller's buffers moved (fresh tensors every step): after three such re-captures (each one is a
                    # device-wide synchronisation + ~1000 node instantiations) stage the inputs instead
                    self._captures += 1
                    if self._captures >= 3 and self._stage is None:
                        self._stage = [t.clone() for t in (d0, vy0, vx0, re, gt_vy, gt_vx)]
                        return self._fwd_bwd(d0, vy0, vx0, re, gt_vy, gt_vx, want_final, eager)
                    check(self.lib.sol_train_graph_destroy(cur[1]))
                    del self._graphs[slot]
                h = C.c_void_p()
                torch.cuda.synchronize()
                with _lib.no_gc_during_capture():
                    check(self.lib.sol_train_graph_create(C.byref(self.cfg), *args, C.byref(h)))
                self._graphs[slot] = (key, h)
            check(self.lib.sol_train_graph_launch(self._graphs[slot][1], stream()))
        else:
            check(self.lib.sol_train_fwd_bwd(C.byref(self.cfg), stream(), *args))
        self.final = fin if want_final else None
        return self.loss_steps.sum() / ms

    def apply_gradients(self, lr):
        """tf.compat.v1.train.AdamOptimizer(lr) update (+ optional per-tensor clip_by_norm)."""
        self.t += 1
        n = self.net.n_params
        check(self.lib.sol_adam_tf_step(stream(), ptr(self.net.params.detach()), ptr(self.grads), ptr(self.m), ptr(self.v),
                                        n, self.t, float(lr), self.beta1, self.beta2, self.eps, float(self.clip_norm),
                                        self._offsets, len(self.net.shapes), ptr(self.scratch)))

    # ---- data-parallel composition --------------------------------------------------------
    def _fwd_bwd_flat(self, *batch):
        loss = self.fwd_bwd(*batch, want_final=self._want_final, eager=self._eager)
        return loss, self.grads

    def _apply_flat(self, grads, lr):
        assert grads is self.grads
        self.apply_gradients(lr)

    def train_step(self, d0, vy0, vx0, re, gt_vy, gt_vx, lr, want_final=False, eager=False):
        """One training step on this rank's shard; returns the GLOBAL loss tensor.  want_final=True also produces
        the state after the last unrolled step in self.final = [density, vy, vx] (this is what makes the engine
        advect the passive density at all: like the TF graph of the reference, nothing that no output needs is run)."""
        self._want_final = want_final
        self._eager = eager
        return self._dp(d0, vy0, vx0, re, gt_vy, gt_vx, lr=lr)

    # ---- algorithmic traffic of the solver part ---------------------
    def solver_algorithmic_bytes(self):
        """4*(10*Nf + 9*N + 11*N*k) per forward sample-step and 4*(2*(10*Nf+9*N) + 11*N*k_bwd)
        per backward sample-step with the MEASURED CG iteration counts (k = 0 with the direct solver:
        only the stencil / advection traffic of SURVEY 8d remains)."""
        N = self.Y * self.X
        Nf = (self.Y + 1) * self.X + self.Y * (self.X + 1)
        kf = self.iters_fwd.double().sum().item()
        kb = self.iters_bwd.double().sum().item()
        nss = self.msteps * self.B
        fwd = 4.0 * ((10 * Nf + 9 * N) * nss + 11.0 * N * kf)
        bwd = 4.0 * (2 * (10 * Nf + 9 * N) * (nss - self.B) + 11.0 * N * kb)
        return fwd, bwd, kf / nss, kb / max(1, nss - self.B)


class SolRollout:
    """No-grad roll-out of solver step + CNN correction (karman_apply.py:138-158)."""

    def __init__(self, net, masks, B, Y, X, dx, std_v, std_re, dt=1.0, res=None, in_std_v=None, out_std_v=None,
                 conv_precision="split", **solver):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.conv_precision = _conv_precision_code(conv_precision)
        self.net, self.masks, self.B = net, masks, B
        kc = ops.karman_cfg(B, Y, X, dx, dt=dt, res=res, masks=masks, **solver)
        i0, i1 = (float(in_std_v[0]), float(in_std_v[1])) if in_std_v is not None else (0.0, 0.0)
        o0, o1 = (float(out_std_v[0]), float(out_std_v[1])) if out_std_v is not None else (0.0, 0.0)
        self.cfg = TrainCfg(kc, 1, float(std_v[0]), float(std_v[1]), float(std_re), float(net.slope), i0, i1, o0, o1)
        nbytes = self.lib.sol_rollout_workspace_bytes(C.byref(self.cfg))
        self.workspace = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=net.params.device)
        self.workspace_bytes = nbytes

    def run(self, d, vy, vx, re, nsteps):
        """Advances (d, vy, vx) in place by nsteps; returns CG iterations [nsteps,B]."""
        iters = torch.zeros(nsteps * self.B, dtype=torch.int32, device=d.device)
        mk = self.masks
        with _conv_precision_scope(self.conv_precision):
            check(self.lib.sol_rollout(C.byref(self.cfg), stream(), ptr(self.net.params.detach()), ptr(d), ptr(vy), ptr(vx),
                                       ptr(re), ptr(mk.active), ptr(mk.inflow), ptr(mk.velBCy), ptr(mk.velBCyMask),
                                       mk.bc_stride, nsteps, ptr(self.workspace), self.workspace_bytes, ptr(iters)))
        return iters.reshape(nsteps, self.B)


class GraphTrainer:
    """The SolTrainer call surface for networks the C++ trainer has no fused schedule for (`model_mercury`,
    karman_train.py:92-99 / `eval('model_'+...)` at :394).  The unrolled step of karman_train.py:397-457 runs as a HAND-WRITTEN
    schedule over the C ABI (`schedule="manual"`, default since round 6: _unrolled_schedule -- forward unroll keeping what the
    reverse sweep needs, reverse sweep with the weight gradients accumulated over the steps, no autograd graph) or, as the
    cross-check, COMPOSED from the differentiable HIP ops (KarmanFlow.step, to_feature, the network, to_staggered) by torch autograd
    (`schedule="autograd"`, rounds 2-5); either way captured once into a hipGraph over static buffers: a step copies the batch in
    and replays.  Same outputs as SolTrainer: the loss, `grads` (flat, Keras get_weights() order), `loss_steps`, `final` =
    [density, vy, vx] after the last step; TF-Adam with optional per-tensor clip; data parallel through the same DPStep (one SUM
    all-reduce of `grads`)."""

    def __init__(self, net, B, Y, X, msteps, std_v, std_re, res=None, clip_grad=False, beta1=0.9, beta2=0.999, eps=1e-8,
                 group=None, use_graph=True, comm=None, in_std_v=None, out_std_v=None, pressure_solver=None,
                 dx=None, dt=1.0, masks=None, cg_rtol=1e-6, cg_atol=1e-9, cg_max_iter=2000, grad_pad="replicate",
                 inflow_order="after", conv_precision="split", schedule="manual"):
        """dx: cell size (default 100 / X, the reference's `--len 100`); the domain is box[0:Y*dx, 0:X*dx] as the scripts
        build it (karman_train.py:363: box[0:len*2, 0:len]).  masks: optional SceneMasks of the caller -- its boundary
        arrays are used; its scene must be the one KarmanFlow derives from the domain (checked).  The solver options are
        those of SolTrainer and are forwarded to KarmanFlow.  schedule: "manual" | "autograd" (see the class docstring)."""
        if schedule not in ("manual", "autograd"):
            raise ValueError("schedule must be 'manual' or 'autograd'")
        self.schedule = schedule
        self._sched = None
        from . import fluid, karman
        _lib.require_gpu()
        self.lib = _lib.load()
        self.conv_precision = _conv_precision_code(conv_precision)
        self.net, self.B, self.Y, self.X, self.msteps = net, B, Y, X, msteps
        dev = self.device = net.params.device
        dx = 100.0 / X if dx is None else float(dx)
        self.dt = float(dt)
        self.dom = fluid.Domain([Y, X], box=fluid.box[0:Y * dx, 0:X * dx])
        self.sim = karman.KarmanFlow(pressure_solver=pressure_solver, cg_rtol=cg_rtol, cg_atol=cg_atol, cg_max_iter=cg_max_iter,
                                     grad_pad=grad_pad, inflow_order=inflow_order)
        self.res = X if res is None else res
        if masks is not None:
            import numpy as np
            active, inflow = self.sim.scene_arrays(self.dom)
            if not (np.array_equal(masks.active.reshape(Y, X).cpu().numpy(), active.astype(np.float32)) and
                    np.array_equal(masks.inflow.reshape(Y, X).cpu().numpy(), inflow.astype(np.float32))):
                raise ValueError("GraphTrainer composes KarmanFlow.step, whose scene (inflow box, sphere) follows from the domain; "
                                 "the given masks describe a different scene")
            self.bcv = masks.velBCy.reshape(-1, Y + 1, X, 1).cpu().numpy()
            self.bcm = masks.velBCyMask.reshape(-1, Y + 1, X, 1).cpu().numpy()
        else:
            self.bcv, self.bcm = karman.velocity_bc_masks(Y, X, batch_size=B)
        t = lambda v: torch.tensor([float(a) for a in v], dtype=torch.float32, device=dev)
        self.scale_loss = t(std_v)
        self._std_loss_host = (float(std_v[0]), float(std_v[1]))
        self.scale_in = t(list(in_std_v if in_std_v is not None else std_v) + [std_re])
        self.scale_out = t(out_std_v if out_std_v is not None else std_v)
        # (host copies: a .tolist() of a device tensor is a synchronising copy -- illegal inside a capture)
        self._scale_in_host = tuple(float(a) for a in (list(in_std_v if in_std_v is not None else std_v) + [std_re]))
        self._scale_out_host = tuple(float(a) for a in (out_std_v if out_std_v is not None else std_v))
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self._in = [f32(B, Y, X), f32(B, Y + 1, X), f32(B, Y, X + 1), f32(B), f32(msteps, B, Y + 1, X), f32(msteps, B, Y, X + 1)]
        n = net.n_params
        self._flat = f32(n + 1)                   # gradient + loss slot: one all-reduce per step (dist.DPStep)
        self.grads, self.m, self.v = self._flat[:n], f32(n), f32(n)
        self.loss_steps = f32(msteps)
        self._fin = [f32(B, Y, X), f32(B, Y + 1, X), f32(B, Y, X + 1)]
        self.final = None
        self.scratch = f32(64)
        self.t = 0
        self.clip_norm = 1e-3 if clip_grad else 0.0
        self.beta1, self.beta2, self.eps = beta1, beta2, eps
        self._offsets = (C.c_int64 * len(net.offsets))(*[int(o) for o in net.offsets])
        self.use_graph, self._graph = use_graph, None
        self._want_final, self._eager = False, False
        self._dp = DPStep(self._fwd_bwd_flat, self._apply_flat, group=group, comm=comm, flat=self._flat)

    def _unrolled(self):
        if self.schedule == "manual":
            with torch.no_grad():                   # nothing here is differentiated by torch: the reverse sweep is written out
                return self._unrolled_schedule()
        return self._unrolled_autograd()

    def _unrolled_schedule(self):
        """karman_train.py:397-457 differentiated by hand (the 2-D counterpart of karman3d.Karman3DTrainer._unrolled_schedule; train.hip does
        the same in C++ for model_mars_moon).  Forward, per unrolled step i: sol_karman_step_fwd (saves the post-diffusion velocity, writes
        the SCALED features itself) -> the network's forward launches (schedule2d.NetSchedule2D) -> velocity += out_std * to_staggered(out)
        -> loss_i and d loss_i / d v_i in one pass (sol_l2_loss_fwd_bwd, gradient pre-scaled by 1 / msteps).  Reverse, i = n-1 .. 0:
        G = d loss_i / d v_i + (adjoint of step i+1 w.r.t. its input) -> d out = out_std * G at the corrected faces -> the network's reverse
        sweep (weight gradients accumulated over the steps in the layers' partial buffers) -> sol_karman_step_bwd, which adds the feature
        gradient / in_std itself (dfeat).  One reduce per layer at the end."""
        from .schedule2d import NetSchedule2D
        d, vy, vx, re, gt_vy, gt_vx = self._in
        B, Y, X, ms = self.B, self.Y, self.X, self.msteps
        dev = self.device
        if self._sched is None:
            self._sched = NetSchedule2D(self.net, B, Y, X)
            self._mk = self.sim._masks(self.dom, self.bcv, self.bcm, dev)
            self._kcfg = ops.karman_cfg(B, Y, X, self.dom.dx[1], dt=self.dt, res=self.res, masks=self._mk, **self.sim._solver)
            self._fs = [1.0 / float(v) for v in self._scale_in_host]
        sch, mk, cfg, lib = self._sched, self._mk, self._kcfg, self.lib
        fs3 = (C.c_float * 3)(*self._fs)
        so, sl = self._scale_out_host, self._std_loss_host
        sch.begin_step()
        keep, losses = [], []
        for i in range(ms):
            d2, vy2, vx2, svy, svx = torch.empty_like(d), torch.empty_like(vy), torch.empty_like(vx), torch.empty_like(vy), torch.empty_like(vx)
            feat = torch.empty(B, Y, X, 4, dtype=torch.float32, device=dev)
            check(lib.sol_karman_step_fwd(C.byref(cfg), stream(), ptr(d), ptr(vy), ptr(vx), ptr(re), ptr(mk.active), ptr(mk.inflow), ptr(mk.velBCy),
                                          ptr(mk.velBCyMask), mk.bc_stride, ptr(d2), ptr(vy2), ptr(vx2), ptr(svy), ptr(svx), ptr(feat), fs3, None))
            out, state = sch.forward(feat)
            vy2[:, :Y].add_(out[..., 0], alpha=so[0])             # to_staggered + add (karman_train.py:88-90, 424-426): the last row / column gets no correction
            vx2[:, :, :X].add_(out[..., 1], alpha=so[1])
            li, gi = ops.l2_loss_fwd_bwd((vy2, vx2), (gt_vy[i], gt_vx[i]), sl, gscale=1.0 / ms)
            losses.append(li.reshape(()))
            keep.append((svy, svx, state, gi))
            d, vy, vx = d2, vy2, vx2
        gin = None
        for i in range(ms - 1, -1, -1):
            svy, svx, state, G = keep[i]
            if gin is not None:
                G[0].add_(gin[0])
                G[1].add_(gin[1])
            dO = torch.stack([G[0][:, :Y] * so[0], G[1][:, :, :X] * so[1]], dim=-1)
            dfeat = sch.backward(state, dO)[..., :2].contiguous()
            oy, ox = torch.empty_like(svy), torch.empty_like(svx)
            check(lib.sol_karman_step_bwd(C.byref(cfg), stream(), ptr(svy), ptr(svx), ptr(re), ptr(mk.active), ptr(mk.velBCyMask), mk.bc_stride,
                                          ptr(G[0]), ptr(G[1]), ptr(dfeat), fs3, ptr(oy), ptr(ox), None))
            gin = (oy, ox)
            keep[i] = None
        losses = _lib.stack0(losses)
        _lib.dcopy_(self.loss_steps, losses)
        _lib.dcopy_(self.grads, sch.end_step())
        for dst, src in zip(self._fin, (d, vy, vx)):
            _lib.dcopy_(dst, src)

    def _unrolled_autograd(self):
        from . import fluid, karman
        d0, vy0, vx0, re, gt_vy, gt_vx = self._in
        B, Y, X = self.B, self.Y, self.X
        stag = lambda vy, vx: torch.stack([_lib.pad_high(vy, 2), _lib.pad_high(vx, 1)], dim=-1)     # [B,Y+1,X+1,2]
        st = fluid.Fluid(self.dom, density=d0.reshape(B, Y, X, 1), velocity=stag(vy0, vx0), batch_size=B)
        losses = []
        for i in range(self.msteps):
            st = self.sim.step(st, re=re, res=self.res, velBCy=self.bcv, velBCyMask=self.bcm, dt=self.dt)
            corr = karman.to_staggered(self.net(karman.to_feature(st, re) / self.scale_in) * self.scale_out, self.dom.box)
            st = st.copied_with(velocity=st.velocity + corr)
            # l2_loss((gt.staggered - prd.staggered) / std_v), karman_train.py:428-436, channel by channel over the padded staggered tensors.
            # One kernel per step, no torch reduction (a multi-workgroup torch .sum() puts a memset node into the captured graph: ops.L2LossFn)
            vt, gt_t = st.velocity.staggered_tensor(), stag(gt_vy[i], gt_vx[i])
            losses.append(ops.l2_loss((vt[..., 0].contiguous(), vt[..., 1].contiguous()), (gt_t[..., 0].contiguous(), gt_t[..., 1].contiguous()), self._std_loss_host))
        losses = _lib.stack0(losses)
        self.net.params.grad = None
        (losses.sum() / self.msteps).backward()
        # (kernel copies: a contiguous tensor.copy_ is a hipMemcpyAsync = a memcpy node, refused by the capture guard -- _lib.dcopy_)
        _lib.dcopy_(self.loss_steps, losses)
        _lib.dcopy_(self.grads, self.net.params.grad)
        vt = st.velocity.staggered_tensor().detach()
        _lib.dcopy_(self._fin[0], st.density.data.detach().reshape(B, Y, X))
        self._fin[1].copy_(vt[:, :, :X, 0])
        self._fin[2].copy_(vt[:, :Y, :, 1])

    def fwd_bwd(self, d0, vy0, vx0, re, gt_vy, gt_vx, want_final=False, eager=False):
        with _conv_precision_scope(self.conv_precision):
            return self._fwd_bwd(d0, vy0, vx0, re, gt_vy, gt_vx, want_final, eager)

    def _fwd_bwd(self, d0, vy0, vx0, re, gt_vy, gt_vx, want_final, eager):
        for dst, src in zip(self._in, (d0, vy0, vx0, re, gt_vy, gt_vx)):
            dst.copy_(src, non_blocking=True)
        if eager or not self.use_graph:
            self._unrolled()
        else:
            if self._graph is None:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._unrolled()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                self.net.params.grad = None
                self._graph = _lib.capture_graph(self._unrolled, "GraphTrainer")      # kernel nodes only (sol_graph_check), then instantiated
            self._graph.replay()
        self.final = self._fin if want_final else None
        return self.loss_steps.sum() / self.msteps

    apply_gradients = SolTrainer.apply_gradients
    _fwd_bwd_flat = SolTrainer._fwd_bwd_flat
    _apply_flat = SolTrainer._apply_flat
    train_step = SolTrainer.train_step


def make_trainer(net, masks, B, Y, X, msteps, dx, std_v, std_re, **kw):
    """SolTrainer (the C++ schedule: model_mars_moon) or GraphTrainer (autograd composition in a hipGraph: everything else)."""
    if net.name == "mars_moon":
        return SolTrainer(net, masks, B, Y, X, msteps, dx, std_v, std_re, **kw)
    return GraphTrainer(net, B, Y, X, msteps, std_v, std_re, dx=dx, masks=masks, **kw)     # unknown keywords raise TypeError
