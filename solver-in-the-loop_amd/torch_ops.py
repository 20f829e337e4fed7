"""torch.library registration of the C-ABI ops: `torch.ops.sol.*` (north star: "exposed to Python
through PyTorch-ROCm custom ops with a hand-written backward for each solver op").

The ops call the same entry points of libsol_hip.so as ops.py (ctypes, raw device pointers; no torch types cross the C
ABI) and carry the hand-written adjoints (sol_karman_step_bwd, sol_burgers_step_bwd, sol_conv5x5 backward-data / -weight)
through torch.library.register_autograd, so they compose with any other PyTorch op and show up in the dispatcher
(torch.ops.sol.karman_step, .conv5x5, .burgers_step, .adam_tf_step).  Scene constants (masks, solver blobs, the cfg
struct) are not tensors: they are registered once with register_scene() and referred to by an integer handle."""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import check, ptr, stream

_SCENES = {}
_LIB = torch.library.Library("sol", "DEF")
_LIB.define("karman_step(Tensor d, Tensor vy, Tensor vx, Tensor re, int scene) -> (Tensor, Tensor, Tensor)")
_LIB.define("karman_step_bwd(Tensor svy, Tensor svx, Tensor re, Tensor gvy, Tensor gvx, int scene) -> (Tensor, Tensor)")
_LIB.define("karman_step_fwd_saved(Tensor d, Tensor vy, Tensor vx, Tensor re, int scene) -> (Tensor, Tensor, Tensor, Tensor, Tensor)")
_LIB.define("conv5x5(Tensor x, Tensor w, Tensor b, Tensor? residual, bool lrelu, float slope) -> Tensor")
_LIB.define("karman3d_step(Tensor d, Tensor vy, Tensor vx, Tensor vz, Tensor re, int scene) -> (Tensor, Tensor, Tensor, Tensor)")
_LIB.define("conv3d(Tensor x, Tensor w, Tensor b, Tensor? residual, bool lrelu, float slope) -> Tensor")
_LIB.define("burgers_step(Tensor vy, Tensor vx, Tensor? fy, Tensor? fx, float dx, float dt, float nu) -> (Tensor, Tensor)")
_LIB.define("l2_loss(Tensor vy, Tensor vx, Tensor gt_vy, Tensor gt_vx, float std_vy, float std_vx) -> Tensor")
_LIB.define("adam_tf_step(Tensor(a!) params, Tensor grads, Tensor(b!) m, Tensor(c!) v, int t, float lr, float beta1, float beta2, float eps) -> ()")


def register_scene(cfg, masks):
    """-> handle of (sol_karman_cfg, SceneMasks) for torch.ops.sol.karman_step."""
    h = len(_SCENES) + 1
    _SCENES[h] = (cfg, masks)
    return h


def _karman_fwd_saved(d, vy, vx, re, scene):
    cfg, masks = _SCENES[scene]
    lib = _lib.load()
    d, vy, vx, re = (_lib.f32(t) for t in (d, vy, vx, re))
    d_out, vy_out, vx_out = torch.empty_like(d), torch.empty_like(vy), torch.empty_like(vx)
    svy, svx = torch.empty_like(vy), torch.empty_like(vx)
    check(lib.sol_karman_step_fwd(C.byref(cfg), stream(), ptr(d), ptr(vy), ptr(vx), ptr(re), ptr(masks.active), ptr(masks.inflow),
                                  ptr(masks.velBCy), ptr(masks.velBCyMask), masks.bc_stride, ptr(d_out), ptr(vy_out), ptr(vx_out),
                                  ptr(svy), ptr(svx), None, None, None))
    return d_out, vy_out, vx_out, svy, svx


def _karman_bwd(svy, svx, re, gvy, gvx, scene):
    cfg, masks = _SCENES[scene]
    lib = _lib.load()
    gvy, gvx = gvy.contiguous(), gvx.contiguous()
    oy, ox = torch.empty_like(svy), torch.empty_like(svx)
    check(lib.sol_karman_step_bwd(C.byref(cfg), stream(), ptr(svy), ptr(svx), ptr(re), ptr(masks.active), ptr(masks.velBCyMask),
                                  masks.bc_stride, ptr(gvy), ptr(gvx), None, None, ptr(oy), ptr(ox), None))
    return oy, ox


def _karman_step(d, vy, vx, re, scene):
    return _karman_fwd_saved(d, vy, vx, re, scene)[:3]


_LIB.impl("karman_step_fwd_saved", _karman_fwd_saved, "CUDA")
_LIB.impl("karman_step_bwd", _karman_bwd, "CUDA")
_LIB.impl("karman_step", _karman_step, "CUDA")


class _KarmanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, vy, vx, re, scene):
        d_out, vy_out, vx_out, svy, svx = torch.ops.sol.karman_step_fwd_saved(d, vy, vx, re, scene)
        ctx.save_for_backward(svy, svx, re)
        ctx.scene = scene
        ctx.mark_non_differentiable(d_out)
        return d_out, vy_out, vx_out

    @staticmethod
    def backward(ctx, _gd, gvy, gvx):
        svy, svx, re = ctx.saved_tensors
        gvy = torch.zeros_like(svy) if gvy is None else gvy
        gvx = torch.zeros_like(svx) if gvx is None else gvx
        oy, ox = torch.ops.sol.karman_step_bwd(svy, svx, re, gvy, gvx, ctx.scene)
        return None, oy, ox, None, None


_LIB.impl("karman_step", lambda d, vy, vx, re, scene: _KarmanFn.apply(d, vy, vx, re, scene), "AutogradCUDA")

# conv / burgers: the autograd.Functions of ops.py already are the hand-written forward + backward pairs
_LIB.impl("conv5x5", lambda x, w, b, residual, lrelu, slope: ops.Conv5x5Fn.apply(x, w, b, residual, lrelu, slope), "AutogradCUDA")


def _burgers(vy, vx, fy, fx, dx, dt, nu):
    B, Yp1, X = vy.shape
    cfg = _lib.BurgersCfg(B, Yp1 - 1, X, float(dx), float(dt))
    circ = ops.burgers_circ(Yp1 - 1, X, dt * nu, vy.device)
    return ops.BurgersStepFn.apply(vy, vx, fy, fx, cfg, circ)


_LIB.impl("burgers_step", _burgers, "AutogradCUDA")


def _adam(params, grads, m, v, t, lr, beta1, beta2, eps):
    check(_lib.load().sol_adam_tf_step(stream(), ptr(params), ptr(grads.contiguous()), ptr(m), ptr(v), params.numel(), int(t), float(lr),
                                       float(beta1), float(beta2), float(eps), 0.0, None, 0, None))


_LIB.impl("adam_tf_step", _adam, "CUDA")


_LIB.impl("l2_loss", lambda vy, vx, gt_vy, gt_vx, sy, sx: ops.l2_loss((vy, vx), (gt_vy, gt_vx), (sy, sx)), "AutogradCUDA")


# ---- karman-3d (BASELINE configs[4]): the step with its hand-written adjoint and Conv3D(5) with forward / backward-data / weight gradient ----
_SIMS3D = {}


def register_scene3d(sim):
    """-> handle of a karman3d.Karman3DFlow (scene + direct-solver blob + workspace) for torch.ops.sol.karman3d_step."""
    h = len(_SIMS3D) + 1
    _SIMS3D[h] = sim
    return h


def _karman3d_step(d, vy, vx, vz, re, scene):
    from . import karman3d as k3
    return k3._Karman3DStepFn.apply(_SIMS3D[scene], _lib.f32(d), _lib.f32(vy), _lib.f32(vx), _lib.f32(vz), _lib.f32(re))


def _conv3d(x, w, b, residual, lrelu, slope):
    from . import karman3d as k3
    return k3._Conv3DFn.apply(x, w, b, residual, lrelu, slope, None)


_LIB.impl("karman3d_step", _karman3d_step, "AutogradCUDA")
_LIB.impl("conv3d", _conv3d, "AutogradCUDA")
