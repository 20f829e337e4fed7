"""Differentiable PyTorch-ROCm wrappers over the C ABI (one autograd.Function per op).

These are the composable building blocks behind the reference-shaped Python surface
(karman.py, burgers.py, model.py).  The fused whole-step entry point lives in trainer.py.
Every function here calls libsol_hip.so; there is no CPU implementation.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import KarmanCfg, SolError, check, ptr, stream

EPI_NONE, EPI_LRELU, EPI_DLRELU = 0, 1, 2
CONV_FWD, CONV_BWD_DATA = 0, 1


def karman_cfg(B, Y, X, dx, dt=1.0, res=None, cg_rtol=1e-6, cg_atol=1e-9, cg_max_iter=2000,
               grad_pad="replicate", inflow_order="after", masks=None):
    """sol_karman_cfg; `masks` (SceneMasks) supplies the coarse inverse of the CG preconditioner."""
    cfg = KarmanCfg(B, Y, X, float(dx), float(dt), float(X if res is None else res),
                    float(cg_rtol), float(cg_atol), int(cg_max_iter),
                    {"replicate": 0, "dirichlet0": 1}[grad_pad], {"after": 0, "before": 1}[inflow_order], 0, None, 0, None)
    ci = getattr(masks, "coarse_inv", None)
    if ci is not None:
        cfg.coarse_n = ci.shape[0]
        cfg.coarse_inv = ci.data_ptr()
    db = getattr(masks, "direct", None)
    if db is not None:
        cfg.direct_n = db.numel()
        cfg.direct = db.data_ptr()
    cfg._keep = (ci, db)          # the struct holds raw device pointers
    return cfg


class SceneMasks:
    """Device-resident constant masks of a scene: active (1 - obstacle), inflow rate, velBCy,
    velBCyMask (reference: KarmanFlow.__init__ karman_train.py:166-171 and :366-373)."""

    def __init__(self, active, inflow, velBCy, velBCyMask, device="cuda", precondition=True, pressure_solver="auto"):
        self.active = _lib.f32(active, device)
        self.inflow = _lib.f32(inflow, device)
        self.velBCy = _lib.f32(velBCy, device)
        self.velBCyMask = _lib.f32(velBCyMask, device)
        Y, X = self.active.shape[-2:]
        n = (Y + 1) * X
        assert self.velBCy.numel() % n == 0 and self.velBCy.numel() == self.velBCyMask.numel()
        self.bc_stride = 0 if self.velBCy.numel() == n else n
        # two-level CG preconditioner (host-prepared dense coarse inverse), when the grid allows it
        self.coarse_inv = None
        if precondition and not os.environ.get("SOL_NO_PRECOND") and Y * X <= 8192 and _lib.load().sol_karman_precond_supported(Y, X):
            from .precond import coarse_inverse
            self.coarse_inv = _lib.f32(coarse_inverse(self.active.reshape(Y, X).cpu().numpy()), device)
        # direct pressure solver (fast diagonalisation + capacitance correction) where it is built and the
        # scene qualifies; pressure_solver="cg" (or SOL_PRESSURE_SOLVER=cg) keeps the (preconditioned) CG
        self.direct = None
        self.direct_header = None
        want = os.environ.get("SOL_PRESSURE_SOLVER", pressure_solver)
        if want not in ("auto", "direct", "cg"):
            raise ValueError("pressure_solver must be 'auto', 'direct' or 'cg' (got %r)" % (want,))
        self.large = Y * X > 8192 or X > 64          # beyond the one-workgroup kernels: forward-only multi-launch path
        if want != "cg" and (self.large or _lib.load().sol_karman_direct_supported(Y, X)):
            from .precond import direct_solver_blob
            blob = direct_solver_blob(self.active.reshape(Y, X).cpu().numpy(), max_window=64 if self.large else 16)
            if blob is not None:
                self.direct = torch.from_numpy(blob).to(device)
                self.direct_header = np.ascontiguousarray(blob[:16].view(np.int32))      # host copy: sizes the large-grid launches
        if want == "direct" and self.direct is None:
            raise ValueError("the direct pressure solver does not support this scene (%dx%d)" % (Y, X))


def karman_step_large(d, vy, vx, re, cfg, masks, workspace=None):
    """Forward-only step for grids beyond the one-workgroup kernels (data generation at 256 x 128,
    /root/reference/karman-2d/karman.py:98-159): sol_karman_step_fwd_large.  Returns (d, vy, vx) after the step."""
    _lib.require_gpu()
    lib = _lib.load()
    d, vy, vx, re = (_lib.f32(t) for t in (d, vy, vx, re))
    B, Y, X = cfg.B, cfg.Y, cfg.X
    assert vy.shape == (B, Y + 1, X) and vx.shape == (B, Y, X + 1) and d.shape == (B, Y, X) and re.shape == (B,)
    nbytes = lib.sol_karman_step_large_workspace_bytes(C.byref(cfg))
    if workspace is None or workspace.numel() * 4 < nbytes:
        workspace = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=vy.device)
    d_out, vy_out, vx_out = torch.empty_like(d), torch.empty_like(vy), torch.empty_like(vx)
    check(lib.sol_karman_step_fwd_large(C.byref(cfg), stream(), ptr(d), ptr(vy), ptr(vx), ptr(re),
                                        ptr(masks.active), ptr(masks.inflow), ptr(masks.velBCy), ptr(masks.velBCyMask),
                                        masks.bc_stride, ptr(d_out), ptr(vy_out), ptr(vx_out), None, None,
                                        masks.direct_header.ctypes.data_as(C.c_void_p), ptr(workspace), workspace.numel() * 4))
    return d_out, vy_out, vx_out


def _scale3(vals):
    return (C.c_float * 3)(*[float(v) for v in vals])


class KarmanStepFn(torch.autograd.Function):
    """(d, vy, vx) [B,Y,X] / [B,Y+1,X] / [B,Y,X+1] -> one simulator_lo.step(...)."""

    @staticmethod
    def forward(ctx, d, vy, vx, re, cfg, masks, info):
        _lib.require_gpu()
        lib = _lib.load()
        d, vy, vx, re = (_lib.f32(t) for t in (d, vy, vx, re))
        B, Y, X = cfg.B, cfg.Y, cfg.X
        assert vy.shape == (B, Y + 1, X) and vx.shape == (B, Y, X + 1) and d.shape == (B, Y, X) and re.shape == (B,)
        d_out = torch.empty_like(d)
        vy_out = torch.empty_like(vy)
        vx_out = torch.empty_like(vx)
        svy = torch.empty_like(vy)
        svx = torch.empty_like(vx)
        iters = torch.empty(B, dtype=torch.int32, device=vy.device)
        check(lib.sol_karman_step_fwd(C.byref(cfg), stream(), ptr(d), ptr(vy), ptr(vx), ptr(re),
                                      ptr(masks.active), ptr(masks.inflow), ptr(masks.velBCy), ptr(masks.velBCyMask),
                                      masks.bc_stride, ptr(d_out), ptr(vy_out), ptr(vx_out), ptr(svy), ptr(svx),
                                      None, None, ptr(iters)))
        ctx.save_for_backward(svy, svx, re)
        ctx.cfg, ctx.masks, ctx.info = cfg, masks, info
        if info is not None:
            info["iterations"] = iters
        ctx.mark_non_differentiable(d_out)
        return d_out, vy_out, vx_out

    @staticmethod
    def backward(ctx, _gd, gvy, gvx):
        lib = _lib.load()
        svy, svx, re = ctx.saved_tensors
        cfg, masks = ctx.cfg, ctx.masks
        gvy = torch.zeros_like(svy) if gvy is None else gvy.contiguous()
        gvx = torch.zeros_like(svx) if gvx is None else gvx.contiguous()
        oy = torch.empty_like(svy)
        ox = torch.empty_like(svx)
        iters = torch.empty(cfg.B, dtype=torch.int32, device=svy.device)
        check(lib.sol_karman_step_bwd(C.byref(cfg), stream(), ptr(svy), ptr(svx), ptr(re), ptr(masks.active),
                                      ptr(masks.velBCyMask), masks.bc_stride, ptr(gvy), ptr(gvx), None, None,
                                      ptr(oy), ptr(ox), ptr(iters)))
        if ctx.info is not None:
            ctx.info["iterations_bwd"] = iters
        return None, oy, ox, None, None, None, None


def karman_step(d, vy, vx, re, cfg, masks, info=None):
    return KarmanStepFn.apply(d, vy, vx, re, cfg, masks, info)


# --------------------------------------------------------------------------------------
# conv 5x5
# --------------------------------------------------------------------------------------
def _pack(w_hwio, cin_run, cout_run, mode):
    lib = _lib.load()
    n = lib.sol_conv5x5_packed_floats(cin_run, cout_run, mode)
    out = torch.empty(n, dtype=torch.float32, device=w_hwio.device)
    check(lib.sol_conv5x5_pack(stream(), ptr(w_hwio), cin_run, cout_run, mode, ptr(out)))
    return out


def _pad_channels(x, c):
    if x.shape[-1] == c:
        return x.contiguous()
    return torch.nn.functional.pad(x, (0, c - x.shape[-1])).contiguous()


def conv5x5_raw(x, packed, bias, residual, act_ref, cout, epilogue, slope):
    lib = _lib.load()
    B, H, W, cin = x.shape
    y = torch.empty(B, H, W, cout, dtype=torch.float32, device=x.device)
    check(lib.sol_conv5x5(stream(), ptr(x), ptr(packed), ptr(bias), ptr(residual), ptr(act_ref), ptr(y),
                          B, H, W, cin, cout, epilogue, float(slope)))
    return y


AMAX_SLOTS = 256         # SOL_AMAX_SLOTS of csrc/common.hpp (include/sol_hip.h: sol_conv5x5_scaled)


def absmax_slots(x):
    """[AMAX_SLOTS] int32 slots holding the bit pattern of max|x| (the form sol_conv5x5_scaled consumes).  In the fused
    trainer the producing kernel publishes this; here it costs one reduction pass."""
    assert _lib.load().sol_absmax_slots() == AMAX_SLOTS
    slots = torch.zeros(AMAX_SLOTS, dtype=torch.int32, device=x.device)
    slots[0] = x.detach().abs().max().to(torch.float32).view(torch.int32)
    return slots


def conv5x5_scaled_raw(x, packed, bias, residual, act_ref, cout, epilogue, slope, x_absmax, y_absmax=None):
    """sol_conv5x5_scaled: fp16 three-product MFMA path when x_absmax is given (32 input channels, W % 64 == 0)."""
    lib = _lib.load()
    B, H, W, cin = x.shape
    y = torch.empty(B, H, W, cout, dtype=torch.float32, device=x.device)
    check(lib.sol_conv5x5_scaled(stream(), ptr(x), ptr(packed), ptr(bias), ptr(residual), ptr(act_ref), ptr(y),
                                 B, H, W, cin, cout, epilogue, float(slope), ptr(x_absmax), ptr(y_absmax)))
    return y


class Conv5x5Fn(torch.autograd.Function):
    """y = act(conv5x5_same(x, w) + b (+ residual)), NHWC, w in Keras HWIO layout."""

    @staticmethod
    def forward(ctx, x, w, b, residual, lrelu, slope):
        _lib.require_gpu()
        cin, cout = w.shape[2], w.shape[3]
        x = _lib.f32(x); w = _lib.f32(w); b = _lib.f32(b)
        cin_k = 4 if cin <= 4 else 32
        assert cin in (1, 2, 3, 4, 32), "conv5x5 supports <=4 or 32 input channels"
        ctx.cin_w = cin
        xk = _pad_channels(x, cin_k)
        packed = _pack(w, cin, cout, CONV_FWD)
        res = None if residual is None else _lib.f32(residual)
        y = conv5x5_raw(xk, packed, b, res, None, cout, EPI_LRELU if lrelu else EPI_NONE, slope)
        ctx.save_for_backward(xk, w, y)
        ctx.meta = (cin, cout, cin_k, lrelu, slope, residual is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        xk, w, y = ctx.saved_tensors
        cin, cout, cin_k, lrelu, slope, has_res = ctx.meta
        B, H, W, _ = xk.shape
        dz = gy.contiguous()
        if lrelu:
            dz = dz * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))
        # weight / bias gradient
        cout_k = cout if cout in (2, 32) else None
        assert cout_k is not None, "conv5x5 backward-weight supports 2 or 32 output channels"
        nws = lib.sol_conv5x5_bwd_weight_ws_floats(B, H, W, cin_k, cout)
        part = torch.zeros(nws, dtype=torch.float32, device=xk.device)
        check(lib.sol_conv5x5_bwd_weight(stream(), ptr(xk), ptr(dz), ptr(part), B, H, W, cin_k, cout))
        dw = torch.empty_like(w)
        db = torch.empty(cout, dtype=torch.float32, device=xk.device)
        check(lib.sol_conv5x5_bwd_weight_reduce(stream(), ptr(part), ptr(dw), ptr(db), B, H, W, cin, cout, 0))
        # data gradient: run-conv with cin_run = cout (padded to 4/32), cout_run = cin
        dzk = _pad_channels(dz, 4 if cout <= 4 else 32)
        packed = _pack(w, cout, cin, CONV_BWD_DATA)
        dx = conv5x5_raw(dzk, packed, None, None, None, cin, EPI_NONE, slope)
        return dx, dw, db, (dz if has_res else None), None, None


def conv5x5(x, w, b, residual=None, lrelu=False, slope=0.3):
    return Conv5x5Fn.apply(x, w, b, residual, lrelu, slope)


# --------------------------------------------------------------------------------------
# Burgers step
# --------------------------------------------------------------------------------------
def circulant_diffusion_matrix(n, amount):
    """Real symmetric circulant matrix of PhiFlow's periodic diffusion along one axis of
    length n: ifft(fft(.) * exp(-(2 pi k)^2 amount)), k = fftfreq(n)  (dx = 1)."""
    k = np.fft.fftfreq(n)
    col = np.fft.ifft(np.exp(-(2 * np.pi) ** 2 * k ** 2 * amount)).real
    idx = (np.arange(n)[:, None] - np.arange(n)[None, :]) % n
    return col[idx]


class BurgersStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vy, vx, fy, fx, cfg, circ):
        _lib.require_gpu()
        lib = _lib.load()
        vy = _lib.f32(vy); vx = _lib.f32(vx)
        fy = None if fy is None else _lib.f32(fy)
        fx = None if fx is None else _lib.f32(fx)
        oy = torch.empty_like(vy)
        ox = torch.empty_like(vx)
        check(lib.sol_burgers_step_fwd(C.byref(cfg), stream(), ptr(vy), ptr(vx), ptr(fy), ptr(fx),
                                       ptr(circ[0]), ptr(circ[1]), ptr(circ[2]), ptr(circ[3]), ptr(oy), ptr(ox)))
        ctx.save_for_backward(vy, vx)
        ctx.cfg, ctx.circ, ctx.has_f = cfg, circ, fy is not None
        return oy, ox

    @staticmethod
    def backward(ctx, gy, gx):
        lib = _lib.load()
        vy, vx = ctx.saved_tensors
        cfg, circ = ctx.cfg, ctx.circ
        gy = gy.contiguous(); gx = gx.contiguous()
        oy = torch.empty_like(vy)
        ox = torch.empty_like(vx)
        check(lib.sol_burgers_step_bwd(C.byref(cfg), stream(), ptr(vy), ptr(vx),
                                       ptr(circ[0]), ptr(circ[1]), ptr(circ[2]), ptr(circ[3]),
                                       ptr(gy), ptr(gx), ptr(oy), ptr(ox)))
        dt = cfg.dt
        return oy, ox, (gy * dt if ctx.has_f else None), (gx * dt if ctx.has_f else None), None, None


def burgers_circ(Y, X, amount, device="cuda"):
    mk = lambda n: torch.as_tensor(circulant_diffusion_matrix(n, amount), dtype=torch.float32, device=device).contiguous()
    return (mk(Y + 1), mk(X), mk(Y), mk(X + 1))


def burgers_step(vy, vx, fy, fx, cfg, circ):
    return BurgersStepFn.apply(vy, vx, fy, fx, cfg, circ)


BURGERS_LDS_MAX = 64      # largest grid edge of the one-workgroup (differentiable) Burgers kernels


def burgers_step_large(vy, vx, fy, fx, cfg, circ, workspace=None):
    """Forward-only Burgers step for grids beyond the one-workgroup kernels (the reference's 128 x 128 data generation,
    /root/reference/burgers/Makefile:19-29): sol_burgers_step_fwd_large.  Returns (vy, vx) after the step."""
    _lib.require_gpu()
    lib = _lib.load()
    vy, vx = _lib.f32(vy), _lib.f32(vx)
    fy = None if fy is None else _lib.f32(fy)
    fx = None if fx is None else _lib.f32(fx)
    nbytes = lib.sol_burgers_step_large_workspace_bytes(C.byref(cfg))
    if workspace is None or workspace.numel() * 4 < nbytes:
        workspace = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=vy.device)
    oy, ox = torch.empty_like(vy), torch.empty_like(vx)
    check(lib.sol_burgers_step_fwd_large(C.byref(cfg), stream(), ptr(vy), ptr(vx), ptr(fy), ptr(fx),
                                         ptr(circ[0]), ptr(circ[1]), ptr(circ[2]), ptr(circ[3]), ptr(oy), ptr(ox),
                                         ptr(workspace), workspace.numel() * 4))
    return oy, ox


class SplitFlatFn(torch.autograd.Function):
    """1-D tensor -> its consecutive pieces flat[b[k]:b[k+1]] (views) as ONE autograd node whose backward assembles the gradient with
    KERNEL copies (_lib.dcopy_).  Why not plain slicing: the backward of a 1-D slice is zeros(n) + narrow.copy_(g), and a copy into a
    contiguous narrow is a hipMemcpyAsync -- one MEMCPY NODE per parameter tensor and unrolled step in a captured trainer (refused by
    sol_graph_check), next to an n-sized zero fill and an n-sized add each.  Used for the flat parameter buffer of the networks
    (ConvNet.tensors, MarsMoon3D.tensors) and for the 1-D halves model_mercury takes of a bias."""

    @staticmethod
    def forward(ctx, flat, bounds):
        ctx.bounds = tuple(int(b) for b in bounds)
        ctx.n = flat.numel()
        return tuple(flat[ctx.bounds[k]:ctx.bounds[k + 1]] for k in range(len(ctx.bounds) - 1))

    @staticmethod
    def backward(ctx, *gs):
        b = ctx.bounds
        ref = next(g for g in gs if g is not None)
        out = torch.empty(ctx.n, dtype=ref.dtype, device=ref.device)
        if b[0] > 0:
            out[:b[0]].zero_()
        if b[-1] < ctx.n:
            out[b[-1]:].zero_()
        for k, g in enumerate(gs):
            seg = out[b[k]:b[k + 1]]
            if g is None:
                seg.zero_()
            elif g.is_cuda and g.dtype in (torch.float32, torch.int32):
                _lib.dcopy_(seg, g.contiguous().reshape(-1))
            else:
                seg.copy_(g.reshape(-1))
        return out, None


def split_flat(flat, bounds):
    """pieces flat[bounds[k]:bounds[k+1]] of a 1-D tensor; differentiable through SplitFlatFn when `flat` requires grad"""
    if torch.is_grad_enabled() and flat.requires_grad:
        return SplitFlatFn.apply(flat, tuple(int(b) for b in bounds))
    return tuple(flat[int(bounds[k]):int(bounds[k + 1])] for k in range(len(bounds) - 1))


def l2_loss_fwd_bwd(pred, gt, std, gscale=1.0, want_grad=True, loss=None, grads=None):
    """sol_l2_loss_fwd_bwd: the loss of ONE unrolled step, karman_train.py:428-436 -- 0.5 * sum(((gt - pred) / std)^2) over the
    staggered components `pred` / `gt` (tuples of 1..3 device tensors, e.g. (v_y [B,Y+1,X], v_x [B,Y,X+1])), `std` one scale per
    component.  Returns (loss [1] device tensor, tuple of d loss / d pred times gscale, or None).  `loss` / `grads` given: accumulated into."""
    lib = _lib.load()
    nc = len(pred)
    assert 1 <= nc <= 3 and len(gt) == nc and len(std) == nc
    pred = [_lib.f32(t) for t in pred]
    gt = [_lib.f32(t) for t in gt]
    for c in range(nc):     # the kernel reads n[c] = pred[c].numel() elements of BOTH: a mismatch would be an out-of-bounds device read
        if gt[c].shape != pred[c].shape or gt[c].device != pred[c].device:
            raise SolError("l2_loss_fwd_bwd: component %d: gt %s on %s does not match pred %s on %s" % (
                c, tuple(gt[c].shape), gt[c].device, tuple(pred[c].shape), pred[c].device))
        if grads is not None and (grads[c].shape != pred[c].shape or grads[c].device != pred[c].device):
            raise SolError("l2_loss_fwd_bwd: component %d: grads %s does not match pred %s" % (c, tuple(grads[c].shape), tuple(pred[c].shape)))
    acc_g = grads is not None
    g = list(grads) if acc_g else ([torch.empty_like(t) for t in pred] if want_grad else None)
    acc_l = loss is not None
    if loss is None:
        loss = torch.empty(1, dtype=torch.float32, device=pred[0].device)
    scratch = torch.empty(lib.sol_l2_loss_scratch_floats(), dtype=torch.float32, device=pred[0].device)
    arr = lambda ts: (C.c_void_p * nc)(*[ptr(t) for t in ts])
    check(lib.sol_l2_loss_fwd_bwd(stream(), nc, arr(pred), arr(gt), arr(g) if g is not None else None,
                                  (C.c_int64 * nc)(*[t.numel() for t in pred]), (C.c_float * nc)(*[float(v) for v in std]),
                                  float(gscale), int(acc_g), ptr(loss), int(acc_l), ptr(scratch)))
    return loss, (tuple(g) if g is not None else None)


class L2LossFn(torch.autograd.Function):
    """tf.nn.l2_loss((gt - prd) / std) of one unrolled step over 1..3 staggered components (karman_train.py:428-436) as ONE
    autograd node: sol_l2_loss_fwd_bwd computes the value and d loss / d prd in one pass, backward scales the saved gradient.
    Why the trainers that are captured into a hipGraph use THIS and not `(diff * diff).sum()`: a torch reduction over more
    elements than one workgroup handles allocates semaphores and clears them with cudaMemsetAsync -- a MEMSET NODE in the captured
    graph, and memset nodes of a replayed hipGraph are unreliable on ROCm 7.2: after a few replays the
    reduction folded its partial sums early and the reported per-step losses were 0.5x / 2x the true values while state and
    gradient stayed right (found by the full-size SOL-16 test).  This kernel has no memset, no atomics and a fixed summation order."""

    @staticmethod
    def forward(ctx, std, n, *tensors):
        pred, gt = tensors[:n], tensors[n:]
        loss, g = l2_loss_fwd_bwd(pred, gt, std)
        ctx.save_for_backward(*g)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        return (None, None) + tuple(gl * g for g in ctx.saved_tensors) + (None,) * len(ctx.saved_tensors)


def l2_loss(pred, gt, std):
    """0.5 * sum_c sum(((gt_c - pred_c) / std_c)^2), differentiable w.r.t. the `pred` tensors (tuples of 1..3 device tensors)."""
    pred = tuple(_lib.f32(t) for t in pred)
    gt = tuple(_lib.f32(t.detach()) for t in gt)
    return L2LossFn.apply(tuple(float(v) for v in std), len(pred), *pred, *gt)
