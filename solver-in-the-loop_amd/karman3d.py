"""karman-3d (BASELINE.json configs[4]): host surface of the 3-D forward path.

The reference has no 3-D code (/root/reference/README.md:37-38).  This module is the dimension-generic twin of karman.py /
model.py / trainer.SolRollout: the same scene (karman-2d/karman_train.py:166-171, 363-373) with a third axis, the same step
(`KarmanFlow.step`, :173-185), `model_mars_moon` (:101-138) with Conv3D(5) layers over 4 input channels (three velocity
components + Re) and 3 output channels, and the roll-out loop of karman_apply.py:138-158.  Forward only in this version.

Layout: density [B,Y,X,Z], v_y [B,Y+1,X,Z], v_x [B,Y,X+1,Z], v_z [B,Y,X,Z+1] (y = flow direction, z contiguous); CNN
tensors [B,Y,X,Z,C].  Everything calls libsol_hip.so (csrc/karman3d.hip); there is no CPU implementation.

Training (SOL-n, the reverse sweep of karman_train.py:397-457 in 3-D): `Karman3DTrainer` runs a hand-written schedule over the C ABI
(forward unroll + reverse sweep, no autograd graph; `schedule="autograd"` keeps the torch composition as the cross-check).  For other hosts
`Karman3DFlow.step` and `conv3d` are also torch.autograd Functions over the HIP adjoints (sol_karman3d_step_bwd; Conv3D backward-data = sol_conv3d on flipped weights, weight gradient
= five passes of the 2-D weight-gradient kernels), composed by `Karman3DTrainer` exactly as the reference composes its graph.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import Karman3DCfg, check, ptr, stream

EPI_NONE, EPI_LRELU, EPI_DLRELU = 0, 1, 2


def scene_arrays3d(Y, X, Z, length=100.0, obstacle="sphere"):
    """(active, inflow) [Y,X,Z] float64: Inflow(box[5:10, 25:75, 25:75]) and Obstacle(Sphere([50, 50, 50], 10)) -- the
    dimension-generic form of karman_train.py:169-170 -- on the domain box[0:2*len, 0:len, 0:len] (:363).  Cell centres
    inside count (inclusive box bounds, dist^2 <= r^2), as in the 2-D scene."""
    if obstacle != "sphere":
        raise NotImplementedError("the HIP path takes window-sized obstacles (sphere); obstacle=%r is an oracle-only option" % (obstacle,))
    if not (Y == 2 * X and Z == X):
        raise ValueError("karman-3d domain is box[0:2*len, 0:len, 0:len]: resolution must be (2*res, res, res)")
    dx = length / X
    c = [(np.arange(n) + 0.5) * dx for n in (Y, X, Z)]
    YC, XC, ZC = np.meshgrid(*c, indexing="ij")
    s = length / 100.0
    inflow = ((YC >= 5 * s) & (YC <= 10 * s) & (XC >= 25 * s) & (XC <= 75 * s) & (ZC >= 25 * s) & (ZC <= 75 * s)).astype(np.float64)
    obst = (((YC - 50 * s) ** 2 + (XC - 50 * s) ** 2 + (ZC - 50 * s) ** 2) <= (10 * s) ** 2).astype(np.float64)
    return 1.0 - obst, inflow


def velocity_bc_masks3d(Y, X, Z):
    """velBCy / velBCyMask of karman_train.py:366-373 with the third axis: the two inflow-side planes and the four lateral
    walls of the flow component, [Y+1,X,Z], values 1."""
    vn = np.zeros((Y + 1, X, Z))
    vn[0:2, :, :] = 1.0
    vn[:, 0, :] = 1.0
    vn[:, -1, :] = 1.0
    vn[:, :, 0] = 1.0
    vn[:, :, -1] = 1.0
    return vn, np.copy(vn)


class Scene3D:
    """Device-resident constants of a karman-3d scene: masks + the direct pressure-solver blob (precond3d)."""

    def __init__(self, Y, X, Z, length=100.0, device="cuda", velBCy=None, velBCyMask=None):
        from .precond3d import direct_solver_blob3d
        self.Y, self.X, self.Z = Y, X, Z
        self.dx = length / X
        active, inflow = scene_arrays3d(Y, X, Z, length)
        bcv, bcm = velocity_bc_masks3d(Y, X, Z)
        if velBCy is not None:
            bcv, bcm = np.asarray(velBCy, dtype=np.float64), np.asarray(velBCyMask, dtype=np.float64)
        n = (Y + 1) * X * Z
        if bcv.size % n or bcv.size != bcm.size:
            raise ValueError("velBCy / velBCyMask must be [Y+1,X,Z] or [B,Y+1,X,Z]")
        self.bc_stride = 0 if bcv.size == n else n
        self.active_np = active
        self.active = _lib.f32(active, device)
        self.inflow = _lib.f32(inflow, device)
        self.velBCy = _lib.f32(bcv, device)
        self.velBCyMask = _lib.f32(bcm, device)
        blob = direct_solver_blob3d(active)
        if blob is None:
            raise ValueError("the direct pressure solver does not support this scene (%dx%dx%d)" % (Y, X, Z))
        self.direct = torch.from_numpy(blob).to(device)
        self.direct_header = np.ascontiguousarray(blob[:16].view(np.int32))


class Karman3DFlow:
    """`simulator.step(...)` of the 3-D scene: one sol_karman3d_step_fwd."""

    def __init__(self, scene, batch_size, dt=1.0, res=None, grad_pad="replicate", inflow_order="after"):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.scene, self.B = scene, batch_size
        s = scene
        self.cfg = Karman3DCfg(batch_size, s.Y, s.X, s.Z, float(s.dx), float(dt), float(s.X if res is None else res),
                               {"replicate": 0, "dirichlet0": 1}[grad_pad], {"after": 0, "before": 1}[inflow_order],
                               s.direct.numel(), s.direct.data_ptr())
        nbytes = max(self.lib.sol_karman3d_step_workspace_bytes(C.byref(self.cfg)), self.lib.sol_karman3d_step_bwd_workspace_bytes(C.byref(self.cfg)))
        self.workspace = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=s.active.device)
        self.workspace_bytes = nbytes

    def _fwd(self, d, vy, vx, vz, re, saved=None, feat_out=None, feat_scale=None):
        s, B = self.scene, self.B
        Y, X, Z = s.Y, s.X, s.Z
        assert d.shape == (B, Y, X, Z) and vy.shape == (B, Y + 1, X, Z) and vx.shape == (B, Y, X + 1, Z) and vz.shape == (B, Y, X, Z + 1)
        assert re.shape == (B,)
        out = [torch.empty_like(t) for t in (d, vy, vx, vz)]
        fs = None
        if feat_out is not None:
            assert feat_out.shape == (B, Y, X, Z, 4) and feat_out.is_contiguous()
            fs = (C.c_float * 4)(*[float(v) for v in feat_scale])
        sv = saved if saved is not None else (None, None, None)
        check(self.lib.sol_karman3d_step_fwd(C.byref(self.cfg), stream(), ptr(d), ptr(vy), ptr(vx), ptr(vz), ptr(re),
                                             ptr(s.active), ptr(s.inflow), ptr(s.velBCy), ptr(s.velBCyMask), s.bc_stride,
                                             ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), ptr(sv[0]), ptr(sv[1]), ptr(sv[2]),
                                             ptr(feat_out), fs,
                                             s.direct_header.ctypes.data_as(C.c_void_p), ptr(self.workspace), self.workspace_bytes))
        return tuple(out)

    def _bwd(self, saved, re, gvy, gvx, gvz):
        s = self.scene
        gi = [torch.empty_like(t) for t in saved]
        check(self.lib.sol_karman3d_step_bwd(C.byref(self.cfg), stream(), ptr(saved[0]), ptr(saved[1]), ptr(saved[2]), ptr(re),
                                             ptr(s.active), ptr(s.velBCyMask), s.bc_stride, ptr(gvy), ptr(gvx), ptr(gvz),
                                             ptr(gi[0]), ptr(gi[1]), ptr(gi[2]),
                                             s.direct_header.ctypes.data_as(C.c_void_p), ptr(self.workspace), self.workspace_bytes))
        return gi

    def step(self, d, vy, vx, vz, re, feat_out=None, feat_scale=None):
        """(d, vy, vx, vz) -> new tensors after one solver step.  Differentiable with respect to the velocity (the density is
        a passive tracer) when a velocity input requires grad; feat_out [B,Y,X,Z,4] (optional, no-grad use) receives the
        scaled features."""
        d, vy, vx, vz, re = (_lib.f32(t) for t in (d, vy, vx, vz, re))
        if torch.is_grad_enabled() and (vy.requires_grad or vx.requires_grad or vz.requires_grad):
            if feat_out is not None:
                raise ValueError("feat_out is the fused no-grad feature output: build the features with to_feature3d() when training")
            return _Karman3DStepFn.apply(self, d, vy, vx, vz, re)
        return self._fwd(d, vy, vx, vz, re, None, feat_out, feat_scale)


class _Karman3DStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sim, d, vy, vx, vz, re):
        saved = [torch.empty_like(t) for t in (vy, vx, vz)]
        out = sim._fwd(d, vy.contiguous(), vx.contiguous(), vz.contiguous(), re, saved)
        ctx.sim = sim
        ctx.save_for_backward(re, *saved)
        ctx.mark_non_differentiable(out[0])
        return out

    @staticmethod
    def backward(ctx, _gd, gvy, gvx, gvz):
        re, sy, sx, sz = ctx.saved_tensors
        z = lambda g, t: torch.zeros_like(t) if g is None else g.contiguous()
        gi = ctx.sim._bwd((sy, sx, sz), re, z(gvy, sy), z(gvx, sx), z(gvz, sz))
        return None, None, gi[0], gi[1], gi[2], None


class _ToFeature3DFn(torch.autograd.Function):
    """to_feature3d as one node with a backward made of kernels only: the autograd backward of `vy[:, :Y]` is zeros + a copy into a
    narrow, and with B = 1 that narrow is contiguous -- a hipMemcpyAsync, i.e. a memcpy node per unrolled step in the captured trainer
    (refused by sol_graph_check).  Here: zero-padding of the strided channel slices (fill + strided copy kernels)."""

    @staticmethod
    def forward(ctx, vy, vx, vz, re):
        Y, X, Z = vx.shape[1], vy.shape[2], vy.shape[3]
        st = torch.stack([vy[:, :Y], vx[:, :, :X], vz[..., :Z]], dim=-1)
        return torch.cat([st, re.reshape(-1, 1, 1, 1, 1).expand(-1, Y, X, Z, 1).to(st.dtype)], dim=-1)

    @staticmethod
    def backward(ctx, g):
        return _lib.pad_high(g[..., 0], 1), _lib.pad_high(g[..., 1], 2), _lib.pad_high(g[..., 2], 3), None


def to_feature3d(vy, vx, vz, re):
    """karman_train.py:77-86 with three components: [B,Y,X,Z,4] = the components at the low faces of every cell + Re."""
    return _ToFeature3DFn.apply(vy, vx, vz, re)


def to_staggered3d(t):
    """karman_train.py:88-90 with three components: zero padding at the high end of each component's own axis."""
    # (cat with zeros, not F.pad: its backward clone()s a narrow of the gradient -- contiguous at B = 1, i.e. a memcpy node: _lib.pad_high)
    return _lib.pad_high(t[..., 0], 1), _lib.pad_high(t[..., 1], 2), _lib.pad_high(t[..., 2], 3)


# ---------------------------------------------------------------------------------------------------------------------
# model_mars_moon with Conv3D(5) layers
# ---------------------------------------------------------------------------------------------------------------------
class MarsMoon3D:
    """12 Conv3D(5, padding='same') layers, 32 features, five residual blocks (karman_train.py:101-138 in 3-D): 4 -> 32,
    10 x 32 -> 32, 32 -> 3; 1 308 355 parameters in ONE flat fp32 buffer in Keras get_weights() order (kernels DHWIO).
    Keras defaults: glorot_uniform kernels, zero biases, LeakyReLU(alpha=0.3)."""
    name = "mars_moon3d"
    slope = 0.3

    def __init__(self, cin=4, cout=3, seed=0, device="cuda"):
        self.cin, self.cout = cin, cout
        chans = [cin] + [32] * 11 + [cout]
        self.chans = chans
        self.shapes = []
        for l in range(12):
            self.shapes += [(5, 5, 5, chans[l], chans[l + 1]), (chans[l + 1],)]
        self.offsets = np.concatenate([[0], np.cumsum([int(np.prod(s)) for s in self.shapes])]).astype(np.int64)
        gen = torch.Generator().manual_seed(seed)
        parts = []
        for s in self.shapes:
            if len(s) == 5:
                lim = math.sqrt(6.0 / (125 * (s[3] + s[4])))
                parts.append(((torch.rand(s, generator=gen, dtype=torch.float64) * 2 - 1) * lim).reshape(-1))
            else:
                parts.append(torch.zeros(s, dtype=torch.float64))
        self.params = torch.cat(parts).to(device=device, dtype=torch.float32).contiguous()
        self._packed = None

    def train_packs(self):
        """Per layer (forward-packed, backward-data-packed) weights, built once and reused by every unrolled step of a
        training step (the weights only change between steps: whoever updates them resets `_tpacks`)."""
        key = self._params_key()
        if getattr(self, "_tpacks", None) is None or getattr(self, "_tpacks_key", None) != key:
            t = self.tensors()
            self._tpacks = []
            for l in range(12):
                cin, cout = self.chans[l], self.chans[l + 1]
                cin_k = 4 if cin <= 4 else 32
                w = t[2 * l].detach()
                wf = w if cin_k == cin else torch.nn.functional.pad(w, (0, 0, 0, cin_k - cin))   # the forward kernels read cin_k input channels (as pack())
                self._tpacks.append((_pack3d(wf, cin_k, cout, 0), _pack3d(w, cout, cin, 1)))
            # the two thin-INPUT layers (first layer 4 -> 32; the output layer's data gradient cout -> 32) as ONE 2-D launch with the depth
            # taps packed into the channel axis (sol_conv3d_thin): packed for the convolution that is RUN
            self._kpacks = None
            self._kpacks_out = None
            if self.thin_kpack and self.cin <= 4 and self.cout <= 4:
                self._kpacks = (_pack3d_thin(t[0].detach(), self.cin, 0), _pack3d_thin(t[22].detach(), self.cout, 1))
                # ... and into the OUTPUT channel axis (sol_conv3d_thin_out): the output layer's forward, the first layer's data gradient
                self._kpacks_out = (_pack3d_thin_out(t[22].detach(), self.cout, 0), _pack3d_thin_out(t[0].detach(), self.cin, 1))
            self._tpacks_key = key
        return self._tpacks

    thin_kpack = True          # False: the thin-input layers as five passes of the 2-D fp32-MFMA kernel (sol_conv3d), rounds 3-6

    thin_out_kpack = True      # False: the 32 -> (<= 4) layers on the eight-row Conv3D kernel with one channel tile (k_conv3d_sb8<1, 0>), rounds 4-6

    def thin_out_packs(self):
        """(output layer forward, first layer backward-data) packed for sol_conv3d_thin_out, or None"""
        self.train_packs()
        return self._kpacks_out if self.thin_out_kpack else None

    def thin_packs(self):
        """(first layer forward, output layer backward-data) packed for sol_conv3d_thin, or None (thin_kpack off / more than four channels)"""
        self.train_packs()
        return self._kpacks

    def _params_key(self):
        """Identity of the weight buffer's CONTENT as far as torch can see it: an optimizer that updates `params` in place through
        torch bumps `_version`; the library's own Adam (ctypes) does not -- Karman3DTrainer.apply_gradients resets the caches itself."""
        return (self.params.data_ptr(), self.params._version)

    def invalidate(self):
        """Drop the packed-weight caches.  MUST be called by whoever changes the CONTENT of `params` in a way torch's version counter does
        not see: writes through `params.data` (`p.data.add_()`, `load_state_dict`-style `.data.copy_`), ctypes / HIP writes into the
        buffer (the library's own Adam: Karman3DTrainer.apply_gradients calls this), external kernels.  `set_weights` calls it itself;
        in-place torch ops on `params` bump `_version` and need nothing."""
        self._packed = None
        self._packed_thin = None
        self._packed_thin_out = None
        self._tpacks = None

    fused_backward = True      # one autograd node with a hand-written reverse sweep (False: one node per layer, torch glue)

    def __call__(self, x):
        """Differentiable forward (training): x [B,Y,X,Z,4] -> [B,Y,X,Z,cout]; gradients flow to self.params (set
        `net.params.requires_grad_(True)`) and to x."""
        if self.fused_backward:
            return _MarsMoon3DFn.apply(x, self.params, self)
        p = self.tensors()
        pk = self.train_packs()
        sl = self.slope
        h = conv3d_fn(x, p[0], p[1], None, True, sl, pk[0])
        for k in range(5):
            a = conv3d_fn(h, p[2 + 4 * k], p[3 + 4 * k], None, True, sl, pk[1 + 2 * k])
            h = conv3d_fn(a, p[4 + 4 * k], p[5 + 4 * k], h, True, sl, pk[2 + 2 * k])
        return conv3d_fn(h, p[22], p[23], None, False, sl, pk[11])

    def activations(self, x):
        """The eleven 32-channel activations h0, a1, h1, ..., a5, h5 of the forward pass (no gradient; the same launches as
        __call__) -- for checks that need the LeakyReLU masks the backward pass will use."""
        with torch.no_grad():
            return _MarsMoon3DFn.run_forward(self, x)[3]

    @property
    def n_params(self):
        return int(self.offsets[-1])

    def tensors(self):
        from .ops import split_flat
        return [t.reshape(s) for t, s in zip(split_flat(self.params, self.offsets), self.shapes)]

    def get_weights(self):
        return [t.detach().cpu().numpy() for t in self.tensors()]

    def set_weights(self, weights):
        flat = np.concatenate([np.asarray(w, dtype=np.float32).reshape(-1) for w in weights])
        assert flat.size == self.n_params, "weight list does not match %s" % self.name
        with torch.no_grad():
            self.params.copy_(torch.as_tensor(flat, device=self.params.device))
        self.invalidate()

    def pack_thin(self):
        """the first layer packed for sol_conv3d_thin (inference path; cached like pack()), or None"""
        if not (self.thin_kpack and self.cin <= 4):
            return None
        key = self._params_key()
        if getattr(self, "_packed_thin", None) is None or getattr(self, "_packed_thin_key", None) != key:
            self._packed_thin_key = key
            self._packed_thin = _pack3d_thin(self.tensors()[0].detach(), self.cin, 0)
        return self._packed_thin

    def pack_thin_out(self):
        """the output layer packed for sol_conv3d_thin_out (inference path; cached like pack()), or None"""
        if not (self.thin_kpack and self.thin_out_kpack and self.cout <= 4):
            return None
        key = self._params_key()
        if getattr(self, "_packed_thin_out", None) is None or getattr(self, "_packed_thin_out_key", None) != key:
            self._packed_thin_out_key = key
            self._packed_thin_out = _pack3d_thin_out(self.tensors()[22].detach(), self.cout, 0)
        return self._packed_thin_out

    def pack(self):
        """(packed weights, padded biases) per layer in the layout the conv kernels consume; cached until set_weights."""
        key = self._params_key()
        if self._packed is None or getattr(self, "_packed_key", None) != key:
            self._packed_key = key
            lib = _lib.load()
            t = self.tensors()
            self._packed = []
            for l in range(12):
                cin, cout = self.chans[l], self.chans[l + 1]
                cin_k = 4 if cin <= 4 else 32
                w = t[2 * l]
                if cin_k != cin:
                    w = torch.nn.functional.pad(w, (0, 0, 0, cin_k - cin))
                w = w.contiguous()
                buf = torch.empty(lib.sol_conv3d_packed_floats(cin_k, cout), dtype=torch.float32, device=w.device)
                check(lib.sol_conv3d_pack(stream(), ptr(w), cin_k, cout, 0, ptr(buf)))
                self._packed.append((buf, t[2 * l + 1].contiguous(), cin_k, cout))
        return self._packed


def conv3d(x, packed, bias, residual, cout, lrelu, slope, x_absmax=None, y_absmax=None, out=None, act_ref=None):
    """sol_conv3d: x [B,Y,X,Z,cin] -> [B,Y,X,Z,cout].  act_ref: the result (conv + residual) is multiplied by LeakyReLU'(act_ref)
    (SOL_EPI_DLRELU, the backward pass's fused form) instead of being activated."""
    lib = _lib.load()
    B, D, H, W, cin = x.shape
    y = out if out is not None else torch.empty(B, D, H, W, cout, dtype=torch.float32, device=x.device)
    epi = EPI_DLRELU if act_ref is not None else (EPI_LRELU if lrelu else EPI_NONE)
    check(lib.sol_conv3d(stream(), ptr(x), ptr(packed), ptr(bias), ptr(residual), ptr(act_ref), ptr(y), B, D, H, W, cin, cout,
                         epi, float(slope), ptr(x_absmax), ptr(y_absmax)))
    return y


def conv3d_thin(x4, packed, bias, lrelu, slope, y_absmax=None, act_ref=None, out=None, ws=None):
    """sol_conv3d_thin: x4 [B,Y,X,Z,4] (zero padded input channels) -> [B,Y,X,Z,32]; the depth taps are gathered into the channel axis
    (scratch: 32 channels of the volume) and the layer runs as ONE 2-D 32 -> 32 convolution over the (X, Z) planes."""
    lib = _lib.load()
    B, D, H, W, c = x4.shape
    assert c == 4, "conv3d_thin reads four (zero padded) input channels"
    y = out if out is not None else torch.empty(B, D, H, W, 32, dtype=torch.float32, device=x4.device)
    if ws is None:
        ws = torch.empty(lib.sol_conv3d_thin_ws_floats(B, D, H, W), dtype=torch.float32, device=x4.device)
    assert ws.numel() >= lib.sol_conv3d_thin_ws_floats(B, D, H, W)
    epi = EPI_DLRELU if act_ref is not None else (EPI_LRELU if lrelu else EPI_NONE)
    check(lib.sol_conv3d_thin(stream(), ptr(x4), ptr(packed), ptr(bias), ptr(act_ref), ptr(y), ptr(ws), B, D, H, W, epi, float(slope), ptr(y_absmax)))
    return y


def conv3d_thin_out(x, packed, bias, cout, x_absmax=None, out=None, ws=None):
    """y [B,D,H,W,cout (<= 4)] = conv3d(x [B,D,H,W,32], w) + bias with the depth taps packed into the OUTPUT channel axis: one 2-D 32 -> 32
    launch + a gather over five planes (sol_conv3d_thin_out; `packed` from _pack3d_thin_out).  No activation."""
    lib = _lib.load()
    B, D, H, W, c = x.shape
    assert c == 32 and 1 <= cout <= 4
    y = out if out is not None else torch.empty(B, D, H, W, cout, dtype=torch.float32, device=x.device)
    if ws is None:
        ws = torch.empty(lib.sol_conv3d_thin_ws_floats(B, D, H, W), dtype=torch.float32, device=x.device)
    check(lib.sol_conv3d_thin_out(stream(), ptr(x), ptr(packed), ptr(bias), ptr(y), ptr(ws), B, D, H, W, cout, ptr(x_absmax)))
    return y


def _pack3d_thin_out(w, cr, mode):
    """sol_conv3d_thin_out_pack: w = the FORWARD kernel ([5,5,5,32,cr] for mode 0, [5,5,5,cr,32] for mode 1 = backward-data), cr <= 4"""
    lib = _lib.load()
    buf = torch.empty(lib.sol_conv3d_thin_packed_floats(), dtype=torch.float32, device=w.device)
    check(lib.sol_conv3d_thin_out_pack(stream(), ptr(w.contiguous()), cr, mode, ptr(buf)))
    return buf


def _pack3d_thin(w, cin_run, mode):
    """w: the FORWARD kernel ([5,5,5,cin_run,32] for mode 0, [5,5,5,32,cin_run] for mode 1 = backward-data)"""
    lib = _lib.load()
    buf = torch.empty(lib.sol_conv3d_thin_packed_floats(), dtype=torch.float32, device=w.device)
    check(lib.sol_conv3d_thin_pack(stream(), ptr(w.contiguous()), cin_run, mode, ptr(buf)))
    return buf


def _pack3d(w, cin_run, cout_run, mode):
    lib = _lib.load()
    buf = torch.empty(lib.sol_conv3d_packed_floats(cin_run, cout_run), dtype=torch.float32, device=w.device)
    check(lib.sol_conv3d_pack(stream(), ptr(w.contiguous()), cin_run, cout_run, mode, ptr(buf)))
    return buf


def _absmax(x):
    """absmax slots of x (sol_absmax: one pass at HBM speed; the tensor must stay referenced by the caller until the consumer is enqueued)"""
    slots = torch.empty(256, dtype=torch.int32, device=x.device)
    check(_lib.load().sol_absmax(stream(), ptr(x), x.numel(), ptr(slots)))
    return slots


def _pad_ch(x, c):
    return x.contiguous() if x.shape[-1] == c else torch.nn.functional.pad(x, (0, c - x.shape[-1])).contiguous()


def conv3d_thin_bwd_weight(x4, dz, cin, zmax=None, acc=None, thin_out=False):
    """dW [5,5,5,cin,32], db [32] of a thin-input layer y = conv3d(x, W) + b from x4 [B,D,H,W,4] (zero padded) and dz [B,D,H,W,32], W == 64:
    sol_conv3d_thin_bwd_weight_acc -- the depth taps packed into the channel axis, ONE pass of the 2-D 32 -> 32 weight-gradient kernel.
    thin_out=True: the thin-OUTPUT layer (32 -> cin <= 4 channels; `x4` is then the zero-padded output gradient [B,D,H,W,4], `dz` the layer's
    32-channel input and `zmax` its absmax slots): dW [5,5,5,32,cin], db [cin] (sol_conv3d_thin_out_bwd_weight_acc).
    acc = (state dict, first, last) as in conv3d_bwd_weight."""
    lib = _lib.load()
    B, D, H, W, c = x4.shape
    assert c == 4 and dz.shape[-1] == 32 and W == 64
    dev = x4.device
    shape_key = ("thin_out" if thin_out else "thin", B, D, H, W, cin, str(dev))
    if acc is not None and "part" in acc[0]:
        if acc[0].get("shape") != shape_key:
            raise _lib.SolError("conv3d_thin_bwd_weight: the accumulation state was allocated for %s, this call has %s" % (acc[0].get("shape"), shape_key))
        if not acc[1] and not acc[0].get("open"):
            raise _lib.SolError("conv3d_thin_bwd_weight: first=False on a state whose sequence is not open (the previous sweep ended with last=True)")
        part, dW, db, ws = (acc[0][k] for k in ("part", "dW", "db", "ws"))
    else:
        if acc is not None and not acc[1]:
            raise _lib.SolError("conv3d_thin_bwd_weight: the first call on a fresh accumulation state must have first=True (nothing to add onto yet)")
        part = torch.empty(lib.sol_conv3d_thin_bwd_weight_ws_floats(B, D, H, W), dtype=torch.float32, device=dev)
        ws = torch.empty(lib.sol_conv3d_thin_ws_floats(B, D, H, W), dtype=torch.float32, device=dev)
        dW = torch.empty((5, 5, 5, 32, cin) if thin_out else (5, 5, 5, cin, 32), dtype=torch.float32, device=dev)
        db = torch.empty(cin if thin_out else 32, dtype=torch.float32, device=dev)
        if acc is not None:
            acc[0].update(part=part, dW=dW, db=db, ws=ws, shape=shape_key)
    first, last = (True, True) if acc is None else (acc[1], acc[2])
    if acc is not None:
        acc[0]["open"] = not last
    if thin_out:
        check(lib.sol_conv3d_thin_out_bwd_weight_acc(stream(), ptr(dz), ptr(zmax), ptr(x4), ptr(ws), ptr(part), ptr(dW), ptr(db), B, D, H, W, cin,
                                                     0 if first else 1, 1 if last else 0))
    else:
        check(lib.sol_conv3d_thin_bwd_weight_acc(stream(), ptr(x4), ptr(dz), ptr(zmax), ptr(ws), ptr(part), ptr(dW), ptr(db), B, D, H, W, cin,
                                                 0 if first else 1, 1 if last else 0))
    return (dW, db) if last else (None, None)


def conv3d_bwd_weight(xk, dz, cin, cout, xmax=None, zmax=None, acc=None):
    """dW [5,5,5,cin,cout], db [cout] of y = conv3d(x, W) + b from xk [B,D,H,W,cin_k] (channels padded to 4 / 32) and dz
    [B,D,H,W,cout]: sol_conv3d_bwd_weight (five passes of the batched 2-D weight-gradient kernels over the shifted plane
    ranges; fp16 three-product operands for the 32 -> 32 case, scaled by the absmax of x and dz).
    acc = (state dict, first, last): the unrolled trainer's form (sol_conv3d_bwd_weight_acc) -- the partial sums of this layer live in
    `state` across the calls of a reverse sweep, the first call overwrites them, the last one reduces; returns (None, None) before the
    last call and the sum over all calls with it."""
    lib = _lib.load()
    B, D, H, W, cin_k = xk.shape
    co_k = cout if cout in (2, 32) else 32                 # the 2-D weight-gradient kernels take 2 or 32 output channels
    dzk = _pad_ch(dz, co_k)
    dev = xk.device
    shape_key = (B, D, H, W, cin_k, co_k, cin, str(dev))
    if acc is not None and "part" in acc[0]:
        # the partial sums of a reverse sweep live in the caller's state dict: they belong to ONE shape and to a sequence that was opened
        # with first=True (the C entry point cannot see the size of `partial`, so a mismatch would overrun it or add onto stale sums)
        if acc[0].get("shape") != shape_key:
            raise _lib.SolError("conv3d_bwd_weight: the accumulation state was allocated for %s, this call has %s" % (acc[0].get("shape"), shape_key))
        if not acc[1] and not acc[0].get("open"):
            raise _lib.SolError("conv3d_bwd_weight: first=False on a state whose sequence is not open (the previous sweep ended with last=True)")
        part, dW, db, scratch = (acc[0][k] for k in ("part", "dW", "db", "scratch"))
    else:
        if acc is not None and not acc[1]:
            raise _lib.SolError("conv3d_bwd_weight: the first call on a fresh accumulation state must have first=True (nothing to add onto yet)")
        part = torch.empty(lib.sol_conv3d_bwd_weight_ws_floats(B, D, H, W, cin_k, co_k), dtype=torch.float32, device=dev)
        dW = torch.empty(5, 5, 5, cin, co_k, dtype=torch.float32, device=dev)
        db = torch.empty(co_k, dtype=torch.float32, device=dev)
        scratch = torch.empty(5 * co_k, dtype=torch.float32, device=dev)
        if acc is not None:
            acc[0].update(part=part, dW=dW, db=db, scratch=scratch, shape=shape_key)
    both32 = cin_k == 32 and co_k == 32
    # (the slot tensors must outlive the call: a temporary inside ptr(...) is freed -- and its block handed to the next
    # allocation -- before the launch is even enqueued)
    # (xmax / zmax given: the slots the producing conv launches published -- no pass over the tensors)
    xmax = (xmax if xmax is not None else _absmax(xk)) if both32 else None
    zmax = (zmax if zmax is not None and dzk is dz else _absmax(dzk)) if both32 else None
    if acc is not None:
        _, first, last = acc
        acc[0]["open"] = not last
        check(lib.sol_conv3d_bwd_weight_acc(stream(), ptr(xk), ptr(dzk), ptr(xmax), ptr(zmax), ptr(part), ptr(dW), ptr(db), ptr(scratch),
                                            B, D, H, W, cin_k, co_k, cin, co_k, 0 if first else 1, 1 if last else 0))
        if not last:
            return None, None
    else:
        check(lib.sol_conv3d_bwd_weight(stream(), ptr(xk), ptr(dzk), ptr(xmax), ptr(zmax),
                                        ptr(part), ptr(dW), ptr(db), ptr(scratch), B, D, H, W, cin_k, co_k, cin, co_k))
    return (dW if co_k == cout else dW[..., :cout].contiguous()), (db if co_k == cout else _lib.dclone(db[:cout]))


class _Conv3DFn(torch.autograd.Function):
    """y = act(conv3d_same(x, w) + b (+ residual)), NDHWC, w in Keras DHWIO layout."""

    @staticmethod
    def forward(ctx, x, w, b, residual, lrelu, slope, packs):
        _lib.require_gpu()
        cin, cout = w.shape[3], w.shape[4]
        assert cin in (1, 2, 3, 4, 32), "conv3d supports <= 4 or 32 input channels"
        cin_k = 4 if cin <= 4 else 32
        xk = _pad_ch(_lib.f32(x), cin_k)
        if packs is not None and cin_k == cin:
            packed = packs[0]
        else:
            wk = _pad_ch(_lib.f32(w).permute(0, 1, 2, 4, 3), cin_k).permute(0, 1, 2, 4, 3).contiguous() if cin_k != cin else _lib.f32(w)
            packed = _pack3d(wk, cin_k, cout, 0)
        ctx.packed_bwd = packs[1] if packs is not None else None
        res = None if residual is None else _lib.f32(residual)
        # the operand's absmax selects the fp16 three-product kernels (and, for 32 -> 32 layers, the one-launch 5x5x5 kernel)
        if MarsMoon3D.thin_kpack and cin <= 4 and cout == 32 and res is None:      # the depth-packed one-launch form, as the fused network runs it
            y = conv3d_thin(xk, _pack3d_thin(_lib.f32(w), cin, 0), _lib.f32(b), lrelu, slope)
        elif MarsMoon3D.thin_kpack and MarsMoon3D.thin_out_kpack and cin == 32 and cout <= 4 and res is None and not lrelu:      # ... and the thin-output form
            y = conv3d_thin_out(xk, _pack3d_thin_out(_lib.f32(w), cout, 0), _lib.f32(b), cout, _absmax(xk))
        else:
            y = conv3d(xk, packed, _lib.f32(b), res, cout, lrelu, slope, _absmax(xk) if cin_k == 32 else None)
        ctx.save_for_backward(xk, w, y)
        ctx.meta = (cin, cout, cin_k, lrelu, slope, residual is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        xk, w, y = ctx.saved_tensors
        cin, cout, cin_k, lrelu, slope, has_res = ctx.meta
        dz = gy.contiguous()
        if lrelu:
            dz = dz * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))
        if MarsMoon3D.thin_kpack and cin <= 4 and cout == 32 and xk.shape[3] == 64:
            dW, db = conv3d_thin_bwd_weight(xk, dz, cin)
        elif MarsMoon3D.thin_kpack and cin == 32 and cout <= 4 and xk.shape[3] == 64:
            dW, db = conv3d_thin_bwd_weight(_pad_ch(dz, 4), xk, cout, thin_out=True)
        else:
            dW, db = conv3d_bwd_weight(xk, dz, cin, cout)
        # data gradient: the flipped kernel, run channels (cout -> cin)
        co_k = 4 if cout <= 4 else 32
        packed = ctx.packed_bwd if ctx.packed_bwd is not None else _pack3d(_lib.f32(w), cout, cin, 1)
        dzk = _pad_ch(dz, co_k)
        if MarsMoon3D.thin_kpack and cout <= 4 and cin == 32:
            dx = conv3d_thin(dzk, _pack3d_thin(_lib.f32(w), cout, 1), None, False, slope)
        elif MarsMoon3D.thin_kpack and MarsMoon3D.thin_out_kpack and cin <= 4 and cout == 32:
            dx = conv3d_thin_out(dzk, _pack3d_thin_out(_lib.f32(w), cin, 1), None, 4, _absmax(dzk))[..., :cin]
        else:
            dx = conv3d(dzk, packed, None, None, cin, False, slope, _absmax(dzk) if co_k == 32 else None)
        return dx, dW, db, (dz if has_res else None), None, None, None


class _MarsMoon3DFn(torch.autograd.Function):
    """model_mars_moon (3-D) as ONE autograd node.  Forward: the twelve sol_conv3d launches, every layer handing the absmax slots
    it published to its consumer (no pass over a tensor just to find its maximum).  Reverse sweep, written out by hand
    (karman_train.py:101-138 differentiated): per layer the weight gradient (sol_conv3d_bwd_weight) and ONE data-gradient launch
    whose epilogue adds the skip gradient of the residual block and multiplies by LeakyReLU'(saved activation)
    (SOL_EPI_DLRELU) -- the compare / where / multiply / add / absmax passes of the per-layer form are gone."""

    @staticmethod
    def run_forward(net, x):
        """(out, xk, amax, acts): the twelve launches; acts = the eleven 32-channel activations h0, a1, h1, ..., a5, h5"""
        _lib.require_gpu()
        p = [t.detach() for t in net.tensors()]
        pk = net.train_packs()
        sl, cout = net.slope, net.cout
        xk = _pad_ch(_lib.f32(x.detach()), 4)
        amax = torch.zeros(11, 256, dtype=torch.int32, device=xk.device)       # absmax slots of the eleven 32-channel activations
        kp = net.thin_packs()
        acts = [conv3d_thin(xk, kp[0], p[1], True, sl, amax[0]) if kp else conv3d(xk, pk[0][0], p[1], None, 32, True, sl, None, amax[0])]
        for k in range(5):
            a = conv3d(acts[-1], pk[1 + 2 * k][0], p[3 + 4 * k], None, 32, True, sl, amax[2 * k], amax[2 * k + 1])
            acts.append(a)
            acts.append(conv3d(a, pk[2 + 2 * k][0], p[5 + 4 * k], acts[-2], 32, True, sl, amax[2 * k + 1], amax[2 * k + 2]))
        ko = net.thin_out_packs()
        if ko:
            out = conv3d_thin_out(acts[-1], ko[0], p[23], cout, amax[10])
        else:
            out = conv3d(acts[-1], pk[11][0], p[23], None, cout, False, sl, amax[10], None)
        return out, xk, amax, acts

    @staticmethod
    def forward(ctx, x, params, net):
        out, xk, amax, acts = _MarsMoon3DFn.run_forward(net, x)
        ctx.net = net
        ctx.save_for_backward(xk, amax, *acts)
        return out

    @staticmethod
    def run_backward(net, xk, amax, acts, g_out, acc=None):
        """(dx [B,Y,X,Z,4 (padded input channels)], flat gradient in get_weights() order) of the network for the output gradient g_out:
        the reverse sweep written out by hand (used by the autograd node below and by Karman3DTrainer's hand-written schedule).
        acc = (list of 12 per-layer state dicts, first, last): the weight gradients are ACCUMULATED over the calls of an unrolled reverse
        sweep inside the kernels' partial buffers and reduced once, with the last call (conv3d_bwd_weight); the flat gradient is None before."""
        A = (lambda l: None) if acc is None else (lambda l: (acc[0][l], acc[1], acc[2]))
        pk = net.train_packs()
        sl, cin, cout = net.slope, net.cin, net.cout
        grads = [None] * 24
        zm = torch.zeros(11, 256, dtype=torch.int32, device=xk.device)         # absmax slots of the eleven pre-activation gradients
        g = _lib.f32(g_out).contiguous()                      # [.., cout], or already zero padded to 4 channels (the trainer's fused glue)
        kp = net.thin_packs()
        if g.shape[-1] == 4 and cout != 4:
            g4, gc = g, None
        else:
            g4, gc = (_pad_ch(g, 4) if kp else None), g
        if kp and g.shape[3] == 64:
            grads[22], grads[23] = conv3d_thin_bwd_weight(g4, acts[10], cout, zmax=amax[10], acc=A(11), thin_out=True)
        else:
            gc = gc if gc is not None else g4[..., :cout].contiguous()
            grads[22], grads[23] = conv3d_bwd_weight(acts[10], gc, 32, cout, xmax=amax[10], acc=A(11))
        # d loss / d (pre-activation of the last residual block's output) = conv3d(g, flip(w11)^T) * lrelu'(h5)
        if kp:
            dz = conv3d_thin(g4, kp[1], None, False, sl, zm[10], act_ref=acts[10])
        else:
            dz = conv3d(g4 if g4 is not None else _pad_ch(gc, 4), pk[11][1], None, None, 32, False, sl, None, zm[10], act_ref=acts[10])
        for k in range(4, -1, -1):
            a, hprev = acts[1 + 2 * k], acts[2 * k]
            # block k: h_k = lrelu(conv_b(a) + h_{k-1}), a = lrelu(conv_a(h_{k-1}))
            grads[4 + 4 * k], grads[5 + 4 * k] = conv3d_bwd_weight(a, dz, 32, 32, xmax=amax[2 * k + 1], zmax=zm[2 * k + 2], acc=A(2 + 2 * k))
            dz1 = conv3d(dz, pk[2 + 2 * k][1], None, None, 32, False, sl, zm[2 * k + 2], zm[2 * k + 1], act_ref=a)
            grads[2 + 4 * k], grads[3 + 4 * k] = conv3d_bwd_weight(hprev, dz1, 32, 32, xmax=amax[2 * k], zmax=zm[2 * k + 1], acc=A(1 + 2 * k))
            dz = conv3d(dz1, pk[1 + 2 * k][1], None, dz, 32, False, sl, zm[2 * k + 1], zm[2 * k], act_ref=hprev)
        if kp and xk.shape[3] == 64 and cin <= 4:
            grads[0], grads[1] = conv3d_thin_bwd_weight(xk, dz, cin, zmax=zm[0], acc=A(0))
        else:
            grads[0], grads[1] = conv3d_bwd_weight(xk, dz, cin, 32, acc=A(0))
        ko = net.thin_out_packs()
        if ko and xk.shape[-1] == 4:
            dx = conv3d_thin_out(dz, ko[1], None, 4, zm[0])           # (channels >= cin of the packed kernel are zero)
        else:
            dx = conv3d(dz, pk[0][1], None, None, xk.shape[-1], False, sl, zm[0], None)
        return dx, (None if grads[0] is None else torch.cat([t.reshape(-1) for t in grads]))

    @staticmethod
    def backward(ctx, g_out):
        net = ctx.net
        xk, amax, *acts = ctx.saved_tensors
        dx, flat = _MarsMoon3DFn.run_backward(net, xk, amax, acts, g_out)
        return dx[..., :net.cin], flat, None


def conv3d_fn(x, w, b, residual=None, lrelu=False, slope=0.3, packs=None):
    """packs: optional (forward-packed, backward-data-packed) weight buffers of this layer (MarsMoon3D.train_packs)."""
    return _Conv3DFn.apply(x, w, b, residual, lrelu, slope, packs)


def ops_l2_loss(pred, gt, std):
    from .ops import l2_loss
    return l2_loss(pred, gt, std)


class Karman3DTrainer:
    """SOL-n training step of the 3-D scene: msteps x [solver step -> features / std -> mars_moon3d -> velocity += std *
    to_staggered(correction)], loss = sum_i l2_loss((gt_i - prd_i) / std_v) / msteps (karman_train.py:397-447 with three
    components), reverse sweep through the HIP adjoints by torch autograd, TF-Adam on the flat parameter buffer
    (sol_adam_tf_step).  gts: [msteps] of (vy, vx, vz) ground-truth frames."""

    def __init__(self, net, scene, B, msteps, std_v, std_re, dt=1.0, res=None, beta1=0.9, beta2=0.999, eps=1e-8, conv_precision="split",
                 use_graph=False, group=None, comm=None, schedule="manual", glue="fused", **solver):
        """schedule: "manual" (default) = the hand-written forward unroll + reverse sweep over the C ABI (_unrolled_schedule), "autograd" = the
        torch-autograd composition of the differentiable HIP ops (rounds 3-4; kept as the cross-check).
        use_graph: capture the whole forward unroll + reverse sweep ONCE into a hipGraph over static input buffers (the
        TF1 "build the graph, sess.run many" shape, as trainer.GraphTrainer does for the 2-D mercury model): a step then
        copies the batch in and replays ~3000 launches with one host call."""
        from .trainer import _conv_precision_code
        from .dist import DPStep
        _lib.require_gpu()
        self.lib = _lib.load()
        self.net, self.scene, self.B, self.ms = net, scene, B, msteps
        assert glue in ("fused", "torch")
        self.glue = glue          # reverse-sweep glue of the manual schedule: sol_karman3d_correct_bwd / _feature_bwd, or the torch elementwise composition
        self.sim = Karman3DFlow(scene, B, dt=dt, res=res, **solver)
        dev = scene.active.device
        self.std_v = torch.tensor([float(v) for v in std_v], dtype=torch.float32, device=dev)
        self._std_v_host = tuple(float(v) for v in std_v)
        self._std_in_host = tuple(float(v) for v in std_v) + (float(std_re),)
        self.std_in = torch.tensor([float(v) for v in std_v] + [float(std_re)], dtype=torch.float32, device=dev)
        self.conv_precision = _conv_precision_code(conv_precision)
        net.params.requires_grad_(True)
        self.m = torch.zeros_like(net.params.detach())
        self.v = torch.zeros_like(net.params.detach())
        self.t = 0
        self.beta1, self.beta2, self.eps = beta1, beta2, eps
        Y, X, Z = scene.Y, scene.X, scene.Z
        f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        # static buffers: inputs, ground-truth frames, outputs (the graph bakes their addresses in)
        self._in = [f(B, Y, X, Z), f(B, Y + 1, X, Z), f(B, Y, X + 1, Z), f(B, Y, X, Z + 1), f(B)]
        self._gt = [f(msteps, B, Y + 1, X, Z), f(msteps, B, Y, X + 1, Z), f(msteps, B, Y, X, Z + 1)]
        self.loss_steps = f(msteps)
        self._loss = f(())
        # gradient + one slot for the loss: the data-parallel exchange (B simulations per rank, BASELINE configs[4]: 8 GPUs) is ONE
        # all-reduce(SUM) of this buffer per training step, as for the 2-D trainers (dist.DPStep; group / comm as there)
        self._flat = f(net.n_params + 1)
        self._grads = self._flat[:net.n_params]
        self._dp = DPStep(lambda *b: (self.fwd_bwd(*b), self._grads), lambda g, lr: self.apply_gradients(lr), group=group, comm=comm, flat=self._flat)
        self._fin = [f(B, Y, X, Z), f(B, Y + 1, X, Z), f(B, Y, X + 1, Z), f(B, Y, X, Z + 1)]
        self.final = None
        self.use_graph, self._graph = bool(use_graph), None
        if schedule not in ("manual", "autograd"):
            raise ValueError("schedule must be 'manual' or 'autograd'")
        self.schedule = schedule

    def _unrolled(self):
        if self.schedule == "manual":
            with torch.no_grad():                   # nothing here is differentiated by torch: the reverse sweep is written out
                return self._unrolled_schedule()
        return self._unrolled_autograd()

    def _unrolled_schedule(self):
        """The SOL-n step as a HAND-WRITTEN schedule over the C ABI (karman_train.py:397-457 differentiated by hand, as train.hip does for
        the 2-D scene): forward unroll with everything the reverse sweep needs kept per step, then the reverse sweep -- no autograd
        graph, no torch op between the launches except a handful of elementwise kernels (slicing adds, one stack per step).
        Per unrolled step i, forward:   solver step (saves the post-diffusion velocity, writes the SCALED features itself) -> network
        (twelve launches, absmax handed from layer to layer) -> velocity += std * to_staggered(out) (sol_karman3d_correct) -> loss_i and
        d loss_i / d v_i in one pass (sol_l2_loss_fwd_bwd, gradient pre-scaled by 1 / msteps).
        Reverse, i = n-1 .. 0:   G = d loss_i / d v_i + (adjoint of step i+1 w.r.t. its input)  ->  d out = std * G at the corrected faces
        -> network reverse sweep (weight gradients accumulated over the steps, d features) -> G += d features / std_in at the low faces
        (to_feature's adjoint) -> solver adjoint (sol_karman3d_step_bwd)."""
        from .ops import l2_loss_fwd_bwd
        d, vy, vx, vz, re = self._in
        net, sim, ms = self.net, self.sim, self.ms
        sc = self.scene
        B, Y, X, Z = self.B, sc.Y, sc.X, sc.Z
        net._tpacks = None                          # the weights moved since the last step: re-pack once (inside the graph when captured)
        fs = [1.0 / t for t in self._std_in_host]          # (host copies: a .tolist() of a device tensor is a synchronising copy -- illegal inside a capture)
        sv = self._std_v_host
        v = (vy, vx, vz)
        keep = []                                   # per step: (saved velocities, network state, d loss_i / d v_i)
        losses = []
        for i in range(ms):
            saved = [torch.empty_like(t) for t in v]
            feat = torch.empty(B, Y, X, Z, 4, dtype=torch.float32, device=vy.device)
            d, *v = sim._fwd(d, v[0], v[1], v[2], re, saved, feat, fs)
            out, xk, amax, acts = _MarsMoon3DFn.run_forward(net, feat)
            check(self.lib.sol_karman3d_correct(stream(), ptr(out), net.cout, sv[0], sv[1], sv[2], ptr(v[0]), ptr(v[1]), ptr(v[2]), B, Y, X, Z))
            li, gi = l2_loss_fwd_bwd(v, tuple(g[i] for g in self._gt), sv, gscale=1.0 / ms)
            losses.append(li.reshape(()))
            keep.append((saved, xk, amax, acts, gi))
        flat = None
        gin = None
        wstate = [dict() for _ in range(12)]         # per layer: the weight-gradient partials of the whole reverse sweep (reduced once, at i = 0)
        inv_in = fs[:3]
        for i in range(ms - 1, -1, -1):
            saved, xk, amax, acts, G = keep[i]
            # G += gin and the adjoint of  v += std * to_staggered(out):  d out[..., c] = std_c * G_c restricted to the faces that received a correction
            # (one launch, zero padded to the four channels the thin-layer launches read; glue="torch": the elementwise composition of rounds 5-6)
            fused_glue = self.glue == "fused" and net.cout == 3
            if fused_glue:
                dO = torch.empty(B, Y, X, Z, 4, dtype=torch.float32, device=G[0].device)
                gi = gin if gin is not None else (None, None, None)
                check(self.lib.sol_karman3d_correct_bwd(stream(), ptr(G[0]), ptr(G[1]), ptr(G[2]), ptr(gi[0]), ptr(gi[1]), ptr(gi[2]), sv[0], sv[1], sv[2], ptr(dO), B, Y, X, Z))
            else:
                if gin is not None:
                    for c in range(3):
                        G[c].add_(gin[c])
                dO = torch.stack([G[0][:, :Y] * sv[0], G[1][:, :, :X] * sv[1], G[2][..., :Z] * sv[2]], dim=-1)
            dx, gflat = _MarsMoon3DFn.run_backward(net, xk, amax, acts, dO, acc=(wstate, i == ms - 1, i == 0))
            flat = gflat if gflat is not None else flat
            # adjoint of the feature map (the three components at the low faces of every cell, divided by std_in; the Re channel has no gradient)
            if fused_glue and dx.shape[-1] == 4:
                check(self.lib.sol_karman3d_feature_bwd(stream(), ptr(dx), inv_in[0], inv_in[1], inv_in[2], ptr(G[0]), ptr(G[1]), ptr(G[2]), B, Y, X, Z))
            else:
                G[0][:, :Y].add_(dx[..., 0], alpha=inv_in[0])
                G[1][:, :, :X].add_(dx[..., 1], alpha=inv_in[1])
                G[2][..., :Z].add_(dx[..., 2], alpha=inv_in[2])
            gin = sim._bwd(saved, re, G[0], G[1], G[2])
            keep[i] = None                          # (eager runs: the step's activations can go)
        losses = _lib.stack0(losses)
        loss = losses.sum() / ms
        # (kernel copies: a contiguous tensor.copy_ is a hipMemcpyAsync = a memcpy node, refused by the capture guard -- _lib.dcopy_)
        _lib.dcopy_(self.loss_steps, losses)
        _lib.dcopy_(self._loss, loss)
        _lib.dcopy_(self._grads, flat)
        for dst, src in zip(self._fin, (d,) + tuple(v)):
            _lib.dcopy_(dst, src)

    def _unrolled_autograd(self):
        """The same step as a torch-autograd composition of the differentiable HIP ops (schedule="autograd"): the cross-check of the
        hand-written schedule (test_karman3d_manual_schedule_equals_autograd_composition) and the form of rounds 3-4."""
        d, vy, vx, vz, re = self._in
        self.net.params.grad = None
        self.net._tpacks = None                     # the weights moved since the last step: re-pack once (inside the graph when captured)
        v = (_lib.dclone(vy).requires_grad_(True), vx, vz)     # the state enters the graph (the step's autograd Function needs a grad-requiring input)
        losses = []
        for i in range(self.ms):
            d, *v = self.sim.step(d, v[0], v[1], v[2], re)
            out = self.net(to_feature3d(v[0], v[1], v[2], re) / self.std_in) * self.std_v
            v = tuple(a + c for a, c in zip(v, to_staggered3d(out)))
            # (one kernel, no torch reduction: a multi-workgroup torch .sum() puts a memset node into the captured graph, ops.L2LossFn)
            losses.append(ops_l2_loss(v, tuple(g[i] for g in self._gt), self._std_v_host))
        losses = _lib.stack0(losses)
        loss = losses.sum() / self.ms
        loss.backward()
        # (kernel copies: a contiguous tensor.copy_ is a hipMemcpyAsync = a memcpy node, refused by the capture guard -- _lib.dcopy_)
        _lib.dcopy_(self.loss_steps, losses)
        _lib.dcopy_(self._loss, loss)
        _lib.dcopy_(self._grads, self.net.params.grad)
        for dst, src in zip(self._fin, (d,) + tuple(v)):
            _lib.dcopy_(dst, src)

    def fwd_bwd(self, d, vy, vx, vz, re, gts):
        """gts: [msteps] of (vy, vx, vz) frames, or the three stacked tensors [msteps, B, ...]."""
        from .trainer import _conv_precision_scope
        for dst, src in zip(self._in, (d, vy, vx, vz, re)):
            dst.copy_(_lib.f32(src), non_blocking=True)
        if isinstance(gts, (list, tuple)) and len(gts) == self.ms and isinstance(gts[0], (list, tuple)):
            for i, fr in enumerate(gts):
                for c in range(3):
                    self._gt[c][i].copy_(_lib.f32(fr[c]), non_blocking=True)
        else:
            for c in range(3):
                self._gt[c].copy_(_lib.f32(gts[c]), non_blocking=True)
        with _conv_precision_scope(self.conv_precision):
            if not self.use_graph:
                self._unrolled()
            else:
                if self._graph is None:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):          # warm-up off the capture: library initialisation, allocator pools
                        self._unrolled()
                    torch.cuda.current_stream().wait_stream(side)
                    torch.cuda.synchronize()
                    self.net.params.grad = None
                    self._graph = _lib.capture_graph(self._unrolled, "Karman3DTrainer")      # kernel nodes only (sol_graph_check), then instantiated
                self._graph.replay()
        self.final = tuple(self._fin)
        return self._loss

    @property
    def grads(self):
        return self._grads

    def apply_gradients(self, lr):
        self.t += 1
        p = self.net.params.detach()
        check(self.lib.sol_adam_tf_step(stream(), ptr(p), ptr(self._grads), ptr(self.m), ptr(self.v),
                                        self.net.n_params, self.t, float(lr), self.beta1, self.beta2, self.eps, 0.0, None, 0, None))
        self.net.invalidate()                       # both pack caches are stale now (the forward packs AND the training packs)

    def train_step(self, d, vy, vx, vz, re, gts, lr):
        """One training step on this rank's simulations; returns the GLOBAL loss tensor (the sum over all ranks' simulations)."""
        return self._dp(d, vy, vx, vz, re, gts, lr=lr)


class Karman3DRollout:
    """No-grad roll-out: nsteps x [solver step -> features / std -> CNN -> velocity += std * correction]
    (karman_apply.py:138-158 / the forward half of karman_train.py:397-426, three components)."""

    def __init__(self, net, scene, B, std_v, std_re, dt=1.0, res=None, conv_precision="split", **solver):
        from .trainer import _conv_precision_code
        _lib.require_gpu()
        self.lib = _lib.load()
        self.net, self.scene, self.B = net, scene, B
        self.sim = Karman3DFlow(scene, B, dt=dt, res=res, **solver)
        self.std_v = tuple(float(v) for v in std_v)
        self.feat_scale = [1.0 / self.std_v[0], 1.0 / self.std_v[1], 1.0 / self.std_v[2], 1.0 / float(std_re)]
        self.conv_precision = _conv_precision_code(conv_precision)
        dev = scene.active.device
        Y, X, Z = scene.Y, scene.X, scene.Z
        f = lambda c: torch.empty(B, Y, X, Z, c, dtype=torch.float32, device=dev)
        self.feat = f(4)
        self.h = [f(32), f(32), f(32)]          # block input / intermediate / block output, rotated
        self.out = f(net.cout)
        self.amax = torch.zeros(12, 256, dtype=torch.int32, device=dev)      # absmax slots of the 32-channel activations

    def correction(self):
        """self.feat -> self.out (the network), publishing / consuming the per-tensor absmax of every 32-channel tensor."""
        from .trainer import _conv_precision_scope
        with _conv_precision_scope(self.conv_precision):
            return self._correction()

    def _correction(self):
        pk = self.net.pack()
        sl = self.net.slope
        self.amax.zero_()
        am = lambda k: self.amax[k]
        h, a, n = self.h
        kp = self.net.pack_thin()
        if kp is not None:
            if getattr(self, "_thin_ws", None) is None:
                self._thin_ws = torch.empty(self.lib.sol_conv3d_thin_ws_floats(*self.feat.shape[:4]), dtype=torch.float32, device=self.feat.device)
            conv3d_thin(self.feat, kp, pk[0][1], True, sl, am(0), out=h, ws=self._thin_ws)
        else:
            conv3d(self.feat, pk[0][0], pk[0][1], None, 32, True, sl, None, am(0), out=h)
        for k in range(5):
            conv3d(h, pk[1 + 2 * k][0], pk[1 + 2 * k][1], None, 32, True, sl, am(2 * k), am(2 * k + 1), out=a)
            conv3d(a, pk[2 + 2 * k][0], pk[2 + 2 * k][1], h, 32, True, sl, am(2 * k + 1), am(2 * k + 2), out=n)
            h, n = n, h
        ko = self.net.pack_thin_out()
        if ko is not None:
            if getattr(self, "_thin_ws", None) is None:
                self._thin_ws = torch.empty(self.lib.sol_conv3d_thin_ws_floats(*self.feat.shape[:4]), dtype=torch.float32, device=self.feat.device)
            conv3d_thin_out(h, ko, pk[11][1], self.net.cout, am(10), out=self.out, ws=self._thin_ws)
        else:
            conv3d(h, pk[11][0], pk[11][1], None, self.net.cout, False, sl, am(10), None, out=self.out)
        return self.out

    def step(self, d, vy, vx, vz, re):
        d, vy, vx, vz = self.sim.step(d, vy, vx, vz, re, feat_out=self.feat, feat_scale=self.feat_scale)
        out = self.correction()
        s = self.scene
        check(self.lib.sol_karman3d_correct(stream(), ptr(out), self.net.cout, self.std_v[0], self.std_v[1], self.std_v[2],
                                            ptr(vy), ptr(vx), ptr(vz), self.B, s.Y, s.X, s.Z))
        return d, vy, vx, vz

    def run(self, d, vy, vx, vz, re, nsteps):
        re = _lib.f32(re)
        for _ in range(nsteps):
            d, vy, vx, vz = self.step(d, vy, vx, vz, re)
        return d, vy, vx, vz
