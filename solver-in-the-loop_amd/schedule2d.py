"""Hand-written forward pass and reverse sweep of the 2-D correction networks over the C ABI.

`model_mars_moon` (/root/reference/karman-2d/karman_train.py:101-138) and `model_mercury` (:92-99) differentiated by hand, layer by
layer, exactly as csrc/train.hip does it for the fused C++ trainer and karman3d._MarsMoon3DFn for the 3-D network -- for the
trainers that are composed in Python (trainer.GraphTrainer: model_mercury and any network on the karman scene, burgers.BurgersTrainer):

  * the weights are packed ONCE per training step (forward and backward-data form of every layer), not once per convolution call;
  * a backward-data launch applies the skip gradient and LeakyReLU'(saved activation) in its epilogue (SOL_EPI_DLRELU): no
    compare / where / multiply / add kernels between the launches;
  * every layer's weight gradient is ACCUMULATED over the unrolled steps in that layer's partial buffer (sol_conv5x5_bwd_weight adds
    into `partial`) and reduced once per step (sol_conv5x5_bwd_weight_reduce);
  * on 64-pixel rows the 32-channel layers run the fp16 three-product kernels, each producer publishing the absmax slots its
    consumer scales by (sol_conv5x5_scaled) -- no pass over a tensor just to find its maximum.

No autograd graph is built; the torch operations left between the launches are allocations, zero fills and concatenations (kernels).
The autograd compositions (model.MarsMoon.__call__, model.Mercury.__call__ over ops.Conv5x5Fn) stay as the cross-check.
"""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import check, ptr, stream


class _Unit:
    """one convolution of the network: its kernel w [5,5,cin,cout] and bias (views of the flat parameter buffer, or persistent copies of
    the halves model_mercury splits a kernel into), its packed forms and its slice of the weight-gradient partial buffer"""
    __slots__ = ("cin", "cout", "cin_k", "w", "b", "pf", "pb", "part", "ws", "dw", "db", "src")


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _int_array(vals):
    return (C.c_int32 * len(vals))(*[int(v) for v in vals])


class NetSchedule2D:
    def __init__(self, net, B, H, W):
        if net.name not in ("mars_moon", "mercury"):
            raise _lib.SolError("NetSchedule2D: no hand-written schedule for network %r" % net.name)
        _lib.require_gpu()
        self.lib = lib = _lib.load()
        self.net, self.B, self.H, self.W = net, int(B), int(H), int(W)
        self.scaled = self.W % 64 == 0               # 64-pixel rows: the split-precision kernels + absmax hand-over
        if net.cin > 4:
            raise _lib.SolError("NetSchedule2D: at most 4 input channels (got %d)" % net.cin)
        if net.cout != 2:
            raise _lib.SolError("NetSchedule2D: the weight-gradient kernels take 2 or 32 output channels (network output: %d)" % net.cout)
        mm = net.name == "mars_moon"
        dims = [(net.cin, 32)] + [(32, 32)] * 10 + [(32, net.cout)] if mm else [(net.cin, 32), (32, 32), (32, 32), (32, net.cout), (32, net.cout)]
        dev = net.params.device
        f = lambda n: torch.empty(int(n), dtype=torch.float32, device=dev)
        self.units = []
        total = 0
        for cin, cout in dims:
            u = _Unit()
            u.cin, u.cout, u.cin_k = cin, cout, (4 if cin <= 4 else 32)
            u.ws = int(lib.sol_conv5x5_bwd_weight_ws_floats(self.B, self.H, self.W, u.cin_k, cout))
            total += (u.ws + 3) // 4 * 4
            u.pf = f(lib.sol_conv5x5_packed_floats(cin, cout, ops.CONV_FWD))
            u.pb = f(lib.sol_conv5x5_packed_floats(cout, cin, ops.CONV_BWD_DATA))
            self.units.append(u)
        self._partials = f(total)
        off = 0
        for u in self.units:
            u.part = self._partials[off:off + u.ws]
            off += (u.ws + 3) // 4 * 4
        self._zero_bias = torch.zeros(net.cout, dtype=torch.float32, device=dev)
        # kernels / biases: views of the flat parameter buffer (updated in place by Adam: the addresses are stable), or persistent copies of the
        # halves of model_mercury's 32 -> 64 / 64 -> 2 kernels (refreshed by begin_step)
        p = [t.detach() for t in net.tensors()]
        self.flat = f(net.n_params)                   # the step's gradient, Keras get_weights() order
        g = [t for t in self.net.tensors(self.flat)]
        if mm:
            for l, u in enumerate(self.units):
                u.w, u.b, u.src = p[2 * l], p[2 * l + 1], None
                u.dw, u.db = g[2 * l], g[2 * l + 1]    # the reduction writes straight into the flat gradient
        else:
            U = self.units
            U[0].w, U[0].b, U[0].src = p[0], p[1], None
            U[0].dw, U[0].db = g[0], g[1]
            for u, src, b in ((U[1], p[2][..., :32], p[3][:32]), (U[2], p[2][..., 32:], p[3][32:]), (U[3], p[4][:, :, :32], p[5]), (U[4], p[4][:, :, 32:], self._zero_bias)):
                u.src, u.w, u.b = src, torch.empty(src.shape, dtype=torch.float32, device=dev), b
                u.dw, u.db = torch.empty(src.shape, dtype=torch.float32, device=dev), f(u.cout)
        n = len(self.units)
        self._pack_args = (2 * n, _ptr_array([u.w for u in self.units] * 2), _int_array([u.cin for u in self.units] + [u.cout for u in self.units]),
                           _int_array([u.cout for u in self.units] + [u.cin for u in self.units]), _int_array([ops.CONV_FWD] * n + [ops.CONV_BWD_DATA] * n),
                           _ptr_array([u.pf for u in self.units] + [u.pb for u in self.units]))
        self._reduce_args = (n, _ptr_array([u.part for u in self.units]), _ptr_array([u.dw for u in self.units]), _ptr_array([u.db for u in self.units]),
                             self.B, self.H, self.W, _int_array([u.cin for u in self.units]), _int_array([u.cout for u in self.units]), 0)

    # ---- once per training step -------------------------------------------------------------------------------------------
    def begin_step(self):
        """pack every layer's weights (forward + backward-data form) from the CURRENT parameters -- ONE launch (sol_conv5x5_pack_jobs) -- and
        clear the weight-gradient partials (one launch)"""
        for u in self.units:
            if u.src is not None:
                u.w.copy_(u.src)                      # (strided halves of a mercury kernel: an elementwise copy kernel)
        check(self.lib.sol_conv5x5_pack_jobs(stream(), *self._pack_args))
        check(self.lib.sol_copy_words(stream(), ptr(self._partials), None, self._partials.numel()))

    # ---- launches -----------------------------------------------------------------------------------------------------------
    def _conv(self, x, packed, bias, residual, act_ref, cout, epi, xmax, ymax):
        B, H, W, cin = x.shape
        y = torch.empty(B, H, W, cout, dtype=torch.float32, device=x.device)
        sl = float(self.net.slope)
        if self.scaled:
            check(self.lib.sol_conv5x5_scaled(stream(), ptr(x), ptr(packed), ptr(bias), ptr(residual), ptr(act_ref), ptr(y),
                                              B, H, W, cin, cout, epi, sl, ptr(xmax), ptr(ymax)))
        else:
            check(self.lib.sol_conv5x5(stream(), ptr(x), ptr(packed), ptr(bias), ptr(residual), ptr(act_ref), ptr(y),
                                       B, H, W, cin, cout, epi, sl))
        return y

    def _bww(self, u, xk, dz):
        """u.part += the weight-gradient partial sums of this step"""
        check(self.lib.sol_conv5x5_bwd_weight(stream(), ptr(xk), ptr(dz), ptr(u.part), self.B, self.H, self.W, u.cin_k, u.cout))

    def _slots(self, n, dev):
        return torch.zeros(n, ops.AMAX_SLOTS, dtype=torch.int32, device=dev) if self.scaled else [None] * n

    # ---- forward ------------------------------------------------------------------------------------------------------------
    def forward(self, x):
        """x [B,H,W,cin] -> (out [B,H,W,cout], state for backward)"""
        xk = ops._pad_channels(_lib.f32(x.detach()), 4)
        U, L, N = self.units, ops.EPI_LRELU, ops.EPI_NONE
        if self.net.name == "mars_moon":
            am = self._slots(11, xk.device)
            acts = [self._conv(xk, U[0].pf, U[0].b, None, None, 32, L, None, am[0])]
            for k in range(5):
                a = self._conv(acts[-1], U[1 + 2 * k].pf, U[1 + 2 * k].b, None, None, 32, L, am[2 * k], am[2 * k + 1])
                acts.append(a)
                acts.append(self._conv(a, U[2 + 2 * k].pf, U[2 + 2 * k].b, acts[-2], None, 32, L, am[2 * k + 1], am[2 * k + 2]))
            out = self._conv(acts[-1], U[11].pf, U[11].b, None, None, self.net.cout, N, am[10], None)
        else:
            am = self._slots(3, xk.device)
            h = self._conv(xk, U[0].pf, U[0].b, None, None, 32, L, None, am[0])
            ha = self._conv(h, U[1].pf, U[1].b, None, None, 32, L, am[0], am[1])
            hb = self._conv(h, U[2].pf, U[2].b, None, None, 32, L, am[0], am[2])
            oa = self._conv(ha, U[3].pf, U[3].b, None, None, self.net.cout, N, am[1], None)
            out = self._conv(hb, U[4].pf, U[4].b, oa, None, self.net.cout, N, am[2], None)
            acts = [h, ha, hb]
        return out, (xk, am, acts)

    # ---- reverse sweep ------------------------------------------------------------------------------------------------------
    def backward(self, state, g_out):
        """d loss / d x [B,H,W,cin] for the output gradient g_out [B,H,W,cout]; the weight gradients of this call are added to the layers'
        partial buffers (end_step reduces them)."""
        xk, am, acts = state
        U, D, N = self.units, ops.EPI_DLRELU, ops.EPI_NONE
        g = _lib.f32(g_out).contiguous()
        g4 = ops._pad_channels(g, 4)
        if self.net.name == "mars_moon":
            zm = self._slots(11, xk.device)
            self._bww(U[11], acts[10], g)
            dz = self._conv(g4, U[11].pb, None, None, acts[10], 32, D, None, zm[10])
            for k in range(4, -1, -1):
                a, hprev = acts[1 + 2 * k], acts[2 * k]
                self._bww(U[2 + 2 * k], a, dz)
                dz1 = self._conv(dz, U[2 + 2 * k].pb, None, None, a, 32, D, zm[2 * k + 2], zm[2 * k + 1])
                self._bww(U[1 + 2 * k], hprev, dz1)
                dz = self._conv(dz1, U[1 + 2 * k].pb, None, dz, hprev, 32, D, zm[2 * k + 1], zm[2 * k])
            self._bww(U[0], xk, dz)
            return self._conv(dz, U[0].pb, None, None, None, U[0].cin, N, zm[0], None)
        h, ha, hb = acts
        zm = self._slots(3, xk.device)
        self._bww(U[3], ha, g)
        self._bww(U[4], hb, g)
        dha = self._conv(g4, U[3].pb, None, None, ha, 32, D, None, zm[1])
        dhb = self._conv(g4, U[4].pb, None, None, hb, 32, D, None, zm[2])
        self._bww(U[1], h, dha)
        self._bww(U[2], h, dhb)
        t = self._conv(dha, U[1].pb, None, None, None, 32, N, zm[1], None)
        dh = self._conv(dhb, U[2].pb, None, t, h, 32, D, zm[2], zm[0])
        self._bww(U[0], xk, dh)
        return self._conv(dh, U[0].pb, None, None, None, U[0].cin, N, zm[0], None)

    # ---- once per training step ---------------------------------------------------------------------------------------------
    def end_step(self):
        """reduce every layer's partial sums (sol_conv5x5_bwd_weight_reduce_jobs: two launches for all layers); returns the flat gradient in
        Keras get_weights() order (self.flat: model_mars_moon's layers reduce straight into it)"""
        check(self.lib.sol_conv5x5_bwd_weight_reduce_jobs(stream(), *self._reduce_args))
        if self.net.name == "mercury":
            # kernel 2 is [5,5,32,64] = the two output halves side by side, kernel 4 is [5,5,64,2] = the two input halves stacked;
            # the output bias receives its gradient once (both last-layer halves see the same dz)
            U = self.units
            g = self.net.tensors(self.flat)
            g[2][..., :32].copy_(U[1].dw)
            g[2][..., 32:].copy_(U[2].dw)
            _lib.dcopy_(g[3][:32], U[1].db)          # (contiguous pieces: kernel copies -- a contiguous copy_ is a memcpy node, refused under capture)
            _lib.dcopy_(g[3][32:], U[2].db)
            g[4][:, :, :32].copy_(U[3].dw)
            g[4][:, :, 32:].copy_(U[4].dw)
            _lib.dcopy_(g[5], U[3].db)
        return self.flat
