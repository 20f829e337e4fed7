"""Correction networks of the reference on the fp32 MFMA conv kernels.

model_mars_moon / model_mercury: /root/reference/karman-2d/karman_train.py:92-138.
Parameters live in ONE flat fp32 CUDA buffer in Keras get_weights() order
([kernel0 HWIO, bias0, kernel1, bias1, ...]); that buffer is what the fused trainer, Adam and
the RCCL all-reduce operate on.  Keras defaults are kept: glorot_uniform kernels, zero
biases, LeakyReLU(alpha=0.3).
"""
import math

import numpy as np
import torch

from . import ops


def _glorot(shape, gen):
    fan_in = shape[0] * shape[1] * shape[2]
    fan_out = shape[0] * shape[1] * shape[3]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim


class ConvNet:
    """Sequential description: layers = [(cin, cout, act, skip_from)]"""

    name = "net"
    layer_channels = ()
    slope = 0.3

    def __init__(self, cin=3, cout=2, seed=0, device="cuda"):
        self.cin, self.cout = cin, cout
        self.shapes = []
        chans = self.channels(cin, cout)
        for l in range(len(chans) - 1):
            self.shapes.append((5, 5, chans[l], chans[l + 1]))
            self.shapes.append((chans[l + 1],))
        self.offsets = np.concatenate([[0], np.cumsum([int(np.prod(s)) for s in self.shapes])]).astype(np.int64)
        gen = torch.Generator().manual_seed(seed)
        flat = torch.cat([(_glorot(s, gen) if len(s) == 4 else torch.zeros(s, dtype=torch.float64)).reshape(-1)
                          for s in self.shapes])
        self.params = flat.to(device=device, dtype=torch.float32).contiguous().requires_grad_(True)
        self.losses = []   # no regularisers (karman_train.py:439-444)

    @property
    def n_params(self):
        return int(self.offsets[-1])

    def tensors(self, flat=None):
        flat = self.params if flat is None else flat
        # (ops.split_flat: ONE autograd node whose backward assembles the flat gradient with kernel copies -- plain slices cost a
        #  memcpy node, a zero fill and an add of the whole buffer per tensor and unrolled step in a captured trainer)
        return [t.reshape(s) for t, s in zip(ops.split_flat(flat, self.offsets), self.shapes)]

    # Keras-style API used by the reference scripts
    def get_weights(self):
        return [t.detach().cpu().numpy() for t in self.tensors()]

    def set_weights(self, weights):
        flat = np.concatenate([np.asarray(w, dtype=np.float32).reshape(-1) for w in weights])
        assert flat.size == self.n_params, "weight list does not match %s" % self.name
        with torch.no_grad():
            self.params.copy_(torch.as_tensor(flat, device=self.params.device))

    def clone(self):
        """Independent copy (own parameter buffer) on the same device."""
        other = type(self)(self.cin, self.cout, 0, self.params.device)
        with torch.no_grad():
            other.params.copy_(self.params)
        return other

    def save(self, path):
        torch.save({"name": self.name, "cin": self.cin, "cout": self.cout,
                    "weights": [torch.as_tensor(w) for w in self.get_weights()]}, path)

    @classmethod
    def load(cls, path, device="cuda"):
        """`.pt` written by save(), or `.npz` holding the arrays of Keras `model.get_weights()` in order (arr_0, arr_1, ...:
        the bridge from the reference's model.h5 -- h5py is not part of this image -- is `np.savez(path, *model.get_weights())`
        on the TensorFlow side).  The architecture is recognised from the kernel shapes."""
        if str(path).endswith(".npz"):
            z = np.load(path)
            ws = [z[k] for k in sorted(z.files, key=lambda k: int(k.split("_")[-1]))]
            cin, cout = ws[0].shape[2], ws[-2].shape[3]
            name = "mercury" if len(ws) == 6 else "mars_moon"
            net = MODELS[name](cin, cout, device=device)
            net.set_weights(ws)
            return net
        blob = torch.load(path, map_location="cpu")
        net = MODELS[blob["name"]](blob["cin"], blob["cout"], device=device)
        net.set_weights([w.numpy() for w in blob["weights"]])
        return net

    def summary(self, print_fn=print):
        print_fn("Model %s: %d parameters" % (self.name, self.n_params))
        for k in range(0, len(self.shapes), 2):
            print_fn("  conv5x5 %-18s bias %s" % (self.shapes[k], self.shapes[k + 1]))

    def predict(self, x):
        with torch.no_grad():
            return self(x)


class MarsMoon(ConvNet):
    """model_mars_moon, karman_train.py:101-138: 12 convs, 260,354 parameters."""
    name = "mars_moon"

    @staticmethod
    def channels(cin, cout):
        return [cin] + [32] * 11 + [cout]

    def __call__(self, x, flat=None):
        p = self.tensors(flat)
        s = self.slope
        h = ops.conv5x5(x, p[0], p[1], None, True, s)
        for k in range(5):
            a = ops.conv5x5(h, p[2 + 4 * k], p[3 + 4 * k], None, True, s)
            h = ops.conv5x5(a, p[4 + 4 * k], p[5 + 4 * k], h, True, s)
        return ops.conv5x5(h, p[22], p[23], None, False, s)


class Mercury(ConvNet):
    """model_mercury, karman_train.py:92-99: Conv5x5(cin->32)+ReLU, Conv5x5(32->64)+ReLU, Conv5x5(64->2).
    The conv kernels are built for <= 32 channels per side, so the 64-channel layer runs as two 32-channel
    halves (output split for 32->64, input split + residual accumulation for 64->2).  Training: trainer.GraphTrainer."""
    name = "mercury"
    slope = 0.0                                      # ReLU = LeakyReLU with slope 0

    @staticmethod
    def channels(cin, cout):
        return [cin, 32, 64, cout]

    def __call__(self, x, flat=None):
        p = self.tensors(flat)
        h = ops.conv5x5(x, p[0], p[1], None, True, 0.0)
        b3a, b3b = ops.split_flat(p[3], (0, 32, 64))           # (1-D halves: not plain slices, see ConvNet.tensors)
        ha = ops.conv5x5(h, p[2][..., :32].contiguous(), b3a, None, True, 0.0)
        hb = ops.conv5x5(h, p[2][..., 32:].contiguous(), b3b, None, True, 0.0)
        oa = ops.conv5x5(ha, p[4][:, :, :32].contiguous(), p[5], None, False, 0.0)
        return ops.conv5x5(hb, p[4][:, :, 32:].contiguous(), torch.zeros_like(p[5]), oa, False, 0.0)


def model_mercury(tensor_in=None, cin=3, cout=2, seed=0, device="cuda"):
    if tensor_in is not None:
        cin = tensor_in.shape[-1]
    return Mercury(cin, cout, seed, device)


def model_mars_moon(tensor_in=None, cin=3, cout=2, seed=0, device="cuda"):
    if tensor_in is not None:
        cin = tensor_in.shape[-1]
    return MarsMoon(cin, cout, seed, device)


MODELS = {"mars_moon": MarsMoon, "mercury": Mercury}
