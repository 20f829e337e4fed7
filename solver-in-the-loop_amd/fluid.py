"""Minimal mirror of the PhiFlow-1.x objects the reference scripts touch on the hot path.

Only what karman_train.py / karman_apply.py / burgers_train.py use:
`box[...]`, `Domain`, `OPEN` / `PERIODIC`, `CenteredGrid`, `StaggeredGrid`
(`.data[c].data`, `.staggered_tensor()`, `+`, `.box`), `Fluid` / `BurgersVelocitySMAC`
(`.density`, `.velocity`, `._batch_size`, `.copied_with`), geometry `Sphere`, `Inflow`,
`Obstacle`.  Tensors are torch CUDA fp32 in the reference's layouts:
density [B,Y,X,1], v_y [B,Y+1,X,1], v_x [B,Y,X+1,1], staggered tensor [B,Y+1,X+1,2]
(component 0 = y, /root/reference/karman-2d/karman_train.py:367).
"""
import numpy as np
import torch

from . import _lib

OPEN = "open"
PERIODIC = "periodic"


class Box:
    def __init__(self, lower, upper):
        self.lower = tuple(float(v) for v in lower)
        self.upper = tuple(float(v) for v in upper)

    @property
    def size(self):
        return tuple(u - l for l, u in zip(self.lower, self.upper))

    def value_at(self, yc, xc):
        """1 where the point lies inside (inclusive bounds, PhiFlow Box.value_at)."""
        return ((yc >= self.lower[0]) & (yc <= self.upper[0]) &
                (xc >= self.lower[1]) & (xc <= self.upper[1])).astype(np.float64)

    def __repr__(self):
        return "Box(%s, %s)" % (self.lower, self.upper)


class _BoxFactory:
    """`box[5:10, 25:75]` and `box([32, 32])` as in `from phi.flow import box`."""

    def __getitem__(self, item):
        if isinstance(item, slice):
            item = (item,)
        return Box([s.start or 0 for s in item], [s.stop for s in item])

    def __call__(self, size):
        return Box([0] * len(size), size)


box = _BoxFactory()


class Sphere:
    def __init__(self, center, radius):
        self.center = tuple(float(c) for c in center)
        self.radius = float(radius)

    def value_at(self, yc, xc):
        return (((yc - self.center[0]) ** 2 + (xc - self.center[1]) ** 2) <= self.radius ** 2).astype(np.float64)


class Inflow:
    def __init__(self, geometry, rate=1.0):
        self.geometry = geometry
        self.rate = float(rate)


class Obstacle:
    def __init__(self, geometry):
        self.geometry = geometry


class Gravity:
    def __init__(self, gravity=-9.81):
        self.gravity = gravity


class Domain:
    def __init__(self, resolution, box=None, boundaries=OPEN):
        self.resolution = tuple(int(r) for r in resolution)
        self.box = box if box is not None else Box([0] * len(self.resolution), self.resolution)
        self.boundaries = boundaries

    @property
    def dx(self):
        return tuple(s / r for s, r in zip(self.box.size, self.resolution))

    def cell_centers(self):
        Y, X = self.resolution
        dy, dx = self.dx
        yc = self.box.lower[0] + (np.arange(Y) + 0.5) * dy
        xc = self.box.lower[1] + (np.arange(X) + 0.5) * dx
        return np.meshgrid(yc, xc, indexing="ij")


class CenteredGrid:
    def __init__(self, data, box=None):
        self.data = data
        self.box = box

    def __mul__(self, other):
        return CenteredGrid(self.data * _raw(other), self.box)

    def __add__(self, other):
        return CenteredGrid(self.data + _raw(other), self.box)


def _raw(v):
    if isinstance(v, CenteredGrid):
        return v.data
    if isinstance(v, np.ndarray):
        return torch.as_tensor(v, dtype=torch.float32, device="cuda")
    return v


def unstack_staggered_tensor(t):
    """[B,Y+1,X+1,2] -> (v_y [B,Y+1,X,1], v_x [B,Y,X+1,1])  (PhiFlow unstack_staggered_tensor)."""
    return t[:, :, :-1, 0:1], t[:, :-1, :, 1:2]


class StaggeredGrid:
    """`StaggeredGrid(tensor, box=)` (to_staggered, karman_train.py:90) or
    `StaggeredGrid([vy, vx], box)` (karman_train.py:183)."""

    def __init__(self, data, box=None):
        if isinstance(data, (list, tuple)):
            comps = [d.data if isinstance(d, CenteredGrid) else d for d in data]
        else:
            comps = list(unstack_staggered_tensor(data))
        self.data = [CenteredGrid(c.contiguous(), box) for c in comps]
        self.box = box

    def staggered_tensor(self):
        vy, vx = self.data[0].data, self.data[1].data
        vy = _lib.pad_high(vy, 2)                                 # pad x at the high end  (cat with zeros: no memcpy node under capture)
        vx = _lib.pad_high(vx, 1)                                 # pad y at the high end
        return torch.cat([vy, vx], dim=-1)

    def __add__(self, other):
        return StaggeredGrid([a.data + b.data for a, b in zip(self.data, other.data)], self.box)

    def __mul__(self, s):
        return StaggeredGrid([a.data * s for a in self.data], self.box)

    __rmul__ = __mul__

    @property
    def resolution(self):
        b, yp1, x, _ = self.data[0].data.shape
        return (yp1 - 1, x)


class Fluid:
    """`Fluid(Domain(...), buoyancy_factor=0, batch_size=B)` (karman_train.py:363)."""

    def __init__(self, domain, density=0.0, velocity=0.0, buoyancy_factor=0.0, batch_size=1, device="cuda"):
        self.domain = domain
        self.buoyancy_factor = buoyancy_factor
        self._batch_size = batch_size
        Y, X = domain.resolution
        self.density = self._centered(density, (batch_size, Y, X, 1), device)
        self.velocity = self._staggered(velocity, batch_size, Y, X, device)
        if buoyancy_factor != 0:
            raise NotImplementedError("buoyancy is unused on the reference path (buoyancy_factor=0)")

    def _centered(self, v, shape, device):
        if isinstance(v, CenteredGrid):
            return v
        if isinstance(v, (int, float)):
            return CenteredGrid(torch.full(shape, float(v), dtype=torch.float32, device=device), self.domain.box)
        t = torch.as_tensor(v, dtype=torch.float32, device=device)
        return CenteredGrid(t.reshape(shape).contiguous(), self.domain.box)

    def _staggered(self, v, B, Y, X, device):
        if isinstance(v, StaggeredGrid):
            return v
        if isinstance(v, (int, float)):
            t = torch.full((B, Y + 1, X + 1, 2), float(v), dtype=torch.float32, device=device)
        else:
            t = torch.as_tensor(v, dtype=torch.float32, device=device)
        return StaggeredGrid(t, self.domain.box)

    def copied_with(self, density=None, velocity=None):
        new = object.__new__(type(self))
        new.__dict__.update(self.__dict__)
        if density is not None:
            Y, X = self.domain.resolution
            new.density = self._centered(density, (self._batch_size, Y, X, 1), self.density.data.device)
        if velocity is not None:
            Y, X = self.domain.resolution
            new.velocity = self._staggered(velocity, self._batch_size, Y, X, self.density.data.device)
        return new


class BurgersVelocitySMAC(Fluid):
    """burgers_train.py:172-176: a state holding only a staggered velocity."""

    def __init__(self, domain, velocity=0.0, batch_size=1, device="cuda"):
        Fluid.__init__(self, domain, 0.0, velocity, 0.0, batch_size, device)
