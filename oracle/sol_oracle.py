"""CPU oracle for the solver-in-the-loop hot path  --  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic of this path lives in PhiFlow 1.5.1 (commit 4f5e678)
and TensorFlow 1.15 (/root/reference/README.md:19-24).  Neither is installable in the
build container and the reference ships no tests / golden vectors,
so this file is a float64 *restatement* of the published PhiFlow-1.x algorithm, anchored
on the reference's own call sites.  It is pinned by analytic known-answer tests
(tests/test_oracle_kat.py) and by the fixtures in tests/golden/ that it generated itself.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (solver-in-the-loop_amd/) never does.

Everything is written with torch (CPU) so that autograd provides the reference
gradients for the hand-written HIP backward kernels.  Default dtype is float64; the
cpu_baseline leg of bench.py runs the same code in float32.

Layout conventions (reference: karman-2d/karman_train.py:367 "velocity.data[0] is
considered as the velocity field in y axis"):
    density  d  [B, Y, X]        cell centres   ((j+.5)dx, (i+.5)dx)
    v_y         [B, Y+1, X]      y-faces        ( j    dx, (i+.5)dx)
    v_x         [B, Y, X+1]      x-faces        ((j+.5)dx,  i    dx)
    staggered_tensor  [B, Y+1, X+1, 2]  (component 0 = v_y, 1 = v_x, zero padded high end)
"""
import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

try:  # scipy is only needed for the direct pressure solve
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
except Exception:  # pragma: no cover
    sp = None
    spla = None


# --------------------------------------------------------------------------------------
# geometry / constant masks
# --------------------------------------------------------------------------------------
@dataclass
class KarmanGeometry:
    """Constant masks of the karman-2d scene.

    reference: KarmanFlow.__init__  karman-2d/karman_train.py:166-171
               (Inflow(box[5:10, 25:75]), Obstacle(Sphere([50, 50], 10)));
               domain box [0:2*len, 0:len], OPEN  karman_train.py:363;
               velBCy / velBCyMask  karman_train.py:366-373.
    """
    Y: int
    X: int
    length: float = 100.0
    inflow_antialias: bool = False       # Q3: GeometryMask anti-aliasing of the Inflow box; recalled default off
    dx: float = field(init=False)
    inflow: np.ndarray = field(init=False)      # [Y,X]   1 where cell centre in box[5:10,25:75]
    obstacle: np.ndarray = field(init=False)    # [Y,X]   1 where cell centre inside the sphere
    active: np.ndarray = field(init=False)      # [Y,X]   1 - obstacle
    my: np.ndarray = field(init=False)          # [Y+1,X] hard-BC face mask (y faces)
    mx: np.ndarray = field(init=False)          # [Y,X+1] hard-BC face mask (x faces)
    diag: np.ndarray = field(init=False)        # [Y,X]   pressure-matrix diagonal (negative)
    bc_mask: np.ndarray = field(init=False)     # [Y+1,X] velBCyMask (values 1)

    def __post_init__(self):
        Y, X = self.Y, self.X
        assert Y == 2 * X, "reference domain is box[0:2*len, 0:len] with res (2*res, res)"
        self.dx = self.length / X
        yc = (np.arange(Y) + 0.5) * self.dx
        xc = (np.arange(X) + 0.5) * self.dx
        YC, XC = np.meshgrid(yc, xc, indexing="ij")
        # Box.value_at: inclusive on both sides  [EXT-RECALL A.4]
        self.inflow = ((YC >= 5.0) & (YC <= 10.0) & (XC >= 25.0) & (XC <= 75.0)).astype(np.float64)
        if self.inflow_antialias:
            # anti-aliased variant: linear ramp of one cell width across the box surface, clip(0.5 - sdf/dx, 0, 1) with the
            # signed distance of the cell centre to the box [EXT-RECALL: later PhiFlow 1.x GeometryMask(antialias=True)]
            qy = np.abs(YC - 7.5) - 2.5
            qx = np.abs(XC - 50.0) - 25.0
            sdf = np.sqrt(np.maximum(qy, 0) ** 2 + np.maximum(qx, 0) ** 2) + np.minimum(np.maximum(qy, qx), 0)
            self.inflow = np.clip(0.5 - sdf / self.dx, 0.0, 1.0)
        # Sphere.value_at: dist^2 <= r^2  [EXT-RECALL A.6]
        self.obstacle = (((YC - 50.0) ** 2 + (XC - 50.0) ** 2) <= 10.0 ** 2).astype(np.float64)
        self.active = 1.0 - self.obstacle
        # accessible mask: same data, extrapolation 'boundary' for OPEN (outside = edge value)
        acc = self.active
        acc_pad = np.pad(acc, 1, mode="edge")
        # face mask = min(accessible_lo, accessible_hi)  [EXT-RECALL A.6]
        self.my = np.minimum(acc_pad[0:Y + 1, 1:X + 1], acc_pad[1:Y + 2, 1:X + 1])
        self.mx = np.minimum(acc_pad[1:Y + 1, 0:X + 1], acc_pad[1:Y + 1, 1:X + 2])
        # matrix diagonal: -(number of accessible neighbours), clipped to <= -1 [EXT-RECALL A.7]
        nacc = (acc_pad[0:Y, 1:X + 1] + acc_pad[2:Y + 2, 1:X + 1] +
                acc_pad[1:Y + 1, 0:X] + acc_pad[1:Y + 1, 2:X + 2])
        self.diag = np.minimum(-nacc, -1.0)
        # velBCy mask, karman_train.py:367-371
        vn = np.zeros((Y + 1, X))
        vn[0:2, 0:X - 1] = 1.0
        vn[:, 0:1] = 1.0
        vn[:, -1:] = 1.0
        self.bc_mask = vn

    # pressure matrix A (negative semi-definite 5-point Laplacian with the masks above)
    def pressure_matrix(self):
        """A[c,c] = diag[c];  A[c,n] = active[c]*active[n] for the 4 neighbours inside the
        domain  (outside an OPEN domain: active = 0 -> Dirichlet p = 0).  [EXT-RECALL A.7]"""
        Y, X = self.Y, self.X
        N = Y * X
        idx = np.arange(N).reshape(Y, X)
        rows, cols, vals = [idx.ravel()], [idx.ravel()], [self.diag.ravel()]
        act = self.active
        for (sj, si) in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            j0, j1 = max(0, -sj), min(Y, Y - sj)
            i0, i1 = max(0, -si), min(X, X - si)
            r = idx[j0:j1, i0:i1]
            c = idx[j0 + sj:j1 + sj, i0 + si:i1 + si]
            v = act[j0:j1, i0:i1] * act[j0 + sj:j1 + sj, i0 + si:i1 + si]
            rows.append(r.ravel()); cols.append(c.ravel()); vals.append(v.ravel())
        A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N))
        return A


_GEOM_CACHE = {}


def geometry(Y, X, length=100.0, inflow_antialias=False):
    key = (Y, X, float(length), bool(inflow_antialias))
    if key not in _GEOM_CACHE:
        _GEOM_CACHE[key] = KarmanGeometry(Y, X, length, inflow_antialias)
    return _GEOM_CACHE[key]


def _t(a, like):
    return torch.as_tensor(a, dtype=like.dtype, device=like.device)


# --------------------------------------------------------------------------------------
# C.1  diffusion (explicit, replicate padding) + velocity BC
# --------------------------------------------------------------------------------------
def laplace_replicate(f):
    """5-point Laplacian with replicate padding, dx = 1.  f: [B,H,W].
    reference call site: diffuse(CenteredGrid(...), alpha) karman_train.py:177-178;
    PhiFlow diffuse -> CenteredGrid.laplace with 'boundary' extrapolation [EXT-RECALL A.3]."""
    fp = F.pad(f.unsqueeze(1), (1, 1, 1, 1), mode="replicate").squeeze(1)
    return (fp[:, 2:, 1:-1] + fp[:, :-2, 1:-1] + fp[:, 1:-1, 2:] + fp[:, 1:-1, :-2] - 4.0 * f)


def diffuse_bc(vy, vx, re, res, dt, geom):
    """KarmanFlow.step lines karman_train.py:175-183."""
    alpha = (1.0 / re * dt * res * res).reshape(-1, 1, 1)
    cy = vy + alpha * laplace_replicate(vy)
    cx = vx + alpha * laplace_replicate(vx)
    m = _t(geom.bc_mask, vy)
    cy = cy * (1.0 - m) + m          # velBCy == velBCyMask (values 1)  karman_train.py:372-373
    return cy, cx


# --------------------------------------------------------------------------------------
# C.2  semi-Lagrangian advection on the MAC grid
# --------------------------------------------------------------------------------------
def _sample(fld, ly, lx, mode):
    """Bilinear sample of fld [B,H,W] at local (index-space) coords ly,lx [B,...].
    mode 'replicate': clamp indices (extrapolation 'boundary');
    mode 'zero': one ring of zero ghost cells, coords clamped onto it (extrapolation
                 'constant', PhiFlow pad_constant_boundaries)  [EXT-RECALL A.5];
    mode 'zero_box': Q4 alternative -- zero strictly outside the field's box (local coordinate beyond the outer cell FACES),
                 edge-clamped bilinear inside it (no half-cell blend towards a ghost ring);
    mode 'circular': indices modulo the array length (extrapolation 'periodic')."""
    B, H, W = fld.shape
    if mode == "zero_box":
        inside = ((ly >= -0.5) & (ly <= H - 0.5) & (lx >= -0.5) & (lx <= W - 0.5)).to(fld.dtype)
        return inside * _sample(fld, ly, lx, "replicate")
    if mode == "zero":
        fld = F.pad(fld, (1, 1, 1, 1))
        ly = ly + 1.0
        lx = lx + 1.0
        H, W = H + 2, W + 2
        mode = "replicate"
    fy = torch.floor(ly)
    fx = torch.floor(lx)
    wy = ly - fy
    wx = lx - fx
    y0 = fy.long()
    x0 = fx.long()
    y1 = y0 + 1
    x1 = x0 + 1
    if mode == "replicate":
        y0 = y0.clamp(0, H - 1); y1 = y1.clamp(0, H - 1)
        x0 = x0.clamp(0, W - 1); x1 = x1.clamp(0, W - 1)
    elif mode == "circular":
        y0 = y0 % H; y1 = y1 % H
        x0 = x0 % W; x1 = x1 % W
    else:
        raise ValueError(mode)
    flat = fld.reshape(B, -1)
    shp = ly.shape

    def g(yy, xx):
        return torch.gather(flat, 1, (yy * W + xx).reshape(B, -1)).reshape(shp)

    return ((1 - wy) * ((1 - wx) * g(y0, x0) + wx * g(y0, x1)) +
            wy * ((1 - wx) * g(y1, x0) + wx * g(y1, x1)))


def _points(kind, Y, X, dx, like):
    """Physical sample points of each grid kind, (y, x) each [H,W]."""
    if kind == "c":
        y = (torch.arange(Y, dtype=like.dtype) + 0.5) * dx
        x = (torch.arange(X, dtype=like.dtype) + 0.5) * dx
    elif kind == "y":
        y = torch.arange(Y + 1, dtype=like.dtype) * dx
        x = (torch.arange(X, dtype=like.dtype) + 0.5) * dx
    else:
        y = (torch.arange(Y, dtype=like.dtype) + 0.5) * dx
        x = torch.arange(X + 1, dtype=like.dtype) * dx
    return torch.meshgrid(y, x, indexing="ij")


def _local(kind, py, px, dx):
    """physical -> index-space coords of grid `kind`:  local = (x - box.lower)/dx - 0.5
    with the component box shifted by half a cell along its own axis [EXT-RECALL A.5]."""
    if kind == "c":
        return py / dx - 0.5, px / dx - 0.5
    if kind == "y":
        return (py + 0.5 * dx) / dx - 0.5, px / dx - 0.5
    return py / dx - 0.5, (px + 0.5 * dx) / dx - 0.5


def advect_mac(d, vy, vx, dt, dx, vel_mode="replicate", den_mode="zero"):
    """semi_lagrangian(density, v), semi_lagrangian(v, v)  [EXT-RECALL A.5]:
       x0 = field.points; u = velocity.at(x0); x = x0 - u*dt; data = field.sample_at(x)."""
    B, Yp1, X = vy.shape
    Y = Yp1 - 1
    out = []
    for kind, fld, mode in (("c", d, den_mode), ("y", vy, vel_mode), ("x", vx, vel_mode)):
        if fld is None:
            out.append(None)
            continue
        py, px = _points(kind, Y, X, dx, vy)
        py = py.unsqueeze(0).expand(B, -1, -1)
        px = px.unsqueeze(0).expand(B, -1, -1)
        uy = _sample(vy, *_local("y", py, px, dx), vel_mode)
        ux = _sample(vx, *_local("x", py, px, dx), vel_mode)
        qy = py - uy * dt
        qx = px - ux * dt
        out.append(_sample(fld, *_local(kind, qy, qx, dx), mode))
    return tuple(out)


# --------------------------------------------------------------------------------------
# C.3/C.4  pressure projection
# --------------------------------------------------------------------------------------
class _PressureSolve(torch.autograd.Function):
    """p = A^-1 rhs by sparse LU (float64).  Backward = second solve with the same
    symmetric matrix (PhiFlow's custom gradient of the CG solve [EXT-RECALL A.7])."""

    @staticmethod
    def forward(ctx, rhs, geom):
        key = "_lu"
        if not hasattr(geom, key):
            setattr(geom, key, spla.splu(geom.pressure_matrix().astype(np.float64)))
        lu = getattr(geom, key)
        ctx.geom = geom
        B = rhs.shape[0]
        sol = lu.solve(rhs.detach().double().reshape(B, -1).numpy().T).T
        return torch.as_tensor(sol, dtype=rhs.dtype).reshape(rhs.shape)

    @staticmethod
    def backward(ctx, g):
        return _PressureSolve.apply(g, ctx.geom), None


def apply_A(p, geom):
    """A p  for p [B,Y,X] (matrix-free form of KarmanGeometry.pressure_matrix)."""
    act = _t(geom.active, p)
    diag = _t(geom.diag, p)
    pa = F.pad(p * act, (1, 1, 1, 1))
    nb = pa[:, 2:, 1:-1] + pa[:, :-2, 1:-1] + pa[:, 1:-1, 2:] + pa[:, 1:-1, :-2]
    return diag * p + act * nb


def cg_reference(rhs, geom, accuracy=1e-5, max_iterations=2000, batch_global=True):
    """Restatement of PhiFlow SparseCG's loop [EXT-RECALL A.7]: x0 = 0, stop when
    max|r| < accuracy (max over batch and cells).  Returns (p, iterations[B])."""
    x = torch.zeros_like(rhs)
    r = rhs.clone()
    p = r.clone()
    B = rhs.shape[0]
    its = torch.zeros(B, dtype=torch.long)
    rr = (r * r).sum(dim=(1, 2), keepdim=True)
    for it in range(max_iterations):
        rmax = r.abs().amax(dim=(1, 2))
        live = (rmax >= accuracy)
        if batch_global:
            live = live.any().expand(B)
        if not bool(live.any()):
            break
        Ap = apply_A(p, geom)
        pAp = (p * Ap).sum(dim=(1, 2), keepdim=True)
        a = torch.where(pAp != 0, rr / pAp, torch.zeros_like(pAp))
        lv = live.reshape(B, 1, 1).to(rhs.dtype)
        x = x + lv * a * p
        r = r - lv * a * Ap
        rr_new = (r * r).sum(dim=(1, 2), keepdim=True)
        b = torch.where(rr != 0, rr_new / rr, torch.zeros_like(rr))
        p = torch.where(live.reshape(B, 1, 1), r + b * p, p)
        rr = torch.where(live.reshape(B, 1, 1), rr_new, rr)
        its += live.long()
    return x, its


def divergence(vy, vx):
    """unit-less divergence  sum_d (upper - lower)  [EXT-RECALL A.6]."""
    return (vy[:, 1:, :] - vy[:, :-1, :]) + (vx[:, :, 1:] - vx[:, :, :-1])


def grad_p(p, grad_pad="replicate"):
    """face differences of p.  grad_pad 'replicate' (PhiFlow 1.x: pressure CenteredGrid has
    extrapolation 'boundary' -> boundary-face gradient 0) or 'dirichlet0' (p = 0 outside).
    Open question Q5 of SURVEY.md appendix A; default = recalled PhiFlow behaviour."""
    if grad_pad == "replicate":
        pp = F.pad(p.unsqueeze(1), (1, 1, 1, 1), mode="replicate").squeeze(1)
    else:
        pp = F.pad(p, (1, 1, 1, 1))
    gy = pp[:, 1:, 1:-1] - pp[:, :-1, 1:-1]      # [B,Y+1,X]
    gx = pp[:, 1:-1, 1:] - pp[:, 1:-1, :-1]      # [B,Y,X+1]
    return gy, gx


def project(vy, vx, geom, grad_pad="replicate", solver="direct", return_info=False):
    """divergence_free(v, domain, obstacles=[sphere])  [EXT-RECALL A.6]."""
    my = _t(geom.my, vy)
    mx = _t(geom.mx, vx)
    vy = vy * my
    vx = vx * mx
    div = divergence(vy, vx)
    if solver == "direct":
        p = _PressureSolve.apply(div, geom)
        its = None
    else:
        p, its = cg_reference(div, geom)
    gy, gx = grad_p(p, grad_pad)
    vy = vy - my * gy
    vx = vx - mx * gx
    if return_info:
        return vy, vx, {"pressure": p, "divergence": div, "iterations": its}
    return vy, vx


# --------------------------------------------------------------------------------------
# full solver step  (KarmanFlow.step, karman_train.py:173-185)
# --------------------------------------------------------------------------------------
def karman_step(d, vy, vx, re, geom, dt=1.0, res=None, grad_pad="replicate", solver="direct",
                inflow_order="after", den_mode="zero"):
    """One `simulator_lo.step(...)`:
       diffuse + BC (karman_train.py:175-183)  ->  IncompressibleFlow.step [EXT-RECALL A.4]:
       density = SL(density, v); v = SL(v, v); density += inflow*dt; v = divergence_free(v).
       inflow_order 'before' reproduces the phi2 variant (karman-2d-phi2/karman_train.py:182).
       Switchable recalled choices: Q2 inflow_order, Q3 geometry(..., inflow_antialias=), Q4 den_mode
       ('zero' ghost ring | 'zero_box'), Q5 grad_pad, Q6 solver ('direct' = converged solve | 'cg' = SparseCG restatement with
       accuracy 1e-5 and the batch-global stop); Q7 is burgers_step(periodic_faces=).  The HIP path implements the defaults
       plus Q2, Q3 (the inflow mask is an input array), Q5 and any converged solve for Q6."""
    res = geom.X if res is None else res
    cy, cx = diffuse_bc(vy, vx, re, res, dt, geom)
    infl = _t(geom.inflow, vy)
    if inflow_order == "before":
        d = d + infl            # phi2: advect(density + inflow)
    d2, ay, ax = advect_mac(d, cy, cx, dt, geom.dx, den_mode=den_mode)
    if inflow_order == "after":
        d2 = d2 + infl * dt
    py, px = project(ay, ax, geom, grad_pad=grad_pad, solver=solver)
    return d2, py, px


# --------------------------------------------------------------------------------------
# feature / pad glue   (karman_train.py:77-90)
# --------------------------------------------------------------------------------------
def staggered_tensor(vy, vx):
    """[B,Y+1,X+1,2]: both components zero padded at the high end [EXT-RECALL A.1]."""
    return torch.stack([F.pad(vy, (0, 1, 0, 0)), F.pad(vx, (0, 0, 0, 1))], dim=-1)


def unstack_staggered(t):
    return t[:, :, :-1, 0], t[:, :-1, :, 1]


def to_feature(vy, vx, re):
    """karman_train.py:77-86: staggered_tensor()[:, :-1, :-1, 0:2] ++ ones*Re  -> [B,Y,X,3]."""
    st = staggered_tensor(vy, vx)[:, :-1, :-1, :]
    rech = re.reshape(-1, 1, 1, 1).expand(-1, st.shape[1], st.shape[2], 1).to(st.dtype)
    return torch.cat([st, rech], dim=-1)


def to_staggered(t):
    """karman_train.py:88-90: pad [B,Y,X,2] -> [B,Y+1,X+1,2] then unstack."""
    tp = F.pad(t, (0, 0, 0, 1, 0, 1))
    return unstack_staggered(tp)


# --------------------------------------------------------------------------------------
# CNN  (model_mars_moon, karman_train.py:101-138)
# --------------------------------------------------------------------------------------
MARS_MOON_SHAPES = [(5, 5, 3, 32)] + [(5, 5, 32, 32)] * 10 + [(5, 5, 32, 2)]   # HWIO


def mars_moon_param_shapes(cin=3, cout=2):
    shapes = []
    chans = [cin] + [32] * 11 + [cout]
    for l in range(12):
        shapes.append((5, 5, chans[l], chans[l + 1]))   # kernel HWIO
        shapes.append((chans[l + 1],))                  # bias
    return shapes


def init_params(seed=0, cin=3, cout=2, dtype=torch.float64):
    """Keras defaults: glorot_uniform kernels, zero biases [EXT-RECALL A.10]."""
    g = torch.Generator().manual_seed(seed)
    ps = []
    for shp in mars_moon_param_shapes(cin, cout):
        if len(shp) == 4:
            fan_in = shp[0] * shp[1] * shp[2]
            fan_out = shp[0] * shp[1] * shp[3]
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            ps.append(((torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * lim).to(dtype))
        else:
            ps.append(torch.zeros(shp, dtype=dtype))
    return ps


def _conv(x_nhwc, w_hwio, b):
    x = x_nhwc.permute(0, 3, 1, 2)
    w = w_hwio.permute(3, 2, 0, 1)
    y = F.conv2d(x, w, b, padding=2)
    return y.permute(0, 2, 3, 1)


def mars_moon(params, x, slope=0.3):
    """5x5 SAME convs, LeakyReLU(alpha=0.3 Keras default) [EXT-RECALL A.10]."""
    act = lambda t: F.leaky_relu(t, slope)
    h = act(_conv(x, params[0], params[1]))
    for k in range(5):
        a = act(_conv(h, params[2 + 4 * k], params[3 + 4 * k]))
        c = _conv(a, params[4 + 4 * k], params[5 + 4 * k])
        h = act(h + c)
    return _conv(h, params[22], params[23])


def mercury_param_shapes(cin=3, cout=2):
    """model_mercury, karman_train.py:92-99: Conv2D(32, 5, same, relu), Conv2D(64, 5, same, relu), Conv2D(2, 5, same)."""
    chans = [cin, 32, 64, cout]
    shapes = []
    for l in range(3):
        shapes.append((5, 5, chans[l], chans[l + 1]))
        shapes.append((chans[l + 1],))
    return shapes


def init_params_mercury(seed=0, cin=3, cout=2, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    ps = []
    for shp in mercury_param_shapes(cin, cout):
        if len(shp) == 4:
            lim = math.sqrt(6.0 / (shp[0] * shp[1] * (shp[2] + shp[3])))
            ps.append(((torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * lim).to(dtype))
        else:
            ps.append(torch.zeros(shp, dtype=dtype))
    return ps


def mercury(params, x):
    """karman_train.py:92-99."""
    h = F.relu(_conv(x, params[0], params[1]))
    h = F.relu(_conv(h, params[2], params[3]))
    return _conv(h, params[4], params[5])


# --------------------------------------------------------------------------------------
# unrolled loop + loss  (karman_train.py:397-436)
# --------------------------------------------------------------------------------------
def correction(params, vy, vx, re, std_v, std_re, in_std_v=None, out_std_v=None):
    """karman_train.py:413-424; in_std_v / out_std_v = dataStats['in.std'][1] / ['out.std'], present only with --pretf (:351-355).
    The network is the one the parameter list belongs to (`eval('model_'+params['model'])`, :394): 24 tensors = mars_moon,
    6 = mercury."""
    si = std_v if in_std_v is None else in_std_v
    so = std_v if out_std_v is None else out_std_v
    feat = to_feature(vy, vx, re) / torch.tensor([si[0], si[1], std_re], dtype=vy.dtype)
    net = mercury if len(params) == 6 else mars_moon
    out = net(params, feat) * torch.tensor([so[0], so[1]], dtype=vy.dtype)
    return to_staggered(out)


def unrolled_loss(params, d0, vy0, vx0, re, gt_vy, gt_vx, geom, std_v, std_re, dt=1.0,
                  return_states=False, in_std_v=None, out_std_v=None, **step_kw):
    """loss = sum_i l2_loss((gt_i - prd_i)/std_v)/msteps   karman_train.py:428-436.
    gt_vy/gt_vx: lists of msteps ground-truth frames."""
    msteps = len(gt_vy)
    d, vy, vx = d0, vy0, vx0
    losses = []
    states = []
    sv = torch.tensor([std_v[0], std_v[1]], dtype=vy0.dtype)
    for i in range(msteps):
        d, vy, vx = karman_step(d, vy, vx, re, geom, dt=dt, **step_kw)
        cy, cx = correction(params, vy, vx, re, std_v, std_re, in_std_v, out_std_v)
        vy = vy + cy
        vx = vx + cx
        diff = (staggered_tensor(gt_vy[i], gt_vx[i]) - staggered_tensor(vy, vx)) / sv
        losses.append(0.5 * (diff * diff).sum())
        states.append((d, vy, vx))
    loss = torch.stack(losses).sum() / msteps
    if return_states:
        return loss, losses, states
    return loss


def adam_tf(params, grads, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8, clip_norm=None):
    """tf.compat.v1.train.AdamOptimizer update (epsilon-hat form) [EXT-RECALL A.10];
    optional per-tensor clip_by_norm (karman_train.py:451-454).  t is the 1-based step."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    out_p, out_m, out_v = [], [], []
    for p, g, mi, vi in zip(params, grads, m, v):
        if clip_norm is not None:
            n = torch.sqrt((g * g).sum())
            g = torch.where(n > clip_norm, g * (clip_norm / n), g)
        mi = beta1 * mi + (1 - beta1) * g
        vi = beta2 * vi + (1 - beta2) * g * g
        out_p.append(p - lr_t * mi / (torch.sqrt(vi) + eps))
        out_m.append(mi)
        out_v.append(vi)
    return out_p, out_m, out_v


# --------------------------------------------------------------------------------------
# Burgers  (burgers/burgers_train.py:172-187)
# --------------------------------------------------------------------------------------
def diffuse_periodic_fft(f, amount):
    """PhiFlow diffuse(), periodic branch [EXT-RECALL A.3]: ifft(fft(f) * exp(-(2 pi)^2 |k|^2 amount)),
    k = fftfreq(array shape) (dx = 1 for the component grids, A.9)."""
    H, W = f.shape[-2:]
    ky = torch.fft.fftfreq(H, dtype=f.dtype)
    kx = torch.fft.fftfreq(W, dtype=f.dtype)
    k2 = ky[:, None] ** 2 + kx[None, :] ** 2
    ker = torch.exp(-(2 * math.pi) ** 2 * k2 * amount)
    return torch.fft.ifft2(torch.fft.fft2(f) * ker).real


def burgers_diffusion_matrices(H, W, amount, dtype=torch.float64):
    """The periodic FFT diffusion is separable: out = Cy @ f @ Cx^T with real symmetric
    circulant matrices.  Used by tests to pin the HIP kernel's host-side constants."""
    def circ(n):
        k = torch.fft.fftfreq(n, dtype=torch.float64)
        ker = torch.exp(-(2 * math.pi) ** 2 * k ** 2 * amount)
        col = torch.fft.ifft(ker).real
        idx = (torch.arange(n)[:, None] - torch.arange(n)[None, :]) % n
        return col[idx].to(dtype)
    return circ(H), circ(W)


def burgers_step(vy, vx, dt, nu=0.1, fy=None, fx=None, dx=1.0, diffusion="fft", periodic_faces="array"):
    """BurgersTest.step / step_with_f (burgers_train.py:182-187) -> PhiFlow Burgers.step:
    v = SL(v, v, dt); v = diffuse(v, dt*nu) (periodic -> FFT) [EXT-RECALL A.9]; v += dt*f.
    Q7, handling of the duplicated +1 face of a periodic staggered component:
      periodic_faces 'array' (default, recalled): the component arrays keep Y+1 / X+1 faces and sampling / the FFT wrap
        modulo the ARRAY length;
      periodic_faces 'domain': the duplicated face is a copy of face 0 -- advection and diffusion act on the Y (X) distinct
        faces, wrapping modulo the domain resolution, and the copy is re-attached afterwards."""
    if periodic_faces == "domain":
        cy, cx = vy[:, :-1, :], vx[:, :, :-1]
        B, Y, X = cy.shape
        out = []
        for kind, fld in (("y", cy), ("x", cx)):
            py, px = _points(kind, Y, X, dx, vy)
            py = py[:Y, :X].unsqueeze(0).expand(B, -1, -1)
            px = px[:Y, :X].unsqueeze(0).expand(B, -1, -1)
            uy = _sample(cy, *_local("y", py, px, dx), "circular")
            ux = _sample(cx, *_local("x", py, px, dx), "circular")
            a = _sample(fld, *_local(kind, py - uy * dt, px - ux * dt, dx), "circular")
            out.append(diffuse_periodic_fft(a, dt * nu))
        ay = torch.cat([out[0], out[0][:, :1, :]], dim=1)
        ax = torch.cat([out[1], out[1][:, :, :1]], dim=2)
    elif periodic_faces == "array":
        _, ay, ax = advect_mac(None, vy, vx, dt, dx, vel_mode="circular")
        if diffusion == "fft":
            ay = diffuse_periodic_fft(ay, dt * nu)
            ax = diffuse_periodic_fft(ax, dt * nu)
        else:
            raise ValueError(diffusion)
    else:
        raise ValueError(periodic_faces)
    if fy is not None:
        ay = ay + dt * fy
        ax = ax + dt * fx
    return ay, ax


def burgers_unrolled_loss(params, vy0, vx0, fy, fx, gt_vy, gt_vx, std_v, std_f, dt, nu=0.1, noforce=False):
    """The unrolled Burgers graph of burgers/burgers_train.py:379-437: msteps x [step_with_f (or step) -> CNN correction
    on to_feature(velocity (, force)) / std -> velocity += to_staggered(out * std_v)], loss = sum_i l2_loss((gt_i - prd_i)
    / std_v) / msteps.  fy/fx/gt_*: lists of msteps frames [B,Y+1,X] / [B,Y,X+1]; std_v, std_f: (std_y, std_x) pairs
    (dataStats['std'][0], [1]).  params: mars_moon weights with 4 (2 with noforce) input channels."""
    msteps = len(gt_vy)
    vy, vx = vy0, vx0
    losses = []
    sv = torch.tensor([std_v[0], std_v[1]], dtype=vy0.dtype)
    for i in range(msteps):
        vy, vx = burgers_step(vy, vx, dt, nu, None if noforce else fy[i], None if noforce else fx[i])
        feat = staggered_tensor(vy, vx)[:, :-1, :-1, :] / sv                      # to_feature_noforce: drop the duplicated edges
        if not noforce:
            ff = staggered_tensor(fy[i], fx[i])[:, :-1, :-1, :] / torch.tensor([std_f[0], std_f[1]], dtype=vy0.dtype)
            feat = torch.cat([feat, ff], dim=-1)
        cy, cx = to_staggered(mars_moon(params, feat) * sv)
        vy = vy + cy
        vx = vx + cx
        diff = (staggered_tensor(gt_vy[i], gt_vx[i]) - staggered_tensor(vy, vx)) / sv
        losses.append(0.5 * (diff * diff).sum())
    return torch.stack(losses).sum() / msteps


# --------------------------------------------------------------------------------------
# synthetic inputs    -- shared by tests, smoke() and bench.py
# --------------------------------------------------------------------------------------
def _smooth(g, sweeps=4):
    for _ in range(sweeps):
        gp = F.pad(g.unsqueeze(1), (1, 1, 1, 1), mode="replicate").squeeze(1)
        g = 0.2 * (g + gp[:, 2:, 1:-1] + gp[:, :-2, 1:-1] + gp[:, 1:-1, 2:] + gp[:, 1:-1, :-2])
    return g


def synthetic_state(B, Y, X, seed, dtype=torch.float64, project_it=True):
    """v_y = 1 + 0.2*G1, v_x = 0.2*G2, density = U(0,1); G = smoothed seeded Gaussian noise,
    then projected once so it is divergence free."""
    gen = torch.Generator().manual_seed(seed)
    vy = 1.0 + 0.2 * _smooth(torch.randn(B, Y + 1, X, generator=gen, dtype=torch.float64))
    vx = 0.2 * _smooth(torch.randn(B, Y, X + 1, generator=gen, dtype=torch.float64))
    d = torch.rand(B, Y, X, generator=gen, dtype=torch.float64)
    if project_it:
        vy, vx = project(vy, vx, geometry(Y, X))
    return d.to(dtype), vy.to(dtype), vx.to(dtype)


RE_TRAIN = [160000.0, 320000.0, 640000.0, 1280000.0, 2560000.0, 5120000.0]   # karman-2d/Makefile:22
STD_RE = float(np.std(RE_TRAIN))


# --------------------------------------------------------------------------------------
# the synthetic SOL-<msteps> training workload bench.py times (BASELINE.json configs[2])
# --------------------------------------------------------------------------------------
def bench_workload(B, Y, X, msteps, rank=0, gt_perturb=0.05, last_layer_scale=0.01):
    """Inputs of the benchmark / SOL-32 fixture, generated in float64 and rounded to fp32 values:
    start state = seeded smooth noise (seed 1234+rank, NOT projected) passed through one solver step
    (spin-up: divergence free and consistent with the boundary conditions); ground truth = plain solver
    roll-out of the start state perturbed by 0.05 x a second noise field (seed 4321+rank); weights =
    init_params(0) with the output layer scaled by 0.01 (an untrained Glorot corrector fed back through 32
    solver steps blows the roll-out up).  bench.py builds the same workload with the HIP solver step."""
    r32 = lambda t: t.detach().float().double()
    g = geometry(Y, X)
    re = torch.tensor([RE_TRAIN[i % len(RE_TRAIN)] for i in range(B)], dtype=torch.float64)
    d, vy, vx = (r32(t) for t in synthetic_state(B, Y, X, 1234 + rank, project_it=False))
    with torch.no_grad():
        d0, vy0, vx0 = (r32(t) for t in karman_step(d, vy, vx, re, g))
        _, py, px = (r32(t) for t in synthetic_state(B, Y, X, 4321 + rank, project_it=False))
        gd, gy, gx = d0, r32(vy0 + gt_perturb * (py - 1.0)), r32(vx0 + gt_perturb * px)
        gts_y, gts_x = [], []
        for _ in range(msteps):
            gd, gy, gx = (r32(t) for t in karman_step(gd, gy, gx, re, g))
            gts_y.append(gy)
            gts_x.append(gx)
    params = [r32(p) for p in init_params(0)]
    params[22] = r32(params[22] * last_layer_scale)
    return {"geom": g, "re": re, "d0": d0, "vy0": vy0, "vx0": vx0, "gt_vy": gts_y, "gt_vx": gts_x, "params": params,
            "std_v": (0.2, 0.2), "std_re": STD_RE}
