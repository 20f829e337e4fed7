"""CPU oracle for the karman-3d configuration (BASELINE.json configs[4])  --  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED, and more so than the 2-D oracle: the reference contains NO 3-D code
(/root/reference/README.md:37-38 announces it as future work), PhiFlow 1.5.1 / TensorFlow are not
installable here and no golden vectors exist.  This file is the
DIMENSION-GENERIC restatement of the 2-D path, one spatial axis added:

    step order        /root/reference/karman-2d/karman_train.py:173-185   (KarmanFlow.step)
    scene             karman_train.py:166-171  Inflow(box[5:10, 25:75]) -> box[5:10, 25:75, 25:75],
                      Obstacle(Sphere([50, 50], 10)) -> Sphere([50, 50, 50], 10)  (option obstacle="cylinder":
                      the 2-D disc extruded along z, i.e. the classic wake geometry)
    domain            karman_train.py:363  box[0:2*len, 0:len] -> box[0:2*len, 0:len, 0:len], OPEN
    velocity BC       karman_train.py:366-373  inflow-side planes 0:2 and the lateral walls of the flow component
    network           karman_train.py:101-138  model_mars_moon with Conv3D(5) layers, 4 input channels
                      (three velocity components + Re), 3 output channels
    feature / pad     karman_train.py:77-90
    loss              karman_train.py:428-436

with PhiFlow's dimension-generic operators as recalled in SURVEY.md appendix A (each op below is the 2-D function of
oracle/sol_oracle.py with one more axis).  It is pinned by analytic known-answer tests (tests/test_oracle3d_kat.py), by
reducing to the 2-D oracle on z-invariant inputs, and by the fixture it generated itself (tests/golden/).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Layout (axes y = flow direction, x, z; z contiguous):
    density  d    [B, Y, X, Z]       cell centres
    v_y           [B, Y+1, X, Z]     faces normal to y          component 0 (karman_train.py:367)
    v_x           [B, Y, X+1, Z]     faces normal to x          component 1
    v_z           [B, Y, X, Z+1]     faces normal to z          component 2
"""
import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

try:
    import scipy.fft as sfft
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
except Exception:  # pragma: no cover
    sfft = sp = spla = None


# --------------------------------------------------------------------------------------
# geometry / constant masks
# --------------------------------------------------------------------------------------
@dataclass
class Karman3DGeometry:
    """Constant masks of the karman-3d scene (dimension-generic KarmanFlow.__init__, karman_train.py:166-171, 363-373)."""
    Y: int
    X: int
    Z: int
    length: float = 100.0
    obstacle_kind: str = "sphere"          # "sphere" (dimension-generic Sphere([50]*3, 10)) or "cylinder" (axis along z)
    dx: float = field(init=False)
    inflow: np.ndarray = field(init=False)      # [Y,X,Z]
    obstacle: np.ndarray = field(init=False)    # [Y,X,Z]
    active: np.ndarray = field(init=False)      # [Y,X,Z]
    masks: tuple = field(init=False)            # hard-BC face masks (my [Y+1,X,Z], mx [Y,X+1,Z], mz [Y,X,Z+1])
    diag: np.ndarray = field(init=False)        # pressure-matrix diagonal (negative)
    bc_mask: np.ndarray = field(init=False)     # [Y+1,X,Z]  velBCyMask (values 1)

    def __post_init__(self):
        Y, X, Z = self.Y, self.X, self.Z
        assert Y == 2 * X and Z == X, "karman-3d domain is box[0:2*len, 0:len, 0:len] with res (2*res, res, res)"
        self.dx = self.length / X
        c = [(np.arange(n) + 0.5) * self.dx for n in (Y, X, Z)]
        YC, XC, ZC = np.meshgrid(*c, indexing="ij")
        self.inflow = ((YC >= 5.0) & (YC <= 10.0) & (XC >= 25.0) & (XC <= 75.0) & (ZC >= 25.0) & (ZC <= 75.0)).astype(np.float64)
        if self.obstacle_kind == "sphere":
            r2 = (YC - 50.0) ** 2 + (XC - 50.0) ** 2 + (ZC - 50.0) ** 2
        elif self.obstacle_kind == "cylinder":
            r2 = (YC - 50.0) ** 2 + (XC - 50.0) ** 2 + 0.0 * ZC
        else:
            raise ValueError(self.obstacle_kind)
        self.obstacle = (r2 <= 10.0 ** 2).astype(np.float64)
        self.active = 1.0 - self.obstacle
        acc = np.pad(self.active, 1, mode="edge")           # accessible: 'boundary' extrapolation (OPEN)  [EXT-RECALL A.6]
        core = (slice(1, -1),) * 3
        ms = []
        nacc = np.zeros_like(self.active)
        for ax in range(3):
            lo = [slice(1, -1)] * 3
            hi = [slice(1, -1)] * 3
            lo[ax] = slice(0, -1)
            hi[ax] = slice(1, None)
            ms.append(np.minimum(acc[tuple(lo)], acc[tuple(hi)]))         # faces of axis ax: n+1 along ax
            dn = [slice(1, -1)] * 3
            up = [slice(1, -1)] * 3
            dn[ax] = slice(0, -2)
            up[ax] = slice(2, None)
            nacc += acc[tuple(dn)] + acc[tuple(up)]
        del core
        self.masks = tuple(ms)
        self.diag = np.minimum(-nacc, -1.0)
        vn = np.zeros((Y + 1, X, Z))
        vn[0:2, :, :] = 1.0                 # 2-D: vn[0:2, 0:X-1] u the two side columns = the two inflow-side rows + the walls
        vn[:, 0, :] = 1.0
        vn[:, -1, :] = 1.0
        vn[:, :, 0] = 1.0
        vn[:, :, -1] = 1.0
        self.bc_mask = vn

    def pressure_matrix(self):
        """A[c,c] = diag[c]; A[c,n] = active[c]*active[n] for the 6 neighbours inside the domain [EXT-RECALL A.7]."""
        Y, X, Z = self.Y, self.X, self.Z
        N = Y * X * Z
        idx = np.arange(N).reshape(Y, X, Z)
        rows, cols, vals = [idx.ravel()], [idx.ravel()], [self.diag.ravel()]
        act = self.active
        for ax in range(3):
            lo = [slice(None)] * 3
            hi = [slice(None)] * 3
            lo[ax] = slice(0, -1)
            hi[ax] = slice(1, None)
            r, c = idx[tuple(lo)].ravel(), idx[tuple(hi)].ravel()
            v = (act[tuple(lo)] * act[tuple(hi)]).ravel()
            rows += [r, c]; cols += [c, r]; vals += [v, v]
        return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N))


_GEOM_CACHE = {}


def geometry(Y, X, Z, length=100.0, obstacle="sphere"):
    key = (Y, X, Z, float(length), obstacle)
    if key not in _GEOM_CACHE:
        _GEOM_CACHE[key] = Karman3DGeometry(Y, X, Z, length, obstacle)
    return _GEOM_CACHE[key]


def _t(a, like):
    return torch.as_tensor(a, dtype=like.dtype, device=like.device)


# --------------------------------------------------------------------------------------
# diffusion (explicit, replicate padding, dx = 1) + velocity BC      karman_train.py:175-183
# --------------------------------------------------------------------------------------
def laplace_replicate(f):
    """7-point Laplacian with replicate padding.  f: [B,n0,n1,n2]  (PhiFlow laplace with 'boundary' extrapolation, A.3)."""
    fp = F.pad(f.unsqueeze(1), (1, 1, 1, 1, 1, 1), mode="replicate").squeeze(1)
    c = fp[:, 1:-1, 1:-1, 1:-1]
    return (fp[:, 2:, 1:-1, 1:-1] + fp[:, :-2, 1:-1, 1:-1] + fp[:, 1:-1, 2:, 1:-1] + fp[:, 1:-1, :-2, 1:-1] +
            fp[:, 1:-1, 1:-1, 2:] + fp[:, 1:-1, 1:-1, :-2] - 6.0 * c)


def diffuse_bc(v, re, res, dt, geom):
    """v = (vy, vx, vz).  alpha = dt*res^2/Re per simulation (karman_train.py:175); the BC blend acts on the flow component."""
    alpha = (1.0 / re * dt * res * res).reshape(-1, 1, 1, 1)
    out = [c + alpha * laplace_replicate(c) for c in v]
    m = _t(geom.bc_mask, v[0])
    out[0] = out[0] * (1.0 - m) + m              # velBCy == velBCyMask (values 1), karman_train.py:372-373
    return tuple(out)


# --------------------------------------------------------------------------------------
# semi-Lagrangian advection on the MAC grid      [EXT-RECALL A.5]
# --------------------------------------------------------------------------------------
def _sample(fld, loc, mode):
    """Trilinear sample of fld [B,n0,n1,n2] at index-space coordinates loc = (l0, l1, l2), each [B,...].
    mode 'replicate': clamp indices; 'zero': one ring of zero ghost cells, coordinates clamped onto it."""
    if mode == "zero":
        fld = F.pad(fld, (1, 1, 1, 1, 1, 1))
        loc = tuple(l + 1.0 for l in loc)
    elif mode != "replicate":
        raise ValueError(mode)
    B = fld.shape[0]
    n = fld.shape[1:]
    fl = [torch.floor(l) for l in loc]
    w = [l - f for l, f in zip(loc, fl)]
    i0 = [f.long() for f in fl]
    idx = [(a.clamp(0, m - 1), (a + 1).clamp(0, m - 1)) for a, m in zip(i0, n)]
    flat = fld.reshape(B, -1)
    shp = loc[0].shape
    out = 0.0
    for a in (0, 1):
        for b in (0, 1):
            for c in (0, 1):
                lin = (idx[0][a] * n[1] + idx[1][b]) * n[2] + idx[2][c]
                val = torch.gather(flat, 1, lin.reshape(B, -1)).reshape(shp)
                wt = (w[0] if a else 1 - w[0]) * (w[1] if b else 1 - w[1]) * (w[2] if c else 1 - w[2])
                out = out + wt * val
    return out


def _points(kind, n, dx, like):
    """Physical sample points of grid `kind` ('c' or the face axis 0/1/2): tuple of three [n0',n1',n2'] arrays."""
    ax = []
    for a in range(3):
        if kind == a:
            ax.append(torch.arange(n[a] + 1, dtype=like.dtype) * dx)
        else:
            ax.append((torch.arange(n[a], dtype=like.dtype) + 0.5) * dx)
    return torch.meshgrid(*ax, indexing="ij")


def _local(kind, p, dx):
    """physical -> index-space coordinates of grid `kind` (component boxes are shifted by half a cell along their axis)."""
    return tuple((p[a] + (0.5 * dx if kind == a else 0.0)) / dx - 0.5 for a in range(3))


def advect_mac(d, v, dt, dx, vel_mode="replicate", den_mode="zero"):
    """semi_lagrangian(density, v), semi_lagrangian(v, v):  x0 = points; u = v.at(x0); x = x0 - u*dt; sample_at(x)."""
    B = v[0].shape[0]
    n = (v[1].shape[1], v[0].shape[2], v[0].shape[3])
    out = []
    for kind, fld, mode in (("c", d, den_mode), (0, v[0], vel_mode), (1, v[1], vel_mode), (2, v[2], vel_mode)):
        if fld is None:
            out.append(None)
            continue
        p = [q.unsqueeze(0).expand(B, -1, -1, -1) for q in _points(kind, n, dx, v[0])]
        u = [_sample(v[a], _local(a, p, dx), vel_mode) for a in range(3)]
        q = [p[a] - u[a] * dt for a in range(3)]
        out.append(_sample(fld, _local(kind, q, dx), mode))
    return out[0], (out[1], out[2], out[3])


# --------------------------------------------------------------------------------------
# pressure projection      [EXT-RECALL A.6-A.8]
# --------------------------------------------------------------------------------------
def apply_A(p, geom):
    act = _t(geom.active, p)
    diag = _t(geom.diag, p)
    pa = F.pad(p * act, (1, 1, 1, 1, 1, 1))
    nb = (pa[:, 2:, 1:-1, 1:-1] + pa[:, :-2, 1:-1, 1:-1] + pa[:, 1:-1, 2:, 1:-1] + pa[:, 1:-1, :-2, 1:-1] +
          pa[:, 1:-1, 1:-1, 2:] + pa[:, 1:-1, 1:-1, :-2])
    return diag * p + act * nb


def _rect_eigenvalues(n):
    lam = 0.0
    for a, m in enumerate(n):
        k = np.arange(1, m + 1)
        shape = [1, 1, 1]
        shape[a] = m
        lam = lam + (2.0 - 2.0 * np.cos(np.pi * k / (m + 1))).reshape(shape)
    return lam


def rect_solve(b, lam):
    """(M_r)^-1 b for the Dirichlet 7-point Laplacian M_r = -A_r of the empty OPEN box: orthonormal DST-I diagonalises it."""
    bh = sfft.dstn(b, type=1, axes=(-3, -2, -1), norm="ortho")
    return sfft.dstn(bh / lam, type=1, axes=(-3, -2, -1), norm="ortho")


def solve_pcg(rhs, geom, rtol=1e-13, max_iter=500):
    """A p = rhs in float64 by conjugate gradients on M = -A preconditioned with the empty-box solve (the obstacle is a
    low-rank perturbation: a handful of iterations).  numpy [B,Y,X,Z] -> numpy.  Used where sparse LU does not fit."""
    lam = _rect_eigenvalues(rhs.shape[1:])
    Mv = lambda x: -apply_A(torch.as_tensor(x), geom).numpy()
    b = -np.asarray(rhs, dtype=np.float64)
    x = np.zeros_like(b)
    r = b.copy()
    z = rect_solve(r, lam)
    p = z.copy()
    rz = (r * z).sum(axis=(1, 2, 3), keepdims=True)
    bn = np.sqrt((b * b).sum(axis=(1, 2, 3), keepdims=True)) + 1e-300
    for _ in range(max_iter):
        Mp = Mv(p)
        a = rz / np.maximum((p * Mp).sum(axis=(1, 2, 3), keepdims=True), 1e-300)
        x += a * p
        r -= a * Mp
        if np.all(np.sqrt((r * r).sum(axis=(1, 2, 3), keepdims=True)) <= rtol * bn):
            break
        z = rect_solve(r, lam)
        rz_new = (r * z).sum(axis=(1, 2, 3), keepdims=True)
        p = z + (rz_new / np.maximum(rz, 1e-300)) * p
        rz = rz_new
    return x


LU_MAX_CELLS = 20000


class _PressureSolve(torch.autograd.Function):
    """p = A^-1 rhs (float64): sparse LU on small grids, preconditioned CG beyond.  Backward = second solve with the same
    symmetric matrix (PhiFlow's custom gradient of the CG solve, A.7)."""

    @staticmethod
    def forward(ctx, rhs, geom):
        ctx.geom = geom
        B = rhs.shape[0]
        r = rhs.detach().double().numpy()
        if geom.Y * geom.X * geom.Z <= LU_MAX_CELLS:
            if not hasattr(geom, "_lu"):
                geom._lu = spla.splu(geom.pressure_matrix().astype(np.float64))
            sol = geom._lu.solve(r.reshape(B, -1).T).T.reshape(r.shape)
        else:
            sol = solve_pcg(r, geom)
        return torch.as_tensor(sol, dtype=rhs.dtype)

    @staticmethod
    def backward(ctx, g):
        return _PressureSolve.apply(g, ctx.geom), None


def divergence(v):
    return ((v[0][:, 1:] - v[0][:, :-1]) + (v[1][:, :, 1:] - v[1][:, :, :-1]) + (v[2][:, :, :, 1:] - v[2][:, :, :, :-1]))


def grad_p(p, grad_pad="replicate"):
    """face differences of p; 'replicate': boundary-face gradient 0 (PhiFlow 1.x), 'dirichlet0': p = 0 outside (Q5)."""
    if grad_pad == "replicate":
        pp = F.pad(p.unsqueeze(1), (1, 1, 1, 1, 1, 1), mode="replicate").squeeze(1)
    else:
        pp = F.pad(p, (1, 1, 1, 1, 1, 1))
    gy = pp[:, 1:, 1:-1, 1:-1] - pp[:, :-1, 1:-1, 1:-1]
    gx = pp[:, 1:-1, 1:, 1:-1] - pp[:, 1:-1, :-1, 1:-1]
    gz = pp[:, 1:-1, 1:-1, 1:] - pp[:, 1:-1, 1:-1, :-1]
    return gy, gx, gz


def project(v, geom, grad_pad="replicate", return_info=False):
    m = [_t(a, v[0]) for a in geom.masks]
    v = [c * mk for c, mk in zip(v, m)]
    div = divergence(v)
    p = _PressureSolve.apply(div, geom)
    g = grad_p(p, grad_pad)
    out = tuple(c - mk * gc for c, mk, gc in zip(v, m, g))
    if return_info:
        return out, {"pressure": p, "divergence": div}
    return out


# --------------------------------------------------------------------------------------
# full solver step   (dimension-generic KarmanFlow.step, karman_train.py:173-185)
# --------------------------------------------------------------------------------------
def karman3d_step(d, v, re, geom, dt=1.0, res=None, grad_pad="replicate", inflow_order="after"):
    res = geom.X if res is None else res
    c = diffuse_bc(v, re, res, dt, geom)
    infl = _t(geom.inflow, v[0])
    if inflow_order == "before":
        d = d + infl
    d2, a = advect_mac(d, c, dt, geom.dx)
    if inflow_order == "after":
        d2 = d2 + infl * dt
    return d2, project(a, geom, grad_pad=grad_pad)


# --------------------------------------------------------------------------------------
# feature / pad glue  (karman_train.py:77-90) and the network (karman_train.py:101-138 with Conv3D)
# --------------------------------------------------------------------------------------
def to_feature(v, re):
    """[B,Y,X,Z,4]: the three components at the low faces of every cell + Re."""
    Y, X, Z = v[1].shape[1], v[0].shape[2], v[0].shape[3]
    st = torch.stack([v[0][:, :Y], v[1][:, :, :X], v[2][:, :, :, :Z]], dim=-1)
    rech = re.reshape(-1, 1, 1, 1, 1).expand(-1, Y, X, Z, 1).to(st.dtype)
    return torch.cat([st, rech], dim=-1)


def to_staggered(t):
    """[B,Y,X,Z,3] -> zero-padded at the high end of each component's own axis (to_staggered, karman_train.py:88-90)."""
    return (F.pad(t[..., 0], (0, 0, 0, 0, 0, 1)), F.pad(t[..., 1], (0, 0, 0, 1)), F.pad(t[..., 2], (0, 1)))


def mars_moon3d_param_shapes(cin=4, cout=3):
    chans = [cin] + [32] * 11 + [cout]
    shapes = []
    for l in range(12):
        shapes.append((5, 5, 5, chans[l], chans[l + 1]))      # Keras Conv3D kernel: (k_y, k_x, k_z, in, out)
        shapes.append((chans[l + 1],))
    return shapes


def init_params(seed=0, cin=4, cout=3, dtype=torch.float64):
    """Keras defaults: glorot_uniform kernels, zero biases [EXT-RECALL A.10]."""
    g = torch.Generator().manual_seed(seed)
    ps = []
    for shp in mars_moon3d_param_shapes(cin, cout):
        if len(shp) == 5:
            lim = math.sqrt(6.0 / (125 * (shp[3] + shp[4])))
            ps.append(((torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * lim).to(dtype))
        else:
            ps.append(torch.zeros(shp, dtype=dtype))
    return ps


def conv3d_same(x, w, b):
    """x [B,Y,X,Z,Ci], w [5,5,5,Ci,Co] -> [B,Y,X,Z,Co], zero padding 2."""
    y = F.conv3d(x.permute(0, 4, 1, 2, 3), w.permute(4, 3, 0, 1, 2), b, padding=2)
    return y.permute(0, 2, 3, 4, 1)


def mars_moon3d(params, x, slope=0.3):
    act = lambda t: F.leaky_relu(t, slope)
    h = act(conv3d_same(x, params[0], params[1]))
    for k in range(5):
        a = act(conv3d_same(h, params[2 + 4 * k], params[3 + 4 * k]))
        c = conv3d_same(a, params[4 + 4 * k], params[5 + 4 * k])
        h = act(h + c)
    return conv3d_same(h, params[22], params[23])


def correction(params, v, re, std_v, std_re):
    """karman_train.py:413-424 with three velocity channels."""
    feat = to_feature(v, re) / torch.tensor(list(std_v) + [std_re], dtype=v[0].dtype)
    out = mars_moon3d(params, feat) * torch.tensor(list(std_v), dtype=v[0].dtype)
    return to_staggered(out)


def rollout(params, d, v, re, geom, std_v, std_re, nsteps, dt=1.0, **step_kw):
    """nsteps x [solver step -> CNN correction -> add]  (the forward half of karman_train.py:397-426 / karman_apply.py:138-158)."""
    states = []
    for _ in range(nsteps):
        d, v = karman3d_step(d, v, re, geom, dt=dt, **step_kw)
        c = correction(params, v, re, std_v, std_re)
        v = tuple(a + b for a, b in zip(v, c))
        states.append((d, v))
    return states


def unrolled_loss(params, d, v, re, gts, geom, std_v, std_re, dt=1.0, **step_kw):
    """loss = sum_i l2_loss((gt_i - prd_i)/std_v)/msteps  (karman_train.py:428-436); gts: list of (vy, vx, vz) frames."""
    losses = []
    for gt in gts:
        d, v = karman3d_step(d, v, re, geom, dt=dt, **step_kw)
        c = correction(params, v, re, std_v, std_re)
        v = tuple(a + b for a, b in zip(v, c))
        losses.append(sum(0.5 * (((g - a) / s) ** 2).sum() for g, a, s in zip(gt, v, std_v)))
    return torch.stack(losses).sum() / len(gts)


# --------------------------------------------------------------------------------------
# synthetic inputs (the 3-D twin of sol_oracle.synthetic_state / SURVEY.md section 8d)
# --------------------------------------------------------------------------------------
def _smooth(g, sweeps=4):
    for _ in range(sweeps):
        gp = F.pad(g.unsqueeze(1), (1, 1, 1, 1, 1, 1), mode="replicate").squeeze(1)
        g = (g + gp[:, 2:, 1:-1, 1:-1] + gp[:, :-2, 1:-1, 1:-1] + gp[:, 1:-1, 2:, 1:-1] + gp[:, 1:-1, :-2, 1:-1] +
             gp[:, 1:-1, 1:-1, 2:] + gp[:, 1:-1, 1:-1, :-2]) / 7.0
    return g


def synthetic_state(B, Y, X, Z, seed, dtype=torch.float64):
    """v_y = 1 + 0.2*G1, v_x = 0.2*G2, v_z = 0.2*G3, density = U(0,1); G = smoothed seeded Gaussian noise (NOT projected:
    the first solver step makes the state divergence free, as bench_workload does in 2-D)."""
    gen = torch.Generator().manual_seed(seed)
    vy = 1.0 + 0.2 * _smooth(torch.randn(B, Y + 1, X, Z, generator=gen, dtype=torch.float64))
    vx = 0.2 * _smooth(torch.randn(B, Y, X + 1, Z, generator=gen, dtype=torch.float64))
    vz = 0.2 * _smooth(torch.randn(B, Y, X, Z + 1, generator=gen, dtype=torch.float64))
    d = torch.rand(B, Y, X, Z, generator=gen, dtype=torch.float64)
    return d.to(dtype), (vy.to(dtype), vx.to(dtype), vz.to(dtype))


RE_TRAIN = [160000.0, 320000.0, 640000.0, 1280000.0, 2560000.0, 5120000.0]   # karman-2d/Makefile:22
STD_RE = float(np.std(RE_TRAIN))
