import os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sol_amd, sol_oracle as o
DEV = "cuda"
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))
B, Y, X, dt, ms = 5, 32, 32, 0.1, 1
gen = torch.Generator().manual_seed(13)
dom = sol_amd.Domain([Y, X], box=sol_amd.box([32, 32]), boundaries=sol_amd.PERIODIC)
velo = (0.3 * torch.randn(ms + 1, B, Y + 1, X + 1, 2, generator=gen, dtype=torch.float32))
forc = (0.1 * torch.randn(ms, B, Y + 1, X + 1, 2, generator=gen, dtype=torch.float32))
params = [p.clone() for p in o.init_params(2, cin=4)]
gb = torch.Generator().manual_seed(9)
params = [(p + 0.01 * torch.randn(p.shape, generator=gb, dtype=torch.float64)).float().double().requires_grad_(True) for p in params]
std_v, std_f = (0.21, 0.19), (0.09, 0.11)
vy = [velo[k][:, :, :X, 0].double() for k in range(ms + 1)]
vx = [velo[k][:, :Y, :, 1].double() for k in range(ms + 1)]
fy = [forc[k][:, :, :X, 0].double() for k in range(ms)]
fx = [forc[k][:, :Y, :, 1].double() for k in range(ms)]
loss = o.burgers_unrolled_loss(params, vy[0], vx[0], fy, fx, vy[1:], vx[1:], std_v, std_f, dt, noforce=False)
loss.backward()
gref = torch.cat([p.grad.reshape(-1) for p in params])
res = {}
for rep in range(2):
    for sched in ("manual", "autograd"):
        net = sol_amd.model_mars_moon(cin=4, cout=2, seed=0, device=DEV)
        net.set_weights([p.detach().numpy() for p in params])
        tr = sol_amd.BurgersTrainer(net, dom, B, ms, dt, std_v, std_f, use_graph=False, schedule=sched)
        # note: the oracle's padded-loss constant: frames' padding enters the trainer's loss only
        l = float(tr.fwd_bwd(velo.to(DEV), forc.to(DEV)))
        g = net.params.grad.detach().clone()
        res[(sched, rep)] = g
        off = net.offsets
        print(sched, rep, "loss", l, float(loss), "grad rel vs oracle %.2e" % rel(g, gref))
        print("   per tensor:", " ".join("%.1e" % rel(g[off[k]:off[k + 1]], gref[off[k]:off[k + 1]]) for k in range(24)))
print("manual rep0 vs rep1 equal:", torch.equal(res[("manual", 0)], res[("manual", 1)]), " autograd rep0 vs rep1 equal:", torch.equal(res[("autograd", 0)], res[("autograd", 1)]))

# ---- which fused backward launch disagrees with its unfused composition? ------------------------------------------------------------
from sol_amd import schedule2d, ops as _ops
orig = schedule2d.NetSchedule2D._conv
count = [0]
def checked(self, x, packed, bias, residual, act_ref, cout, epi, xmax, ymax):
    y = orig(self, x, packed, bias, residual, act_ref, cout, epi, xmax, ymax)
    if epi == _ops.EPI_DLRELU:
        plain = orig(self, x, packed, bias, None, None, cout, _ops.EPI_NONE, xmax, None)
        if residual is not None:
            plain = plain + residual
        ref = plain * torch.where(act_ref > 0, torch.ones_like(act_ref), torch.full_like(act_ref, self.net.slope))
        bad = (y - ref).abs() > 1e-5 * ref.abs().max()
        print("DLRELU launch %d: x %s cin %d  residual %s  rel diff %.2e  bad elements %d  zeros in act_ref %d  min|act_ref| %.2e" % (
            count[0], tuple(x.shape), x.shape[-1], residual is not None, rel(y, ref), int(bad.sum()), int((act_ref == 0).sum()), float(act_ref.abs().min())))
        if bad.any():
            idx = bad.nonzero()[:5]
            for i in idx:
                i = tuple(int(v) for v in i)
                print("      at", i, "fused", float(y[i]), "ref", float(ref[i]), "act_ref", float(act_ref[i]), "plain", float(plain[i]))
        count[0] += 1
    return y
schedule2d.NetSchedule2D._conv = checked
net = sol_amd.model_mars_moon(cin=4, cout=2, seed=0, device=DEV)
net.set_weights([p.detach().numpy() for p in params])
tr = sol_amd.BurgersTrainer(net, dom, B, ms, dt, std_v, std_f, use_graph=False, schedule="manual")
tr.fwd_bwd(velo.to(DEV), forc.to(DEV))

# ---- per unit: the accumulated partial's reduction against a fresh single-call weight gradient on the same operands ------------------
schedule2d.NetSchedule2D._conv = orig
rec = []
orig_bww = schedule2d.NetSchedule2D._bww
def rec_bww(self, u, xk, dz):
    rec.append((u, xk, dz))
    return orig_bww(self, u, xk, dz)
schedule2d.NetSchedule2D._bww = rec_bww
net = sol_amd.model_mars_moon(cin=4, cout=2, seed=0, device=DEV)
net.set_weights([p.detach().numpy() for p in params])
tr = sol_amd.BurgersTrainer(net, dom, B, ms, dt, std_v, std_f, use_graph=False, schedule="manual")
tr.fwd_bwd(velo.to(DEV), forc.to(DEV))
g = net.params.grad.detach().clone()
print("HOOKED (bww recorded) run: grad vs oracle %.2e" % rel(g, gref))
lib = sol_amd.load()
from sol_amd._lib import ptr, stream, check
off = net.offsets
sch = tr._sched
for (u, xk, dz) in rec:
    l = sch.units.index(u)
    part = torch.zeros(u.ws, dtype=torch.float32, device=DEV)
    check(lib.sol_conv5x5_bwd_weight(stream(), ptr(xk), ptr(dz), ptr(part), B, Y, X, u.cin_k, u.cout))
    dw = torch.empty(5, 5, u.cin, u.cout, dtype=torch.float32, device=DEV); db = torch.empty(u.cout, dtype=torch.float32, device=DEV)
    check(lib.sol_conv5x5_bwd_weight_reduce(stream(), ptr(part), ptr(dw), ptr(db), B, Y, X, u.cin, u.cout, 0))
    # float64 reference of the weight gradient from the same operands
    xx = xk[..., :u.cin].double().permute(0, 3, 1, 2); zz = dz.double().permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(xx, (u.cout, u.cin, 5, 5), zz, padding=2).permute(2, 3, 1, 0)
    print("unit %2d: schedule dW vs fresh %.2e  fresh vs float64 %.2e  schedule vs float64 %.2e   db: schedule vs sum(dz) %.2e  (part offset %d floats, ws %d)" % (
        l, rel(g[off[2 * l]:off[2 * l + 1]], dw.reshape(-1)), rel(dw, ref), rel(g[off[2 * l]:off[2 * l + 1]].reshape(ref.shape), ref),
        rel(g[off[2 * l + 1]:off[2 * l + 2]], dz.double().sum((0, 1, 2))), (u.part.data_ptr() - sch._partials.data_ptr()) // 4, u.ws))

# ---- are the saved activations intact after the step, and equal to a plain layer-by-layer forward? ------------------------------------
schedule2d.NetSchedule2D._bww = orig_bww
orig_fwd = schedule2d.NetSchedule2D.forward
saved = {}
def fwd(self, x):
    out, st = orig_fwd(self, x)
    torch.cuda.synchronize()
    saved["x"] = x.clone(); saved["acts"] = [a.clone() for a in st[2]]; saved["live"] = st[2]; saved["out"] = out.clone(); saved["outlive"] = out
    return out, st
schedule2d.NetSchedule2D.forward = fwd
net = sol_amd.model_mars_moon(cin=4, cout=2, seed=0, device=DEV)
net.set_weights([p.detach().numpy() for p in params])
tr = sol_amd.BurgersTrainer(net, dom, B, ms, dt, std_v, std_f, use_graph=False, schedule="manual")
tr.fwd_bwd(velo.to(DEV), forc.to(DEV))
torch.cuda.synchronize()
print("activations intact after the step:", [bool(torch.equal(a, b)) for a, b in zip(saved["acts"], saved["live"])], bool(torch.equal(saved["out"], saved["outlive"])))
# plain forward in float64 torch
p64 = [p.detach().double() for p in params]
def c64(x, w, b):
    return torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, padding=2).permute(0, 2, 3, 1)
lre = lambda t: torch.nn.functional.leaky_relu(t, 0.3)
x = saved["x"].double().cpu()
ref = [lre(c64(x, p64[0], p64[1]))]
for k in range(5):
    a = lre(c64(ref[-1], p64[2 + 4 * k], p64[3 + 4 * k]))
    ref.append(a)
    ref.append(lre(c64(a, p64[4 + 4 * k], p64[5 + 4 * k]) + ref[-2]))
print("activations vs float64 forward:", " ".join("%.1e" % rel(a, r) for a, r in zip(saved["acts"], ref)))

# ---- every backward-data launch against a float64 torch transposed convolution -------------------------------------------------------
schedule2d.NetSchedule2D.forward = orig_fwd
cnt = [0]
def checked2(self, x, packed, bias, residual, act_ref, cout, epi, xmax, ymax):
    y = orig(self, x, packed, bias, residual, act_ref, cout, epi, xmax, ymax)
    for l, u in enumerate(self.units):
        if packed is u.pb:
            w = u.w.double().cpu()                                    # [5,5,cin,cout] of the forward layer
            g = x[..., :u.cout].double().cpu().permute(0, 3, 1, 2)
            ref = torch.nn.functional.conv_transpose2d(g, w.permute(3, 2, 0, 1), padding=2).permute(0, 2, 3, 1)     # [B,H,W,cin]
            if residual is not None:
                ref = ref + residual.double().cpu()
            if epi == _ops.EPI_DLRELU:
                a = act_ref.double().cpu()
                ref = ref * torch.where(a > 0, torch.ones_like(a), torch.full_like(a, self.net.slope))
            fresh = _ops._pack(u.w, u.cout, u.cin, _ops.CONV_BWD_DATA)
            print("bwd-data launch %2d (unit %2d): vs float64 %.2e   packed buffer equals a fresh pack: %s" % (cnt[0], l, rel(y[..., :ref.shape[-1]], ref), bool(torch.equal(fresh, packed))))
            cnt[0] += 1
    return y
schedule2d.NetSchedule2D._conv = checked2
net = sol_amd.model_mars_moon(cin=4, cout=2, seed=0, device=DEV)
net.set_weights([p.detach().numpy() for p in params])
tr = sol_amd.BurgersTrainer(net, dom, B, ms, dt, std_v, std_f, use_graph=False, schedule="manual")
tr.fwd_bwd(velo.to(DEV), forc.to(DEV))

print("HOOKED (bwd-data checked) run: grad vs oracle %.2e" % rel(net.params.grad, gref))
schedule2d.NetSchedule2D._conv = orig
for trial in range(3):
    net = sol_amd.model_mars_moon(cin=4, cout=2, seed=0, device=DEV)
    net.set_weights([p.detach().numpy() for p in params])
    tr = sol_amd.BurgersTrainer(net, dom, B, ms, dt, std_v, std_f, use_graph=False, schedule="manual")
    tr.fwd_bwd(velo.to(DEV), forc.to(DEV))
    g1 = net.params.grad.detach().clone()
    tr.fwd_bwd(velo.to(DEV), forc.to(DEV))
    print("plain run %d: grad vs oracle %.2e, second call %.2e" % (trial, rel(g1, gref), rel(net.params.grad, gref)))

# ---- the network's reverse sweep alone: NetSchedule2D against float64 torch autograd on the same features and output gradient -------
sch = schedule2d.NetSchedule2D(net, B, Y, X)
gen2 = torch.Generator().manual_seed(1)
feat = saved["x"]
dO = torch.randn(B, Y, X, 2, generator=gen2, dtype=torch.float32).to(DEV)
with torch.no_grad():
    sch.begin_step()
    out, st = sch.forward(feat)
    dx = sch.backward(st, dO)
    flat = sch.end_step()
pp = [p.detach().double().clone().requires_grad_(True) for p in params]
xx = feat.double().cpu().requires_grad_(True)
h = lre(c64(xx, pp[0], pp[1]))
for k in range(5):
    a = lre(c64(h, pp[2 + 4 * k], pp[3 + 4 * k]))
    h = lre(c64(a, pp[4 + 4 * k], pp[5 + 4 * k]) + h)
o64 = c64(h, pp[22], pp[23])
(o64 * dO.double().cpu()).sum().backward()
g64 = torch.cat([p.grad.reshape(-1) for p in pp])
print("NETWORK ALONE: out %.2e  dx %.2e  flat gradient %.2e" % (rel(out, o64), rel(dx, xx.grad), rel(flat, g64)))
print("   per tensor:", " ".join("%.1e" % rel(flat[off[k]:off[k + 1]], g64[off[k]:off[k + 1]]) for k in range(24)))

# ---- forward activations: schedule vs the autograd composition's own launches, bit for bit ---------------------------------------------
with torch.no_grad():
    p = net.tensors()
    s_ = net.slope
    hh = _ops.conv5x5(feat, p[0], p[1], None, True, s_)
    auto = [hh]
    for k in range(5):
        aa = _ops.conv5x5(hh, p[2 + 4 * k], p[3 + 4 * k], None, True, s_)
        hh = _ops.conv5x5(aa, p[4 + 4 * k], p[5 + 4 * k], hh, True, s_)
        auto += [aa, hh]
acts = st[2]
print("activations schedule == autograd launches:", [bool(torch.equal(a, b)) for a, b in zip(acts, auto)])
for i, (a, b) in enumerate(zip(acts, auto)):
    r64 = ref[i].to(DEV) if False else None
    flips = int(((a > 0) != (b > 0)).sum())
    a64 = None
    print("   act %2d: sign flips schedule/autograd %d, |min| %.2e, max abs diff %.2e" % (i, flips, float(a.abs().min()), float((a - b).abs().max())))

# ---- the autograd composition on the same features / output gradient, with the gradient of every activation recorded -----------------
net.params.grad = None
grads_auto = {}
def hk(name):
    def f(g):
        grads_auto[name] = g.detach().clone()
    return f
p = net.tensors()
hh = _ops.conv5x5(feat, p[0], p[1], None, True, s_); hh.register_hook(hk("act0"))
for k in range(5):
    aa = _ops.conv5x5(hh, p[2 + 4 * k], p[3 + 4 * k], None, True, s_); aa.register_hook(hk("act%d" % (1 + 2 * k)))
    hh = _ops.conv5x5(aa, p[4 + 4 * k], p[5 + 4 * k], hh, True, s_); hh.register_hook(hk("act%d" % (2 + 2 * k)))
oo = _ops.conv5x5(hh, p[22], p[23], None, False, s_)
(oo * dO).sum().backward()
ga = net.params.grad.detach().clone()
print("AUTOGRAD network alone: flat gradient vs float64 %.2e ; schedule vs autograd %.2e" % (rel(ga, g64), rel(flat, ga)))
print("   per tensor autograd vs float64:", " ".join("%.1e" % rel(ga[off[k]:off[k + 1]], g64[off[k]:off[k + 1]]) for k in range(24)))
# float64 gradients of the activations
