#!/usr/bin/env python
"""Experiment: the batch of 6 as TWO independent 3-simulation training graphs replayed concurrently on two streams
(the solver kernels of one chain can then overlap with the convolutions of the other)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sol_amd
from sol_amd import ops, synthetic
dev = torch.device("cuda", 0)
X, Y, B, ms = 64, 128, 6, 32
dom = sol_amd.Domain([Y, X], box=sol_amd.box[0:200, 0:100])
flow = sol_amd.KarmanFlow()
active, inflow = flow.scene_arrays(dom)
bcv, bcm = sol_amd.velocity_bc_masks(Y, X)
masks = ops.SceneMasks(active, inflow, bcv.reshape(Y + 1, X), bcm.reshape(Y + 1, X), dev)
net = sol_amd.model_mars_moon(cin=3, cout=2, seed=0, device=dev)
with torch.no_grad():
    net.tensors()[22].mul_(0.01)
std_v = (0.2, 0.2)
f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
d0, vy0, vx0 = (f(t) for t in synthetic.state(B, Y, X, 1234))
re = f(synthetic.reynolds(B))
cfgk = ops.karman_cfg(B, Y, X, dom.dx[1], masks=masks)
d0, vy0, vx0 = (t.detach().contiguous() for t in ops.karman_step(d0, vy0, vx0, re, cfgk, masks))
gt_vy = torch.stack([vy0] * ms).contiguous(); gt_vx = torch.stack([vx0] * ms).contiguous()
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
Bs = B // nsplit
trs = [sol_amd.SolTrainer(net, masks, Bs, Y, X, ms, dom.dx[1], std_v, synthetic.STD_RE, use_graph=True) for _ in range(nsplit)]
streams = [torch.cuda.Stream() for _ in range(nsplit)]
sl = [slice(k * Bs, (k + 1) * Bs) for k in range(nsplit)]
args = [(d0[s].contiguous(), vy0[s].contiguous(), vx0[s].contiguous(), re[s].contiguous(), gt_vy[:, s].contiguous(), gt_vx[:, s].contiguous()) for s in sl]
def step():
    cur = torch.cuda.current_stream()
    for k in range(nsplit):
        streams[k].wait_stream(cur)
        with torch.cuda.stream(streams[k]):
            trs[k].fwd_bwd(*args[k], want_final=True)
    for k in range(nsplit):
        cur.wait_stream(streams[k])
    g = trs[0].grads
    for k in range(1, nsplit):
        g = g + trs[k].grads
    return g
for _ in range(3): step()
torch.cuda.synchronize(); t = time.time()
for _ in range(10): g = step()
torch.cuda.synchronize()
print("nsplit %d: %.2f ms per step (fwd+bwd only), %.0f sim-steps/s" % (nsplit, (time.time() - t) / 10 * 1e3, B * ms / ((time.time() - t) / 10)))
