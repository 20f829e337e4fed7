"""Do two HIP streams overlap kernels here?  solver step (6 workgroups, ~150 us) on stream A,
conv (256 workgroups, ~28 us each, x6) on stream B."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sol_amd
from sol_amd import ops, synthetic

B, Y, X = 6, 128, 64
dom = sol_amd.Domain([Y, X], box=sol_amd.box[0:200, 0:100])
flow = sol_amd.KarmanFlow()
active, inflow = flow.scene_arrays(dom)
bcv, bcm = sol_amd.velocity_bc_masks(Y, X)
masks = ops.SceneMasks(active, inflow, bcv.reshape(Y + 1, X), bcm.reshape(Y + 1, X))
f = lambda t: t.to(device="cuda", dtype=torch.float32).contiguous()
d0, vy0, vx0 = (f(t) for t in synthetic.state(B, Y, X, 1234))
re = f(synthetic.reynolds(B))
cfg = ops.karman_cfg(B, Y, X, dom.dx[1], masks=masks)
x = torch.randn(B, Y, X, 32, device="cuda")
w = torch.randn(5, 5, 32, 32, device="cuda") * 0.05
packed = ops._pack(w, 32, 32, ops.CONV_FWD)
bias = torch.zeros(32, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run(mode, n=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        if mode in ("step", "both_serial"):
            ops.karman_step(d0, vy0, vx0, re, cfg, masks)
        if mode in ("conv", "both_serial"):
            for _ in range(6):
                ops.conv5x5_raw(x, packed, bias, None, None, 32, ops.EPI_LRELU, 0.3)
        if mode == "both_streams":
            sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(sa):
                ops.karman_step(d0, vy0, vx0, re, cfg, masks)
            with torch.cuda.stream(sb):
                for _ in range(6):
                    ops.conv5x5_raw(x, packed, bias, None, None, 32, ops.EPI_LRELU, 0.3)
            torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for m in ("step", "conv", "both_serial", "both_streams"):
    run(m, 3)
    print("%-13s %.1f us per iteration" % (m, run(m)))
