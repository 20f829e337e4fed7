import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sol_amd
from sol_amd import karman3d as k3, _lib
from sol_amd._lib import ptr, stream, check
lib = _lib.load()
dev = "cuda"
gen = torch.Generator().manual_seed(0)
for (B, D, H, W) in ((1, 8, 64, 64), (1, 64, 64, 64), (1, 128, 64, 64)):
    x = torch.randn(B, D, H, W, 32, generator=gen, dtype=torch.float32).to(dev)
    dz = (torch.randn(B, D, H, W, 32, generator=gen, dtype=torch.float32) * 0.01).to(dev)
    outs = []
    for use in (True, False):
        part = torch.empty(lib.sol_conv3d_bwd_weight_ws_floats(B, D, H, W, 32, 32), dtype=torch.float32, device=dev)
        dW = torch.empty(5, 5, 5, 32, 32, dtype=torch.float32, device=dev)
        db = torch.empty(32, dtype=torch.float32, device=dev); sc = torch.empty(32, dtype=torch.float32, device=dev)
        xm, zm = (k3._absmax(x), k3._absmax(dz)) if use else (None, None)
        check(lib.sol_conv3d_bwd_weight(stream(), ptr(x), ptr(dz), ptr(xm), ptr(zm),
                                        ptr(part), ptr(dW), ptr(db), ptr(sc), B, D, H, W, 32, 32, 32, 32))
        torch.cuda.synchronize()
        outs.append(dW)
        print((B, D, H, W), "absmax" if use else "bf16x6", "finite", bool(torch.isfinite(dW).all()), "per-slice nonfinite", [int((~torch.isfinite(dW[k])).sum()) for k in range(5)],
              "norm", float(dW[torch.isfinite(dW)].norm()))
    a, b = outs
    m = torch.isfinite(a) & torch.isfinite(b)
    print("  rel diff fp16 vs bf16x6 (finite part)", float((a[m] - b[m]).norm() / b[m].norm()))
