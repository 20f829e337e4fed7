import os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import sol_amd
from sol_amd import ops, _lib, synthetic
import test_gpu_determinism as T
import itertools
cases = {"vy40": lambda vy, vx: vy.__setitem__((1, slice(60, 70), slice(20, 30)), 40.0),
         "vy-17": lambda vy, vx: vy.__setitem__((0, slice(100, 110), slice(5, 50)), -17.0),
         "vy-17b": lambda vy, vx: vy.__setitem__((0, slice(70, 80), slice(5, 50)), -17.0),
         "vx30": lambda vy, vx: vx.__setitem__((0, slice(30, 40), slice(10, 20)), 30.0),
         "vx-55": lambda vy, vx: vx.__setitem__((1, slice(5, 9), slice(None)), -55.0),
         "one": lambda vy, vx: vy.__setitem__((1, 66, 25), 40.0)}
for name, fn in cases.items():
    out = {}
    for bands in (0, 1):
        _lib.set_option("fwd_bands", bands)
        tr, (d, vy, vx, re, gy, gx) = T._trainer2d(2, 128, 64, 1, False)
        vy, vx = vy.clone(), vx.clone()
        fn(vy, vx)
        tr.grads.zero_()
        tr.fwd_bwd(d, vy, vx, re, gy, gx, want_final=True)
        torch.cuda.synchronize()
        out[bands] = [t.clone() for t in tr.final]
    print(name, [int((a.view(torch.int32) != b.view(torch.int32)).sum()) for a, b in zip(out[0], out[1])], [float((a - b).abs().max()) for a, b in zip(out[0], out[1])], float(out[0][1].abs().max()))
