"""Which torch operations leave non-kernel nodes in a captured hipGraph (ROCm 7.2, torch 2.10)?  One capture per operation, census by
sol_graph_census.  Used to make the torch-composed trainers' graphs kernel-only (sol_graph_check refuses memset / memcpy nodes).
    python tools/graph_census.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import sol_amd  # noqa: F401
from sol_amd import _lib

dev = "cuda"
x = torch.randn(64, 1024, device=dev)
xs = torch.randn(4, device=dev)
y = torch.empty_like(x)
big = torch.randn(1 << 20, device=dev)
p = torch.randn(1000, device=dev, requires_grad=True)
idx = torch.zeros(5, dtype=torch.int32, device=dev)


def census(fn):
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()                        # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with _lib.no_gc_during_capture(), torch.cuda.graph(g):
        fn()
    c = _lib.graph_census(g.raw_cuda_graph())
    g.reset()
    return c


def accgrad():
    p.grad = None
    (p * 2.0).sum().backward()


def accgrad_twice():
    p.grad = None
    (p * 2.0).sum().backward()
    (p * 3.0).sum().backward()


cases = {
    "torch.zeros(64,1024)": lambda: torch.zeros(64, 1024, device=dev),
    "torch.zeros(11,256,int32)": lambda: torch.zeros(11, 256, dtype=torch.int32, device=dev),
    "zeros_like": lambda: torch.zeros_like(x),
    "x.zero_()": lambda: y.zero_(),
    "x.fill_(0)": lambda: y.fill_(0.0),
    "torch.ones_like": lambda: torch.ones_like(x),
    "full_like": lambda: torch.full_like(x, 0.3),
    "clone (contiguous)": lambda: x.clone(),
    "copy_ (contiguous, same dtype)": lambda: y.copy_(x),
    "copy_ (strided source)": lambda: y[:, :512].copy_(x[:, 512:]),
    "copy_ scalar": lambda: xs[0:1].copy_(xs[1:2]),
    "contiguous() of a slice": lambda: x[:, :512].contiguous(),
    "torch.mul(x, 1, out=y)": lambda: torch.mul(x, 1.0, out=y),
    "pad": lambda: torch.nn.functional.pad(x, (0, 1)),
    "cat": lambda: torch.cat([x, x], dim=0),
    "stack": lambda: torch.stack([x, x], dim=-1),
    "sum of 16": lambda: xs.sum(),
    "sum of 64K": lambda: x.sum(),
    "sum of 1M": lambda: big.sum(),
    "where": lambda: torch.where(x > 0, x, 0.3 * x),
    "x * y (alloc)": lambda: x * x,
    "AccumulateGrad (first)": accgrad,
    "AccumulateGrad (second)": accgrad_twice,
    "div by tensor": lambda: x / xs[0],
    "expand + cat": lambda: torch.cat([x.reshape(64, 1024, 1), xs[:1].reshape(1, 1, 1).expand(64, 1024, 1)], dim=-1),
}
for name, fn in cases.items():
    try:
        print("%-34s %s" % (name, census(fn)))
    except Exception as e:      # noqa: BLE001
        print("%-34s FAILED: %s" % (name, str(e).splitlines()[0][:120]))
