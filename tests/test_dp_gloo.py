"""N>1 path on CPU: world_size-2 gloo run of the data-parallel step (sol_amd.dist.DPStep).
The compute plugged in is the oracle (test infrastructure); what is tested is the product's
sharding + all-reduce(SUM) + identical-update logic: 2-rank result == single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sol_oracle as o
    torch.set_default_dtype(torch.float64)
    B, Y, X, ms = 4, 16, 8, 2
    g = o.geometry(Y, X)
    d, vy, vx = o.synthetic_state(B, Y, X, 1234)
    re = torch.tensor(o.RE_TRAIN[:B])
    gts = [o.synthetic_state(B, Y, X, 4321 + i, project_it=False) for i in range(ms)]
    return o, g, d, vy, vx, re, gts


def _make_step(o, g, params, m, v, state, group=None, use_flat=False):
    """use_flat: the gradient lives in the first n elements of a persistent n + 1 buffer handed to DPStep (the trainers'
    layout: the exchange then needs no packing copy); otherwise DPStep packs [grads | loss] itself."""
    import sol_amd
    offs = np.concatenate([[0], np.cumsum([p.numel() for p in params])])
    n = int(offs[-1])
    flat = torch.zeros(n + 1, dtype=torch.float64) if use_flat else None

    def fwd_bwd(d, vy, vx, re, gy, gx):
        ps = [p.detach().clone().requires_grad_(True) for p in params]
        loss = o.unrolled_loss(ps, d, vy, vx, re, list(gy), list(gx), g, (0.2, 0.25), o.STD_RE)
        loss.backward()
        grads = torch.cat([p.grad.reshape(-1) for p in ps])
        if use_flat:
            flat[:n].copy_(grads)
            grads = flat[:n]
        return loss.detach(), grads

    def apply(flat, lr):
        state["t"] += 1
        grads = [flat[offs[k]:offs[k + 1]].reshape(params[k].shape) for k in range(len(params))]
        p2, m2, v2 = o.adam_tf(params, grads, m, v, state["t"], lr)
        for a, b in zip(params + m + v, p2 + m2 + v2):
            a.copy_(b)

    return sol_amd.dist.DPStep(fwd_bwd, apply, group=group, flat=flat)


def _worker(rank, world, port, out, use_flat=False):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    o, g, d, vy, vx, re, gts = _problem()
    import sol_amd
    r, w, _ = sol_amd.dist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = sol_amd.dist.shard_range(d.shape[0], rank, world)
    params = [p.clone() for p in o.init_params(0)]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    step = _make_step(o, g, params, m, v, {"t": 0}, use_flat=use_flat)
    gy = torch.stack([s[1][lo:hi] for s in gts])
    gx = torch.stack([s[2][lo:hi] for s in gts])
    # count every collective the step issues (exactly ONE all-reduce per training step)
    calls = {"all_reduce": 0, "other": 0}
    real = {}
    for name in ("all_reduce", "broadcast", "all_gather", "reduce", "all_gather_into_tensor", "reduce_scatter_tensor"):
        real[name] = getattr(torch.distributed, name)

        def counted(*a, _n=name, **k):
            calls["all_reduce" if _n == "all_reduce" else "other"] += 1
            return real[_n](*a, **k)
        setattr(torch.distributed, name, counted)
    losses = []
    for _ in range(2):
        losses.append(float(step(d[lo:hi], vy[lo:hi], vx[lo:hi], re[lo:hi], gy, gx, lr=1e-4)))
    for name, fn in real.items():
        setattr(torch.distributed, name, fn)
    torch.save({"loss": losses, "params": torch.cat([p.reshape(-1) for p in params]), "calls": dict(calls),
                "dp_collectives": step.collectives}, os.path.join(out, "r%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("use_flat", [False, True])
def test_two_rank_gloo_equals_single_process(tmp_path, use_flat):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), use_flat), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    # ONE collective per training step: the loss rides behind the gradient
    for r in (r0, r1):
        assert r["calls"] == {"all_reduce": 2, "other": 0} and r["dp_collectives"] == 2, r["calls"]
    # every rank holds bit-identical weights after the identical Adam update
    assert torch.equal(r0["params"], r1["params"]) and r0["loss"] == r1["loss"]
    # and they equal the single-process large-batch step (loss is a batch SUM -> grads add up)
    o, g, d, vy, vx, re, gts = _problem()
    params = [p.clone() for p in o.init_params(0)]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    step = _make_step(o, g, params, m, v, {"t": 0})
    gy = torch.stack([s[1] for s in gts])
    gx = torch.stack([s[2] for s in gts])
    losses = [float(step(d, vy, vx, re, gy, gx, lr=1e-4)) for _ in range(2)]
    ref = torch.cat([p.reshape(-1) for p in params])
    assert np.allclose(losses, r0["loss"], rtol=1e-10)
    assert float((ref - r0["params"]).abs().max()) < 1e-9
