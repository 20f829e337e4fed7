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


def _make_step(o, g, params, m, v, state, group=None):
    import sol_amd
    offs = np.concatenate([[0], np.cumsum([p.numel() for p in params])])

    def fwd_bwd(d, vy, vx, re, gy, gx):
        ps = [p.detach().clone().requires_grad_(True) for p in params]
        loss = o.unrolled_loss(ps, d, vy, vx, re, list(gy), list(gx), g, (0.2, 0.25), o.STD_RE)
        loss.backward()
        return loss.detach(), torch.cat([p.grad.reshape(-1) for p in ps])

    def apply(flat, lr):
        state["t"] += 1
        grads = [flat[offs[k]:offs[k + 1]].reshape(params[k].shape) for k in range(len(params))]
        p2, m2, v2 = o.adam_tf(params, grads, m, v, state["t"], lr)
        for a, b in zip(params + m + v, p2 + m2 + v2):
            a.copy_(b)

    return sol_amd.dist.DPStep(fwd_bwd, apply, group=group)


def _worker(rank, world, port, out):
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
    step = _make_step(o, g, params, m, v, {"t": 0})
    gy = torch.stack([s[1][lo:hi] for s in gts])
    gx = torch.stack([s[2][lo:hi] for s in gts])
    losses = []
    for _ in range(2):
        losses.append(float(step(d[lo:hi], vy[lo:hi], vx[lo:hi], re[lo:hi], gy, gx, lr=1e-4)))
    torch.save({"loss": losses, "params": torch.cat([p.reshape(-1) for p in params])}, os.path.join(out, "r%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gloo_equals_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
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
