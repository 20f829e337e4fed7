"""Data-parallel plumbing: one process per GPU, torch.distributed (backend 'nccl' = RCCL over
xGMI on ROCm; 'gloo' in the CPU test-suite).

The reference is single-device (`--gpu` only sets CUDA_VISIBLE_DEVICES,
/root/reference/karman-2d/karman_train.py:22,49); sharding independent simulations over the
GPUs of a node is the new capability asked for by BASELINE.json.  The path has exactly one
exchange step per training step: all-reduce(SUM) of the flat CNN gradient (1.04 MB), because
the loss is a SUM over the batch axis (tf.nn.l2_loss, karman_train.py:430).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  No-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("SOL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def shard_range(n_global, rank, world):
    """Contiguous shard of `n_global` simulations for `rank` (n_global % world == 0)."""
    if n_global % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (n_global, world))
    per = n_global // world
    return rank * per, (rank + 1) * per


def allreduce_sum_(flat, group=None):
    """In-place SUM all-reduce of one flat buffer (no bucketing: the whole gradient is 1 MB,
    latency bound; it cannot overlap with backward because the weights are shared by all
    unrolled steps and the gradient is complete only at the end of the reverse sweep)."""
    if world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class DPStep:
    """train step = local fwd_bwd -> all-reduce(SUM) grads (+ loss) -> identical optimizer
    update on every rank.  `fwd_bwd(*batch) -> (loss_tensor, flat_grads)`, `apply(grads, lr)`."""

    def __init__(self, fwd_bwd, apply, group=None):
        self.fwd_bwd, self.apply, self.group = fwd_bwd, apply, group

    def __call__(self, *batch, lr):
        loss, grads = self.fwd_bwd(*batch)
        if world_size(self.group) > 1:
            allreduce_sum_(grads, self.group)
            loss = loss.clone()
            allreduce_sum_(loss, self.group)
        self.apply(grads, lr)
        return loss
