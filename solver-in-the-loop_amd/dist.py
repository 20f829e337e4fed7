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


AFFINITY = None        # what bind_cpu_affinity did for this process (bench.py prints it)


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(device_index, sysfs="/sys"):
    """NUMA node of the GPU `device_index` (torch's numbering) from /sys/class/drm/card*/device/numa_node, matched by PCI address;
    None when sysfs has no answer (no amdgpu cards visible, node -1 = the platform does not say)."""
    import glob
    try:
        pr = torch.cuda.get_device_properties(device_index)
        want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        return None
    for c in sorted(glob.glob(os.path.join(sysfs, "class/drm/card[0-9]*/device"))):
        if want in os.path.realpath(c):
            try:
                with open(os.path.join(c, "numa_node")) as f:
                    n = int(f.read().strip())
                return n if n >= 0 else None
            except (OSError, ValueError):
                return None
    return None


def bind_cpu_affinity(device_index, local_rank=0, local_world=1, sysfs="/sys"):
    """Pin this process to the CPUs of its GPU's NUMA node (os.sched_setaffinity): with a 12 ms step and a host-issued all-reduce + Adam
    between two graph replays, a rank whose launcher thread migrates to the other socket shows up as rank skew.
    Ranks that share a node split its CPUs evenly among themselves (local_rank / local_world).  Does nothing -- and says so -- when the
    platform gives no NUMA node, when the intersection with the inherited mask is empty, or when SOL_NO_AFFINITY is set.
    Returns a record {"numa_node", "cpus", "bound", "why"}; also kept in dist.AFFINITY."""
    global AFFINITY
    rec = {"numa_node": None, "cpus": None, "bound": False, "why": "", "inherited": None}
    try:
        if hasattr(os, "sched_getaffinity"):
            rec["inherited"] = sorted(os.sched_getaffinity(0))      # (bench.py's CPU baseline puts this mask back for its host threads)
        if os.environ.get("SOL_NO_AFFINITY"):
            rec["why"] = "SOL_NO_AFFINITY set"
        elif not hasattr(os, "sched_setaffinity"):
            rec["why"] = "no sched_setaffinity on this platform"
        else:
            node = gpu_numa_node(device_index, sysfs)
            rec["numa_node"] = node
            if node is None:
                rec["why"] = "sysfs gives no NUMA node for device %d" % device_index
            else:
                with open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)) as f:
                    cpus = sorted(_parse_cpulist(f.read()) & set(os.sched_getaffinity(0)))
                if local_world > 1 and len(cpus) >= 2 * local_world:
                    per = len(cpus) // local_world
                    cpus = cpus[(local_rank % local_world) * per:(local_rank % local_world + 1) * per]
                if not cpus:
                    rec["why"] = "the node's CPUs are outside the inherited affinity mask"
                else:
                    os.sched_setaffinity(0, cpus)
                    rec.update(cpus="%d-%d (%d)" % (cpus[0], cpus[-1], len(cpus)), bound=True)
    except Exception as e:          # never let a binding problem stop a run
        rec["why"] = "failed: %s" % e
    AFFINITY = rec
    return rec


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  No-op for single-process runs (except the CPU binding of a GPU process)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available() and AFFINITY is None:
        ndev = torch.cuda.device_count()
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        # ranks that share a NUMA node split it; the split is by local rank among ALL local ranks (a conservative partition: at most
        # local_world slices per node), only when there is more than one rank
        bind_cpu_affinity(local % ndev, local, lw if world > 1 else 1)
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


class SolComm:
    """RCCL communicator owned by libsol_hip.so (sol_comm_init / sol_allreduce_grads, include/sol_hip.h): the exchange
    step of the path without torch in the data path.  The 128-byte unique id travels over the already initialised
    torch.distributed group (any backend).  Must be created with this rank's device current."""

    def __init__(self, group=None):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        lib = _lib.load()
        world = world_size(group)
        rank = dist.get_rank(group) if world > 1 else 0
        buf = C.create_string_buffer(128)
        if rank == 0:
            _lib.check(lib.sol_comm_unique_id(buf))
        ids = [buf.raw]
        if world > 1:
            dist.broadcast_object_list(ids, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self.handle = C.c_void_p()
        _lib.check(lib.sol_comm_init(ids[0], world, rank, C.byref(self.handle)))
        self.world, self.rank = world, rank

    def allreduce_sum_(self, flat):
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        self._lib.check(self._lib.load().sol_allreduce_grads(self.handle, self._lib.stream(), self._lib.ptr(flat), flat.numel()))
        return flat

    def close(self):
        if self.handle:
            self._lib.load().sol_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DPStep:
    """train step = local fwd_bwd -> ONE all-reduce(SUM) of [grads | loss] -> identical optimizer update on every rank
    (one exchange per training step).  `fwd_bwd(*batch) -> (loss_tensor, flat_grads)`,
    `apply(grads, lr)`.

    The scalar loss travels in the slot behind the gradient (count n + 1) instead of a collective of its own.  `flat`:
    optional persistent buffer of n + 1 elements whose first n elements ARE the gradient tensor fwd_bwd returns (the
    trainers allocate their gradient that way: no copy at all); without it the step packs grads and loss into an internal
    buffer (one copy of the gradient each way).  `collectives` counts the all-reduces issued so far."""

    def __init__(self, fwd_bwd, apply, group=None, comm=None, flat=None):
        """comm: optional SolComm -- the all-reduce then goes through sol_allreduce_grads (the library's own RCCL
        communicator) instead of torch.distributed.all_reduce; both are RCCL on the GPU."""
        self.fwd_bwd, self.apply, self.group, self.comm, self.flat = fwd_bwd, apply, group, comm, flat
        self.collectives = 0
        self._pack = None

    def _exchange(self, loss, grads):
        n = grads.numel()
        own = self.flat is not None and self.flat.numel() == n + 1 and self.flat.data_ptr() == grads.data_ptr()
        if own:
            buf = self.flat
        else:
            if self._pack is None or self._pack.numel() != n + 1 or self._pack.dtype != grads.dtype or self._pack.device != grads.device:
                self._pack = torch.empty(n + 1, dtype=grads.dtype, device=grads.device)
            buf = self._pack
            buf[:n].copy_(grads.reshape(-1))
        buf[n:].copy_(loss.detach().reshape(1).to(buf.dtype))
        if self.comm is not None:
            self.comm.allreduce_sum_(buf)
        else:
            allreduce_sum_(buf, self.group)
        self.collectives += 1
        if not own:
            grads.reshape(-1).copy_(buf[:n])
        return buf[n].clone()

    def __call__(self, *batch, lr):
        loss, grads = self.fwd_bwd(*batch)
        if world_size(self.group) > 1:
            loss = self._exchange(loss, grads)
        self.apply(grads, lr)
        return loss
