#!/usr/bin/env python
"""bench.py -- SOL-32 karman-2d 128x64 training-step throughput on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU over RCCL.  Under torch.distributed.run (RANK/WORLD_SIZE set) the process is a
  rank; called bare as `python bench.py --gpus N` it re-executes itself under torch.distributed.run with
  N ranks on 127.0.0.1.  Rank 0 prints ONE JSON line.
  A "step" = one full training step (msteps=32 unrolled solver+CNN forward, loss, reverse sweep, gradient
  all-reduce, TF-Adam) on a synthetic batch of 6 simulations per GPU (BASELINE.json configs[2]);
  value = sim-steps/s of the whole job = N*B*msteps*K / time.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_MFMA16_TF = 2500.0        # dense 16-bit MFMA peak
PEAK_MFMA32_TF = 157.3         # fp32 matrix peak
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")   # written by tools/pmc_summary.py --json (with the library's source hash)
PRECISION_FILE = os.path.join(ROOT, "profiles", "r06_precision_at_c3.json")   # written by tests/test_gpu_parity.py::test_sol32_workload_error_of_each_conv_arithmetic_against_float64
RANK_SKEW_LIMIT = 0.25         # N > 1: the line is flagged "valid": false when the slowest rank's timed region is this much longer than the fastest one's


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--msteps", type=int, default=32)
    p.add_argument("--res", type=int, default=64, help="cells in x (Y = 2*res)")
    p.add_argument("--batch", type=int, default=6, help="simulations per GPU")
    p.add_argument("--lr", type=float, default=1e-6,
                   help="Adam learning rate of the timed steps (the reference's 1e-4 makes THIS synthetic workload diverge "
                        "within ~10 steps: see workload_deviations in the JSON line)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="skip the strict-fp32 leg, the 64x32 recipe, the roll-out and the full-chip solver launch")
    p.add_argument("--all-legs", action="store_true", help="N > 1: rank 0 also runs the single-GPU side legs (strict fp32, bf16x6, recipe, roll-out, karman-3d) "
                   "while the other ranks wait; by default an N > 1 run spends its lease on the data-parallel measurement only")
    p.add_argument("--no-graph", action="store_true", help="launch the ~1000 kernels of a step eagerly instead of replaying the hipGraph")
    p.add_argument("--k3d-dp", action="store_true", help="N > 1: also time the karman-3d SOL-16 step data parallel (one simulation per rank, BASELINE "
                                                         "configs[4]); off by default: a second collective path must not be able to hang the contract line")
    p.add_argument("--precision", default="split", choices=["split", "bf16x6", "fp32"], help="conv arithmetic of the timed steps")
    p.add_argument("--cpu-msteps", type=int, default=2, help="msteps of the bounded CPU-baseline sample")
    p.add_argument("--no-cpu-sol32", action="store_true", help="skip the ONE CPU training step at the full SOL-<msteps> depth that calibrates the bounded sample")
    p.add_argument("--prewarm", type=int, default=40,
                   help="untimed training steps BEFORE the W warm-up steps (clock / power-state ramp of a freshly leased GPU: ~0.5 s of the "
                        "workload itself, so that the K timed steps do not start on a cold clock); reported in the line as pre_warmup_steps")
    p.add_argument("--no-device-state", action="store_true", help="skip the clock / power samples around the timed region (device_state_before / _after)")
    p.add_argument("--comm", default="torch", choices=["torch", "lib"],
                   help="N > 1: the gradient all-reduce through torch.distributed.all_reduce (RCCL via PyTorch) or through the library's own "
                        "RCCL communicator (sol_allreduce_grads, csrc/comm.hip; needs one device per rank)")
    return p.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def cpu_baseline(args, Y, X, B):
    """Oracle (CPU restatement of the PhiFlow-1.5.1 algorithm, NOT TF-PhiFlow) timed on the host
    cores on a bounded sample: one fp32 training step (fwd + autograd bwd) of SOL-<cpu_msteps>."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sol_oracle as o
    try:        # the GPU process was pinned to its GPU's NUMA node (dist.bind_cpu_affinity): the host baseline gets the whole inherited mask back
        from sol_amd import dist as _dist
        if _dist.AFFINITY and _dist.AFFINITY.get("bound") and _dist.AFFINITY.get("inherited"):
            os.sched_setaffinity(0, _dist.AFFINITY["inherited"])
    except Exception:
        pass
    host = os.cpu_count() or 1
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    ms = args.cpu_msteps
    dt = torch.float32
    g = o.geometry(Y, X)
    d, vy, vx = o.synthetic_state(B, Y, X, 1234, dtype=dt)
    re = torch.tensor([o.RE_TRAIN[i % 6] for i in range(B)], dtype=dt)
    gts = [o.synthetic_state(B, Y, X, 4321 + i, dtype=dt, project_it=False) for i in range(ms)]
    params = [p.requires_grad_(True) for p in o.init_params(0, dtype=dt)]

    def one_step(m):
        loss = o.unrolled_loss(params, d, vy, vx, re, [s[1] for s in gts[:m]], [s[2] for s in gts[:m]], g, (0.2, 0.2), o.STD_RE)
        loss.backward()

    # n = os.cpu_count() of the host.  The 128x64 convolutions of this restatement do not scale past ~16 threads
    # (more threads are SLOWER on most hosts), so min(n, 16) and min(n, 32) are tried on two steps each and the faster count is used: the
    # baseline is the best the host does, and the line says which counts were tried and what the host has.
    # (the trial is bounded: more than 32 threads are never tried -- on a 256-thread host ONE SOL-2 step of these small convolutions takes
    #  166 s with 256 torch threads against 0.16 s with 16, measured on the round-5 GPU box; that trial alone ran for seven minutes)
    tried = {}
    for n in sorted({min(host, 16), min(host, 32)}):
        torch.set_num_threads(n)
        one_step(1)                   # warm-up: LU factorisation, oneDNN primitive caches
        t0 = time.time()
        one_step(ms)
        one_step(ms)
        tried[n] = (time.time() - t0) / 2
    cores = min(tried, key=tried.get)
    torch.set_num_threads(cores)
    reps, t0 = 0, time.time()
    while True:
        one_step(ms)
        reps += 1
        sec = time.time() - t0
        if sec > 12.0 or reps >= 200:
            break
    out = {"value": reps * B * ms / sec, "unit": "sim-steps/s", "cores": cores, "threads_used": cores, "host_cpu_count": host, "cpu_model": model,
           "threads_tried_s_per_step": {str(k): round(v, 4) for k, v in tried.items()}, "kind": "port",
           "sample": "%d fp32 training steps of SOL-%d (fwd + autograd bwd, B=%d, %dx%d) = %d sim-steps in %.1f s on %d "
                     "threads; torch-CPU restatement of the PhiFlow-1.5.1 algorithm (oracle/sol_oracle.py), not TF-PhiFlow"
                     % (reps, ms, B, Y, X, reps * B * ms, sec, cores)}
    if not args.no_cpu_sol32 and args.msteps > ms and out["value"] * 40.0 > B * args.msteps:
        # ONE step at the depth of the metric (SOL-32): is the shallow sample representative?  (skipped when it would take > ~40 s)
        full = args.msteps
        gts = gts + [o.synthetic_state(B, Y, X, 4321 + i, dtype=dt, project_it=False) for i in range(ms, full)]
        t0 = time.time()
        one_step(full)
        sec_full = time.time() - t0
        out["full_depth"] = {"msteps": full, "value": B * full / sec_full, "unit": "sim-steps/s", "seconds_per_training_step": sec_full,
                             "ratio_to_sample": (B * full / sec_full) / out["value"],
                             "note": "one fp32 training step at SOL-%d on the same threads (the autograd graph of the full unroll, 192 simulation steps)" % full}
    return out


class Workload:
    """Synthetic SOL-<msteps> training workload on one rank (same construction as oracle.bench_workload, with the HIP
    solver step instead of the oracle's)."""

    def __init__(self, sol_amd, dev, B, Y, X, ms, rank, precision="split", use_graph=True, comm=None):
        from sol_amd import ops, synthetic
        self.B, self.Y, self.X, self.ms = B, Y, X, ms
        dom = sol_amd.Domain([Y, X], box=sol_amd.box[0:200, 0:100])
        flow = sol_amd.KarmanFlow()
        active, inflow = flow.scene_arrays(dom)
        bcv, bcm = sol_amd.velocity_bc_masks(Y, X)
        self.masks = masks = ops.SceneMasks(active, inflow, bcv.reshape(Y + 1, X), bcm.reshape(Y + 1, X), dev)
        self.net = net = sol_amd.model_mars_moon(cin=3, cout=2, seed=0, device=dev)
        # Glorot-uniform weights of the reference architecture; the output layer is scaled by 0.01 so that the
        # untrained corrector starts as a small perturbation (a raw random corrector fed back through 32 solver
        # steps blows the roll-out up)
        with torch.no_grad():
            net.tensors()[22].mul_(0.01)
        self.std_v = (0.2, 0.2)
        self.dx = dom.dx[1]
        self.trainer = sol_amd.SolTrainer(net, masks, B, Y, X, ms, dom.dx[1], self.std_v, synthetic.STD_RE,
                                          use_graph=use_graph, conv_precision=precision, comm=comm)
        f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        d0, vy0, vx0 = (f(t) for t in synthetic.state(B, Y, X, 1234 + rank))
        self.re = re = f(synthetic.reynolds(B))
        # spin-up: one solver step makes the random start state divergence free / consistent
        self.cfgk = cfgk = ops.karman_cfg(B, Y, X, dom.dx[1], masks=masks)
        self.d0, self.vy0, self.vx0 = (t.detach().contiguous() for t in ops.karman_step(d0, vy0, vx0, re, cfgk, masks))
        # ground truth = plain solver roll-out of a slightly perturbed start state: the loss and its gradients are
        # non-zero but the targets are reachable (random frames as targets make Adam drive the corrector -- and with
        # it the 32-step unroll -- to blow up)
        _, py, px = synthetic.state(B, Y, X, 4321 + rank)
        gd, gy, gx = self.d0, self.vy0 + 0.05 * f(py - 1.0), self.vx0 + 0.05 * f(px)
        gts_y, gts_x = [], []
        for _ in range(ms):
            gd, gy, gx = (t.detach() for t in ops.karman_step(gd, gy, gx, re, cfgk, masks))
            gts_y.append(gy)
            gts_x.append(gx)
        self.gt_vy, self.gt_vx = torch.stack(gts_y).contiguous(), torch.stack(gts_x).contiguous()

    def step(self, lr, trainer=None):
        tr = trainer or self.trainer
        return tr.train_step(self.d0, self.vy0, self.vx0, self.re, self.gt_vy, self.gt_vx, lr, want_final=True)   # final state incl. the passive density


def timed_steps(wl, lr, steps, warmup, barrier, trainer=None, per_step=False):
    """W untimed steps, then K steps between barrier + synchronize pairs (wall clock).  per_step: one HIP event on the launch stream in
    front of every timed step and one behind the last (recorded asynchronously: no synchronisation is added to the timed region) -> the
    K individual step durations, read after the closing barrier."""
    trace = []
    for _ in range(warmup):
        trace.append(float(wl.step(lr, trainer)))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if per_step else None
    stamps = None
    if per_step:
        import ctypes as C
        import sol_amd
        stamps = torch.zeros(steps + 1, 32, dtype=torch.int32, device=wl.d0.device)      # per XCD (s_memtime, s_memrealtime) in front of every step and behind the last
        lib, strm = sol_amd._lib.load(), sol_amd._lib.stream
        stamp = lambda i: sol_amd._lib.check(lib.sol_clock_stamp(strm(), C.c_void_p(stamps.data_ptr() + 128 * i)))
        stamp(0)                                                                           # (code object load outside the timed region)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        if evs:
            evs[i].record()
            stamp(i)
        loss = wl.step(lr, trainer)
    if evs:
        evs[steps].record()
        stamp(steps)
    barrier()
    sec = time.perf_counter() - t0
    if per_step:
        w = stamps.cpu().numpy().view("uint64").reshape(steps + 1, 8, 2).astype("float64")
        ok = [x for x in range(8) if all(w[i, x, 1] > 0 for i in range(steps + 1))]                      # XCDs that took every stamp
        mhz = lambda a, b, x: (w[b, x, 0] - w[a, x, 0]) / max(w[b, x, 1] - w[a, x, 1], 1.0) * 100.0
        if ok:
            per_xcd = [mhz(0, steps, x) for x in ok]
            clk = [sum(mhz(i, i + 1, x) for x in ok) / len(ok) for i in range(steps)]                      # MHz per step, mean over the XCDs
            CLOCK_DURING.update(mhz_mean=sum(per_xcd) / len(per_xcd), mhz_min_step=min(clk), mhz_max_step=max(clk), mhz_per_xcd=[round(v, 1) for v in per_xcd], xcds=ok)
        return sec, float(loss), trace, [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    return sec, float(loss), trace


CLOCK_DURING = {}      # average shader clock of the timed region (sol_clock_stamp brackets), filled by timed_steps(per_step=True)


def _quantile(v, q):
    v = sorted(v)
    if not v:
        return None
    x = q * (len(v) - 1)
    lo = int(math.floor(x))
    hi = min(lo + 1, len(v) - 1)
    return v[lo] + (v[hi] - v[lo]) * (x - lo)


def step_distribution(per_step_ms):
    """median / p10 / p90 / min / max of the K per-step durations (HIP events on the launch stream): what lets a reader tell a slow
    box (the whole distribution moves, and the device_state probe with it) from a 1 % kernel regression (the median moves, the probe does not)"""
    if not per_step_ms:
        return None
    return {"median": _quantile(per_step_ms, 0.5), "p10": _quantile(per_step_ms, 0.1), "p90": _quantile(per_step_ms, 0.9),
            "min": min(per_step_ms), "max": max(per_step_ms), "mean": sum(per_step_ms) / len(per_step_ms), "n": len(per_step_ms),
            "all": [round(v, 4) for v in per_step_ms],
            "note": "HIP events on the launch stream around every timed step (graph replay + all-reduce + Adam), recorded asynchronously; ms"}


def _sysfs_gpu(dev_index):
    """best-effort clock / power / temperature of the device from amdgpu's sysfs files (no root needed); {} when nothing is readable"""
    import glob
    out = {}
    try:
        want = None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        cards = [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk"))]
        pick = None
        for c in cards:
            if want and want in os.path.realpath(c):
                pick = c
        if pick is None and len(cards) == 1:
            pick = cards[0]
        if pick is None:
            return out
        for name, key in (("pp_dpm_sclk", "sclk_mhz"), ("pp_dpm_mclk", "mclk_mhz")):
            try:
                with open(os.path.join(pick, name)) as f:
                    for line in f:
                        if line.strip().endswith("*"):
                            out[key] = float("".join(ch for ch in line.split(":")[1] if ch.isdigit() or ch == "."))
            except (OSError, ValueError, IndexError):
                pass
        for hw in glob.glob(os.path.join(pick, "hwmon", "hwmon*")):
            for name, key, scale in (("power1_average", "power_w", 1e-6), ("power1_input", "power_w", 1e-6), ("temp1_input", "temp_c", 1e-3),
                                     ("freq1_input", "sclk_hwmon_mhz", 1e-6)):
                try:
                    with open(os.path.join(hw, name)) as f:
                        out.setdefault(key, float(f.read().strip()) * scale)
                except (OSError, ValueError):
                    pass
    except Exception:
        pass
    return out


_DISPATCH_GRAPH = None


def device_state(sol_amd, dev, iters=4000000):
    """Clock / power sample of the device: (a) ~90 ms of a dependent-MFMA chain on every SIMD (sol_clock_probe): ns per dependent
    v_mfma_f32_16x16x16_f16 follows the engine clock the box holds UNDER MATRIX LOAD and nothing else, and the ratio of the two device
    counters (s_memtime / s_memrealtime) is that clock in MHz; (b) amdgpu sysfs (sclk, power, temperature) where readable; (c) the clock
    the runtime advertises.  Taken before and after the timed region."""
    import ctypes as C
    out = {"iters": iters}
    try:
        buf = torch.zeros(8, dtype=torch.int32, device=dev)
        lib = sol_amd._lib.load()
        sol_amd._lib.check(lib.sol_clock_probe(sol_amd._lib.stream(), C.c_void_p(buf.data_ptr()), 1000))      # warm: code object load
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sol_amd._lib.check(lib.sol_clock_probe(sol_amd._lib.stream(), C.c_void_p(buf.data_ptr()), iters))
        torch.cuda.synchronize()
        out["probe_wall_ms"] = (time.perf_counter() - t0) * 1e3
        w = buf.cpu().numpy().view("uint64")
        out["ns_per_dependent_mfma"] = float(w[0]) * 10.0 / float(w[2])
        # s_memtime counts engine clocks on gfx950 (23.8 per 10 ns tick of s_memrealtime on a 2.4 GHz part): the ratio IS the shader clock held
        # during the probe, measured by the device itself
        out["shader_clock_mhz_measured"] = float(w[1]) / max(float(w[0]), 1.0) * 100.0
        out["cycles_per_dependent_mfma"] = out["ns_per_dependent_mfma"] * out["shader_clock_mhz_measured"] * 1e-3
    except Exception as e:
        out["probe_error"] = str(e)
    # (d) the cost of a launch boundary: a chain of dependent one-word copy kernels in a replayed graph -- us per node is what EVERY one of
    #     the ~920 launches of a step pays between the end of one kernel and the start of the next (command processor, not shader clock);
    # (e) device-to-device copy bandwidth of a 256 MB buffer (HBM + fabric clocks)
    try:
        global _DISPATCH_GRAPH
        lib = sol_amd._lib.load()
        if _DISPATCH_GRAPH is None:
            word = torch.zeros(64, dtype=torch.int32, device=dev)

            def chain():
                for i in range(400):
                    sol_amd._lib.check(lib.sol_copy_words(sol_amd._lib.stream(), C.c_void_p(word.data_ptr() + 4 * (i & 1)), C.c_void_p(word.data_ptr() + 8 + 4 * (i & 1)), 1))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                chain()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            _DISPATCH_GRAPH = (sol_amd._lib.capture_graph(chain, "bench.py dispatch probe"), word)
        g = _DISPATCH_GRAPH[0]
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        out["us_per_dependent_launch_in_graph"] = a.elapsed_time(b) * 1e3 / (5 * 400)
        big = torch.empty(2, 64 * 1024 * 1024, dtype=torch.float32, device=dev)
        for rep in range(2):
            a.record()
            for _ in range(4):
                sol_amd._lib.check(lib.sol_copy_words(sol_amd._lib.stream(), C.c_void_p(big[1].data_ptr()), C.c_void_p(big[0].data_ptr()), big[0].numel()))
            b.record()
            torch.cuda.synchronize()
        out["copy_GBps_256MB"] = 4 * 2 * big[0].numel() * 4 / (a.elapsed_time(b) * 1e-3) / 1e9
        # (f) dependent-load latency: one lane walking a 64 MB (memory-side cache) and a 512 MB (HBM) working set of the same zero-filled buffer
        big.zero_()
        res = torch.zeros(8, dtype=torch.int32, device=dev)
        for key, nlines in (("load_latency_ns_64MB", 1 << 19), ("load_latency_ns_512MB", 1 << 22)):
            sol_amd._lib.check(lib.sol_latency_probe(sol_amd._lib.stream(), C.c_void_p(big.data_ptr()), nlines, 20000, C.c_void_p(res.data_ptr())))
            torch.cuda.synchronize()
            w = res.cpu().numpy().view("uint64")
            out[key] = float(w[0]) * 10.0 / float(w[1])
        del big
    except Exception as e:
        out["dispatch_probe_error"] = str(e)
    out.update(_sysfs_gpu(dev.index if dev.index is not None else 0))
    try:
        out["advertised_max_clock_mhz"] = torch.cuda.get_device_properties(dev).clock_rate / 1e3
    except Exception:
        pass
    return out


def burgers_leg(sol_amd, dev, steps=30):
    """BASELINE configs[0] and the reference's two Burgers training recipes (burgers/Makefile:69-77: 32x32, `-b 5`, dt 0.1; NON = `-m 1`,
    SOL-04 = `-m 4`): one training step = unrolled msteps x [BurgersTest.step_with_f -> model_mars_moon(4 -> 2) correction -> l2 loss], reverse
    sweep, TF-Adam; captured once, replayed (sol_amd.BurgersTrainer).  Synthetic frames of the recipe's shape."""
    import numpy as np
    B, Y, X, dt = 5, 32, 32, 0.1
    dom = sol_amd.Domain([Y, X], box=sol_amd.box([32, 32]), boundaries=sol_amd.PERIODIC)
    rng = np.random.default_rng(0)
    out = {"workload": "burgers 32x32, batch 5, dt 0.1, model_mars_moon(4 -> 2), lr 1e-4 (burgers/Makefile:69-77)", "data": "synthetic"}
    for leg, ms in (("non_m1", 1), ("sol04_m4", 4)):
        try:
            velo = torch.as_tensor(0.3 * rng.standard_normal((ms + 1, B, Y + 1, X + 1, 2)).astype(np.float32), device=dev)
            forc = torch.as_tensor(0.1 * rng.standard_normal((ms, B, Y + 1, X + 1, 2)).astype(np.float32), device=dev)
            net = sol_amd.model_mars_moon(cin=4, cout=2, seed=0, device=dev)
            tr = sol_amd.BurgersTrainer(net, dom, B, ms, dt, (0.2, 0.2), (0.1, 0.1), use_graph=True)
            for _ in range(5):
                tr.train_step(velo, forc, 1e-4)
            torch.cuda.synchronize()
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            t0 = time.perf_counter()
            for i in range(steps):
                evs[i].record()
                loss = tr.train_step(velo, forc, 1e-4)
            evs[steps].record()
            torch.cuda.synchronize()
            sec = time.perf_counter() - t0
            per = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
            out[leg] = {"msteps": ms, "ms_per_step": sec / steps * 1e3, "ms_per_step_median": _quantile(per, 0.5), "sim_steps_per_s": B * ms * steps / sec,
                        "loss": float(loss), "finite": bool(math.isfinite(float(loss))), "schedule": getattr(tr, "schedule", "autograd")}
            del tr
        except Exception as e:
            out[leg] = {"error": str(e)}
    return out


def profile_kernels(wl, lr, trainer=None):
    """One EAGER forward + reverse sweep with a pair of HIP events around every kernel launch (on the launch stream):
    {kernel: {"calls", "avg_us", "total_us"}}.  Same kernels, same order and same (cold) operands as the replayed graph."""
    from sol_amd import _lib
    tr = trainer or wl.trainer
    with _lib.profile() as p:          # forward + reverse sweep only: no collective (this runs on rank 0 alone), no optimizer update
        tr.fwd_bwd(wl.d0, wl.vy0, wl.vx0, wl.re, wl.gt_vy, wl.gt_vx, want_final=True, eager=True)
    return {k.strip("()"): {"calls": c, "avg_us": t / max(c, 1), "total_us": t} for k, (c, t) in p.kernels.items()}


def karman3d_leg(sol_amd, dev, B=1, steps=8):
    """BASELINE configs[4] grid, 128 x 64 x 64: (a) forward roll-out of solver step + Conv3D correction -- ms per simulation
    step, the per-kernel table of one step (per-launch HIP events) and, for the stencil kernels, their ALGORITHMIC bytes per
    launch against the 8 TB/s HBM peak; (b) the SOL-16 training step (train_sol16).  Stencil traffic accounting:
      k3_diffuse      reads the 3 components + the 2 BC arrays of the flow component, writes 3 components
      k3_advect_*     reads 3 diffused components + density, writes 3 components + density
      k3_div          reads 3 components, writes the right-hand side
      k3_project      reads pressure + 3 components, writes 3 components + the 4-channel feature tensor"""
    from sol_amd import karman3d as k3, synthetic, _lib
    Y, X, Z = 128, 64, 64
    sc = k3.Scene3D(Y, X, Z, device=dev)
    net = k3.MarsMoon3D(device=dev)
    w = net.get_weights()
    w[22] = w[22] * 0.01
    net.set_weights(w)
    ro = k3.Karman3DRollout(net, sc, B, (0.2, 0.2, 0.2), synthetic.STD_RE)
    gen = torch.Generator().manual_seed(4)
    r = lambda *s: torch.randn(*s, generator=gen)
    st = (torch.rand(B, Y, X, Z, generator=gen).to(dev), (1.0 + 0.1 * r(B, Y + 1, X, Z)).to(dev), (0.1 * r(B, Y, X + 1, Z)).to(dev),
          (0.1 * r(B, Y, X, Z + 1)).to(dev))
    re = synthetic.reynolds(B).float().to(dev)
    st = ro.step(*st, re)                   # warm-up + spin-up
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    s2 = st
    for _ in range(steps):
        s2 = ro.step(*s2, re)
    b.record()
    torch.cuda.synchronize()
    ms_step = a.elapsed_time(b) / steps
    a.record()
    for _ in range(steps):
        ro.correction()
    b.record()
    torch.cuda.synchronize()
    ms_cnn = a.elapsed_time(b) / steps               # the twelve CNN launches back to back (warm caches)
    with _lib.profile() as p:
        ro.step(*st, re)
    N = Y * X * Z
    nV = [(Y + 1) * X * Z, Y * (X + 1) * Z, Y * X * (Z + 1)]
    alg = {"k3_diffuse": 4.0 * B * (2 * sum(nV) + 2 * nV[0]), "k3_advect_tile": 4.0 * B * (2 * sum(nV) + 2 * N), "k3_advect": 4.0 * B * (2 * sum(nV) + 2 * N),
           "k3_div": 4.0 * B * (sum(nV) + N), "k3_project": 4.0 * B * (N + 2 * sum(nV) + 4 * N)}
    kern = {}
    tot = sum(t for _, t in p.kernels.values())
    for k, (c, t) in sorted(p.kernels.items(), key=lambda kv: -kv[1][1]):
        nm = k.strip("()")
        e = {"calls": c, "avg_us": round(t / max(c, 1), 2), "share": round(t / tot, 4)}
        if nm in alg:
            e["algorithmic_GBps"] = alg[nm] / (t / c * 1e-6) / 1e9
            e["frac_of_hbm_peak"] = e["algorithmic_GBps"] / PEAK_HBM_GBS
        kern[nm] = e
    flop = 2.0 * 125 * (4 * 32 + 10 * 32 * 32 + 32 * 3) * B * N
    # roofline of the dominant 3-D kernel (the one-launch 32 -> 32 Conv3D): algorithmic fp32 FLOP / event time / the 16-bit matrix peak
    roof3 = None
    for nm, e in kern.items():
        if nm.startswith("k_conv3d_sb") and "<1" not in nm:
            fl = 2.0 * 125 * 32 * 32 * B * N
            roof3 = {"kernel": nm, "bound": "mfma", "achieved": fl / (e["avg_us"] * 1e-6) / 1e12, "peak": PEAK_MFMA16_TF, "unit": "TFLOP/s",
                     "launch_us": e["avg_us"], "algorithmic_fp32_flop_per_launch": fl}
            roof3["frac"] = roof3["achieved"] / roof3["peak"]
            roof3["mfma_pipe_busy"] = 3.0 * roof3["frac"]
            roof3["note"] = ("launch_us from per-launch HIP events with a synchronisation after every launch (cold operands); back to back the twelve "
                             "launches of a CNN pass take cnn_ms_back_to_back (tools/k3d_time.py: 279-292 us per 32 -> 32 launch)")
            break
    t_cnn = sum(t for k, (c, t) in p.kernels.items() if "conv" in k or k == "k3_fill")
    # SOL-16 training step (BASELINE configs[4]: forward unroll, loss, reverse sweep through the HIP adjoints, TF-Adam)
    train = None
    try:
        del ro
        ms3 = 16
        tr = k3.Karman3DTrainer(net, sc, B, ms3, (0.2, 0.2, 0.2), synthetic.STD_RE, use_graph=True)
        gts = []
        gs = st
        with torch.no_grad():
            sim = tr.sim
            for _ in range(ms3):                      # ground truth = plain solver roll-out of a perturbed start (as the 2-D workload)
                gs = sim.step(gs[0], gs[1], gs[2], gs[3], re)
                gts.append(tuple(t + 0.01 * torch.randn_like(t) for t in gs[1:]))
        torch.cuda.reset_peak_memory_stats()
        l0 = float(tr.train_step(*st, re, gts, lr=1e-7))
        torch.cuda.synchronize()
        nrep = 2
        t0 = time.perf_counter()
        for _ in range(nrep):
            l1 = float(tr.train_step(*st, re, gts, lr=1e-7))
        torch.cuda.synchronize()
        t_tr = (time.perf_counter() - t0) / nrep
        train = {"msteps": ms3, "ms_per_step": t_tr * 1e3, "sim_steps_per_s": B * ms3 / t_tr, "loss": l1, "loss_first": l0, "finite": bool(math.isfinite(l1)),
                 "peak_memory_GB": torch.cuda.max_memory_allocated() / 2 ** 30,
                 "conv_fp32_equiv_TFLOPs_of_step": 3.0 * flop * ms3 / t_tr / 1e12,
                 "note": "hand-written schedule over the C ABI (forward unroll + reverse sweep, no autograd graph), captured once into a kernel-nodes-only hipGraph and replayed + TF-Adam"}
    except Exception as e:
        train = {"error": str(e)}
    return {"train_sol16": train,"workload": "karman-3d %dx%dx%d, batch %d, forward roll-out (solver step + Conv3D(5) mars_moon correction; BASELINE configs[4] grid)" % (Y, X, Z, B),
            "ms_per_sim_step": ms_step, "sim_steps_per_s": B * 1e3 / ms_step, "finite": bool(torch.isfinite(s2[1]).all()),
            "cnn_fp32_equiv_TFLOPs": flop / (t_cnn * 1e-6) / 1e12, "solver_us": sum(t for k, (c, t) in p.kernels.items() if "conv" not in k and k not in ("k3_fill", "k3_correct")),
            "cnn_ms_back_to_back": ms_cnn, "cnn_fp32_equiv_TFLOPs_back_to_back": flop / (ms_cnn * 1e-3) / 1e12, "roofline": roof3, "kernels": kern}


def karman3d_dp_leg(sol_amd, dev, world, barrier):
    """karman-3d 128x64x64 SOL-16 training step, one simulation per rank, ONE all-reduce of [gradient | loss] per step (every rank calls this)."""
    from sol_amd import karman3d as k3, synthetic
    Y, X, Z, ms3 = 128, 64, 64, 16
    sc = k3.Scene3D(Y, X, Z, device=dev)
    net = k3.MarsMoon3D(device=dev)
    w = net.get_weights()
    w[22] = w[22] * 0.01
    net.set_weights(w)
    tr = k3.Karman3DTrainer(net, sc, 1, ms3, (0.2, 0.2, 0.2), synthetic.STD_RE, use_graph=True)
    gen = torch.Generator().manual_seed(1 + (torch.distributed.get_rank() if world > 1 else 0))
    f = lambda *s: torch.randn(*s, generator=gen)
    st = (torch.rand(1, Y, X, Z, generator=gen).to(dev), (1.0 + 0.1 * f(1, Y + 1, X, Z)).to(dev), (0.1 * f(1, Y, X + 1, Z)).to(dev), (0.1 * f(1, Y, X, Z + 1)).to(dev))
    re = synthetic.reynolds(1).float().to(dev)
    gts, gs = [], st
    with torch.no_grad():
        for _ in range(ms3):
            gs = tr.sim.step(gs[0], gs[1], gs[2], gs[3], re)
            gts.append(tuple(t + 0.01 * torch.randn_like(t) for t in gs[1:]))
    c0 = tr._dp.collectives
    l0 = float(tr.train_step(*st, re, gts, lr=1e-7))
    per_step = tr._dp.collectives - c0
    barrier()
    t0 = time.perf_counter()
    nrep = 3
    for _ in range(nrep):
        l1 = float(tr.train_step(*st, re, gts, lr=1e-7))
    barrier()
    t = torch.tensor([(time.perf_counter() - t0) / nrep], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    t = float(t.item())
    return {"workload": "karman-3d 128x64x64 SOL-16, 1 simulation per rank, %d rank(s)" % world, "ms_per_step": t * 1e3, "sim_steps_per_s": world * ms3 / t,
            "collectives_per_step": per_step, "allreduce_bytes": tr._flat.numel() * 4, "loss": l1, "loss_first": l0, "finite": bool(math.isfinite(l1))}


def load_traffic():
    """{'kernel|grid': {'FETCH_SIZE': KB, 'WRITE_SIZE': KB}} from the committed rocprofv3 --pmc summary -- but only when it was
    collected for THIS build: the file carries the content hash of the library sources it was taken at
    (sol_amd._build._source_hash()); counters of other kernels are not this run's traffic, the fields stay null then."""
    try:
        with open(TRAFFIC_FILE) as f:
            tab = json.load(f)
    except (OSError, ValueError):
        return {}
    from sol_amd import _build
    if tab.get("source_hash") != _build._source_hash():
        return {"stale": True, "collected_at_source_hash": tab.get("source_hash")}
    return tab


def load_precision():
    """the committed workload-level precision record (errors of the three convolution arithmetics against the float64 fixture at C3, written
    by the GPU test-suite); None when absent"""
    try:
        with open(PRECISION_FILE) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def precision_note():
    p = load_precision()
    if not p:
        return "not recorded for this tree (profiles/r06_precision_at_c3.json missing)"
    r, e = p["ratio_to_strict_fp32"]["split"], p["relative_l2_error"]["split"]
    return "%.2fx / %.2fx the strict trainer's error (%.1e / %.1e absolute; tolerances 1e-5 / 1e-4)" % (
        r["loss_steps"], r["gradient_every_16th"], e["loss_steps"], e["gradient_every_16th"])


def traffic_bytes(tab, kernel, grid):
    # (rocprofv3 prints defaulted template arguments, the launch macro's name does not: k_conv5x5_dx<3, 2, false> vs <3, 2>)
    kern = {k.replace(", false>", ">"): v for k, v in tab.get("kernels", {}).items()}
    e = kern.get("%s|%d" % (kernel, grid))
    if not e or "FETCH_SIZE" not in e or "WRITE_SIZE" not in e:
        return None
    return (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0      # gfx950: FETCH_SIZE under-counts 2x (MI355X_MICROARCH.md)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)
    import sol_amd
    from sol_amd import ops, synthetic
    sol_amd._lib.require_gpu()
    ndev = torch.cuda.device_count()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    backend = None
    if world_env > ndev:                 # more ranks than devices (debugging the N>1 path on a 1-GPU box): RCCL refuses shared devices
        backend = "gloo"
    rank, world, local = sol_amd.dist.init_from_env(backend)
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if world > 1 and world <= ndev and torch.distributed.get_backend() != "nccl":
        # one device per rank is available: a weak-scaling point measured over gloo (host staging) would say nothing about RCCL / xGMI
        raise SystemExit("bench.py: %d ranks on %d devices must run the nccl (RCCL) backend, got %r -- refusing to print a scaling point over a "
                         "fallback backend" % (world, ndev, torch.distributed.get_backend()))
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    X = args.res
    Y = 2 * X
    B, ms = args.batch, args.msteps
    comm = None
    if args.comm == "lib" and world > 1:
        if torch.distributed.get_backend() != "nccl":
            raise SystemExit("--comm lib needs one device per rank (RCCL refuses ranks that share a device); this run uses the %s backend"
                             % torch.distributed.get_backend())
        comm = sol_amd.dist.SolComm()
    wl = Workload(sol_amd, dev, B, Y, X, ms, rank, args.precision, use_graph=not args.no_graph, comm=comm)
    tr = wl.trainer

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(0, args.prewarm)):
        wl.step(args.lr)
    state_before = device_state(sol_amd, dev) if (rank == 0 and not args.no_device_state) else None
    sec, loss, trace, per_step_ms = timed_steps(wl, args.lr, args.steps, args.warmup, barrier, per_step=True)
    state_after = device_state(sol_amd, dev) if (rank == 0 and not args.no_device_state) else None
    tsec = torch.tensor([sec], dtype=torch.float64, device=dev)
    rank_ms = [sec / args.steps * 1e3]
    if world > 1:
        allsec = [torch.zeros_like(tsec) for _ in range(world)]
        torch.distributed.all_gather(allsec, tsec)
        rank_ms = [float(t.item()) / args.steps * 1e3 for t in allsec]
        torch.distributed.all_reduce(tsec, op=torch.distributed.ReduceOp.MAX)
        if rank == 0:
            print("bench.py: per-rank ms/step " + " ".join("%.3f" % v for v in rank_ms), file=sys.stderr, flush=True)
    # a rank that is throttled, shares its device or lost its binding makes the weak-scaling number that rank's, not the job's: the
    # line is still printed (an 8-GPU run always yields a number), flagged "valid": false
    rank_skew = max(rank_ms) / min(rank_ms) - 1.0
    valid = rank_skew <= RANK_SKEW_LIMIT
    if not valid and rank == 0:
        print("bench.py: WARNING rank skew %.0f %% (> %d %%): per-rank ms/step %s" % (rank_skew * 100, int(RANK_SKEW_LIMIT * 100), rank_ms), file=sys.stderr, flush=True)
    sec = float(tsec.item())
    if not math.isfinite(loss) or not all(math.isfinite(v) for v in trace):
        raise SystemExit("bench.py: non-finite loss in the timed run (warm-up trace %s, final %s): the measurement is invalid" % (trace, loss))
    ms_per_step = sec / args.steps * 1e3
    value = world * B * ms * args.steps / sec

    # ---- multi-GPU bookkeeping: all-reduce cost and replica consistency -------------------------
    dp = None
    if world > 1:
        ev = lambda: torch.cuda.Event(enable_timing=True)
        a, b = ev(), ev()
        g = tr._flat.clone()                  # what a training step exchanges: [gradient | loss], ONE collective
        c0 = tr._dp.collectives
        wl.step(args.lr)
        collectives_per_step = tr._dp.collectives - c0
        ar = comm.allreduce_sum_ if comm is not None else sol_amd.dist.allreduce_sum_
        for _ in range(3):
            ar(g)
        torch.cuda.synchronize()
        a.record()
        for _ in range(20):
            ar(g)
        b.record()
        torch.cuda.synchronize()
        w32 = wl.net.params.detach().view(torch.int32).to(torch.int64)
        sig = torch.stack([w32.sum(), (w32 * torch.arange(1, w32.numel() + 1, device=dev) % 1000003).sum()])
        sigs = [torch.zeros_like(sig) for _ in range(world)]
        torch.distributed.all_gather(sigs, sig)
        dp = {"backend": torch.distributed.get_backend(), "comm": "sol_allreduce_grads (library RCCL communicator)" if comm is not None else "torch.distributed.all_reduce",
              "allreduce_us": a.elapsed_time(b) / 20 * 1e3, "allreduce_bytes": g.numel() * 4,
              "collectives_per_step": collectives_per_step, "rank_ms_per_step": rank_ms, "rank_skew": rank_skew,
              "weights_bit_identical_across_ranks": bool(all(bool((s == sigs[0]).all()) for s in sigs))}
        if rank == 0:       # the one-line summary of every N > 1 run (stderr: the contract line stays the only line on stdout)
            print("bench.py: dp summary backend=%s comm=%s world=%d ms/step[min %.3f max %.3f] allreduce=%.1f us (%d B) collectives/step=%d bit_identical=%s"
                  % (dp["backend"], args.comm, world, min(rank_ms), max(rank_ms), dp["allreduce_us"], dp["allreduce_bytes"], collectives_per_step,
                     dp["weights_bit_identical_across_ranks"]), file=sys.stderr, flush=True)
        if not dp["weights_bit_identical_across_ranks"]:
            valid = False
            if rank == 0:
                print("bench.py: WARNING the replicas' weights diverged", file=sys.stderr, flush=True)

    k3d_dp = None
    if args.k3d_dp and world > 1:
        k3d_dp = karman3d_dp_leg(sol_amd, dev, world, barrier)
    out = None
    if rank == 0:
        N = Y * X
        Nf = (Y + 1) * X + Y * (X + 1)
        ev = lambda: torch.cuda.Event(enable_timing=True)

        def time_call(fn, reps):
            fn()
            torch.cuda.synchronize()
            a, b = ev(), ev()
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps * 1e-3

        traffic = load_traffic()
        fwd_b, bwd_b, kf_tr, kb_tr = tr.solver_algorithmic_bytes()
        direct = getattr(wl.masks, "direct", None) is not None
        prof = profile_kernels(wl, args.lr)
        tot_prof = sum(v["total_us"] for v in prof.values())

        def pick(*names):
            for n in names:
                if n in prof:
                    return n, prof[n]
            return None, None

        # (1) dominant kernel: the 32->32 conv (forward and backward-data launches share one kernel)
        conv_names = {"split": ("k_conv5x5_dx<3, 2>", "k_conv5x5_dx<1, 1, true>", "k_conv5x5_dx<1, 2>", "k_conv5x5_sb<2, 2>", "k_conv5x5_sb<2, 0>"), "bf16x6": ("k_conv5x5_sb<2, 0>",), "fp32": ("k_conv5x5_r3<2>",)}[args.precision]
        cname, cst = pick(*conv_names)
        flop_conv = 2.0 * 25 * 32 * 32 * B * N
        nprod = 3 if cname.startswith("k_conv5x5_dx") else {"k_conv5x5_sb<2, 2>": 3, "k_conv5x5_sb<2, 0>": 6}.get(cname, 1)
        roof_conv = None
        if cst:
            t_conv = cst["avg_us"] * 1e-6
            peak = PEAK_MFMA16_TF if nprod > 1 else PEAK_MFMA32_TF
            alg = flop_conv / t_conv / 1e12              # ALGORITHMIC fp32 FLOP of the convolution per launch / launch duration
            roof_conv = {"kernel": cname, "bound": "mfma", "achieved": alg, "peak": peak, "unit": "TFLOP/s", "frac": alg / peak,
                         "mfma_pipe_busy": nprod * alg / peak,
                         "traffic": traffic_bytes(traffic, cname, ((B * Y + 2) // 3) * max(1, X // 64) * (512 if cname.startswith("k_conv5x5_dx") else 768)),
                         "launch_us": cst["avg_us"], "launches_per_train_step": cst["calls"], "share_of_step_kernel_time": cst["total_us"] / tot_prof,
                         "algorithmic_fp32_flop_per_launch": flop_conv, "executed_mfma_flop_per_launch": nprod * flop_conv,
                         "frac_of_fp32_matrix_peak": alg / PEAK_MFMA32_TF,
                         "note": ("achieved = algorithmic fp32 FLOPs of the convolution (2*25*32*32 per output pixel) / the kernel's average duration "
                                  "inside an eager training step (per-launch HIP events on the launch stream); peak = dense peak of the pipe the kernel "
                                  "runs on (16-bit MFMA).  Each fp32 product is evaluated as %d exact 16-bit MFMA products with fp32 accumulation: "
                                  "mfma_pipe_busy = %d x frac is the occupancy of that pipe, frac_of_fp32_matrix_peak compares with v_mfma_f32_*_f32." % (nprod, nprod))
                                 if nprod > 1 else "fp32 MFMA (v_mfma_f32_16x16x4_f32); achieved = algorithmic fp32 FLOPs / average in-pipeline duration"}
        # (2) the fused advect+pressure step the north star names, as launched in the training graph at this batch size
        sname, sst = pick("k_karman_fwd_bands", "k_karman_fwd_dens", "k_karman_fwd")
        bands = sname == "k_karman_fwd_bands"
        solver_wgs = (40 * ((B + 7) // 8) + B) if bands else ((2 if sname and sname.endswith("dens") else 1) * B)      # workgroups of the launch (bands: 8 x 5 slots per 8 simulations, B of each 8 used, + B density workgroups)
        solver_cus = (5 * B + B) if bands else solver_wgs
        roof_solver = None
        if sst:
            bytes_step = fwd_b / ms            # algorithmic bytes of one forward launch (SURVEY 8d formula, measured k; 0 with the direct solver)
            t_s = sst["avg_us"] * 1e-6
            roof_solver = {"kernel": sname, "bound": "hbm", "achieved": bytes_step / t_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": bytes_step / t_s / (PEAK_HBM_GBS * 1e9), "traffic": traffic_bytes(traffic, sname, solver_wgs * 512),
                           "launch_us": sst["avg_us"], "launches_per_train_step": sst["calls"], "share_of_step_kernel_time": sst["total_us"] / tot_prof,
                           "cg_iters": kf_tr, "algorithmic_bytes_per_launch": bytes_step,
                           "accounting": "SURVEY 8d formula 4*(10*Nf + 9*N + 11*N*k) per sample-step with the MEASURED k of this solver (k = 0 for the direct solve)",
                           "reference_cg_bytes_note": {
                               "k_planning": 180, "algorithmic_bytes_per_launch": 4.0 * (10 * Nf + 9 * N + 11.0 * N * 180) * B,
                               "note": "byte count only: the traffic the reference's unpreconditioned CG (k ~ 180 at 128x64, SURVEY 8d worked example) "
                                       "would move for the same result.  This launch does not move it, so no fraction of peak is quoted for it"},
                           "pressure_solver": "direct (sine-transform diagonalisation + capacitance correction, no iteration)" if direct
                                              else "two-level preconditioned CG",
                           "workgroups_per_simulation": 5 if bands else 1,
                           "note": ("LDS-resident, FIVE workgroups per simulation since round 6 (four row bands for the stencil phases with recomputed halos + one "
                                    "solver workgroup; hand-offs of divergence and pressure through global memory) + one density workgroup: B = %d simulations "
                                    "occupy %d of 256 CUs; the launch is a latency chain (bands 5.7 us -> solve 15.4 us -> projection 2 us), not a streaming kernel"
                                    % (B, solver_cus)) if bands else
                                   ("LDS-resident, one workgroup (CU) per simulation: B = %d simulations occupy %d of 256 CUs, so the fraction of "
                                    "the CHIP's HBM roofline at this batch size is bounded by B/256 = %.3f" % (B, B, B / 256.0))}
        bname, bst = pick("k_karman_bwd_bww", "k_karman_bwd")
        kern_tab = {k: {"calls": v["calls"], "avg_us": round(v["avg_us"], 2), "share": round(v["total_us"] / tot_prof, 4)}
                    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_us"])}
        out = {
            "metric": "sim-steps/s, SOL-32 training (karman-2d 128x64, fwd+bwd+Adam)",
            "value": value, "unit": "sim-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            # the distribution of the K timed steps (HIP events around every step, no synchronisation added) and the clock the box held
            "ms_per_step_median": _quantile(per_step_ms, 0.5), "ms_per_step_p10": _quantile(per_step_ms, 0.1), "ms_per_step_p90": _quantile(per_step_ms, 0.9),
            # the two reference-width companions of the headline arithmetic FIRST (filled in below): the same step with true 24-bit operand
            # splits (bf16x6) and with every convolution on v_mfma_f32_*_f32 (strict fp32)
            "bf16x6_ms_per_step": None,
            # the conservative companions of the headline, as top-level scalars (filled in below; null when a leg did not run):
            # the same step with every convolution on v_mfma_f32_*_f32 (no operand splits), that kernel's fraction of the fp32 matrix
            # peak, and the fused advect + pressure launch's fraction of the HBM peak at this batch size (measured k)
            "strict_fp32_ms_per_step": None, "strict_fp32_frac": None, "solver_step_frac": roof_solver["frac"] if roof_solver else None,
            # ... and the other shapes of the same path that the extras time (filled in below): the reference's own recipe (64x32, B = 3,
            # SOL-32), the no-grad roll-out at B = 1 and at the bench batch, one karman-3d SOL-16 step
            "recipe_64x32_b3_ms_per_step": None, "rollout_b1_us_per_step": None, "rollout_b6_us_per_step": None, "karman3d_sol16_ms_per_step": None,
            "burgers_non_ms_per_step": None, "burgers_sol04_ms_per_step": None,
            "valid": bool(valid), "rank_skew": rank_skew, "pre_warmup_steps": max(0, args.prewarm),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "scaling_status": ("this line is the N = %d point of a weak-scaling series (6 simulations per GPU); the driver computes the curve from its own "
                               "N = 1, 2, 4, 8 runs.  No N > 1 run on separate GPUs had executed when this file was committed (no multi-GPU node in "
                               "rounds 1-6): the 1 -> 8 curve is UNMEASURED, nothing is extrapolated here; tools/scale.sh runs the four points + the "
                               "DESIGN.md section 6 checklist on a node that has them" % world),
            "dtype": {"split": "f32 (fp16x3 split MFMA in 32-ch convs)", "bf16x6": "f32 (bf16x6 split MFMA in 32-ch convs)", "fp32": "f32"}[args.precision],
            "data": "synthetic",
            "dtype_note": "fp32 tensors and fp32 accumulation everywhere; with precision=split the 32-channel convolutions evaluate each fp32 "
                          "product as three exact fp16 MFMA products of power-of-two scaled 22-bit operand splits (single convolution: error vs float64 <= the "
                          "fp32-MFMA kernel's, tests/test_gpu_parity.py::test_split_conv_relative_l2_not_worse_than_fp32_mfma); strict_fp32 = same step on "
                          "v_mfma_f32_*_f32.  AT THIS WORKLOAD (C3, float64 fixture; precision_at_c3 below): final fields equal to the strict trainer's error, "
                          "per-step losses / gradient %s -- the weights' two-fp16-plane representation (<= 2^-23 relative, a fixed perturbation of "
                          "the network), not the arithmetic: test_sol32_split_trainer_equals_strict_fp32_on_weights_representable_in_two_fp16_planes"
                          % precision_note(),
            "precision_at_c3": load_precision(),
            "config": {"workload": "karman-2d %dx%d SOL-%d, batch %d Re values per GPU (BASELINE configs[2])" % (Y, X, ms, B),
                       "global_batch": world * B, "msteps": ms, "parallelism": "dp%d" % world, "lr": args.lr},
            "workload_deviations_from_survey_8d": [
                "ground truth = plain solver roll-out of the start state perturbed by 0.05 x seeded noise (seed 4321+rank), not independent "
                "seed-4321 noise frames: unreachable random targets make Adam blow the 32-step unroll up",
                "start state passed through one solver step (spin-up) instead of one oracle projection",
                "output layer of the Glorot-initialised corrector scaled by 0.01",
                "lr = %g instead of 1e-4: at 1e-4 this synthetic workload diverges (loss 2386 -> 118280 -> 11701 -> ... -> NaN after ~10 steps, "
                "float64 oracle agrees: tests/golden/train_128x64_sol32.npz)" % args.lr,
                "value / ms_per_step = wall clock over the K timed steps / K (barrier + synchronize on both sides); the median and the p10 / p90 of "
                "per-step HIP events are printed beside it (ms_per_step_median, step_time_distribution_ms)",
            ],
            "loss": loss, "loss_warmup": trace,
            "step_time_distribution_ms": step_distribution(per_step_ms),
            # the shader clock the device HELD during the timed steps (s_memtime / s_memrealtime stamps in front of every step): boxes of this pool
            # run the same build at 11.8 .. 12.8 ms per step with identical idle / probe clocks; this is the number that moves with them
            "shader_clock_during_timed_steps": dict(CLOCK_DURING, note="100 MHz x d(s_memtime) / d(s_memrealtime) between stamp kernels on the launch stream; "
                                                    "ms_per_step x mhz_mean / 2400 = the step time normalised to a 2.4 GHz clock",
                                                    ms_per_step_at_2400_mhz=(ms_per_step * CLOCK_DURING["mhz_mean"] / 2400.0) if CLOCK_DURING.get("mhz_mean") else None),
            "device_state_before": state_before, "device_state_after": state_after,
            "roofline": roof_conv if roof_conv and (not roof_solver or cst["total_us"] >= sst["total_us"]) else roof_solver,
            "roofline_solver_step": roof_solver,
            "roofline_conv": dict(roof_conv) if roof_conv else None,
            "traffic_source": {"file": os.path.relpath(TRAFFIC_FILE, ROOT), "matches_this_build": bool(traffic.get("kernels")),
                               "note": "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc passes, gfx950 correction); null when the "
                                       "counters were collected for another build of the library"},
            "kernels_in_step": kern_tab,
            "profiled_step_kernel_time_ms": tot_prof * 1e-3,
            "train_step_breakdown": {"solver_alg_bytes_fwd": fwd_b, "solver_alg_bytes_bwd": bwd_b,
                                     "cg_iters_fwd_mean": kf_tr, "cg_iters_bwd_mean": kb_tr,
                                     "solver_fwd_ms": sst["total_us"] * 1e-3 if sst else None,
                                     "solver_bwd_fused_ms": bst["total_us"] * 1e-3 if bst else None,
                                     "conv_flop_per_train_step": 3.0 * 520000.0 * N * B * ms,
                                     "conv_fp32_equiv_TFLOPs_of_step": 3.0 * 520000.0 * N * B * ms / (ms_per_step * 1e-3) / 1e12},
            "data_parallel": dp,
            "cpu_affinity": {k: v for k, v in (sol_amd.dist.AFFINITY or {}).items() if k != "inherited"} or None,
        }
        if out["roofline"] is roof_conv and roof_solver:     # the north star's second roofline, inside the object the driver's `parsed` keeps
            out["roofline"]["solver_step"] = {"kernel": sname, "bound": "hbm", "frac": roof_solver["frac"], "achieved_GBps": roof_solver["achieved"],
                                              "launch_us": roof_solver["launch_us"], "share_of_step_kernel_time": roof_solver["share_of_step_kernel_time"],
                                              "note": "fused advect + pressure launch as it runs in the training graph; algorithmic bytes with the measured k (0: direct solve)"}
        extras = not args.no_extras and (world == 1 or args.all_legs)
        if extras:       # single-GPU extras (they train / roll out on rank 0 only: no collectives in them; N > 1: only with --all-legs)
            # the same kernel with one simulation per CU (256 simulations): what the LDS-resident design delivers per chip
            try:
                Bf = 256
                df, vyf, vxf = (t[:1].expand(Bf, -1, -1).contiguous() for t in (wl.d0, wl.vy0, wl.vx0))
                ref_ = wl.re[:1].expand(Bf).contiguous()
                cfgf = ops.karman_cfg(Bf, Y, X, wl.dx, masks=wl.masks)
                infof = {}
                t_full = time_call(lambda: ops.karman_step(df, vyf, vxf, ref_, cfgf, wl.masks, infof), 10)
                kff = float(infof["iterations"].double().mean().item())
                bytes_full = 4.0 * (10 * Nf + 9 * N + 11.0 * N * kff) * Bf
                out["roofline_solver_step_full_chip_256_sims"] = {
                    "launch_us": t_full * 1e6, "sim_steps_per_s": Bf / t_full, "algorithmic_GBps": bytes_full / t_full / 1e9,
                    "frac_of_hbm_peak": bytes_full / t_full / (PEAK_HBM_GBS * 1e9), "note": "not a BASELINE config: occupancy reference only"}
                del df, vyf, vxf
            except Exception as e:
                out["roofline_solver_step_full_chip_256_sims"] = {"error": str(e)}
            # the same training step in the two other convolution arithmetics: strict fp32 MFMA (every convolution on
            # v_mfma_f32_*_f32) and bf16x6 (true 24-bit operand splits, six exact bf16 products per fp32 product)
            for leg, prec, kname, npr in (("strict_fp32", "fp32", "k_conv5x5_r3<2>", 1), ("bf16x6", "bf16x6", "k_conv5x5_sb<2, 0>", 6)):
                if args.precision == prec:
                    continue
                try:
                    trp = sol_amd.SolTrainer(wl.net.clone(), wl.masks, B, Y, X, ms, wl.dx, wl.std_v,
                                             synthetic.STD_RE, conv_precision=prec)
                    np_ = max(3, min(5, args.steps))
                    sp, lp, _ = timed_steps(wl, args.lr, np_, 2, lambda: torch.cuda.synchronize(), trainer=trp)
                    pp = profile_kernels(wl, args.lr, trainer=trp)
                    cp = pp.get(kname)
                    pk = PEAK_MFMA32_TF if npr == 1 else PEAK_MFMA16_TF
                    out[leg] = {"ms_per_step": sp / np_ * 1e3, "sim_steps_per_s": B * ms * np_ / sp, "loss": lp,
                                "roofline": None if not cp else {
                                    "kernel": kname, "bound": "mfma", "achieved": flop_conv / (cp["avg_us"] * 1e-6) / 1e12,
                                    "peak": pk, "unit": "TFLOP/s", "frac": flop_conv / (cp["avg_us"] * 1e-6) / (pk * 1e12),
                                    "mfma_pipe_busy": npr * flop_conv / (cp["avg_us"] * 1e-6) / (pk * 1e12),
                                    "launch_us": cp["avg_us"], "traffic": traffic_bytes(traffic, kname, ((B * Y + 2) // 3) * max(1, X // 64) * 768)}}
                    del trp
                    if leg == "bf16x6":
                        out["bf16x6_ms_per_step"] = out[leg]["ms_per_step"]
                    if leg == "strict_fp32":
                        out["strict_fp32_ms_per_step"] = out[leg]["ms_per_step"]
                        out["strict_fp32_frac"] = out[leg]["roofline"]["frac"] if out[leg]["roofline"] else None
                        if isinstance(out.get("roofline"), dict):      # the driver's `parsed` keeps the roofline object: carry the conservative figures there too
                            out["roofline"]["strict_fp32"] = {"ms_per_step": out[leg]["ms_per_step"], "kernel": kname,
                                                              "frac_of_fp32_matrix_peak": out["strict_fp32_frac"],
                                                              "launch_us": out[leg]["roofline"]["launch_us"] if out[leg]["roofline"] else None}
                except Exception as e:
                    out[leg] = {"error": str(e)}
            # the reference's own training recipe (karman-2d/Makefile:78-80): 64x32, batch 3, SOL-32 -- with the per-layer
            # conv launches (default) and with the ten 32->32 layers of a CNN pass as ONE persistent launch (option
            # cnn_persistent: 32 workgroups at this size, where a launch is pure latency)
            if (Y, X, B) == (128, 64, 6):
                for leg, persistent in (("reference_recipe_64x32_b3", 0), ("reference_recipe_64x32_b3_persistent_cnn", 1)):
                    if persistent and args.precision != "split":
                        continue
                    try:
                        sol_amd._lib.set_option("cnn_persistent", persistent)          # enters the workspace size: set before the trainer exists
                        wl2 = Workload(sol_amd, dev, 3, 64, 32, ms, rank, args.precision)
                        s2, l2, _ = timed_steps(wl2, args.lr, 10, 3, lambda: torch.cuda.synchronize())
                        out[leg] = {"workload": "karman-2d 64x32 SOL-%d, batch 3 (karman-2d/Makefile:78-80)" % ms, "cnn_persistent": persistent,
                                    "ms_per_step": s2 / 10 * 1e3, "sim_steps_per_s": 3 * ms * 10 / s2, "loss": l2,
                                    "cg_iters_fwd_mean": wl2.trainer.solver_algorithmic_bytes()[2]}
                        del wl2
                    except Exception as e:
                        out[leg] = {"error": str(e)}
                    finally:
                        sol_amd._lib.set_option("cnn_persistent", 0)
            # second half of BASELINE.json's metric: no-grad roll-out (karman_apply.py:138-158).  B = 1 is the reference's
            # script shape (one latency-bound launch after the other on 43 workgroups); B = 6 is the same engine on the
            # bench batch; "persistent_cnn" = the ten 32->32 layers of a step as one launch (option cnn_persistent)
            out["rollout"] = {}
            for leg, rb, persistent in (("b1", 1, 0), ("b1_persistent_cnn", 1, 1), ("b6", B, 0)):
                try:
                    sol_amd._lib.set_option("cnn_persistent", persistent)
                    ro = sol_amd.SolRollout(wl.net, wl.masks, rb, Y, X, wl.dx, wl.std_v, synthetic.STD_RE)
                    rd, ry, rx = wl.d0[:rb].clone(), wl.vy0[:rb].clone(), wl.vx0[:rb].clone()
                    re1 = wl.re[:rb].contiguous()
                    ro.run(rd, ry, rx, re1, 6)
                    t_ro = time_call(lambda: ro.run(rd, ry, rx, re1, 50), 2)
                    out["rollout"][leg] = {"sim_steps_per_s": 50.0 * rb / t_ro, "batch": rb, "steps": 50, "us_per_step": t_ro / 50 * 1e6,
                                           "finite": bool(torch.isfinite(ry).all())}
                    del ro
                except Exception as e:          # never let the extra line break the contract line
                    out["rollout"][leg] = {"error": str(e)}
                finally:
                    sol_amd._lib.set_option("cnn_persistent", 0)
            if "b1" in out["rollout"] and "us_per_step" in out["rollout"]["b1"]:
                out["rollout"].update({k: out["rollout"]["b1"][k] for k in ("sim_steps_per_s", "batch", "steps", "us_per_step")})
            out["recipe_64x32_b3_ms_per_step"] = out.get("reference_recipe_64x32_b3", {}).get("ms_per_step")
            out["rollout_b1_us_per_step"] = out["rollout"].get("b1", {}).get("us_per_step")
            out["rollout_b6_us_per_step"] = out["rollout"].get("b6", {}).get("us_per_step")
        if extras:
            try:
                out["burgers"] = burgers_leg(sol_amd, dev)
            except Exception as e:
                out["burgers"] = {"error": str(e)}
            out["burgers_non_ms_per_step"] = (out["burgers"].get("non_m1") or {}).get("ms_per_step")
            out["burgers_sol04_ms_per_step"] = (out["burgers"].get("sol04_m4") or {}).get("ms_per_step")
        if extras:
            try:
                out["karman3d"] = karman3d_leg(sol_amd, dev)
            except Exception as e:
                out["karman3d"] = {"error": str(e)}
            out["karman3d_sol16_ms_per_step"] = (out["karman3d"].get("train_sol16") or {}).get("ms_per_step")
        if k3d_dp is not None:
            out["karman3d_data_parallel"] = k3d_dp
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, Y, X, B)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
