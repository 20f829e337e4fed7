"""ctypes binding of libsol_hip.so (C ABI declared in include/sol_hip.h).

The product path has NO CPU fallback: if the library is missing, or a call is made without
a ROCm device, this module raises.  PyTorch is used only as the owner of device memory and
streams (tensor.data_ptr(), torch.cuda.current_stream()).
"""
import ctypes as C
import os
import re

import torch

from . import _build

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(_HERE, "..", "include", "sol_hip.h")


class SolError(RuntimeError):
    pass


class KarmanCfg(C.Structure):
    """sol_karman_cfg"""
    _fields_ = [("B", C.c_int32), ("Y", C.c_int32), ("X", C.c_int32),
                ("dx", C.c_float), ("dt", C.c_float), ("res", C.c_float),
                ("cg_rtol", C.c_float), ("cg_atol", C.c_float), ("cg_max_iter", C.c_int32),
                ("grad_pad", C.c_int32), ("inflow_before", C.c_int32),
                ("coarse_n", C.c_int32), ("coarse_inv", C.c_void_p),
                ("direct_n", C.c_int32), ("direct", C.c_void_p)]


class BurgersCfg(C.Structure):
    """sol_burgers_cfg"""
    _fields_ = [("B", C.c_int32), ("Y", C.c_int32), ("X", C.c_int32), ("dx", C.c_float), ("dt", C.c_float)]


class Karman3DCfg(C.Structure):
    """sol_karman3d_cfg"""
    _fields_ = [("B", C.c_int32), ("Y", C.c_int32), ("X", C.c_int32), ("Z", C.c_int32),
                ("dx", C.c_float), ("dt", C.c_float), ("res", C.c_float),
                ("grad_pad", C.c_int32), ("inflow_before", C.c_int32),
                ("direct_n", C.c_int32), ("direct", C.c_void_p)]


class TrainCfg(C.Structure):
    """sol_train_cfg"""
    _fields_ = [("karman", KarmanCfg), ("msteps", C.c_int32),
                ("std_v0", C.c_float), ("std_v1", C.c_float), ("std_re", C.c_float),
                ("lrelu_slope", C.c_float),
                ("in_std_v0", C.c_float), ("in_std_v1", C.c_float), ("out_std_v0", C.c_float), ("out_std_v1", C.c_float)]


_P = C.c_void_p
_SIGS = {
    "sol_last_error": (C.c_char_p, []),
    "sol_version": (C.c_int, []),
    "sol_abi_sizes": (C.c_int, [C.POINTER(C.c_int32)] * 3),
    "sol_set_option": (C.c_int, [C.c_char_p, C.c_int32]),
    "sol_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32)]),
    "sol_prof_begin": (C.c_int, []),
    "sol_prof_end": (C.c_int, [C.c_int32, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "sol_karman_direct_supported": (C.c_int, [C.c_int32] * 2),
    "sol_karman_step_large_workspace_bytes": (C.c_size_t, [_P]),
    "sol_karman_step_fwd_large": (C.c_int, [_P] * 10 + [C.c_int64] + [_P] * 7 + [C.c_size_t]),
    "sol_karman_precond_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "sol_karman_step_fwd": (C.c_int, [C.POINTER(KarmanCfg), _P] + [_P] * 8 + [C.c_int64] + [_P] * 6 + [C.POINTER(C.c_float), _P]),
    "sol_karman_step_bwd": (C.c_int, [C.POINTER(KarmanCfg), _P] + [_P] * 5 + [C.c_int64] + [_P] * 3 + [C.POINTER(C.c_float)] + [_P] * 3),
    "sol_burgers_step_fwd": (C.c_int, [C.POINTER(BurgersCfg), _P] + [_P] * 10),
    "sol_burgers_step_bwd": (C.c_int, [C.POINTER(BurgersCfg), _P] + [_P] * 10),
    "sol_burgers_step_large_workspace_bytes": (C.c_size_t, [C.POINTER(BurgersCfg)]),
    "sol_burgers_step_fwd_large": (C.c_int, [C.POINTER(BurgersCfg), _P] + [_P] * 10 + [_P, C.c_size_t]),
    "sol_conv5x5_packed_floats": (C.c_size_t, [C.c_int32] * 3),
    "sol_conv5x5_pack": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "sol_conv5x5_pack_jobs": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "sol_conv5x5_bwd_weight_reduce_jobs": (C.c_int, [_P, C.c_int32] + [C.POINTER(C.c_void_p)] * 3 + [C.c_int32] * 3 + [C.POINTER(C.c_int32)] * 2 + [C.c_int32]),
    "sol_conv5x5": (C.c_int, [_P] * 7 + [C.c_int32] * 6 + [C.c_float]),
    "sol_absmax_slots": (C.c_int32, []),
    "sol_absmax": (C.c_int, [_P, _P, C.c_int64, _P]),
    "sol_conv5x5_scaled": (C.c_int, [_P] * 7 + [C.c_int32] * 6 + [C.c_float] + [_P] * 2),
    "sol_conv5x5_bwd_weight_ws_floats": (C.c_size_t, [C.c_int32] * 5),
    "sol_conv5x5_bwd_weight": (C.c_int, [_P] * 4 + [C.c_int32] * 5),
    "sol_conv5x5_bwd_weight_reduce": (C.c_int, [_P] * 4 + [C.c_int32] * 6),
    "sol_train_workspace_bytes": (C.c_size_t, [C.POINTER(TrainCfg)]),
    "sol_train_fwd_bwd": (C.c_int, [C.POINTER(TrainCfg), _P] + [_P] * 9 + [C.c_int64] + [_P] * 3 + [C.c_size_t] + [_P] * 7),
    "sol_train_graph_create": (C.c_int, [C.POINTER(TrainCfg)] + [_P] * 9 + [C.c_int64] + [_P] * 3 + [C.c_size_t] + [_P] * 7 + [C.POINTER(C.c_void_p)]),
    "sol_train_graph_launch": (C.c_int, [_P, _P]),
    "sol_train_graph_destroy": (C.c_int, [_P]),
    "sol_copy_words": (C.c_int, [_P, _P, _P, C.c_int64]),
    "sol_clock_probe": (C.c_int, [_P, _P, C.c_int32]),
    "sol_clock_stamp": (C.c_int, [_P, _P]),
    "sol_latency_probe": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    "sol_graph_census": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int32]),
    "sol_graph_check": (C.c_int, [_P, C.c_char_p]),
    "sol_graph_node_type_name": (C.c_char_p, [C.c_int32]),
    "sol_rollout_workspace_bytes": (C.c_size_t, [C.POINTER(TrainCfg)]),
    "sol_rollout": (C.c_int, [C.POINTER(TrainCfg), _P] + [_P] * 9 + [C.c_int64, C.c_int32, _P, C.c_size_t, _P]),
    "sol_l2_loss_scratch_floats": (C.c_int32, []),
    "sol_l2_loss_fwd_bwd": (C.c_int, [_P, C.c_int32] + [C.POINTER(C.c_void_p)] * 3 + [C.POINTER(C.c_int64), C.POINTER(C.c_float), C.c_float, C.c_int32, _P, C.c_int32, _P]),
    "sol_adam_tf_step": (C.c_int, [_P] * 5 + [C.c_int64, C.c_int32] + [C.c_float] * 5 + [C.POINTER(C.c_int64), C.c_int32, _P]),
    "sol_comm_unique_id": (C.c_int, [C.c_char_p]),
    "sol_comm_init": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "sol_allreduce_grads": (C.c_int, [_P, _P, _P, C.c_int64]),
    "sol_comm_destroy": (C.c_int, [_P]),
    "sol_abi_size_karman3d": (C.c_int32, []),
    "sol_karman3d_step_workspace_bytes": (C.c_size_t, [C.POINTER(Karman3DCfg)]),
    "sol_karman3d_step_fwd": (C.c_int, [C.POINTER(Karman3DCfg), _P] + [_P] * 9 + [C.c_int64] + [_P] * 8 + [C.POINTER(C.c_float), _P, _P, C.c_size_t]),
    "sol_karman3d_step_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(Karman3DCfg)]),
    "sol_karman3d_step_bwd": (C.c_int, [C.POINTER(Karman3DCfg), _P] + [_P] * 6 + [C.c_int64] + [_P] * 6 + [_P, _P, C.c_size_t]),
    "sol_karman3d_correct": (C.c_int, [_P, _P, C.c_int32] + [C.c_float] * 3 + [_P] * 3 + [C.c_int32] * 4),
    "sol_karman3d_correct_bwd": (C.c_int, [_P] * 7 + [C.c_float] * 3 + [_P] + [C.c_int32] * 4),
    "sol_karman3d_feature_bwd": (C.c_int, [_P, _P] + [C.c_float] * 3 + [_P] * 3 + [C.c_int32] * 4),
    "sol_conv3d_packed_floats": (C.c_size_t, [C.c_int32] * 2),
    "sol_conv3d_pack": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "sol_conv3d": (C.c_int, [_P] * 7 + [C.c_int32] * 7 + [C.c_float] + [_P] * 2),
    "sol_conv3d_thin_packed_floats": (C.c_size_t, []),
    "sol_conv3d_thin_ws_floats": (C.c_size_t, [C.c_int32] * 4),
    "sol_conv3d_thin_pack": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    "sol_conv3d_thin": (C.c_int, [_P] * 7 + [C.c_int32] * 5 + [C.c_float, _P]),
    "sol_conv3d_thin_out_pack": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    "sol_conv3d_thin_out": (C.c_int, [_P] * 6 + [C.c_int32] * 5 + [_P]),
    "sol_conv3d_thin_bwd_weight_ws_floats": (C.c_size_t, [C.c_int32] * 4),
    "sol_conv3d_thin_bwd_weight_acc": (C.c_int, [_P] * 8 + [C.c_int32] * 7),
    "sol_conv3d_thin_out_bwd_weight_acc": (C.c_int, [_P] * 8 + [C.c_int32] * 7),
    "sol_conv3d_bwd_weight_ws_floats": (C.c_size_t, [C.c_int32] * 6),
    "sol_conv3d_bwd_weight": (C.c_int, [_P] * 9 + [C.c_int32] * 8),
    "sol_conv3d_bwd_weight_acc": (C.c_int, [_P] * 9 + [C.c_int32] * 10),
    "sol_mars_moon_layer": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}

_lib = None


def declared_symbols():
    """Every function name declared in include/sol_hip.h (used by the CPU test-suite)."""
    with open(HEADER) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sol_[a-z0-9_]+)\s*\(", text)))


def lib_path():
    return _build.LIB


ABI_VERSION = 215     # sol_version() of the library these bindings were written against

# Debugging overrides: environment variable -> (option, value).  Read ONCE here, in Python, when the library is loaded;
# the library itself never reads the environment (options are set through sol_set_option, include/sol_hip.h).
_ENV_OPTIONS = {
    "SOL_CONV_NO_SB": ("conv_precision", 2), "SOL_CONV_NO_FP16": ("conv_precision", 1), "SOL_CONV_SPLIT3": ("conv_split3", 1),
    "SOL_CONV_NO_R3": ("conv_r3", 0), "SOL_CONV_NO_THIN": ("conv_thin", 0), "SOL_CONV_NO_BWW32": ("conv_bww32", 0),
    "SOL_CORRECT_NO_FUSE": ("correct_fuse", 0), "SOL_BWW_NO_FUSE": ("bww_fuse", 0), "SOL_BWW_NO_SIDE": ("bww_side", 0),
    "SOL_DENSITY_INLINE": ("density_mode", 1), "SOL_DENSITY_NO_FUSE": ("density_mode", 2), "SOL_STEP_PROF": ("step_prof", 1),
    "SOL_CNN_NO_PERSISTENT": ("cnn_persistent", 0), "SOL_CONV_NO_DX": ("conv_dx", 0), "SOL_CONV_NO_THIN_VALU": ("conv_thin_valu", 0),
}
_ENV_INT_OPTIONS = {"SOL_FWD_BANDS": "fwd_bands", "SOL_CONV_THIN_VALU": "conv_thin_valu", "SOL_BWW_CHUNK": "bww_chunk", "SOL_STREAMS": "streams", "SOL_CPT": "cpt", "SOL_DBG_SKIP": "dbg_skip"}


def load():
    """Load (building on demand when hipcc is available) and type the library."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if _build._stale():
        # missing, or built from other sources than the ones in the tree (content hash): rebuild, or refuse to load a
        # library whose struct layouts / signatures may differ from the ctypes mirror in this file
        try:
            _build.build()
        except Exception as e:  # no silent fallback: the HIP extension IS the product
            if not os.path.exists(path):
                raise SolError("libsol_hip.so is missing and could not be built: %s" % e)
            if _build.have_hipcc():
                raise SolError("libsol_hip.so is stale and the rebuild failed: %s" % e)
            # no compiler on this box (a GPU box running a shipped library): the ABI checks below decide
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.sol_version() != ABI_VERSION:
        raise SolError("libsol_hip.so reports ABI version %d, these bindings need %d: rebuild (python -m sol_amd._build --force)"
                       % (lib.sol_version(), ABI_VERSION))
    kc, bc, tc = C.c_int32(), C.c_int32(), C.c_int32()
    lib.sol_abi_sizes(C.byref(kc), C.byref(bc), C.byref(tc))
    if (kc.value, bc.value, tc.value) != (C.sizeof(KarmanCfg), C.sizeof(BurgersCfg), C.sizeof(TrainCfg)):
        raise SolError("struct layout mismatch between libsol_hip.so %r and the ctypes mirror %r"
                       % ((kc.value, bc.value, tc.value), (C.sizeof(KarmanCfg), C.sizeof(BurgersCfg), C.sizeof(TrainCfg))))
    if lib.sol_abi_size_karman3d() != C.sizeof(Karman3DCfg):
        raise SolError("struct layout mismatch: sol_karman3d_cfg is %d bytes in libsol_hip.so, %d in the ctypes mirror"
                       % (lib.sol_abi_size_karman3d(), C.sizeof(Karman3DCfg)))
    _lib = lib
    for env, (opt, val) in _ENV_OPTIONS.items():
        if os.environ.get(env):
            set_option(opt, val)
    for env, opt in _ENV_INT_OPTIONS.items():
        if os.environ.get(env):
            set_option(opt, int(os.environ[env]))
    return lib


def set_option(name, value):
    """sol_set_option: process-wide kernel-variant option (see include/sol_hip.h)."""
    check(load().sol_set_option(name.encode(), int(value)))


def get_option(name):
    v = C.c_int32()
    check(load().sol_get_option(name.encode(), C.byref(v)))
    return v.value


class profile:
    """with profile() as p: ...eager library calls...  ->  p.kernels = {name: (calls, total_us)}.
    Per-launch HIP events on the launch stream (sol_prof_begin / sol_prof_end)."""
    MAXC = 64

    def __enter__(self):
        check(load().sol_prof_begin())
        self.kernels = {}
        return self

    def __exit__(self, *exc):
        names = C.create_string_buffer(self.MAXC * 64)
        tot = (C.c_double * self.MAXC)()
        calls = (C.c_int32 * self.MAXC)()
        n = load().sol_prof_end(self.MAXC, names, tot, calls)
        if n < 0:
            check(n)
        for k in range(n):
            nm = names.raw[k * 64:(k + 1) * 64].split(b"\0")[0].decode()
            self.kernels[nm] = (calls[k], tot[k])
        return False


class no_gc_during_capture:
    """A stream capture must not be interrupted by Python's cyclic garbage collector: a collected object whose finaliser
    calls the HIP runtime (a trainer destroying its hipGraph, a torch CUDAGraph, a freed allocator segment) makes the runtime
    refuse the call -- or abort -- while ANY stream of the process is capturing in global mode.  Collect what is collectable
    first, then keep the collector off for the duration of the capture."""

    def __enter__(self):
        import gc
        gc.collect()
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False


def dcopy_(dst, src):
    """dst.copy_(src) for contiguous 32-bit device tensors of equal size AS A KERNEL (sol_copy_words): torch's copy_ / clone() of a
    contiguous tensor is a hipMemcpyAsync, i.e. a MEMCPY NODE in a captured graph, which sol_graph_check refuses.  Returns dst."""
    if dst.numel() != src.numel() or dst.dtype != src.dtype:
        raise SolError("dcopy_: %s %s <- %s %s" % (tuple(dst.shape), dst.dtype, tuple(src.shape), src.dtype))
    if dst.dtype not in (torch.float32, torch.int32) or not (dst.is_cuda and src.is_cuda):
        # checked HERE, not by ptr() in the middle of a stream capture
        raise SolError("dcopy_: 32-bit (float32 / int32) CUDA tensors only (got %s on %s <- %s on %s)" % (dst.dtype, dst.device, src.dtype, src.device))
    src = src.detach()
    if not (dst.is_contiguous() and src.is_contiguous()):
        # a strided copy is an elementwise kernel already; equal numel with different shapes means "the same words", as in the kernel form
        dst.copy_(src if src.shape == dst.shape else src.reshape(dst.shape))
        return dst
    check(load().sol_copy_words(stream(), ptr(dst), ptr(src), dst.numel()))
    return dst


def dclone(t):
    """t.detach().clone() without a memcpy node (see dcopy_)."""
    return dcopy_(torch.empty_like(t, memory_format=torch.contiguous_format), t.detach().contiguous())


def pad_high(t, *dims):
    """Zero padding by ONE element at the high end of each of `dims` -- torch.nn.functional.pad(t, ...) written as concatenation with
    zeros.  Same values; the difference is what a captured graph holds: constant_pad_nd copies `t` into a narrow of the result, and its
    backward clone()s a narrow of the gradient -- with batch size 1 (or padding along dim 0) those narrows are CONTIGUOUS, the copies are
    hipMemcpyAsync, and the captured trainer holds memcpy nodes (refused by sol_graph_check).  cat / its backward are kernels / views."""
    for d in dims:
        shape = list(t.shape)
        shape[d] = 1
        t = torch.cat([t, t.new_zeros(shape)], dim=d)
    return t


def stack0(ts):
    """torch.stack(ts) of 0-d tensors; a stack of ONE tensor is a plain contiguous copy (a memcpy node under capture): reshape instead."""
    return ts[0].reshape(1) if len(ts) == 1 else torch.stack(ts)


def graph_census(raw_graph):
    """{node type name: count} of a hipGraph_t (child graphs entered) -- sol_graph_census."""
    lib = load()
    counts = (C.c_int32 * 32)()
    check(lib.sol_graph_census(C.c_void_p(int(raw_graph)), counts, 32))
    return {lib.sol_graph_node_type_name(t).decode(): int(counts[t]) for t in range(32) if counts[t]}


MIN_TORCH_FOR_CAPTURE = "2.8"


def capture_graph(fn, what):
    """Capture `fn()` (launches on torch's current stream) into a torch CUDAGraph and return it, instantiated -- THE way this package
    captures torch-composed work (GraphTrainer, BurgersTrainer, BurgersRollout, Karman3DTrainer).  The captured graph is passed through
    sol_graph_check BEFORE it is instantiated: anything but kernel nodes (a memset node = a multi-workgroup torch reduction's semaphore
    clear or a torch.zeros() inside the capture; a memcpy node = data staged inside the replayed region) raises SolError naming the node
    types.  Memset nodes replay unreliably on ROCm 7.2 (per-step losses 0.5x / 2x the true values after a few
    replays); with this guard that defect class is refused at capture time instead of being tested for after the fact."""
    try:
        g = torch.cuda.CUDAGraph(keep_graph=True)
        if not (hasattr(g, "raw_cuda_graph") and hasattr(g, "instantiate")):
            raise TypeError("CUDAGraph lacks raw_cuda_graph() / instantiate()")
    except TypeError as e:
        # keep_graph / raw_cuda_graph / instantiate exist from torch 2.8 on; without them the captured graph cannot be inspected before it
        # is instantiated, and an unchecked capture is exactly what this function exists to refuse
        raise SolError("capturing %s needs torch >= %s (torch.cuda.CUDAGraph(keep_graph=True), raw_cuda_graph(), instantiate()); this is torch %s "
                       "(%s).  Run the trainer with use_graph=False, or upgrade torch." % (what, MIN_TORCH_FOR_CAPTURE, torch.__version__, e))
    with no_gc_during_capture(), torch.cuda.graph(g):
        fn()
    try:
        check(load().sol_graph_check(C.c_void_p(int(g.raw_cuda_graph())), what.encode()))
    except SolError:
        g.reset()
        raise
    g.instantiate()
    return g


def check(code):
    if code != 0:
        raise SolError("libsol_hip error %d: %s" % (code, load().sol_last_error().decode()))


def require_gpu():
    if not torch.cuda.is_available():
        raise SolError("no ROCm device visible: the solver-in-the-loop engine has no CPU fallback")


def ptr(t):
    """Device pointer of a contiguous fp32/int32 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SolError("expected a CUDA tensor")
    if not t.is_contiguous():
        raise SolError("expected a contiguous tensor")
    if t.dtype not in (torch.float32, torch.int32):
        raise SolError("expected a float32 / int32 tensor (got %s): the C ABI takes 32-bit buffers only" % t.dtype)
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def f32(t, device=None):
    """contiguous fp32 CUDA tensor view/copy of t"""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    return t.to(device=device or "cuda", dtype=torch.float32).contiguous()

