"""Builds libsol_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the
library is a plain C-ABI shared object (include/sol_hip.h) loaded through ctypes."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
DEFAULT_LIB = os.path.join(LIBDIR, "libsol_hip.so")
LIB = os.environ.get("SOL_HIP_LIB") or DEFAULT_LIB      # SOL_HIP_LIB: an explicitly named PREBUILT variant (tools/ab_lib.py); never built, never stamped
SOURCES = ["karman_step.hip", "karman_large.hip", "karman3d.hip", "conv3d_sb.hip", "burgers_step.hip", "conv5x5.hip", "conv5x5_sb.hip", "conv5x5_dx.hip", "conv5x5_thin.hip", "train.hip", "comm.hip", "cnn_chain.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]
# the CG loop packs its vector updates by hand (float2); the SLP vectoriser only adds v_mov traffic there
# conv3d_sb.hip: packed-f32 VALU (v_pk_mul/fma_f32, what SLP makes of the fp16 split of the row staging) costs ~22 cycles each
# beside MFMAs (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
# The conv / trainer sources likewise: A/B on one box (tools/ab_lib.py) 13.725 -> 13.586 ms per training step.
_NOSLP = ["-fno-slp-vectorize"]
# LLVM's iterative-ILP machine scheduler for the 2-D solver and conv sources: A/B on one box 13.91 -> 13.75 ms per training step
# (solver alone 13.81).  NOT for conv3d_sb.hip: its kernels are scheduled by hand with sched_barrier, and that strategy makes the CNN
# pass 14 % slower (3.30 -> 3.78 ms).
_ITILP = ["-mllvm", "--amdgpu-sched-strategy=iterative-ilp"]
EXTRA = {"karman_step.hip": _NOSLP + _ITILP, "conv3d_sb.hip": _NOSLP, "conv5x5_sb.hip": _NOSLP + _ITILP, "conv5x5.hip": _NOSLP + _ITILP,
         "train.hip": _NOSLP, "cnn_chain.hip": _NOSLP,
         # hand-scheduled with sched_barrier like conv3d_sb.hip: no iterative-ILP strategy; the leading scalar kernel arguments
         # (pointers + tile geometry, 11 dwords) are preloaded into SGPRs by the command processor
         "conv5x5_dx.hip": _NOSLP + ["-mllvm", "-amdgpu-kernarg-preload-count=11"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libsol_hip.so)")


def have_hipcc():
    try:
        _hipcc()
        return True
    except RuntimeError:
        return False


STAMP = os.path.join(LIBDIR, "libsol_hip.sources.sha1")


def _source_hash():
    """Content hash of everything the library is built from (sources, headers, flags): a copy of the tree that scrambles
    the modification times (rsync without -t, a fresh checkout next to a shipped .so) must not trigger a rebuild on every
    rank of a node, and an edited source must trigger one whatever its time stamp says."""
    import hashlib
    h = hashlib.sha1(repr((FLAGS, sorted(EXTRA.items()), SOURCES)).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "sol_hip.h")]
    for d in deps:
        if os.path.isfile(d):
            h.update(os.path.basename(d).encode())
            with open(d, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def _stale():
    if os.environ.get("SOL_HIP_LIB"):             # an explicitly named prebuilt library (tools/ab_lib.py variants): never rebuilt
        if not os.path.exists(LIB):
            raise RuntimeError("SOL_HIP_LIB=%s does not exist (the override names a prebuilt library; it is never built)" % LIB)
        return False
    if not os.path.exists(LIB):
        return True
    try:
        with open(STAMP) as f:
            return f.read().strip() != _source_hash()
    except OSError:
        return True


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libsol_hip.so.  Returns the path.
    Serialised by a lock file: several ranks of one node may import the package at the same time."""
    if os.environ.get("SOL_HIP_LIB"):
        # the stamp belongs to the default library: building "over" the override would link a product build onto the named
        # variant and mark the untouched default library as current
        if force:
            raise RuntimeError("build(force=True) with SOL_HIP_LIB set: unset the override to rebuild the product library")
        _stale()                                  # diagnoses a missing file
        return LIB
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():          # another process built it while this one waited
            return LIB
        return _build_locked(verbose)


def _build_locked(verbose):
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    procs = []
    objs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + EXTRA.get(src, []) + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, out))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", LIB]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    with open(STAMP, "w") as f:
        f.write(_source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
