#!/usr/bin/env python
"""Static check of the compiled kernels' MFMA streams: distance (in MFMA issue slots) between an MFMA and the previous MFMA of the same basic
block that wrote its accumulator.  A dependent v_mfma_f32_16x16x32_* issues ~4 slots after its producer on gfx950 (one wave per SIMD with two
alternating accumulators runs at half rate, profiles/r01_ubench_notes.txt), so distances 1..3 are pipe bubbles unless another wave fills them.
    python tools/mfma_dep_distance.py [file.hip ...]       (no GPU; compiles every source to assembly with the product flags)"""
import collections, importlib.util, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "solver-in-the-loop_amd")
spec = importlib.util.spec_from_file_location("_b", os.path.join(PKG, "_build.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
srcs = sys.argv[1:] or [s for s in b.SOURCES if s not in ("comm.hip",)]
os.makedirs("/tmp/asm", exist_ok=True)
procs = []
for s in srcs:
    out = "/tmp/asm/" + s.replace(".hip", ".s")
    procs.append((s, out, subprocess.Popen([b._hipcc()] + b.FLAGS + b.EXTRA.get(s, []) + ["--cuda-device-only", "-S", os.path.join(PKG, "csrc", s), "-o", out],
                                           stderr=subprocess.DEVNULL)))
rx = re.compile(r"\s+(v_mfma_\w+)\s+([av])\[(\d+):(\d+)\]")
for s, out, p in procs:
    p.wait()
    kern, hist, seq = None, None, []
    res = {}
    def flush():
        last = {}
        for i, (lo, hi) in enumerate(seq):
            d = min((i - last[r] for r in range(lo, hi + 1) if r in last), default=None)
            if d is not None:
                hist[min(d, 9)] += 1
            for r in range(lo, hi + 1):
                last[r] = i
        seq.clear()
    for line in open(out):
        if line.startswith("_Z") and line.rstrip().endswith(":") or (line.startswith("_Z") and ": " in line and "@" in line):
            if kern is not None:
                flush()
            kern = line.split(":")[0]
            hist = res.setdefault(kern, collections.Counter())
            continue
        if kern is None:
            continue
        if line.startswith(".LBB") or "s_cbranch" in line or "s_barrier" in line or "s_endpgm" in line:
            flush()
            continue
        m = rx.match(line)
        if m:
            seq.append((int(m.group(3)), int(m.group(4))))
    if kern is not None:
        flush()
    for k, h in res.items():
        n = sum(h.values())
        if n >= 8:
            name = subprocess.run(["c++filt", k], stdout=subprocess.PIPE, text=True).stdout.strip().replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            print("%-18s %-46s dependent MFMAs %5d   distance 1: %4d  2: %4d  3: %4d  >=4: %5d" % (s, name[:46], n, h[1], h[2], h[3], sum(v for d, v in h.items() if d >= 4)))
