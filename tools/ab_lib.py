#!/usr/bin/env python
"""A/B of a build variant against the product library ON ONE BOX (box-to-box variance is 2-4 %, more than most kernel changes).
    python tools/ab_lib.py --build NAME file.hip:-flag[,-Dmacro] [file2.hip:...]     (needs hipcc; no GPU)
        -> solver-in-the-loop_amd/lib/libsol_NAME.so: the product objects with the named sources recompiled with the extra flags
    python tools/ab_lib.py --run NAME [NAME2 ...] [--reps 3] [-- bench args]          (on the GPU box)
        alternates `bench.py --no-extras --no-cpu-baseline` between the product library and each variant (SOL_HIP_LIB), prints
        ms_per_step of every run and the medians."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "solver-in-the-loop_amd")


def variant_path(name):
    return os.path.join(PKG, "lib", "libsol_%s.so" % name)


if "--build" in sys.argv:
    i = sys.argv.index("--build")
    name, specs = sys.argv[i + 1], sys.argv[i + 2:]
    import importlib.util
    spec = importlib.util.spec_from_file_location("_b", os.path.join(PKG, "_build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    b.build()
    objdir = os.path.join(PKG, "build")
    hipcc = b._hipcc()
    changed = {}
    for sp in specs:
        src, flags = sp.split(":", 1)
        obj = os.path.join(objdir, src.replace(".hip", "_%s.o" % name))
        cmd = [hipcc] + b.FLAGS + b.EXTRA.get(src, []) + flags.split(",") + ["-c", os.path.join(PKG, "csrc", src), "-o", obj]
        print(" ".join(cmd)); subprocess.check_call(cmd)
        changed[src] = obj
    objs = [changed.get(s, os.path.join(objdir, s.replace(".hip", ".o"))) for s in b.SOURCES]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", variant_path(name)])
    print(variant_path(name))
    sys.exit(0)

i = sys.argv.index("--run")
rest = sys.argv[i + 1:]
extra = []
if "--" in rest:
    j = rest.index("--"); extra = rest[j + 1:]; rest = rest[:j]
reps = 3
if "--reps" in rest:
    j = rest.index("--reps"); reps = int(rest[j + 1]); rest = rest[:j] + rest[j + 2:]
names = ["product"] + rest
res = {n: [] for n in names}
for r in range(reps):
    for n in names:
        env = dict(os.environ)
        if n != "product":
            env["SOL_HIP_LIB"] = variant_path(n)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--no-cpu-baseline"] + extra, env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            ms = json.loads(out.stdout.strip().splitlines()[-1])["ms_per_step"]
        except Exception:
            print(out.stdout[-2000:], out.stderr[-2000:]); raise
        res[n].append(ms)
        print("rep %d %-12s %.3f ms" % (r, n, ms), flush=True)
print(json.dumps({n: {"median_ms": statistics.median(v), "runs": v} for n, v in res.items()}))
