#!/usr/bin/env python
"""Mean counter value per (kernel, grid size) from a rocprofv3 --pmc run written with --output-format csv.
Usage: python tools/pmc_summary.py <dir with *counter_collection.csv> [substring filter] [--json out.json]
--json writes {"kernels": {"<kernel>|<grid>": {"<counter>": mean, "dispatches": n}}} -- the file bench.py reads its
roofline `traffic` from (profiles/r03_pmc_traffic.json); several runs (one counter each) merge into one file.  The file
records the content hash of the library sources it was collected at (sol_amd._build._source_hash()): bench.py reports the
traffic only while the build it runs matches."""
import csv, glob, json, os, re, sys
from collections import defaultdict

argv = list(sys.argv[1:])
jout = None
if "--json" in argv:
    k = argv.index("--json")
    jout = argv[k + 1]
    del argv[k:k + 2]
root = argv[0]
flt = argv[1] if len(argv) > 1 else ""
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name", "")
            if flt not in name:
                continue
            m = re.search(r"(k_\w+(<[^>]*>)?|__amd_\w+)", name)
            key = (m.group(1) if m else name[:48], r.get("Grid_Size", "?"), r.get("Counter_Name", "?"))
            a = acc[key]
            a[0] += float(r.get("Counter_Value", 0) or 0)
            a[1] += 1
for (name, grid, ctr), (s, n) in sorted(acc.items()):
    print("%-50s grid %-8s %-12s mean %12.1f over %d dispatches" % (name, grid, ctr, s / n, n))
if jout:
    tab = {"kernels": {}}
    if os.path.exists(jout):
        with open(jout) as f:
            tab = json.load(f)
    for (name, grid, ctr), (s, n) in sorted(acc.items()):
        e = tab["kernels"].setdefault("%s|%s" % (name, grid), {})
        e[ctr] = s / n
        e["dispatches"] = n
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import sol_amd
    h = sol_amd._build._source_hash()
    if tab.get("source_hash") not in (None, h):
        tab["kernels"] = {k: v for k, v in tab["kernels"].items() if False}      # other build: start over
        for (name, grid, ctr), (s, n) in sorted(acc.items()):
            e = tab["kernels"].setdefault("%s|%s" % (name, grid), {})
            e[ctr] = s / n
            e["dispatches"] = n
    tab["source_hash"] = h
    tab["unit"] = "as reported by rocprofv3 (FETCH_SIZE / WRITE_SIZE: KB per dispatch; gfx950: FETCH_SIZE under-counts wide reads 2x)"
    with open(jout, "w") as f:
        json.dump(tab, f, indent=1, sort_keys=True)
