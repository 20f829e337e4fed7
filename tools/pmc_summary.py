#!/usr/bin/env python
"""Mean counter value per (kernel, grid size) from a rocprofv3 --pmc run written with --output-format csv.
Usage: python tools/pmc_summary.py <dir with *counter_collection.csv> [substring filter]"""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
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
