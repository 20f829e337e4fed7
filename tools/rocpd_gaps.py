#!/usr/bin/env python
"""From a rocprofv3 (rocpd SQLite) kernel trace: per kernel name, the average duration AND the average idle gap between the end of a
dispatch and the start of the next one on the device (what a sum of durations does not show).  Usage: python tools/rocpd_gaps.py results.db"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end from kernels order by start").fetchall()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for (n, s, e), nxt in zip(rows, rows[1:] + [None]):
    a = agg[n[:70]]
    a[0] += 1
    a[1] += e - s
    if nxt is not None and nxt[1] - e < 50000:          # ignore host-side pauses between graph replays
        a[2] += nxt[1] - e
tot_d = sum(a[1] for a in agg.values()); tot_g = sum(a[2] for a in agg.values())
print("%-72s %7s %9s %9s" % ("kernel", "calls", "avg_us", "gap_after_us"))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-72s %7d %9.2f %9.2f" % (n, a[0], a[1] / a[0] / 1e3, a[2] / a[0] / 1e3))
print("sum of durations %.3f ms, sum of gaps %.3f ms" % (tot_d / 1e6, tot_g / 1e6))
