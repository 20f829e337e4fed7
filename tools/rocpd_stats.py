#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace: per-kernel calls, total,
average, min, max duration.  Usage: python tools/rocpd_stats.py results.db [> profiles/xxx.txt]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-78s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, n, s, a, mn, mx in rows:
        print("%-78s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (name[:78], n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    print("total kernel time: %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))


if __name__ == "__main__":
    main(sys.argv[1])
