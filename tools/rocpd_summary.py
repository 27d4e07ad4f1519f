#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table (text).
Usage: python tools/rocpd_summary.py <results.db> [> profiles/rNN_xxx.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    stats = {}
    for name, s, e in rows:
        d = (e - s) / 1e3  # ns -> us
        st = stats.setdefault(name, [0, 0.0, 1e30, 0.0])
        st[0] += 1; st[1] += d; st[2] = min(st[2], d); st[3] = max(st[3], d)
    total = sum(v[1] for v in stats.values()) or 1.0
    print("%-90s %8s %12s %12s %12s %12s %7s" % ("KERNEL", "CALLS", "TOTAL_us", "AVG_us", "MIN_us", "MAX_us", "PCT"))
    for name, (n, tot, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print("%-90s %8d %12.1f %12.2f %12.2f %12.2f %6.1f%%" % (name[:90], n, tot, tot / n, mn, mx, 100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1])
