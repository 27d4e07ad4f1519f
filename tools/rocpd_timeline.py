#!/usr/bin/env python
"""Timeline of the last evaluation pass in a rocprofv3 rocpd SQLite database (kernel-trace): start offset / duration / end of every kernel
between the last two k_clear launches.   Usage: python tools/rocpd_timeline.py <results.db> [pass index from the end, default 2]"""
import sqlite3
import sys


def main(path, back=2):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    extra = [c for c in ("queue_id", "stream_id") if c in cols]
    rows = cur.execute("select %s, start, end%s from kernels order by start" % (name_col, "".join(", " + c for c in extra))).fetchall()
    clears = [i for i, r in enumerate(rows) if "k_clear" in r[0]]
    if len(clears) < back + 1:
        print("not enough passes"); return
    a, b = clears[-back - 1], clears[-back]
    t0 = rows[a][1]
    print("%-60s %10s %10s %10s  %s" % ("KERNEL", "start_us", "dur_us", "end_us", " ".join(extra)))
    for r in rows[a:b]:
        print("%-60s %10.1f %10.1f %10.1f  %s" % (r[0][:60], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3, " ".join(str(x) for x in r[3:])))
    print("pass span: %.1f us (to the next pass's k_clear: %.1f us)" % ((max(r[2] for r in rows[a:b]) - t0) / 1e3, (rows[b][1] - t0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
