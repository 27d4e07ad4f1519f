#!/usr/bin/env python
"""Idle time between the kernels of the LM iterations in a rocprofv3 rocpd database (kernel-trace of tools/lm_scale_probe.py / lm_iter_probe.py): for every stretch between
two k_bcr_build launches (= one solve + the evaluation that follows), the span, the sum of the kernel durations on the busiest timeline (union of busy intervals) and the
largest gaps with the kernels on either side.   Usage: python tools/rocpd_gaps.py <results.db>"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    marks = [i for i, r in enumerate(rows) if "k_bcr_build" in r[0]]
    if len(marks) < 3:
        print("not enough solves"); return
    for a, b in list(zip(marks[:-1], marks[1:]))[-3:]:
        seg = rows[a:b]
        t0 = seg[0][1]; span = (seg[-1][2] - t0) / 1e3
        busy, cur_end, gaps = 0.0, t0, []
        for i, r in enumerate(seg):
            s, e = r[1], r[2]
            if s > cur_end:
                gaps.append(((s - cur_end) / 1e3, seg[i - 1][0][:40], r[0][:40]))
                busy += (e - s) / 1e3
            elif e > cur_end:
                busy += (e - cur_end) / 1e3
            cur_end = max(cur_end, e)
        gaps.sort(reverse=True)
        print("iteration: %d kernels, span %.1f us, busy %.1f us, idle %.1f us (%.1f %%), %d gaps > 3 us" % (len(seg), span, busy, span - busy, 100 * (span - busy) / span, sum(1 for g in gaps if g[0] > 3)))
        for g in gaps[:8]:
            print("   gap %7.1f us  after %-40s before %s" % g)


if __name__ == "__main__":
    main(sys.argv[1])
