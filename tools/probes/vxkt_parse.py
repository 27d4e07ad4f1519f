import sys, re
rows = []
for l in sys.stdin:
    m = re.search(r"VXKT b (\d+) hb (\d+) pts (\d+): heads (\d+) endsearch (\d+) sums (\d+) finalize (-?\d+) total (\d+) rt (\d+) (\d+)", l)
    if m: rows.append([int(x) for x in m.groups()])
if not rows: sys.exit("no rows")
t0 = min(r[8] for r in rows)
print("blocks", len(rows), "span (100 MHz ticks -> us):", (max(r[9] for r in rows) - t0) / 100.0)
import statistics as st
for name, idx in (("heads", 3), ("endsearch", 4), ("sums", 5), ("finalize", 6), ("total", 7)):
    v = sorted(r[idx] for r in rows)
    print("%-10s median %8d  p90 %8d  max %8d cycles" % (name, v[len(v)//2], v[int(0.9*len(v))], v[-1]))
st_ = sorted((r[8] - t0) / 100.0 for r in rows); en = sorted((r[9] - t0) / 100.0 for r in rows)
print("start us: median %.1f p90 %.1f max %.1f | end us: median %.1f p90 %.1f max %.1f" % (st_[len(st_)//2], st_[int(0.9*len(st_))], st_[-1], en[len(en)//2], en[int(0.9*len(en))], en[-1]))
d = sorted((r[9] - r[8]) / 100.0 for r in rows)
print("duration us: median %.1f p90 %.1f max %.1f" % (d[len(d)//2], d[int(0.9*len(d))], d[-1]))
late = sorted(rows, key=lambda r: -r[9])[:8]
for r in late: print("late block", r[0], "hb", r[1], "pts", r[2], "start %.1f end %.1f" % ((r[8]-t0)/100.0, (r[9]-t0)/100.0), "heads", r[3], "sums", r[5], "fin", r[6])
