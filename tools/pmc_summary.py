#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch, per kernel.
usage: pmc_summary.py DIR [DIR ...]   (every *counter_collection.csv under the directories is read)"""
import csv, glob, os, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
res = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            res[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"])
for k in sorted(acc):
    g = res[k]
    print(f"{k[:110]}\n    grid {g[0]} wg {g[1]} lds {g[2]} scratch {g[3]} vgpr {g[4]} agpr {g[5]} sgpr {g[6]}")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f} min={min(v):16.1f} max={max(v):16.1f}")
