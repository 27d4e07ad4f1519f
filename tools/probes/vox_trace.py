"""Kernel timeline of one voxel covariance build (100 k points by default): run under rocprofv3 --kernel-trace, then `python tools/probes/vox_trace.py --db <results.db>`."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def run(n):
    import importlib
    import torch
    torch.cuda.init()
    lvx = importlib.import_module("lvi-exc_amd.lvx"); synth = importlib.import_module("lvi-exc_amd.synth")
    ctx = lvx.Context()
    cloud = synth.make_voxel_cloud(seed=2, n=min(n, 100000))
    if n > 100000:
        cloud = synth.tile_voxel_cloud(cloud, n // 100000)
        n = len(cloud)
    t = lvx.upstream_bench(ctx, "voxel_build", (cloud, 0.5), 30)
    print("voxel_build %d: %.1f us" % (n, 1e6 * t), ctx.voxel_info())


def show(path):
    import sqlite3
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kt = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t][0]
    rows = cur.execute("select name, start, end from %s order by start" % kt).fetchall()
    inits = [i for i, r in enumerate(rows) if "k_vx_extent" in r[0]]
    a, b = inits[-3], inits[-2]
    t0 = rows[a][1]
    for r in rows[a:b]:
        print("%-90s %8.1f %8.1f" % (r[0][:90], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3))
    print("span %.1f us, to next build %.1f us" % ((max(r[2] for r in rows[a:b]) - t0) / 1e3, (rows[b][1] - t0) / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--db":
        show(sys.argv[2])
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 100000)
