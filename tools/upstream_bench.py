"""Upstream kernels with device-resident inputs, repeated (for rocprofv3 --kernel-trace / --pmc and for quick timing).
Usage: python tools/upstream_bench.py [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lvi-exc_amd"))
import lvx  # noqa: E402
import synth  # noqa: E402

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    torch.cuda.init()   # before the first lvx context (one HIP runtime per process initialises the device)
    ctx = lvx.Context(0)
    scan, p4, bmin, bmax = synth.make_assoc_problem(seed=5, H=16, W=1800, n_planes=2000)
    t1 = lvx.upstream_bench(ctx, "surfel_assoc", (scan, p4, bmin, bmax), reps)
    batch = np.stack([scan] * 16)
    t16 = lvx.upstream_bench(ctx, "surfel_assoc", (batch, p4, bmin, bmax), reps)
    t64 = lvx.upstream_bench(ctx, "surfel_assoc", (np.stack([scan] * 64), p4, bmin, bmax), reps)
    cloud = synth.make_voxel_cloud(seed=2, n=100_000)
    tv = lvx.upstream_bench(ctx, "voxel_build", (cloud, 0.5), reps)
    tl = lvx.upstream_bench(ctx, "voxel_lookup7", synth.rigid_move(cloud), reps)
    pts = synth.make_vlp16_sweep(seed=1)
    import time
    lvx.scan_register(ctx, pts, 16, 0.3)
    t0 = time.perf_counter()
    for _ in range(reps):
        lvx.scan_register(ctx, pts, 16, 0.3)
    ts = (time.perf_counter() - t0) / reps
    n = scan.shape[0] * scan.shape[1]
    print("surfel_assoc 1 scan  : %.1f us  %.0f Mpts/s" % (1e6 * t1, n / t1 / 1e6))
    print("surfel_assoc 16 scans: %.1f us  %.0f Mpts/s" % (1e6 * t16, 16 * n / t16 / 1e6))
    print("surfel_assoc 64 scans: %.1f us  %.0f Mpts/s" % (1e6 * t64, 64 * n / t64 / 1e6))
    print("voxel_build 100k     : %.1f us  %.0f Mpts/s" % (1e6 * tv, len(cloud) / tv / 1e6))
    print("voxel_lookup7 100k   : %.1f us  %.0f Mq/s" % (1e6 * tl, len(cloud) / tl / 1e6))
    print("scan_register 28.8k  : %.1f us per sweep (host buffers in and out)" % (1e6 * ts))
    ctx.close()
