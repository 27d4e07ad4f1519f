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
    big = {}
    for tiles in (4, 40):   # the map cloud of a DataAssociation round (~410 k points), a 4 M-point map: same local density
        bc = synth.tile_voxel_cloud(cloud, tiles)
        tb_ = lvx.upstream_bench(ctx, "voxel_build", (bc, 0.5), max(3, reps // 2))
        nl_ = ctx.voxel_info()["n_leaves"]
        tq_ = lvx.upstream_bench(ctx, "voxel_lookup7", synth.rigid_move(bc), max(3, reps // 2))
        big[len(bc)] = (tb_, nl_, tq_)
    pts = synth.make_vlp16_sweep(seed=1)
    import time
    lvx.scan_register(ctx, pts, 16, 0.3)
    t0 = time.perf_counter()
    for _ in range(reps):
        lvx.scan_register(ctx, pts, 16, 0.3)
    ts = (time.perf_counter() - t0) / reps
    tb = {}
    for S in (1, 16, 64):   # batched scan registration with the points resident on the device: S sweeps per call, results stay on the device (counts come back)
        sw = [synth.make_vlp16_sweep(seed=1 + (k % 4)) for k in range(S)]
        off = np.concatenate([[0], np.cumsum([len(p) for p in sw])]).astype(np.int32)
        pd = torch.from_numpy(np.concatenate(sw).view(np.uint8).reshape(-1)).to("cuda")
        lvx.scan_register_batch_d(ctx, pd.data_ptr(), off, 16, 0.3)
        t0 = time.perf_counter()
        for _ in range(reps):
            lvx.scan_register_batch_d(ctx, pd.data_ptr(), off, 16, 0.3)
        tb[S] = ((time.perf_counter() - t0) / reps, int(off[-1]))
    # next-row kernels (host buffers in and out: read their kernel durations from the rocprofv3 trace of this script)
    src = cloud[::7].copy()
    lvx.voxel_build(ctx, cloud, 1.0, fetch=False)
    for _ in range(reps):
        lvx.ndt_derivatives(ctx, src, src, np.zeros(6))
    Pq = synth.make_problem(seed=41, duration=1.5, n_surfel=0, n_planes=1, n_landmarks=0)
    c2 = lvx.Context(0); lvx.load_problem(c2, Pq, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    rng = np.random.default_rng(3)
    raw = np.zeros(28800 * 8, dtype=lvx.POINT_XYZIT)
    raw["x"], raw["y"], raw["z"] = (rng.uniform(-20, 20, (3, len(raw)))).astype(np.float32)
    raw["timestamp"] = np.sort(rng.uniform(Pq["t_start"], Pq["t_end"], len(raw)))
    for _ in range(reps):
        lvx.undistort(c2, Pq["state_true"], raw, np.array([0, 0, 0, 1.0]), np.zeros(3), True)
    c2.close()
    try:   # one DataAssociation round of the stage driver, device-resident (lvx_data_association)
        import ctypes as C
        S = synth.make_sequence(seed=50)
        g = lvx.Context(0)
        g.set_spline(S["t0"], S["dt"], S["n_knots"])
        rawd = np.zeros(S["scans"].shape, dtype=lvx.POINT_XYZIT)
        for k in ("x", "y", "z", "timestamp"):
            rawd[k] = S["scans"][k]
        g._ck(g._l.lvx_set_scans(g._h, C.c_int(len(rawd)), C.c_int(S["H"]), C.c_int(S["W"]), rawd.ctypes.data_as(C.c_void_p)))
        st_ = np.ascontiguousarray(S["state0"], np.float64)
        npl, npt = C.c_int32(0), C.c_int32(0)
        da = lambda: g._ck(g._l.lvx_data_association(g._h, st_.ctypes.data_as(C.c_void_p), C.c_double(S["t_map"]), None, C.byref(npl), C.byref(npt)))
        for _ in range(3):   # round 1 = the four-stop chain, round 2 sizes the one-stop chain's buffers
            da()
        t0 = time.perf_counter()
        for _ in range(20):
            da()
        tda = (time.perf_counter() - t0) / 20
        print("data_association     : %.1f us per round, %d scans x %d points, %d surfels, %d SurfelPoints" % (1e6 * tda, len(rawd), rawd[0].size, npl.value, npt.value))
        g.close()
    except Exception as e:   # noqa: BLE001
        print("data_association     : error", e)
    n = scan.shape[0] * scan.shape[1]
    print("surfel_assoc 1 scan  : %.1f us  %.0f Mpts/s" % (1e6 * t1, n / t1 / 1e6))
    print("surfel_assoc 16 scans: %.1f us  %.0f Mpts/s" % (1e6 * t16, 16 * n / t16 / 1e6))
    print("surfel_assoc 64 scans: %.1f us  %.0f Mpts/s" % (1e6 * t64, 64 * n / t64 / 1e6))
    print("voxel_build 100k     : %.1f us  %.0f Mpts/s" % (1e6 * tv, len(cloud) / tv / 1e6))
    print("voxel_lookup7 100k   : %.1f us  %.0f Mq/s" % (1e6 * tl, len(cloud) / tl / 1e6))
    for npts, (tb_, nl_, tq_) in big.items():
        by = 36.0 * npts + 268.0 * nl_
        print("voxel_build %8d  : %.1f us  %.0f Mpts/s  %d leaves  %.0f GB/s algorithmic = %.3f of 8 TB/s | lookup7 %.1f us  %.0f Mq/s  %.0f GB/s = %.3f" % (
            npts, 1e6 * tb_, npts / tb_ / 1e6, nl_, by / tb_ / 1e9, by / tb_ / 8e12, 1e6 * tq_, npts / tq_ / 1e6, 100.0 * npts / tq_ / 1e9, 100.0 * npts / tq_ / 8e12))
    print("scan_register 28.8k  : %.1f us per sweep (host buffers in and out)  %.1f Mpts/s" % (1e6 * ts, len(pts) / ts / 1e6))
    for S, (t, npts) in tb.items():
        print("scan_register_batch_d %2d sweeps: %.1f us per call, %.1f us per sweep (%d points resident on the device, counts back)  %.1f Mpts/s" % (S, 1e6 * t, 1e6 * t / S, npts, npts / t / 1e6))
    ctx.close()
