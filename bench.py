#!/usr/bin/env python
"""bench.py — residual+Jacobian+normal-equation evaluations per second of the LVI-ExC calibration solve on MI355X.

One "step" = one pass of the hot path over the whole synthetic problem of BASELINE.json config 4
(1 M LiDAR surfel + 200 k IMU samples [= 200 k gyro + 200 k accel blocks] + 50 k ORB reprojection blocks):
every residual block's residual, analytic Jacobian, Huber scaling, and the J^T J / J^T r assembly into the
structured normal equations, with the state and all measurement arrays already resident in HBM.
N > 1: one independent calibration sequence per GPU (weak scaling, seed 40 + rank), no data-path collective;
the only exchange is one RCCL all-reduce per step of the shared-calibration block of the normal equations.

The second half of BASELINE.json's metric — surfel association Mpts/s at 1/2/4/8 GPUs — is measured in the same run (`secondary.surfel_assoc`):
every rank associates its own shard of scans against the same surfel map (lvx_surfel_assoc_batch_d) and the per-point flags are all-gathered.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the time-dominant kernel
(measured with HIP events on the library's own stream) plus every family's and the whole pass's fraction, and `cpu_baseline`
(the oracle — a restatement of the reference's per-block autodiff evaluator — timed on a bounded sample on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lvi-exc_amd"))

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6      # MI355X FP64: vector 78.6 TFLOP/s = matrix (v_mfma_f64_16x16x4_f64) 78.6 TFLOP/s
# algorithmic bytes per residual block, SURVEY.md §8(d)
# roofline.traffic = HBM bytes per launch of the dominant kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes).  It is NOT measured by this run (a PMC pass
# serialises the kernels): tools/profile_round.sh writes profiles/latest_traffic.json next to the counter summary, with the SHA-256 of the kernel sources it was taken
# from; bench.py reports the figure only while that hash still matches the tree, otherwise `traffic` is null (a constant in this file would go stale silently).
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "latest_traffic.json")
KERNEL_SOURCES = ("lvx_eval.hip", "lvx_resid.h", "lvx_math.h", "lvx_ctx.h")


def kernel_sources_sha():
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "lvi-exc_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def measured_traffic(kernel_key):
    """(bytes per launch, source string) of `kernel_key` ('imu' | 'surfel' | ...) from the committed profile, or (None, why)."""
    try:
        d = json.load(open(TRAFFIC_JSON))
    except Exception:   # noqa: BLE001
        return None, "no profiles/latest_traffic.json"
    if d.get("sources_sha256") != kernel_sources_sha():
        return None, "profiles/latest_traffic.json was taken from other kernel sources (hash mismatch): not reported"
    k = d.get("kernels", {}).get(kernel_key)
    if not k:
        return None, "kernel not in profiles/latest_traffic.json"
    return k["fetch_bytes"] + k["write_bytes"], "%s: %.1f MB fetched + %.1f MB written per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, mean over dispatches with normal equations)" % (
        d.get("source", "profiles/"), k["fetch_bytes"] / 1e6, k["write_bytes"] / 1e6)


def measured_executed(kernel_key, avg_launch_ms):
    """What the dominant kernel EXECUTES per launch, from the SQ counter passes of the committed profile (same source-hash rule as the traffic): MFMA instructions x 2 048
    FLOP (v_mfma_f64_16x16x4_f64), vector instructions (an upper bound of 128 FLOP each if every one were a 64-lane FP64 FMA; moves, integer and LDS address arithmetic are
    in that count), and the share of the chip's matrix-pipe cycles the MFMAs occupy (64 cycles each, 256 CUs x 4 SIMDs at 2.4 GHz)."""
    try:
        d = json.load(open(TRAFFIC_JSON))
    except Exception:   # noqa: BLE001
        return None
    if d.get("sources_sha256") != kernel_sources_sha():
        return None
    q = d.get("sq", {}).get(kernel_key)
    if not q:
        return None
    simd_cycles = avg_launch_ms * 1e-3 * 2.4e9 * 1024
    return {"mfma_insts": q["insts_mfma"], "mfma_flops": q["insts_mfma"] * 2048.0, "mfma_TFLOPs": q["insts_mfma"] * 2048.0 / (avg_launch_ms * 1e-3) / 1e12,
            "matrix_pipe_frac": q["insts_mfma"] * 64.0 / simd_cycles, "valu_insts": q["insts_valu"], "valu_issue_frac": q["insts_valu"] * 4.0 / simd_cycles,
            "valu_flops_upper_bound": q["insts_valu"] * 128.0, "source": d.get("sq_source", "profiles/"),
            "note": "per launch; matrix_pipe_frac = MFMA instructions x 64 cycles / (launch duration x 1024 SIMDs x 2.4 GHz); valu_issue_frac = vector instructions x 4 cycles "
                    "(a 64-lane wavefront on a 16-lane SIMD) over the same denominator; the algorithmic fraction above divides SURVEY 8d's FLOP count by the FP64 peak instead"}


BYTES_PER_EVAL = {"imu": 32, "surfel": 60, "reproj": 60}
FLOPS_PER_EVAL = {"imu": 4e3, "surfel": 9e3, "reproj": 11e3}


def assoc_metric(ctx, world, rank, scans_per_gpu=64):
    """surfel-assoc Mpts/s: every rank associates its own shard of scans (weak scaling: scans_per_gpu each) against one surfel map, flags all-gathered
    (scan-level kernels shard by scan, no arithmetic collective)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    import lvx
    import sharded
    import synth
    scan, p4, bmin, bmax = synth.make_assoc_problem(seed=5, H=16, W=1800, n_planes=2000)
    H, W, P = scan.shape[0], scan.shape[1], len(p4)
    n_scans = scans_per_gpu * world
    lo, hi = sharded.scan_shard(n_scans, rank, world)
    dev = torch.device("cuda", torch.cuda.current_device())
    local = torch.from_numpy(np.ascontiguousarray(scan, np.float32)).to(dev).unsqueeze(0).repeat(hi - lo, 1, 1, 1).contiguous()
    local[:, :, :, 0] += 1e-3 * torch.arange(lo, hi, device=dev, dtype=torch.float32).view(-1, 1, 1)     # the scans differ
    pl = torch.from_numpy(np.concatenate([p4.ravel(), bmin.ravel(), bmax.ravel()])).to(dev)
    flags = torch.empty((hi - lo, H * W), dtype=torch.int32, device=dev)

    ctx._ck(ctx._l.lvx_surfel_map_prepare_d(ctx._h, C.c_int(P), C.c_void_p(pl.data_ptr())))   # setSurfelMap: once per data association, not per scan

    def step():
        ctx._ck(ctx._l.lvx_surfel_assoc_batch_d(ctx._h, C.c_int(hi - lo), C.c_int(H), C.c_int(W), C.c_void_p(local.data_ptr()), C.c_int(P), C.c_void_p(pl.data_ptr()), C.c_double(0.05), C.c_int(2),
                                                C.c_void_p(flags.data_ptr())))
        return sharded.all_gather_scan_hits(dist, flags, n_scans) if world > 1 else flags   # hits only: ~1/8 of the dense flags' bytes per rank
    for _ in range(3):
        step()
    ctx.synchronize(); torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        out = step()
    ctx.synchronize(); torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tm = torch.tensor([dt], dtype=torch.float64, device=dev); dist.all_reduce(tm, op=dist.ReduceOp.MAX); dt = float(tm.item())
    pts = n_scans * H * W
    res = {"Mpts_per_s": pts * reps / dt / 1e6, "scans": n_scans, "scans_per_gpu": scans_per_gpu, "points_per_scan": H * W, "planes": P, "ms_per_call": 1e3 * dt / reps,
           "associated_points": int((out >= 0).sum().item()), "scaling": "weak",
           "roofline": {"bound": "hbm", "achieved": 20.0 * pts * reps / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 20.0 * pts * reps / dt / 1e9 / HBM_PEAK_GBS,
                        "note": "20 B per point algorithmic (16 B read + 4 B flag, SURVEY 8d); the kernels are bound by dependent lookups (cell -> list -> box) and the bitmask atomics, not by HBM; "
                                "the surfel map's grid is built once before the timed calls (lvx_surfel_map_prepare_d), as setSurfelMap is in the reference"}}
    if rank == 0:   # the reference's OpenMP loop over planes (restated in the oracle) on the host cores, one scan
        from oracle import oracle as O
        cores = usable_cores()
        O.surfel_assoc_omp(scan, p4, bmin, bmax, 0.05, 2, cores)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 3.0:
            O.surfel_assoc_omp(scan, p4, bmin, bmax, 0.05, 2, cores); n += 1
        res["cpu_openmp"] = {"Mpts_per_s": H * W * n / (time.perf_counter() - t0) / 1e6, "cores": cores, "kind": "port", "sample": "%d passes over one scan, OpenMP over planes as surfel_association.cpp:122" % n}
    ctx._ck(ctx._l.lvx_surfel_map_release(ctx._h))   # the table `pl` dies with this frame
    return res


def secondary_metrics(ctx, P, lo):
    """Side measurements (not the headline value): one full LM iteration incl. the linear solve, the converged two-stage solve, a pass with
    free sensor time offsets, and the upstream kernels of BASELINE.json configs 1-2, inputs resident in HBM, wall-clock over repeated calls."""
    import lvx
    import synth
    sec = {}
    try:
        ctx.lm_solve(P["state0"], max_iterations=1)      # untimed: rocBLAS handle creation and kernel loading (~160 ms, once per process)
        t0 = time.perf_counter()
        _, sm = ctx.lm_solve(P["state0"], max_iterations=8)      # (8 iterations: the solve's initial evaluation, 0.6 ms, is a twelfth of the time, not a quarter as with 3)
        dt = time.perf_counter() - t0
        it = max(1, sm["iterations"])
        sec["lm_iteration"] = {"ms_per_iteration": 1e3 * dt / it, "iterations": it, "Mevals_per_s_incl_solve": lo["n_blocks"] * (it + sm["successful_steps"] + 1) / dt / 1e6,
                               "note": "evaluate(+J^T J) + landmark elimination + band solve (leaves + separators elimination where the band's column profile allows it, else the uniform block chain) + candidate cost evaluation per iteration; host-synchronised",
                               "solver_plan": {k: ctx.layout()[k] for k in ("solver_separators", "solver_leaves", "bandwidth", "n_band")}}
        try:   # the same iterations on the uniform block chain (block cyclic reduction with b = bandwidth: the solver of rounds 2-5), for the record
            ctx.set_switch("SOLVER_ND", -1)
            ctx.lm_solve(P["state0"], max_iterations=1)
            t0 = time.perf_counter()
            _, smu = ctx.lm_solve(P["state0"], max_iterations=8)
            sec["lm_iteration"]["uniform_chain_ms_per_iteration"] = 1e3 * (time.perf_counter() - t0) / max(1, smu["iterations"])
        finally:
            ctx.set_switch("SOLVER_ND", 0)
        # the headline counts evaluations per second of the evaluation pass alone; this is what someone who ITERATES gets: blocks per second of whole Gauss-Newton / LM
        # iterations (one evaluation with normal equations + the exact SPARSE_SCHUR-equivalent step each)
        sec["gn_iteration_Mevals_per_s"] = lo["n_blocks"] * it / dt / 1e6
    except Exception as e:   # noqa: BLE001
        sec["lm_iteration"] = {"error": str(e)[:200]}
    try:   # config 4 "to convergence": the reference's stage schedule to a Ceres termination (lvi-exc_amd/stages.py, tests/test_gpu_converge.py)
        import stages as cs
        x, log = cs.run_stages_gpu(P, P["state0"])
        sec["converged_solve"] = {"seconds": sum(dt for _, _, dt in log), "stages": [{"stage": n, "iterations": s["iterations"], "termination": s["termination"], "final_cost": s["final_cost"], "seconds": dt}
                                                                                        for n, s, dt in log],
                                  "distance_to_truth": cs.extrinsic_errors(x, P["state_true"], P["n_knots"]),
                                  "note": "trajInitFromSurfel (<= 30 it) then trajInitFromLVIdata (<= 80 it) from the 3 deg / 3 cm perturbed start, wall time incl. problem upload"}
    except Exception as e:   # noqa: BLE001
        sec["converged_solve"] = {"error": str(e)[:200]}
    try:   # BASELINE config 3: IMU-only normal equations, 200 k samples (gyroscope + accelerometer block each), its own CPU time
        from oracle import oracle as O
        Q = dict(P)
        Q.update(surf_pt=P["surf_pt"][:0], surf_t=P["surf_t"][:0], surf_plane=P["surf_plane"][:0], rep_lm=P["rep_lm"][:0], rep_uv=P["rep_uv"][:0], rep_t0=P["rep_t0"][:0],
                 cs_lm=P["cs_lm"][:0], cs_plane=P["cs_plane"][:0])
        g = lvx.Context(0)
        lvx.load_problem(g, Q, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU | lvx.LOCK_LIDAR_Q | lvx.LOCK_LIDAR_P | lvx.LOCK_CAM_Q | lvx.LOCK_CAM_P | lvx.LOCK_LANDMARKS)
        g.set_state(P["state0"])
        for _ in range(3):
            g.evaluate_resident(lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ)
        g.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.evaluate_resident(lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ)
        g.synchronize()
        t3 = (time.perf_counter() - t0) / 20
        nb3 = g.layout()["n_blocks"]
        g.close()
        o = O.Oracle()
        lvx.load_problem(o, Q, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU | lvx.LOCK_LIDAR_Q | lvx.LOCK_LIDAR_P | lvx.LOCK_CAM_Q | lvx.LOCK_CAM_P | lvx.LOCK_LANDMARKS)
        cores = usable_cores(); o.set_threads(cores)
        o.evaluate_products(P["state0"])
        t0 = time.perf_counter(); nrep = 0
        while time.perf_counter() - t0 < 4.0:
            o.evaluate_products(P["state0"]); nrep += 1
        tc3 = (time.perf_counter() - t0) / nrep
        sec["config3_imu_only"] = {"ms_per_step": 1e3 * t3, "blocks": int(nb3), "Mevals_per_s": nb3 / t3 / 1e6, "frac_fp64_peak": FLOPS_PER_EVAL["imu"] * nb3 / t3 / 1e12 / FP64_PEAK_TFLOPS,
                                   "cpu": {"Mevals_per_s": o.num_blocks / tc3 / 1e6, "ms_per_pass": 1e3 * tc3, "cores": cores, "kind": "port",
                                           "sample": "%d full passes over the 200 k-sample IMU-only problem: stride-4 dual-number Jacobians + J^T r + diag(J^T J), OpenMP over blocks" % nrep},
                                   "note": "BASELINE config 3: gyroscope + accelerometer blocks only (k_clear + k_imu_own + fold), whole pass wall time, state resident"}
    except Exception as e:   # noqa: BLE001
        sec["config3_imu_only"] = {"error": str(e)[:200]}
    try:   # free LiDAR / camera time offsets (the reference's opt_time_offset_ stages): one more global column on the fused kernels
        g = lvx.Context(0)
        lvx.load_problem(g, P, 0)
        g.set_state(P["state0"])
        for _ in range(2):
            g.evaluate_resident(lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ)
        g.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.evaluate_resident(lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ)
        g.synchronize()
        dtf = (time.perf_counter() - t0) / 5
        g.close()
        sec["free_time_offsets"] = {"ms_per_step": 1e3 * dtf, "Mevals_per_s": lo["n_blocks"] / dtf / 1e6, "note": "lock mask 0: both sensor time offsets free (not the headline configuration); fused time-offset column in the LiDAR, camera-surfel and reprojection kernels"}
    except Exception as e:   # noqa: BLE001
        sec["free_time_offsets"] = {"error": str(e)[:200]}
    try:
        from oracle import oracle as O
        cloud = synth.make_voxel_cloud(seed=2, n=100_000)
        t = lvx.upstream_bench(ctx, "voxel_build", (cloud, 0.5))
        t0 = time.perf_counter(); O.voxel_build(cloud, 0.5); tc = time.perf_counter() - t0
        sec["voxel_build"] = {"Mpts_per_s": len(cloud) / t / 1e6, "points": len(cloud), "ms": 1e3 * t, "hbm_frac": (36.0 * len(cloud)) / t / 1e9 / HBM_PEAK_GBS,
                              "hbm_frac_is_at": "BASELINE config 2 size (100 k points): launch-latency bound, see larger_maps for the saturating size",
                              "cpu": {"Mpts_per_s": len(cloud) / tc / 1e6, "cores": 1, "kind": "port", "sample": "one build (the reference's applyFilter is serial)"}}
        t = lvx.upstream_bench(ctx, "voxel_lookup7", synth.rigid_move(cloud))
        sec["voxel_lookup7"] = {"Mqueries_per_s": len(cloud) / t / 1e6, "ms": 1e3 * t, "hbm_frac": 100.0 * len(cloud) / t / 1e9 / HBM_PEAK_GBS,
                                "hbm_frac_is_at": "BASELINE config 2 size (100 k queries); voxel_build.larger_maps.*.lookup7 has 400 k and 4 M"}
        # the sizes the pipeline really builds: the map cloud of a DataAssociation round (~410 k points) and a 4 M-point map; same local density (tiled config-2 cloud).
        # One launch chain without a host hop, replayed as a HIP graph; algorithmic traffic 36 B per point + 268 B per occupied leaf
        sizes = {}
        for tiles in (4, 40):
            big = synth.tile_voxel_cloud(cloud, tiles)
            tb = lvx.upstream_bench(ctx, "voxel_build", (big, 0.5), reps=10)
            nl = ctx.voxel_info()["n_leaves"]
            by = 36.0 * len(big) + 268.0 * nl
            tq = lvx.upstream_bench(ctx, "voxel_lookup7", synth.rigid_move(big), reps=10)
            sizes["%d_points" % len(big)] = {"ms": 1e3 * tb, "Mpts_per_s": len(big) / tb / 1e6, "leaves": int(nl), "GBps": by / tb / 1e9, "hbm_frac": by / tb / 1e9 / HBM_PEAK_GBS,
                                             "lookup7": {"ms": 1e3 * tq, "Mqueries_per_s": len(big) / tq / 1e6, "GBps": 100.0 * len(big) / tq / 1e9, "hbm_frac": 100.0 * len(big) / tq / 1e9 / HBM_PEAK_GBS}}
        sec["voxel_build"]["larger_maps"] = sizes
        big_key = max(sizes, key=lambda k: int(k.split("_")[0]))
        sec["upstream_hbm_fractions"] = {   # every streaming upstream kernel at the BASELINE config size AND at a size that saturates the chip, side by side
            "voxel_build": {"config2_100k_points": sec["voxel_build"]["hbm_frac"], big_key: sizes[big_key]["hbm_frac"]},
            "voxel_lookup7": {"config2_100k_queries": sec["voxel_lookup7"]["hbm_frac"], big_key: sizes[big_key]["lookup7"]["hbm_frac"]},
            "note": "fraction of 8 TB/s by ALGORITHMIC bytes (36 B/pt + 268 B/leaf; 100 B/query, SURVEY 8d); the config sizes are bound by launch count / dependent round trips, not by HBM"}
        pts = synth.make_vlp16_sweep(seed=1)
        lvx.scan_register(ctx, pts, 16, 0.3)
        t0 = time.perf_counter()
        for _ in range(10):
            lvx.scan_register(ctx, pts, 16, 0.3)
        t = (time.perf_counter() - t0) / 10
        t0 = time.perf_counter(); nrep = 0
        while time.perf_counter() - t0 < 1.0:
            O.scan_register(pts, 16, 0.3); nrep += 1
        tcs = (time.perf_counter() - t0) / nrep
        sec["scan_registration"] = {"ms_per_sweep": 1e3 * t, "points": len(pts), "Mpts_per_s": len(pts) / t / 1e6, "hbm_frac": 28.0 * len(pts) / t / 1e9 / HBM_PEAK_GBS,
                                    "cpu": {"ms_per_sweep": 1e3 * tcs, "Mpts_per_s": len(pts) / tcs / 1e6, "cores": 1, "kind": "port", "sample": "%d sweeps (the reference's laserCloudHandler is serial)" % nrep},
                                    "note": "host buffers in/out (PCIe inclusive), reference budget 100 ms per sweep; 28 B per point algorithmic"}
        import torch
        batched = {}
        for S in (1, 16, 64):   # lvx_scan_register_batch_d: S sweeps per call, points resident on the device, results stay there (counts come back)
            sw = [synth.make_vlp16_sweep(seed=1 + (k % 4)) for k in range(S)]
            off = np.concatenate([[0], np.cumsum([len(q) for q in sw])]).astype(np.int32)
            pd = torch.from_numpy(np.concatenate(sw).view(np.uint8).reshape(-1)).to("cuda")
            lvx.scan_register_batch_d(ctx, pd.data_ptr(), off, 16, 0.3)
            t0 = time.perf_counter()
            for _ in range(10):
                lvx.scan_register_batch_d(ctx, pd.data_ptr(), off, 16, 0.3)
            tb = (time.perf_counter() - t0) / 10
            batched["%d_sweeps" % S] = {"ms_per_call": 1e3 * tb, "us_per_sweep": 1e6 * tb / S, "Mpts_per_s": int(off[-1]) / tb / 1e6, "hbm_frac": 28.0 * int(off[-1]) / tb / 1e9 / HBM_PEAK_GBS}
        sec["scan_registration"]["batched_device_resident"] = batched
        # real VLP-16 ranges come in 2 mm steps: equal curvatures inside a sector are then common and the kernel reproduces libstdc++'s std::sort order of equal keys
        # with a restated introsort on ONE lane per sector (lvx_stdsort.h) — the continuous-noise sweeps above never take that path, these do
        sw = [synth.make_vlp16_sweep(seed=1 + (k % 4), range_quantum=0.002) for k in range(64)]
        off = np.concatenate([[0], np.cumsum([len(q) for q in sw])]).astype(np.int32)
        pd = torch.from_numpy(np.concatenate(sw).view(np.uint8).reshape(-1)).to("cuda")
        lvx.scan_register_batch_d(ctx, pd.data_ptr(), off, 16, 0.3)
        t0 = time.perf_counter()
        for _ in range(10):
            lvx.scan_register_batch_d(ctx, pd.data_ptr(), off, 16, 0.3)
        tq = (time.perf_counter() - t0) / 10
        sec["scan_registration"]["batched_device_resident_quantised_2mm"] = {"sweeps": 64, "ms_per_call": 1e3 * tq, "us_per_sweep": 1e6 * tq / 64, "Mpts_per_s": int(off[-1]) / tq / 1e6,
                                                                              "hbm_frac": 28.0 * int(off[-1]) / tq / 1e9 / HBM_PEAK_GBS}
    except Exception as e:   # noqa: BLE001
        sec["upstream_error"] = str(e)[:200]
    try:   # the reference's ONLY published benchmark (src/ndt_omp/README.md:8-41, apps/align.cpp on its two scans): NDT registration, 0.1 m VoxelGrid, resolution 1.0, identity guess
        from oracle import ndt_align as NA
        from oracle import oracle as O
        gold = os.path.join(ROOT, "tests", "golden")
        tgt = O.voxelgrid_xyzi(np.load(os.path.join(gold, "ndt_data_251370668.npz"))["xyzi"], 0.1)
        src = O.voxelgrid_xyzi(np.load(os.path.join(gold, "ndt_data_251371071.npz"))["xyzi"], 0.1)
        readme = {7: {"fitness": 0.214205, "ms_1_thread": 139.433, "ms_8_threads": 63.1442}, 1: {"fitness": 0.208511, "ms_1_thread": 34.6418, "ms_8_threads": 17.2353}}
        nd = {"points": {"target": len(tgt), "source": len(src)}, "note": "pclomp::NormalDistributionsTransform::align (setInputTarget = voxel covariance grid at 1.0 m, then the Newton / "
              "More-Thuente loop) + getFitnessScore; GPU: lvx_voxel_build + lvx_ndt_align + lvx_ndt_fitness with host buffers in and out; README = Core i7-6700K, "
              "src/ndt_omp/README.md:18-26,33-41; the fitness of the README run is reproduced to 4.5 % / 7.9 % by the restated default loop and to 1.1 % at its fixed point "
              "(tests/test_ndt_align_oracle.py: the statistic moves 0.8 % per milliradian and the default loop stops 0.12 m short of the optimum)"}
        for search in (7, 1):
            lvx.voxel_build(ctx, tgt, 1.0, fetch=False)
            lvx.ndt_align(ctx, src, search=search)
            t0 = time.perf_counter()
            for _ in range(10):
                r = lvx.ndt_align(ctx, src, search=search)
            ta = (time.perf_counter() - t0) / 10
            t0 = time.perf_counter()
            for _ in range(10):
                lvx.voxel_build(ctx, tgt, 1.0, fetch=False); ctx.voxel_info()
            tv = (time.perf_counter() - t0) / 10
            fg = lvx.ndt_fitness(ctx, src, r["final_transformation"], tgt)
            a = NA.NdtAligner(tgt, 1.0, search)
            t0 = time.perf_counter(); a.align(src); tc = time.perf_counter() - t0
            nd["DIRECT%d" % search] = {"align_ms": 1e3 * ta, "target_grid_ms": 1e3 * tv, "iterations": r["iterations"], "evaluations": r["n_evaluations"], "fitness": fg,
                                        "readme": readme[search], "speedup_vs_readme_8_threads": readme[search]["ms_8_threads"] / (1e3 * ta),
                                        "fitness_vs_readme": fg / readme[search]["fitness"],
                                        "cpu": {"align_ms": 1e3 * tc, "fitness": a.fitness(), "iterations": a.nr_iterations, "cores": 1, "kind": "port",
                                                "sample": "one align of the oracle's restated loop (scalar C per-point arithmetic, Python loop around it)"}}
        sec["ndt_align"] = nd
    except Exception as e:   # noqa: BLE001
        sec["ndt_align"] = {"error": str(e)[:200]}
    try:   # one DataAssociation round of the stage driver, device-resident (lvx_data_association): 57 scans x 16 x 450 points of synth.make_sequence
        import ctypes as C
        S = synth.make_sequence(seed=50)
        g = lvx.Context(0)
        g.set_spline(S["t0"], S["dt"], S["n_knots"])
        raw = np.zeros(S["scans"].shape, dtype=lvx.POINT_XYZIT)
        for k in ("x", "y", "z", "timestamp"):
            raw[k] = S["scans"][k]
        nsc = len(raw)
        g._ck(g._l.lvx_set_scans(g._h, C.c_int(nsc), C.c_int(S["H"]), C.c_int(S["W"]), raw.ctypes.data_as(C.c_void_p)))
        st = np.ascontiguousarray(S["state0"], np.float64)
        npl, npt = C.c_int32(0), C.c_int32(0)
        call = lambda: g._ck(g._l.lvx_data_association(g._h, st.ctypes.data_as(C.c_void_p), C.c_double(S["t_map"]), None, C.byref(npl), C.byref(npt)))
        for _ in range(3):   # round 1 = the four-stop chain, round 2 sizes the one-stop chain's buffers
            call()
        t0 = time.perf_counter()
        for _ in range(20):
            call()
        tda = (time.perf_counter() - t0) / 20
        g.close()
        sec["data_association"] = {"ms": 1e3 * tda, "scans": nsc, "points": int(raw.size), "surfels": int(npl.value), "surfel_points": int(npt.value), "Mpts_per_s": raw.size / tda / 1e6,
                                   "note": "de-skew of every scan into the map frame + voxel covariance grid of the map cloud + surfel extraction + association of every scan + chronological SurfelPoint "
                                           "emission, one call, raw scans resident on the device (lvi_initialize_surfel_orb.cpp:1180-1201); rounds after the first run over the capacities of the "
                                           "round before with ONE host stop (DESIGN 3.3)"}
    except Exception as e:   # noqa: BLE001
        sec["data_association"] = {"error": str(e)[:200]}
    return sec


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (os.cpu_count() reports the host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, n)


class Emitter:
    """Prints the JSON line exactly once.  The side measurements after the timed region (joint solve over ranks, association all-gather, LM legs)
    run under a watchdog: a collective that never returns must not cost the headline line — on timeout rank 0 prints what it has and every rank exits."""

    def __init__(self, rank):
        import threading
        self.rank, self.out, self.done, self.lock, self.timer = rank, None, False, threading.Lock(), None

    def emit(self):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.rank == 0 and self.out is not None:
                print(json.dumps(self.out), flush=True)

    def arm(self, seconds):
        import threading

        def fire():
            if self.out is not None:
                self.out.setdefault("secondary", {})["watchdog"] = "side measurements did not finish within %d s; line emitted by the watchdog" % seconds
            self.emit()
            os._exit(0)
        self.timer = threading.Timer(seconds + (0 if self.rank == 0 else 10), fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()


def cpu_baseline(P, seconds_budget=12.0):
    """The reference's CPU path restated (oracle/), timed on the host cores on a bounded sample of the same workload: mode (i) per-block stride-4 dual numbers
    (the cost profile of ceres::DynamicAutoDiffCostFunction) — the headline `value` — and mode (ii) closed-form Jacobians + block products."""
    from oracle import oracle as O
    import lvx
    ns, ni, nr = 200000, 40000, 10000   # 1/5 of the problem, same 20 : 4(+4) : 1 mix
    rng = np.random.default_rng(0)
    si = np.sort(rng.choice(len(P["surf_t"]), min(ns, len(P["surf_t"])), replace=False))
    ii = np.sort(rng.choice(len(P["t_imu"]), min(ni, len(P["t_imu"])), replace=False))
    lm_sel = np.arange(max(1, nr // 10))
    rmask = np.isin(P["rep_lm"], lm_sel)
    Q = dict(P)
    Q.update(surf_pt=P["surf_pt"][si], surf_t=P["surf_t"][si], surf_plane=P["surf_plane"][si], t_imu=P["t_imu"][ii], gyro=P["gyro"][ii], acc=P["acc"][ii],
             rep_lm=P["rep_lm"][rmask], rep_uv=P["rep_uv"][rmask], rep_t0=P["rep_t0"][rmask])
    o = O.Oracle()
    lvx.load_problem(o, Q, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    cores = usable_cores()
    o.set_threads(cores)
    blocks = o.num_blocks
    o.set_block_products(True)        # every block's dense J^T J upper triangle is formed too (what mode (ii) below does): the two legs price the same work
    o.evaluate_products(P["state0"])  # warm
    t0 = time.perf_counter()
    reps = 0
    while True:
        o.evaluate_products(P["state0"])   # residuals + dual-number Jacobians + J^T r + the block's J^T J products, OpenMP over the blocks
        reps += 1
        if time.perf_counter() - t0 > seconds_budget or reps >= 50:
            break
    dt = time.perf_counter() - t0
    out = {"value": blocks * reps / dt / 1e6, "unit": "Mevals/s", "cores": cores, "kind": "port",
           "sample": "%d surfel + %d gyro + %d accel + %d reprojection blocks x %d passes (1/5 of the workload): residual + stride-4 dual-number Jacobian per block "
                     "(cost profile of ceres::DynamicAutoDiffCostFunction) + J^T r + the block's dense J^T J upper triangle (formed, folded into a checksum: no sparse assembly), "
                     "OpenMP over blocks, g++ -O3 -msse4.2" % (len(si), len(ii), len(ii), int(rmask.sum()), reps)}
    try:   # mode (ii): closed-form Jacobians + every block's J^T J / J^T r products
        n2, _ = O.analytic_pass(o, P["state0"], cores)
        t0 = time.perf_counter(); reps2 = 0
        while True:
            O.analytic_pass(o, P["state0"], cores); reps2 += 1
            if time.perf_counter() - t0 > seconds_budget or reps2 >= 200:
                break
        dt2 = time.perf_counter() - t0
        out["optimised"] = {"value": n2 * reps2 / dt2 / 1e6, "unit": "Mevals/s", "cores": cores, "kind": "port",
                            "sample": "same sample x %d passes: closed-form Jacobians on the group (host build of the residual headers) + the block's J^T J / J^T r products, "
                                      "OpenMP over blocks — BASELINE.md 3 mode (ii)" % reps2}
    except Exception as e:   # noqa: BLE001
        out["optimised"] = {"error": str(e)[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--small", action="store_true", help="1/10 size problem (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the LM-iteration and upstream-kernel side measurements")
    ap.add_argument("--shard-size", choices=("full", "eighth"), default="full",
                    help="N > 1 (SURVEY 8d config 5): every rank's sequence at full config-4 size (default, weak scaling) or at 1/8 size; the other variant is reported under secondary")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import lvx
    import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("LVX_BENCH_BACKEND", "nccl")   # "gloo": functional check of the N > 1 path with several ranks on one GPU
    if torch.cuda.is_available() and backend != "nccl":
        local_rank %= torch.cuda.device_count()
    # LVX_BENCH_FORCE_DIST=1: take the N > 1 code path with WORLD_SIZE = 1 (process group, exported border block, all-reduce, in-library RCCL, joint LM) — the readiness
    # check of tests/test_gpu_bench_ranks.py on a one-GPU box: every line the 8-GPU run executes runs once before that run
    dist_on = world > 1 or bool(os.environ.get("LVX_BENCH_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")

    scale = (10 if args.small else 1) * (8 if args.shard_size == "eighth" else 1)
    n_surf, n_imu, n_rep = 1_000_000 // scale, 200_000 // scale, 50_000 // scale
    P = synth.make_bench_problem(seed=4 if world == 1 else 40 + rank, n_imu=n_imu, n_surfel=n_surf, n_reproj=n_rep, n_planes=max(8, 2000 // scale))
    ctx = lvx.Context(local_rank)
    lvx.load_problem(ctx, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    lo = ctx.layout()
    ctx.set_state(P["state0"])
    if dist_on:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)   # evaluation, export and the all-reduce share one stream
    what = lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ
    nbd = lo["border_ld"]
    red = torch.zeros(nbd * nbd + nbd + 2, dtype=torch.float64, device="cuda") if dist_on else None
    transport = {"inlib": False}   # the per-step all-reduce: torch.distributed (RCCL behind it) or the library's own communicator (lvx_rccl_allreduce_d)

    def step():
        ctx.evaluate_resident(what)
        if dist_on:
            ctx.export_border(red.data_ptr())   # shared-calibration block of J^T J, J^T r, cost
            if transport["inlib"]:
                ctx.rccl_allreduce(red.data_ptr(), red.numel())
            else:
                dist.all_reduce(red)

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    # live HIP-event timing of the dominant kernel (surfel) over the timed region; the other kernels' durations come from a separate
    # profiled run below (an event pair around every launch costs ~5 % of a pass).  LVX_BENCH_NOPROF=1: no events at all (replayed graph).
    live = not os.environ.get("LVX_BENCH_NOPROF")
    # which kernel is time-dominant: a short run with an event pair around every launch (untimed), then the timed region carries events around that kernel only
    live_k = lvx.FAM_SURFEL
    if live:
        ctx.set_profiling(True); ctx.kernel_ms()
        for _ in range(3):
            ctx.evaluate_resident(what)
        ctx.synchronize()
        ms0, l0 = ctx.kernel_ms()
        ctx.set_profiling(False)
        cand = [i for i in range(len(ms0)) if l0[i] and lvx.KERNEL_NAMES[i] not in ("fold", "clear", "solve", "upstream")]
        if cand:
            live_k = max(cand, key=lambda i: ms0[i] / l0[i])
    ctx.set_profiling(live, only=live_k)
    ctx.kernel_ms()
    def timed_region():
        """EXACTLY args.steps steps between barrier + synchronize on both sides; the maximum over the ranks."""
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    elapsed = timed_region()
    ms, launches = ctx.kernel_ms()
    ctx.set_profiling(False)
    cost = ctx.evaluate_resident(lvx.EVAL_COST, want_cost=True)
    if dist_on and float(red[-1].item()) != 0.0:     # summed device error words of the last step: some rank's sums were incomplete
        raise SystemExit("a rank reported a device-side evaluation error (range / non-unit quaternion / fallback)")
    blocks = lo["n_blocks"]
    value = blocks * world * args.steps / elapsed / 1e6
    out = {
        "metric": "M residual+Jacobian evals/s per GN iter", "value": value, "unit": "Mevals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "config4-full-LVI: %d surfel + %d IMU samples (gyro+accel blocks) + %d ORB reprojection blocks, %d knots @ dt=0.02; "
                               "step = residuals + analytic Jacobians + Huber + J^T J/J^T r assembly (no linear solve)" % (n_surf, n_imu, len(P["rep_lm"]), lo["n_knots"]),
                   "blocks_per_step_per_gpu": int(blocks), "n_tangent": lo["n_tangent"], "bandwidth": lo["bandwidth"], "n_border": lo["n_border"],
                   "locks": "LIDAR_TAU | CAM_TAU (sensor time offsets constant, everything else free: trajInitFromLVIdata with lvi.yaml's opt_time_offset false)",
                   "tracks": P.get("tracks", "orb"), "parallelism": "sequence-per-gpu x%d" % world, "shard_size": args.shard_size, "cost": cost,
                   "allreduce_transport": "torch.distributed all_reduce (backend %s) of the exported border block" % backend if dist_on else "none (one sequence)"},
    }
    em = Emitter(rank)
    em.out = out
    em.arm(300)   # everything below is a side measurement: it must not be able to cost the line
    # ---- the same step over the LIBRARY'S OWN RCCL communicator (lvx_rccl_init / lvx_rccl_allreduce_d).  It runs AFTER the torch-transport measurement above and under the
    # watchdog: a second communicator that cannot be created, or hangs, costs nothing — the line with the torch transport is already in `out`.  When it works its
    # measurement becomes the headline (`value`, `ms_per_step`), the torch-transport one is kept beside it. ----
    rccl_ok = False
    if dist_on and backend == "nccl":
        try:
            uid = torch.tensor(list(ctx.rccl_unique_id()) if rank == 0 else [0] * 128, dtype=torch.uint8, device="cuda")
            dist.broadcast(uid, src=0)
            ctx.rccl_init(bytes(uid.cpu().tolist()), rank, world)
            ok = torch.ones(1, device="cuda")
        except Exception as e:   # noqa: BLE001
            ok = torch.zeros(1, device="cuda")
            out.setdefault("secondary", {})["inlib_rccl_init_error"] = str(e)[:300]
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)       # every rank takes the same transport
        rccl_ok = bool(ok.item() > 0)
        if not rccl_ok:
            try:
                ctx.rccl_finalize()
            except Exception:   # noqa: BLE001
                pass
        if rccl_ok:
            try:
                transport["inlib"] = True
                ref = red.clone()
                for _ in range(max(1, args.warmup)):
                    step()
                ctx.synchronize(); torch.cuda.synchronize()
                same = bool(torch.allclose(red[:-2], ref[:-2], rtol=1e-9, atol=0.0)) if world == 1 else None   # one rank: the reduced block is the exported block
                elapsed_b = timed_region()
                if float(red[-1].item()) != 0.0:
                    raise RuntimeError("device-side evaluation error under the in-library transport")
                hb = {"torch_transport": {"value": value, "ms_per_step": 1e3 * elapsed / args.steps},
                      "inlib_transport": {"value": blocks * world * args.steps / elapsed_b / 1e6, "ms_per_step": 1e3 * elapsed_b / args.steps}}
                if same is not None:
                    hb["single_rank_identity"] = same
                out.setdefault("secondary", {})["headline_transports"] = hb
                elapsed, value = elapsed_b, hb["inlib_transport"]["value"]
                out["value"], out["ms_per_step"] = value, 1e3 * elapsed / args.steps
                out["config"]["allreduce_transport"] = "RCCL inside liblvx on the context's stream (lvx_rccl_init + lvx_rccl_allreduce_d) of the exported border block; torch's beside it under secondary.headline_transports"
            except Exception as e:   # noqa: BLE001
                out.setdefault("secondary", {})["inlib_rccl_step_error"] = str(e)[:300]
            transport["inlib"] = False
    if rank == 0:
        k = live_k
        surf_ms = ms[k] / max(1, launches[k])   # the live-timed (dominant) kernel's mean launch duration inside the timed region
        # durations of every kernel family in the pass's own schedule: a separate run with an event pair around every launch (outside the timed region);
        # the surfel entry is the live one
        ctx.set_profiling(True); ctx.kernel_ms()
        for _ in range(5):
            ctx.evaluate_resident(what)   # rank 0 only: no collective here
        ctx.synchronize()
        msa, la = ctx.kernel_ms()
        ctx.set_profiling(False)
        out["kernel_ms"] = {lvx.KERNEL_NAMES[i]: msa[i] / max(1, la[i]) for i in range(len(msa)) if la[i]}
        if launches[k]:
            out["kernel_ms"][lvx.KERNEL_NAMES[k]] = surf_ms
        live_name = lvx.KERNEL_NAMES[k]
        solo = dict(out["kernel_ms"])
        if world == 1:   # solo durations: the same step with every family kernel on one stream (outside the timed region)
            ctx.set_switch("SERIAL", 1)
            ctx.set_profiling(True); ctx.kernel_ms()
            for _ in range(5):
                step()
            ctx.synchronize()
            ms1, l1 = ctx.kernel_ms()
            ctx.set_profiling(False)
            ctx.set_switch("SERIAL", 0)
            solo = {lvx.KERNEL_NAMES[i]: ms1[i] / max(1, l1[i]) for i in range(len(ms1)) if l1[i]}
            out["kernel_ms_solo"] = solo
        # algorithmic work of every family against its solo duration, and of the whole pass against the step time
        work = {"surfel": (FLOPS_PER_EVAL["surfel"] * n_surf, BYTES_PER_EVAL["surfel"] * n_surf), "gyro": (FLOPS_PER_EVAL["imu"] * n_imu, BYTES_PER_EVAL["imu"] * n_imu),
                "accel": (FLOPS_PER_EVAL["imu"] * n_imu, BYTES_PER_EVAL["imu"] * n_imu), "reproj": (FLOPS_PER_EVAL["reproj"] * len(P["rep_lm"]), BYTES_PER_EVAL["reproj"] * len(P["rep_lm"]))}
        if "accel" not in solo and "gyro" in solo:   # fused IMU kernel (k_imu_own): one launch evaluates the gyroscope AND the accelerometer block of every sample
            solo["imu"] = solo.pop("gyro")
            if "gyro" in out["kernel_ms"]:
                out["kernel_ms"]["imu"] = out["kernel_ms"].pop("gyro")
            if "kernel_ms_solo" in out:
                out["kernel_ms_solo"] = dict(solo)
            work["imu"] = (2 * FLOPS_PER_EVAL["imu"] * n_imu, (BYTES_PER_EVAL["imu"] + 24) * n_imu)   # t 8 + gyro 24 + accel 24 B per sample
            del work["gyro"], work["accel"]
        rep_parts = ["reproj", "reproj_jac", "reproj_obs", "reproj_ref", "reproj_cross", "reproj_lmrows"]   # the fused reprojection path is five kernels
        solo["reproj_all"] = sum(solo.get(k2, 0.0) for k2 in rep_parts)
        fam = {}
        live_fam = {"gyro": "imu" if "imu" in work else "gyro"}.get(live_name, live_name)
        for name, (fl, by) in work.items():
            d = (surf_ms if (name == live_fam and launches[k]) else solo.get("reproj_all" if name == "reproj" else name, 0.0)) * 1e-3
            if d > 0:
                fam[name] = {"ms": 1e3 * d, "TFLOPs": fl / d / 1e12, "frac_fp64_peak": fl / d / 1e12 / FP64_PEAK_TFLOPS, "GBps": by / d / 1e9}
        # the time-dominant KERNEL: a family's duration is one kernel's except for reprojection (five kernels, the longest counts)
        longest = {n: (max(solo.get(k2, 0.0) for k2 in rep_parts) if n == "reproj" else fam[n]["ms"]) for n in fam}
        dominant = max(fam, key=lambda n: longest[n])
        step_s = elapsed / args.steps
        tot_fl = sum(v[0] for v in work.values())
        names = {"surfel": "k_family_mfma<SurfAcc>", "gyro": "k_family_mfma<GyroAcc>", "accel": "k_family_mfma<AccelAcc>", "imu": "k_imu_own (gyroscope + accelerometer blocks fused, owner-computes)",
                 "reproj": "reprojection path (k_reproj_jac + k_family_mfma<RepSideAcc<1>> + k_family_mfma<RepSideAcc<0>> + k_reproj_cross + k_reproj_lmrows)"}
        fl_d, by_d = work[dominant]
        traffic_bytes, traffic_src = measured_traffic(dominant)
        out["roofline"] = {"bound": "mfma", "kernel": names[dominant], "achieved": fam[dominant]["TFLOPs"], "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fam[dominant]["frac_fp64_peak"],
                           "traffic": traffic_bytes if scale == 1 else None, "traffic_source": traffic_src,
                           "duration_source": "HIP events on the kernel's own stream around every launch of the timed region" if (dominant == live_fam and launches[k]) else "solo duration (every kernel on one stream), profiled run outside the timed region",
                           "avg_launch_ms": fam[dominant]["ms"], "algorithmic_flops_per_launch": fl_d, "algorithmic_bytes_per_launch": by_d,
                           "hbm": {"achieved_GBps": fam[dominant]["GBps"], "peak_GBps": HBM_PEAK_GBS},
                           "executed": measured_executed(dominant, fam[dominant]["ms"]) if scale == 1 else None,
                           "families": fam,
                           "whole_pass": {"TFLOPs": tot_fl / step_s / 1e12, "frac_fp64_peak": tot_fl / step_s / 1e12 / FP64_PEAK_TFLOPS, "ms": 1e3 * step_s},
                           "note": "time-dominant kernel family of the pass (solo durations; the surfel kernel's from HIP events on its own stream around every launch of the timed region); "
                                   "ALGORITHMIC FP64 work per block from SURVEY.md 8d (surfel 9 k, IMU 4 k, reprojection 11 k FLOP) over the duration — the kernels execute less than that "
                                   "(hoisted hub pose, precomputed control-point pairs); FP64 matrix = FP64 vector peak = 78.6 TFLOP/s on MI355X; `families` has every family's fraction, "
                                   "`whole_pass` the sum over the step time"}
    # ---- side measurements that involve every rank ----
    if dist_on and not args.no_secondary:
        # LM iterations of the JOINT problem — shared rig extrinsics, one sequence per GPU (lvx_lm_solve_shared).  Transport: RCCL inside the library on the context's
        # stream when the job runs on RCCL (no host round trip); the host callback over torch.distributed otherwise, and as the fallback when a second communicator next
        # to torch's cannot be created.  Which one ran is recorded.
        import sharded
        sj = P["state0"].copy()
        Nk = lo["n_knots"]
        sec_all = out.setdefault("secondary", {})
        try:
            ext = torch.from_numpy(sj[7 * Nk + 16:7 * Nk + 32].copy()).cuda()
            dist.broadcast(ext, src=0)                      # the shared extrinsics start from rank 0's guess
            sj[7 * Nk + 16:7 * Nk + 32] = ext.cpu().numpy()
        except Exception as e:   # noqa: BLE001
            sec_all["joint_lm_iteration"] = {"error": str(e)[:300]}
        if "joint_lm_iteration" not in sec_all:
            try:
                ctx.collective_count(reset=True)
                dist.barrier(); torch.cuda.synchronize()
                tj = time.perf_counter()
                xj, smj = ctx.lm_solve_shared(sj, None if rccl_ok else sharded.dist_all_reduce(dist, torch.device("cuda", local_rank)), max_iterations=3)
                torch.cuda.synchronize(); dist.barrier()
                tj = time.perf_counter() - tj
                # the shared extrinsics every rank ended with: they must be THE SAME numbers (each rank solves the same reduced 14 x 14 system from the same reduction)
                ext_all = torch.zeros(world, 16, dtype=torch.float64, device="cuda")       # (a sum of one-hot rows: all_reduce is what both backends offer on device tensors)
                ext_all[rank] = torch.from_numpy(np.ascontiguousarray(xj[7 * Nk + 16:7 * Nk + 32])).cuda()
                dist.all_reduce(ext_all)
                ext_all = ext_all.cpu().numpy()
                ext_spread = float(np.abs(ext_all - ext_all[0]).max())
                sec_all["joint_lm_iteration"] = {"ms_per_iteration": 1e3 * tj / max(1, smj["iterations"]), "iterations": smj["iterations"], "initial_cost": smj["initial_cost"], "final_cost": smj["final_cost"],
                                                 "collectives": int(ctx.collective_count()), "ranks": world, "shared_extrinsics_max_spread_over_ranks": ext_spread,
                                                 "transport": "RCCL inside liblvx on the context's stream (lvx_rccl_init): reduced system packed, reduced and solved on the device" if rccl_ok else
                                                              "host callback (torch.distributed all_reduce of <= 211 doubles per reduction)",
                                                 "note": "joint LM over %d sequences with shared extrinsics: evaluate + private elimination per GPU, reduced 14 x 14 system + decision scalars over the ranks" % world}
            except Exception as e:   # noqa: BLE001
                sec_all["joint_lm_iteration"] = {"error": str(e)[:300]}
            if rccl_ok:
                try:
                    ctx.rccl_finalize()
                except Exception:   # noqa: BLE001
                    pass
        fail_rank = int(os.environ.get("LVX_BENCH_FAIL_RANK", "-1"))
        if 0 <= fail_rank < world:   # readiness check (tests/test_gpu_bench_ranks.py): one rank fails locally (non-unit control quaternion) -> EVERY rank must leave the joint solve together
            try:
                bad = sj.copy()
                if rank == fail_rank:
                    bad[3 * Nk + 4 * (Nk // 2):3 * Nk + 4 * (Nk // 2) + 4] *= 1.5
                code = 0
                try:
                    ctx.lm_solve_shared(bad, sharded.dist_all_reduce(dist, torch.device("cuda", local_rank)), max_iterations=3)
                except lvx.LvxError as e:
                    code = int(e.code)
                codes = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(world)]
                dist.all_gather(codes, torch.tensor([code], dtype=torch.int64, device="cuda"))
                sec_all["joint_failure_vote"] = {"failing_rank": fail_rank, "return_codes": [int(c.item()) for c in codes]}
                ctx.set_state(P["state0"]); ctx.evaluate_resident(what, want_cost=True)   # the failing rank's context remembers its last evaluation's device error (lvx_synchronize): replace it
            except Exception as e:   # noqa: BLE001
                sec_all["joint_failure_vote"] = {"error": str(e)[:300]}
        try:   # SURVEY 8d config 5 asks for both shard sizes: the variant that is not the headline, same step (evaluation + all-reduce of the border block), fewer steps
            oscale = (10 if args.small else 1) * (1 if args.shard_size == "eighth" else 8)
            P2 = synth.make_bench_problem(seed=40 + rank, n_imu=200_000 // oscale, n_surfel=1_000_000 // oscale, n_reproj=50_000 // oscale, n_planes=max(8, 2000 // oscale))
            c2 = lvx.Context(local_rank)
            lvx.load_problem(c2, P2, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
            l2 = c2.layout()
            c2.set_state(P2["state0"])
            c2.set_stream(torch.cuda.current_stream().cuda_stream)
            red2 = torch.zeros(l2["border_ld"] ** 2 + l2["border_ld"] + 2, dtype=torch.float64, device="cuda")

            def step2():
                c2.evaluate_resident(what); c2.export_border(red2.data_ptr()); dist.all_reduce(red2)
            for _ in range(2):
                step2()
            c2.synchronize(); torch.cuda.synchronize(); dist.barrier()
            t2 = time.perf_counter()
            for _ in range(10):
                step2()
            c2.synchronize(); torch.cuda.synchronize(); dist.barrier()
            t2 = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device="cuda"); dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            nb2 = torch.tensor([float(l2["n_blocks"])], dtype=torch.float64, device="cuda"); dist.all_reduce(nb2)
            sec_all["other_shard_size"] = {"shard_size": "full" if args.shard_size == "eighth" else "eighth", "value": float(nb2.item()) * 10 / float(t2.item()) / 1e6, "unit": "Mevals/s",
                                           "ms_per_step": 1e2 * float(t2.item()), "blocks_all_ranks": int(nb2.item())}
            c2.close()
        except Exception as e:   # noqa: BLE001
            sec_all["other_shard_size"] = {"error": str(e)[:300]}
    if not args.no_secondary:
        try:
            assoc = assoc_metric(ctx, world, rank)      # every rank takes part (all-gather of the flags)
        except Exception as e:   # noqa: BLE001
            assoc = {"error": str(e)[:300]}
            print("[bench rank %d] surfel_assoc: %s" % (rank, str(e)[:300]), file=sys.stderr, flush=True)
        out.setdefault("secondary", {})["surfel_assoc"] = assoc
    # ---- rank 0 alone ----
    if rccl_ok:
        try:
            ctx.rccl_finalize()
        except Exception:   # noqa: BLE001
            pass
    if rank == 0:
        if world == 1 and not dist_on and not args.no_secondary:
            sec = secondary_metrics(ctx, P, lo)
            sec.update(out.get("secondary", {}))
            out["secondary"] = sec
        if not args.no_cpu_baseline and world == 1 and not dist_on:
            out["cpu_baseline"] = cpu_baseline(P)
    em.disarm()
    em.emit()
    ctx.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
