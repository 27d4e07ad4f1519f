#!/usr/bin/env python
"""bench.py — residual+Jacobian+normal-equation evaluations per second of the LVI-ExC calibration solve on MI355X.

One "step" = one pass of the hot path over the whole synthetic problem of BASELINE.json config 4
(1 M LiDAR surfel + 200 k IMU samples [= 200 k gyro + 200 k accel blocks] + 50 k ORB reprojection blocks):
every residual block's residual, analytic Jacobian, Huber scaling, and the J^T J / J^T r assembly into the
structured normal equations, with the state and all measurement arrays already resident in HBM.
N > 1: one independent calibration sequence per GPU (weak scaling, seed 40 + rank), no data-path collective;
the only exchange is one RCCL all-reduce per step of the shared-calibration block of the normal equations.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant kernel
(LiDAR surfel, measured with HIP events on the library's own stream) and `cpu_baseline` (the oracle — a
restatement of the reference's per-block autodiff evaluator — timed on a bounded sample on the host cores).
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
# HBM bytes per launch of k_family_mfma<SurfAcc> at config 4 from the PMC counters (FETCH_SIZE 32.8 MB + WRITE_SIZE 59.0 MB per dispatch with
# normal equations; KiB -> bytes; FETCH_SIZE is not doubled: these are 8-byte strided reads, not the 16 B/lane streams the guide's x2
# correction was calibrated on).  Reads: 56 B of row inputs per block (t, point, row-ordered plane) = 56 MB would be the cold figure, the
# counter sees 33 MB (the rest hits the 256 MB Infinity Cache from the previous pass); writes: the accumulator flushes (one atomic per touched
# band / border entry per workgroup, 1954 workgroups) — the register-spill scratch of the earlier rounds (232 MB) is gone.
PMC_TRAFFIC_BYTES = 91.8e6
PMC_SOURCE = "profiles/r01j_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per dispatch)"
BYTES_PER_EVAL = {"imu": 32, "surfel": 60, "reproj": 60}
FLOPS_PER_EVAL = {"imu": 4e3, "surfel": 9e3, "reproj": 11e3}


def secondary_metrics(ctx, P, lo):
    """Side measurements (not the headline value): one full LM iteration incl. the linear solve, and the upstream kernels of
    BASELINE.json configs 1-2 / the surfel-association metric, inputs resident in HBM, wall-clock over repeated calls."""
    import lvx
    import synth
    sec = {}
    try:
        import time
        ctx.lm_solve(P["state0"], max_iterations=1)      # untimed: rocBLAS / rocSOLVER handle creation and kernel loading (~160 ms, once per process)
        t0 = time.perf_counter()
        _, sm = ctx.lm_solve(P["state0"], max_iterations=3)
        dt = time.perf_counter() - t0
        it = max(1, sm["iterations"])
        sec["lm_iteration"] = {"ms_per_iteration": 1e3 * dt / it, "iterations": it, "Mevals_per_s_incl_solve": lo["n_blocks"] * (it + sm["successful_steps"] + 1) / dt / 1e6,
                               "note": "evaluate(+J^T J) + block-cyclic-reduction solve + candidate cost evaluation per iteration; host-synchronised"}
    except Exception as e:   # noqa: BLE001
        sec["lm_iteration"] = {"error": str(e)[:200]}
    try:
        scan, p4, bmin, bmax = synth.make_assoc_problem(seed=5, H=16, W=1800, n_planes=2000)
        t = lvx.upstream_bench(ctx, "surfel_assoc", (scan, p4, bmin, bmax))
        sec["surfel_assoc"] = {"Mpts_per_s": scan.shape[0] * scan.shape[1] / t / 1e6, "planes": 2000, "points": scan.shape[0] * scan.shape[1], "ms": 1e3 * t,
                               "hbm_frac": (20.0 * scan.shape[0] * scan.shape[1] + 80.0 * 2000) / t / 1e9 / HBM_PEAK_GBS}
        cloud = synth.make_voxel_cloud(seed=2, n=100_000)
        t = lvx.upstream_bench(ctx, "voxel_build", (cloud, 0.5))
        sec["voxel_build"] = {"Mpts_per_s": len(cloud) / t / 1e6, "points": len(cloud), "ms": 1e3 * t}
        t = lvx.upstream_bench(ctx, "voxel_lookup7", synth.rigid_move(cloud))
        sec["voxel_lookup7"] = {"Mqueries_per_s": len(cloud) / t / 1e6, "ms": 1e3 * t, "hbm_frac": 100.0 * len(cloud) / t / 1e9 / HBM_PEAK_GBS}
        pts = synth.make_vlp16_sweep(seed=1)
        import time
        lvx.scan_register(ctx, pts, 16, 0.3)
        t0 = time.perf_counter()
        for _ in range(10):
            lvx.scan_register(ctx, pts, 16, 0.3)
        t = (time.perf_counter() - t0) / 10
        sec["scan_registration"] = {"ms_per_sweep": 1e3 * t, "points": len(pts), "note": "host buffers in/out (PCIe inclusive), reference budget 100 ms per sweep"}
    except Exception as e:   # noqa: BLE001
        sec["upstream_error"] = str(e)[:200]
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


def cpu_baseline(P, seconds_budget=20.0):
    """Oracle (port of the reference's per-block dual-number evaluator) on a bounded sample of the same workload."""
    from oracle import oracle as O
    import lvx
    import synth
    ns, ni, nr = 20000, 4000, 1000   # same 20 : 4(+4) : 1 mix as the full problem
    rng = np.random.default_rng(0)
    si = np.sort(rng.choice(len(P["surf_t"]), ns, replace=False))
    ii = np.sort(rng.choice(len(P["t_imu"]), ni, replace=False))
    lm_sel = np.arange(nr // 10)
    rmask = np.isin(P["rep_lm"], lm_sel)
    Q = dict(P)
    Q.update(surf_pt=P["surf_pt"][si], surf_t=P["surf_t"][si], surf_plane=P["surf_plane"][si], t_imu=P["t_imu"][ii], gyro=P["gyro"][ii], acc=P["acc"][ii],
             rep_lm=P["rep_lm"][rmask], rep_uv=P["rep_uv"][rmask], rep_t0=P["rep_t0"][rmask])
    o = O.Oracle()
    lvx.load_problem(o, Q, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    cores = usable_cores()
    o.set_threads(cores)
    blocks = o.num_blocks
    o.evaluate(P["state0"], jac=True)  # warm
    t0 = time.perf_counter()
    reps = 0
    while True:
        o.evaluate(P["state0"], jac=True)
        reps += 1
        if time.perf_counter() - t0 > seconds_budget or reps >= 50:
            break
    dt = time.perf_counter() - t0
    return {"value": blocks * reps / dt / 1e6, "unit": "Mevals/s", "cores": cores, "kind": "port",
            "sample": "%d surfel + %d gyro + %d accel + %d reprojection blocks x %d passes: residual + stride-4 dual-number Jacobian per block "
                      "(cost profile of ceres::DynamicAutoDiffCostFunction), OpenMP over blocks; J^T J not included" % (ns, ni, ni, int(rmask.sum()), reps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--small", action="store_true", help="1/10 size problem (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the LM-iteration and upstream-kernel side measurements")
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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

    scale = 10 if args.small else 1
    n_surf, n_imu, n_rep = 1_000_000 // scale, 200_000 // scale, 50_000 // scale
    P = synth.make_bench_problem(seed=4 if world == 1 else 40 + rank, n_imu=n_imu, n_surfel=n_surf, n_reproj=n_rep, n_planes=2000 // scale)
    ctx = lvx.Context(local_rank)
    lvx.load_problem(ctx, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    lo = ctx.layout()
    ctx.set_state(P["state0"])
    if world > 1:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)   # evaluation, export and the all-reduce share one stream
    what = lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ
    nbd = lo["border_ld"]
    red = torch.zeros(nbd * nbd + nbd + 2, dtype=torch.float64, device="cuda") if world > 1 else None

    def step():
        ctx.evaluate_resident(what)
        if world > 1:
            ctx.export_border(red.data_ptr())   # shared-calibration block of J^T J, J^T r, cost
            dist.all_reduce(red)

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    # live HIP-event timing of the dominant kernel (surfel) over the timed region; the other kernels' durations come from a separate
    # profiled run below (an event pair around every launch costs ~5 % of a pass).  LVX_BENCH_NOPROF=1: no events at all (replayed graph).
    live = not os.environ.get("LVX_BENCH_NOPROF")
    ctx.set_profiling(live, only=lvx.FAM_SURFEL)
    ctx.kernel_ms()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ms, launches = ctx.kernel_ms()
    ctx.set_profiling(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    cost = ctx.evaluate_resident(lvx.EVAL_COST, want_cost=True)
    joint = None
    if world > 1 and not args.no_secondary:
        # side measurement (outside the timed region): LM iterations of the JOINT problem — shared rig extrinsics, one sequence per GPU;
        # per iteration ONE all-reduce of the 14 x 14 reduced system + a few scalars (lvx_lm_solve_shared)
        try:
            import sharded
            sj = P["state0"].copy()
            Nk = lo["n_knots"]
            ext = torch.from_numpy(sj[7 * Nk + 16:7 * Nk + 32].copy()).cuda()
            dist.broadcast(ext, src=0)                      # the shared extrinsics start from rank 0's guess
            sj[7 * Nk + 16:7 * Nk + 32] = ext.cpu().numpy()
            dist.barrier(); torch.cuda.synchronize()
            tj = time.perf_counter()
            _, smj = ctx.lm_solve_shared(sj, sharded.dist_all_reduce(dist, torch.device("cuda", local_rank)), max_iterations=3)
            torch.cuda.synchronize(); dist.barrier()
            tj = time.perf_counter() - tj
            joint = {"ms_per_iteration": 1e3 * tj / max(1, smj["iterations"]), "iterations": smj["iterations"], "initial_cost": smj["initial_cost"], "final_cost": smj["final_cost"],
                     "note": "joint LM over %d sequences with shared extrinsics: evaluate + private elimination per GPU, one 211-double all-reduce + 4 scalar reductions per iteration" % world}
        except Exception as e:   # noqa: BLE001
            joint = {"error": str(e)[:300]}

    blocks = lo["n_blocks"]
    value = blocks * world * args.steps / elapsed / 1e6
    out = {
        "metric": "M residual+Jacobian evals/s per GN iter", "value": value, "unit": "Mevals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "config4-full-LVI: %d surfel + %d IMU samples (gyro+accel blocks) + %d ORB reprojection blocks, %d knots @ dt=0.02; "
                               "step = residuals + analytic Jacobians + Huber + J^T J/J^T r assembly (no linear solve)" % (n_surf, n_imu, len(P["rep_lm"]), lo["n_knots"]),
                   "blocks_per_step_per_gpu": int(blocks), "n_tangent": lo["n_tangent"], "bandwidth": lo["bandwidth"], "n_border": lo["n_border"],
                   "parallelism": "sequence-per-gpu x%d" % world, "cost": cost},
    }
    if rank == 0:
        k = lvx.FAM_SURFEL
        surf_ms = ms[k] / max(1, launches[k])
        alg_bytes = BYTES_PER_EVAL["surfel"] * n_surf
        alg_flops = FLOPS_PER_EVAL["surfel"] * n_surf
        achieved = alg_flops / (surf_ms * 1e-3) / 1e12 if surf_ms > 0 else 0.0
        out["roofline"] = {"bound": "mfma", "kernel": "k_family_mfma<SurfAcc>", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                           "traffic": PMC_TRAFFIC_BYTES if scale == 1 else None, "traffic_source": PMC_SOURCE, "avg_launch_ms": surf_ms,
                           "algorithmic_flops_per_launch": alg_flops, "algorithmic_bytes_per_launch": alg_bytes,
                           "hbm": {"achieved_GBps": alg_bytes / (surf_ms * 1e-3) / 1e9 if surf_ms > 0 else 0.0, "peak_GBps": HBM_PEAK_GBS},
                           "note": "fused residual + analytic Jacobian + FP64-MFMA J^T J kernel of the LiDAR surfel family (1 M of the 1.45 M blocks); FP64 matrix = FP64 vector peak = 78.6 TFLOP/s on MI355X; "
                                   "9 kFLOP / 60 B per block (SURVEY.md 8d) => compute bound, the HBM figure is reported for completeness; duration from HIP events on the kernel's own stream around every launch of the timed region; the LiDAR kernels run first and alone, the IMU and "
                                   "reprojection kernels concurrently after them (solo durations of all kernels: kernel_ms_solo)"}
        # durations of every kernel in the pass's own schedule: a separate run with an event pair around every launch (outside the timed region);
        # the surfel entry is the live one
        ctx.set_profiling(True); ctx.kernel_ms()
        for _ in range(5):
            ctx.evaluate_resident(what)   # rank 0 only: no collective here
        ctx.synchronize()
        msa, la = ctx.kernel_ms()
        ctx.set_profiling(False)
        out["kernel_ms"] = {lvx.KERNEL_NAMES[i]: msa[i] / max(1, la[i]) for i in range(len(msa)) if la[i]}
        out["kernel_ms"][lvx.KERNEL_NAMES[k]] = surf_ms
        if world == 1:   # solo durations: the same step with every family kernel on one stream (outside the timed region)
            ctx.set_switch("SERIAL", 1)
            ctx.set_profiling(True); ctx.kernel_ms()
            for _ in range(5):
                step()
            ctx.synchronize()
            ms1, l1 = ctx.kernel_ms()
            ctx.set_profiling(False)
            ctx.set_switch("SERIAL", 0)
            out["kernel_ms_solo"] = {lvx.KERNEL_NAMES[i]: ms1[i] / max(1, l1[i]) for i in range(len(ms1)) if l1[i]}
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary_metrics(ctx, P, lo)
        if joint is not None:
            out["secondary"] = {"joint_lm_iteration": joint}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(P)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
