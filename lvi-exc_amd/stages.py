"""The reference's solve stages over the lvx binding (host-side mirror of lvi-exc_amd/host/lvx_loaders.hpp::StageLocks and lvx_calibrate.hpp).

Lock masks: which Lock* calls TrajectoryManagerLVI makes before building each stage's estimator (src/lvi_exc/src/core/trajectory_manager_lvi.cpp):
  initialSO3TrajWithGyro (:43-62)    SO3 spline + one orientation prior; R3 absent, biases locked                                      <= 30 iterations
  trajInitFromSurfel (:311-351)      gyro + accel + surfel blocks; camera and landmarks constant                                        <= 30
  trajInitFromLVIdata (:138-195)     + reprojection blocks, everything free (lvi.yaml: lock_traj_lidar_in_2nd_stage false)              <= 80
  trajInitFromLVIdata + lm_splane (:197-257)  + camera-landmark-to-surfel blocks; trajectory and LiDAR locked (lock_traj_lidar_in_3rd_stage) <= 80
Time offsets stay locked unless opt_time_offset (lvi.yaml:32).  A fresh problem is built for every stage, as the reference does
(make_shared<SplitTrajEstimator> per stage).
"""
import time

import numpy as np

import lvx
import synth


def stage_locks(stage, opt_time_offset=False):
    tau = 0 if opt_time_offset else (lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    if stage == "SO3FromGyro":
        return lvx.LOCK_R3 | lvx.LOCK_ACC_BIAS | lvx.LOCK_GYRO_BIAS | lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU
    if stage == "TrajFromSurfel":
        return lvx.LOCK_CAM_Q | lvx.LOCK_CAM_P | lvx.LOCK_CAM_TAU | lvx.LOCK_LANDMARKS | (0 if opt_time_offset else lvx.LOCK_LIDAR_TAU)
    if stage == "TrajFromLVI":
        return tau
    if stage == "TrajFromLVILandmarksOnly":
        return tau | lvx.LOCK_TRAJ | lvx.LOCK_LIDAR_Q | lvx.LOCK_LIDAR_P
    raise ValueError(stage)


TAU = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU
STAGE_SURFEL = stage_locks("TrajFromSurfel")
# (name, lock mask, max iterations, with reprojection blocks) of the two stages a config-4 style problem (fixed surfel list) goes through
STAGES = (("trajInitFromSurfel", STAGE_SURFEL, 30, False), ("trajInitFromLVIdata", stage_locks("TrajFromLVI"), 80, True))


def without_reprojection(P):
    Q = dict(P)
    Q["rep_lm"], Q["rep_uv"], Q["rep_t0"] = P["rep_lm"][:0], P["rep_uv"][:0], P["rep_t0"][:0]
    return Q


def extrinsic_errors(x, x_ref, n_knots):
    """(rad, m) of the lidar and camera extrinsics of state x against x_ref."""
    b = 7 * n_knots
    out = {}
    for name, o in (("lidar", 16), ("cam", 24)):
        d = synth.qmul(x[b + o:b + o + 4], synth.qconj(x_ref[b + o:b + o + 4]))
        out[name + "_rad"] = float(2 * np.arctan2(np.linalg.norm(d[:3]), abs(d[3])))
        out[name + "_m"] = float(np.linalg.norm(x[b + o + 4:b + o + 7] - x_ref[b + o + 4:b + o + 7]))
    return out


def run_stages_gpu(P, x0, legacy=False, verbose=0, device=0):
    """Returns (state, [(stage, summary, seconds)]) of the two stages on the GPU through lvx_lm_solve."""
    x, log = np.array(x0, dtype=np.float64), []
    for name, locks, iters, with_rep in STAGES:
        g = lvx.Context(device)
        lvx.load_problem(g, P if with_rep else without_reprojection(P), locks)
        if legacy:
            g.set_switch("FORCE_LEGACY", 1)
        t0 = time.perf_counter()
        x, s = g.lm_solve(x, max_iterations=iters, verbose=verbose)
        log.append((name, s, time.perf_counter() - t0))
        g.close()
    return x, log


def cost_at(P, x, device=0):
    g = lvx.Context(device)
    lvx.load_problem(g, P, TAU)
    c = g.evaluate(x, residuals=False)["cost"]
    g.close()
    return c
