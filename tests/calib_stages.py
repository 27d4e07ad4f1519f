"""The reference's stage schedule on a synthetic sequence (test helper; also used by tools/fullsize_converge.py and bench.py's secondary line).

trajInitFromSurfel (src/lvi_exc/src/core/trajectory_manager_lvi.cpp:311-351): gyro + accel + surfel blocks, camera and landmarks constant, <= 30 iterations;
trajInitFromLVIdata (:138-195): + reprojection blocks, everything free (lvi.yaml: lock_traj_lidar_in_2nd_stage false, time offsets locked), <= 80 iterations.
A fresh problem is built for every stage, as the reference does (make_shared<SplitTrajEstimator> per stage)."""
import time

import numpy as np

import lvx
import synth

TAU = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU
STAGE_SURFEL = lvx.LOCK_CAM_Q | lvx.LOCK_CAM_P | lvx.LOCK_LANDMARKS | TAU
STAGES = (("trajInitFromSurfel", STAGE_SURFEL, 30, False), ("trajInitFromLVIdata", TAU, 80, True))


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


def run_stages_oracle(P, x0, threads=None):
    """The same schedule through the CPU oracle and the numpy LM restatement (oracle/lm.py): dense linear algebra, small problems only."""
    from oracle import lm
    from oracle import oracle as O
    x, log = np.array(x0, dtype=np.float64), []
    for name, locks, iters, with_rep in STAGES:
        o = O.Oracle()
        lvx.load_problem(o, P if with_rep else without_reprojection(P), locks)
        if threads:
            o.set_threads(threads)
        free = lm.free_tangent_indices(P["n_knots"], P["n_landmarks"], locks)
        t0 = time.perf_counter()
        x, s = lm.lm_solve(o, x, free, max_iterations=iters, n_knots=P["n_knots"], n_landmarks=P["n_landmarks"])
        log.append((name, s, time.perf_counter() - t0))
    return x, log


def cost_at(P, x, device=0):
    g = lvx.Context(device)
    lvx.load_problem(g, P, TAU)
    c = g.evaluate(x, residuals=False)["cost"]
    g.close()
    return c
