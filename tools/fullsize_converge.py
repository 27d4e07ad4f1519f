"""Probe: config 4 at full size through the reference's stage schedule (trajInitFromSurfel -> trajInitFromLVIdata) to an LM termination.
Usage: python tools/fullsize_converge.py [scale] [legacy]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lvi-exc_amd"))
import lvx  # noqa: E402
import synth  # noqa: E402


def ext_err(x, xt, N):
    b = 7 * N
    out = {}
    for name, o in (("lidar", 16), ("cam", 24)):
        q, qt = x[b + o:b + o + 4], xt[b + o:b + o + 4]
        d = synth.qmul(q, synth.qconj(qt))
        out[name + "_rad"] = 2 * np.arctan2(np.linalg.norm(d[:3]), abs(d[3]))
        out[name + "_m"] = np.linalg.norm(x[b + o + 4:b + o + 7] - xt[b + o + 4:b + o + 7])
    return out


def stages(P, x0, verbose=0, legacy=False):
    N = P["n_knots"]
    TAU = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU
    S1 = lvx.LOCK_CAM_Q | lvx.LOCK_CAM_P | lvx.LOCK_LANDMARKS | TAU
    Q = dict(P)
    Q["rep_lm"], Q["rep_uv"], Q["rep_t0"] = P["rep_lm"][:0], P["rep_uv"][:0], P["rep_t0"][:0]
    x = x0
    log = []
    for name, prob, locks, iters in (("surfel", Q, S1, 30), ("lvi", P, TAU, 80)):
        g = lvx.Context(0)
        lvx.load_problem(g, prob, locks)
        if legacy:
            g.set_switch("FORCE_LEGACY", 1)
        t0 = time.perf_counter()
        x, s = g.lm_solve(x, max_iterations=iters, verbose=verbose)
        dt = time.perf_counter() - t0
        g.close()
        e = ext_err(x, P["state_true"], N)
        log.append((name, s, dt, e))
        print("%-7s it %3d ok %3d term %-20s cost %.6e -> %.6e  %.3f s  %s" % (name, s["iterations"], s["successful_steps"], s["termination"], s["initial_cost"], s["final_cost"], dt,
                                                                                   " ".join("%s=%.3e" % kv for kv in e.items())), flush=True)
    return x, log


if __name__ == "__main__":
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    P = synth.make_bench_problem(seed=4, n_imu=200_000 // scale, n_surfel=1_000_000 // scale, n_reproj=50_000 // scale, n_planes=max(50, 2000 // scale))
    if "sparse" in sys.argv:
        P = synth.make_bench_problem(seed=4, n_imu=200_000 // scale, n_surfel=1_000_000 // scale, n_reproj=50_000 // scale, n_planes=max(50, 2000 // scale), tracks="sparse")
    x, log = stages(P, P["state0"], verbose=1 if "verbose" in sys.argv else 0, legacy="legacy" in sys.argv)
    np.save(os.path.join(ROOT, "gpurun_out", "converged_%s_%d.npy" % ("legacy" if "legacy" in sys.argv else "mfma", scale)), x)
