import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "lvi-exc_amd")):
    sys.path.insert(0, p)
import numpy as np
import lvx, synth
P = synth.make_bench_problem(seed=4, n_surfel=0, n_reproj=0)
locks = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU | lvx.LOCK_LIDAR_Q | lvx.LOCK_LIDAR_P | lvx.LOCK_CAM_Q | lvx.LOCK_CAM_P
for nd in (0, -1):
    g = lvx.Context(0)
    g.set_switch("SOLVER_ND", nd)
    lvx.load_problem(g, P, locks)
    g.lm_solve(P["state0"], max_iterations=1)
    t0 = time.perf_counter()
    x, s = g.lm_solve(P["state0"], max_iterations=8)
    dt = time.perf_counter() - t0
    lo = g.layout()
    print("config 3 (IMU only) SOLVER_ND %2d: %.3f ms per iteration, %d iterations, %s, band %d x %d, %d separators, final cost %.6e" % (nd, 1e3 * dt / max(1, s["iterations"]), s["iterations"], s["termination"], lo["n_band"], lo["bandwidth"], lo["solver_separators"], s["final_cost"]))
    g.close()
