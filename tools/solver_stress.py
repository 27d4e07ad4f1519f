"""Stress: the cyclic-reduction step against the sequential band Cholesky over many seeds / bandwidths / trust radii (tools; the fixed cases are tests/test_gpu_solver_sizes.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lvi-exc_amd")):
    sys.path.insert(0, p)
import numpy as np
import lvx, synth
LOCKS = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU
worst, nfb = 0.0, 0
rng = np.random.default_rng(0)
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    views = int(rng.choice([4, 5, 6, 7, 8, 9, 10, 11, 12, 13]))
    cam_rate = float(rng.choice([18.5, 19.0, 20.0, 21.0]))
    n_imu = int(rng.choice([3000, 6000, 12000]))
    P = synth.make_bench_problem(seed=100 + trial, n_imu=n_imu, n_surfel=5 * n_imu, n_reproj=int(0.4 * n_imu), n_planes=60, views_per_lm=views, cam_rate=cam_rate)
    g = lvx.Context(0)
    lvx.load_problem(g, P, LOCKS)
    x = P["state0"] if trial % 2 else P["state_true"]
    radius = float(10.0 ** rng.uniform(1, 7))
    g.evaluate(x, normal_eq=True, dense=False, residuals=False)
    d, m = g.solve_step(radius, True)
    fb = g.layout()["solver_fallbacks"]
    bw = g.layout()["bandwidth"]
    g.set_switch("SOLVER_SEQ", 1)
    g.evaluate(x, normal_eq=True, dense=False, residuals=False)
    ds, ms = g.solve_step(radius, True)
    g.close()
    err = np.abs(d - ds).max() / max(1.0, np.abs(ds).max())
    worst = max(worst, err); nfb += fb
    print("trial %2d views %2d rate %.1f n_imu %5d bw %3d radius %.1e: step diff %.2e, model diff %.1e, fallbacks %d" % (trial, views, cam_rate, n_imu, bw, radius, err, abs(m - ms) / abs(ms), fb), flush=True)
print("worst %.3e, fallbacks %d" % (worst, nfb))
