"""Probe: config-4 pass with free LiDAR / camera time offsets (lock mask 0) vs locked, fused vs per-segment TAU kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lvi-exc_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
torch.cuda.init()
import lvx, synth

P = synth.make_bench_problem(seed=4)
for name, locks, legacy in (("locked", lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU, 0), ("lidar tau free", lvx.LOCK_CAM_TAU, 0), ("both free", 0, 0), ("both free, per-segment kernels", 0, 1)):
    g = lvx.Context(0)
    lvx.load_problem(g, P, locks)
    g.set_switch("TAU_LEGACY", legacy)
    g.set_state(P["state0"])
    for _ in range(3):
        g.evaluate_resident(lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ)
    c = g.evaluate_resident(lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ, want_cost=True)
    t0 = time.perf_counter()
    for _ in range(10):
        g.evaluate_resident(lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ)
    g.synchronize()
    dt = (time.perf_counter() - t0) / 10
    gg, dd = g.gradient()
    N = P["n_knots"]
    print("%-34s %.3f ms per pass, cost %.9e, fallback %d, g[tau_L] %.6e g[tau_C] %.6e" % (name, 1e3 * dt, c, g.layout()["exact_fallback"], gg[6 * N + 14], gg[6 * N + 21]))
    g.close()
