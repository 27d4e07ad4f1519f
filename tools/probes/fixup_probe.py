"""Probe: cost of the row-level exact fallback at config 4 — one control-point pair beyond 0.8 rad; per-kernel HIP-event times with and without it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "lvi-exc_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
torch.cuda.init()
import lvx, synth
P = synth.make_bench_problem(seed=4)
N = P["n_knots"]
s = P["state0"].copy(); k = N // 2
s[3 * N + 4 * k:3 * N + 4 * k + 4] = synth.qmul(synth.q_from_rotvec(np.array([0.0, 0.0, 2.0])), s[3 * N + 4 * k:3 * N + 4 * k + 4].copy())
g = lvx.Context(0)
lvx.load_problem(g, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
what = lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ
for name, st in (("plain", P["state0"]), ("wide pair", s)):
    g.set_state(st)
    g.evaluate_resident(what, want_cost=True)
    for _ in range(3):
        g.evaluate_resident(what)
    g.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.evaluate_resident(what)
    g.synchronize()
    dt = (time.perf_counter() - t0) / 20
    g.set_profiling(True); g.kernel_ms()
    for _ in range(10):
        g.evaluate_resident(what)
    g.synchronize()
    ms, n = g.kernel_ms()
    g.set_profiling(False)
    print("%-10s pass %.4f ms, fallback rows %d | " % (name, 1e3 * dt, g.layout()["fallback_rows"]) + "  ".join("%s %.1f" % (lvx.KERNEL_NAMES[i], 1e3 * ms[i] / 10) for i in range(len(ms)) if n[i]))
