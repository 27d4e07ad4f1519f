"""Probe: wall time per LM iteration of config 4 (evaluate + landmark elimination + BCR solve + candidate evaluation) and the rocprof kernel summary of it.
Usage: python tools/lm_iter_probe.py [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lvi-exc_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
torch.cuda.init()
import lvx, synth

it = int(sys.argv[1]) if len(sys.argv) > 1 else 4
P = synth.make_bench_problem(seed=4)
g = lvx.Context(0)
lvx.load_problem(g, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
g.lm_solve(P["state0"], max_iterations=1)          # warm-up: layout, allocations, graph capture
t0 = time.perf_counter()
x, s = g.lm_solve(P["state0"], max_iterations=it)
dt = time.perf_counter() - t0
print("lm %d iterations: %.3f ms per iteration (%s, accepted %s)" % (s["iterations"], 1e3 * dt / max(1, s["iterations"]), s["termination"], list(s["accepted"])))
print("cost history", s["cost_history"])
g.evaluate(P["state0"], normal_eq=True, dense=False, residuals=False)
for k in range(3):
    t0 = time.perf_counter(); d, m = g.solve_step(1e4, True); dt = time.perf_counter() - t0
print("solve_step: %.3f ms (incl. diag / scaling / host copy of delta), model cost change %.9e" % (1e3 * dt, m))
