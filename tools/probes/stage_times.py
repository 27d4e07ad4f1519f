"""Where the wall time of the two-stage config-4 solve goes: layout (first evaluation), LM iterations, rejected steps."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "lvi-exc_amd"))
import torch
torch.cuda.init()
import numpy as np
import lvx, synth, stages as cs
P = synth.make_bench_problem(seed=4)
x = np.array(P["state0"], dtype=np.float64)
for name, locks, iters, with_rep in cs.STAGES:
    g = lvx.Context(0)
    t0 = time.perf_counter(); lvx.load_problem(g, P if with_rep else cs.without_reprojection(P), locks); t_load = time.perf_counter() - t0
    t0 = time.perf_counter(); g.evaluate(x, residuals=False); t_first = time.perf_counter() - t0
    t0 = time.perf_counter(); g.evaluate(x, residuals=False); t_second = time.perf_counter() - t0
    t0 = time.perf_counter(); x1, s = g.lm_solve(x, max_iterations=iters); t_lm = time.perf_counter() - t0
    print("%s: load %.1f ms, first evaluate (layout) %.1f ms, second %.1f ms, lm_solve %.1f ms: %d iterations, accepted %s" % (name, 1e3 * t_load, 1e3 * t_first, 1e3 * t_second, 1e3 * t_lm, s["iterations"], list(map(int, s.get("accepted", [])))))
    x = x1
    g.close()
