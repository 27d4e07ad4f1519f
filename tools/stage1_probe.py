"""Probe: LM iterations of the surfel stage of config 4 (trajInitFromSurfel: no reprojection blocks, half-bandwidth 23 -> block size 24, 13 levels of cyclic reduction)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lvi-exc_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
torch.cuda.init()
import lvx, synth, stages
P = stages.without_reprojection(synth.make_bench_problem(seed=4))
g = lvx.Context(0)
lvx.load_problem(g, P, stages.STAGE_SURFEL)
g.lm_solve(P["state0"], max_iterations=1)
t0 = time.perf_counter(); x, s = g.lm_solve(P["state0"], max_iterations=6); dt = time.perf_counter() - t0
print("stage 1: %d iterations, %.3f ms per iteration" % (s["iterations"], 1e3 * dt / s["iterations"]), g.layout()["bandwidth"], flush=True)
