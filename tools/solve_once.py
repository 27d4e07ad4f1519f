"""Probe: one evaluation + ONE LM step solve of config 4 (for the instrumented solver builds: tools/build_kt_bcr.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lvi-exc_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
torch.cuda.init()
import lvx, synth
P = synth.make_bench_problem(seed=4)
g = lvx.Context(0)
lvx.load_problem(g, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
g.evaluate(P["state0"], normal_eq=True, dense=False, residuals=False)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    t0 = time.perf_counter(); d, m = g.solve_step(1e4, True); dt = time.perf_counter() - t0
    print("solve_step: %.3f ms, model cost change %.9e" % (1e3 * dt, m), flush=True)
