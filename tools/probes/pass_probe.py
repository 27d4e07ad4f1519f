"""Probe: per-kernel HIP-event times of one config-4 evaluation pass (LVX_LIB selects a variant library; REP_FUSED / other switches from the environment).
Usage: python tools/probes/pass_probe.py [n_reproj] [passes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "lvi-exc_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
torch.cuda.init()
import lvx, synth

n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P = synth.make_bench_problem(seed=4, n_reproj=n_rep)
g = lvx.Context(0)
lvx.load_problem(g, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
g.set_state(P["state0"])
what = lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ
for _ in range(3):
    g.evaluate_resident(what)
g.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    g.evaluate_resident(what)
g.synchronize()
print("pass %.4f ms (%d blocks, lib %s)" % (1e3 * (time.perf_counter() - t0) / passes, g.layout()["n_blocks"], os.environ.get("LVX_LIB", "liblvx.so")))
g.set_profiling(True); g.kernel_ms()
for _ in range(passes):
    g.evaluate_resident(what)
g.synchronize()
ms, n = g.kernel_ms()
print("  ".join("%s %.1f us" % (lvx.KERNEL_NAMES[i], 1e3 * ms[i] / n[i]) for i in range(len(ms)) if n[i]))
