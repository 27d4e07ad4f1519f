"""Deterministic mode at full size (config 4): checksums of three passes and the time per pass.  python tools/det_check.py"""
import sys, time
sys.path.insert(0, "/root/repo/lvi-exc_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, synth, lvx
P = synth.make_bench_problem(seed=4)
for det in (1, 0):
    g = lvx.Context(0)
    if det: g.set_switch("DETERMINISTIC", 1)
    lvx.load_problem(g, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    g.set_state(P["state0"])
    out = []
    for _ in range(3):
        c = g.evaluate_resident(want_cost=True); out.append((c,) + g.normal_eq_checksum())
    g.synchronize(); t0 = time.perf_counter()
    for _ in range(5): g.evaluate_resident()
    g.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("deterministic" if det else "default", "ms per pass %.3f" % (1e3 * dt), "repeatable" if all(o == out[0] for o in out) else "differs", out[0][0])
    g.close()
