"""Per-evaluation latency of small problems (launch-bound regime): tools/small_latency.py"""
import sys, time
sys.path.insert(0, '/root/repo/lvi-exc_amd'); sys.path.insert(0, '/root/repo')
import numpy as np, synth, lvx
for name, kw in (("tiny", dict(duration=2.0, n_surfel=600, n_planes=12, n_landmarks=30)), ("small", dict(duration=10.0, n_surfel=20000, n_planes=40, n_landmarks=200)),
                 ("medium", dict(duration=60.0, n_surfel=200000, n_planes=200, n_landmarks=1000))):
    P = synth.make_problem(seed=5, **kw)
    g = lvx.Context(0); lvx.load_problem(g, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    g.set_state(P["state0"])
    for _ in range(5): g.evaluate_resident()
    g.synchronize()
    t = time.perf_counter(); n = 200
    for _ in range(n): g.evaluate_resident()
    g.synchronize()
    dt = (time.perf_counter() - t) / n
    lo = g.layout()
    print("%-7s blocks %8d  %.1f us per evaluate (async queue)  %.2f M evals/s" % (name, lo["n_blocks"], dt * 1e6, lo["n_blocks"] / dt / 1e6))
    t = time.perf_counter(); n = 50
    for _ in range(n): g.evaluate_resident(want_cost=True)
    dt = (time.perf_counter() - t) / n
    print("%-7s                  %.1f us per evaluate + cost readback (synchronous)" % (name, dt * 1e6))
    g.close()
