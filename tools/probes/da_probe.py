"""lvx_data_association: wall time per round on the one-stop chain (capacities of the round before) against the four-stop chain (LVX_DA_SYNC=1), same scans and state."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lvi-exc_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lvx  # noqa: E402
import synth  # noqa: E402

S = synth.make_sequence(seed=50)
raw = np.zeros(S["scans"].shape, dtype=lvx.POINT_XYZIT)
for k in ("x", "y", "z", "timestamp"):
    raw[k] = S["scans"][k]
x = np.ascontiguousarray(S["state0"], np.float64)
for sync in (1, 0, 1, 0):
    g = lvx.Context(0)
    g.set_switch("DA_SYNC", sync)
    g.set_spline(S["t0"], S["dt"], S["n_knots"])
    lvx.set_scans(g, raw, S["H"], S["W"])
    for _ in range(3):
        n = lvx.data_association(g, x, S["t_map"])
    t0 = time.perf_counter()
    for _ in range(20):
        lvx.data_association(g, x, S["t_map"])
    dt = (time.perf_counter() - t0) / 20
    print("%s chain: %.3f ms per round, %s surfels / points, stats %s" % ("four-stop" if sync else "one-stop ", 1e3 * dt, n, lvx.data_association_stats(g)))
    g.close()
