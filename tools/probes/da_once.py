"""Two lvx_data_association rounds (the second on the one-stop chain), for instrumented builds that print from the kernels."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lvi-exc_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lvx, synth  # noqa: E402
S = synth.make_sequence(seed=50)
raw = np.zeros(S["scans"].shape, dtype=lvx.POINT_XYZIT)
for k in ("x", "y", "z", "timestamp"):
    raw[k] = S["scans"][k]
g = lvx.Context(0)
g.set_spline(S["t0"], S["dt"], S["n_knots"])
lvx.set_scans(g, raw, S["H"], S["W"])
x = np.ascontiguousarray(S["state0"], np.float64)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    print(lvx.data_association(g, x, S["t_map"]), flush=True)
g.close()
