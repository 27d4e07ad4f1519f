"""Debug: residual rows of the make_sequence(seed=50) surfel-stage problem, GPU vs oracle, per family."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lvi-exc_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch  # noqa
torch.cuda.init()
import lvx, synth
from oracle import pipeline, oracle as O

S = synth.make_sequence(seed=50)
x = S["state0"]
da = pipeline.data_association(S, x)
xs, s = pipeline.solve_stage(S, x, "TrajFromSurfel", da["planes"], da["points"], max_iterations=4)
pt, t, pid = pipeline.select_surfels(da["points"], S["t_map"], 10)
locks = pipeline.stage_locks("TrajFromSurfel")
g = lvx.Context(0); o = pipeline._base_oracle(S)
for obj in (g, o):
    if obj is g:
        obj.set_spline(S["t0"], S["dt"], S["n_knots"]); c = S["camera"]
        obj.set_camera(c["rows"], c["cols"], c["readout"], c["fx"], c["fy"], c["cx"], c["cy"], c["k1"], c["k2"], c["p1"], c["p2"], c["k3"])
        obj.set_landmarks(S["lm_uv"], S["lm_t0"])
    obj.set_imu(S["t_imu"], S["gyro"], S["acc"], 28.0, 18.0)
    obj.set_planes(da["planes"]["Pi"])
    obj.set_surfel(pt, t, pid, S["t_map"], 5.0, 10.0)
    obj.set_locks(locks)
rows = g.family_rows()
for name, st in (("state0", x), ("after 4 it", xs)):
    rg = g.evaluate(st, normal_eq=True, dense=True); ro = o.evaluate(st, normal_eq=True)
    print(name, "cost gpu %.12e oracle %.12e diff %.3e" % (rg["cost"], ro["cost"], rg["cost"] - ro["cost"]))
    for f, fam in enumerate(("gyro", "accel", "prior", "surfel", "reproj", "camsurf")):
        a, b = rows[f], rows[f + 1]
        if a == b: continue
        e = np.abs(rg["residuals"][a:b] - ro["residuals"][a:b])
        i = int(np.argmax(e))
        print("  %-7s rows %6d max err %.3e at row %d (r = %.6e / %.6e), max |r| %.3e, cost part gpu %.12e oracle %.12e" % (fam, b - a, e.max(), i, rg["residuals"][a + i], ro["residuals"][a + i],
              np.abs(ro["residuals"][a:b]).max(), 0.5 * np.sum(rg["residuals"][a:b] ** 2), 0.5 * np.sum(ro["residuals"][a:b] ** 2)))
    print("  H err %.3e g err %.3e" % (np.abs(rg["H"] - ro["H"]).max() / np.abs(ro["H"]).max(), np.abs(rg["g"] - ro["g"]).max() / np.abs(ro["g"]).max()))
