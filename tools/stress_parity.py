"""Repeat the fused-path evaluation of a small full-LVI problem and compare every result with the first one (atomic summation order is the
only legitimate run-to-run difference): python tools/stress_parity.py [n]"""
import sys
sys.path.insert(0, "/root/repo/lvi-exc_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, synth, lvx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
worst = 0.0; bad = 0
for seed in (4, 5):
    P = synth.make_problem(seed=seed, duration=2.0, n_surfel=700, n_planes=12, n_landmarks=30, n_camsurf=10)
    g = lvx.Context(0)
    lvx.load_problem(g, P, lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU)
    g.set_orientation_prior(P["t0"], np.array([np.cos(5e-5), 0, 0, np.sin(5e-5)]), 28.0)
    ref = None
    for it in range(n):
        st = P["state0"] if it % 2 == 0 else P["state_true"]
        r = g.evaluate(st, jac=False, normal_eq=True)
        key = it % 2
        if ref is None: ref = {}
        if key not in ref: ref[key] = r; continue
        Hs = np.abs(ref[key]["H"]).max()
        e = max(np.abs(r["H"] - ref[key]["H"]).max() / Hs, np.abs(r["g"] - ref[key]["g"]).max() / np.abs(ref[key]["g"]).max(), abs(r["cost"] - ref[key]["cost"]) / abs(ref[key]["cost"]))
        worst = max(worst, e)
        if e > 1e-10: bad += 1; print("MISMATCH seed", seed, "iter", it, "rel", e)
    g.close()
print("worst relative deviation", worst, "mismatches", bad)
