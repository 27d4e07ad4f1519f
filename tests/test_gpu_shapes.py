"""Randomised problem shapes through both assembly paths (FP64-MFMA chunks / windows and per-segment kernels) against the oracle: knot spacings
from 5 ms to 100 ms, IMU rates from 50 Hz to 1 kHz (fewer or far more than one panel of samples per knot interval), dense and very sparse
LiDAR (thousands of rows per window down to empty intervals and empty chunks), few / many views per landmark, short splines (fewer
control points than one chunk), with and without camera-surfel rows — the edge cases of the window / panel / chunk logic."""
import numpy as np
import pytest

import lvx
import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TAU = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU

CASES = [
    dict(duration=0.25, dt=0.005, imu_rate=1000.0, n_surfel=3000, n_planes=5, n_landmarks=0, views_per_lm=3, n_camsurf=0),     # many rows per interval
    dict(duration=3.0, dt=0.1, imu_rate=50.0, n_surfel=40, n_planes=3, n_landmarks=6, views_per_lm=4, n_camsurf=3),          # 5 IMU samples / interval, sparse lidar
    dict(duration=0.4, dt=0.02, imu_rate=400.0, n_surfel=1, n_planes=1, n_landmarks=0, views_per_lm=2, n_camsurf=0),          # a single surfel row
    dict(duration=6.0, dt=0.02, imu_rate=100.0, n_surfel=25, n_planes=4, n_landmarks=40, views_per_lm=10, n_camsurf=20),      # empty lidar chunks, many views
    dict(duration=1.0, dt=0.013, imu_rate=333.0, n_surfel=700, n_planes=9, n_landmarks=15, views_per_lm=2, n_camsurf=0),      # stamps not aligned with the knots
    dict(duration=0.12, dt=0.02, imu_rate=400.0, n_surfel=64, n_planes=2, n_landmarks=0, views_per_lm=2, n_camsurf=0),        # spline shorter than one chunk
    dict(duration=2.0, dt=0.05, imu_rate=800.0, n_surfel=5000, n_planes=20, n_landmarks=25, views_per_lm=6, n_camsurf=25),    # 40 IMU samples and ~125 surfels / interval
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_random_shapes_both_paths(case):
    kw = CASES[case]
    P = synth.make_problem(seed=100 + case, cam_rate=20.0, **kw)
    o = O.Oracle(); g = lvx.Context(0)
    for obj in (o, g):
        lvx.load_problem(obj, P, TAU)
    scale = None
    for state in (P["state0"], P["state_true"]):
        ro = o.evaluate(state, normal_eq=True)
        if scale is None:   # magnitudes at the perturbed start; at the true state residuals and gradient are differences of much larger terms
            scale = (np.abs(ro["H"]).max(), np.abs(ro["g"]).max(), np.abs(ro["residuals"]).max(), abs(ro["cost"]))
        Hs, gs, rs, cs = scale
        fast = g.evaluate(state, jac=False, normal_eq=True)             # MFMA path
        slow = g.evaluate(state, jac=True, normal_eq=True)              # per-segment kernels (debug Jacobian requested)
        for r in (fast, slow):
            assert abs(r["cost"] - ro["cost"]) <= 1e-12 * cs
            assert np.abs(r["residuals"] - ro["residuals"]).max() <= 1e-11 * rs
            assert np.abs(r["H"] - ro["H"]).max() <= 1e-10 * Hs
            assert np.abs(r["g"] - ro["g"]).max() <= 1e-10 * gs
    g.close()
