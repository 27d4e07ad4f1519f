"""Known-answer tests that pin the CPU oracle (no GPU).  The reference ships no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned by identities derivable from the cited reference code, an independent numpy
restatement (synth.Spline) and derivative checks of its dual-number Jacobians."""
import numpy as np
import pytest

import lvx
import synth
from oracle import lm
from oracle import oracle as O

TAU = O.LOCK_LIDAR_TAU | O.LOCK_CAM_TAU


def _oracle(n_knots, t0=0.0, dt=0.02):
    o = O.Oracle()
    o.set_spline(t0, dt, n_knots)
    o.set_locks(TAU)
    return o


def _state(r3, so3, imu=None, lidar=None, cam=None, rho=()):
    ident = synth.sensor_block([0, 0, 0, 1], [0, 0, 0])
    return synth.pack_state(r3, so3, synth.imu_block(0.0, 0.0) if imu is None else imu, ident if lidar is None else lidar, ident if cam is None else cam, rho)


def test_constant_orientation_gives_zero_angular_velocity():
    # all SO3 control points equal => q(t) = cp, omega = 0 (M_cumul first column (1,0,0,0): spline_base.h:25-29)
    N = 12
    q = synth.q_from_rpy(0.3, -0.2, 1.1)
    o = _oracle(N)
    s = _state(np.zeros((N, 3)), np.tile(q, (N, 1)))
    e = o.eval_pose(s, np.linspace(0.0, 0.17, 9))
    assert np.abs(e["quat"] - q).max() < 1e-15
    assert np.abs(e["angvel"]).max() < 1e-13


def test_affine_control_points_give_affine_position():
    # p_i = a + b i  =>  p(t) = a + b (i0 + u + 1), v = b/dt, acc = 0 (rows of M sum to 1: spline_base.h:19-23)
    N, dt = 16, 0.02
    a, b = np.array([1.0, -2.0, 0.5]), np.array([0.1, 0.03, -0.07])
    r3 = a + b * np.arange(N)[:, None]
    o = _oracle(N, 0.0, dt)
    s = _state(r3, np.tile([0, 0, 0, 1.0], (N, 1)))
    t = np.array([0.003, 0.05, 0.1234, 0.2])
    e = o.eval_pose(s, t)
    assert np.abs(e["pos"] - (a + b * (t[:, None] / dt + 1.0))).max() < 1e-12
    assert np.abs(e["vel"] - b / dt).max() < 1e-10
    assert np.abs(e["acc"]).max() < 1e-7


def test_constant_rate_rotation_gyro_reads_the_rate():
    # control points exp(k * w dt / 2 axis): constant body rate w about a fixed axis => gyro prediction = w axis
    N, dt, w = 20, 0.02, 0.8
    axis = np.array([0.0, 0.6, 0.8])
    so3 = synth.q_from_rotvec(np.arange(N)[:, None] * (w * dt) * axis)
    o = _oracle(N, 0.0, dt)
    t = np.array([0.05, 0.11, 0.2])
    o.set_imu(t, np.tile(w * axis, (3, 1)), np.zeros((3, 3)), 28.0, 18.0)
    o.set_so3_only(True)
    o.set_locks(TAU | O.LOCK_R3 | O.LOCK_ACC_BIAS | O.LOCK_GYRO_BIAS)
    r = o.evaluate(_state(np.zeros((N, 3)), so3))
    assert np.abs(r["residuals"]).max() < 1e-11


def test_static_pose_accelerometer_reads_gravity_plus_bias():
    # roll = pitch = 0: refined_gravity = (0, 0, +9.79) (imu.h:25,61-70); prediction = g + b_a
    N = 10
    ba = np.array([0.05, 0.02, -0.03])
    o = _oracle(N)
    t = np.array([0.03, 0.09])
    o.set_imu(t, np.zeros((2, 3)), np.tile(np.array([0, 0, 9.79]) + ba, (2, 1)), 28.0, 18.0)
    s = _state(np.zeros((N, 3)), np.tile([0, 0, 0, 1.0], (N, 1)), imu=synth.imu_block(0.0, 0.0, ba, (0, 0, 0)))
    r = o.evaluate(s)
    assert np.abs(r["residuals"]).max() < 1e-12


def test_surfel_point_on_plane_has_zero_residual_and_normal_gradient():
    N = 12
    o = _oracle(N)
    Pi = np.array([[0.0, 0.0, 2.0]])                       # plane z = 2, closest point (0,0,2)
    o.set_planes(Pi)
    pts = np.array([[1.0, -0.5, 2.0], [0.3, 0.2, 2.1]])
    o.set_surfel(pts, np.array([0.10, 0.15]), np.array([0, 0]), 0.05, 5.0, 10.0)
    s = _state(np.zeros((N, 3)), np.tile([0, 0, 0, 1.0], (N, 1)))
    r = o.evaluate(s, jac=True)
    assert abs(r["residuals"][0]) < 1e-13 and abs(r["residuals"][1] - 10.0 * 0.1) < 1e-12
    J = O.dense_jacobian(r["jac_cols"], r["jac_vals"], o.tangent_size)
    # d r / d p_LinI = w (m - nL) = 0 for identity poses; d r / d(position control points of the k-eval) sums to w n
    kcols = [6 * k + 2 for k in range(N)]
    assert abs(J[0, kcols].sum()) < 1e-12                   # hub (-w n) and k (+w n) position weights cancel in z


def test_reprojection_of_the_reference_observation_is_identity():
    P = synth.make_problem(seed=31, duration=1.0, n_surfel=0, n_planes=1, n_landmarks=15, noise=False)
    o = O.Oracle(); lvx.load_problem(o, P, TAU)
    r = o.evaluate(P["state_true"])
    nI = len(P["t_imu"])
    rep = r["residuals"][6 * nI:].reshape(-1, 2)
    first = np.concatenate([[True], np.diff(P["rep_lm"]) != 0])     # each landmark's first observation is its reference
    assert np.abs(rep[first]).max() < 1e-9


def test_numpy_spline_agrees_with_oracle():
    P = synth.make_problem(seed=32, duration=1.5, n_surfel=0, n_planes=1, n_landmarks=0)
    o = O.Oracle(); lvx.load_problem(o, P, TAU)
    u = synth.unpack_state(P["state_true"], P["n_knots"], 0)
    sp = synth.Spline(P["t0"], P["dt"], u["r3"], u["so3"])
    t = np.linspace(P["t_start"], P["t_end"] - 1e-3, 101) + 1.234e-4
    a, b = o.eval_pose(P["state_true"], t), sp.eval(t)
    for k, tol in (("pos", 1e-13), ("vel", 1e-11), ("acc", 1e-9), ("quat", 1e-14), ("angvel", 1e-12)):
        assert np.abs(a[k] - b[k]).max() < tol, k


def test_dual_number_jacobians_match_finite_differences():
    P = synth.make_problem(seed=33, duration=1.0, n_surfel=150, n_planes=6, n_landmarks=12, n_camsurf=0)
    o = O.Oracle(); lvx.load_problem(o, P, TAU)
    s0 = P["state0"]
    r0 = o.evaluate(s0, jac=True)
    J = O.dense_jacobian(r0["jac_cols"], r0["jac_vals"], o.tangent_size)
    free = lm.free_tangent_indices(P["n_knots"], P["n_landmarks"], TAU)
    rng = np.random.default_rng(0)
    for _ in range(3):
        d = np.zeros(o.tangent_size); d[free] = rng.standard_normal(len(free))
        h = 1e-6
        fd = (o.evaluate(o.plus(s0, h * d))["residuals"] - o.evaluate(o.plus(s0, -h * d))["residuals"]) / (2 * h)
        assert np.abs(fd - J @ d).max() < 1e-6 * np.abs(J @ d).max()


def test_normal_equations_are_jtj_with_huber_scaling():
    P = synth.make_problem(seed=34, duration=1.0, n_surfel=200, n_planes=6, n_landmarks=10)
    o = O.Oracle(); lvx.load_problem(o, P, TAU)
    r = o.evaluate(P["state0"], jac=True, normal_eq=True)
    J = O.dense_jacobian(r["jac_cols"], r["jac_vals"], o.tangent_size)
    res = r["residuals"]
    nI, nS, nR = len(P["t_imu"]), len(P["surf_t"]), len(P["rep_lm"])
    sc = np.ones(len(res))
    o0 = 6 * nI
    s = res[o0:o0 + nS] ** 2
    sc[o0:o0 + nS] = np.where(s > 25.0, np.sqrt(5.0 / np.sqrt(np.maximum(s, 1e-300))), 1.0)
    o1 = o0 + nS
    s2 = (res[o1:o1 + 2 * nR].reshape(-1, 2) ** 2).sum(axis=1)
    sc[o1:o1 + 2 * nR] = np.repeat(np.where(s2 > 25.0, np.sqrt(5.0 / np.sqrt(np.maximum(s2, 1e-300))), 1.0), 2)
    Js = J * sc[:, None]
    assert (sc < 1).any()                                   # the perturbed start has Huber outliers
    assert np.abs(Js.T @ Js - r["H"]).max() < 1e-12 * np.abs(r["H"]).max()
    assert np.abs(Js.T @ (res * sc) - r["g"]).max() < 1e-12 * np.abs(r["g"]).max()


def test_range_and_unit_errors():
    P = synth.make_problem(seed=35, duration=1.0, n_surfel=20, n_planes=2, n_landmarks=0)
    o = O.Oracle(); lvx.load_problem(o, P, TAU)
    t = P["surf_t"].copy(); t[0] = P["t_map"] - 0.1          # spans must be ordered: t_k >= t_map (trajectory_estimator.h:102-127)
    o.set_surfel(P["surf_pt"], t, P["surf_plane"], P["t_map"], 5.0, 10.0)
    with pytest.raises(IndexError):
        o.evaluate(P["state0"])
    o.set_surfel(P["surf_pt"], P["surf_t"], P["surf_plane"], P["t_map"], 5.0, 10.0)
    s = P["state0"].copy(); N = P["n_knots"]
    s[3 * N + 4 * 20: 3 * N + 4 * 20 + 4] *= 1.001           # |q| - 1 > 1e-5 (quaternion_math.h:19-23)
    with pytest.raises(ValueError):
        o.evaluate(s)


def test_oracle_lm_recovers_extrinsics_on_noise_free_data():
    P = synth.make_problem(seed=36, duration=2.0, n_surfel=500, n_planes=12, n_landmarks=0, noise=False)
    locks = TAU | O.LOCK_CAM_Q | O.LOCK_CAM_P | O.LOCK_LANDMARKS
    o = O.Oracle(); lvx.load_problem(o, P, locks)
    free = lm.free_tangent_indices(P["n_knots"], 0, locks)
    x, s = lm.lm_solve(o, P["state0"], free, max_iterations=25, n_knots=P["n_knots"], n_landmarks=0)
    assert s["final_cost"] < 1e-6 * s["initial_cost"]
    ut, ux = synth.unpack_state(P["state_true"], P["n_knots"], 0), synth.unpack_state(x, P["n_knots"], 0)
    d = synth.qmul(ux["lidar"][:4], synth.qconj(ut["lidar"][:4]))
    assert 2 * np.arctan2(np.linalg.norm(d[:3]), abs(d[3])) < 5e-3


def test_matrix_free_products_equal_the_dense_normal_equations():
    # orc_evaluate_products (the full-size checker of tests/test_gpu_fullsize_oracle.py) against the dense assembly of orc_evaluate
    P = synth.make_problem(seed=21, duration=1.0, n_surfel=500, n_planes=8, n_landmarks=20, n_camsurf=5)
    o = O.Oracle(); lvx.load_problem(o, P, TAU)
    d = o.evaluate(P["state0"], normal_eq=True)
    V = np.random.default_rng(0).standard_normal((2, o.tangent_size))
    m = o.evaluate_products(P["state0"], V)
    assert abs(m["cost"] - d["cost"]) <= 1e-13 * d["cost"]
    assert np.array_equal(m["residuals"], d["residuals"])
    assert np.abs(m["g"] - d["g"]).max() <= 1e-13 * np.abs(d["g"]).max()
    assert np.abs(m["diag"] - np.diag(d["H"])).max() <= 1e-13 * np.diag(d["H"]).max()
    assert np.abs(m["HV"] - V @ d["H"]).max() <= 1e-13 * np.abs(V @ d["H"]).max()


# ---- first map from odometry poses (oracle/pipeline.py: first_data_association) ----
def test_key_scan_rule_known_answers():
    """LiDAROdometry::checkKeyScan (src/core/lidar_odometry.cpp:107-128): first scan; > 0.2 m from the LAST KEY scan (not the last scan); > 5 deg in yaw / pitch / roll,
    the difference wrapped once by 360 (so a 358 deg jump is a 2 deg turn); scans that were not fed do not count."""
    from oracle import pipeline

    def pose(x=0.0, yaw_deg=0.0):
        c, s = np.cos(np.deg2rad(yaw_deg)), np.sin(np.deg2rad(yaw_deg))
        T = np.eye(4); T[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]; T[0, 3] = x
        return T.ravel()
    P = [pose(5.0), pose(5.15), pose(5.19), pose(5.21), pose(5.3), pose(5.3, 4.0), pose(5.3, 5.5), pose(5.3, 179.0), pose(5.3, -179.0), pose(9.0)]
    present = [1, 1, 1, 1, 1, 1, 1, 1, 1, 0]
    # 0: first; 3: 0.21 m from scan 0; 4: only 0.09 from scan 3; 6: 5.5 deg from key 3's yaw 0; 7: 173.5 deg; 8: 179 -> -179 is 2 deg after the wrap; 9: absent
    assert pipeline.key_scans(P, present) == [0, 3, 6, 7]
    assert np.allclose(pipeline.r2ypr_deg(np.array(pose(0, 30.0)).reshape(4, 4)[:3, :3]), [30.0, 0.0, 0.0])


def test_transform_cloud_is_the_scalar_pcl_form():
    """pcl::transformPointCloud(Matrix4d) on float points: double products summed left to right, rounded to float once; non-finite points untouched."""
    from oracle import pipeline
    rng = np.random.default_rng(3)
    pts = np.zeros((50, 4), np.float32); pts[:, :3] = rng.uniform(-30, 30, (50, 3)); pts[:, 3] = 7.0
    pts[4, 0] = np.nan
    T = np.eye(4); T[:3, :3] = np.linalg.qr(rng.standard_normal((3, 3)))[0]; T[:3, 3] = [1.5, -2.25, 0.125]
    out = pipeline.transform_cloud(pts, T.ravel())
    ref = (pts[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    ok = np.arange(50) != 4
    assert np.abs(out[ok, :3] - ref[ok]).max() <= 4e-6 and np.array_equal(out[:, 3], pts[:, 3])
    assert np.isnan(out[4, 0]) and np.array_equal(out[4, 1:], pts[4, 1:])
    assert out.dtype == np.float32
