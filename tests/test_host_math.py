"""CPU check of the DEVICE math: the __host__ __device__ analytic-Jacobian functions of lvx_math.h / lvx_resid.h, built with
g++ (tests/native), against the oracle's dual-number Jacobians.  No GPU needed; this is not a product code path."""
import ctypes as C

import numpy as np
import pytest

import lvx
import synth
from oracle import oracle as O

TAU = O.LOCK_LIDAR_TAU | O.LOCK_CAM_TAU


def _hc_eval(hc, o, state):
    nr, mc = o.num_residuals, 128
    cost = C.c_double(0)
    res = np.zeros(nr); jc = np.full((nr, mc), -1, np.int32); jv = np.zeros((nr, mc))
    rc = hc.hc_evaluate(o._h, O._p(O._d(state)), C.byref(cost), O._p(res), O._p(jc), O._p(jv))
    return rc, cost.value, res, jc, jv


def _compare(hc, P, locks, prior=True, so3_only=False):
    o = O.Oracle(); lvx.load_problem(o, P, locks)
    if prior:
        o.set_orientation_prior(P["t0"], np.array([np.cos(5e-5), 0, 0, np.sin(5e-5)]), 28.0)
    o.set_so3_only(so3_only)
    for name in ("state0", "state_true"):
        r = o.evaluate(P[name], jac=True)
        rc, cost, res, jc, jv = _hc_eval(hc, o, P[name])
        assert rc == 0
        assert abs(cost - r["cost"]) <= 1e-12 * abs(r["cost"])
        assert np.abs(res - r["residuals"]).max() <= 1e-11 * np.abs(r["residuals"]).max()
        Jo = O.dense_jacobian(r["jac_cols"], r["jac_vals"], o.tangent_size)
        Jh = O.dense_jacobian(jc, jv, o.tangent_size)
        assert np.abs(Jo - Jh).max() <= 1e-12 * np.abs(Jo).max()


@pytest.mark.parametrize("seed,noise", [(4, True), (5, False)])
def test_all_families(host_check_lib, seed, noise):
    P = synth.make_problem(seed=seed, duration=2.0, n_surfel=400, n_planes=10, n_landmarks=30, n_camsurf=10, noise=noise)
    _compare(host_check_lib, P, TAU)


def test_locks_and_so3_only(host_check_lib):
    P = synth.make_problem(seed=6, duration=1.0, n_surfel=100, n_planes=5, n_landmarks=10, n_camsurf=4)
    _compare(host_check_lib, P, TAU | O.LOCK_TRAJ | O.LOCK_LIDAR_Q | O.LOCK_LIDAR_P)
    P2 = synth.make_problem(seed=7, duration=1.0, n_surfel=0, n_planes=1, n_landmarks=0)
    _compare(host_check_lib, P2, TAU | O.LOCK_R3 | O.LOCK_ACC_BIAS | O.LOCK_GYRO_BIAS, so3_only=True)


def test_distortion(host_check_lib):
    cam = dict(synth.DEFAULT_CAMERA, k1=-0.0397646985948, k2=0.00802944041788, p1=-0.0043042199686, p2=-0.0001040279967, k3=0.00030608999077)
    P = synth.make_problem(seed=8, duration=1.0, n_surfel=50, n_planes=4, n_landmarks=20, n_camsurf=5, camera=cam)
    _compare(host_check_lib, P, TAU)


def test_large_relative_rotation_between_control_points(host_check_lib):
    """logq has no hemisphere handling (quaternion_math.h:46-52): exercise relative rotations well past the series region."""
    P = synth.make_problem(seed=9, duration=1.0, n_surfel=60, n_planes=4, n_landmarks=0)
    N = P["n_knots"]
    rng = np.random.default_rng(1)
    for key in ("state0", "state_true"):
        s = P[key].copy()
        so3 = s[3 * N:7 * N].reshape(N, 4)
        so3[:] = synth.qmul(synth.q_from_rotvec(0.9 * rng.standard_normal((N, 3))), so3)
        so3 /= np.linalg.norm(so3, axis=1, keepdims=True)
        P[key] = s
    _compare(host_check_lib, P, TAU, prior=False)


@pytest.mark.parametrize("spread", [1e-9, 1e-3, 0.05, 0.9, 2.5])
def test_so3_precomputed_pairs_match_direct_evaluation(host_check_lib, spread):
    """so3_eval_pre (what the fused kernels run: log / J_r^-1 per control-point pair hoisted, double-angle forms for J_r) against so3_eval
    (the reference's arithmetic order), from the series region up to relative rotations > pi/2."""
    import ctypes as C
    rng = np.random.default_rng(11)
    out = np.zeros(4)
    worst = np.zeros(4)
    n_large = 0
    for _ in range(200):
        base = synth.q_from_rotvec(rng.standard_normal(3))
        cps = np.stack([synth.qmul(synth.q_from_rotvec(spread * rng.standard_normal(3)), base) for _ in range(4)])
        cps /= np.linalg.norm(cps, axis=1, keepdims=True)
        u = rng.uniform(0, 1)
        rc = host_check_lib.hc_so3_pre_diff(cps.ctypes.data_as(C.c_void_p), C.c_double(u), C.c_double(0.02), out.ctypes.data_as(C.c_void_p))
        # rc 2: a control-point pair beyond the small-angle polynomials (|Omega| > 0.8 rad half-angle) — such rows take the exact kernel
        assert rc == 0 or (rc == 2 and spread >= 0.5)
        n_large += rc == 2
        if rc == 0:
            worst = np.maximum(worst, out)
    # q, dxi are O(1); w_body and dw carry 1/dt = 50
    assert worst[0] <= 1e-14 and worst[2] <= 2e-12
    assert worst[1] <= 1e-11 * max(1.0, spread / 0.02) and worst[3] <= 1e-9 * max(1.0, spread / 0.02)
    assert (n_large < 200 or spread >= 2.0) and (spread < 2.0 or n_large > 0)


@pytest.mark.parametrize("spread", [1e-9, 1e-3, 0.05, 0.9, 2.5])
def test_so3_reverse_mode_pullback_matches_forward_jacobian(host_check_lib, spread):
    """so3_pullback_pre (what the single-row LiDAR residuals of the fused kernels use: dxi[k]^T g by rotations and cross products) against
    the 3x3 Jacobian blocks of so3_eval."""
    import ctypes as C
    rng = np.random.default_rng(12)
    out = np.zeros(2)
    worst = np.zeros(2)
    n_large = 0
    for _ in range(200):
        base = synth.q_from_rotvec(rng.standard_normal(3))
        cps = np.stack([synth.qmul(synth.q_from_rotvec(spread * rng.standard_normal(3)), base) for _ in range(4)])
        cps /= np.linalg.norm(cps, axis=1, keepdims=True)
        g = rng.standard_normal(3)
        rc = host_check_lib.hc_so3_pull_diff(cps.ctypes.data_as(C.c_void_p), C.c_double(rng.uniform(0, 1)), C.c_double(0.02), g.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        # rc 2: a control-point pair beyond the small-angle polynomials (|Omega| > 0.8 rad half-angle) — such rows take the exact kernel
        assert rc == 0 or (rc == 2 and spread >= 0.5)
        n_large += rc == 2
        if rc == 0:
            worst = np.maximum(worst, out)
    assert worst[0] <= 1e-14 and worst[1] <= 5e-12
    assert (n_large < 200 or spread >= 2.0) and (spread < 2.0 or n_large > 0)


@pytest.mark.parametrize("spread", [1e-9, 1e-3, 0.05, 0.5])
def test_so3_angular_velocity_pullback_matches_forward_jacobian(host_check_lib, spread):
    """so3_pullback_w_pre (the gyroscope rows of k_imu_rot: dw[k]^T g by rotations and cross products) against the 3x3 blocks dw[k] of so3_eval_pre."""
    import ctypes as C
    rng = np.random.default_rng(13)
    out = np.zeros(2)
    worst = np.zeros(2)
    n_ok = 0
    for _ in range(200):
        base = synth.q_from_rotvec(rng.standard_normal(3))
        cps = np.stack([synth.qmul(synth.q_from_rotvec(spread * rng.standard_normal(3)), base) for _ in range(4)])
        cps /= np.linalg.norm(cps, axis=1, keepdims=True)
        g = rng.standard_normal(3)
        rc = host_check_lib.hc_so3_pullw_diff(cps.ctypes.data_as(C.c_void_p), C.c_double(rng.uniform(0, 1)), C.c_double(0.02), g.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        assert rc == 0 or (rc == 2 and spread >= 0.5)
        if rc == 0:
            n_ok += 1
            worst = np.maximum(worst, out)
    assert n_ok > 20
    # angular velocities are O(spread / dt) = up to 25 rad/s here, their derivatives O(1 / dt) = 50
    assert worst[0] <= 1e-12 and worst[1] <= 2e-10


def test_two_point_lookup_matches_segment_construction(host_check_lib):
    """The locked-offset LiDAR rows skip build_segments / seg_lookup (lvx_resid.h: two_point_lookup): same status and bit-identical knot
    reference on random times, knot-aligned times, offsets that leave the segment, out-of-range and unsorted spans."""
    import ctypes as C
    rng = np.random.default_rng(21)
    t0, dt, n = 3.25, 0.02, 200
    tmax = t0 + (n - 3) * dt
    code = C.c_int(0)
    seen = {-1: 0, 0: 0, 1: 0, 2: 0}
    for it in range(20000):
        mode = it % 5
        if mode == 0:
            ta, tb = np.sort(rng.uniform(t0 - 0.1, tmax + 0.1, 2))
        elif mode == 1:   # knot-aligned times
            ta = t0 + dt * rng.integers(0, n - 3); tb = t0 + dt * rng.integers(0, n - 3)
        elif mode == 2:   # close together (merged segments)
            ta = rng.uniform(t0, tmax - 0.2); tb = ta + rng.uniform(-0.01, 0.12)
        else:
            ta = rng.uniform(t0, t0 + 0.5); tb = rng.uniform(ta, tmax)
        tau = [0.0, 1e-3, -1e-3, 0.019, -0.019, 1e-5, -1e-5, 0.07, -0.07, 5e-6][it % 10] if mode != 3 else rng.uniform(-0.03, 0.03)
        rc = host_check_lib.hc_two_point_check(C.c_double(t0), C.c_double(dt), C.c_int(n), C.c_double(ta), C.c_double(tb), C.c_double(tau), C.byref(code))
        assert rc == 0, (ta, tb, tau, code.value)
        seen[code.value] += 1
    assert seen[0] > 8000 and seen[-1] > 100 and seen[1] > 100 and seen[2] > 100, seen
