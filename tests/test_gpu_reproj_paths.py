"""GPU parity of BOTH reprojection paths against the oracle: the single-launch kernel (k_reproj_fused, round 6: a wavefront per (reference window, observation window)
group, rows never leave the chip; opt-in, measured slower) and the five-launch chain (k_reproj_jac -> side passes -> cross terms -> landmark rows).  Each is FORCED
(switch REP_FUSED = 1 / -1) on the same problems, and the launch counters prove which one ran.
Semantics: kontiki/measurements/static_rscamera_measurement.h:20-60,135-203; tolerances as tests/test_gpu_eval.py."""
import numpy as np
import pytest

import lvx
import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TAU_LOCKS = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU


def _pair(P, locks, mode):
    o = O.Oracle()
    g = lvx.Context(0)
    g.set_switch("REP_FUSED", mode)
    for obj in (o, g):
        lvx.load_problem(obj, P, locks)
    return o, g


def _blockscaled(Hg, Ho, tol=1e-9):
    d = np.sqrt(np.maximum(np.diag(Ho), 0.0))
    scale = np.outer(d, d)
    bad = np.abs(Hg - Ho) > tol * scale + 1e-300
    assert not bad.any(), "worst entry-scaled error %.3e" % (np.abs(Hg - Ho)[bad] / np.maximum(scale[bad], 1e-300)).max()


def _check(o, g, state, mode, res_floor=100.0):   # a reprojection residual is the difference of two pixel coordinates of order 1e3: 1e-9 px is their rounding floor
    ro = o.evaluate(state, normal_eq=True)
    g.set_profiling(True); g.kernel_ms()
    rg = g.evaluate(state, normal_eq=True)
    _, launches = g.kernel_ms()
    g.set_profiling(False)
    assert g.layout()["exact_fallback"] == 0
    if mode > 0:
        assert launches[lvx.KERNEL_REP_FUSED] == 1 and launches[lvx.KERNEL_REP_JAC] == 0
    else:
        assert launches[lvx.KERNEL_REP_FUSED] == 0 and launches[lvx.KERNEL_REP_JAC] == 1
    rs = max(np.abs(ro["residuals"]).max(), res_floor)
    assert np.abs(rg["residuals"] - ro["residuals"]).max() <= 1e-11 * rs
    assert abs(rg["cost"] - ro["cost"]) <= 1e-12 * abs(ro["cost"])
    assert np.abs(rg["H"] - ro["H"]).max() <= 1e-10 * np.abs(ro["H"]).max()
    assert np.abs(rg["g"] - ro["g"]).max() <= 1e-10 * np.abs(ro["g"]).max()
    _blockscaled(rg["H"], ro["H"])
    # cost-only evaluation (no normal equations): the same kernel without its assembly half
    rc = g.evaluate(state, normal_eq=False)
    assert abs(rc["cost"] - ro["cost"]) <= 1e-12 * abs(ro["cost"])
    assert np.abs(rc["residuals"] - ro["residuals"]).max() <= 1e-11 * rs


@pytest.mark.parametrize("mode", [1, -1])
@pytest.mark.parametrize("seed", [4, 5])
def test_full_lvi_both_paths(seed, mode):
    P = synth.make_problem(seed=seed, duration=2.0, n_surfel=700, n_planes=12, n_landmarks=30, n_camsurf=10)
    o, g = _pair(P, TAU_LOCKS, mode)
    _check(o, g, P["state0"], mode)
    _check(o, g, P["state_true"], mode)
    g.close()


@pytest.mark.parametrize("mode", [1, -1])
@pytest.mark.parametrize("tracks", ["orb", "sparse"])
def test_bench_tracks_both_paths(tracks, mode):
    """ORB-like co-visibility (frame pairs with tens of blocks: long runs) and one block per frame pair (every run is one block) through either path."""
    P = synth.make_bench_problem(seed=11, n_imu=1200, n_surfel=600, n_reproj=1500, n_planes=10, tracks=tracks, obs_per_frame=40)
    o, g = _pair(P, TAU_LOCKS, mode)
    _check(o, g, P["state0"], mode)
    _check(o, g, P["state_true"], mode, res_floor=100.0)
    g.close()


def test_default_is_the_chain():
    """The single-launch kernel is opt-in (measured slower at config 4: DESIGN.md 3.1): without the switch the five-launch chain runs."""
    P = synth.make_bench_problem(seed=11, n_imu=1200, n_surfel=600, n_reproj=1500, n_planes=10, tracks="orb", obs_per_frame=40)
    g = lvx.Context(0)
    lvx.load_problem(g, P, TAU_LOCKS)
    g.set_profiling(True); g.kernel_ms()
    g.evaluate(P["state0"], normal_eq=True)
    _, launches = g.kernel_ms()
    assert launches[lvx.KERNEL_REP_FUSED] == 0 and launches[lvx.KERNEL_REP_JAC] == 1
    g.close()


@pytest.mark.parametrize("mode", [1, -1])
@pytest.mark.parametrize("free", [True, False])
def test_camera_time_offset_both_paths(free, mode):
    """A non-zero camera time offset (free: its column 6 N + 21 rides along; locked at a non-zero value) moves views into neighbouring knot intervals: the fused kernel
    forms its runs from the intervals the evaluation returns, the chain handles them as strays."""
    P = synth.make_bench_problem(seed=11, n_imu=1200, n_surfel=600, n_reproj=1500, n_planes=10, tracks="orb", obs_per_frame=40)
    N = P["n_knots"]
    locks = lvx.LOCK_LIDAR_TAU | (0 if free else lvx.LOCK_CAM_TAU)
    o, g = _pair(P, locks, mode)
    for tau in (8e-4, -7e-4):
        s = P["state0"].copy()
        s[7 * N + 24 + 7] = tau
        _check(o, g, s, mode)
    g.close()


@pytest.mark.parametrize("mode", [1, -1])
def test_locked_landmarks_and_locked_trajectory_both_paths(mode):
    """Lock masks of the calibration stages (trajectory_manager_lvi.cpp:138-257): landmarks constant (no landmark rows at all), and the third stage's
    trajectory + LiDAR lock (only camera extrinsics, landmarks, IMU calibration free: every knot column is dead)."""
    P = synth.make_problem(seed=6, duration=2.0, n_surfel=500, n_planes=10, n_landmarks=40, n_camsurf=10)
    for locks in (TAU_LOCKS | lvx.LOCK_LANDMARKS, TAU_LOCKS | lvx.LOCK_TRAJ | lvx.LOCK_LIDAR_Q | lvx.LOCK_LIDAR_P):
        o, g = _pair(P, locks, mode)
        _check(o, g, P["state0"], mode)
        g.close()


@pytest.mark.parametrize("mode", [1, -1])
def test_radtan_camera_both_paths(mode):
    cam = dict(synth.DEFAULT_CAMERA, k1=-0.0397646985948, k2=0.00802944041788, p1=-0.0043042199686, p2=-0.0001040279967, k3=0.00030608999077)   # lvi.yaml:54-78
    P = synth.make_problem(seed=8, duration=2.0, n_surfel=300, n_planes=8, n_landmarks=40, n_camsurf=0, camera=cam)
    o, g = _pair(P, TAU_LOCKS, mode)
    _check(o, g, P["state0"], mode, res_floor=100.0)
    g.close()


def test_fused_kernel_solves_like_the_chain():
    """The LM loop over either path from the same start: same iteration count and termination, cost history to 1e-9 relative, final states to 1e-9."""
    P = synth.make_bench_problem(seed=12, n_imu=1600, n_surfel=800, n_reproj=2000, n_planes=10, tracks="orb", obs_per_frame=40)
    out = []
    for mode in (1, -1):
        g = lvx.Context(0)
        g.set_switch("REP_FUSED", mode)
        lvx.load_problem(g, P, TAU_LOCKS)
        x, res = g.lm_solve(P["state0"], max_iterations=8)
        res["state"] = x
        out.append(res)
        g.close()
    a, b = out
    assert a["iterations"] == b["iterations"] and a["termination"] == b["termination"]
    ca, cb = np.asarray(a["cost_history"]), np.asarray(b["cost_history"])
    assert np.abs(ca - cb).max() <= 1e-9 * np.abs(cb).max()
    assert np.abs(a["state"] - b["state"]).max() <= 1e-9 * max(1.0, np.abs(b["state"]).max())


@pytest.mark.parametrize("mode", [1, -1])
def test_wide_control_point_pair_under_a_camera_frame_both_paths(mode):
    """A control point turned by 2 rad against its neighbours in the middle of a co-visibility window: the reprojection blocks whose views touch it cannot use the
    control-point-pair table (small-angle polynomials) — k_reproj_jac / k_reproj_fused put them on the reprojection fallback list, the exact per-segment kernel adds them
    behind the landmark rows; IMU samples and surfel points around the same knot take their own lists.  Same numbers as the oracle, the pass stays on the fused kernels."""
    P = synth.make_bench_problem(seed=11, n_imu=1200, n_surfel=600, n_reproj=1500, n_planes=10, tracks="orb", obs_per_frame=40)
    N = P["n_knots"]
    k = int((P["lm_t0"][0] - P["t0"]) / P["dt"]) + 2
    s = P["state0"].copy()
    s[3 * N + 4 * k:3 * N + 4 * k + 4] = synth.qmul(synth.q_from_rotvec(np.array([0.0, 0.0, 2.0])), s[3 * N + 4 * k:3 * N + 4 * k + 4].copy())
    o, g = _pair(P, TAU_LOCKS, mode)
    g.evaluate(s, normal_eq=True)      # the pass that discovers the rows switches the lists on and is repeated: _check counts the launches of a settled pass
    _check(o, g, s, mode)
    lo = g.layout()
    assert lo["exact_fallback"] == 0 and lo["fallback_rows"] > 50      # IMU samples + surfel points + the views of the frames around knot k
    # the reprojection blocks were among them: with the reprojection family alone the list is not empty either
    Q = dict(P, t_imu=P["t_imu"][:0], gyro=P["gyro"][:0], acc=P["acc"][:0], surf_pt=P["surf_pt"][:0], surf_t=P["surf_t"][:0], surf_plane=P["surf_plane"][:0])
    o2, g2 = _pair(Q, TAU_LOCKS, mode)
    g2.evaluate(s, normal_eq=True)
    _check(o2, g2, s, mode)
    assert g2.layout()["fallback_rows"] > 0
    g.close(); g2.close()
