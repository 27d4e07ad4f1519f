"""GPU parity: HIP evaluator (through the C ABI) vs the CPU oracle on identical seeded inputs.

Tolerances (FP64): residuals 1e-11 relative to the largest residual, Jacobian rows 1e-9 relative to the
largest Jacobian entry, normal equations 1e-10 relative to max|H| (atomic summation order differs run to run).
"""
import numpy as np
import pytest

import lvx
import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TAU_LOCKS = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU


def _pair(P, locks, prior=True):
    o = O.Oracle()
    g = lvx.Context(0)
    for obj in (o, g):
        lvx.load_problem(obj, P, locks)
        if prior:
            obj.set_orientation_prior(P["t0"], np.array([np.cos(5e-5), 0, 0, np.sin(5e-5)]), 28.0)
    return o, g


def _assert_blockscaled(Hg, Ho, tol=1e-9):
    """Every entry against ITS OWN scale sqrt(H_ii H_jj) (the Cauchy-Schwarz bound of |H_ij|): small blocks — calibration x landmark,
    landmark x knot — sit many orders below the IMU diagonal and are invisible to a max|H|-relative check."""
    d = np.sqrt(np.maximum(np.diag(Ho), 0.0))
    scale = np.outer(d, d)
    bad = np.abs(Hg - Ho) > tol * scale + 1e-300
    assert not bad.any(), "worst entry-scaled error %.3e" % (np.abs(Hg - Ho)[bad] / np.maximum(scale[bad], 1e-300)).max()


def _compare(o, g, state, check_jac=True, res_floor=0.0):
    """res_floor: magnitude the residual tolerance refers to when the residuals themselves are small — a reprojection residual is the
    difference of two pixel coordinates of order 1e3, so at the ground truth (|r| ~ 1 px) 1e-11 |r| would ask for 1e-14 of the projection."""
    ro = o.evaluate(state, jac=True, normal_eq=True)
    rg = g.evaluate(state, jac=check_jac, normal_eq=True)
    nt = o.tangent_size
    assert rg["residuals"].shape == ro["residuals"].shape
    rs = max(np.abs(ro["residuals"]).max(), res_floor)
    assert np.abs(rg["residuals"] - ro["residuals"]).max() <= 1e-11 * rs
    assert abs(rg["cost"] - ro["cost"]) <= 1e-12 * abs(ro["cost"])
    if check_jac:
        Jo = O.dense_jacobian(ro["jac_cols"], ro["jac_vals"], nt)
        Jg = O.dense_jacobian(rg["jac_cols"], rg["jac_vals"], nt)
        assert np.abs(Jo - Jg).max() <= 1e-9 * np.abs(Jo).max()
    Hs = np.abs(ro["H"]).max()
    assert np.abs(rg["H"] - ro["H"]).max() <= 1e-10 * Hs
    assert np.abs(rg["g"] - ro["g"]).max() <= 1e-10 * np.abs(ro["g"]).max()
    _assert_blockscaled(rg["H"], ro["H"])
    if check_jac:   # the debug-Jacobian evaluation takes the per-segment kernels; without it the MFMA assembly path runs
        rf = g.evaluate(state, jac=False, normal_eq=True)
        assert np.abs(rf["residuals"] - ro["residuals"]).max() <= 1e-11 * rs
        assert np.abs(rf["H"] - ro["H"]).max() <= 1e-10 * Hs
        assert np.abs(rf["g"] - ro["g"]).max() <= 1e-10 * np.abs(ro["g"]).max()
    return ro, rg


@pytest.mark.parametrize("seed", [4, 5])
def test_full_lvi_parity(seed):
    P = synth.make_problem(seed=seed, duration=2.0, n_surfel=700, n_planes=12, n_landmarks=30, n_camsurf=10)
    o, g = _pair(P, TAU_LOCKS)
    _compare(o, g, P["state0"])
    _compare(o, g, P["state_true"])
    g.close()


@pytest.mark.parametrize("tracks", ["orb", "sparse"])
def test_bench_generator_shapes_small(tracks):
    """The bench generator (ORB-like co-visibility windows / one block per frame pair) at a size the oracle handles: the grouped
    cross-term kernel sees groups of many blocks (orb) and of one block (sparse)."""
    P = synth.make_bench_problem(seed=11, n_imu=1200, n_surfel=600, n_reproj=1500, n_planes=10, tracks=tracks, obs_per_frame=40)
    assert len(P["rep_lm"]) > 1000
    o, g = _pair(P, TAU_LOCKS, prior=False)
    _compare(o, g, P["state0"], check_jac=False)
    _compare(o, g, P["state_true"], res_floor=100.0)   # 1e-9 px on coordinates up to 1280 px
    g.close()


@pytest.mark.parametrize("free", [True, False])
def test_camera_time_offset_moves_views_out_of_their_cross_term_windows(free):
    """Non-zero camera time offset on the bench generator's co-visibility tracks: the cross-term groups were fixed at layout time from the view times at tau = 0, and
    a view within |tau| of a knot now evaluates in the neighbouring interval — for some blocks outside the 4-interval window of their group.  Those blocks take the
    kernel's entry-by-entry path (k_reproj_cross, "strays") and the pass stays on the fused kernels: H, g and the residuals against the oracle, no fallback;
    with the offset FREE its column (tangent 6 N + 21) rides along the fused reprojection path, with it LOCKED at that value the spans' +-1 ms margins keep it legal."""
    P = synth.make_bench_problem(seed=11, n_imu=1200, n_surfel=600, n_reproj=1500, n_planes=10, tracks="orb", obs_per_frame=40)
    N = P["n_knots"]
    locks = lvx.LOCK_LIDAR_TAU | (0 if free else lvx.LOCK_CAM_TAU)
    o, g = _pair(P, locks, prior=False)
    for tau in (8e-4, -7e-4):
        s = P["state0"].copy()
        s[7 * N + 24 + 7] = tau
        # blocks whose reference or observation view leaves its aligned 4-interval window
        rd = P["camera"]["readout"] / P["camera"]["rows"]
        lm = P["rep_lm"]
        moved = 0
        for t0v, vv in ((P["lm_t0"][lm], P["lm_uv"][lm, 1]), (P["rep_t0"], P["rep_uv"][:, 1])):
            k_a = np.floor((t0v + vv * rd - P["t0"]) / P["dt"]); k_b = np.floor((t0v + vv * rd + tau - P["t0"]) / P["dt"])
            moved += int(np.count_nonzero(k_a // 4 != k_b // 4))
        assert moved >= 3
        ro = o.evaluate(s, normal_eq=True)
        rg = g.evaluate(s, normal_eq=True)
        assert g.layout()["exact_fallback"] == 0
        assert np.abs(rg["residuals"] - ro["residuals"]).max() <= 1e-11 * np.abs(ro["residuals"]).max()
        assert np.abs(rg["H"] - ro["H"]).max() <= 1e-10 * np.abs(ro["H"]).max()
        assert np.abs(rg["g"] - ro["g"]).max() <= 1e-10 * np.abs(ro["g"]).max()
        _assert_blockscaled(rg["H"], ro["H"])
        assert (np.abs(ro["H"][6 * N + 21]).max() > 0) == free
    g.close()


def test_imu_only_config3_shape():
    P = synth.make_problem(seed=3, duration=3.0, n_surfel=0, n_planes=1, n_landmarks=0)
    o, g = _pair(P, TAU_LOCKS, prior=False)
    _compare(o, g, P["state0"])
    g.close()


@pytest.mark.parametrize("locks", [
    TAU_LOCKS | lvx.LOCK_CAM_Q | lvx.LOCK_CAM_P,                                  # Solve #1: camera locked
    TAU_LOCKS | lvx.LOCK_TRAJ | lvx.LOCK_LIDAR_Q | lvx.LOCK_LIDAR_P,              # Solve #3: trajectory + lidar locked
    TAU_LOCKS | lvx.LOCK_ACC_BIAS | lvx.LOCK_GYRO_BIAS | lvx.LOCK_LANDMARKS,
])
def test_lock_masks(locks):
    P = synth.make_problem(seed=7, duration=1.5, n_surfel=300, n_planes=8, n_landmarks=20, n_camsurf=6)
    o, g = _pair(P, locks)
    _compare(o, g, P["state0"])
    g.close()


@pytest.mark.parametrize("locks", [0, lvx.LOCK_LIDAR_TAU, lvx.LOCK_CAM_TAU])
def test_free_time_offsets(locks):
    """Free sensor time offsets (Sensor::LockTimeOffset(false), sensors.h:70-85): spans are padded by the offset bound and the residuals
    are differentiated through the spline time argument; tau columns vs the oracle's dual numbers."""
    P = synth.make_problem(seed=9, duration=1.5, n_surfel=300, n_planes=8, n_landmarks=20, n_camsurf=6)
    o, g = _pair(P, locks)
    N = P["n_knots"]
    for state in (P["state0"], P["state_true"]):
        s = state.copy()
        if not (locks & lvx.LOCK_LIDAR_TAU):
            s[7 * N + 16 + 7] = 3e-4      # lidar tau (a LOCKED non-zero offset leaves the 4-knot segment of {t, t} spans: range_error, as the reference)
        if not (locks & lvx.LOCK_CAM_TAU):
            s[7 * N + 24 + 7] = -2e-4     # cam tau
        ro, rg = _compare(o, g, s)
        assert g.layout()["exact_fallback"] == 0     # free offsets stay on the fused kernels (time-offset column in SurfAccT / CamSurfAccT / the reprojection passes)
        nt = o.tangent_size
        Jo = O.dense_jacobian(ro["jac_cols"], ro["jac_vals"], nt)
        for col, bit in ((6 * N + 14, lvx.LOCK_LIDAR_TAU), (6 * N + 21, lvx.LOCK_CAM_TAU)):
            assert (np.abs(Jo[:, col]).max() > 0) == (not (locks & bit))
    g.close()


def test_so3_only_solve0():
    P = synth.make_problem(seed=8, duration=1.5, n_surfel=0, n_planes=1, n_landmarks=0)
    o, g = _pair(P, TAU_LOCKS | lvx.LOCK_R3 | lvx.LOCK_ACC_BIAS | lvx.LOCK_GYRO_BIAS)
    o.set_so3_only(True)
    _compare(o, g, P["state0"])
    g.close()


def test_distortion_camera():
    cam = dict(synth.DEFAULT_CAMERA, k1=-0.0397646985948, k2=0.00802944041788, p1=-0.0043042199686, p2=-0.0001040279967, k3=0.00030608999077)
    P = synth.make_problem(seed=9, duration=1.5, n_surfel=100, n_planes=5, n_landmarks=25, n_camsurf=5, camera=cam)
    o, g = _pair(P, TAU_LOCKS)
    _compare(o, g, P["state0"])
    g.close()


def test_range_error_maps_to_code():
    P = synth.make_problem(seed=10, duration=1.0, n_surfel=50, n_planes=4, n_landmarks=0)
    P["t_imu"] = P["t_imu"].copy()
    P["t_imu"][-1] = P["t0"] + (P["n_knots"] - 3) * P["dt"] + 0.5     # beyond MaxTime -> std::range_error in the reference
    o, g = _pair(P, TAU_LOCKS, prior=False)
    with pytest.raises(IndexError):
        o.evaluate(P["state0"])
    with pytest.raises(lvx.LvxError) as ei:
        g.evaluate(P["state0"])
    assert ei.value.code == lvx.E_RANGE
    g.close()


def test_nonunit_quaternion_maps_to_code():
    P = synth.make_problem(seed=11, duration=1.0, n_surfel=0, n_planes=1, n_landmarks=0)
    s = P["state0"].copy()
    N = P["n_knots"]
    s[3 * N + 4 * (N // 2): 3 * N + 4 * (N // 2) + 4] *= 1.01
    o, g = _pair(P, TAU_LOCKS, prior=False)
    with pytest.raises(ValueError):
        o.evaluate(s)
    with pytest.raises(lvx.LvxError) as ei:
        g.evaluate(s)
    assert ei.value.code == lvx.E_NONUNIT_QUAT
    g.close()


def test_empty_problem():
    P = synth.make_problem(seed=12, duration=1.0, n_surfel=0, n_planes=1, n_landmarks=0)
    g = lvx.Context(0)
    g.set_spline(P["t0"], P["dt"], P["n_knots"])
    r = g.evaluate(P["state0"][: 7 * P["n_knots"] + 32], normal_eq=True)
    assert r["cost"] == 0.0 and r["residuals"].size == 0 and not r["H"].any()
    g.close()


def test_merged_hub_segment_corner_takes_the_exact_fallback():
    """t_map 5 us before a knot and a locked lidar offset of 8 us: evaluated alone, the map-time pose leaves its 4-knot segment and is found by
    the reference's t - 1e-5 retry (spline_base.h:196-203); in a residual whose point time lies in the next interval the two spans MERGE into
    a longer segment that contains t_map + tau directly, i.e. a different interpolation amount.  The shared-hub fast path detects the
    mismatch and the evaluation is redone by the per-segment kernels; either way the result must equal the oracle's."""
    P = synth.make_problem(seed=31, duration=1.0, n_surfel=400, n_planes=6, n_landmarks=0, n_camsurf=0)
    P["t_map"] = P["t0"] + 12 * P["dt"] - 5e-6
    assert P["surf_t"].min() > P["t_map"]
    # some rows in the interval right after t_map (merged segment), most further away (separate segments)
    P["surf_t"] = np.sort(np.concatenate([P["t_map"] + np.linspace(2e-3, 0.03, 40), P["surf_t"][40:]]))
    o, g = _pair(P, TAU_LOCKS, prior=False)
    N = P["n_knots"]
    s = P["state0"].copy()
    s[7 * N + 16 + 7] = 8e-6
    ro = o.evaluate(s, normal_eq=True)
    assert g.layout()["exact_fallback"] == 0
    for jac in (False, True, False):
        rg = g.evaluate(s, jac=jac, normal_eq=True)
        assert abs(rg["cost"] - ro["cost"]) <= 1e-12 * abs(ro["cost"])
        assert np.abs(rg["residuals"] - ro["residuals"]).max() <= 1e-11 * np.abs(ro["residuals"]).max()
        assert np.abs(rg["H"] - ro["H"]).max() <= 1e-10 * np.abs(ro["H"]).max()
    lo = g.layout()
    assert lo["exact_fallback"] == 0 and 0 < lo["fallback_rows"] <= 40   # the corner was detected, not missed — and only ITS rows left the fused kernels
    g.close()


def test_large_control_point_rotation_takes_the_exact_fallback():
    """The fused kernels evaluate the SO3 factors with small-angle polynomials (|Omega| <= 0.8 rad half-angle between neighbouring control
    points); a control point 2 rad away from its neighbours is marked in the pass's pair table and the evaluation is redone by the per-segment
    kernels (generic sin / cos / log) — the result must equal the oracle's either way."""
    P = synth.make_problem(seed=33, duration=1.5, n_surfel=500, n_planes=8, n_landmarks=20, n_camsurf=6)
    o, g = _pair(P, TAU_LOCKS, prior=False)
    N = P["n_knots"]
    s = P["state0"].copy()
    k = N // 2
    q = s[3 * N + 4 * k:3 * N + 4 * k + 4].copy()            # (x, y, z, w) of control point k
    s[3 * N + 4 * k:3 * N + 4 * k + 4] = synth.qmul(synth.q_from_rotvec(np.array([0.0, 0.0, 2.0])), q)
    ro = o.evaluate(s, normal_eq=True)
    assert g.layout()["exact_fallback"] == 0
    for jac in (False, True, False):
        rg = g.evaluate(s, jac=jac, normal_eq=True)
        assert abs(rg["cost"] - ro["cost"]) <= 1e-12 * abs(ro["cost"])
        assert np.abs(rg["residuals"] - ro["residuals"]).max() <= 1e-11 * np.abs(ro["residuals"]).max()
        assert np.abs(rg["H"] - ro["H"]).max() <= 1e-10 * np.abs(ro["H"]).max()
        assert np.abs(rg["g"] - ro["g"]).max() <= 1e-10 * np.abs(ro["g"]).max()
    lo = g.layout()
    # the rows whose 4-knot window holds the wide pair — IMU samples, surfel points, views — went to the exact kernel one by one; everything else stayed fused
    assert lo["exact_fallback"] == 0 and 0 < lo["fallback_rows"] < 0.2 * lo["n_blocks"]
    g.close()


def test_more_fallback_rows_than_the_lists_hold_take_the_per_segment_kernels():
    """Every control point turned by 2 rad against its neighbour: every row of every family is beyond the fused kernels' polynomials, the lists (4 096 rows per family)
    overflow and the whole pass is redone by the per-segment kernels — same numbers."""
    P = synth.make_problem(seed=34, duration=12.0, n_surfel=300, n_planes=6, n_landmarks=10, n_camsurf=0)
    assert len(P["t_imu"]) > 4096
    o, g = _pair(P, TAU_LOCKS, prior=False)
    N = P["n_knots"]
    s = P["state0"].copy()
    for k in range(1, N, 2):
        q = s[3 * N + 4 * k:3 * N + 4 * k + 4].copy()
        s[3 * N + 4 * k:3 * N + 4 * k + 4] = synth.qmul(synth.q_from_rotvec(np.array([0.0, 0.0, 2.0])), q)
    ro = o.evaluate(s, normal_eq=True)
    rg = g.evaluate(s, normal_eq=True)
    assert g.layout()["exact_fallback"] == 1
    assert abs(rg["cost"] - ro["cost"]) <= 1e-12 * abs(ro["cost"])
    assert np.abs(rg["residuals"] - ro["residuals"]).max() <= 1e-11 * np.abs(ro["residuals"]).max()
    assert np.abs(rg["H"] - ro["H"]).max() <= 1e-10 * np.abs(ro["H"]).max()
    g.close()


def test_inplace_landmark_fill_is_cleared_between_solves():
    """Free camera time offset (5-knot segments), frames one knot apart, the reference observation in the MIDDLE of a 3-view track: no single reprojection block
    reaches further than the IMU band, the landmark's row (first view .. last view) does.  The in-place landmark elimination of a single-sequence solve writes its
    Schur fill -w E^T E across that whole reach into the band itself; the next evaluation must start from a band where that fill is gone (k_clear's per-column
    full-height flag covers a free landmark's reach, not only single blocks) — H after a solve against the oracle at the same state."""
    # frames in the middle of knot intervals (pad 0.21 s = 10.5 intervals) and a 1 ms readout: a view's padded span stays inside ONE interval, a block of two views
    # one knot apart touches 5 knots (= the reach of the 5-control-point IMU / LiDAR segments), the three-view landmark 6
    cam = dict(synth.DEFAULT_CAMERA, readout=0.001)
    P = synth.make_problem(seed=41, duration=1.2, n_surfel=200, n_planes=6, n_landmarks=24, views_per_lm=3, cam_rate=50.0, n_camsurf=0, pad=0.21, camera=cam)
    lm = P["rep_lm"]
    P["lm_uv"] = P["lm_uv"].copy(); P["lm_t0"] = P["lm_t0"].copy()
    n_mid = 0
    for l in range(P["n_landmarks"]):
        rows = np.flatnonzero(lm == l)
        if len(rows) == 3:   # reference := the middle view (the pixel it was seen at there; the depth stays the first view's: consistent enough for a parity test)
            P["lm_uv"][l] = P["rep_uv"][rows[1]]; P["lm_t0"][l] = P["rep_t0"][rows[1]]; n_mid += 1
    assert n_mid >= 10
    locks = lvx.LOCK_LIDAR_TAU
    o, g = _pair(P, locks, prior=False)
    s0 = P["state0"].copy()
    ro = o.evaluate(s0, normal_eq=True)
    rg = g.evaluate(s0, normal_eq=True)
    assert np.abs(rg["H"] - ro["H"]).max() <= 1e-10 * np.abs(ro["H"]).max()
    s1, summ = g.lm_solve(s0, max_iterations=3)
    for s in (s1, s0):
        ro = o.evaluate(s, normal_eq=True)
        rg = g.evaluate(s, normal_eq=True)
        assert np.abs(rg["H"] - ro["H"]).max() <= 1e-10 * np.abs(ro["H"]).max()
        assert np.abs(rg["g"] - ro["g"]).max() <= 1e-9 * np.abs(ro["g"]).max()   # (mid-track references with the first view's depth: residuals of tens of pixels, heavy cancellation in g)
        _assert_blockscaled(rg["H"], ro["H"])
    g.close()
