"""The leaves + separators elimination of the band (csrc/lvx_nd.h: narrow band leaves, dense leaves for the co-visibility windows, a chain of 32-wide separators) against
the uniform block chain (SOLVER_ND = -1: block cyclic reduction with b = bandwidth) and the sequential band Cholesky (SOLVER_SEQ) on the SAME normal equations: three
elimination orders of one SPD system.  Reference semantics of the step: Ceres' SPARSE_SCHUR LM step (kontiki/trajectory_estimator.h:38-68)."""
import numpy as np
import pytest

import lvx
import synth

pytestmark = pytest.mark.gpu
TAU = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU
RADIUS = 1e4


def _step(P, locks, x, nd, seq=False, leaf=None, radius=RADIUS):
    g = lvx.Context(0)
    g.set_switch("SOLVER_ND", nd)
    if seq:
        g.set_switch("SOLVER_SEQ", 1)
    lvx.load_problem(g, P, locks)
    g.evaluate(x, normal_eq=True, dense=False, residuals=False)
    d, m = g.solve_step(radius, True)
    lo = g.layout()
    g.close()
    return d, m, lo


def _same(d, m, dr, mr, tol=1e-7):
    assert np.all(np.isfinite(d)) and np.abs(d).max() > 0
    assert np.abs(d - dr).max() <= tol * max(1.0, np.abs(dr).max()), "step differs by %.3e" % (np.abs(d - dr).max() / max(1.0, np.abs(dr).max()))
    assert abs(m - mr) <= 1e-8 * abs(mr)


@pytest.mark.parametrize("seed,n_reproj,obs", [(31, 2400, 40), (32, 1200, 60), (33, 4000, 40)])
def test_step_matches_the_uniform_chain_and_the_sequential_solver(seed, n_reproj, obs):
    """Narrow stretches + co-visibility windows (dense leaves) + hub border (53 right-hand sides)."""
    P = synth.make_bench_problem(seed=seed, n_imu=16000, n_surfel=40000, n_reproj=n_reproj, n_planes=60, obs_per_frame=obs)
    x = P["state0"]
    d, m, lo = _step(P, TAU, x, 1)
    assert lo["solver_separators"] > 2 and lo["solver_leaves"] == lo["solver_separators"] + 1 and lo["solver_fallbacks"] == 0
    du, mu, lou = _step(P, TAU, x, -1)
    assert lou["solver_separators"] == 0 and lou["solver_fallbacks"] == 0
    ds, ms, _ = _step(P, TAU, x, -1, seq=True)
    _same(d, m, ds, ms)
    _same(du, mu, ds, ms)
    _same(d, m, du, mu)


def test_imu_only_band_has_narrow_leaves_only():
    """Config 3's shape: no camera, no LiDAR — no wide run, no hub knots (23 right-hand sides: the two-tile instance)."""
    P = synth.make_problem(seed=7, duration=40.0, n_surfel=0, n_landmarks=0, n_camsurf=0)
    locks = TAU | lvx.LOCK_LIDAR_Q | lvx.LOCK_LIDAR_P | lvx.LOCK_CAM_Q | lvx.LOCK_CAM_P
    x = P["state0"]
    d, m, lo = _step(P, locks, x, 1)
    assert lo["solver_separators"] >= 2 and lo["n_border"] == 22
    ds, ms, _ = _step(P, locks, x, -1, seq=True)
    _same(d, m, ds, ms)


def test_free_time_offsets_widen_the_separators_to_30_columns():
    P = synth.make_bench_problem(seed=34, n_imu=16000, n_surfel=40000, n_reproj=2400, n_planes=60, obs_per_frame=40)
    N = P["n_knots"]
    x = P["state0"].copy()
    x[7 * N + 23], x[7 * N + 31] = 4e-4, -5e-4
    d, m, lo = _step(P, 0, x, 1)
    assert lo["solver_separators"] > 2
    ds, ms, _ = _step(P, 0, x, -1, seq=True)
    _same(d, m, ds, ms)


def test_small_radius_and_locked_landmarks():
    P = synth.make_bench_problem(seed=35, n_imu=16000, n_surfel=40000, n_reproj=2400, n_planes=60, obs_per_frame=40)
    x = P["state0"]
    for locks, radius in ((TAU | lvx.LOCK_LANDMARKS, 1e4), (TAU, 3.0)):
        d, m, lo = _step(P, locks, x, 1, radius=radius)
        assert lo["solver_separators"] > 2
        ds, ms, _ = _step(P, locks, x, -1, seq=True, radius=radius)
        _same(d, m, ds, ms)


def test_lm_loop_takes_the_same_path_with_either_elimination():
    P = synth.make_bench_problem(seed=36, n_imu=16000, n_surfel=40000, n_reproj=2400, n_planes=60, obs_per_frame=40)
    out = []
    for nd in (1, -1):
        g = lvx.Context(0)
        g.set_switch("SOLVER_ND", nd)
        lvx.load_problem(g, P, TAU)
        x, res = g.lm_solve(P["state0"], max_iterations=8)
        res["state"] = x
        res["layout"] = g.layout()
        out.append(res)
        g.close()
    a, b = out
    assert a["layout"]["solver_separators"] > 2 and b["layout"]["solver_separators"] == 0
    assert a["iterations"] == b["iterations"] and a["termination"] == b["termination"]
    ca, cb = np.asarray(a["cost_history"]), np.asarray(b["cost_history"])
    assert np.abs(ca - cb).max() <= 1e-9 * np.abs(cb).max()
    assert np.abs(a["state"] - b["state"]).max() <= 1e-8 * max(1.0, np.abs(b["state"]).max())


def test_profile_that_does_not_fit_keeps_the_uniform_chain():
    """A continuous camera stream ('sparse' tracks: every column is wide): no plan, the chain of b = bandwidth blocks runs and nothing changes."""
    P = synth.make_bench_problem(seed=37, n_imu=6000, n_surfel=20000, n_reproj=3000, n_planes=40, tracks="sparse")
    d, m, lo = _step(P, TAU, P["state0"], 1)
    assert lo["solver_separators"] == 0 and lo["solver_fallbacks"] == 0
    ds, ms, _ = _step(P, TAU, P["state0"], -1, seq=True)
    _same(d, m, ds, ms)


def test_deterministic_mode_is_bitwise_reproducible_on_this_path():
    """No atomics anywhere in the leaves + separators elimination: with the deterministic evaluation (one adder per accumulator entry) two steps are bit-identical."""
    P = synth.make_bench_problem(seed=38, n_imu=16000, n_surfel=40000, n_reproj=2400, n_planes=60, obs_per_frame=40)
    out = []
    for _ in range(2):
        g = lvx.Context(0)
        g.set_switch("DETERMINISTIC", 1)
        g.set_switch("SOLVER_ND", 1)
        lvx.load_problem(g, P, TAU)
        g.evaluate(P["state0"], normal_eq=True, dense=False, residuals=False)
        d, m = g.solve_step(RADIUS, True)
        assert g.layout()["solver_separators"] > 2
        out.append((d, m))
        g.close()
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]


def test_joint_step_of_two_sequences_is_the_same_with_either_elimination():
    """lvx_solve_step_shared (sequence per GPU, shared rig extrinsics): the band of every rank goes through its own elimination plan; the 14 shared columns sit in the
    border and ride through it as right-hand sides.  Two ranks in two threads, in-process all-reduce (tests/test_gpu_shared.py)."""
    import threading
    import sharded
    seqs = []
    ref = None
    for r in range(2):
        P = synth.make_bench_problem(seed=50 + r, n_imu=12000 + 4000 * r, n_surfel=30000, n_reproj=2000, n_planes=50, obs_per_frame=40)
        N = P["n_knots"]
        s = P["state0"].copy()
        if ref is None:
            ref = s[7 * N + 16:7 * N + 32].copy()
        s[7 * N + 16:7 * N + 32] = ref
        seqs.append((P, s))

    def run(nd):
        ar = sharded.ThreadAllReduce(2)
        res, err = [None, None], [None, None]

        def work(r):
            try:
                g = lvx.Context(0)
                g.set_switch("SOLVER_ND", nd)
                lvx.load_problem(g, seqs[r][0], TAU)
                g.evaluate(seqs[r][1], normal_eq=True, dense=False, residuals=False)
                d, m = g.solve_step_shared(RADIUS, ar.rank_fn(r))
                res[r] = (d, m, g.layout()["solver_separators"])
                g.close()
            except BaseException as e:   # noqa: BLE001
                err[r] = e
                ar.bar.abort()
        th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=600)
        for e in err:
            if e is not None:
                raise e
        return res
    a, b = run(1), run(-1)
    for r in range(2):
        assert a[r][2] > 2 and b[r][2] == 0
        _same(a[r][0], a[r][1], b[r][0], b[r][1])
    sh0, sh1 = sharded.shared_tangent_indices(seqs[0][0]["n_knots"]), sharded.shared_tangent_indices(seqs[1][0]["n_knots"])
    assert np.array_equal(a[0][0][sh0], a[1][0][sh1])      # the shared step is bitwise identical on both ranks


@pytest.mark.parametrize("seed", range(12))
def test_random_window_layouts_agree_with_the_sequential_solver(seed):
    """Co-visibility windows of different widths and spacings (views per landmark 3 .. 8 at 10 .. 20 Hz: dense leaves of ~70 .. 190 columns; wider or merged windows: no plan, the uniform chain; windows close to the ends of the
    sequence, narrow stretches of very different lengths), different trust radii; whatever plan comes out — or none — the step is the sequential solver's."""
    rng = np.random.default_rng(900 + seed)
    views = int(rng.integers(3, 9))
    P = synth.make_bench_problem(seed=60 + seed, n_imu=int(rng.integers(9000, 20000)), n_surfel=30000, n_reproj=int(rng.integers(300, 1500)), n_planes=50,
                                 views_per_lm=views, cam_rate=float(rng.choice([10.0, 15.0, 20.0])), obs_per_frame=int(rng.integers(20, 60)), pad=float(rng.choice([0.1, 0.2, 0.4])))
    radius = float(rng.choice([3.0, 1e2, 1e4, 1e6]))
    x = P["state0"]
    d, m, lo = _step(P, TAU, x, 1, radius=radius)
    ds, ms, _ = _step(P, TAU, x, -1, seq=True, radius=radius)
    print("seed %d: views %d, band %d x %d, %d separators / %d leaves, radius %g" % (seed, views, lo["n_band"], lo["bandwidth"], lo["solver_separators"], lo["solver_leaves"], radius))
    assert lo["solver_fallbacks"] == 0
    _same(d, m, ds, ms)


def test_failed_pivot_sends_the_step_to_the_sequential_solver(monkeypatch):
    """The pivot codes of the elimination reach the host with the step's sums; a failure voids the step, which the sequential band Cholesky redoes (counted in
    lvx_layout::solver_fallbacks).  LVX_TEST_BAD_PIVOT declares a failure where there is none: the redone step is the step."""
    P = synth.make_bench_problem(seed=39, n_imu=16000, n_surfel=40000, n_reproj=2400, n_planes=60, obs_per_frame=40)
    x = P["state0"]
    d0, m0, lo0 = _step(P, TAU, x, 1)
    assert lo0["solver_fallbacks"] == 0 and lo0["solver_separators"] > 2
    monkeypatch.setenv("LVX_TEST_BAD_PIVOT", "1")
    d1, m1, lo1 = _step(P, TAU, x, 1)
    monkeypatch.delenv("LVX_TEST_BAD_PIVOT")
    assert lo1["solver_fallbacks"] == 1
    _same(d1, m1, d0, m0)
    # and the LM loop keeps going through such steps
    monkeypatch.setenv("LVX_TEST_BAD_PIVOT", "1")
    g = lvx.Context(0)
    g.set_switch("SOLVER_ND", 1)
    lvx.load_problem(g, P, TAU)
    xs, res = g.lm_solve(x, max_iterations=3)
    n_fb = g.layout()["solver_fallbacks"]
    g.close()
    monkeypatch.delenv("LVX_TEST_BAD_PIVOT")
    g = lvx.Context(0)
    g.set_switch("SOLVER_ND", 1)
    lvx.load_problem(g, P, TAU)
    xr, ref = g.lm_solve(x, max_iterations=3)
    g.close()
    assert n_fb == res["iterations"] and res["iterations"] == ref["iterations"]
    assert np.abs(np.asarray(res["cost_history"]) - np.asarray(ref["cost_history"])).max() <= 1e-9 * np.abs(np.asarray(ref["cost_history"])).max()
