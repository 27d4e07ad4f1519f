"""LVX_DETERMINISTIC: every addition of a pass in a fixed order => bitwise repeatable normal equations (SURVEY.md section 7 asks for a
deterministic reduction order); the default path differs from it only by summation order."""
import numpy as np
import pytest

import lvx
import synth

pytestmark = pytest.mark.gpu
LOCKS = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU


def _ctx(P, det):
    g = lvx.Context(0)
    if det:
        g.set_switch("DETERMINISTIC", 1)
    lvx.load_problem(g, P, LOCKS)
    g.set_orientation_prior(P["t0"], np.array([np.cos(5e-5), 0, 0, np.sin(5e-5)]), 28.0)
    return g


def test_small_problem_bitwise_repeatable_dense():
    P = synth.make_problem(seed=7, duration=2.0, n_surfel=900, n_planes=12, n_landmarks=30, n_camsurf=10)
    g = _ctx(P, True)
    ref = {}
    for it in range(12):
        key = it % 2
        r = g.evaluate(P["state0"] if key == 0 else P["state_true"], normal_eq=True)
        if key not in ref:
            ref[key] = r
            continue
        assert r["cost"] == ref[key]["cost"]
        assert np.array_equal(r["H"], ref[key]["H"]) and np.array_equal(r["g"], ref[key]["g"]) and np.array_equal(r["residuals"], ref[key]["residuals"])
    h = _ctx(P, False)
    d = h.evaluate(P["state0"], normal_eq=True)
    assert abs(d["cost"] - ref[0]["cost"]) <= 1e-13 * abs(d["cost"])
    assert np.abs(d["H"] - ref[0]["H"]).max() <= 1e-12 * np.abs(d["H"]).max()
    assert np.abs(d["g"] - ref[0]["g"]).max() <= 1e-12 * np.abs(d["g"]).max()
    g.close(); h.close()


def test_tenth_of_config4_checksums_repeat():
    """~1900 workgroups per family and co-visible reprojection groups: several colours per launch sequence"""
    P = synth.make_bench_problem(seed=4, n_imu=20000, n_surfel=100000, n_reproj=5000, n_planes=200)
    g = _ctx(P, True)
    g.set_state(P["state0"])
    sums = []
    for _ in range(4):
        c = g.evaluate_resident(lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ, want_cost=True)
        sums.append((c,) + g.normal_eq_checksum())
    assert all(s == sums[0] for s in sums[1:]), sums
    # the default (atomic) path is NOT expected to repeat bit for bit, but it must agree: same cost to rounding, same solve step
    h = _ctx(P, False)
    h.set_state(P["state0"])
    c2 = h.evaluate_resident(lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ, want_cost=True)
    assert abs(c2 - sums[0][0]) <= 1e-12 * abs(c2)
    d1, _ = g.solve_step(1e4)
    d2, _ = h.solve_step(1e4)
    assert np.abs(d1 - d2).max() <= 1e-7 * max(1.0, np.abs(d2).max())
    g.close(); h.close()


@pytest.mark.parametrize("size", ["small", "tenth"])
def test_lm_solve_bitwise_repeatable(size):
    """The SOLVER in deterministic mode (round 4): the landmark elimination's group products, the norms / dot products, the cyclic reduction's backward sweep and the
    border Gram add in a fixed order (tickets passed from workgroup to workgroup in blockIdx order) => two LM solves from the same start give the same bits: steps,
    cost history, final state.  The default path agrees with it to rounding."""
    if size == "small":
        P = synth.make_problem(seed=7, duration=2.0, n_surfel=900, n_planes=12, n_landmarks=30, n_camsurf=10)
    else:
        P = synth.make_bench_problem(seed=4, n_imu=20000, n_surfel=100000, n_reproj=5000, n_planes=200)
    runs = []
    for _ in range(3):
        g = _ctx(P, True)
        g.set_state(P["state0"])
        g.evaluate_resident(lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ, want_cost=True)
        d, info = g.solve_step(1e4)
        x, s = g.lm_solve(P["state0"], max_iterations=6)
        runs.append((d.copy(), x.copy(), [float(v) for v in s["cost_history"]] if "cost_history" in s else [s["initial_cost"], s["final_cost"]], s["iterations"]))
        g.close()
    for r in runs[1:]:
        assert np.array_equal(runs[0][0], r[0]), "solve_step not repeatable: max diff %.3e" % np.abs(runs[0][0] - r[0]).max()
        assert runs[0][2] == r[2] and runs[0][3] == r[3]
        assert np.array_equal(runs[0][1], r[1]), "lm_solve not repeatable: max diff %.3e" % np.abs(runs[0][1] - r[1]).max()
    h = _ctx(P, False)
    xd, sd = h.lm_solve(P["state0"], max_iterations=6)
    assert sd["iterations"] == runs[0][3]
    assert abs(sd["final_cost"] - runs[0][2][-1]) <= 1e-5 * abs(sd["final_cost"])   # (summation order only; six LM iterations amplify the last bits)
    h.close()
