"""Full-size (BASELINE.json config 4: 1 M surfel + 200 k IMU + 50 k reprojection blocks, 25 k knots) checks through properties that do not
need the oracle (it would take minutes at this size):

* two independent assemblies agree: the FP64-MFMA path (chunk accumulators, pseudo-pose hub, reprojection passes) against the
  per-segment kernels (LVX_FORCE_LEGACY) — same cost, and the same damped step out of the block-cyclic-reduction solver;
* additivity ("a checksum of checksums"): cost and the dense calibration block of J^T J / J^T r of the whole problem equal the sum over
  the problem split by measurement family;
* one LM step from the perturbed start reduces the cost and predicts the decrease (gain ratio near 1).
"""
import os

import numpy as np
import pytest
import torch

import lvx
import synth

pytestmark = pytest.mark.gpu
TAU = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU


@pytest.fixture(scope="module")
def bench_problem():
    return synth.make_bench_problem(seed=4)


def _border(ctx, lo):
    n = lo["border_ld"]
    buf = torch.zeros(n * n + n + 2, dtype=torch.float64, device="cuda")
    ctx.evaluate_resident(lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ)
    ctx.export_border(buf.data_ptr())
    ctx.synchronize()
    torch.cuda.synchronize()
    out = buf.cpu().numpy()
    assert out[-1] == 0.0          # device error word of the pass
    return out


def test_mfma_path_equals_per_segment_kernels_at_full_size(bench_problem):
    P = bench_problem
    g = lvx.Context(0)
    lvx.load_problem(g, P, TAU)
    g.set_state(P["state0"])
    out = {}
    for mode in ("mfma", "legacy"):
        g.set_switch("FORCE_LEGACY", 1 if mode == "legacy" else 0)
        c = g.evaluate_resident(lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ, want_cost=True)
        d, m = g.solve_step(1e4, True)
        out[mode] = (c, d, m)
    (c1, d1, m1), (c2, d2, m2) = out["mfma"], out["legacy"]
    assert abs(c1 - c2) <= 1e-12 * abs(c2)
    assert abs(m1 - m2) <= 1e-9 * abs(m2)
    assert np.abs(d1 - d2).max() <= 1e-7 * np.abs(d2).max()
    g.close()


def _calib_block(buf, lo):
    """(22 x 22 lower triangle, 22 gradient entries, cost) of the calibration scalars: they follow the hub knots in the border."""
    n, h = lo["border_ld"], 6 * lo["n_hub_knots"]
    C = buf[:n * n].reshape(n, n)[h:h + 22, h:h + 22]
    return np.tril(C), buf[n * n + h:n * n + h + 22], buf[n * n + n]


def test_cost_and_calibration_block_are_additive_over_families(bench_problem):
    P = bench_problem
    whole = lvx.Context(0)
    lvx.load_problem(whole, P, TAU)
    whole.set_state(P["state0"])
    lo = whole.layout()
    C_ref, g_ref, c_ref = _calib_block(_border(whole, lo), lo)
    whole.close()
    C_sum, g_sum, c_sum = np.zeros((22, 22)), np.zeros(22), 0.0
    for keep in ("imu", "surfel", "reproj"):
        Q = dict(P)
        if keep != "imu":
            Q["t_imu"], Q["gyro"], Q["acc"] = P["t_imu"][:0], P["gyro"][:0], P["acc"][:0]
        if keep != "surfel":
            Q["surf_t"], Q["surf_pt"], Q["surf_plane"] = P["surf_t"][:0], P["surf_pt"][:0], P["surf_plane"][:0]
        if keep != "reproj":
            Q["rep_lm"], Q["rep_uv"], Q["rep_t0"] = P["rep_lm"][:0], P["rep_uv"][:0], P["rep_t0"][:0]
        c = lvx.Context(0)
        lvx.load_problem(c, Q, TAU)
        c.set_state(P["state0"])
        lq = c.layout()
        Cp, gp, cp = _calib_block(_border(c, lq), lq)
        C_sum += Cp; g_sum += gp; c_sum += cp
        c.close()
    assert np.abs(C_ref - C_sum).max() <= 1e-10 * np.abs(C_ref).max()
    assert np.abs(g_ref - g_sum).max() <= 1e-10 * np.abs(g_ref).max()
    assert abs(c_ref - c_sum) <= 1e-12 * abs(c_ref)


def test_one_lm_step_reduces_the_cost_as_predicted(bench_problem):
    P = bench_problem
    g = lvx.Context(0)
    lvx.load_problem(g, P, TAU)
    x, s = g.lm_solve(P["state0"], max_iterations=1)
    assert list(s["accepted"]) == [1]
    assert s["final_cost"] < 0.2 * s["initial_cost"]
    c0 = g.evaluate(P["state0"], normal_eq=True, dense=False)["cost"]
    d, m = g.solve_step(1e4, True)
    c1 = g.evaluate(g.plus(P["state0"], d))["cost"]
    assert 0.5 < (c0 - c1) / m < 1.5          # gain ratio of the first (strongly nonlinear) step
    g.close()


def test_one_wide_control_point_pair_does_not_slow_the_pass(bench_problem):
    """one control-point pair beyond the fused kernels' small-angle polynomials (0.8 rad) used to send EVERY family of EVERY later pass to the per-segment
    kernels (5-9 x slower), silently.  Now only the rows whose 4-knot window holds that pair leave the fused kernels (row-level fallback lists, lvx_layout::fallback_rows):
    the pass must cost less than 10 % more than without the wide pair, and agree with the per-segment kernels' result."""
    import time
    P = bench_problem
    N = P["n_knots"]
    s = P["state0"].copy()
    k = N // 2
    q = s[3 * N + 4 * k:3 * N + 4 * k + 4].copy()
    s[3 * N + 4 * k:3 * N + 4 * k + 4] = synth.qmul(synth.q_from_rotvec(np.array([0.0, 0.0, 2.0])), q)
    what = lvx.EVAL_COST | lvx.EVAL_NORMAL_EQ

    def timed(ctx, passes=30):
        for _ in range(3):
            ctx.evaluate_resident(what)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(passes):
            ctx.evaluate_resident(what)
        ctx.synchronize()
        return (time.perf_counter() - t0) / passes
    g = lvx.Context(0)
    lvx.load_problem(g, P, TAU)
    g.set_state(P["state0"])
    g.evaluate_resident(what, want_cost=True)
    assert g.layout()["fallback_rows"] == 0
    t_plain = min(timed(g) for _ in range(3))
    g.set_state(s)
    c_fast = g.evaluate_resident(what, want_cost=True)          # discovers the wide pair, switches the lists on, repeats the pass
    d_fast, m_fast = g.solve_step(1e4, True)
    lo = g.layout()
    assert lo["exact_fallback"] == 0 and 0 < lo["fallback_rows"] < 2000      # ~4 intervals of IMU samples, surfel points and views
    t_wide = min(timed(g) for _ in range(3))
    print("pass %.4f ms, with one wide pair %.4f ms (%d rows on the fallback lists)" % (1e3 * t_plain, 1e3 * t_wide, lo["fallback_rows"]))
    assert t_wide < 1.10 * t_plain
    g.set_switch("FORCE_LEGACY", 1)
    c_ref = g.evaluate_resident(what, want_cost=True)
    d_ref, m_ref = g.solve_step(1e4, True)
    assert abs(c_fast - c_ref) <= 1e-12 * abs(c_ref) and abs(m_fast - m_ref) <= 1e-9 * abs(m_ref)
    assert np.abs(d_fast - d_ref).max() <= 1e-7 * np.abs(d_ref).max()
    g.close()
