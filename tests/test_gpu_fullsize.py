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
