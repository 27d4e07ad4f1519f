"""The kept switches compute the same normal equations and the same LM step as the default path: every kernel on one stream, launch by launch instead of the
replayed graph, the per-segment (exact fallback) kernels, everything cleared instead of the structural / ownership-aware clear, the deterministic schedule, the sequential
band Cholesky instead of block cyclic reduction."""
import numpy as np
import pytest

import lvx
import synth

pytestmark = pytest.mark.gpu
LOCKS = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU


@pytest.fixture(scope="module")
def problem():
    P = synth.make_bench_problem(seed=9, n_imu=2000, n_surfel=10000, n_reproj=800, n_planes=30)
    return P, _eval(P, {})


def _eval(P, switches):
    g = lvx.Context(0)
    for k, v in switches.items():
        g.set_switch(k, v)
    lvx.load_problem(g, P, LOCKS)
    r = g.evaluate(P["state0"], normal_eq=True)
    d, mcc = g.solve_step(1e4)
    g.close()
    return r, d, mcc


@pytest.mark.parametrize("switches", [{"SERIAL": 1}, {"NO_GRAPH": 1}, {"FORCE_LEGACY": 1}, {"CLEAR_ALL": 1}, {"DETERMINISTIC": 1}, {"SOLVER_SEQ": 1}],
                         ids=lambda s: "+".join("%s=%d" % kv for kv in s.items()))
def test_variant_matches_default(problem, switches):
    P, (r0, d0, m0) = problem
    r1, d1, m1 = _eval(P, switches)
    assert abs(r1["cost"] - r0["cost"]) <= 1e-12 * abs(r0["cost"])
    assert np.abs(r1["residuals"] - r0["residuals"]).max() <= 1e-11 * np.abs(r0["residuals"]).max()
    assert np.abs(r1["H"] - r0["H"]).max() <= 1e-10 * np.abs(r0["H"]).max()
    assert np.abs(r1["g"] - r0["g"]).max() <= 1e-10 * np.abs(r0["g"]).max()
    assert np.abs(d1 - d0).max() <= 1e-7 * max(1.0, np.abs(d0).max())
    assert abs(m1 - m0) <= 1e-8 * abs(m0)
