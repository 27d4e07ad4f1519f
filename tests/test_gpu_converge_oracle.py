"""BASELINE.json config 4 at FULL size (1 M surfel + 200 k gyro + 200 k accel + 50 k reprojection blocks, 25 k knots, 155 k unknowns) driven TO A CERES TERMINATION through
the reference's two stages (trajInitFromSurfel <= 30 iterations, trajInitFromLVIdata <= 80: trajectory_manager_lvi.cpp:311-351, 138-195) by

  * the GPU: lvx_lm_solve — FP64-MFMA assembly, landmark elimination, block-cyclic-reduction solver (lvx_solver.hip, lvx_bcr.hip), and
  * the CPU oracle: oracle/lm_sparse.py — the oracle's dual-number Jacobian as a generic CSR matrix, generic A^T A, Schur complement of the inverse-depth e-blocks,
    dense border found by column degree, reverse Cuthill-McKee + LAPACK band Cholesky.  It shares neither the evaluator nor one line of the linear algebra with the GPU.

north_star's acceptance clause at the size it is stated on: same termination, same accept / reject sequence, cost history within 1e-7 (relative), converged extrinsics
within 1e-6 rad / 1e-4 m of the oracle's end point.  What one ceres::Solve does: kontiki/trajectory_estimator.h:38-68.  Wall time is printed (pytest -s): the oracle
takes a few seconds per iteration on 16 host cores."""
import time

import numpy as np
import pytest

import stages as cs   # noqa: E402
import synth   # noqa: E402
from oracle import pipeline   # noqa: E402

pytestmark = pytest.mark.gpu


def test_config4_full_size_converges_like_the_sparse_oracle_lm():
    P = synth.make_bench_problem(seed=4)
    N = P["n_knots"]
    t0 = time.perf_counter()
    xg, logg = cs.run_stages_gpu(P, P["state0"])
    tg = time.perf_counter() - t0
    t0 = time.perf_counter()
    xo, logo = pipeline.run_fixed_stages(P, P["state0"], sparse=True)
    to = time.perf_counter() - t0
    print("GPU %.2f s, oracle %.1f s" % (tg, to))
    for (name, sg, _), (_, so, dt) in zip(logg, logo):
        print("%-20s gpu: %s %d it, cost %.9e | oracle: %s %d it, cost %.9e (%.0f s: %s, %s)" % (name, sg["termination"], sg["iterations"], sg["final_cost"], so["termination"], so["iterations"],
                                                                                                   so["final_cost"], dt, {k: round(v, 1) for k, v in so["timing"].items()}, so["solver"]))
        assert sg["termination"] == so["termination"] and sg["termination"] in ("function_tolerance", "parameter_tolerance", "gradient_tolerance")
        assert sg["iterations"] == so["iterations"]
        assert list(sg["accepted"]) == list(so["accepted"])
        rel = np.abs(sg["cost_history"] - so["cost_history"]) / so["cost_history"]
        print("   cost history: max relative difference %.3e" % rel.max())
        assert rel.max() <= 1e-7
        assert so["solver"]["n_border"] >= 30 and so["solver"]["bandwidth"] < 400       # the oracle found the arrowhead on its own
    e = cs.extrinsic_errors(xg, xo, N)
    print("gpu vs sparse oracle LM:", e)
    assert e["lidar_rad"] <= 1e-6 and e["cam_rad"] <= 1e-6 and e["lidar_m"] <= 1e-4 and e["cam_m"] <= 1e-4
    et = cs.extrinsic_errors(xo, P["state_true"], N)
    assert et["lidar_rad"] <= 5e-4 and et["lidar_m"] <= 2e-3 and et["cam_rad"] <= 2e-3 and et["cam_m"] <= 1e-2   # and the oracle's end point is the calibration
