"""Golden regression vectors (tests/golden, made by tests/golden/make_golden.py from this repo's oracle): the oracle must reproduce
them on CPU; on the GPU box the HIP path is checked against the same files (no /root/reference needed at run time)."""
import os

import numpy as np
import pytest

import lvx
from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAU = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU


def _problem(z):
    P = {k: z[k] for k in z.files}
    for k in ("t0", "dt", "w_gyro", "w_acc", "t_map", "huber_surf", "w_surf", "huber_rep", "w_rep", "huber_cs", "w_cs"):
        P[k] = float(P[k])
    for k in ("n_knots", "n_landmarks"):
        P[k] = int(P[k])
    c = z["camera"]
    P["camera"] = dict(rows=int(c[0]), cols=int(c[1]), readout=c[2], fx=c[3], fy=c[4], cx=c[5], cy=c[6], k1=c[7], k2=c[8], p1=c[9], p2=c[10], k3=c[11])
    return P


def _check_solve(obj, z):
    P = _problem(z)
    lvx.load_problem(obj, P, TAU)
    r = obj.evaluate(P["state0"], normal_eq=True)
    assert abs(r["cost"] - float(z["cost"])) <= 1e-12 * float(z["cost"])
    assert np.abs(r["residuals"] - z["residuals"]).max() <= 1e-11 * np.abs(z["residuals"]).max()
    assert np.abs(r["g"] - z["g"]).max() <= 1e-10 * np.abs(z["g"]).max()
    assert np.abs(np.diag(r["H"]) - z["H_diag"]).max() <= 1e-10 * np.abs(z["H_diag"]).max()
    assert abs(np.linalg.norm(r["H"]) - float(z["H_frob"])) <= 1e-10 * float(z["H_frob"])


def test_oracle_reproduces_solve_golden():
    _check_solve(O.Oracle(), np.load(os.path.join(G, "solve_small.npz")))


def test_oracle_reproduces_upstream_golden():
    z = np.load(os.path.join(G, "scanreg_small.npz"))
    r = O.scan_register(z["pts"], 16, 0.3)
    for k in ("label", "sort_ind", "picked", "sharp", "less_sharp", "flat", "less_flat", "scan_start", "scan_end"):
        assert np.array_equal(r[k], z[k]), k
    assert np.array_equal(r["curvature"].view(np.uint32), z["curvature"].view(np.uint32))
    z = np.load(os.path.join(G, "voxel_small.npz"))
    v = O.voxel_build(z["cloud"], 1.0)
    assert np.array_equal(v["leaf_key"], z["leaf_key"]) and np.array_equal(v["leaf_n"], z["leaf_n"]) and np.array_equal(v["grid"], z["grid"])
    assert np.allclose(v["mean"], z["mean"], rtol=1e-13, atol=1e-13) and np.allclose(v["evals"], z["evals"], rtol=1e-12, atol=1e-18)
    assert np.array_equal(O.voxel_lookup7(v, z["queries"], 1.0), z["ids7"])
    z = np.load(os.path.join(G, "assoc_small.npz"))
    assert np.array_equal(O.surfel_assoc(z["scan"], z["p4"], z["bmin"], z["bmax"], 0.05, 2), z["flag"])


def _check_next_rows(obj, mod, ctx_for_upstream):
    z = np.load(os.path.join(G, "tau_deskew_small.npz"))
    P = _problem(z)
    lvx.load_problem(obj, P, 0)
    r = obj.evaluate(z["state"], jac=True, normal_eq=True)
    N = P["n_knots"]
    J = O.dense_jacobian(r["jac_cols"], r["jac_vals"], obj.tangent_size)
    assert abs(r["cost"] - float(z["cost"])) <= 1e-12 * float(z["cost"])
    assert np.abs(r["residuals"] - z["residuals"]).max() <= 1e-11 * np.abs(z["residuals"]).max()
    for col, key in ((6 * N + 14, "J_tau_lidar"), (6 * N + 21, "J_tau_cam")):
        assert np.abs(z[key]).max() > 0 and np.abs(J[:, col] - z[key]).max() <= 1e-9 * np.abs(z[key]).max()
    assert np.abs(r["g"] - z["g"]).max() <= 1e-10 * np.abs(z["g"]).max()
    raw = z["raw"].view(lvx.POINT_XYZIT)
    und = mod.undistort(obj, z["state_true"], raw, synth_qconj(z["q_map"]), z["p_map"], True)
    assert np.array_equal(np.isnan(und), np.isnan(z["undistorted"]))
    m = ~np.isnan(und)
    assert np.abs(und[m] - z["undistorted"][m]).max() <= 4e-6
    z = np.load(os.path.join(G, "surfel_extract_small.npz"))
    if ctx_for_upstream is None:
        e = O.surfel_extract(z["cloud"], O.voxel_build(z["cloud"], 0.5))
    else:
        lvx.voxel_build(ctx_for_upstream, z["cloud"], 0.5, fetch=False)
        e, n = lvx.surfel_extract(ctx_for_upstream, max_planes=len(z["leaf"]) + 8)
        assert n == len(z["leaf"])
    assert np.array_equal(e["leaf"], z["leaf"]) and np.array_equal(e["n_inliers"], z["n_inliers"]) and np.array_equal(e["plane_type"], z["plane_type"])
    assert np.abs(e["p4"] - z["p4"]).max() <= 1e-9 and np.array_equal(e["box_min"], z["box_min"]) and np.array_equal(e["box_max"], z["box_max"])


def synth_qconj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def test_oracle_reproduces_next_rows_golden():
    _check_next_rows(O.Oracle(), O, None)


@pytest.mark.gpu
def test_gpu_reproduces_next_rows_golden():
    ctx = lvx.Context(0)
    _check_next_rows(ctx, lvx, ctx)
    ctx.close()


@pytest.mark.gpu
def test_gpu_reproduces_golden():
    ctx = lvx.Context(0)
    _check_solve(ctx, np.load(os.path.join(G, "solve_small.npz")))
    z = np.load(os.path.join(G, "scanreg_small.npz"))
    r = lvx.scan_register(ctx, z["pts"], 16, 0.3)
    for k in ("label", "sort_ind", "picked", "sharp", "less_sharp", "flat", "less_flat", "scan_start", "scan_end"):
        assert np.array_equal(r[k], z[k]), k
    assert np.array_equal(r["curvature"].view(np.uint32), z["curvature"].view(np.uint32))
    z = np.load(os.path.join(G, "voxel_small.npz"))
    v = lvx.voxel_build(ctx, z["cloud"], 1.0)
    assert np.array_equal(v["leaf_key"], z["leaf_key"]) and np.array_equal(v["leaf_n"], z["leaf_n"]) and np.array_equal(v["grid"], z["grid"])
    assert np.allclose(v["mean"], z["mean"], rtol=1e-12, atol=1e-12) and np.allclose(v["evals"], z["evals"], rtol=1e-10, atol=1e-16)
    assert np.array_equal(lvx.voxel_lookup7(ctx, z["queries"]), z["ids7"])
    z = np.load(os.path.join(G, "assoc_small.npz"))
    assert np.array_equal(lvx.surfel_assoc(ctx, z["scan"], z["p4"], z["bmin"], z["bmax"], 0.05, 2), z["flag"])
    ctx.close()
