"""The oracle (CPU test) and the HIP evaluator (GPU test) against the INDEPENDENT ambient-coordinate restatement of the reference's residual
functors (tests/golden/make_ambient.py: written from the reference headers in 50-digit arithmetic, derivatives by central differences, Ceres'
parameter blocks 4-wide for quaternions).  What is compared:
  * residuals of every block (gyro, accel, orientation prior, surfel, rolling-shutter reprojection, camera-landmark-to-surfel);
  * Jacobians: J_ambient . P, with P the Jacobian of ceres::EigenQuaternionParameterization::Plus at delta = 0 (columns e_j (x) q — restated
    here from Ceres' documentation) against the tangent rows the oracle (dual numbers) and lvx_get_jacobian (analytic, on the group) produce;
  * cost, J^T J and J^T r with ceres::HuberLoss restated here (rho(s) = s for s <= a^2, 2 a sqrt(s) - a^2 above; rows scaled by sqrt(rho')).
This does not lift "parity unpinned" (the reference itself cannot be built here) but removes the single-restatement failure mode."""
import os

import numpy as np
import pytest

import lvx
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
TAU = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU


def _qmul(a, b):
    return synth.qmul(np.asarray(a, float), np.asarray(b, float))


VARIANTS = ["small", "tau", "radtan", "solve0"]   # tests/golden/make_ambient.py: every family | free time offsets | radial-tangential camera | Solve #0 (SO3-only estimator)


def _locks(P):
    if P.get("so3_only"):
        return lvx.LOCK_R3 | lvx.LOCK_ACC_BIAS | lvx.LOCK_GYRO_BIAS | TAU      # initialSO3TrajWithGyro (trajectory_manager_lvi.cpp:43-62)
    return 0 if P.get("free_tau") else TAU


def _load(variant="small"):
    z = np.load(os.path.join(HERE, "golden", "ambient_%s.npz" % variant))
    P = {k: z[k] for k in z.files if not k.startswith("camera_")}
    P["camera"] = {k[7:]: (int(z[k]) if k[7:] in ("rows", "cols") else float(z[k])) for k in z.files if k.startswith("camera_")}
    for k in ("t0", "dt", "w_gyro", "w_acc", "t_map", "huber_surf", "w_surf", "huber_rep", "w_rep", "huber_cs", "w_cs", "prior_t", "prior_w"):
        P[k] = float(P[k])
    P["n_knots"], P["n_landmarks"] = int(P["n_knots"]), int(P["n_landmarks"])
    P["free_tau"], P["so3_only"] = bool(int(P.get("free_tau", 0))), bool(int(P.get("so3_only", 0)))
    return P


def _tangent_map(state, N, L):
    """T [state x tangent]: ambient Jacobian rows times T = rows in the manifold's tangent (what Ceres forms as J * P)."""
    ns, nt = 7 * N + 32 + L, 6 * N + 22 + L
    T = np.zeros((ns, nt))

    def quat(so, to):   # d (delta (+) q) / d delta at 0, delta = half-angle vector, left multiplication; storage (x, y, z, w)
        q = state[so:so + 4]
        for j in range(3):
            e = np.zeros(4); e[j] = 1.0
            T[so:so + 4, to + j] = _qmul(e, q)

    for k in range(N):
        T[3 * k:3 * k + 3, 6 * k:6 * k + 3] = np.eye(3)
        quat(3 * N + 4 * k, 6 * k + 3)
    b, c = 7 * N, 6 * N
    for so, to, n in ((8, 0, 1), (9, 1, 1), (10, 2, 3), (13, 5, 3), (20, 11, 3), (23, 14, 1), (28, 18, 3), (31, 21, 1)):
        T[b + so:b + so + n, c + to:c + to + n] = np.eye(n)
    quat(b + 16, c + 8); quat(b + 24, c + 15)
    for l in range(L):
        T[b + 32 + l, c + 22 + l] = 1.0
    return T


def _reference_system(P, free):
    """(residuals, tangent Jacobian, cost, H, g) from the fixture alone."""
    r, Jt = P["residuals"], P["J_ambient"] @ _tangent_map(P["state"], P["n_knots"], P["n_landmarks"])
    locked = np.ones(Jt.shape[1], bool); locked[free] = False
    Jt = Jt.copy(); Jt[:, locked] = 0.0
    huber = {3: P["huber_surf"], 4: P["huber_rep"], 5: P["huber_cs"]}
    cost, H, g = 0.0, np.zeros((Jt.shape[1],) * 2), np.zeros(Jt.shape[1])
    for b in np.unique(P["row_block"]):
        rows = np.where(P["row_block"] == b)[0]
        rb, Jb = r[rows], Jt[rows]
        s = float(rb @ rb)
        a = huber.get(int(P["row_family"][rows[0]]))
        rho, sc = s, 1.0
        if a is not None and s > a * a:
            rho, sc = 2 * a * np.sqrt(s) - a * a, np.sqrt(a / np.sqrt(s))   # ceres::HuberLoss; Corrector with rho'' <= 0: sqrt(rho') on rows
        cost += 0.5 * rho
        H += (sc * Jb).T @ (sc * Jb); g += (sc * Jb).T @ (sc * rb)
    return r, Jt, cost, H, g


def _check(ev, dense_jac, P, free, tol_h=1e-9):
    r, Jt, cost, H, g = _reference_system(P, free)
    assert ev["residuals"].shape == r.shape
    assert np.abs(ev["residuals"] - r).max() <= 1e-11 * max(np.abs(r).max(), 100.0)   # reprojection residuals are differences of ~1e3 px coordinates
    J = dense_jac(ev["jac_cols"], ev["jac_vals"], Jt.shape[1])
    # every row against its own largest entry; rows that vanish analytically (the reference observation reprojected into its own view) against
    # 1e-6 of their family's largest entry, or rounding noise would be divided by rounding noise
    fmax = np.array([max(np.abs(Jt[P["row_family"] == f]).max(), 1e-300) for f in P["row_family"]])[:, None]
    scale = np.maximum(np.abs(Jt).max(axis=1, keepdims=True), 1e-6 * fmax)
    assert (np.abs(J - Jt) / scale).max() <= 1e-9, "worst row-relative Jacobian error %.3e" % (np.abs(J - Jt) / scale).max()
    assert abs(ev["cost"] - cost) <= 1e-11 * cost
    d = np.sqrt(np.maximum(np.diag(H), 0)); sc = np.outer(d, d) + 1e-300
    assert (np.abs(ev["H"] - H) / sc).max() <= tol_h
    assert np.abs(ev["g"] - g).max() <= 1e-9 * np.abs(g).max()


def _free(P, locks):
    from oracle import lm
    return lm.free_tangent_indices(P["n_knots"], P["n_landmarks"], locks)


def test_fixture_exercises_every_family_and_the_huber_branch():
    P = _load()
    assert set(np.unique(P["row_family"])) == {0, 1, 2, 3, 4, 5}
    assert len(P["rep_lm"]) > 0 and (P["rep_t0"] == P["lm_t0"][P["rep_lm"]]).any()      # a block whose two views are the same frame
    r, _, _, _, _ = _reference_system(P, _free(P, TAU))
    big = [np.sum(r[P["row_block"] == b] ** 2) > 25.0 for b in np.unique(P["row_block"][P["row_family"] >= 3])]
    assert any(big) and not all(big)                                                        # both branches of the Huber loss


def test_variants_cover_what_they_claim():
    P = _load("tau")
    N = P["n_knots"]
    assert P["free_tau"] and P["state"][7 * N + 23] != 0 and P["state"][7 * N + 31] != 0
    Jt = P["J_ambient"] @ _tangent_map(P["state"], N, P["n_landmarks"])
    assert np.abs(Jt[P["row_family"] == 3][:, 6 * N + 14]).max() > 0        # d r / d tau_lidar on the surfel rows
    assert np.abs(Jt[P["row_family"] >= 4][:, 6 * N + 21]).max() > 0        # d r / d tau_cam on the reprojection / camera-surfel rows
    P = _load("radtan")
    assert abs(P["camera"]["k1"]) > 1e-5 and abs(P["camera"]["p2"]) > 0
    P = _load("solve0")
    assert P["so3_only"] and set(np.unique(P["row_family"])) == {0, 2}


@pytest.mark.parametrize("variant", VARIANTS)
def test_oracle_matches_the_ambient_restatement(variant):
    from oracle import oracle as O
    P = _load(variant)
    locks = _locks(P)
    o = O.Oracle()
    lvx.load_problem(o, P, locks)
    o.set_so3_only(P["so3_only"])
    o.set_orientation_prior(P["prior_t"], P["prior_q_wxyz"], P["prior_w"])
    ev = o.evaluate(P["state"], jac=True, normal_eq=True)
    _check(ev, O.dense_jacobian, P, _free(P, locks))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_gpu_matches_the_ambient_restatement(variant):
    from oracle import oracle as O   # dense_jacobian helper only
    P = _load(variant)
    locks = _locks(P)
    g = lvx.Context(0)
    lvx.load_problem(g, P, locks)
    g.set_orientation_prior(P["prior_t"], P["prior_q_wxyz"], P["prior_w"])
    ev = g.evaluate(P["state"], jac=True, normal_eq=True)          # per-segment kernels (debug Jacobian)
    _check(ev, O.dense_jacobian, P, _free(P, locks))
    ev2 = g.evaluate(P["state"], jac=False, normal_eq=True)        # fused MFMA path: residuals, cost, H, g
    ev2["jac_cols"], ev2["jac_vals"] = ev["jac_cols"], ev["jac_vals"]
    _check(ev2, O.dense_jacobian, P, _free(P, locks))
    assert g.layout()["exact_fallback"] == 0
    g.close()
