"""Config 4 at FULL size (1 M surfel + 200 k gyro + 200 k accel + 50 k reprojection blocks, 25 k knots, 150 k tangent scalars) against the CPU oracle,
matrix-free — the dense J^T J (180 GB) is the only thing that does not fit, so everything else is compared:

(i)   every residual row, the cost, g = J^T r and diag(J^T J) of the HIP pass against the oracle's dual-number evaluation of the same 1.45 M blocks
      (one OpenMP pass over all host cores: oracle/lvx_oracle.cpp::orc_evaluate_products);
(ii)  the damped step of the block-cyclic-reduction solver is checked against the linear system IT SHOULD SOLVE, built from the ORACLE's Jacobian:
      || S J^T (J delta) + D_s S^-1 delta + S g ||  <=  tol * || S g ||, D_s = clamp(S^2 diag(J^T J), 1e-6, 1e32) / radius, with and without Jacobi scaling
      (what one iteration of ceres::Solve computes: TRUST_REGION / LEVENBERG_MARQUARDT / SPARSE_SCHUR exact step, kontiki/trajectory_estimator.h:38-68);
(iii) the same step from the sequential band Cholesky (LVX_SOLVER_SEQ=1) — an independent elimination order — agrees with the cyclic-reduction step.
"""
import os

import numpy as np
import pytest

import lvx
import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TAU = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU
RADIUS = 1e4


@pytest.fixture(scope="module")
def full():
    P = synth.make_bench_problem(seed=4)
    g = lvx.Context(0)
    o = O.Oracle()
    for obj in (g, o):
        lvx.load_problem(obj, P, TAU)
    x = P["state0"]
    rg = g.evaluate(x, normal_eq=True, dense=False)
    gg, dg = g.gradient()
    steps = {}
    for name, scaling in (("scaled", True), ("unscaled", False)):
        steps[name] = g.solve_step(RADIUS, scaling)
    assert g.layout()["solver_fallbacks"] == 0
    g.set_switch("SOLVER_SEQ", 1)
    g.evaluate(x, normal_eq=True, dense=False, residuals=False)
    steps["scaled_seq"] = g.solve_step(RADIUS, True)
    g.set_switch("SOLVER_SEQ", 0)
    lo = g.layout()
    rows = g.family_rows()
    g.close()
    ro = o.evaluate_products(x, V=np.stack([steps["scaled"][0], steps["unscaled"][0]]))
    return dict(P=P, rg=rg, gg=gg, dg=dg, steps=steps, ro=ro, lo=lo, rows=rows, n_blocks=o.num_blocks)


def test_every_residual_row_and_the_cost_match_the_oracle(full):
    rg, ro, lo = full["rg"], full["ro"], full["lo"]
    assert lo["n_blocks"] == full["n_blocks"] >= 1_449_000 and lo["exact_fallback"] == 0
    assert len(rg["residuals"]) == len(ro["residuals"]) == lo["n_residuals"] >= 2_299_000   # 3 x (200 k + 200 k) IMU rows + 1 M surfel rows + 2 x 50 k reprojection rows
    err = np.abs(rg["residuals"] - ro["residuals"])
    rows = full["rows"]
    for f, name in enumerate(("gyro", "accel", "prior", "surfel", "reproj", "camsurf")):
        a, b = rows[f], rows[f + 1]
        if b == a:
            continue
        # every row of a family against the family's residual scale (a pixel residual is the difference of two ~600 px numbers: rows near zero carry the
        # rounding of their operands, not of their own size)
        fam_scale = np.abs(ro["residuals"][a:b]).max()
        print("%-7s %8d rows: max |err| / max |r| = %.3e (max |r| %.3e)" % (name, b - a, err[a:b].max() / fam_scale, fam_scale))
        assert err[a:b].max() <= 1e-11 * fam_scale
    print("cost rel err %.3e" % (abs(rg["cost"] - ro["cost"]) / ro["cost"]))
    assert abs(rg["cost"] - ro["cost"]) <= 1e-12 * abs(ro["cost"])


def _block_scale(v, n_knots):
    """max |v| per parameter block kind (positions, rotations, each calibration block, inverse depths): entries are compared relative to their own block's scale."""
    s = np.empty_like(v)
    k = v[:6 * n_knots].reshape(n_knots, 6)
    sk = np.empty_like(k)
    sk[:, :3] = np.abs(k[:, :3]).max()
    sk[:, 3:] = np.abs(k[:, 3:]).max()
    s[:6 * n_knots] = sk.ravel()
    b = 6 * n_knots
    for lo_, hi_ in ((0, 2), (2, 5), (5, 8), (8, 11), (11, 14), (14, 15), (15, 18), (18, 21), (21, 22)):
        s[b + lo_:b + hi_] = max(np.abs(v[b + lo_:b + hi_]).max(), 1e-300)
    s[b + 22:] = max(np.abs(v[b + 22:]).max(), 1e-300) if len(v) > b + 22 else 1.0
    return s


def test_gradient_and_diagonal_match_the_oracle(full):
    n = full["lo"]["n_knots"]
    for name, a, b in (("g", full["gg"], full["ro"]["g"]), ("diag", full["dg"], full["ro"]["diag"])):
        rel = np.abs(a - b) / _block_scale(b, n)
        print("%s: max block-scaled err %.3e" % (name, rel.max()))
        assert rel.max() <= 1e-10
    assert (full["ro"]["diag"] >= 0).all() and np.count_nonzero(full["ro"]["diag"]) >= 150_000


@pytest.mark.parametrize("which,idx,scaling", [("scaled", 0, True), ("unscaled", 1, False)])
def test_cyclic_reduction_step_solves_the_oracles_damped_system(full, which, idx, scaling):
    ro = full["ro"]
    delta, mcc = full["steps"][which]
    H_delta, g, diag = ro["HV"][idx], ro["g"], ro["diag"]
    S = 1.0 / (1.0 + np.sqrt(diag)) if scaling else np.ones_like(diag)
    D = np.clip(S * S * diag, 1e-6, 1e32) / RADIUS
    res = S * H_delta + D * (delta / S) + S * g
    rel = np.linalg.norm(res) / np.linalg.norm(S * g)
    model = -(g @ delta) - 0.5 * (delta @ H_delta)
    print("%s step: |residual| / |S g| = %.3e, model cost change %.9e (lvx %.9e)" % (which, rel, model, mcc))
    assert rel <= 1e-8
    assert abs(model - mcc) <= 1e-8 * abs(model) and model > 0


def test_sequential_band_cholesky_gives_the_same_step(full):
    d1, m1 = full["steps"]["scaled"]
    d2, m2 = full["steps"]["scaled_seq"]
    n = full["lo"]["n_knots"]
    rel = np.abs(d1 - d2) / _block_scale(d2, n)
    print("BCR vs sequential band Cholesky: max block-scaled step difference %.3e" % rel.max())
    assert rel.max() <= 1e-7
    assert abs(m1 - m2) <= 1e-9 * abs(m2)


# ---- BASELINE config 3 at full size: the IMU-only layout bench.py times (secondary.config3_imu_only) — no hub knots, k_imu_own owns every band column ----
C3_LOCKS = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU | lvx.LOCK_LIDAR_Q | lvx.LOCK_LIDAR_P | lvx.LOCK_CAM_Q | lvx.LOCK_CAM_P | lvx.LOCK_LANDMARKS


@pytest.fixture(scope="module")
def config3():
    P = synth.make_bench_problem(seed=4)
    Q = dict(P)
    Q.update(surf_pt=P["surf_pt"][:0], surf_t=P["surf_t"][:0], surf_plane=P["surf_plane"][:0], rep_lm=P["rep_lm"][:0], rep_uv=P["rep_uv"][:0], rep_t0=P["rep_t0"][:0],
             cs_lm=P["cs_lm"][:0], cs_plane=P["cs_plane"][:0])      # exactly bench.py's construction
    g = lvx.Context(0)
    o = O.Oracle()
    for obj in (g, o):
        lvx.load_problem(obj, Q, C3_LOCKS)
    x = P["state0"]
    rg = g.evaluate(x, normal_eq=True, dense=False)
    gg, dg = g.gradient()
    step = g.solve_step(RADIUS, True)
    lo, rows = g.layout(), g.family_rows()
    g.close()
    ro = o.evaluate_products(x, V=np.stack([step[0]]))
    return dict(rg=rg, gg=gg, dg=dg, step=step, ro=ro, lo=lo, rows=rows, n_blocks=o.num_blocks)


def test_config3_full_size_matches_the_oracle(config3):
    """200 k gyroscope + 200 k accelerometer blocks on 25 k knots: every residual row, the cost, g = J^T r and diag(J^T J) of the HIP pass against the oracle's
    dual-number evaluation (kontiki/measurements/gyroscope_measurement.h:36-38, accelerometer_measurement.h:39-41, sensors/imu.h:61-101)."""
    rg, ro, lo = config3["rg"], config3["ro"], config3["lo"]
    assert lo["n_blocks"] == config3["n_blocks"] == 400_000 and lo["exact_fallback"] == 0 and lo["n_hub_knots"] == 0
    assert len(rg["residuals"]) == len(ro["residuals"]) == 1_200_000
    err = np.abs(rg["residuals"] - ro["residuals"])
    rows = config3["rows"]
    for f, name in ((0, "gyro"), (1, "accel")):
        a, b = rows[f], rows[f + 1]
        scale = np.abs(ro["residuals"][a:b]).max()
        print("%-5s %7d rows: max |err| / max |r| = %.3e" % (name, b - a, err[a:b].max() / scale))
        assert b - a == 600_000 and err[a:b].max() <= 1e-11 * scale
    assert abs(rg["cost"] - ro["cost"]) <= 1e-12 * abs(ro["cost"])
    n = lo["n_knots"]
    for name, a, b in (("g", config3["gg"], ro["g"]), ("diag", config3["dg"], ro["diag"])):
        rel = np.abs(a - b) / _block_scale(b, n)
        print("%s: max block-scaled err %.3e" % (name, rel.max()))
        assert rel.max() <= 1e-10
    # locked blocks stay out: LiDAR / camera extrinsics and the landmarks have no gradient, the IMU calibration has
    base = 6 * n
    assert not config3["gg"][base + 8:].any() and config3["gg"][base:base + 8].all()


def test_config3_step_solves_the_oracles_damped_system(config3):
    ro = config3["ro"]
    delta, mcc = config3["step"]
    H_delta, g, diag = ro["HV"][0], ro["g"], ro["diag"]
    S = 1.0 / (1.0 + np.sqrt(diag))
    D = np.clip(S * S * diag, 1e-6, 1e32) / RADIUS
    res = S * H_delta + D * (delta / S) + S * g
    free = diag > 0
    rel = np.linalg.norm(res[free]) / np.linalg.norm((S * g)[free])
    model = -(g @ delta) - 0.5 * (delta @ H_delta)
    print("config 3 step: |residual| / |S g| = %.3e, model cost change %.9e (lvx %.9e)" % (rel, model, mcc))
    assert rel <= 1e-8 and not delta[~free].any()
    assert abs(model - mcc) <= 1e-8 * abs(model) and model > 0


# ---- the camera-landmark-to-surfel family at scale: config 4 + 20 000 cam-surfel blocks, the third stage's lock mask and everything free ----
@pytest.mark.parametrize("locks_name", ["all_free", "stage3"])
def test_config4_with_camera_surfel_blocks_matches_the_oracle(locks_name):
    """The fused cam-surfel kernel (k_family_mfma<CamSurfAcc>) is otherwise only compared on small problems: here 20 000 blocks (every landmark against four surfels) ride on
    the full config-4 problem — residual rows, cost, gradient and diag(J^T J) against the oracle's dual-number pass; `stage3` = trajInitFromLVIdata with lm_splane
    (trajectory_manager_lvi.cpp:197-257: trajectory and LiDAR extrinsics locked, camera extrinsics + inverse depths + IMU calibration free)."""
    P = dict(synth.make_bench_problem(seed=4))
    rng = np.random.default_rng(9)
    L = P["n_landmarks"]
    P["cs_lm"] = np.repeat(np.arange(L, dtype=np.int32), 4)[:20_000]
    P["cs_plane"] = rng.integers(0, len(P["planes"]), len(P["cs_lm"])).astype(np.int32)
    locks = TAU if locks_name == "all_free" else (TAU | lvx.LOCK_TRAJ | lvx.LOCK_LIDAR_Q | lvx.LOCK_LIDAR_P)
    g = lvx.Context(0)
    o = O.Oracle()
    for obj in (g, o):
        lvx.load_problem(obj, P, locks)
    x = P["state0"]
    rg = g.evaluate(x, normal_eq=True, dense=False)
    gg, dg = g.gradient()
    lo, rows = g.layout(), g.family_rows()
    g.close()
    ro = o.evaluate_products(x)
    assert lo["exact_fallback"] == 0 and lo["fallback_rows"] == 0 and rows[6] - rows[5] == len(P["cs_lm"])
    err = np.abs(rg["residuals"] - ro["residuals"])
    for f, name in enumerate(("gyro", "accel", "prior", "surfel", "reproj", "camsurf")):
        a, b = rows[f], rows[f + 1]
        if b > a:
            scale = np.abs(ro["residuals"][a:b]).max()
            print("%-7s %8d rows: max |err| / max |r| = %.3e" % (name, b - a, err[a:b].max() / scale))
            assert err[a:b].max() <= 1e-11 * scale
    assert abs(rg["cost"] - ro["cost"]) <= 1e-12 * abs(ro["cost"])
    n = lo["n_knots"]
    for name, a, b in (("g", gg, ro["g"]), ("diag", dg, ro["diag"])):
        sc = _block_scale(b, n)
        live = sc > 0                                   # (a locked block kind has scale 0: it must be exactly zero on both sides)
        assert not a[~live].any() and not b[~live].any()
        rel = np.abs(a - b)[live] / sc[live]
        print("%s (%s): max block-scaled err %.3e" % (name, locks_name, rel.max()))
        assert rel.max() <= 1e-10
    if locks_name == "stage3":
        assert not gg[:6 * n].any() and not gg[6 * n + 8:6 * n + 14].any() and gg[6 * n + 15:6 * n + 21].any()   # knots and LiDAR extrinsics constant, camera extrinsics live


# ---- variants of config 4 at full size that the default fixture does not reach ----
@pytest.mark.parametrize("variant", ["free_time_offsets", "radtan", "solve0"])
def test_config4_variants_match_the_oracle_at_full_size(variant):
    """free_time_offsets: lvi.yaml's opt_time_offset — both sensor offsets free and NON-ZERO (the fused kernels' time-offset column, 5-control-point segments, views that
    change knot interval); radtan: the distortion model of lvi.yaml:54-78 (8 fixed-point iterations in Unproject); solve0: initialSO3TrajWithGyro — 200 k gyroscope blocks +
    the orientation prior on the SO3 spline alone (k_family_mfma<GyroAcc>).  Residual rows 1e-11 per family, cost 1e-12, g and diag(J^T J) 1e-10 block-scaled."""
    if variant == "radtan":
        P = dict(synth.make_bench_problem(seed=4))
        P["camera"] = dict(P["camera"], k1=-0.0397646985948, k2=0.00802944041788, p1=-0.0043042199686, p2=-0.0001040279967, k3=0.00030608999077)
    else:
        P = dict(synth.make_bench_problem(seed=4))
    N = P["n_knots"]
    x = P["state0"].copy()
    locks = TAU
    if variant == "free_time_offsets":
        locks = 0
        x[7 * N + 23], x[7 * N + 31] = 6.5e-4, -8e-4
    g = lvx.Context(0)
    o = O.Oracle()
    for obj in (g, o):
        if variant == "solve0":
            Q = dict(P, surf_pt=P["surf_pt"][:0], surf_t=P["surf_t"][:0], surf_plane=P["surf_plane"][:0], rep_lm=P["rep_lm"][:0], rep_uv=P["rep_uv"][:0], rep_t0=P["rep_t0"][:0],
                     acc=np.zeros_like(P["acc"]))
            lvx.load_problem(obj, Q, lvx.LOCK_R3 | lvx.LOCK_ACC_BIAS | lvx.LOCK_GYRO_BIAS | TAU)
            obj.set_so3_only(True)
            obj.set_orientation_prior(P["t0"], np.array([np.cos(5e-5), 0, 0, np.sin(5e-5)]), 28.0)
        else:
            lvx.load_problem(obj, P, locks)
    rg = g.evaluate(x, normal_eq=True, dense=False)
    gg, dg = g.gradient()
    lo, rows = g.layout(), g.family_rows()
    g.close()
    ro = o.evaluate_products(x)
    assert lo["exact_fallback"] == 0 and lo["fallback_rows"] == 0
    err = np.abs(rg["residuals"] - ro["residuals"])
    for f, name in enumerate(("gyro", "accel", "prior", "surfel", "reproj", "camsurf")):
        a, b = rows[f], rows[f + 1]
        if b > a:
            scale = max(np.abs(ro["residuals"][a:b]).max(), 100.0 if name == "reproj" else 0.0)   # (a pixel residual is the difference of two ~1e3 px numbers)
            print("%-18s %-7s %8d rows: max |err| / scale = %.3e" % (variant, name, b - a, err[a:b].max() / scale))
            assert err[a:b].max() <= 1e-11 * scale
    assert abs(rg["cost"] - ro["cost"]) <= 1e-12 * abs(ro["cost"])
    for name, a, b in (("g", gg, ro["g"]), ("diag", dg, ro["diag"])):
        sc = _block_scale(b, N)
        live = sc > 0
        assert not a[~live].any() and not b[~live].any()
        rel = np.abs(a - b)[live] / sc[live]
        print("%-18s %s: max block-scaled err %.3e" % (variant, name, rel.max()))
        assert rel.max() <= 1e-10
    if variant == "free_time_offsets":
        assert gg[6 * N + 14] != 0.0 and gg[6 * N + 21] != 0.0 and dg[6 * N + 14] > 0 and dg[6 * N + 21] > 0
    if variant == "solve0":
        assert rows[1] - rows[0] == 600_000 and rows[3] - rows[2] == 1 and not gg[:6 * N].reshape(N, 6)[:, :3].any()
