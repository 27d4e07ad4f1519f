"""oracle/lm_sparse.py (the full-size oracle LM: generic sparse normal equations, e-block Schur complement, RCM + LAPACK band Cholesky) against the dense oracle
LM of oracle/lm.py and against scipy's SuperLU — on the CPU, small sizes.  tests/test_gpu_converge_oracle.py then holds the GPU against it at full size."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import lvx
import synth
from oracle import lm, lm_sparse, pipeline
from oracle import oracle as O

TAU = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU


@pytest.fixture(scope="module")
def small():
    P = synth.make_problem(seed=33, duration=2.0, n_surfel=1500, n_planes=12, n_landmarks=40, n_camsurf=8)
    o = O.Oracle()
    lvx.load_problem(o, P, TAU)
    return P, o


def test_csr_jacobian_reproduces_the_dense_normal_equations(small):
    P, o = small
    x = P["state0"]
    ev = o.jacobian_csr(x)
    ref = o.evaluate(x, jac=True, normal_eq=True)
    assert ev["cost"] == pytest.approx(ref["cost"], rel=1e-14)
    assert np.array_equal(ev["residuals"], ref["residuals"])
    J = ev["J"]
    assert J.shape == (o.num_residuals, o.tangent_size)
    Hl = O.ata_lower(J)
    H = lm_sparse.sym_from_lower(Hl).toarray()
    s = np.abs(ref["H"]).max()
    assert np.abs(H - ref["H"]).max() <= 1e-13 * s
    g = J.T @ ev["r"]
    assert np.abs(g - ref["g"]).max() <= 1e-13 * np.abs(ref["g"]).max()
    # rows are the raw rows of orc_evaluate times the Corrector's scale: unscaled rows of the non-robust families agree exactly
    Jd = O.dense_jacobian(ref["jac_cols"], ref["jac_vals"], o.tangent_size)
    n_imu_rows = 6 * len(P["t_imu"])
    assert np.abs(J[:n_imu_rows].toarray() - Jd[:n_imu_rows]).max() == 0.0


def test_ata_lower_generic_matrix():
    rng = np.random.default_rng(5)
    A = sp.random(400, 60, density=0.08, random_state=rng, format="csr")
    A = sp.vstack([A, A[:7]]).tocsr()        # repeated rows
    L = O.ata_lower(A, threads=3)
    ref = (A.T @ A).toarray()
    assert np.abs(L.toarray() - np.tril(ref)).max() <= 1e-14 * np.abs(ref).max()
    assert (np.diff(L.indptr) >= 0).all() and all((np.diff(L.indices[L.indptr[j]:L.indptr[j + 1]]) > 0).all() for j in range(60))


@pytest.mark.parametrize("radius", [1e4, 3.0])
def test_sparse_step_equals_dense_step_and_superlu(small, radius):
    P, o = small
    N, L = P["n_knots"], P["n_landmarks"]
    x = P["state0"]
    free = lm.free_tangent_indices(N, L, TAU)
    ev = o.jacobian_csr(x)
    Hl = O.ata_lower(ev["J"])
    g = ev["J"].T @ ev["r"]
    ref = o.evaluate(x, normal_eq=True)
    scale = 1.0 / (1.0 + np.sqrt(np.diag(ref["H"])[free]))
    d_ref, m_ref, diag_ref = lm.solve_step(ref["H"], ref["g"], free, radius, scale)
    eb = np.nonzero(free >= 6 * N + 22)[0]
    assert len(eb) == L
    st = {}
    d_sp, m_sp, diag_sp = lm_sparse.solve_step(Hl, g, free, radius, scale, eblocks_free=eb, force_sparse=True, stats=st)
    d_dn, m_dn, _ = lm_sparse.solve_step(Hl, g, free, radius, scale, eblocks_free=eb)
    print("sparse path:", st)
    sc = np.abs(d_ref).max()
    assert np.abs(d_sp - d_ref).max() <= 1e-9 * sc and np.abs(d_dn - d_ref).max() <= 1e-9 * sc
    assert m_sp == pytest.approx(m_ref, rel=1e-10) and np.allclose(diag_sp, diag_ref, rtol=1e-12)
    # SuperLU on the same damped system
    Hs = sp.diags(scale) @ lm_sparse.sym_from_lower(sp.csc_matrix(Hl)[free][:, free]) @ sp.diags(scale)
    A = (Hs + sp.diags(diag_sp / radius)).tocsc()
    y = spla.splu(A).solve(-(g[free] * scale))
    assert np.abs(y * scale - d_sp[free]).max() <= 1e-9 * sc


def test_arrowhead_is_found_from_the_matrix_alone():
    """1/16 of config 4 (9.6 k unknowns): border = the map-time knots + the calibration scalars, band of a few knots after RCM; step against SuperLU."""
    P = synth.make_bench_problem(seed=4, n_imu=200_000 // 16, n_surfel=1_000_000 // 16, n_reproj=50_000 // 16, n_planes=64)
    N, L = P["n_knots"], P["n_landmarks"]
    o = O.Oracle()
    lvx.load_problem(o, P, TAU)
    free = lm.free_tangent_indices(N, L, TAU)
    ev = o.jacobian_csr(P["state0"])
    Hl = O.ata_lower(ev["J"])
    g = ev["J"].T @ ev["r"]
    scale = 1.0 / (1.0 + np.sqrt(Hl.diagonal()[free]))
    eb = np.nonzero(free >= 6 * N + 22)[0]
    st = {}
    d, m, diag = lm_sparse.solve_step(Hl, g, free, 1e4, scale, eblocks_free=eb, stats=st)
    print("sparse path:", st, "landmarks", len(eb))
    assert 20 + 18 <= st["n_border"] <= 22 + 36 and st["bandwidth"] <= 600 and st["n"] == len(free) - len(eb)
    Hs = sp.diags(scale) @ lm_sparse.sym_from_lower(sp.csc_matrix(Hl)[free][:, free]) @ sp.diags(scale)
    A = (Hs + sp.diags(diag / 1e4)).tocsc()
    res = A @ (d[free] / scale) + g[free] * scale
    assert np.linalg.norm(res) <= 1e-9 * np.linalg.norm(g[free] * scale)
    assert m > 0


def test_eblocks_must_be_uncoupled():
    A = sp.csc_matrix(np.array([[4.0, 0, 0], [1.0, 3.0, 0], [0.5, 0.2, 5.0]]))
    with pytest.raises(ValueError):
        lm_sparse.spd_solve(A, np.ones(3), eblocks=[0, 1], force_sparse=True)


def test_sparse_lm_follows_the_dense_lm():
    P = synth.make_bench_problem(seed=4, n_imu=200_000 // 256, n_surfel=1_000_000 // 256, n_reproj=50_000 // 256, n_planes=16, obs_per_frame=20)
    xd, logd = pipeline.run_fixed_stages(P, P["state0"])
    xs, logs = pipeline.run_fixed_stages(P, P["state0"], sparse=True)
    import scipy  # noqa: F401
    for (_, sd, _), (_, ss, _) in zip(logd, logs):
        assert sd["termination"] == ss["termination"] and sd["iterations"] == ss["iterations"]
        assert list(sd["accepted"]) == list(ss["accepted"])
        assert np.abs(sd["cost_history"] - ss["cost_history"]).max() <= 1e-9 * sd["cost_history"].max()
    assert np.abs(xd - xs).max() <= 1e-7
