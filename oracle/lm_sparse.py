"""CPU oracle of the Levenberg-Marquardt loop at FULL problem size (TEST INFRASTRUCTURE ONLY — PARITY UNPINNED).

The same trust-region loop as oracle/lm.py — Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy as configured by
kontiki/trajectory_estimator.h:38-68 (TRUST_REGION, LEVENBERG_MARQUARDT, SPARSE_SCHUR = an EXACT linear solve, Ceres defaults otherwise), restated from
Ceres' public semantics — but with sparse linear algebra, so that BASELINE config 4 (155 k unknowns) can be driven to a Ceres termination on the CPU
and the GPU solver (lvi-exc_amd/csrc/lvx_solver.hip: structured band / border / landmark rows, block cyclic reduction) is held against a step that
shares NOTHING with it:

  J            the oracle's robustified Jacobian as one generic CSR matrix (oracle.jacobian_csr: per-block stride-4 dual numbers, manifold, Corrector);
  H = J^T J    generic sparse product (oracle.ata_lower);
  e-blocks     what SPARSE_SCHUR eliminates first: the free inverse depths (no two share a residual, so their block is diagonal — asserted, not assumed);
               Schur complement S = B - E^T C^-1 E by the same generic product on C^-1/2 E;
  ordering     from the MATRIX alone: columns whose degree is far above the median are a dense border (the arrowhead: map-time knots, calibration),
               reverse Cuthill-McKee (scipy.sparse.csgraph) on the rest;
  factor       LAPACK band Cholesky (scipy.linalg.cholesky_banded) + a dense border Schur complement.

Small systems (or dense=True) take numpy's dense Cholesky like oracle/lm.py; tests/test_lm_sparse.py holds the two against each other and against
scipy's SuperLU.  Bounds (rho >= 0, |free tau| <= max) by projection inside oracle.plus, as oracle/lm.py.
"""
import time

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee

from . import lm
from . import oracle as O


def sym_from_lower(L):
    """full symmetric CSR from a lower-triangular sparse matrix"""
    L = sp.csr_matrix(L)
    return (L + sp.tril(L, -1).T).tocsr()


def spd_solve(A_lower, b, eblocks=None, dense_limit=3000, force_sparse=False, stats=None):
    """Solve A y = b for a sparse SPD A given by its lower triangle (any sparse format).  eblocks: indices of pairwise uncoupled scalars to eliminate first."""
    n = A_lower.shape[0]
    A_lower = sp.csc_matrix(A_lower)
    if n <= dense_limit and not force_sparse:
        Ad = A_lower.toarray()
        Ad = Ad + np.tril(Ad, -1).T
        Lc = np.linalg.cholesky(Ad)
        return sla.solve_triangular(Lc.T, sla.solve_triangular(Lc, b, lower=True, check_finite=False), lower=False, check_finite=False)
    A = sym_from_lower(A_lower)
    y = np.zeros(n)
    if eblocks is not None and len(eblocks):
        e = np.asarray(eblocks)
        is_e = np.zeros(n, bool); is_e[e] = True
        rest = np.nonzero(~is_e)[0]
        Cee = A[e][:, e]
        c = Cee.diagonal()
        if Cee.nnz != np.count_nonzero(c):   # an e-block coupled to another one: not a diagonal block
            raise ValueError("e-blocks are not pairwise uncoupled")
        if not (c > 0).all():
            raise np.linalg.LinAlgError("non-positive e-block pivot")
        E = A[e][:, rest].tocsr()                                   # [n_e, n_rest]
        F = sp.diags(1.0 / np.sqrt(c)) @ E
        B_lower = sp.tril(A[rest][:, rest], 0, format="csc")
        S_lower = (B_lower - O.ata_lower(F.tocsr())).tocsc()
        bs = b[rest] - E.T @ (b[e] / c)
        yr = _arrow_solve(S_lower, bs, stats)
        y[rest] = yr
        y[e] = (b[e] - E @ yr) / c
        return y
    return _arrow_solve(A_lower, b, stats)


def _arrow_solve(S_lower, b, stats=None):
    """SPD solve by ordering: dense border (degree > 10 x median) last, RCM + LAPACK band Cholesky on the rest."""
    n = S_lower.shape[0]
    S = sym_from_lower(S_lower)
    deg = np.diff(S.indptr)
    # dense columns: far above the typical column (a calibration scalar that only the rotation part of every knot sees has degree n / 2, the gyroscope bias)
    border = np.nonzero(deg > max(10 * int(np.median(deg)), 64))[0]
    is_b = np.zeros(n, bool); is_b[border] = True
    rest = np.nonzero(~is_b)[0]
    R = S[rest][:, rest].tocsr()
    perm = np.asarray(reverse_cuthill_mckee(R, symmetric_mode=True))
    Rp = R[perm][:, perm].tocoo()
    lo = Rp.row >= Rp.col
    i, j, v = Rp.row[lo], Rp.col[lo], Rp.data[lo]
    bw = int((i - j).max(initial=0))
    nr = len(rest)
    if (bw + 1) * nr * 8 > 8e9:
        raise MemoryError("band storage of %d x %d after RCM: the matrix is not an arrowhead" % (bw + 1, nr))
    ab = np.zeros((bw + 1, nr))
    ab[i - j, j] = v
    if stats is not None:
        stats.update(n=n, n_border=len(border), bandwidth=bw)
    cb = sla.cholesky_banded(ab, lower=True, overwrite_ab=True, check_finite=False)
    rhs = np.empty((nr, 1 + len(border)))
    rhs[:, 0] = b[rest][perm]
    if len(border):
        rhs[:, 1:] = S[rest][:, border].toarray()[perm]
    X = sla.cho_solve_banded((cb, True), rhs, overwrite_b=True, check_finite=False)
    y = np.zeros(n)
    z = X[:, 0]
    if len(border):
        Srb = S[rest][:, border].tocsr()[perm]          # [nr, nb]
        T = S[border][:, border].toarray() - Srb.T @ X[:, 1:]
        T = 0.5 * (T + T.T)
        Lc = np.linalg.cholesky(T)
        yb = sla.cho_solve((Lc, True), b[border] - Srb.T @ z, check_finite=False)
        z = z - X[:, 1:] @ yb
        y[border] = yb
    yr = np.empty(nr); yr[perm] = z
    y[rest] = yr
    return y


def solve_step(H_lower, g, free, radius, scale, lm_diag=None, eblocks_free=None, min_diag=1e-6, max_diag=1e32, force_sparse=False, stats=None):
    """(S H S + D) y = -S g on the free scalars (H_lower: lower triangle over all tangent scalars); returns (delta, model_cost_change, lm_diag)."""
    Hf = sp.csc_matrix(H_lower)[free][:, free]
    s = scale
    Hs = (sp.diags(s) @ Hf @ sp.diags(s)).tocsc()
    gs = g[free] * s
    if lm_diag is None:
        lm_diag = np.clip(Hs.diagonal(), min_diag, max_diag)
    A = (Hs + sp.diags(lm_diag / radius)).tocsc()
    y = spd_solve(A, -gs, eblocks=eblocks_free, force_sparse=force_sparse, stats=stats)
    Hy = Hs @ y + sp.tril(Hs, -1).T @ y
    model = -(gs @ y + 0.5 * (y @ Hy))
    delta = np.zeros(H_lower.shape[0])
    delta[free] = y * s
    return delta, model, lm_diag


def lm_solve(oracle, state, free, n_knots, n_landmarks, max_iterations=50, initial_radius=1e4, max_radius=1e16, min_radius=1e-32, min_relative_decrease=1e-3,
             function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, jacobi_scaling=True, force_sparse=False, verbose=False):
    """oracle/lm.py::lm_solve with the step from sparse linear algebra.  Same return value."""
    x = np.array(state, dtype=np.float64)
    free = np.asarray(free)
    mask = lm.free_state_mask(n_knots, n_landmarks, free)
    lm_first = 6 * n_knots + 22
    eb = np.nonzero(free >= lm_first)[0]          # positions (within `free`) of the free inverse depths: the e-blocks of SPARSE_SCHUR
    timing = {"jacobian": 0.0, "product": 0.0, "solve": 0.0, "cost": 0.0}

    def linearise(xx):
        t0 = time.perf_counter()
        ev = oracle.jacobian_csr(xx)
        t1 = time.perf_counter()
        H = O.ata_lower(ev["J"])
        g = ev["J"].T @ ev["r"]
        timing["jacobian"] += t1 - t0; timing["product"] += time.perf_counter() - t1
        return ev["cost"], H, g

    cost, H, g = linearise(x)
    diagH = H.diagonal()
    scale = 1.0 / (1.0 + np.sqrt(np.maximum(diagH[free], 0.0))) if jacobi_scaling else np.ones(len(free))
    radius, dec = initial_radius, 2.0
    lm_diag = None
    hist = {"cost": [], "radius": [], "accepted": []}
    term = "max_iterations"
    it = 0
    invalid = 0
    if np.abs(g[free]).max(initial=0.0) <= gradient_tolerance:
        return x, dict(termination="gradient_tolerance", iterations=0, initial_cost=cost, final_cost=cost, **hist)
    init_cost = cost
    stats = {}
    while it < max_iterations:
        it += 1
        t0 = time.perf_counter()
        try:
            delta, model, lm_diag = solve_step(H, g, free, radius, scale, lm_diag, eblocks_free=eb, force_sparse=force_sparse, stats=stats)
            ok = np.isfinite(model) and model > 0
        except np.linalg.LinAlgError:
            ok = False
        timing["solve"] += time.perf_counter() - t0
        if not ok:
            invalid += 1
            if invalid >= 5:          # max_num_consecutive_invalid_steps = 5
                term = "failure"
                break
            radius *= 0.5
            hist["cost"].append(cost); hist["radius"].append(radius); hist["accepted"].append(-1)
            continue
        invalid = 0
        xc = oracle.plus(x, delta)
        t0 = time.perf_counter()
        try:
            cand = oracle.evaluate(xc)["cost"]
        except (IndexError, ValueError):
            cand = np.inf
        timing["cost"] += time.perf_counter() - t0
        step_norm = np.linalg.norm((xc - x)[mask])
        x_norm = np.linalg.norm(x[mask])
        if step_norm <= parameter_tolerance * (x_norm + parameter_tolerance):
            term = "parameter_tolerance"
            hist["cost"].append(cost); hist["radius"].append(radius); hist["accepted"].append(0)
            break
        change = cost - cand
        if abs(change) <= function_tolerance * cost:
            term = "function_tolerance"
            hist["cost"].append(cost); hist["radius"].append(radius); hist["accepted"].append(0)
            break
        rho = change / model
        if verbose:
            print("  oracle LM %3d cost %.9e cand %.9e rho %.3f radius %.3e %s" % (it, cost, cand, rho, radius, stats), flush=True)
        if rho > min_relative_decrease:
            x = xc
            cost, H, g = linearise(x)
            radius = min(max_radius, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            dec = 2.0
            lm_diag = None
            hist["cost"].append(cost); hist["radius"].append(radius); hist["accepted"].append(1)
            if np.abs(g[free]).max(initial=0.0) <= gradient_tolerance:
                term = "gradient_tolerance"
                break
        else:
            radius /= dec
            dec *= 2.0
            hist["cost"].append(cost); hist["radius"].append(radius); hist["accepted"].append(0)
            if radius < min_radius:
                term = "min_trust_region_radius"      # Ceres: CONVERGENCE, "minimum trust region radius reached"
                break
    return x, dict(termination=term, iterations=it, initial_cost=init_cost, final_cost=cost, cost_history=np.array(hist["cost"]), radius_history=np.array(hist["radius"]),
                   accepted=np.array(hist["accepted"]), timing=timing, solver=stats)
