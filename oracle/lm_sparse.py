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


class ArrowSolver:
    """Sparse SPD solve  A y = b  (A by its lower triangle) in three generic steps — Schur complement of the e-blocks (pairwise uncoupled scalars), dense border
    (columns whose degree is far above the median) last, reverse Cuthill-McKee + LAPACK band Cholesky on the rest.  The symbolic part (which columns are border,
    the RCM order) depends on the sparsity pattern only: it is computed at the first solve and reused while the pattern still fits it (an LM run solves ~20 systems
    with one pattern)."""

    def __init__(self):
        self.inv = None
        self.stats = {}

    def _analyse(self, A_lower, eblocks):
        n = A_lower.shape[0]
        A = sym_from_lower(A_lower)
        is_e = np.zeros(n, bool)
        if eblocks is not None and len(eblocks):
            is_e[np.asarray(eblocks)] = True
        e = np.nonzero(is_e)[0]
        rest0 = np.nonzero(~is_e)[0]
        S = A[rest0][:, rest0].tocsr()
        S.data[:] = 1.0
        if len(e):
            E = A[e][:, rest0].tocsr()
            E.data[:] = 1.0
            S = (S + sym_from_lower(O.ata_lower(E))).tocsr()     # pattern of B - E^T C^-1 E
        deg = np.diff(S.indptr)
        # dense columns: far above the typical column (a calibration scalar that only the rotation part of every knot sees has degree n / 2, the gyroscope bias)
        is_b = deg > max(10 * int(np.median(deg)), 64)
        border, band = np.nonzero(is_b)[0], np.nonzero(~is_b)[0]
        R = S[band][:, band].tocsr()
        perm = np.asarray(reverse_cuthill_mckee(R, symmetric_mode=True))
        Rp = R[perm][:, perm].tocoo()
        bw = int(np.abs(Rp.row - Rp.col).max(initial=0))
        order = np.concatenate([rest0[band][perm], rest0[border], e])      # new index -> old index
        self.inv = np.empty(n, np.int64); self.inv[order] = np.arange(n)
        self.order, self.n, self.nr, self.nb, self.ne, self.bw = order, n, len(band), len(border), len(e), bw
        if (bw + 1) * self.nr * 8 > 8e9:
            raise MemoryError("band storage of %d x %d after RCM: the matrix is not an arrowhead" % (bw + 1, self.nr))
        self.stats.update(n=n - len(e), n_border=len(border), bandwidth=bw, analyses=self.stats.get("analyses", 0) + 1)

    def solve(self, A_lower, b, eblocks=None):
        A_lower = sp.coo_matrix(A_lower)
        A_lower.eliminate_zeros()                                  # structural zeros (a Jacobian column that is exactly zero in a block) carry no pattern
        n = A_lower.shape[0]
        n_e = 0 if eblocks is None else len(eblocks)
        fresh = self.inv is None or self.n != n or self.ne != n_e
        while True:
            if fresh:
                self._analyse(A_lower, eblocks)
            nr, nb, ne, bw = self.nr, self.nb, self.ne, self.bw
            i, j, v = self.inv[A_lower.row], self.inv[A_lower.col], A_lower.data
            sw = i < j
            i, j = np.where(sw, j, i), np.where(sw, i, j)
            m = nr + nb
            ee = j >= m                                            # both indices in the e-block range
            if (i[ee] != j[ee]).any():
                raise ValueError("e-blocks are not pairwise uncoupled")
            if not ((i < nr) & (i - j > bw)).any():
                break
            if fresh:
                raise MemoryError("pattern does not fit its own analysis")
            fresh = True                                           # the pattern outgrew the cached ordering: analyse again
        bn = b[self.order]
        if ne:
            c = np.bincount(i[ee] - m, weights=v[ee], minlength=ne)          # (duplicate entries add, as everywhere below)
            if not (c > 0).all():
                raise np.linalg.LinAlgError("non-positive e-block pivot")
            fe = (i >= m) & ~ee
            F = sp.csr_matrix((v[fe] / np.sqrt(c[i[fe] - m]), (i[fe] - m, j[fe])), shape=(ne, m))      # C^-1/2 E
            FtF = O.ata_lower(F).tocoo()
            keep = i < m
            i = np.concatenate([i[keep], FtF.row]); j = np.concatenate([j[keep], FtF.col]); v = np.concatenate([v[keep], -FtF.data])
            fs = ((i < nr) & (i - j > bw)).any()
            if fs:
                raise MemoryError("Schur complement of the e-blocks outgrew the cached ordering")
            be = bn[m:]
            bs = bn[:m] - F.T @ (be / np.sqrt(c))
        else:
            bs = bn
        # band part -> LAPACK lower band storage, border rows -> dense
        inb = i < nr
        ab = np.bincount((i[inb] - j[inb]) * nr + j[inb], weights=v[inb], minlength=(bw + 1) * nr).reshape(bw + 1, nr)
        cb = sla.cholesky_banded(ab, lower=True, overwrite_ab=True, check_finite=False)
        rhs = np.empty((nr, 1 + nb))
        rhs[:, 0] = bs[:nr]
        if nb:
            rb = ~inb & (j < nr)
            Srb = np.bincount(j[rb] * nb + (i[rb] - nr), weights=v[rb], minlength=nr * nb).reshape(nr, nb)
            rhs[:, 1:] = Srb
            bb = ~inb & (j >= nr)
            T = np.bincount((i[bb] - nr) * nb + (j[bb] - nr), weights=v[bb], minlength=nb * nb).reshape(nb, nb)
            T = T + np.tril(T, -1).T
        X = sla.cho_solve_banded((cb, True), rhs, overwrite_b=True, check_finite=False)
        z = X[:, 0]
        ym = np.empty(nr + nb)
        if nb:
            T = T - Srb.T @ X[:, 1:]
            T = 0.5 * (T + T.T)
            Lc = np.linalg.cholesky(T)
            yb = sla.cho_solve((Lc, True), bs[nr:] - Srb.T @ z, check_finite=False)
            z = z - X[:, 1:] @ yb
            ym[nr:] = yb
        ym[:nr] = z
        yn = np.empty(n)
        yn[:nr + nb] = ym
        if ne:
            yn[nr + nb:] = be / c - (F @ ym) / np.sqrt(c)
        y = np.empty(n)
        y[self.order] = yn
        return y


def spd_solve(A_lower, b, eblocks=None, dense_limit=3000, force_sparse=False, stats=None, solver=None):
    """Solve A y = b for a sparse SPD A given by its lower triangle (any sparse format).  eblocks: indices of pairwise uncoupled scalars to eliminate first
    (checked); solver: an ArrowSolver whose symbolic analysis is reused."""
    n = A_lower.shape[0]
    if n <= dense_limit and not force_sparse:
        Ad = sp.csc_matrix(A_lower).toarray()
        Ad = Ad + np.tril(Ad, -1).T
        Lc = np.linalg.cholesky(Ad)
        return sla.solve_triangular(Lc.T, sla.solve_triangular(Lc, b, lower=True, check_finite=False), lower=False, check_finite=False)
    solver = solver if solver is not None else ArrowSolver()
    y = solver.solve(A_lower, b, eblocks)
    if stats is not None:
        stats.update(solver.stats)
    return y


def solve_step(H_lower, g, free, radius, scale, lm_diag=None, eblocks_free=None, min_diag=1e-6, max_diag=1e32, force_sparse=False, stats=None, solver=None):
    """(S H S + D) y = -S g on the free scalars (H_lower: lower triangle over all tangent scalars); returns (delta, model_cost_change, lm_diag)."""
    nt = H_lower.shape[0]
    H = sp.coo_matrix(H_lower)
    pos = np.full(nt, -1, np.int64); pos[free] = np.arange(len(free))
    i, j = pos[H.row], pos[H.col]
    k = (i >= 0) & (j >= 0)
    i, j = i[k], j[k]
    s = scale
    v = H.data[k] * s[i] * s[j]                                     # S H S, lower triangle over the free scalars
    gs = g[free] * s
    dg = np.zeros(len(free)); np.add.at(dg, i[i == j], v[i == j])
    if lm_diag is None:
        lm_diag = np.clip(dg, min_diag, max_diag)
    nf = len(free)
    A = sp.coo_matrix((np.concatenate([v, lm_diag / radius]), (np.concatenate([i, np.arange(nf)]), np.concatenate([j, np.arange(nf)]))), shape=(nf, nf))
    y = spd_solve(A, -gs, eblocks=eblocks_free, force_sparse=force_sparse, stats=stats, solver=solver)
    off = i != j
    Hy = np.bincount(i, weights=v * y[j], minlength=nf) + np.bincount(j[off], weights=v[off] * y[i[off]], minlength=nf)
    model = -(gs @ y + 0.5 * (y @ Hy))
    delta = np.zeros(nt)
    delta[free] = y * s
    return delta, model, lm_diag


def lm_solve(oracle, state, free, n_knots, n_landmarks, max_iterations=50, initial_radius=1e4, max_radius=1e16, min_radius=1e-32, min_relative_decrease=1e-3,
             function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, jacobi_scaling=True, force_sparse=False, verbose=False, sensor_mto=1e-3):
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

    bnd = lm.bounds_in_problem(oracle, lm.bounded_scalars(n_knots, n_landmarks, free, sensor_mto), n_knots)    # box constraints of the blocks that are in the problem: Ceres' constrained-problem behaviour (oracle/lm.py header)
    if bnd:
        x = oracle.plus(x, np.zeros(oracle.tangent_size))
    cost, H, g = linearise(x)
    diagH = H.diagonal()
    scale = 1.0 / (1.0 + np.sqrt(np.maximum(diagH[free], 0.0))) if jacobi_scaling else np.ones(len(free))
    radius, dec = initial_radius, 2.0
    lm_diag = None
    hist = {"cost": [], "radius": [], "accepted": []}
    term = "max_iterations"
    it = 0
    invalid = 0
    if lm.projected_gradient_max(x, g, free, bnd) <= gradient_tolerance:
        return x, dict(termination="gradient_tolerance", iterations=0, initial_cost=cost, final_cost=cost, **hist)
    init_cost = cost
    stats = {}
    solver = ArrowSolver()
    while it < max_iterations:
        it += 1
        t0 = time.perf_counter()
        try:
            delta, model, lm_diag = solve_step(H, g, free, radius, scale, lm_diag, eblocks_free=eb, force_sparse=force_sparse, stats=stats, solver=solver)
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
        if bnd:      # projected Armijo line search on the trust-region step (TrustRegionMinimizer::DoLineSearch)
            g0 = float(g @ delta)
            if g0 < 0.0 and np.isfinite(cand) and cand > cost + 1e-4 * g0:
                def eval_fg(a):
                    try:
                        e = oracle.jacobian_csr(oracle.plus(x, a * delta))
                        return e["cost"], float((e["J"].T @ e["r"]) @ delta)
                    except (IndexError, ValueError):
                        return np.inf, 0.0
                a, fa, _ = lm.projected_line_search(eval_fg, cost, g0, cand, eval_fg(1.0)[1], direction_max_norm=float(np.abs(delta).max()))
                if a != 1.0:
                    delta = a * delta
                    xc = oracle.plus(x, delta)
                    cand = fa
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
            if lm.projected_gradient_max(x, g, free, bnd) <= gradient_tolerance:
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
