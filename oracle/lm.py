"""CPU oracle of the Levenberg-Marquardt loop (TEST INFRASTRUCTURE ONLY — PARITY UNPINNED).

Restates, from Ceres' public semantics, what ceres::Solve does for the options set in
kontiki/trajectory_estimator.h:38-68 (TRUST_REGION + LEVENBERG_MARQUARDT, exact linear solve, Ceres defaults
otherwise): Jacobi scaling computed at the first iterate, LM diagonal clamp(diag(J_s^T J_s), 1e-6, 1e32) / radius
(reused after a rejected step), step acceptance by relative decrease > 1e-3, radius update
r / max(1/3, 1 - (2 rho - 1)^3), rejection r / decrease_factor (2, 4, ...), and the parameter / function / gradient
tolerance tests in the order of TrustRegionMinimizer::Minimize.  Dense numpy linear algebra on the oracle's J^T J.
Bounds (rho >= 0, |free tau| <= max) are enforced by projection inside oracle.plus (ceres::ParameterBlock::Plus); the projected line search
Ceres adds for constrained problems is not restated.
"""
import numpy as np


def free_state_mask(n_knots, n_landmarks, free_tangent):
    """Ambient state entries that belong to parameter blocks with at least one free tangent scalar."""
    N, L = n_knots, n_landmarks
    m = np.zeros(7 * N + 32 + L, dtype=bool)
    ft = np.zeros(6 * N + 22 + L, dtype=bool)
    ft[free_tangent] = True
    for k in range(N):
        if ft[6 * k]:
            m[3 * k:3 * k + 3] = True
        if ft[6 * k + 3]:
            m[3 * N + 4 * k:3 * N + 4 * k + 4] = True
    b, c = 7 * N, 6 * N
    for so, to, n in ((8, 0, 1), (9, 1, 1), (10, 2, 3), (13, 5, 3), (16, 8, 4), (20, 11, 3), (23, 14, 1), (24, 15, 4), (28, 18, 3), (31, 21, 1)):
        if ft[c + to]:
            m[b + so:b + so + n] = True
    for l in range(L):
        if ft[c + 22 + l]:
            m[b + 32 + l] = True
    return m


def solve_step(H, g, free, radius, scale=None, lm_diag=None, min_diag=1e-6, max_diag=1e32):
    """(S H S + D^2) y = -S g on the free scalars; returns (delta, model_cost_change, lm_diag)."""
    Hf = H[np.ix_(free, free)]
    gf = g[free]
    s = np.ones(len(free)) if scale is None else scale
    Hs = Hf * s[:, None] * s[None, :]
    gs = gf * s
    if lm_diag is None:
        lm_diag = np.clip(np.diag(Hs), min_diag, max_diag)
    A = Hs + np.diag(lm_diag / radius)
    Lc = np.linalg.cholesky(A)
    y = -np.linalg.solve(Lc.T, np.linalg.solve(Lc, gs))
    model = -(gs @ y + 0.5 * y @ (Hs @ y))
    delta = np.zeros(H.shape[0])
    delta[free] = y * s
    return delta, model, lm_diag


def lm_solve(oracle, state, free, max_iterations=50, initial_radius=1e4, max_radius=1e16, min_radius=1e-32, min_relative_decrease=1e-3,
             function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, jacobi_scaling=True, n_knots=None, n_landmarks=0, mask=None):
    x = np.array(state, dtype=np.float64)
    free = np.asarray(free)
    if mask is None:   # ambient entries of the free parameter blocks (a joint multi-sequence problem passes its own)
        mask = free_state_mask(n_knots, n_landmarks, free)
    ev = oracle.evaluate(x, normal_eq=True)
    cost, H, g = ev["cost"], ev["H"], ev["g"]
    scale = 1.0 / (1.0 + np.sqrt(np.maximum(np.diag(H)[free], 0.0))) if jacobi_scaling else np.ones(len(free))
    radius, dec = initial_radius, 2.0
    lm_diag = None
    hist = {"cost": [], "radius": [], "accepted": []}
    term = "max_iterations"
    it = 0
    invalid = 0
    if np.abs(g[free]).max(initial=0.0) <= gradient_tolerance:
        return x, dict(termination="gradient_tolerance", iterations=0, initial_cost=cost, final_cost=cost, **hist)
    init_cost = cost
    while it < max_iterations:
        it += 1
        try:
            delta, model, lm_diag = solve_step(H, g, free, radius, scale, lm_diag)
            ok = np.isfinite(model) and model > 0
        except np.linalg.LinAlgError:
            ok = False
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
        try:
            cand = oracle.evaluate(xc)["cost"]
        except (IndexError, ValueError):
            cand = np.inf
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
        if rho > min_relative_decrease:
            x = xc
            ev = oracle.evaluate(x, normal_eq=True)
            cost, H, g = ev["cost"], ev["H"], ev["g"]
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
    return x, dict(termination=term, iterations=it, initial_cost=init_cost, final_cost=cost,
                   cost_history=np.array(hist["cost"]), radius_history=np.array(hist["radius"]), accepted=np.array(hist["accepted"]))


def free_tangent_indices(n_knots, n_landmarks, locks):
    """Tangent scalars that are not constant under `locks` (same rule as lvx_resid.h tangent_locked)."""
    N, L = n_knots, n_landmarks
    out = []
    for g in range(6 * N + 22 + L):
        if g < 6 * N:
            locked = bool(locks & 1) or ((g % 6) < 3 and bool(locks & 2))
        else:
            c = g - 6 * N
            bit = (None if c < 2 else 8 if c < 5 else 9 if c < 8 else 2 if c < 11 else 3 if c < 14 else 4 if c < 15 else 5 if c < 18 else 6 if c < 21 else 7 if c < 22 else 10)
            locked = False if bit is None else bool(locks & (1 << bit))
        if not locked:
            out.append(g)
    return np.array(out, dtype=np.int64)
