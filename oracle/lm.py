"""CPU oracle of the Levenberg-Marquardt loop (TEST INFRASTRUCTURE ONLY — PARITY UNPINNED).

Restates, from Ceres' public semantics, what ceres::Solve does for the options set in
kontiki/trajectory_estimator.h:38-68 (TRUST_REGION + LEVENBERG_MARQUARDT, exact linear solve, Ceres defaults
otherwise): Jacobi scaling computed at the first iterate, LM diagonal clamp(diag(J_s^T J_s), 1e-6, 1e32) / radius
(reused after a rejected step), step acceptance by relative decrease > 1e-3, radius update
r / max(1/3, 1 - (2 rho - 1)^3), rejection r / decrease_factor (2, 4, ...), and the parameter / function / gradient
tolerance tests in the order of TrustRegionMinimizer::Minimize.  Dense numpy linear algebra on the oracle's J^T J.
Bounds (rho >= 0: static_rscamera_measurement.h:184-185, |free tau| <= max: sensors.h:70-85) make the problem CONSTRAINED in Ceres' sense
(Program::IsBoundsConstrained).  What TrustRegionMinimizer does then, restated from Ceres' public sources from memory (out-of-tree, PARITY UNPINNED):
  * every candidate is projected onto the box (ceres::ParameterBlock::Plus: oracle.plus), the start point too (IterationZero);
  * the gradient tolerance is tested on the PROJECTED gradient || x - Plus(x, -g) ||_inf (here: bounded scalars projected, unbounded blocks |g|);
  * the trust-region step goes through a projected Armijo line search before it is judged (DoLineSearch: sufficient decrease 1e-4 against g . delta, first trial
    step 1, contraction by the minimiser of the polynomial through every sample so far (values and directional derivatives: CUBIC interpolation evaluates the gradient
    at the trial points), clamped to [1e-3, 0.6] x the last step, at most 20 trials, minimum step 1e-9); delta is scaled by the step found, the step quality still
    divides by the model cost change of the UNSCALED step.  A full step that satisfies Armijo — the rule — changes nothing.
"""
import numpy as np


def interpolating_polynomial(samples):
    """Coefficients (highest power first) of the lowest-degree polynomial through `samples` = [(x, f, df or None), ...] (ceres FindInterpolatingPolynomial)."""
    rows, rhs = [], []
    n = sum(1 + (s[2] is not None) for s in samples)
    for x, f, df in samples:
        rows.append([x ** (n - 1 - k) for k in range(n)]); rhs.append(f)
        if df is not None:
            rows.append([(n - 1 - k) * x ** (n - 2 - k) if n - 1 - k > 0 else 0.0 for k in range(n)]); rhs.append(df)
    A, b = np.array(rows, dtype=np.float64), np.array(rhs, dtype=np.float64)
    # Gaussian elimination with partial pivoting (the same sequence of operations as lvx_solver.hip::poly_fit)
    for k in range(n):
        piv = k + int(np.argmax(np.abs(A[k:, k])))
        if piv != k:
            A[[k, piv]] = A[[piv, k]]; b[[k, piv]] = b[[piv, k]]
        for r in range(k + 1, n):
            m = A[r, k] / A[k, k]
            A[r, k:] -= m * A[k, k:]; b[r] -= m * b[k]
    c = np.zeros(n)
    for k in range(n - 1, -1, -1):
        c[k] = (b[k] - A[k, k + 1:] @ c[k + 1:]) / A[k, k]
    return c


def minimize_polynomial(c, lo, hi):
    """argmin of the polynomial on [lo, hi]: 2001 uniform samples, then 60 golden-section steps on the best bracket (deterministic; ceres MinimizePolynomial takes
    the real roots of the derivative and the end points)."""
    def val(x):
        v = 0.0
        for a in c:
            v = v * x + a
        return v
    n = 2000
    best, bi = None, 0
    for i in range(n + 1):
        v = val(lo + (hi - lo) * i / n)
        if best is None or v < best:
            best, bi = v, i
    a = lo + (hi - lo) * max(bi - 1, 0) / n
    b = lo + (hi - lo) * min(bi + 1, n) / n
    g = 0.6180339887498949
    x1, x2 = b - g * (b - a), a + g * (b - a)
    f1, f2 = val(x1), val(x2)
    for _ in range(60):
        if f1 <= f2:
            b, x2, f2 = x2, x1, f1
            x1 = b - g * (b - a); f1 = val(x1)
        else:
            a, x1, f1 = x1, x2, f2
            x2 = a + g * (b - a); f2 = val(x2)
    return 0.5 * (a + b)


def projected_line_search(eval_fg, f0, g0, f1, g1, max_trials=20, sufficient_decrease=1e-4, max_contraction=1e-3, min_contraction=0.6, min_step=1e-9, verbose=False, direction_max_norm=1.0):
    """Armijo search along the (projected) step: eval_fg(alpha) -> (cost, directional derivative) at Plus(x, alpha delta).  (f1, g1): the full step, already
    evaluated.  Returns (alpha, cost at alpha, trials) — alpha = 1 when the full step is kept (it satisfies Armijo, or no step does).  The search gives up when
    alpha * direction_max_norm (the infinity norm of the step, LineSearchFunction::DirectionInfinityNorm) falls under min_line_search_step_size, as ArmijoLineSearch does."""
    if not (g0 < 0.0) or not np.isfinite(f1) or f1 <= f0 + sufficient_decrease * g0:
        return 1.0, f1, 0
    prev, cur = None, (1.0, f1, g1)
    for trial in range(1, max_trials + 1):
        samples = [(0.0, f0, g0)] + ([prev] if prev is not None else []) + [cur]
        c = interpolating_polynomial(samples)
        a = minimize_polynomial(c, max_contraction * cur[0], min_contraction * cur[0])
        if verbose:
            print("  oracle line search trial %d: f0 %.12e g0 %.12e | last step %.6e f %.12e df %.12e -> step %.12e" % (trial, f0, g0, cur[0], cur[1], cur[2], a))
        if a * direction_max_norm < min_step:
            break
        f, g = eval_fg(a)
        if np.isfinite(f) and f <= f0 + sufficient_decrease * a * g0:
            return a, f, trial
        if not np.isfinite(f):        # a trial that cannot be evaluated: contract without a new sample
            cur = (a, cur[1], cur[2])
            continue
        prev, cur = cur, (a, f, g)
    return 1.0, f1, max_trials


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


def bounded_scalars(n_knots, n_landmarks, free, sensor_mto=1e-3):
    """[(tangent index, state index, lower, upper)] of the free scalars that carry a box constraint: inverse depths rho >= 0 (static_rscamera_measurement.h:184-185,
    camera_surfel_landmark.h:232), a free LiDAR / camera time offset |tau| <= max_time_offset (sensors.h:70-85, trajectory_manager_lvi.h:118-119)."""
    N, L = n_knots, n_landmarks
    fr = set(int(v) for v in np.asarray(free)[np.asarray(free) >= 6 * N + 14])
    out = []
    if 6 * N + 14 in fr:
        out.append((6 * N + 14, 7 * N + 23, -sensor_mto, sensor_mto))
    if 6 * N + 21 in fr:
        out.append((6 * N + 21, 7 * N + 31, -sensor_mto, sensor_mto))
    for l in range(L):
        if 6 * N + 22 + l in fr:
            out.append((6 * N + 22 + l, 7 * N + 32 + l, 0.0, np.inf))
    return out


def bounds_in_problem(oracle, bnd, n_knots):
    """A bounded block makes the problem constrained only if it is PART of the problem: Ceres' Program holds the parameter blocks some residual block uses, and
    Program::IsBoundsConstrained looks at those.  Solve #0 (gyroscope + prior) leaves the landmarks formally unlocked but no reprojection block exists — round 6: with them
    counted, the oracle ran Ceres' projected line search on Solve #0's rejected steps and accepted what the real minimizer (and the GPU's loop) rejects.  Inverse depths need
    a reprojection or camera-surfel block, the LiDAR offset a surfel or camera-surfel block, the camera offset a reprojection or camera-surfel block."""
    if not hasattr(oracle, "n_reproj"):
        return bnd
    has_rho = oracle.n_reproj + oracle.n_camsurf > 0
    has_tau = {6 * n_knots + 14: oracle.n_surfel + oracle.n_camsurf > 0, 6 * n_knots + 21: oracle.n_reproj + oracle.n_camsurf > 0}
    return [b for b in bnd if (has_tau[b[0]] if b[0] in has_tau else has_rho)]


def projected_gradient_max(x, g, free, bnd):
    """|| x - Plus(x, -g) ||_inf over the free scalars: bounded ones projected onto their box, unbounded ones |g| (TrustRegionMinimizer::ComputeGradientNorms for a
    constrained problem; the ambient difference of a quaternion block equals |g| to third order)."""
    pg = np.array(g, dtype=np.float64)
    for ti, si, lo, hi in bnd:
        pg[ti] = x[si] - min(max(x[si] - g[ti], lo), hi)
    return np.abs(pg[free]).max(initial=0.0)


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
             function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, jacobi_scaling=True, n_knots=None, n_landmarks=0, mask=None, constrained=None, sensor_mto=1e-3):
    """constrained: None = derive the box constraints from the free scalars (single sequence), [] = none (a joint multi-sequence problem: projection only)."""
    x = np.array(state, dtype=np.float64)
    free = np.asarray(free)
    if mask is None:   # ambient entries of the free parameter blocks (a joint multi-sequence problem passes its own)
        mask = free_state_mask(n_knots, n_landmarks, free)
    bnd = bounded_scalars(n_knots, n_landmarks, free, sensor_mto) if (constrained is None and n_knots is not None) else (constrained or [])
    if constrained is None and n_knots is not None:
        bnd = bounds_in_problem(oracle, bnd, n_knots)
    if bnd:
        x = oracle.plus(x, np.zeros(oracle.tangent_size))      # IterationZero: the start point projected onto the box
    ev = oracle.evaluate(x, normal_eq=True)
    cost, H, g = ev["cost"], ev["H"], ev["g"]
    scale = 1.0 / (1.0 + np.sqrt(np.maximum(np.diag(H)[free], 0.0))) if jacobi_scaling else np.ones(len(free))
    radius, dec = initial_radius, 2.0
    lm_diag = None
    hist = {"cost": [], "radius": [], "accepted": []}
    term = "max_iterations"
    it = 0
    invalid = 0
    if projected_gradient_max(x, g, free, bnd) <= gradient_tolerance:
        return x, dict(termination="gradient_tolerance", iterations=0, initial_cost=cost, final_cost=cost, **hist)
    init_cost = cost
    line_search_trials = []
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
        if bnd:      # projected Armijo line search on the trust-region step (TrustRegionMinimizer::DoLineSearch)
            g0 = float(g @ delta)
            if g0 < 0.0 and np.isfinite(cand) and cand > cost + 1e-4 * g0:
                def eval_fg(a):
                    try:
                        e = oracle.evaluate(oracle.plus(x, a * delta), normal_eq=True)
                        return e["cost"], float(e["g"] @ delta)
                    except (IndexError, ValueError):
                        return np.inf, 0.0
                a, fa, trials = projected_line_search(eval_fg, cost, g0, cand, eval_fg(1.0)[1], direction_max_norm=float(np.abs(delta).max()))
                line_search_trials.append(trials)
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
        if rho > min_relative_decrease:
            x = xc
            ev = oracle.evaluate(x, normal_eq=True)
            cost, H, g = ev["cost"], ev["H"], ev["g"]
            radius = min(max_radius, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            dec = 2.0
            lm_diag = None
            hist["cost"].append(cost); hist["radius"].append(radius); hist["accepted"].append(1)
            if projected_gradient_max(x, g, free, bnd) <= gradient_tolerance:
                term = "gradient_tolerance"
                break
        else:
            radius /= dec
            dec *= 2.0
            hist["cost"].append(cost); hist["radius"].append(radius); hist["accepted"].append(0)
            if radius < min_radius:
                term = "min_trust_region_radius"      # Ceres: CONVERGENCE, "minimum trust region radius reached"
                break
    return x, dict(termination=term, iterations=it, initial_cost=init_cost, final_cost=cost, line_search_trials=line_search_trials,
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
