"""GPU parity of the sequence-per-GPU JOINT solve (lvx_solve_step_shared / lvx_lm_solve_shared): two calibration
sequences with private trajectories / biases / landmarks and SHARED rig extrinsics.  Each sequence lives in its own lvx.Context; the two
ranks run in two threads of this process (one GPU) and meet in an in-process all-reduce (sharded.ThreadAllReduce) — the same callback
ABI takes torch.distributed (RCCL) on a multi-GPU node.  Reference: the joint problem assembled densely from the oracle's per-sequence
J^T J (private blocks block-diagonal, shared block summed) and solved by the numpy LM of oracle/lm.py.

Tolerances: one damped step within 1e-7 of the dense joint solve (relative to the largest entry); LM — same accept / reject sequence,
cost history within 1e-7 relative, extrinsics within 1e-6 rad / 1e-4 m, and bitwise-identical shared variables on both ranks.
"""
import threading

import numpy as np
import pytest

import lvx
import sharded
import synth
from oracle import lm
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TAU = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU
WORLD = 2


def _sequences():
    seqs = []
    ref = None
    for r in range(WORLD):
        P = synth.make_problem(seed=40 + r, duration=1.0 + 0.3 * r, n_surfel=250, n_planes=8, n_landmarks=12, n_camsurf=0)
        N = P["n_knots"]
        s = P["state0"].copy()
        if ref is None:
            ref = s[7 * N + 16:7 * N + 32].copy()
        s[7 * N + 16:7 * N + 32] = ref          # the shared extrinsics start from ONE guess
        seqs.append((P, s))
    return seqs


class JointOracle:
    """The joint problem for oracle/lm.py: state = [state_0 | state_1], tangent = [tangent_0 | tangent_1] with rank 1's shared scalars
    aliased onto rank 0's."""

    def __init__(self, seqs):
        self.P = [p for p, _ in seqs]
        self.o = []
        for P in self.P:
            o = O.Oracle(); lvx.load_problem(o, P, TAU); self.o.append(o)
        self.ns = [len(s) for _, s in seqs]
        self.nt = [o.tangent_size for o in self.o]
        self.sh = [sharded.shared_tangent_indices(P["n_knots"]) for P in self.P]
        free = [lm.free_tangent_indices(P["n_knots"], P["n_landmarks"], TAU) for P in self.P]
        self.free = np.concatenate([free[0], self.nt[0] + np.setdiff1d(free[1], self.sh[1])])
        m0 = lm.free_state_mask(self.P[0]["n_knots"], self.P[0]["n_landmarks"], free[0])
        m1 = lm.free_state_mask(self.P[1]["n_knots"], self.P[1]["n_landmarks"], free[1])
        N1 = self.P[1]["n_knots"]
        m1[7 * N1 + 16:7 * N1 + 32] = False     # shared blocks counted once
        self.mask = np.concatenate([m0, m1])

    @property
    def tangent_size(self):
        return self.nt[0] + self.nt[1]

    def split(self, x):
        return x[:self.ns[0]], x[self.ns[0]:]

    def evaluate(self, x, normal_eq=False):
        xs = self.split(x)
        ev = [o.evaluate(xi, normal_eq=normal_eq) for o, xi in zip(self.o, xs)]
        out = {"cost": ev[0]["cost"] + ev[1]["cost"]}
        if normal_eq:
            n0, n1 = self.nt
            H = np.zeros((n0 + n1, n0 + n1)); g = np.zeros(n0 + n1)
            H[:n0, :n0] = ev[0]["H"]; g[:n0] = ev[0]["g"]
            H1, g1 = ev[1]["H"], ev[1]["g"]
            idx = n0 + np.arange(n1)
            idx[self.sh[1]] = self.sh[0]        # alias
            np.add.at(H, (idx[:, None], idx[None, :]), H1)
            np.add.at(g, idx, g1)
            out["H"], out["g"] = H, g
        return out

    def plus(self, x, delta):
        n0 = self.nt[0]
        d0 = delta[:n0]
        d1 = delta[n0:].copy()
        d1[self.sh[1]] = d0[self.sh[0]]
        xs = self.split(x)
        return np.concatenate([self.o[0].plus(xs[0], d0), self.o[1].plus(xs[1], d1)])


def _joint_bounds(J, seqs):
    """The joint problem's box constraints for oracle/lm.py: every free inverse depth of every sequence, rho >= 0 (tangent index, state index, lower, upper)."""
    bnd = []
    for r, (P, _) in enumerate(seqs):
        N, L = P["n_knots"], P["n_landmarks"]
        t0, s0 = (0, 0) if r == 0 else (J.nt[0], J.ns[0])
        bnd += [(t0 + 6 * N + 22 + l, s0 + 7 * N + 32 + l, 0.0, np.inf) for l in range(L)]
    return bnd


def _run_ranks(seqs, body, locks=TAU):
    """body(rank, ctx, allreduce) in one thread per rank; returns the per-rank results."""
    ar = sharded.ThreadAllReduce(WORLD)
    res, err = [None] * WORLD, [None] * WORLD

    def work(r):
        try:
            g = lvx.Context(0)
            lvx.load_problem(g, seqs[r][0], locks)
            res[r] = body(r, g, ar.rank_fn(r))
            g.close()
        except BaseException as e:   # noqa: BLE001
            err[r] = e
            ar.bar.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(WORLD)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    for e in err:
        if e is not None:
            raise e
    return res


@pytest.mark.parametrize("radius", [1e4, 3.0])
def test_shared_step_equals_joint_dense_step(radius):
    seqs = _sequences()
    J = JointOracle(seqs)
    x = np.concatenate([s for _, s in seqs])
    ev = J.evaluate(x, normal_eq=True)
    scale = 1.0 / (1.0 + np.sqrt(np.maximum(np.diag(ev["H"])[J.free], 0)))
    d_ref, m_ref, _ = lm.solve_step(ev["H"], ev["g"], J.free, radius, scale)

    def body(r, g, allreduce):
        g.evaluate(seqs[r][1], normal_eq=True, dense=False)
        return g.solve_step_shared(radius, allreduce)
    res = _run_ranks(seqs, body)
    n0 = J.nt[0]
    d1_ref = d_ref[n0:].copy(); d1_ref[J.sh[1]] = d_ref[:n0][J.sh[0]]
    tol = 1e-7 * np.abs(d_ref).max()
    assert np.abs(res[0][0] - d_ref[:n0]).max() <= tol
    assert np.abs(res[1][0] - d1_ref).max() <= tol
    assert np.array_equal(res[0][0][J.sh[0]], res[1][0][J.sh[1]])       # the shared step is bitwise identical on both ranks
    for r in range(WORLD):
        assert abs(res[r][1] - m_ref) <= 1e-8 * abs(m_ref)               # model cost change of the JOINT problem


def test_shared_lm_matches_joint_oracle_lm():
    seqs = _sequences()
    J = JointOracle(seqs)
    x0 = np.concatenate([s for _, s in seqs])
    xo, so = lm.lm_solve(J, x0, J.free, max_iterations=12, mask=J.mask, constrained=_joint_bounds(J, seqs))   # free inverse depths: the joint problem is constrained (line search)
    res = _run_ranks(seqs, lambda r, g, allreduce: g.lm_solve_shared(seqs[r][1], allreduce, max_iterations=12))
    for r in range(WORLD):
        sg = res[r][1]
        assert sg["iterations"] == so["iterations"] and sg["termination"] == so["termination"]
        assert list(sg["accepted"]) == list(so["accepted"])
        assert np.abs(sg["cost_history"] - so["cost_history"]).max() <= 1e-7 * so["cost_history"].max()
    N = [P["n_knots"] for P, _ in seqs]
    e = [res[r][0][7 * N[r] + 16:7 * N[r] + 32] for r in range(WORLD)]
    assert np.array_equal(e[0], e[1])                                    # one set of extrinsics
    eo = J.split(xo)[0][7 * N[0] + 16:7 * N[0] + 32]
    for off in (0, 8):
        d = synth.qmul(e[0][off:off + 4], synth.qconj(eo[off:off + 4]))
        assert 2 * np.arctan2(np.linalg.norm(d[:3]), abs(d[3])) <= 1e-6
        assert np.abs(e[0][off + 4:off + 7] - eo[off + 4:off + 7]).max() <= 1e-4
    assert res[0][1]["final_cost"] < 1e-2 * res[0][1]["initial_cost"]


def test_collectives_per_iteration_and_failure_leaves_together():
    """The joint loop's schedule: ONE reduction after the first evaluation and exactly two per iteration (the reduced 14 x 14 system; the candidate's cost, model
    terms, norms and joint diagonal / gradient block) — and a rank whose evaluation fails (a measurement outside its spline) makes EVERY rank return instead of leaving the other inside a reduction."""
    seqs = _sequences()

    def body(r, g, allreduce):
        g.collective_count(reset=True)
        x, s = g.lm_solve_shared(seqs[r][1], allreduce, max_iterations=6)
        return s, g.collective_count()
    res = _run_ranks(seqs, body, locks=TAU | lvx.LOCK_LANDMARKS)      # nothing bounded is free: the unconstrained schedule
    for s, n in res:
        acc = int(np.sum(np.asarray(s["accepted"]) == 1))
        assert n == 1 + 2 * s["iterations"] + (1 if s["termination"] == "max_iterations" and s["accepted"][-1] != 1 else 0) and acc >= 1
    # free inverse depths (rho >= 0): the joint problem is constrained — one more reduction per iteration (the candidate's joint cost decides on the line search), and when a
    # search runs, one for the directional derivative at the full step and one per trial
    res = _run_ranks(seqs, body)
    for s, n in res:
        base = 1 + 3 * s["iterations"] + (1 if s["termination"] == "max_iterations" and s["accepted"][-1] != 1 else 0)
        assert n >= base and n <= base + 22 * s["iterations"]
    assert res[0][1] == res[1][1]                                        # both ranks met the same collectives
    # rank 1's sequence has an IMU sample beyond the spline: its evaluation returns LVX_E_RANGE; rank 0 must come back with LVX_E_COMM
    P1 = dict(seqs[1][0]); P1["t_imu"] = P1["t_imu"].copy(); P1["t_imu"][-1] = P1["t0"] + (P1["n_knots"] - 3) * P1["dt"] + 0.5
    bad = [seqs[0], (P1, seqs[1][1])]
    codes = [None, None]

    def body2(r, g, allreduce):
        try:
            g.lm_solve_shared(bad[r][1], allreduce, max_iterations=3)
        except lvx.LvxError as e:
            codes[r] = e.code
        return None
    _run_ranks(bad, body2)
    assert codes == [lvx.E_RCCL, lvx.E_RANGE]


def test_rccl_transport_single_rank_equals_plain_solve():
    """The RCCL transport (ncclAllReduce on the context's stream; step block packed, reduced and solved on the device) with a world of one rank — all this
    box has — must reproduce the single-sequence solve; the multi-rank protocol is the one the callback transport tests above."""
    P, x0 = _sequences()[0]
    g = lvx.Context(0)
    # inverse depths constant: the plain two-reductions-per-iteration schedule asserted below (free ones make the problem CONSTRAINED in Ceres' sense — projected line search,
    # projected gradient norm — in the single-sequence AND, since round 5, in the joint solve, which then takes a third reduction per iteration)
    lvx.load_problem(g, P, TAU | lvx.LOCK_LANDMARKS)
    xa, sa = g.lm_solve(x0, max_iterations=8)
    g.rccl_init(g.rccl_unique_id(), 0, 1)
    g.collective_count(reset=True)
    xb, sb = g.lm_solve_shared(x0, None, max_iterations=8)
    assert g.joint_shared_count() == 14       # the 14 extrinsic scalars stayed out of the local elimination: the reduced 14 x 14 system went through ncclAllReduce
    # one reduction after the first evaluation, two per iteration, and one vote when the iteration cap ends the loop right behind a rejected step (its restore has met no collective yet)
    assert g.collective_count() == 1 + 2 * sb["iterations"] + (1 if sb["termination"] == "max_iterations" and sb["accepted"][-1] != 1 else 0)
    g.rccl_finalize()
    assert sa["iterations"] == sb["iterations"] and list(sa["accepted"]) == list(sb["accepted"]) and sa["termination"] == sb["termination"]
    assert np.abs(sa["cost_history"] - sb["cost_history"]).max() <= 1e-9 * sa["cost_history"].max()
    N = P["n_knots"]
    assert np.abs(xa[7 * N:7 * N + 32] - xb[7 * N:7 * N + 32]).max() <= 1e-7
    g.close()


def test_rccl_and_callback_transports_agree_on_one_rank():
    """Both transports of the joint solve on the same sequence, world of one: the host-callback transport (an identity all-reduce) and the RCCL transport
    (pack -> ncclAllReduce -> 14 x 14 solve on the device) run the SAME reduced-system protocol — shared block kept out of the local elimination, shared
    damping added once after the reduction — so their iterates agree to rounding, and both report 14 shared scalars."""
    P, x0 = _sequences()[0]
    N = P["n_knots"]
    g = lvx.Context(0)
    lvx.load_problem(g, P, TAU | lvx.LOCK_LANDMARKS)   # (free inverse depths would make the plain solve below a constrained one: see the test above)
    xa, sa = g.lm_solve_shared(x0, lambda buf, op: None, max_iterations=8)
    assert g.joint_shared_count() == 14
    g.rccl_init(g.rccl_unique_id(), 0, 1)
    xb, sb = g.lm_solve_shared(x0, None, max_iterations=8)
    assert g.joint_shared_count() == 14
    g.rccl_finalize()
    xc, sc = g.lm_solve(x0, max_iterations=8)
    assert g.joint_shared_count() == 0
    assert sa["iterations"] == sb["iterations"] and list(sa["accepted"]) == list(sb["accepted"]) and sa["termination"] == sb["termination"]
    assert np.abs(sa["cost_history"] - sb["cost_history"]).max() <= 1e-9 * sa["cost_history"].max()     # the pass adds with atomics: 1e-11 from run to run
    assert np.abs(xa - xb).max() <= 1e-7
    assert np.abs(xa[7 * N:7 * N + 32] - xc[7 * N:7 * N + 32]).max() <= 1e-7
    g.close()


def _with_landmarks_at_infinity(P, n_inf, seed):
    """Landmarks 0 .. n_inf - 1 moved to infinity (true inverse depth 0): their observations are the oracle's own prediction at rho = 0 plus pixel noise, so about half of
    them want a NEGATIVE inverse depth (tests/test_gpu_solver.py::test_constrained_problem_line_search_and_bound_at_the_solution, here per sequence)."""
    N = P["n_knots"]
    rng = np.random.default_rng(seed)
    xt = P["state_true"].copy()
    xt[7 * N + 32:7 * N + 32 + n_inf] = 0.0
    ot = O.Oracle(); lvx.load_problem(ot, P, TAU)
    r = ot.evaluate(xt)["residuals"]
    n_imu, n_surf = len(P["t_imu"]), len(P["surf_t"])
    r_rep = r[6 * n_imu + n_surf:6 * n_imu + n_surf + 2 * len(P["rep_lm"])].reshape(-1, 2)
    sel = np.isin(P["rep_lm"], np.arange(n_inf))
    Q = dict(P)
    uv = P["rep_uv"].copy()
    uv[sel] = (P["rep_uv"][sel] - r_rep[sel] / P["w_rep"]) + 0.5 * rng.standard_normal((int(sel.sum()), 2))
    Q["rep_uv"] = uv
    return Q


def test_joint_solve_is_constrained_like_ceres_line_search_and_bound():
    """Free inverse depths (rho >= 0) make the JOINT problem constrained in Ceres' sense: projected start, projected gradient, projected Armijo line search on the
    trust-region step — with every quantity the search decides on summed over the ranks.  Two sequences with landmarks at infinity whose observations want negative inverse
    depths: the joint oracle LM (oracle/lm.py with the joint problem's box constraints) and the two-rank GPU solve take the same contracted steps, and the same landmarks
    end ON the bound."""
    seqs = []
    ref = None
    for r in range(WORLD):
        # (one trajectory recorded twice with different pixel noise on the landmarks at infinity: on this pair the joint search contracts a step — 11 trials — and
        #  4 + 2 landmarks end on the bound; pairs of different trajectories scanned with oracle/lm.py never triggered the search)
        P = synth.make_problem(seed=29, duration=2.0, n_surfel=800, n_planes=12, n_landmarks=30, n_camsurf=0)
        P = _with_landmarks_at_infinity(P, 5, 7 + r)
        N = P["n_knots"]
        s = P["state0"].copy()
        if ref is None:
            ref = s[7 * N + 16:7 * N + 32].copy()
        s[7 * N + 16:7 * N + 32] = ref
        seqs.append((P, s))
    J = JointOracle(seqs)
    x0 = np.concatenate([s for _, s in seqs])
    xo, so = lm.lm_solve(J, x0, J.free, max_iterations=30, mask=J.mask, constrained=_joint_bounds(J, seqs))
    assert sum(so["line_search_trials"]) >= 1                            # the search really contracted a joint step
    res = _run_ranks(seqs, lambda r, g, allreduce: g.lm_solve_shared(seqs[r][1], allreduce, max_iterations=30))
    conv = ("parameter_tolerance", "function_tolerance", "gradient_tolerance")
    xs = J.split(xo)
    for r in range(WORLD):
        sg = res[r][1]
        N, L = seqs[r][0]["n_knots"], seqs[r][0]["n_landmarks"]
        rho_o, rho_g = xs[r][7 * N + 32:7 * N + 32 + L], res[r][0][7 * N + 32:7 * N + 32 + L]
        print("rank %d: %s / %s, %d / %d iterations, trials %s, rho on the bound %d / %d" % (r, so["termination"], sg["termination"], so["iterations"], sg["iterations"], so["line_search_trials"],
                                                                                          int((rho_o == 0).sum()), int((rho_g == 0).sum())))
        # Both sides END in a tail of rejected steps whose 20-trial searches find nothing (the landmarks sit on the bound, the cost is flat to 1e-9): how many of those
        # iterations run before a tolerance fires depends on the last bits of the sums (8 - 11 on either side from run to run).  What is compared is everything before it.
        assert sg["termination"] in conv and so["termination"] in conv and abs(sg["iterations"] - so["iterations"]) <= 4
        k = min(len(sg["accepted"]), len(so["accepted"]), 8) - 1
        assert k >= 5 and list(sg["accepted"][:k]) == list(so["accepted"][:k])
        assert abs(sg["final_cost"] - so["final_cost"]) <= 1e-7 * so["final_cost"]
        assert np.abs(sg["cost_history"][:k] - so["cost_history"][:k]).max() <= 1e-7 * so["cost_history"].max()
        assert (rho_g >= 0).all() and list(rho_o == 0) == list(rho_g == 0)
        assert np.abs(rho_g - rho_o).max() <= 1e-6 * max(1.0, np.abs(rho_o).max())
    assert (np.concatenate([xs[r][7 * seqs[r][0]["n_knots"] + 32:] for r in range(WORLD)]) == 0).any()
    N0, N1 = seqs[0][0]["n_knots"], seqs[1][0]["n_knots"]
    assert np.array_equal(res[0][0][7 * N0 + 16:7 * N0 + 32], res[1][0][7 * N1 + 16:7 * N1 + 32])   # one set of extrinsics
