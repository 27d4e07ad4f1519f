"""Host logic for sharding INDEPENDENT calibration sequences one per GPU (BASELINE.json config 5).

Each rank owns one sequence: its trajectory, gravity, biases and landmarks are private; only the rig extrinsics
(lidar theta/p/tau, camera theta/p/tau = 14 tangent scalars) are shared.  One damped Gauss-Newton step of the JOINT problem is
    1. every rank eliminates its private variables (Schur complement onto the 14 shared scalars),
    2. ONE all-reduce (sum) of [S_shared (14x14) | rhs (14) | cost (1)]  — 211 doubles, latency-bound on xGMI / RCCL,
    3. every rank solves the same 14 x 14 system and back-substitutes its private part.
This module is backend-agnostic (torch.distributed: "nccl" = RCCL on ROCm, "gloo" on CPU for tests).  The dense algebra here is the
reference formulation the GPU solver's border reduction follows; it runs on whatever (H, g) the caller provides.
"""
import numpy as np

N_SHARED = 14


def shared_tangent_indices(n_knots):
    """lidar theta(3) p(3) tau, cam theta(3) p(3) tau in the tangent layout of include/lvx.h."""
    return 6 * n_knots + 8 + np.arange(N_SHARED)


def reduce_to_shared(H, g, free, shared, damping):
    """Schur-eliminate the private free scalars.  Returns (S, rhs, solver closure for the back substitution)."""
    free = np.asarray(free)
    shared = np.asarray(shared)
    is_sh = np.isin(free, shared)
    priv = free[~is_sh]
    sh = free[is_sh]
    App = H[np.ix_(priv, priv)] + np.diag(damping[priv])
    Aps = H[np.ix_(priv, sh)]
    Ass = H[np.ix_(sh, sh)]
    L = np.linalg.cholesky(App)
    Z = np.linalg.solve(L, Aps)
    z = np.linalg.solve(L, -g[priv])
    S = np.zeros((N_SHARED, N_SHARED))
    r = np.zeros(N_SHARED)
    pos = np.searchsorted(shared, sh)
    S[np.ix_(pos, pos)] = Ass - Z.T @ Z
    r[pos] = -g[sh] - Z.T @ z

    def back_substitute(y_shared):
        y = np.zeros(H.shape[0])
        y[sh] = y_shared[pos]
        y[priv] = np.linalg.solve(L.T, z - Z @ y_shared[pos])
        return y
    return S, r, back_substitute, pos


def sharded_step(H, g, free, shared, damping, cost, all_reduce):
    """One joint step.  `all_reduce(vec)` sums a float64 numpy vector over ranks in place.  Shared damping is added once, after the sum."""
    S, r, back, pos = reduce_to_shared(H, g, free, shared, damping)
    buf = np.concatenate([S.ravel(), r, [cost]])
    all_reduce(buf)
    S = buf[:N_SHARED * N_SHARED].reshape(N_SHARED, N_SHARED)
    r = buf[N_SHARED * N_SHARED:N_SHARED * N_SHARED + N_SHARED]
    Sd = S + np.diag(damping[np.asarray(shared)])
    live = np.zeros(N_SHARED, dtype=bool)
    live[pos] = True
    y = np.zeros(N_SHARED)
    y[live] = np.linalg.solve(Sd[np.ix_(live, live)], r[live])
    return back(y), float(buf[-1])


def torch_all_reduce(dist):
    import torch

    def fn(vec):
        t = torch.from_numpy(vec)
        dist.all_reduce(t)
        return vec
    return fn


def dist_all_reduce(dist, device=None):
    """allreduce(vec, op) for lvx.Context.lm_solve_shared over torch.distributed: 'gloo' reduces the host buffer directly,
    'nccl' (= RCCL on ROCm) stages the <= 211 doubles through a device tensor."""
    import torch

    def fn(vec, op):
        rop = dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX
        if device is None:
            dist.all_reduce(torch.from_numpy(vec), op=rop)
        else:
            t = torch.from_numpy(vec).to(device)
            dist.all_reduce(t, op=rop)
            vec[:] = t.cpu().numpy()
    return fn


class ThreadAllReduce:
    """In-process all-reduce between `world` threads (one lvx.Context per thread): used to drive several sequences on ONE GPU,
    e.g. the single-GPU parity test of the joint solve."""

    def __init__(self, world):
        import threading
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world

    def rank_fn(self, rank):
        def fn(vec, op):
            self.slots[rank] = vec.copy()
            self.bar.wait()
            stack = np.stack(self.slots)
            res = stack.sum(axis=0) if op == "sum" else stack.max(axis=0)
            self.bar.wait()
            vec[:] = res
        return fn
