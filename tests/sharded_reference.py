"""Dense numpy restatement of the sequence-per-GPU reduction (TEST REFERENCE ONLY): what lvx_solve_step_shared does on the GPU — every rank eliminates
its private variables, one sum over the ranks of [S_shared (14 x 14) | rhs (14) | cost], the same 14 x 14 solve everywhere, back substitution."""
import numpy as np

N_SHARED = 14


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


