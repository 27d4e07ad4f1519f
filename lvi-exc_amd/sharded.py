"""Host logic for sharding INDEPENDENT calibration sequences one per GPU (BASELINE.json config 5).

Each rank owns one sequence: its trajectory, gravity, biases and landmarks are private; only the rig extrinsics
(lidar theta/p/tau, camera theta/p/tau = 14 tangent scalars) are shared.  One damped Gauss-Newton step of the JOINT problem is
    1. every rank eliminates its private variables (Schur complement onto the 14 shared scalars),
    2. ONE all-reduce (sum) of [S_shared (14x14) | rhs (14) | cost (1)]  — 211 doubles, latency-bound on xGMI / RCCL,
    3. every rank solves the same 14 x 14 system and back-substitutes its private part.
This module is backend-agnostic (torch.distributed: "nccl" = RCCL on ROCm, "gloo" on CPU for tests).  (A dense numpy restatement of the
reduction, used only as a test reference, lives in tests/sharded_reference.py.)

Second sharding axis: scan-level work — surfel association, scan registration, de-skew — is independent per scan: the scans of
a sequence are split into contiguous shards, one per rank, each rank runs the batched kernels on its shard against the same surfel map, and the
per-point results are all-gathered (no arithmetic collective).
"""
import numpy as np

N_SHARED = 14


def shared_tangent_indices(n_knots):
    """lidar theta(3) p(3) tau, cam theta(3) p(3) tau in the tangent layout of include/lvx.h."""
    return 6 * n_knots + 8 + np.arange(N_SHARED)


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


def scan_shard(n_scans, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_scans scans for `rank` of `world` (the first n_scans % world ranks get one scan more)."""
    base, extra = divmod(int(n_scans), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_scan_results(dist, local, n_scans, fill=-1):
    """All-gather per-scan result rows (torch tensor [n_local, row_len], e.g. the surfel-association flags of this rank's shard) into
    [n_scans, row_len] in scan order on every rank.  Shards may differ by one scan: they are padded to the largest for the collective."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return local
    per = [scan_shard(n_scans, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in per)
    row = local.shape[1]
    pad = torch.full((nmax, row), fill, dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * nmax, row), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * nmax:r * nmax + (hi - lo)] for r, (lo, hi) in enumerate(per)], dim=0)


def all_gather_scan_hits(dist, local, n_scans, fill=-1, capacity_frac=0.125):
    """The same result as all_gather_scan_results, gathered COMPACTLY: an association flags ~4 % of a scan's points (<= 2 points per ring and surfel), so each rank sends
    its hits as (flat index in its shard, value) pairs in a fixed-capacity buffer — capacity_frac of the shard's points + the count — instead of every point's flag:
    64 scans x 28 800 points are 7.4 MB of flags per rank and 0.9 MB of hits at the default capacity (1.8 MB incl. indices), and at N = 8 the gather, not the 0.11 ms kernel,
    sets the rate.  A rank whose hits outgrow the capacity makes every rank fall back to the dense gather (decided by the gathered counts themselves: no
    extra collective).  Returns [n_scans, row_len] in scan order on every rank, `fill` where nothing was flagged."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return local
    per = [scan_shard(n_scans, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in per)
    row = local.shape[1]
    cap = max(16, int(capacity_frac * nmax * row))
    flat = local.reshape(-1)
    idx = torch.nonzero(flat != fill).reshape(-1)                      # (one host synchronisation: the count)
    n = int(idx.numel())
    buf = torch.full((cap + 1, 2), -1, dtype=torch.int64, device=local.device)
    buf[0, 0] = n
    if n <= cap:
        buf[1:n + 1, 0] = idx
        buf[1:n + 1, 1] = flat[idx].to(torch.int64)
    out = torch.empty((world * (cap + 1), 2), dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    out = out.view(world, cap + 1, 2)
    counts = out[:, 0, 0]
    if bool((counts > cap).any()):                                     # every rank sees the same counts: all take the dense path together
        return all_gather_scan_results(dist, local, n_scans, fill)
    res = torch.full((n_scans, row), fill, dtype=local.dtype, device=local.device)
    for r, (lo, hi) in enumerate(per):
        k = int(counts[r])
        if k > 0:
            res[lo:hi].reshape(-1)[out[r, 1:k + 1, 0]] = out[r, 1:k + 1, 1].to(local.dtype)
    return res


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
