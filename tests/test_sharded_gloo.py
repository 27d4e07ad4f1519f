"""Multi-process (world_size 2, gloo, CPU) test of the sequence-per-GPU sharding protocol (lvi-exc_amd/sharded.py):
the sharded step — private Schur complement, ONE all-reduce of the 14 x 14 shared-extrinsics system, back substitution —
must equal the step of the joint problem assembled in one process."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import lvx
import sharded
import sharded_reference
import synth
from oracle import lm
from oracle import oracle as O

TAU = O.LOCK_LIDAR_TAU | O.LOCK_CAM_TAU


def _sequence(rank):
    P = synth.make_problem(seed=40 + rank, duration=0.8, n_surfel=150, n_planes=8, n_landmarks=10, n_camsurf=0)
    o = O.Oracle(); lvx.load_problem(o, P, TAU)
    # both sequences start from the SAME extrinsics guess (they are shared variables)
    s = P["state0"].copy()
    N = P["n_knots"]
    ref = synth.make_problem(seed=40, duration=0.8, n_surfel=1, n_planes=1, n_landmarks=0)["state0"]
    Nr = synth.make_problem(seed=40, duration=0.8, n_surfel=1, n_planes=1, n_landmarks=0)["n_knots"]
    s[7 * N + 16:7 * N + 32] = ref[7 * Nr + 16:7 * Nr + 32]
    ev = o.evaluate(s, normal_eq=True)
    free = lm.free_tangent_indices(N, P["n_landmarks"], TAU)
    return P, ev, free


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, ev, free = _sequence(rank)
    nt = len(ev["g"])
    damping = np.zeros(nt); damping[free] = np.clip(np.diag(ev["H"])[free], 1e-6, 1e32) / 1e4
    shared = sharded.shared_tangent_indices(P["n_knots"])
    # shared damping must be the JOINT diagonal: sum it once
    dsh = np.diag(ev["H"])[shared].copy()
    t = torch.from_numpy(dsh); dist.all_reduce(t)
    damping[shared] = np.clip(dsh, 1e-6, 1e32) / 1e4
    y, total_cost = sharded_reference.sharded_step(ev["H"], ev["g"], free, shared, damping, ev["cost"], sharded_reference.torch_all_reduce(dist))
    out.put((rank, y, total_cost))
    dist.barrier()
    dist.destroy_process_group()


def test_two_sequences_sharded_step_equals_joint_step():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, y, c = q.get(timeout=300)
        res[r] = (y, c)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # joint problem in one process: block-diagonal private parts + summed shared block
    seqs = [_sequence(r) for r in range(2)]
    sizes = [len(s[1]["g"]) for s in seqs]
    shared = [sharded.shared_tangent_indices(s[0]["n_knots"]) for s in seqs]
    priv = [np.setdiff1d(s[2], sh) for s, sh in zip(seqs, shared)]
    npv = [len(p) for p in priv]
    n = sum(npv) + sharded.N_SHARED
    H = np.zeros((n, n)); g = np.zeros(n)
    off = 0
    sh0 = sum(npv)
    for (P, ev, free), sh, pv, m in zip(seqs, shared, priv, npv):
        H[off:off + m, off:off + m] = ev["H"][np.ix_(pv, pv)]
        H[off:off + m, sh0:] = ev["H"][np.ix_(pv, sh)]
        H[sh0:, off:off + m] = ev["H"][np.ix_(sh, pv)]
        H[sh0:, sh0:] += ev["H"][np.ix_(sh, sh)]
        g[off:off + m] = ev["g"][pv]; g[sh0:] += ev["g"][sh]
        off += m
    D = np.clip(np.diag(H), 1e-6, 1e32) / 1e4
    y_joint = np.linalg.solve(H + np.diag(D), -g)
    off = 0
    for r, ((P, ev, free), sh, pv, m) in enumerate(zip(seqs, shared, priv, npv)):
        y_r, cost = res[r]
        assert np.abs(y_r[pv] - y_joint[off:off + m]).max() <= 1e-7 * np.abs(y_joint).max()
        assert np.abs(y_r[sh] - y_joint[sh0:]).max() <= 1e-7 * np.abs(y_joint).max()
        assert abs(cost - sum(s[1]["cost"] for s in seqs)) <= 1e-9 * cost
        off += m
    assert np.abs(res[0][0][shared[0]] - res[1][0][shared[1]]).max() <= 1e-12 * np.abs(y_joint).max()   # every rank gets the same shared step


def _ar_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fn = sharded.dist_all_reduce(dist)          # the callback lvx.Context.lm_solve_shared hands to lvx_lm_solve_shared
    a = np.arange(5, dtype=np.float64) + 10 * rank
    b = a.copy()
    fn(a, "sum"); fn(b, "max")
    out.put((rank, a, b))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_callback_sum_and_max():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ar_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, a, b in got:
        assert np.array_equal(a, 2 * np.arange(5) + 10.0) and np.array_equal(b, np.arange(5) + 10.0)
    th = sharded.ThreadAllReduce(1).rank_fn(0)
    v = np.array([1.0, 2.0]); th(v, "sum")
    assert np.array_equal(v, [1.0, 2.0])


def _scan_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_scans = 5                                   # 3 + 2: uneven shards
    _, p4, bmin, bmax = synth.make_assoc_problem(seed=70, H=4, W=300, n_planes=60)
    lo, hi = sharded.scan_shard(n_scans, rank, world)
    local = [O.surfel_assoc(synth.make_assoc_problem(seed=70 + s, H=4, W=300, n_planes=60)[0], p4, bmin, bmax, 0.05, 2).ravel() for s in range(lo, hi)]
    mine = torch.from_numpy(np.stack(local).astype(np.int32))
    flags = sharded.all_gather_scan_results(dist, mine, n_scans)
    compact = sharded.all_gather_scan_hits(dist, mine, n_scans)                           # the same table from (index, value) pairs of the hits only
    tiny = sharded.all_gather_scan_hits(dist, mine, n_scans, capacity_frac=1e-4)           # capacity outgrown on some rank: every rank takes the dense gather
    assert torch.equal(compact, flags) and torch.equal(tiny, flags)
    out.put((rank, (lo, hi), flags.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_scans_shard_over_ranks_and_results_gather_in_scan_order():
    """scan-level kernels (surfel association) shard by scan; the per-point flags are all-gathered — here with the CPU oracle per shard over gloo."""
    assert [sharded.scan_shard(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [sharded.scan_shard(8, r, 8) for r in range(8)] == [(r, r + 1) for r in range(8)]
    assert sharded.scan_shard(2, 3, 4) == (2, 2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_scan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, p4, bmin, bmax = synth.make_assoc_problem(seed=70, H=4, W=300, n_planes=60)
    whole = np.stack([O.surfel_assoc(synth.make_assoc_problem(seed=70 + s, H=4, W=300, n_planes=60)[0], p4, bmin, bmax, 0.05, 2).ravel() for s in range(5)])
    assert (whole >= 0).sum() > 10
    for _, _, flags in got:
        assert np.array_equal(flags, whole)
