"""bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), on ONE GPU with the gloo backend (LVX_BENCH_BACKEND=gloo: two
ranks time-share the device; RCCL needs a device per rank): guards the N > 1 path of the bench — sequence-per-rank evaluation + all-reduce of the border block,
the joint LM iteration over the ranks (lvx_lm_solve_shared, host-callback transport) and the scan-sharded surfel association with its all-gather — against rot
while no multi-GPU node is available.  1/10-size problem, 3 steps."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_rank_bench_line():
    env = dict(os.environ, LVX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--small"]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    print("2-rank bench: %.1f s wall, %.1f Mevals/s, ms_per_step %.3f" % (dt, out["value"], out["ms_per_step"]))
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["unit"] == "Mevals/s"
    assert out["value"] > 0 and out["config"]["parallelism"] == "sequence-per-gpu x2"
    sec = out["secondary"]
    j = sec["joint_lm_iteration"]
    assert "error" not in j, j
    assert j["iterations"] >= 1 and j["final_cost"] < j["initial_cost"]
    a = sec["surfel_assoc"]
    assert "error" not in a, a
    assert a["scans"] == 2 * a["scans_per_gpu"] and a["associated_points"] > 0 and a["Mpts_per_s"] > 0
    assert dt < 300


def test_eight_rank_bench_line_with_a_failing_rank():
    """What the driver's 8-GPU run does, on ONE GPU (gloo, 1/10-size sequences, seeds 40..47): eight ranks evaluate + all-reduce, solve the joint LM, all-gather the
    sharded association, time the other shard size of SURVEY's config 5 — and when rank 5 fails locally (non-unit quaternion) every rank leaves the joint solve in the same
    collective: the failing rank with its own code, the others with LVX_E_COMM."""
    env = dict(os.environ, LVX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", LVX_BENCH_FAIL_RANK="5", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--small"]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    sec = out["secondary"]
    print("8-rank bench: %.1f s wall, %.1f Mevals/s, joint LM %s, vote %s, other shard size %s" % (dt, out["value"], sec.get("joint_lm_iteration"), sec.get("joint_failure_vote"), sec.get("other_shard_size")))
    assert out["n_gpus"] == 8 and out["config"]["parallelism"] == "sequence-per-gpu x8" and out["config"]["shard_size"] == "full" and out["value"] > 0
    j = sec["joint_lm_iteration"]
    assert "error" not in j, j
    assert j["iterations"] >= 1 and j["final_cost"] < j["initial_cost"] and j["transport"].startswith("host callback")
    v = sec["joint_failure_vote"]
    assert "error" not in v, v
    codes = v["return_codes"]
    assert len(codes) == 8 and codes[5] == -2 and all(c == -5 for i, c in enumerate(codes) if i != 5)       # LVX_E_NONUNIT_QUAT on the failing rank, LVX_E_COMM everywhere else
    a = sec["surfel_assoc"]
    assert "error" not in a, a
    assert a["scans"] == 8 * a["scans_per_gpu"] and a["associated_points"] > 0
    o = sec["other_shard_size"]
    assert "error" not in o, o
    assert o["shard_size"] == "eighth" and o["value"] > 0
    assert dt < 600


def test_single_rank_runs_the_nccl_branch_with_the_library_transport():
    """Every line the driver's 8-GPU run executes, executed once on a one-GPU box: bench.py under torch.distributed.run with ONE rank on the real `nccl` (= RCCL) backend and
    LVX_BENCH_FORCE_DIST=1 — process group, exported border block, torch all-reduce, then the unique-id broadcast, lvx_rccl_init, the SAME step over the library's own
    communicator (lvx_rccl_allreduce_d, which becomes the headline transport), the joint LM over in-library RCCL, the association all-gather path and lvx_rccl_finalize."""
    env = dict(os.environ, LVX_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.pop("LVX_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--small"]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    sec = out["secondary"]
    print("1-rank nccl bench: %.1f s wall, %s, transports %s, joint LM %s" % (dt, out["config"]["allreduce_transport"][:40], sec.get("headline_transports"), sec.get("joint_lm_iteration")))
    assert "inlib_rccl_init_error" not in sec and "inlib_rccl_step_error" not in sec, sec
    assert out["config"]["allreduce_transport"].startswith("RCCL inside liblvx")
    h = sec["headline_transports"]
    assert h["single_rank_identity"] is True and h["inlib_transport"]["value"] > 0 and h["torch_transport"]["value"] > 0
    assert out["value"] == h["inlib_transport"]["value"] and out["n_gpus"] == 1
    j = sec["joint_lm_iteration"]
    assert "error" not in j, j
    assert j["transport"].startswith("RCCL inside liblvx") and j["iterations"] >= 1 and j["final_cost"] < j["initial_cost"] and j["collectives"] >= 3
    assert "error" not in sec["surfel_assoc"] and "watchdog" not in sec
    assert "roofline" in out and out["roofline"]["frac"] > 0


def test_every_visible_device_runs_the_rccl_bench():
    """The day a multi-GPU box runs this suite, it proves that RCCL saw N ranks BEFORE the driver's scaling run: bench.py --gpus N (N = min(visible devices, 8)) on the real
    `nccl` backend, one rank per device — the step over torch's all-reduce and over the library's own communicator (lvx_rccl_init across N ranks), the joint LM over in-library
    RCCL with its collectives counted, and every rank ending with bit-identical shared extrinsics.  One visible device: skipped (test_single_rank_... covers the code path)."""
    import torch
    n = min(torch.cuda.device_count(), 8)
    if n < 2:
        pytest.skip("one visible device: the N-rank RCCL run needs a device per rank")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.pop("LVX_BENCH_BACKEND", None); env.pop("LVX_BENCH_FORCE_DIST", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "2", "--small"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    sec = out["secondary"]
    print("%d-rank RCCL bench: %.1f Mevals/s, transports %s, joint LM %s" % (n, out["value"], sec.get("headline_transports"), sec.get("joint_lm_iteration")))
    assert out["n_gpus"] == n and out["config"]["parallelism"] == "sequence-per-gpu x%d" % n and out["scaling"] == "weak"
    assert "inlib_rccl_init_error" not in sec and "inlib_rccl_step_error" not in sec, sec
    h = sec["headline_transports"]
    assert h["inlib_transport"]["value"] > 0 and h["torch_transport"]["value"] > 0 and out["value"] == h["inlib_transport"]["value"]
    j = sec["joint_lm_iteration"]
    assert "error" not in j, j
    assert j["transport"].startswith("RCCL inside liblvx") and j["ranks"] == n and j["collectives"] >= 3 and j["final_cost"] < j["initial_cost"]
    assert j["shared_extrinsics_max_spread_over_ranks"] == 0.0
    a = sec["surfel_assoc"]
    assert "error" not in a, a
    assert a["scans"] == n * a["scans_per_gpu"] and a["associated_points"] > 0
