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
