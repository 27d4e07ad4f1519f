"""Text-format loaders of the reference's offline driver (lvi-exc_amd/host/lvx_loaders.hpp): files written the way
write_orb_slam_results.cpp / laserMapping.cpp write them are parsed by the C++ header and compared with an independent reading here."""
import os
import subprocess

import numpy as np
import pytest

import lvx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def demo(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ld") / "host_loader_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "native", "host_loader_demo.cpp"), "-o", out])
    return out


def test_orb_results_and_pose_files(demo, tmp_path):
    rng = np.random.default_rng(3)
    stamps = [1_600_000_000_000_000_000 + 50_000_000 * k for k in range(12)]
    cols, rows, border = 1280, 720, 20
    lm_ids = list(range(100, 140))
    seen = {s: {} for s in stamps}
    ref_of = {}
    for l in lm_ids:
        views = sorted(rng.choice(len(stamps), size=rng.integers(2, 6), replace=False))
        ref_of[l] = stamps[views[0]]
        for v in views:
            seen[stamps[v]][l] = (float("%g" % rng.uniform(0, cols)), float("%g" % rng.uniform(0, rows)))   # ostream default precision, as the writer
    lines = []
    for s in stamps:
        lines.append("FramePose %d 0.1 0.2 0.3 0 0 0 1" % s)
        lines.append("UV %d " % s + " ".join("%g %g %d" % (uv[0], uv[1], l) for l, uv in seen[s].items()) + " ")
    depth = {l: float("%g" % rng.uniform(1, 20)) for l in lm_ids}
    for l in lm_ids:
        lines.append("MapPoint %d 0.5 -0.25 %g %d" % (l, depth[l], ref_of[l]))
    lines.append("MapPoint 100 9 9 9 %d" % ref_of[100])              # duplicate id: ignored
    lines.append("MapPoint 999 1 1 1 %d" % stamps[0])                 # never observed in its reference frame: dropped
    lines.append("MapPoint 998 1 1 1 12345")                          # unknown reference frame: dropped
    orb = tmp_path / "orb.txt"; orb.write_text("\n".join(lines) + "\n")
    poses = []
    t, yaw = np.zeros(3), 0.0
    for k in range(30):
        t = t + (np.array([0.03, 0, 0]) if k % 3 else np.array([0.2, 0.05, 0]))
        yaw += 0.01 if k % 5 else 0.12
        poses.append((1_600_000_000_000_000_000 + 100_000_000 * k, t.copy(), np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])))
    pf = tmp_path / "poses.txt"
    pf.write_text("".join("%d %.7g %.7g %.7g %.7g %.7g %.7g %.7g\n" % (s, p[0], p[1], p[2], q[0], q[1], q[2], q[3]) for s, p, q in poses))
    r = subprocess.run([demo, str(orb), str(pf), str(cols), str(rows), str(border)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    # expected, read independently
    exp_L, exp_O = [], []
    for l in sorted(lm_ids):
        uv = seen[ref_of[l]][l]
        if uv[0] < border or uv[1] < border or uv[0] > cols - border or uv[1] > rows - border:
            continue
        li = len(exp_L)
        exp_L.append((l, uv[0], uv[1], ref_of[l] * 1e-9, 1.0 / (depth[l] + 1e-15)))
        for s in sorted(stamps):
            if s != ref_of[l] and l in seen[s]:
                exp_O.append((li, seen[s][l][0], seen[s][l][1], s * 1e-9))
    assert out[0] == "frames %d views %d landmarks %d observations %d" % (len(stamps), len(stamps), len(exp_L), len(exp_O))
    got_L = [tuple(float(x) for x in ln.split()[1:]) for ln in out if ln.startswith("L ")]
    got_O = [tuple(float(x) for x in ln.split()[1:]) for ln in out if ln.startswith("O ")]
    assert np.allclose(got_L, exp_L, rtol=0, atol=1e-12) and np.allclose(got_O, exp_O, rtol=0, atol=1e-12)
    assert 0 < len(exp_L) < len(lm_ids)                               # the border filter removed some
    # key poses: kept when rotated >= 5 deg or moved >= 0.1 m since the last kept pose
    key = [poses[0]]
    for s, p, q in poses[1:]:
        ls, lp, lq = key[-1]
        d = abs(float(np.dot(lq, q)))
        ang = 2 * np.arctan2(np.sqrt(max(0.0, 1 - min(d, 1.0) ** 2)), d)
        if np.degrees(ang) < 5.0 and np.linalg.norm(lp - p) < 0.1:
            continue
        key.append((s, p, q))
    assert "poses %d key %d" % (len(poses), len(key)) in out
    assert [int(ln.split()[1]) for ln in out if ln.startswith("K ")] == [s for s, _, _ in key]
    tau = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU
    assert out[-1] == "locks %d %d %d %d" % (lvx.LOCK_R3 | lvx.LOCK_ACC_BIAS | lvx.LOCK_GYRO_BIAS | tau, lvx.LOCK_CAM_Q | lvx.LOCK_CAM_P | lvx.LOCK_LANDMARKS | tau, tau,
                                              lvx.LOCK_TRAJ | lvx.LOCK_LIDAR_Q | lvx.LOCK_LIDAR_P)
