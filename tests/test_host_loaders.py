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
    # PoseOfScan: simulation = exact stamp (int64(t * 1e9)); otherwise the nearest of poses idx - 5 .. idx + 4, never the file's last pose
    assoc = {(int(m), int(i)): (int(ok), float(x), float(y), float(r00), float(r01)) for m, i, ok, x, y, r00, r01 in (ln.split()[1:] for ln in out if ln.startswith("A "))}
    n = len(poses)

    def R01(q):
        w, x, y, z = q
        return 1 - 2 * (y * y + z * z), 2 * (x * y - z * w)
    for i in range(n):
        t = poses[i][0] * 1e-9 + (0.004 if i == 3 else 0.0) + (0.06 if i == 7 else 0.0)
        hit = [k for k in range(n) if poses[k][0] == int(t * 1e9)]
        exp = hit[-1] if hit else None
        ok, x, y, r00, r01 = assoc[(0, i)]
        assert ok == (exp is not None)
        if exp is not None:
            assert np.allclose([x, y], [float("%.7g" % poses[exp][1][0]), float("%.7g" % poses[exp][1][1])], rtol=0, atol=1e-12)
        cand = [k for k in range(i - 5, i + 5) if 0 <= k < n - 1]
        ok, x, y, r00, r01 = assoc[(1, i)]
        assert ok == (1 if cand else 0)
        if cand:
            best = min(cand, key=lambda k: (abs(poses[k][0] * 1e-9 - t), k))
            q = [float("%.7g" % v) for v in poses[best][2]]
            assert np.allclose([x, y], [float("%.7g" % poses[best][1][0]), float("%.7g" % poses[best][1][1])], rtol=0, atol=1e-12)
            assert np.allclose([r00, r01], R01(q), rtol=0, atol=1e-12)
    assert assoc[(0, 3)][0] == 0 and assoc[(0, 7)][0] == 0 and assoc[(1, 3)][0] == 1      # a late stamp has no exact pose but a nearest one
    assert assoc[(1, n - 1)][1] == float("%.7g" % poses[n - 2][1][0])                      # the last pose of the file is never taken: scan n - 1 gets pose n - 2
    tau = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU
    assert out[-1] == "locks %d %d %d %d" % (lvx.LOCK_R3 | lvx.LOCK_ACC_BIAS | lvx.LOCK_GYRO_BIAS | tau, lvx.LOCK_CAM_Q | lvx.LOCK_CAM_P | lvx.LOCK_LANDMARKS | tau, tau,
                                              lvx.LOCK_TRAJ | lvx.LOCK_LIDAR_Q | lvx.LOCK_LIDAR_P)
