"""The C++ host mirror of the reference's estimator surface (lvi-exc_amd/host/lvx_estimator.hpp: AddMeasurement / Lock / Solve over the C ABI).
CPU: the header-only mirror compiles with g++ and links against liblvx.so.  GPU: a C++ program that builds the problem measurement by
measurement the way TrajectoryManagerLVI does must give the same LM result as the ctypes path."""
import os
import subprocess

import numpy as np
import pytest

import lvx
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "host_estimator_demo.cpp")
LIBDIR = os.path.join(ROOT, "lvi-exc_amd")


@pytest.fixture(scope="module")
def demo_binary(tmp_path_factory):
    import build as lvx_build   # lvi-exc_amd/build.py
    lvx_build.build()
    out = str(tmp_path_factory.mktemp("host") / "host_estimator_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", SRC, "-o", out, "-L" + LIBDIR, "-llvx", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return out


def _write_problem(path, P, locks, max_it):
    c = P["camera"]
    parts = [np.array([P["t0"], P["dt"], P["n_knots"], locks, max_it, c["rows"], c["cols"], c["readout"], c["fx"], c["fy"], c["cx"], c["cy"], c["k1"], c["k2"], c["p1"], c["p2"], c["k3"],
                       P["w_gyro"], P["w_acc"], P["t_map"], P["huber_surf"], P["w_surf"], P["huber_rep"], P["w_rep"]], dtype=np.float64)]
    for k in ("state0", "t_imu", "gyro", "acc", "planes", "surf_pt", "surf_t", "surf_plane", "lm_uv", "lm_t0", "rep_lm", "rep_uv", "rep_t0"):
        a = np.asarray(P[k], dtype=np.float64).ravel()
        parts += [np.array([len(a)], dtype=np.float64), a]
    np.concatenate(parts).tofile(path)


def test_host_mirror_compiles_and_links(demo_binary):
    assert os.path.exists(demo_binary)
    # without arguments it only prints its usage: nothing touches a device
    assert subprocess.run([demo_binary], capture_output=True).returncode == 2


@pytest.mark.gpu
def test_cpp_estimator_matches_ctypes_path(demo_binary, tmp_path):
    locks = lvx.LOCK_LIDAR_TAU | lvx.LOCK_CAM_TAU
    P = synth.make_problem(seed=21, duration=1.5, n_surfel=400, n_planes=10, n_landmarks=20, n_camsurf=0)
    pin, pout = str(tmp_path / "p.bin"), str(tmp_path / "r.bin")
    _write_problem(pin, P, locks, 10)
    r = subprocess.run([demo_binary, pin, pout], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "lvx LM: iterations" in r.stdout
    out = np.fromfile(pout)
    g = lvx.Context(0)
    lvx.load_problem(g, P, locks)
    x, s = g.lm_solve(P["state0"], max_iterations=10)
    g.close()
    assert int(out[0]) == s["iterations"] and int(out[4]) == s["successful_steps"]
    assert abs(out[2] - s["initial_cost"]) <= 1e-12 * s["initial_cost"] and abs(out[3] - s["final_cost"]) <= 1e-8 * s["final_cost"]
    assert np.abs(out[5:] - x).max() <= 1e-7      # two runs of the same LM differ at the 1e-9 level after 10 iterations: the order of the FP64 atomics is not fixed
    # a measurement outside the spline surfaces as std::range_error, like kontiki's CheckTimeSpans
    Q = dict(P); Q["t_imu"] = P["t_imu"].copy(); Q["t_imu"][0] = P["t0"] - 1.0
    _write_problem(pin, Q, locks, 2)
    r = subprocess.run([demo_binary, pin, pout], capture_output=True, text=True)
    assert r.returncode == 4 and "range_error" in r.stderr
