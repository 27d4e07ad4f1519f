"""f4 parity: lvx_host::Calibrator (lvi-exc_amd/host/lvx_calibrate.hpp, C++ over the C ABI, GPU) against the oracle running THE SAME SCHEDULE
(oracle/pipeline.py: de-skew -> voxel grid -> surfel map -> association -> trajInitFromSurfel, twice; trajInitFromLVIdata; the camera-surfel stage) on
synth.make_sequence(seed=50) — reference: src/lvi_exc/test/lvi_initialize_surfel_orb.cpp:1169-1245, src/core/trajectory_manager_lvi.cpp:138-257,311-351.

Stage by stage, each stage fed with the GPU's inputs so that one float of the de-skew that rounds the other way cannot hide a wrong rule downstream:
  de-skew          GPU vs oracle from the same state: same NaN pattern, every coordinate within 4e-6 m (float32 outputs of FP64 poses), < 0.1 % of them not bit-equal
  surfel map       oracle voxel grid + setSurfelMap on the GPU's map cloud: same number of surfels, same leaves / point counts / inlier counts / plane types / AABBs
                   (exact), planes 1e-9
  association      oracle getAssociation on the GPU's scans and surfels: the SurfelPoint list identical — order, raw points, map points, timestamps, plane ids
  every solve      oracle LM (oracle/lm.py) from the stage's input state on the same blocks: same accept / reject sequence and termination, cost history 1e-7,
                   extrinsics after the stage within 1e-6 rad / 1e-4 m
  landmark <-> plane   the number of camera-landmark-to-surfel blocks equals the oracle's associateVisualPointsWithPlanes
and once free-running (the oracle chains its own outputs): final extrinsics within 1e-5 rad / 1e-4 m of the Calibrator's.
"""
import os
import subprocess
import time

import numpy as np
import pytest

import lvx
import synth
from oracle import pipeline
from test_gpu_pipeline import _write

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "calibrate_demo.cpp")
LIBDIR = os.path.join(ROOT, "lvi-exc_amd")
REFINE = 2


def _build_demo(d):
    import build as lvx_build
    lvx_build.build()
    exe = str(d / "calibrate_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(LIBDIR, "host"), SRC, "-o", exe, "-L" + LIBDIR, "-llvx", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _run_demo(exe, d, S, n_stages, n_assoc, pose_file=None, **wkw):
    """Run the C++ Calibrator on S (tests/native/calibrate_demo.cpp) and parse its result + history files: stages (with state_in / state_out), association records, final state."""
    pin, pout, phist = str(d / "seq.bin"), str(d / "res.bin"), str(d / "hist.bin")
    _write(pin, S, **wkw)
    t0 = time.perf_counter()
    r = subprocess.run([exe, pin, pout, phist] + ([pose_file] if pose_file else []), capture_output=True, text=True)
    print(r.stdout, r.stderr[-2000:], "calibrator wall time %.2f s" % (time.perf_counter() - t0))
    assert r.returncode == 0, r.stderr
    out = np.fromfile(pout)
    ns, nst = int(out[0]), len(S["state0"])
    rep = out[1:1 + 7 * ns].reshape(ns, 7)
    o = 1 + 7 * ns
    x_final = out[o:o + nst]; o += nst
    stages = []
    for k in range(ns):
        n = int(out[o]); o += 1
        cost, radius, acc = out[o:o + n], out[o + n:o + 2 * n], out[o + 2 * n:o + 3 * n].astype(int); o += 3 * n
        m = int(out[o]); o += 1
        stages.append(dict(iterations=int(rep[k, 0]), termination=lvx.LM_TERMINATION[int(rep[k, 1])], initial_cost=rep[k, 2], final_cost=rep[k, 3], n_planes=int(rep[k, 4]),
                           n_used=int(rep[k, 5]), n_camsurf=int(rep[k, 6]), cost_history=cost, accepted=acc, state_in=out[o:o + m])); o += m
    assert o == len(out) and ns == n_stages
    for k in range(ns):
        stages[k]["state_out"] = stages[k + 1]["state_in"] if k + 1 < ns else x_final
    buf = open(phist, "rb").read()
    assoc, o = [], 0
    while o < len(buf):
        n_state, n_pl, n_pt, n_cl = (int(v) for v in np.frombuffer(buf, np.float64, 4, o)); o += 32
        rec = dict(state=np.frombuffer(buf, np.float64, n_state, o)); o += 8 * n_state
        rec["planes"] = np.frombuffer(buf, lvx.SURFEL_PLANE, n_pl, o); o += lvx.SURFEL_PLANE.itemsize * n_pl
        for key, w in (("pt", 3), ("pt_map", 3), ("t", 1)):
            rec[key] = np.frombuffer(buf, np.float64, w * n_pt, o).reshape(n_pt, w) if w > 1 else np.frombuffer(buf, np.float64, n_pt, o); o += 8 * w * n_pt
        rec["plane"] = np.frombuffer(buf, np.int32, n_pt, o); o += 4 * n_pt
        rec["scans_in_map"] = np.frombuffer(buf, np.float32, n_cl, o).reshape(len(S["scans"]), S["H"], S["W"], 4); o += 4 * n_cl
        assoc.append(rec)
    assert len(assoc) == n_assoc
    return dict(S=S, stages=stages, assoc=assoc, x_final=x_final)


@pytest.fixture(scope="module")
def run(tmp_path_factory):
    d = tmp_path_factory.mktemp("calib_oracle")
    exe = _build_demo(d)
    S = synth.make_sequence(seed=50)
    return _run_demo(exe, d, S, REFINE + 2, REFINE, refine_iterations=REFINE, lvi=1, camsurf=1)


def _planes_dict(pl):
    return dict(p4=np.ascontiguousarray(pl["p4"]), Pi=np.ascontiguousarray(pl["Pi"]), box_min=np.ascontiguousarray(pl["box_min"]), box_max=np.ascontiguousarray(pl["box_max"]))


def _points_dict(rec):
    return dict(pt=rec["pt"], pt_map=rec["pt_map"], t=rec["t"], plane=rec["plane"])


def _ext_err(a, b, N):
    out = {}
    for name, o in (("lidar", 7 * N + 16), ("cam", 7 * N + 24)):
        d = synth.qmul(a[o:o + 4], synth.qconj(b[o:o + 4]))
        out[name] = (float(2 * np.arctan2(np.linalg.norm(d[:3]), abs(d[3]))), float(np.linalg.norm(a[o + 4:o + 7] - b[o + 4:o + 7])))
    return out


@pytest.mark.parametrize("k", range(REFINE))
def test_data_association_round_matches_the_oracle(run, k):
    S, rec = run["S"], run["assoc"][k]
    assert np.array_equal(rec["state"], run["stages"][k]["state_in"])            # the association ran at the state the solve then starts from
    # de-skew
    so = pipeline.deskew_into_map(S, rec["state"])
    sg = rec["scans_in_map"]
    assert np.array_equal(np.isnan(so), np.isnan(sg))
    m = ~np.isnan(so)
    diff = np.abs(so[m] - sg[m])
    frac = np.count_nonzero(so[m].view(np.uint32) != sg[m].view(np.uint32)) / m.sum()
    print("round %d de-skew: max |diff| %.2e m, %.4f %% of the floats not bit-equal" % (k, diff.max(), 100 * frac))
    assert diff.max() <= 4e-6 and frac < 1e-3
    # surfel map on the GPU's map cloud
    po = pipeline.surfel_map(sg, pipeline.DEFAULTS)
    pg = rec["planes"]
    assert len(pg) == len(po["p4"]) == run["stages"][k]["n_planes"] and len(pg) > 100
    assert np.array_equal(pg["leaf"], po["leaf"]) and np.array_equal(pg["n_points"], po["n_points"]) and np.array_equal(pg["n_inliers"], po["n_inliers"])
    assert np.array_equal(pg["plane_type"], po["plane_type"])
    assert np.array_equal(pg["box_min"], po["box_min"]) and np.array_equal(pg["box_max"], po["box_max"])
    assert np.abs(pg["p4"] - po["p4"]).max() <= 1e-9 and np.abs(pg["Pi"] - po["Pi"]).max() <= 1e-9 * np.abs(po["Pi"]).max()
    # association of every scan against the GPU's surfels
    eo = pipeline.associate(S, sg, _planes_dict(pg), pipeline.DEFAULTS)
    assert len(eo["t"]) == len(rec["t"]) > 1000
    for key in ("pt", "pt_map", "t", "plane"):
        assert np.array_equal(eo[key], rec[key]), key
    # what the solve uses of it: every 10th point at or after the map time
    assert run["stages"][k]["n_used"] == len(pipeline.select_surfels(eo, S["t_map"], 10)[1])


def _check_solve(run, k, stage, planes, points, camsurf=None):
    S, st = run["S"], run["stages"][k]
    t0 = time.perf_counter()
    xo, so = pipeline.solve_stage(S, st["state_in"], stage, planes, points, camsurf=camsurf)
    print("stage %d %-26s gpu: it %d %s cost %.6e -> %.6e | oracle: it %d %s -> %.6e (%.1f s)" % (k, stage, st["iterations"], st["termination"], st["initial_cost"], st["final_cost"],
                                                                                                     so["iterations"], so["termination"], so["final_cost"], time.perf_counter() - t0))
    assert abs(st["initial_cost"] - so["initial_cost"]) <= 1e-10 * so["initial_cost"]
    assert list(st["accepted"]) == list(so["accepted"]) and st["termination"] == so["termination"] and st["iterations"] == so["iterations"]
    assert np.abs(st["cost_history"] - so["cost_history"]).max() <= 1e-7 * so["cost_history"].max()
    e = _ext_err(st["state_out"], xo, S["n_knots"])
    print("   extrinsics gpu vs oracle after the stage:", e)
    assert e["lidar"][0] <= 1e-6 and e["cam"][0] <= 1e-6 and e["lidar"][1] <= 1e-4 and e["cam"][1] <= 1e-4
    return xo


@pytest.mark.parametrize("k", range(REFINE))
def test_surfel_solves_match_the_oracle_lm(run, k):
    rec = run["assoc"][k]
    _check_solve(run, k, "TrajFromSurfel", _planes_dict(rec["planes"]), _points_dict(rec))


def test_lvi_and_camera_surfel_stages_match_the_oracle_lm(run):
    S, rec = run["S"], run["assoc"][-1]
    planes, points = _planes_dict(rec["planes"]), _points_dict(rec)
    _check_solve(run, REFINE, "TrajFromLVI", planes, points)
    st = run["stages"][REFINE + 1]
    cs = pipeline.landmark_planes(S, st["state_in"], planes)
    assert len(cs[0]) == st["n_camsurf"]
    print("camera-landmark-to-surfel blocks:", len(cs[0]))
    _check_solve(run, REFINE + 1, "TrajFromLVILandmarksOnly", planes, points, camsurf=cs)


def test_free_running_oracle_schedule_ends_at_the_same_extrinsics(run):
    S = run["S"]
    t0 = time.perf_counter()
    xo, log = pipeline.run_schedule(S, S["state0"], refine_iterations=REFINE, lvi_stage=True, camera_surfel_stage=True)
    print("oracle schedule %.1f s; surfels per round: oracle %s, gpu %s" % (time.perf_counter() - t0, [len(l["association"]["planes"]["p4"]) for l in log[:REFINE]],
                                                                              [len(a["planes"]) for a in run["assoc"]]))
    for lo, st in zip(log, run["stages"]):
        assert lo["lm"]["termination"] == st["termination"]
    e = _ext_err(run["x_final"], xo, S["n_knots"])
    print("free-running, gpu vs oracle:", e)
    assert e["lidar"][0] <= 1e-5 and e["cam"][0] <= 1e-5 and e["lidar"][1] <= 1e-4 and e["cam"][1] <= 1e-4


# ---- the reference's DEFAULT route: first map from LOAM poses (lvi_initialize_surfel_orb.cpp:1175-1178, 1262-1300) ----
def _reference_initial_state(S):
    """What the reference starts from (trajectory_manager_lvi.cpp:31-36, imu.h:126-127): every R3 control point 0, every SO3 control point identity, roll = pitch = 0.01,
    zero biases; the extrinsics' initial guesses and the inverse depths of the ORB map as make_sequence's state0 has them."""
    N = S["n_knots"]
    x = np.array(S["state0"], dtype=np.float64)
    x[:3 * N] = 0.0
    x[3 * N:7 * N] = np.tile([0.0, 0.0, 0.0, 1.0], N)
    return x


@pytest.fixture(scope="module")
def run_loam(tmp_path_factory):
    d = tmp_path_factory.mktemp("calib_loam")
    exe = _build_demo(d)
    S = synth.make_sequence(seed=53, duration=4.0, n_reproj=1500)
    scan_t, stamp, p, q_wxyz, T = synth.sequence_loam_poses(S, noise_m=2e-3, noise_rad=5e-4, seed=7)
    stamp, p, q_wxyz, T = stamp[:-1], p[:-1], q_wxyz[:-1], T[:-1]      # LOAM lost the last scan: it has no pose (loam_poses_map_.find fails) and is skipped
    pose_file = str(d / "loam_poses.txt")
    synth.write_loam_pose_file(pose_file, stamp, p, q_wxyz)
    x0 = _reference_initial_state(S)
    out = _run_demo(exe, d, S, 2, 1, pose_file=pose_file, refine_iterations=1, lvi=0, camsurf=0, solve0=1, state=x0, scan_stamps=scan_t)
    out.update(x0=x0, scan_t=scan_t, T=T, has_pose=np.r_[np.ones(len(T), np.int32), 0], poses=np.vstack([T, np.zeros((1, 16))]))
    return out


def test_solve0_from_the_reference_initial_state_matches_the_oracle_lm(run_loam):
    S, st = run_loam["S"], run_loam["stages"][0]
    assert np.array_equal(st["state_in"], run_loam["x0"])
    xo, so = pipeline.solve_stage(S, st["state_in"], "SO3FromGyro", None, None)
    print("Solve #0 gpu: it %d %s cost %.6e -> %.6e | oracle: it %d %s -> %.6e" % (st["iterations"], st["termination"], st["initial_cost"], st["final_cost"], so["iterations"], so["termination"], so["final_cost"]))
    assert st["termination"] == so["termination"] and st["iterations"] == so["iterations"] and list(st["accepted"]) == list(so["accepted"])
    assert np.abs(st["cost_history"] - so["cost_history"]).max() <= 1e-7 * so["cost_history"].max()
    N = S["n_knots"]
    assert np.abs(st["state_out"][3 * N:7 * N] - xo[3 * N:7 * N]).max() <= 1e-7      # the SO3 control points the first map is de-skewed with


def test_first_map_association_from_loam_poses_matches_the_oracle(run_loam):
    S, rec = run_loam["S"], run_loam["assoc"][0]
    assert np.array_equal(rec["state"], run_loam["stages"][1]["state_in"])            # it ran at Solve #0's result: zero positions, SO3 from the gyroscope
    assert np.count_nonzero(rec["state"][:3 * S["n_knots"]]) == 0
    fo = pipeline.first_data_association(S, rec["state"], run_loam["scan_t"], run_loam["poses"], run_loam["has_pose"])
    so, sg = fo["scans_in_map"], rec["scans_in_map"]
    assert np.isnan(sg[-1, :, :, :3]).all() and not np.isnan(sg[0, :, :, 0]).all()    # the scan without a pose is absent
    assert np.array_equal(np.isnan(so), np.isnan(sg))
    m = ~np.isnan(so)
    diff = np.abs(so[m] - sg[m])
    frac = np.count_nonzero(so[m].view(np.uint32) != sg[m].view(np.uint32)) / m.sum()
    print("first map: %d key scans of %d; scans in the map frame: max |diff| %.2e m, %.4f %% of the floats not bit-equal" % (len(fo["key"]), len(sg), diff.max(), 100 * frac))
    assert diff.max() <= 4e-6 and frac < 1e-3
    assert 1 < len(fo["key"]) < len(sg) - 1                                           # the key-scan rule selects: not every scan, not only the first
    # the surfel map of the KEY-SCAN cloud, from the GPU's scans (lambda 0.6: the constructor's)
    opt = dict(pipeline.DEFAULTS, plane_lambda=0.6)
    po = pipeline.surfel_map(sg[fo["key"]], opt)
    pg = rec["planes"]
    assert len(pg) == len(po["p4"]) == run_loam["stages"][1]["n_planes"] and len(pg) > 50
    assert np.array_equal(pg["leaf"], po["leaf"]) and np.array_equal(pg["n_points"], po["n_points"]) and np.array_equal(pg["n_inliers"], po["n_inliers"])
    assert np.array_equal(pg["plane_type"], po["plane_type"])
    assert np.array_equal(pg["box_min"], po["box_min"]) and np.array_equal(pg["box_max"], po["box_max"])
    assert np.abs(pg["p4"] - po["p4"]).max() <= 1e-9 and np.abs(pg["Pi"] - po["Pi"]).max() <= 1e-9 * np.abs(po["Pi"]).max()
    # a map of ALL scans would be another map: the key-scan selection is really what the GPU used
    assert len(pipeline.surfel_map(sg[:-1], opt)["p4"]) != len(pg)
    # association of every scan against the GPU's surfels: the SurfelPoint list identical
    eo = pipeline.associate(S, sg, _planes_dict(pg), opt)
    assert len(eo["t"]) == len(rec["t"]) > 500
    for key in ("pt", "pt_map", "t", "plane"):
        assert np.array_equal(eo[key], rec[key]), key
    assert run_loam["stages"][1]["n_used"] == len(pipeline.select_surfels(eo, S["t_map"], 10)[1])


def test_batch_optimization_after_the_first_map_matches_the_oracle_lm(run_loam):
    """trajInitFromSurfel from ZERO positions on the first map's SurfelPoints: the GPU's LM against the oracle's from the same state on the same blocks."""
    rec = run_loam["assoc"][0]
    xo = _check_solve(run_loam, 1, "TrajFromSurfel", _planes_dict(rec["planes"]), _points_dict(rec))
    S, N = run_loam["S"], run_loam["S"]["n_knots"]
    e = _ext_err(xo, S["state_true"], N)
    print("after the batch optimisation, oracle vs ground truth (LiDAR):", e["lidar"], " start:", _ext_err(run_loam["x0"], S["state_true"], N)["lidar"])
    assert np.abs(xo[:3 * N]).max() > 0.1                                             # the positions left zero


# ---- LIinitializer::CIoptimize (lvi_initialize_surfel_orb.cpp:519-537): camera-IMU calibration alone ----
def test_camera_imu_route_matches_the_oracle_lm(tmp_path_factory):
    """initialSO3TrajWithGyro, then trajInitFromVisualFrames (trajectory_manager_lvi.cpp:99-136: gyroscope + accelerometer + reprojection blocks, LiDAR extrinsics locked,
    <= 200 iterations) through lvx_host::Calibrator::RunCameraImu, each solve against the oracle LM from the same state: accept / reject sequence, termination, cost
    history, camera extrinsics."""
    d = tmp_path_factory.mktemp("calib_ci")
    exe = _build_demo(d)
    S = synth.make_sequence(seed=53, duration=4.0, n_reproj=1500)
    N = S["n_knots"]
    x0 = np.array(S["state0"], dtype=np.float64)
    x0[3 * N:7 * N] = np.tile([0.0, 0.0, 0.0, 1.0], N)      # Solve #0 starts from identity rotations, as the reference does (from rotations already near the truth the gyro-only
    #                                                          problem is a 1e11-conditioned plateau on which two correct LMs part ways within a few iterations)
    out = _run_demo(exe, d, S, 2, 0, refine_iterations=-1, lvi=0, camsurf=0, solve0=1, state=x0)
    for k, stage in ((0, "SO3FromGyro"), (1, "TrajFromVisualFrames")):
        st = out["stages"][k]
        xo, so = pipeline.solve_stage(S, st["state_in"], stage, None, None)
        print("%-22s gpu: it %d %s cost %.6e -> %.6e | oracle: it %d %s -> %.6e" % (stage, st["iterations"], st["termination"], st["initial_cost"], st["final_cost"], so["iterations"], so["termination"], so["final_cost"]))
        assert abs(st["initial_cost"] - so["initial_cost"]) <= 1e-10 * so["initial_cost"]
        assert list(st["accepted"]) == list(so["accepted"]) and st["termination"] == so["termination"] and st["iterations"] == so["iterations"]
        assert np.abs(st["cost_history"] - so["cost_history"]).max() <= 1e-7 * so["cost_history"].max()
        e = _ext_err(st["state_out"], xo, N)
        assert e["cam"][0] <= 1e-6 and e["cam"][1] <= 1e-4 and e["lidar"] == (0.0, 0.0)      # the LiDAR extrinsics are locked: untouched on both sides
    e0, e1 = _ext_err(x0, S["state_true"], N), _ext_err(out["x_final"], S["state_true"], N)
    print("camera extrinsics vs ground truth: start", e0["cam"], "end", e1["cam"])
    assert e1["cam"][0] < 0.2 * e0["cam"][0]


# ---- lvi.yaml's opt_time_offset = true: the sensor time offsets are parameter blocks of the stages (trajectory_manager_lvi.cpp:159-165, 327-335) ----
@pytest.mark.parametrize("stage", ["TrajFromSurfel", "TrajFromLVI"])
def test_stages_with_free_time_offsets_match_the_oracle_lm(run, stage):
    """The same surfels and SurfelPoints as the first association round of the schedule, the stage solved with `opt_time_offset`: the LiDAR (and, with the
    reprojection blocks, the camera) time offset is free, bounded by +- 1 ms (sensors.h:161-162) — the fused kernels' time-offset column, 5-control-point
    segments, the projected line search of the constrained problem — against the oracle LM from the same state."""
    S, rec = run["S"], run["assoc"][0]
    opt = dict(pipeline.DEFAULTS, opt_time_offset=True)
    planes, points = _planes_dict(rec["planes"]), _points_dict(rec)
    x0 = np.array(run["stages"][0]["state_in"])
    N, L = S["n_knots"], len(S["lm_t0"])
    locks = pipeline.stage_locks(stage, True)
    g = lvx.Context(0)
    g.set_spline(S["t0"], S["dt"], N)
    c = S["camera"]
    g.set_camera(c["rows"], c["cols"], c["readout"], c["fx"], c["fy"], c["cx"], c["cy"], c["k1"], c["k2"], c["p1"], c["p2"], c["k3"])
    g.set_landmarks(S["lm_uv"], S["lm_t0"])
    g.set_imu(S["t_imu"], S["gyro"], S["acc"], opt["w_gyro"], opt["w_acc"])
    g.set_planes(planes["Pi"])
    pt, t, pid = pipeline.select_surfels(points, S["t_map"], opt["downsample_step"])
    g.set_surfel(pt, t, pid, S["t_map"], 5.0, opt["w_surfel"])
    if stage != "TrajFromSurfel":
        g.set_reproj(S["rep_lm"], S["rep_uv"], S["rep_t0"], opt["w_cam"], 1.0)
    g.set_locks(locks)
    iters = 30 if stage == "TrajFromSurfel" else 80
    xg, sg = g.lm_solve(x0, max_iterations=iters)
    assert g.layout()["exact_fallback"] == 0
    g.close()
    xo, so = pipeline.solve_stage(S, x0, stage, planes, points, opt)
    print("%s, free offsets: gpu it %d %s cost %.6e -> %.6e | oracle it %d %s -> %.6e; tau_L %.3e tau_C %.3e" % (stage, sg["iterations"], sg["termination"], sg["initial_cost"], sg["final_cost"],
                                                                                                                 so["iterations"], so["termination"], so["final_cost"], xg[7 * N + 23], xg[7 * N + 31]))
    assert list(sg["accepted"]) == list(so["accepted"]) and sg["termination"] == so["termination"] and sg["iterations"] == so["iterations"]
    assert np.abs(np.asarray(sg["cost_history"]) - so["cost_history"]).max() <= 1e-7 * so["cost_history"].max()
    e = _ext_err(xg, xo, N)
    assert e["lidar"][0] <= 1e-6 and e["cam"][0] <= 1e-6 and e["lidar"][1] <= 1e-4 and e["cam"][1] <= 1e-4
    assert abs(xg[7 * N + 23] - xo[7 * N + 23]) <= 1e-9 and abs(xg[7 * N + 31] - xo[7 * N + 31]) <= 1e-9
    assert abs(xg[7 * N + 23]) <= 1e-3 + 1e-15 and abs(xg[7 * N + 31]) <= 1e-3 + 1e-15       # inside the box
    assert xg[7 * N + 23] != 0.0                                                            # the LiDAR offset moved
