"""The reference's offline calibration schedule chained from the oracle's pieces (TEST INFRASTRUCTURE ONLY — PARITY UNPINNED, see orc_core.hpp).

What it restates (stage machine of LIinitializer, src/lvi_exc/test/lvi_initialize_surfel_orb.cpp):
  DataAssociation, refinement branch (:1180-1201)
      ScanUndistortion::undistortScanInMap   (include/core/scan_undistortion.h:59-74, 132-180)   -> oracle.undistort per scan, map cloud = scans concatenated
      LiDAROdometry::ndtInit + setInputTarget(map cloud)                                        -> oracle.voxel_build
      SurfelAssociation::setSurfelMap        (src/core/surfel_association.cpp:50-108)           -> oracle.surfel_extract (deterministic plane fit)
      getAssociation per scan                (:111-158)                                         -> oracle.surfel_assoc + oracle.surfel_emit
      averageTimeDownSmaple(10)              (:240-244)                                         -> every 10th SurfelPoint
  DataAssociation, FIRST-MAP branch (InitializationDone, :1175-1178) -> first_data_association below:
      Mapping() (:1262-1300): ScanUndistortion::undistortScan (scan_undistortion.h:40-57, rotation only) + LiDAROdometry::feedScan(..., using_loam = true) with
      updateKeyScan / checkKeyScan (src/core/lidar_odometry.cpp:45-74, 89-128), undistortScanInMap(odom_data_map) (scan_undistortion.h:95-116),
      setSurfelMap over the voxel grid of the KEY-SCAN map, getAssociation per scan
  BatchOptimization / Refinement (:1212-1243) -> trajInitFromSurfel (src/core/trajectory_manager_lvi.cpp:311-351)
  trajInitFromLVIdata(frames, surfels)        (:138-195), trajInitFromLVIdata(frames, surfels, lm_splane) (:197-257) with associateVisualPointsWithPlanes (:161-214)
every solve through oracle/lm.py (numpy LM on the oracle's dense J^T J).  The product-side mirror is lvx_host::Calibrator (lvi-exc_amd/host/lvx_calibrate.hpp);
tests/test_gpu_pipeline_oracle.py compares the two stage by stage.  Sizes: a few hundred knots (dense linear algebra).
"""
import numpy as np

from . import lm
from . import oracle as O

DEFAULTS = dict(ndt_resolution=0.5, plane_lambda=0.7, fit_threshold=0.05, min_leaf_points=10, min_inliers=20, associated_radius=0.05, selected_per_ring=2, downsample_step=10,
                w_gyro=28.0, w_acc=18.0, w_surfel=10.0, w_cam=5.0, w_cam_surfel=30.0, opt_time_offset=False)

LOCK = dict(TRAJ=1 << 0, R3=1 << 1, LIDAR_Q=1 << 2, LIDAR_P=1 << 3, LIDAR_TAU=1 << 4, CAM_Q=1 << 5, CAM_P=1 << 6, CAM_TAU=1 << 7, ACC_BIAS=1 << 8, GYRO_BIAS=1 << 9, LANDMARKS=1 << 10)


def stage_locks(stage, opt_time_offset=False):
    """Lock masks of the solve stages (which Lock* calls TrajectoryManagerLVI makes before building each estimator)."""
    tau = 0 if opt_time_offset else (LOCK["LIDAR_TAU"] | LOCK["CAM_TAU"])
    if stage == "SO3FromGyro":              # trajectory_manager_lvi.cpp:43-62
        return LOCK["R3"] | LOCK["ACC_BIAS"] | LOCK["GYRO_BIAS"] | LOCK["LIDAR_TAU"] | LOCK["CAM_TAU"]
    if stage == "TrajFromSurfel":           # :311-351
        return LOCK["CAM_Q"] | LOCK["CAM_P"] | LOCK["CAM_TAU"] | LOCK["LANDMARKS"] | (0 if opt_time_offset else LOCK["LIDAR_TAU"])
    if stage == "TrajFromLVI":              # :138-195
        return tau
    if stage == "TrajFromLVILandmarksOnly":  # :197-257 with lock_traj_lidar_in_3rd_stage
        return tau | LOCK["TRAJ"] | LOCK["LIDAR_Q"] | LOCK["LIDAR_P"]
    if stage == "TrajFromVisualFrames":     # :99-136 (LIinitializer::CIoptimize): camera-IMU only, LiDAR extrinsics locked
        return LOCK["LIDAR_Q"] | LOCK["LIDAR_P"] | LOCK["LIDAR_TAU"] | (0 if opt_time_offset else LOCK["CAM_TAU"])
    raise ValueError(stage)


def _raw_scans(S):
    """make_sequence scans -> [n_scans, H * W] PointXYZIT records"""
    sc = S["scans"]
    raw = np.zeros(sc.shape, dtype=O.POINT_XYZIT)
    for k in ("x", "y", "z", "timestamp"):
        raw[k] = sc[k]
    if "intensity" in sc.dtype.names:
        raw["intensity"] = sc["intensity"]
    return raw


def _base_oracle(S):
    o = O.Oracle()
    o.set_spline(S["t0"], S["dt"], S["n_knots"])
    c = S["camera"]
    o.set_camera(c["rows"], c["cols"], c["readout"], c["fx"], c["fy"], c["cx"], c["cy"], c["k1"], c["k2"], c["p1"], c["p2"], c["k3"])
    o.set_landmarks(S["lm_uv"], S["lm_t0"])
    return o


def deskew_into_map(S, state):
    """undistortScanInMap: [n_scans, H, W, 4] float32 in the LiDAR frame at the map time."""
    o = _base_oracle(S)
    q, p, ok = O.eval_lidar_pose(o, state, [S["t_map"]])
    if not ok[0]:
        raise IndexError("map time outside the trajectory")
    q_G_to_L0 = np.array([-q[0, 0], -q[0, 1], -q[0, 2], q[0, 3]])
    raw = _raw_scans(S)
    out = np.stack([O.undistort(o, state, raw[s], q_G_to_L0, p[0], True) for s in range(len(raw))])
    return out.reshape(len(raw), S["H"], S["W"], 4)


def surfel_map(scans_in_map, opt):
    """ndtInit(resolution) + setInputTarget(map cloud) + setSurfelMap: planes dict of oracle.surfel_extract."""
    cloud = np.ascontiguousarray(scans_in_map, np.float32).reshape(-1, 4)      # map_cloud_ += scan, in scan order
    vox = O.voxel_build(cloud, opt["ndt_resolution"], 6, 0.01)
    return O.surfel_extract(cloud, vox, opt["plane_lambda"], opt["fit_threshold"], opt["min_leaf_points"], opt["min_inliers"])


def associate(S, scans_in_map, planes, opt):
    """getAssociation for every scan: the concatenated chronological SurfelPoint list."""
    raw = _raw_scans(S).reshape(len(scans_in_map), S["H"], S["W"])
    parts = []
    for s in range(len(scans_in_map)):
        flag = O.surfel_assoc(scans_in_map[s], planes["p4"], planes["box_min"], planes["box_max"], opt["associated_radius"], opt["selected_per_ring"])
        parts.append(O.surfel_emit(flag, scans_in_map[s], raw[s]))
    return {k: np.concatenate([p[k] for p in parts]) for k in ("pt", "pt_map", "t", "plane")}


def data_association(S, state, opt=None):
    opt = dict(DEFAULTS, **(opt or {}))
    sim = deskew_into_map(S, state)
    planes = surfel_map(sim, opt)
    pts = associate(S, sim, planes, opt) if len(planes["p4"]) else dict(pt=np.zeros((0, 3)), pt_map=np.zeros((0, 3)), t=np.zeros(0), plane=np.zeros(0, np.int32))
    return dict(scans_in_map=sim, planes=planes, points=pts)


def r2ypr_deg(R):
    """mathutils::R2ypr (include/utils/math_utils.h:192-207): yaw, pitch, roll in degrees."""
    n, o, a = R[:, 0], R[:, 1], R[:, 2]
    y = np.arctan2(n[1], n[0])
    p = np.arctan2(-n[2], n[0] * np.cos(y) + n[1] * np.sin(y))
    r = np.arctan2(a[0] * np.sin(y) - a[1] * np.cos(y), -o[0] * np.sin(y) + o[1] * np.cos(y))
    return np.array([y, p, r]) / np.pi * 180.0


def key_scans(poses, present, key_dist=0.2, key_angle_deg=5.0):
    """LiDAROdometry::checkKeyScan over the scans that were fed, in order (src/core/lidar_odometry.cpp:107-128): the first one, or > key_dist from the LAST KEY scan, or
    turned by > key_angle_deg in yaw / pitch / roll (differences wrapped once by +-360: normalize_angle, include/core/lidar_odometry.h:95-102)."""
    pos_last, ypr_last, key = np.zeros(3), np.zeros(3), []
    for s in range(len(poses)):
        if not present[s]:
            continue
        T = np.asarray(poses[s], dtype=np.float64).reshape(4, 4)
        dist = np.linalg.norm(T[:3, 3] - pos_last)
        ypr = r2ypr_deg(T[:3, :3])
        d = ypr - ypr_last
        d = np.where(d > 180, d - 360, d)
        d = np.where(d < -180, d + 360, d)
        if not key or dist > key_dist or (np.abs(d) > key_angle_deg).any():
            pos_last, ypr_last = T[:3, 3].copy(), ypr
            key.append(s)
    return key


def transform_cloud(xyzi, T):
    """pcl::transformPointCloud(cloud, out, Eigen::Matrix4d) on a non-dense cloud (PCL <= 1.8 scalar form): double arithmetic on the float coordinates, the sum taken left
    to right, rounded to float; points with a non-finite coordinate are copied as they are."""
    T = np.asarray(T, dtype=np.float64).reshape(4, 4)
    x, y, z = (xyzi[:, k].astype(np.float64) for k in range(3))
    out = xyzi.copy()
    fin = np.isfinite(x) & np.isfinite(y) & np.isfinite(z)
    for r in range(3):
        v = ((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3]
        out[fin, r] = v[fin].astype(np.float32)
    return out


def first_data_association(S, state, scan_t, poses, has_pose=None, opt=None, key_dist=0.2, key_angle_deg=5.0):
    """The FIRST DataAssociation (lvi_initialize_surfel_orb.cpp:1175-1178): map from per-scan odometry poses [n_scans][16] (row-major scan -> map), only the SO3 spline of
    `state` is used.  Returns scans_in_map (absent scans NaN), key scan list, planes, points."""
    opt = dict(DEFAULTS, plane_lambda=0.6, **(opt or {}))      # SurfelAssociation is constructed with plane_lambda_ = 0.6 (:127, 240); 0.7 is set for the refinement rounds (:1182)
    o = _base_oracle(S)
    raw = _raw_scans(S)
    n = len(raw)
    q, p, ok = O.eval_lidar_pose(o, state, np.asarray(scan_t, dtype=np.float64))
    present = [bool(ok[s]) and (has_pose is None or bool(has_pose[s])) for s in range(n)]
    sim = np.full((n, S["H"] * S["W"], 4), np.nan, np.float32)
    sim[:, :, 3] = 0.0
    for s in range(n):
        if not present[s]:
            continue
        q_G_to_L0 = np.array([-q[s, 0], -q[s, 1], -q[s, 2], q[s, 3]])
        und = O.undistort(o, state, raw[s], q_G_to_L0, p[s], False)      # ScanUndistortion::undistortScan(correct_position = false)
        sim[s] = transform_cloud(und, poses[s])                          # undistortScanInMap(odom_data_map) / updateKeyScan: the same transformPointCloud
    key = key_scans(poses, present, key_dist, key_angle_deg)
    sim = sim.reshape(n, S["H"], S["W"], 4)
    if not key:
        return dict(scans_in_map=sim, key=key, planes=None, points=None)
    planes = surfel_map(sim[key], opt)                                   # map_cloud_ of LiDAROdometry: the key scans, in order
    pts = associate(S, sim, planes, opt) if len(planes["p4"]) else dict(pt=np.zeros((0, 3)), pt_map=np.zeros((0, 3)), t=np.zeros(0), plane=np.zeros(0, np.int32))
    return dict(scans_in_map=sim, key=key, planes=planes, points=pts)


def select_surfels(points, t_map, step):
    """averageTimeDownSmaple(step) over spoints_all_, then what addSurfMeasurement can take: {t_map, t} must be ordered (CheckTimeSpans, kontiki/trajectory_estimator.h:102-127)."""
    idx = np.arange(0, len(points["t"]), max(1, step))
    idx = idx[points["t"][idx] >= t_map]
    return points["pt"][idx], points["t"][idx], points["plane"][idx]


def solve_stage(S, state, stage, planes, points, opt=None, camsurf=None, max_iterations=None):
    """One TrajectoryManagerLVI solve through the oracle evaluator + oracle/lm.py.  stage: TrajFromSurfel | TrajFromLVI | TrajFromLVILandmarksOnly | SO3FromGyro."""
    opt = dict(DEFAULTS, **(opt or {}))
    o = _base_oracle(S)
    locks = stage_locks(stage, opt["opt_time_offset"])
    N, L = S["n_knots"], len(S["lm_t0"])
    if stage == "SO3FromGyro":
        o.set_imu(S["t_imu"], S["gyro"], np.zeros_like(S["acc"]), opt["w_gyro"], opt["w_acc"])
        o.set_so3_only(True)
        o.set_orientation_prior(S["t0"], [np.cos(0.5e-4), 0, 0, np.sin(0.5e-4)], opt["w_gyro"])
    elif stage == "TrajFromVisualFrames":
        o.set_imu(S["t_imu"], S["gyro"], S["acc"], opt["w_gyro"], opt["w_acc"])
        o.set_reproj(S["rep_lm"], S["rep_uv"], S["rep_t0"], opt["w_cam"], 1.0)
    else:
        o.set_imu(S["t_imu"], S["gyro"], S["acc"], opt["w_gyro"], opt["w_acc"])
        o.set_planes(planes["Pi"])
        pt, t, pid = select_surfels(points, S["t_map"], opt["downsample_step"])
        o.set_surfel(pt, t, pid, S["t_map"], 5.0, opt["w_surfel"])
        if stage != "TrajFromSurfel":
            o.set_reproj(S["rep_lm"], S["rep_uv"], S["rep_t0"], opt["w_cam"], 1.0)      # huber = w_cam, weight 1: the reference's argument swap (:525)
        if camsurf is not None:
            o.set_camsurf(camsurf[0], camsurf[1], S["t_map"], 5.0, opt["w_cam_surfel"])
    o.set_locks(locks)
    free = lm.free_tangent_indices(N, L, locks)
    if max_iterations is None:
        max_iterations = 30 if stage in ("SO3FromGyro", "TrajFromSurfel") else (200 if stage == "TrajFromVisualFrames" else 80)
    return lm.lm_solve(o, state, free, max_iterations=max_iterations, n_knots=N, n_landmarks=L)


def landmark_planes(S, state, planes, opt=None):
    """associateVisualPointsWithPlanes with q_LtoC / t_LinC from the current extrinsics: (landmark ids, plane ids) of the camera-landmark-to-surfel blocks."""
    opt = dict(DEFAULTS, **(opt or {}))
    o = _base_oracle(S)
    N = S["n_knots"]
    sl, sc = state[7 * N + 16:7 * N + 24], state[7 * N + 24:7 * N + 32]

    def qmul(a, b):
        ax, ay, az, aw = a; bx, by, bz, bw = b
        return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])

    def qrot(q, v):
        u = 2.0 * np.cross(q[:3], v)
        return v + q[3] * u + np.cross(q[:3], u)
    qc = np.array([-sc[0], -sc[1], -sc[2], sc[3]])
    q_LtoC = qmul(qc, sl[:4])
    t_LinC = qrot(qc, sl[4:7] - sc[4:7])
    pol = O.landmark_assoc(o, state, q_LtoC, t_LinC, S["t_map"], planes["p4"], planes["box_min"], planes["box_max"], opt["associated_radius"])
    lmk = np.nonzero(pol >= 0)[0].astype(np.int32)
    return lmk, pol[lmk].astype(np.int32)


def run_schedule(S, state0, refine_iterations=2, lvi_stage=True, camera_surfel_stage=False, opt=None):
    """Free-running schedule as lvx_host::Calibrator::Run: [DataAssociation -> trajInitFromSurfel] x refine_iterations, trajInitFromLVIdata, optionally the
    camera-surfel stage.  Returns (state, [stage dicts])."""
    x = np.array(state0, dtype=np.float64)
    log = []
    da = None
    for it in range(refine_iterations):
        da = data_association(S, x, opt)
        x, s = solve_stage(S, x, "TrajFromSurfel", da["planes"], da["points"], opt)
        log.append(dict(name="BatchOptimization" if it == 0 else "Refinement", lm=s, association=da))
    if lvi_stage:
        x, s = solve_stage(S, x, "TrajFromLVI", da["planes"], da["points"], opt)
        log.append(dict(name="trajInitFromLVIdata", lm=s))
    if camera_surfel_stage:
        cs = landmark_planes(S, x, da["planes"], opt)
        x, s = solve_stage(S, x, "TrajFromLVILandmarksOnly", da["planes"], da["points"], opt, camsurf=cs)
        log.append(dict(name="trajInitFromLVIdata+lm_splane", lm=s, camsurf=cs))
    return x, log


def run_fixed_stages(P, x0, threads=None, sparse=False, verbose=False):
    """Config-4 style problem (fixed surfel list, synth.make_bench_problem): trajInitFromSurfel (<= 30) then trajInitFromLVIdata (<= 80) through the oracle LM
    (sparse=True: oracle/lm_sparse.py — generic sparse normal equations + band Cholesky, any problem size; else the dense numpy LM of oracle/lm.py).
    Returns (state, [(stage, summary, seconds)])."""
    import time
    x, log = np.array(x0, dtype=np.float64), []
    N, L = P["n_knots"], P["n_landmarks"]
    for name, stage, iters, with_rep in (("trajInitFromSurfel", "TrajFromSurfel", 30, False), ("trajInitFromLVIdata", "TrajFromLVI", 80, True)):
        locks = stage_locks(stage)
        o = O.Oracle()
        o.set_spline(P["t0"], P["dt"], N)
        c = P["camera"]
        o.set_camera(c["rows"], c["cols"], c["readout"], c["fx"], c["fy"], c["cx"], c["cy"], c["k1"], c["k2"], c["p1"], c["p2"], c["k3"])
        o.set_imu(P["t_imu"], P["gyro"], P["acc"], P["w_gyro"], P["w_acc"])
        o.set_planes(P["planes"])
        o.set_surfel(P["surf_pt"], P["surf_t"], P["surf_plane"], P["t_map"], P["huber_surf"], P["w_surf"])
        o.set_landmarks(P["lm_uv"], P["lm_t0"])
        if with_rep:
            o.set_reproj(P["rep_lm"], P["rep_uv"], P["rep_t0"], P["huber_rep"], P["w_rep"])
        o.set_camsurf(P["cs_lm"], P["cs_plane"], P["t_map"], P["huber_cs"], P["w_cs"])
        o.set_locks(locks)
        if threads:
            o.set_threads(threads)
        free = lm.free_tangent_indices(N, L, locks)
        t0 = time.perf_counter()
        if sparse:
            from . import lm_sparse
            x, s = lm_sparse.lm_solve(o, x, free, N, L, max_iterations=iters, verbose=verbose)
        else:
            x, s = lm.lm_solve(o, x, free, max_iterations=iters, n_knots=N, n_landmarks=L)
        log.append((name, s, time.perf_counter() - t0))
    return x, log
