"""Seeded synthetic VLP-16 + IMU + mono streams for the calibration solve.

Also holds a vectorised numpy evaluator of the split R3 + SO3 uniform cubic B-spline
(reference: kontiki/trajectories/uniform_{r3,so3}_spline_trajectory.h, spline_base.h:19-29)
that is used to synthesise measurements which are exactly consistent with a ground-truth state.
All quaternions are stored (x, y, z, w) like Eigen::Map<Quaternion> (spline_base.h:118-127).
"""
import numpy as np

GRAVITY = -9.79  # kontiki/sensors/imu.h:25

M = np.array([[1, 4, 1, 0], [-3, 0, 3, 0], [3, -6, 3, 0], [-1, 3, -3, 1]], dtype=np.float64) / 6.0
M_CUMUL = np.array([[6, 5, 1, 0], [0, 3, 3, 0], [0, -3, 3, 0], [0, 1, -2, 1]], dtype=np.float64) / 6.0


# ---------------------------------------------------------------------------------------------
# quaternion helpers, (x, y, z, w) storage, Hamilton product
# ---------------------------------------------------------------------------------------------
def qmul(a, b):
    ax, ay, az, aw = np.moveaxis(np.asarray(a, dtype=np.float64), -1, 0)
    bx, by, bz, bw = np.moveaxis(np.asarray(b, dtype=np.float64), -1, 0)
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def qconj(q):
    q = np.asarray(q, dtype=np.float64)
    return q * np.array([-1.0, -1.0, -1.0, 1.0])


def qrot(q, v):
    q = np.asarray(q, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    qv = q[..., :3]
    uv = 2.0 * np.cross(qv, v)
    return v + q[..., 3:4] * uv + np.cross(qv, uv)


def qexp_half(v):
    """exp of a pure quaternion with vector part v (half-angle vector)."""
    v = np.asarray(v, dtype=np.float64)
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    s = np.where(n > 1e-8, np.sin(n) / np.where(n > 1e-8, n, 1.0), 1.0)
    return np.concatenate([s * v, np.cos(n)], axis=-1)


def qlog_half(q):
    """log of a unit quaternion -> half-angle vector (no hemisphere handling, as the reference)."""
    q = np.asarray(q, dtype=np.float64)
    v = q[..., :3]
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    k = np.where(n > 1e-8, np.arctan2(n, q[..., 3:4]) / np.where(n > 1e-8, n, 1.0), 1.0)
    return k * v


def q_from_rotvec(phi):
    return qexp_half(0.5 * np.asarray(phi, dtype=np.float64))


def q_from_rpy(roll, pitch, yaw):
    qx = q_from_rotvec([roll, 0, 0])
    qy = q_from_rotvec([0, pitch, 0])
    qz = q_from_rotvec([0, 0, yaw])
    return qmul(qz, qmul(qy, qx))


# ---------------------------------------------------------------------------------------------
# numpy spline evaluator
# ---------------------------------------------------------------------------------------------
class Spline:
    def __init__(self, t0, dt, r3, so3):
        self.t0, self.dt = float(t0), float(dt)
        self.r3 = np.ascontiguousarray(r3, dtype=np.float64)
        self.so3 = np.ascontiguousarray(so3, dtype=np.float64)
        self.n = len(self.r3)

    @property
    def t_min(self):
        return self.t0

    @property
    def t_max(self):
        return self.t0 + (self.n - 3) * self.dt

    def _iu(self, t):
        s = (np.asarray(t, dtype=np.float64) - self.t0) / self.dt
        i0 = np.floor(s).astype(np.int64)
        if np.any(i0 < 0) or np.any(i0 > self.n - 4):
            raise IndexError("time out of range for spline")
        return i0, s - i0

    def eval(self, t):
        t = np.atleast_1d(np.asarray(t, dtype=np.float64))
        i0, u = self._iu(t)
        one = np.ones_like(u)
        zero = np.zeros_like(u)
        U = np.stack([one, u, u * u, u * u * u], axis=-1)
        dU = np.stack([zero, one, 2 * u, 3 * u * u], axis=-1) / self.dt
        ddU = np.stack([zero, zero, 2 * one, 6 * u], axis=-1) / self.dt ** 2
        idx = i0[:, None] + np.arange(4)[None, :]
        cp = self.r3[idx]  # (n, 4, 3)
        pos = np.einsum("nj,njk->nk", U @ M, cp)
        vel = np.einsum("nj,njk->nk", dU @ M, cp)
        acc = np.einsum("nj,njk->nk", ddU @ M, cp)
        B = U @ M_CUMUL
        dB = dU @ M_CUMUL
        qc = self.so3[idx]  # (n, 4, 4)
        q = qc[:, 0]
        parts = [np.tile(np.array([0.0, 0, 0, 1.0]), (len(t), 1)) for _ in range(3)]
        for j in range(1, 4):
            om = qlog_half(qmul(qconj(qc[:, j - 1]), qc[:, j]))
            e = qexp_half(B[:, j:j + 1] * om)
            q = qmul(q, e)
            for m in range(3):
                if m == j - 1:
                    w = np.concatenate([dB[:, j:j + 1] * om, np.zeros((len(t), 1))], axis=-1)
                    parts[m] = qmul(parts[m], w)
                parts[m] = qmul(parts[m], e)
        dq = qmul(qc[:, 0], parts[0] + parts[1] + parts[2])
        angvel = 2.0 * qmul(dq, qconj(q))[:, :3]
        return {"pos": pos, "vel": vel, "acc": acc, "quat": q, "angvel": angvel}


# ---------------------------------------------------------------------------------------------
# state packing (layout documented in include/lvx.h)
# ---------------------------------------------------------------------------------------------
def pack_state(r3, so3, imu, lidar, cam, rho):
    return np.concatenate([np.ravel(r3), np.ravel(so3), np.ravel(imu), np.ravel(lidar), np.ravel(cam), np.ravel(rho)]).astype(np.float64)


def unpack_state(state, n_knots, n_landmarks):
    s = np.asarray(state, dtype=np.float64)
    o = 0
    r3 = s[o:o + 3 * n_knots].reshape(n_knots, 3); o += 3 * n_knots
    so3 = s[o:o + 4 * n_knots].reshape(n_knots, 4); o += 4 * n_knots
    imu = s[o:o + 16]; o += 16
    lidar = s[o:o + 8]; o += 8
    cam = s[o:o + 8]; o += 8
    rho = s[o:o + n_landmarks]
    return {"r3": r3, "so3": so3, "imu": imu, "lidar": lidar, "cam": cam, "rho": rho}


def imu_block(roll=0.01, pitch=0.01, ba=(0, 0, 0), bg=(0, 0, 0)):
    """q_rel(x,y,z,w)=identity, p_rel=0, tau=0, roll, pitch, b_a, b_g (sensors.h:99-110, imu.h:126-127)."""
    return np.array([0, 0, 0, 1, 0, 0, 0, 0, roll, pitch, *ba, *bg], dtype=np.float64)


def sensor_block(q_xyzw, p, tau=0.0):
    return np.array([*q_xyzw, *p, tau], dtype=np.float64)


def gravity_vec(roll, pitch):
    cr, sr, cp, sp = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch)
    return np.array([-sp * cr * GRAVITY, sr * GRAVITY, -cr * cp * GRAVITY])


DEFAULT_CAMERA = dict(rows=720, cols=1280, readout=0.0666, fx=530.174987792968, fy=530.094970703125,
                      cx=635.119995117187, cy=356.522003173828, k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0)  # cfg/lvi.yaml:54-78


def _unproject(cam, uv):
    return np.stack([(uv[..., 0] - cam["cx"]) / cam["fx"], (uv[..., 1] - cam["cy"]) / cam["fy"], np.ones(uv.shape[:-1])], axis=-1)


def _project(cam, X):
    return np.stack([cam["fx"] * X[..., 0] / X[..., 2] + cam["cx"], cam["fy"] * X[..., 1] / X[..., 2] + cam["cy"]], axis=-1)


def make_trajectory(n_knots, t0, dt, rng, pos_amp=2.0, rot_amp=0.6):
    """Smooth sum-of-sinusoids control points (pos <= 0.5 Hz, rot <= 0.7 Hz)."""
    tk = t0 + dt * (np.arange(n_knots) - 1.0)
    r3 = np.zeros((n_knots, 3))
    phi = np.zeros((n_knots, 3))
    for ax in range(3):
        for _ in range(3):
            f = rng.uniform(0.05, 0.5)
            r3[:, ax] += pos_amp / 3 * rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * f * tk + rng.uniform(0, 2 * np.pi))
            f = rng.uniform(0.05, 0.7)
            phi[:, ax] += rot_amp / 3 * rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * f * tk + rng.uniform(0, 2 * np.pi))
    so3 = q_from_rotvec(phi)
    so3 /= np.linalg.norm(so3, axis=1, keepdims=True)
    return r3, so3


def make_problem(seed=4, duration=10.0, dt=0.02, imu_rate=400.0, n_surfel=2000, n_planes=40, n_landmarks=50, views_per_lm=10,
                 cam_rate=20.0, n_camsurf=0, pad=0.2, noise=True, camera=None, t_start=100.0, state_noise=1e-2):
    """Build a consistent synthetic calibration problem.  Returns a dict with measurement arrays, the
    ground-truth state ("state_true") and a perturbed state ("state0")."""
    rng = np.random.default_rng(seed)
    cam = dict(DEFAULT_CAMERA if camera is None else camera)
    t0 = t_start - pad
    t_end = t_start + duration
    n_knots = 4
    while t0 + (n_knots - 3) * dt < t_end + pad:  # SplineEntity::ExtendTo (spline_base.h:374-378)
        n_knots += 1
    r3, so3 = make_trajectory(n_knots, t0, dt, rng)
    sp = Spline(t0, dt, r3, so3)
    # true calibration
    q_LI = q_from_rpy(np.deg2rad(2.0), np.deg2rad(-3.0), np.deg2rad(91.0))
    p_LI = np.array([0.05, -0.10, 0.12])
    q_CI = qmul(q_from_rpy(np.deg2rad(-90.0), 0.0, np.deg2rad(-90.0)), q_from_rotvec(np.deg2rad([2.0, 0, 0])))
    p_CI = np.array([-0.22, 0.02, 0.22])
    roll, pitch = 0.02, -0.015
    ba = np.array([0.05, 0.02, -0.03]); bg = np.array([0.01, -0.02, 0.005])
    g = gravity_vec(roll, pitch)
    ns = 1.0 if noise else 0.0
    # --- IMU ---
    n_imu = int(round(duration * imu_rate))
    t_imu = t_start + np.arange(n_imu) / imu_rate
    e = sp.eval(t_imu)
    gyro = qrot(qconj(e["quat"]), e["angvel"]) + bg + ns * 1.745e-4 * np.sqrt(imu_rate) * rng.standard_normal((n_imu, 3))
    acc = qrot(qconj(e["quat"]), e["acc"] + g) + ba + ns * 5.88e-4 * np.sqrt(imu_rate) * rng.standard_normal((n_imu, 3))
    # --- planes + surfel points ---
    t_map = t_start + 0.05
    e0 = sp.eval([t_map])
    R0q, p0 = e0["quat"][0], e0["pos"][0]
    nrm = rng.standard_normal((n_planes, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    dist = rng.uniform(2.0, 12.0, n_planes)
    Pi = nrm * dist[:, None]  # closest point to the origin of L0; plane: n.x - d = 0
    planes = [Pi]
    sid = rng.integers(0, n_planes, n_surfel)
    t_s = np.sort(rng.uniform(t_map + 1e-3, t_end - 1e-3, n_surfel))
    n_s = nrm[sid]
    e1 = np.cross(n_s, np.array([0.3, -0.5, 0.8])); e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = np.cross(n_s, e1)
    pM = Pi[sid] + rng.uniform(-5, 5, (n_surfel, 1)) * e1 + rng.uniform(-5, 5, (n_surfel, 1)) * e2

    def map_to_lidar(pM_, t_k):
        ek = sp.eval(t_k)
        pI0 = qrot(q_LI, pM_) + p_LI
        pG = qrot(R0q, pI0) + p0
        pIk = qrot(qconj(ek["quat"]), pG - ek["pos"])
        return qrot(qconj(q_LI), pIk - p_LI)

    pt_s = map_to_lidar(pM, t_s) + ns * 0.02 * rng.standard_normal((n_surfel, 3)) if n_surfel else np.zeros((0, 3))
    # --- landmarks + reprojection ---
    row_d = cam["readout"] / cam["rows"]
    t_frames = np.arange(t_start + 0.1, t_end - 0.2, 1.0 / cam_rate)

    def cam_pose(t):
        ee = sp.eval(t)
        return qmul(ee["quat"], np.broadcast_to(q_CI, ee["quat"].shape)), qrot(ee["quat"], p_CI) + ee["pos"]

    lm_uv = np.zeros((n_landmarks, 2)); lm_t0 = np.zeros(n_landmarks); rho = np.zeros(n_landmarks)
    rep_lm, rep_uv, rep_t0 = [], [], []
    PW = np.zeros((n_landmarks, 3))
    n_first = max(1, len(t_frames) - views_per_lm)
    for l in range(n_landmarks):
        f0 = int(rng.integers(0, n_first))
        uv = np.array([rng.uniform(100, cam["cols"] - 100), rng.uniform(100, cam["rows"] - 100)])
        z = rng.uniform(2.0, 15.0)
        tr = t_frames[f0] + uv[1] * row_d
        qc, pc = cam_pose(np.array([tr]))
        Pw = qrot(qc[0], _unproject(cam, uv) * z) + pc[0]
        PW[l] = Pw
        lm_uv[l], lm_t0[l], rho[l] = uv, t_frames[f0], 1.0 / z
        for v in range(views_per_lm):
            fi = f0 + v
            if fi >= len(t_frames):
                break
            if v == 0:
                o = uv.copy()
            else:
                vv = cam["cy"]
                ok = True
                for _ in range(4):  # rolling-shutter fixed point on the row time
                    qo, po = cam_pose(np.array([t_frames[fi] + vv * row_d]))
                    Xc = qrot(qconj(qo[0]), Pw - po[0])
                    if Xc[2] < 0.2:
                        ok = False
                        break
                    o = _project(cam, Xc)
                    vv = o[1]
                if not ok or not (0 <= o[0] < cam["cols"] and 0 <= o[1] < cam["rows"]):
                    continue
                o = o + ns * 0.5 * rng.standard_normal(2)
            rep_lm.append(l); rep_uv.append(o); rep_t0.append(t_frames[fi])
    rep_lm = np.array(rep_lm, dtype=np.int32); rep_uv = np.array(rep_uv).reshape(-1, 2); rep_t0 = np.array(rep_t0)
    # --- camera-landmark-to-surfel: define a plane through the landmark's map-frame point ---
    cs_lm = np.zeros(0, dtype=np.int32); cs_plane = np.zeros(0, dtype=np.int32)
    if n_camsurf and n_landmarks:
        cs_lm = rng.choice(n_landmarks, size=min(n_camsurf, n_landmarks), replace=False).astype(np.int32)
        extra = []
        for l in cs_lm:
            # map-frame point exactly as camera_surfel_landmark.h:38-76 sees it (pose at the view's t0, no row time)
            ek = sp.eval([lm_t0[l]])
            p_I = qrot(q_CI, _unproject(cam, lm_uv[l]) / (rho[l] + 1e-8)) + p_CI
            pI0 = qrot(qconj(R0q), qrot(ek["quat"][0], p_I) + ek["pos"][0] - p0)
            pM_ = qrot(qconj(q_LI), pI0 - p_LI)
            n_ = rng.standard_normal(3); n_ /= np.linalg.norm(n_)
            d_ = float(n_ @ pM_)
            if abs(d_) < 0.5:
                n_ = pM_ / np.linalg.norm(pM_); d_ = float(n_ @ pM_)
            extra.append(n_ * d_)
        cs_plane = (n_planes + np.arange(len(cs_lm))).astype(np.int32)
        planes.append(np.array(extra))
    planes = np.concatenate(planes, axis=0)
    imu_true = imu_block(roll, pitch, ba, bg)
    state_true = pack_state(r3, so3, imu_true, sensor_block(q_LI, p_LI), sensor_block(q_CI, p_CI), rho)
    # perturbed start: control points noise, extrinsics 3 deg / 5 cm, biases 0, default gravity guess
    r3p = r3 + state_noise * rng.standard_normal(r3.shape)
    so3p = qmul(q_from_rotvec(state_noise * rng.standard_normal((n_knots, 3))), so3)
    so3p /= np.linalg.norm(so3p, axis=1, keepdims=True)
    dqL = q_from_rotvec(np.deg2rad(3.0) * np.array([0.6, -0.5, 0.62]))
    dqC = q_from_rotvec(np.deg2rad(3.0) * np.array([-0.4, 0.7, 0.59]))
    state0 = pack_state(r3p, so3p, imu_block(0.01, 0.01), sensor_block(qmul(dqL, q_LI), p_LI + np.array([0.03, -0.03, 0.03])),
                        sensor_block(qmul(dqC, q_CI), p_CI + np.array([-0.03, 0.03, 0.03])), rho * (1.0 + 0.05 * rng.standard_normal(n_landmarks)))
    return dict(t0=t0, dt=dt, n_knots=n_knots, camera=cam, t_imu=t_imu, gyro=gyro, acc=acc, w_gyro=28.0, w_acc=18.0,
                planes=planes, surf_pt=pt_s, surf_t=t_s, surf_plane=sid.astype(np.int32), t_map=t_map, huber_surf=5.0, w_surf=10.0,
                n_landmarks=n_landmarks, lm_uv=lm_uv, lm_t0=lm_t0, rep_lm=rep_lm, rep_uv=rep_uv, rep_t0=rep_t0,
                huber_rep=5.0, w_rep=1.0,  # quirk: w_cam lands in the Huber slot, weight = 1 (trajectory_manager_lvi.cpp:525)
                cs_lm=cs_lm, cs_plane=cs_plane, huber_cs=5.0, w_cs=30.0,
                state_true=state_true, state0=state0, t_start=t_start, t_end=t_end)


def _bench_visual(sp, cam, rng, t_start, t_end, n_reproj, q_CI, p_CI, tracks, views_per_lm, cam_rate, obs_per_frame, noise_px=0.5):
    """Landmarks + rolling-shutter observations, exactly consistent with the trajectory up to pixel noise (vectorised).

    tracks = "orb": ORB-SLAM-like co-visibility: the frames come in windows of `views_per_lm` consecutive frames (1 / cam_rate apart) that
        all see the same ~obs_per_frame map points (>= 100 observations per frame, every frame pair of a window co-visible); the windows are
        spread evenly over the sequence.  The reference observation of a map point is its first view (70 %) or the second / third one.
    tracks = "sparse": SURVEY 8d's literal reading (n_reproj / views landmarks, each first seen in a random frame of a continuous 20 Hz
        stream and tracked over the next views_per_lm frames: ~5 observations per frame, no two blocks share a frame pair).
    Observations that leave the image or come closer than 0.5 m are dropped; landmarks keep the reference's rule of > 5 observations
    (trajectory_manager_lvi.cpp:518-524); the result is trimmed to n_reproj blocks (whole landmarks)."""
    row_d = cam["readout"] / cam["rows"]
    t_frames = np.arange(t_start + 0.1, t_end - 0.2, 1.0 / cam_rate)
    over = 1.5
    if tracks == "orb":
        n_win = max(1, int(round(n_reproj / float(views_per_lm * obs_per_frame))))
        w0 = np.linspace(0, len(t_frames) - views_per_lm - 1, n_win).astype(np.int64)
        per = int(obs_per_frame * over)
        win = np.repeat(np.arange(n_win), per)
        n_lm = len(win)
        ref_k = rng.choice(3, size=n_lm, p=[0.7, 0.2, 0.1])
        f_first = w0[win]
    else:
        n_lm = int(n_reproj // views_per_lm * over)
        f_first = rng.integers(0, len(t_frames) - views_per_lm, n_lm)
        ref_k = np.zeros(n_lm, dtype=np.int64)
    lm_uv = np.stack([rng.uniform(150, cam["cols"] - 150, n_lm), rng.uniform(100, cam["rows"] - 100, n_lm)], axis=1)
    z = rng.uniform(3.0, 15.0, n_lm)
    lm_t0 = t_frames[f_first + ref_k]

    def cam_pose(t):
        e = sp.eval(t)
        return qmul(e["quat"], np.broadcast_to(q_CI, e["quat"].shape)), qrot(e["quat"], p_CI) + e["pos"]

    qc, pc = cam_pose(lm_t0 + lm_uv[:, 1] * row_d)
    PW = qrot(qc, _unproject(cam, lm_uv) * z[:, None]) + pc
    lm = np.repeat(np.arange(n_lm), views_per_lm)
    view = np.tile(np.arange(views_per_lm), n_lm)
    t0o = t_frames[f_first[lm] + view]
    vv = np.full(len(lm), cam["cy"])
    for _ in range(5):   # fixed point on the row time: the observation's row decides when the pose is sampled
        qo, po = cam_pose(t0o + vv * row_d)
        Xc = qrot(qconj(qo), PW[lm] - po)
        uv = _project(cam, np.where(Xc[:, 2:3] > 1e-3, Xc, np.array([0.0, 0.0, 1.0])))
        vv = np.clip(uv[:, 1], 0.0, cam["rows"] - 1.0)
    is_ref = view == ref_k[lm]
    ok = (Xc[:, 2] > 0.5) & (uv[:, 0] >= 1) & (uv[:, 0] < cam["cols"] - 1) & (uv[:, 1] >= 1) & (uv[:, 1] < cam["rows"] - 1)
    uv = uv + noise_px * rng.standard_normal(uv.shape)
    uv[is_ref] = lm_uv[lm[is_ref]]   # the reference observation itself is one of the blocks (r = 0 rows)
    ok |= is_ref
    cnt = np.bincount(lm[ok], minlength=n_lm)
    keep_lm = cnt > 5
    if tracks == "orb":   # the same number of map points per window
        order = np.argsort(np.where(keep_lm, 0, 1) * n_lm + np.arange(n_lm), kind="stable")
        rank_in_win = np.empty(n_lm, dtype=np.int64)
        for w in range(n_win):
            idx = order[win[order] == w]
            rank_in_win[idx] = np.arange(len(idx))
        budget = np.cumsum(np.where(keep_lm, cnt, 0)[np.lexsort((np.arange(n_lm), rank_in_win))])
        sel = np.zeros(n_lm, dtype=bool)
        sel[np.lexsort((np.arange(n_lm), rank_in_win))[budget <= n_reproj]] = True
        keep_lm &= sel
    else:
        csum = np.cumsum(np.where(keep_lm, cnt, 0))
        keep_lm &= csum <= n_reproj
    new_id = np.cumsum(keep_lm) - 1
    rows = ok & keep_lm[lm]
    return dict(n_landmarks=int(keep_lm.sum()), lm_uv=lm_uv[keep_lm], lm_t0=lm_t0[keep_lm], rho=1.0 / z[keep_lm],
                rep_lm=new_id[lm[rows]].astype(np.int32), rep_uv=uv[rows], rep_t0=t0o[rows])


def make_bench_problem(seed=4, n_imu=200_000, n_surfel=1_000_000, n_reproj=50_000, n_planes=2000, dt=0.02, imu_rate=400.0,
                       views_per_lm=10, cam_rate=20.0, pad=0.2, t_start=100.0, tracks="orb", obs_per_frame=100):
    """BASELINE.json config 4 shapes (1 M surfel / 200 k IMU / 50 k ORB) generated fully vectorised; every measurement is consistent with
    the ground-truth state up to its sensor noise, so the problem can be solved to convergence.  `tracks`: see _bench_visual."""
    rng = np.random.default_rng(seed)
    cam = dict(DEFAULT_CAMERA)
    duration = n_imu / imu_rate
    t0 = t_start - pad
    t_end = t_start + duration
    n_knots = int(np.ceil((t_end + pad - t0) / dt)) + 3
    while t0 + (n_knots - 3) * dt < t_end + pad:
        n_knots += 1
    r3, so3 = make_trajectory(n_knots, t0, dt, rng)
    sp = Spline(t0, dt, r3, so3)
    q_LI = q_from_rpy(np.deg2rad(2.0), np.deg2rad(-3.0), np.deg2rad(91.0)); p_LI = np.array([0.05, -0.10, 0.12])
    q_CI = qmul(q_from_rpy(np.deg2rad(-90.0), 0.0, np.deg2rad(-90.0)), q_from_rotvec(np.deg2rad([2.0, 0, 0]))); p_CI = np.array([-0.22, 0.02, 0.22])
    roll, pitch = 0.02, -0.015
    ba = np.array([0.05, 0.02, -0.03]); bg = np.array([0.01, -0.02, 0.005])
    g = gravity_vec(roll, pitch)
    # +0.37 sample: real IMU stamps never sit exactly on knot times; exact coincidences make the reference throw
    # "No segment found" through the 4-knot segment bounds (spline_base.h:196-221) depending on rounding
    t_imu = t_start + (np.arange(n_imu) + 0.37) / imu_rate
    e = sp.eval(t_imu)
    gyro = qrot(qconj(e["quat"]), e["angvel"]) + bg + 1.745e-4 * np.sqrt(imu_rate) * rng.standard_normal((n_imu, 3))
    acc = qrot(qconj(e["quat"]), e["acc"] + g) + ba + 5.88e-4 * np.sqrt(imu_rate) * rng.standard_normal((n_imu, 3))
    t_map = t_start + 0.05
    e0 = sp.eval([t_map]); R0q, p0 = e0["quat"][0], e0["pos"][0]
    nrm = rng.standard_normal((n_planes, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    Pi = nrm * rng.uniform(2.0, 12.0, n_planes)[:, None]
    sid = rng.integers(0, n_planes, n_surfel)
    t_s = np.sort(rng.uniform(t_map + 1e-3, t_end - 1e-3, n_surfel))
    n_s = nrm[sid]
    e1 = np.cross(n_s, np.array([0.3, -0.5, 0.8])); e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = np.cross(n_s, e1)
    pM = Pi[sid] + rng.uniform(-5, 5, (n_surfel, 1)) * e1 + rng.uniform(-5, 5, (n_surfel, 1)) * e2
    ek = sp.eval(t_s)
    pG = qrot(R0q, qrot(q_LI, pM) + p_LI) + p0
    pt_s = qrot(qconj(q_LI), qrot(qconj(ek["quat"]), pG - ek["pos"]) - p_LI) + 0.02 * rng.standard_normal((n_surfel, 3))
    V = _bench_visual(sp, cam, rng, t_start, t_end, n_reproj, q_CI, p_CI, tracks, views_per_lm, cam_rate, obs_per_frame)
    rho = V["rho"]
    state_true = pack_state(r3, so3, imu_block(roll, pitch, ba, bg), sensor_block(q_LI, p_LI), sensor_block(q_CI, p_CI), rho)
    r3p = r3 + 1e-2 * rng.standard_normal(r3.shape)
    so3p = qmul(q_from_rotvec(1e-2 * rng.standard_normal((n_knots, 3))), so3); so3p /= np.linalg.norm(so3p, axis=1, keepdims=True)
    dqL = q_from_rotvec(np.deg2rad(3.0) * np.array([0.6, -0.5, 0.62])); dqC = q_from_rotvec(np.deg2rad(3.0) * np.array([-0.4, 0.7, 0.59]))
    state0 = pack_state(r3p, so3p, imu_block(0.01, 0.01), sensor_block(qmul(dqL, q_LI), p_LI + 0.03), sensor_block(qmul(dqC, q_CI), p_CI - 0.03),
                        rho * (1.0 + 0.05 * rng.standard_normal(len(rho))))
    return dict(t0=t0, dt=dt, n_knots=n_knots, camera=cam, t_imu=t_imu, gyro=gyro, acc=acc, w_gyro=28.0, w_acc=18.0,
                planes=Pi, surf_pt=pt_s, surf_t=t_s, surf_plane=sid.astype(np.int32), t_map=t_map, huber_surf=5.0, w_surf=10.0,
                n_landmarks=V["n_landmarks"], lm_uv=V["lm_uv"], lm_t0=V["lm_t0"], rep_lm=V["rep_lm"], rep_uv=V["rep_uv"], rep_t0=V["rep_t0"], huber_rep=5.0, w_rep=1.0,
                cs_lm=np.zeros(0, dtype=np.int32), cs_plane=np.zeros(0, dtype=np.int32), huber_cs=5.0, w_cs=30.0,
                state_true=state_true, state0=state0, t_start=t_start, t_end=t_end, tracks=tracks)


# ---------------------------------------------------------------------------------------------------------
# upstream kernels: synthetic VLP-16 sweep, voxel cloud, organised scan + surfel planes
# ---------------------------------------------------------------------------------------------------------
RS_POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad", "<f4"), ("intensity", "u1"), ("pad2", "u1"), ("ring", "<u2"),
                     ("pad3", "<u4"), ("timestamp", "<f8")])   # RsPointXYZIRT (aloam scanRegistration.cpp:57-66), 32 bytes


def _raycast_room(dirs, origin, half=(10.0, 7.5), z_lo=-1.5, z_hi=1.5, pillars=((3, 2, 0.3), (-4, 3, 0.4), (5, -3, 0.35), (-2, -4, 0.3))):
    """Range along unit rays to the inside of a box room (20 x 15 x 3 m) with 4 vertical cylinders; origin: one point or one per ray."""
    d = dirs.astype(np.float64)
    o = np.broadcast_to(np.asarray(origin, dtype=np.float64), d.shape)
    t = np.full(len(d), np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        for ax, (lo, hi) in enumerate(((-half[0], half[0]), (-half[1], half[1]), (z_lo, z_hi))):
            for wall in (lo, hi):
                tt = (wall - o[:, ax]) / d[:, ax]
                p = o + tt[:, None] * d
                ok = (tt > 0) & (np.abs(p[:, 0]) <= half[0] + 1e-9) & (np.abs(p[:, 1]) <= half[1] + 1e-9) & (p[:, 2] >= z_lo - 1e-9) & (p[:, 2] <= z_hi + 1e-9)
                t = np.where(ok & (tt < t), tt, t)
        for cx, cy, r in pillars:
            a = d[:, 0] ** 2 + d[:, 1] ** 2
            b = 2 * ((o[:, 0] - cx) * d[:, 0] + (o[:, 1] - cy) * d[:, 1])
            c = (o[:, 0] - cx) ** 2 + (o[:, 1] - cy) ** 2 - r * r
            disc = b * b - 4 * a * c
            tt = (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a)
            z = o[:, 2] + tt * d[:, 2]
            ok = (disc > 0) & (tt > 0) & (z >= z_lo) & (z <= z_hi)
            t = np.where(ok & (tt < t), tt, t)
    return t


def make_vlp16_sweep(seed=1, n_az=1800, n_rings=16, noise=0.02, drop_frac=0.01, range_quantum=0.0, xyz_quantum=0.0):
    """One VLP-16-like sweep, azimuth-major firing order (all rings per azimuth step), range noise, a few NaN and near returns.
    range_quantum: ranges rounded to a multiple of it (a real VLP-16 reports 2 mm steps); xyz_quantum: coordinates rounded to a lattice (recordings that
    store millimetre integers) — both make EQUAL curvatures inside a sector common, which the continuous noise never does."""
    rng = np.random.default_rng(seed)
    az = np.deg2rad(np.arange(n_az) * (360.0 / n_az))
    el = np.deg2rad(np.linspace(-15.0, 15.0, n_rings))
    A, E = np.meshgrid(az, el, indexing="ij")   # (n_az, n_rings)
    dirs = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], axis=-1).reshape(-1, 3)
    rngs = _raycast_room(dirs, (0.3, -0.2, 0.0)) + noise * rng.standard_normal(len(dirs))
    if range_quantum > 0:
        rngs = np.round(rngs / range_quantum) * range_quantum
    pts = np.zeros(len(dirs), dtype=RS_POINT)
    xyz = dirs * rngs[:, None]
    if xyz_quantum > 0:
        xyz = np.round(xyz / xyz_quantum) * xyz_quantum
    xyz = xyz.astype(np.float32)
    pts["x"], pts["y"], pts["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    pts["ring"] = np.tile(np.arange(n_rings, dtype=np.uint16), n_az)
    pts["intensity"] = rng.integers(0, 255, len(dirs)).astype(np.uint8)
    pts["timestamp"] = 1638000000.0 + np.repeat(np.arange(n_az) / n_az * 0.1, n_rings)
    k = rng.random(len(dirs))
    pts["x"][k < drop_frac] = np.nan                       # dropped returns
    near = (k > 1 - drop_frac / 2)
    pts["x"][near] *= 0.01; pts["y"][near] *= 0.01; pts["z"][near] *= 0.01   # inside minimum_range
    return pts


def make_voxel_cloud(seed=2, n=100_000):
    """60 % on 12 planes (sigma 2 cm), 30 % on 6 cylinders, 10 % uniform in 40 x 40 x 6 m; float32 xyzi."""
    rng = np.random.default_rng(seed)
    n_pl, n_cy = int(0.6 * n), int(0.3 * n)
    n_un = n - n_pl - n_cy
    out = []
    pid = rng.integers(0, 12, n_pl)
    nrm = rng.standard_normal((12, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    ctr = rng.uniform([-15, -15, -2], [15, 15, 2], (12, 3))
    e1 = np.cross(nrm, [0.3, -0.5, 0.8]); e1 /= np.linalg.norm(e1, axis=1, keepdims=True); e2 = np.cross(nrm, e1)
    uv = rng.uniform(-4, 4, (n_pl, 2))
    out.append(ctr[pid] + uv[:, :1] * e1[pid] + uv[:, 1:] * e2[pid] + 0.02 * rng.standard_normal((n_pl, 1)) * nrm[pid])
    cid = rng.integers(0, 6, n_cy)
    cc = rng.uniform([-15, -15], [15, 15], (6, 2)); cr = rng.uniform(0.2, 0.8, 6)
    th = rng.uniform(0, 2 * np.pi, n_cy); zz = rng.uniform(-3, 3, n_cy)
    out.append(np.stack([cc[cid, 0] + cr[cid] * np.cos(th), cc[cid, 1] + cr[cid] * np.sin(th), zz], axis=1) + 0.01 * rng.standard_normal((n_cy, 3)))
    out.append(rng.uniform([-20, -20, -3], [20, 20, 3], (n_un, 3)))
    xyz = np.concatenate(out)[rng.permutation(n)]
    return np.concatenate([xyz, rng.uniform(0, 255, (n, 1))], axis=1).astype(np.float32)


def tile_voxel_cloud(cloud, tiles, pitch=40.0):
    """A larger map at the same local density: `tiles` shifted copies of `cloud` on a square grid of `pitch` metres (a 410 k-point map cloud is ~4 tiles of the
    100 k-point config-2 cloud, a 4 M-point one 40)."""
    side = int(np.ceil(np.sqrt(tiles)))
    out = []
    for k in range(tiles):
        c = cloud.copy()
        c[:, 0] += np.float32(pitch * (k % side)); c[:, 1] += np.float32(pitch * (k // side))
        out.append(c)
    return np.concatenate(out)


def rigid_move(xyzi, t=(0.1, 0.05, 0.02), yaw_deg=1.0):
    c, s = np.cos(np.deg2rad(yaw_deg)), np.sin(np.deg2rad(yaw_deg))
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    out = np.array(xyzi, dtype=np.float32)
    out[:, :3] = (xyzi[:, :3].astype(np.float64) @ R.T + np.asarray(t)).astype(np.float32)
    return out


def make_assoc_problem(seed=5, H=16, W=1800, n_planes=400):
    """Organised scan (H x W, float xyzi, NaN holes) in the map frame + surfel planes (p4, AABB) cut from the room's surfaces."""
    rng = np.random.default_rng(seed)
    az = np.deg2rad(np.arange(W) * (360.0 / W)); el = np.deg2rad(np.linspace(-15, 15, H))
    E, A = np.meshgrid(el, az, indexing="ij")
    dirs = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], axis=-1).reshape(-1, 3)
    r = _raycast_room(dirs, (0.3, -0.2, 0.0)) + 0.01 * rng.standard_normal(H * W)
    xyz = dirs * r[:, None]
    scan = np.concatenate([xyz, np.zeros((H * W, 1))], axis=1).astype(np.float32).reshape(H, W, 4)
    scan[rng.random((H, W)) < 0.02, 0] = np.nan
    # surfels: 0.5 m voxels around random scan points, plane = local wall
    p4 = np.zeros((n_planes, 4)); bmin = np.zeros((n_planes, 3)); bmax = np.zeros((n_planes, 3))
    valid = np.argwhere(~np.isnan(scan[..., 0]))
    walls = np.array([[1, 0, 0, -10.0], [1, 0, 0, 10.0], [0, 1, 0, -7.5], [0, 1, 0, 7.5], [0, 0, 1, -1.5], [0, 0, 1, 1.5]])
    for i in range(n_planes):
        h, w = valid[rng.integers(len(valid))]
        c = scan[h, w, :3].astype(np.float64)
        d = np.abs(walls[:, :3] @ c - walls[:, 3])
        k = int(np.argmin(d))
        n_ = walls[k, :3] * (1 if rng.random() < 0.5 else -1)
        off = -n_ @ (walls[k, :3] * walls[k, 3])
        p4[i] = [n_[0], n_[1], n_[2], off + 0.005 * rng.standard_normal()]
        lo = np.floor(c / 1.0) * 1.0
        bmin[i] = lo + rng.uniform(0, 0.05, 3); bmax[i] = lo + 1.0 - rng.uniform(0, 0.05, 3)
    return scan, p4, bmin, bmax


def make_sequence(seed=50, duration=6.0, dt=0.02, imu_rate=400.0, scan_rate=10.0, H=16, W=450, n_reproj=3000, obs_per_frame=50, pad=0.3, t_start=100.0,
                  range_noise=0.01, lidar_err_deg=1.0, lidar_err_m=0.02, cam_err_deg=2.0, cam_err_m=0.03, cp_noise=(2e-3, 2e-3)):
    """A recorded-sequence stand-in for the offline calibration driver (lvx_host::Calibrator): IMU stream, organised LiDAR scans of a box room with
    pillars — every point ray-cast from the pose AT ITS OWN TIMESTAMP (a moving, rolling LiDAR) — and ORB-like visual tracks, all consistent with one
    ground-truth state.  state0: trajectory slightly off (it must be good enough to de-skew, as after the reference's LOAM / NDT initialisation),
    extrinsics off by lidar_err / cam_err."""
    rng = np.random.default_rng(seed)
    cam = dict(DEFAULT_CAMERA)
    t0 = t_start - pad
    t_end = t_start + duration
    n_knots = int(np.ceil((t_end + pad - t0) / dt)) + 3
    while t0 + (n_knots - 3) * dt < t_end + pad:
        n_knots += 1
    r3, so3 = make_trajectory(n_knots, t0, dt, rng, pos_amp=1.2, rot_amp=0.5)
    r3[:, 2] *= 0.25                                                     # stay between floor and ceiling
    sp = Spline(t0, dt, r3, so3)
    q_LI = q_from_rpy(np.deg2rad(2.0), np.deg2rad(-3.0), np.deg2rad(91.0)); p_LI = np.array([0.05, -0.10, 0.12])
    q_CI = qmul(q_from_rpy(np.deg2rad(-90.0), 0.0, np.deg2rad(-90.0)), q_from_rotvec(np.deg2rad([2.0, 0, 0]))); p_CI = np.array([-0.22, 0.02, 0.22])
    roll, pitch = 0.02, -0.015
    ba = np.array([0.05, 0.02, -0.03]); bg = np.array([0.01, -0.02, 0.005])
    g = gravity_vec(roll, pitch)
    n_imu = int(round(duration * imu_rate))
    t_imu = t_start + (np.arange(n_imu) + 0.37) / imu_rate
    e = sp.eval(t_imu)
    gyro = qrot(qconj(e["quat"]), e["angvel"]) + bg + 1.745e-4 * np.sqrt(imu_rate) * rng.standard_normal((n_imu, 3))
    acc = qrot(qconj(e["quat"]), e["acc"] + g) + ba + 5.88e-4 * np.sqrt(imu_rate) * rng.standard_normal((n_imu, 3))
    # scans
    t_map = t_start + 0.05
    n_scans = int(np.floor((duration - 0.3) * scan_rate))
    az = np.deg2rad(np.arange(W) * (360.0 / W)); el = np.deg2rad(np.linspace(-15.0, 15.0, H))
    E, A = np.meshgrid(el, az, indexing="ij")
    dirs_L = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], axis=-1).reshape(-1, 3)     # [H * W], index h * W + w
    col_dt = np.tile(np.arange(W) / W / scan_rate, H)
    scans = np.zeros((n_scans, H * W), dtype=np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad", "<f4"), ("intensity", "<f4"), ("pad2", "<f4"), ("timestamp", "<f8")]))
    for s_ in range(n_scans):
        ts = t_start + 0.1 + s_ / scan_rate + col_dt
        ek = sp.eval(ts)
        org = qrot(ek["quat"], np.broadcast_to(p_LI, (len(ts), 3))) + ek["pos"]
        dW = qrot(ek["quat"], qrot(np.broadcast_to(q_LI, (len(ts), 4)), dirs_L))
        rr = _raycast_room(dW, org) + range_noise * rng.standard_normal(len(ts))
        xyz = (dirs_L * rr[:, None]).astype(np.float32)
        scans["x"][s_], scans["y"][s_], scans["z"][s_], scans["timestamp"][s_] = xyz[:, 0], xyz[:, 1], xyz[:, 2], ts
        drop = rng.random(len(ts)) < 0.01
        scans["x"][s_][drop] = np.nan
    V = _bench_visual(sp, cam, rng, t_start, t_end, n_reproj, q_CI, p_CI, "orb", 10, 20.0, obs_per_frame)
    rho = V["rho"]
    state_true = pack_state(r3, so3, imu_block(roll, pitch, ba, bg), sensor_block(q_LI, p_LI), sensor_block(q_CI, p_CI), rho)
    r3p = r3 + cp_noise[0] * rng.standard_normal(r3.shape)
    so3p = qmul(q_from_rotvec(cp_noise[1] * rng.standard_normal((n_knots, 3))), so3); so3p /= np.linalg.norm(so3p, axis=1, keepdims=True)
    ax = np.array([0.6, -0.5, 0.62]); ax /= np.linalg.norm(ax)
    dqL = q_from_rotvec(np.deg2rad(lidar_err_deg) * ax); dqC = q_from_rotvec(np.deg2rad(cam_err_deg) * np.array([-0.4, 0.7, 0.59]) / np.linalg.norm([-0.4, 0.7, 0.59]))
    dL = lidar_err_m * np.array([1, -1, 1]) / np.sqrt(3); dC = cam_err_m * np.array([-1, 1, 1]) / np.sqrt(3)
    state0 = pack_state(r3p, so3p, imu_block(0.01, 0.01), sensor_block(qmul(dqL, q_LI), p_LI + dL), sensor_block(qmul(dqC, q_CI), p_CI + dC), rho * (1.0 + 0.05 * rng.standard_normal(len(rho))))
    return dict(t0=t0, dt=dt, n_knots=n_knots, camera=cam, t_imu=t_imu, gyro=gyro, acc=acc, H=H, W=W, scans=scans, t_map=t_map,
                n_landmarks=V["n_landmarks"], lm_uv=V["lm_uv"], lm_t0=V["lm_t0"], rep_lm=V["rep_lm"], rep_uv=V["rep_uv"], rep_t0=V["rep_t0"],
                state_true=state_true, state0=state0, t_start=t_start, t_end=t_end)


def sequence_loam_poses(S, noise_m=0.0, noise_rad=0.0, seed=0):
    """What LOAM hands the reference for a recorded sequence (ReadPoseGT, lvi_initialize_surfel_orb.cpp:458-516): one LiDAR pose per scan, stamped with the scan's header
    stamp, in the frame of the LiDAR at the map time — from the ground-truth trajectory of make_sequence(), optionally perturbed.  Returns (scan_t [n], stamp_ns [n] int64,
    p [n, 3], q_wxyz [n, 4], T [n, 16] row-major scan -> map as Eigen builds it from the written numbers)."""
    N = S["n_knots"]
    u = unpack_state(S["state_true"], N, S["n_landmarks"])
    sp = Spline(S["t0"], S["dt"], u["r3"], u["so3"])
    lidar = u["lidar"]
    q_LI, p_LI = lidar[:4], lidar[4:7]
    scan_t = np.array([sc["timestamp"][0] for sc in S["scans"]], dtype=np.float64)
    stamp = (scan_t * 1e9).astype(np.int64)

    def lidar_pose(t):
        e = sp.eval(t)
        return qmul(e["quat"], np.broadcast_to(q_LI, e["quat"].shape)), qrot(e["quat"], np.broadcast_to(p_LI, e["pos"].shape)) + e["pos"]
    q0, p0 = lidar_pose(np.array([S["t_map"]]))
    qk, pk = lidar_pose(scan_t)
    q = qmul(np.broadcast_to(qconj(q0[0]), qk.shape), qk)
    p = qrot(np.broadcast_to(qconj(q0[0]), qk.shape), pk - p0[0])
    if noise_m or noise_rad:
        rng = np.random.default_rng(seed)
        q = qmul(q_from_rotvec(noise_rad * rng.standard_normal((len(q), 3))), q)
        p = p + noise_m * rng.standard_normal(p.shape)
    q_wxyz = np.stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]], axis=1)
    # the file holds '%.9f'-style decimals; T is built from what a reader gets back (write_loam_pose_file writes repr-exact doubles, so this is the identity here)
    T = np.zeros((len(q), 16))
    for i in range(len(q)):
        w, x, y, z = q_wxyz[i]
        tx, ty, tz = 2 * x, 2 * y, 2 * z
        twx, twy, twz, txx, txy, txz, tyy, tyz, tzz = tx * w, ty * w, tz * w, tx * x, ty * x, tz * x, ty * y, tz * y, tz * z
        R = np.array([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]])
        M = np.eye(4); M[:3, :3] = R; M[:3, 3] = p[i]
        T[i] = M.ravel()
    return scan_t, stamp, p, q_wxyz, T


def write_loam_pose_file(path, stamp_ns, p, q_wxyz):
    """A-LOAM laserMapping's pose dump as ReadPoseGT reads it: `stamp_ns x y z qw qx qy qz`, one pose per line (repr-exact doubles: atof gives the same bits back)."""
    with open(path, "w") as f:
        for s, pp, qq in zip(stamp_ns, p, q_wxyz):
            f.write("%d %s %s\n" % (int(s), " ".join(repr(float(v)) for v in pp), " ".join(repr(float(v)) for v in qq)))
