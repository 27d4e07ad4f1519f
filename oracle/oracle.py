"""ctypes binding of the CPU oracle (oracle/liblvx_oracle.so).

TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (see oracle/orc_core.hpp).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FAM_GYRO, FAM_ACCEL, FAM_PRIOR, FAM_SURFEL, FAM_REPROJ, FAM_CAMSURF = range(6)

LOCK_TRAJ = 1 << 0
LOCK_R3 = 1 << 1
LOCK_LIDAR_Q = 1 << 2
LOCK_LIDAR_P = 1 << 3
LOCK_LIDAR_TAU = 1 << 4
LOCK_CAM_Q = 1 << 5
LOCK_CAM_P = 1 << 6
LOCK_CAM_TAU = 1 << 7
LOCK_ACC_BIAS = 1 << 8
LOCK_GYRO_BIAS = 1 << 9
LOCK_LANDMARKS = 1 << 10


def build(force=False):
    so = os.path.join(_HERE, "liblvx_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_create.restype = C.c_void_p
        for name in ("orc_state_size", "orc_tangent_size", "orc_num_residuals", "orc_num_blocks", "orc_max_cols"):
            getattr(_LIB, name).restype = C.c_int
    return _LIB


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Flat problem description -> residuals / Jacobians / normal equations on the CPU."""

    def __init__(self):
        self._l = lib()
        self._h = C.c_void_p(self._l.orc_create())
        self._keep = []
        self.n_surfel = self.n_reproj = self.n_camsurf = 0    # blocks per family: which bounded parameter blocks are part of the problem at all (oracle/lm.py)

    def __del__(self):
        try:
            self._l.orc_destroy(self._h)
        except Exception:
            pass

    def set_spline(self, t0, dt, n_knots):
        self._l.orc_set_spline(self._h, C.c_double(t0), C.c_double(dt), C.c_int(n_knots))

    def set_camera(self, rows, cols, readout, fx, fy, cx, cy, k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0):
        self._l.orc_set_camera(self._h, C.c_int(rows), C.c_int(cols), *[C.c_double(v) for v in (readout, fx, fy, cx, cy, k1, k2, p1, p2, k3)])

    def set_imu(self, t, gyro, acc, w_g, w_a):
        t, gyro, acc = _d(t), _d(gyro), _d(acc)
        self._l.orc_set_imu(self._h, C.c_int(len(t)), _p(t), _p(gyro), _p(acc), C.c_double(w_g), C.c_double(w_a))

    def set_orientation_prior(self, t, q_wxyz, w, enable=True):
        q = _d(q_wxyz)
        self._l.orc_set_orientation_prior(self._h, C.c_int(1 if enable else 0), C.c_double(t), _p(q), C.c_double(w))

    def set_planes(self, pi3):
        pi3 = _d(pi3)
        self._l.orc_set_planes(self._h, C.c_int(len(pi3)), _p(pi3))

    def set_surfel(self, pt, t, plane_id, t_map, huber, w):
        pt, t, plane_id = _d(pt), _d(t), _i(plane_id)
        self.n_surfel = len(t)
        self._l.orc_set_surfel(self._h, C.c_int(len(t)), _p(pt), _p(t), _p(plane_id), C.c_double(t_map), C.c_double(huber), C.c_double(w))

    def set_landmarks(self, uv_ref, t0_ref):
        uv_ref, t0_ref = _d(uv_ref), _d(t0_ref)
        self.n_landmarks = len(t0_ref)
        self._l.orc_set_landmarks(self._h, C.c_int(len(t0_ref)), _p(uv_ref), _p(t0_ref))

    def set_reproj(self, lm, uv_obs, t0_obs, huber, w):
        lm, uv_obs, t0_obs = _i(lm), _d(uv_obs), _d(t0_obs)
        self.n_reproj = len(lm)
        self._l.orc_set_reproj(self._h, C.c_int(len(lm)), _p(lm), _p(uv_obs), _p(t0_obs), C.c_double(huber), C.c_double(w))

    def set_camsurf(self, lm, plane_id, t_map, huber, w):
        lm, plane_id = _i(lm), _i(plane_id)
        self.n_camsurf = len(lm)
        self._l.orc_set_camsurf(self._h, C.c_int(len(lm)), _p(lm), _p(plane_id), C.c_double(t_map), C.c_double(huber), C.c_double(w))

    def set_locks(self, mask):
        self._l.orc_set_locks(self._h, C.c_uint32(mask))

    def set_so3_only(self, flag):
        self._l.orc_set_so3_only(self._h, C.c_int(1 if flag else 0))

    def set_threads(self, n):
        self._l.orc_set_threads(self._h, C.c_int(n))

    def set_block_products(self, on):
        """evaluate_products also forms every block's dense J^T J upper triangle (CPU-baseline timing; folded into products_checksum)."""
        self._l.orc_set_block_products(self._h, C.c_int(1 if on else 0))

    @property
    def products_checksum(self):
        self._l.orc_products_checksum.restype = C.c_double
        return self._l.orc_products_checksum(self._h)

    @property
    def state_size(self):
        return self._l.orc_state_size(self._h)

    @property
    def tangent_size(self):
        return self._l.orc_tangent_size(self._h)

    @property
    def num_residuals(self):
        return self._l.orc_num_residuals(self._h)

    @property
    def num_blocks(self):
        return self._l.orc_num_blocks(self._h)

    def evaluate(self, state, jac=False, normal_eq=False):
        """Returns dict(cost, residuals[, jac_cols, jac_vals][, H, g]); raises on range / non-unit errors."""
        state = _d(state)
        assert state.size == self.state_size
        nr = self.num_residuals
        mc = self._l.orc_max_cols()
        cost = C.c_double(0)
        res = np.zeros(nr)
        jc = np.full((nr, mc), -1, dtype=np.int32) if jac else None
        jv = np.zeros((nr, mc)) if jac else None
        nt = self.tangent_size
        H = np.zeros((nt, nt)) if normal_eq else None
        g = np.zeros(nt) if normal_eq else None
        rc = self._l.orc_evaluate(self._h, _p(state), C.byref(cost), _p(res), _p(jc), _p(jv), _p(H), _p(g))
        if rc == -1:
            raise IndexError("oracle: time span out of range for trajectory (std::range_error)")
        if rc == -2:
            raise ValueError("oracle: logq of a non-unit quaternion (std::runtime_error)")
        out = {"cost": cost.value, "residuals": res}
        if jac:
            out["jac_cols"], out["jac_vals"] = jc, jv
        if normal_eq:
            out["H"], out["g"] = H, g
        return out

    def evaluate_products(self, state, V=None):
        """Matrix-free pass for problems too large for the dense H: dict(cost, residuals, g = J^T r, diag = diag(J^T J)[, HV = J^T J V]) with the
        robustified J; V: [k, n_tangent] or None.  OpenMP over the blocks (set_threads)."""
        state = _d(state)
        assert state.size == self.state_size
        nt, nr = self.tangent_size, self.num_residuals
        V = None if V is None else _d(np.atleast_2d(V))
        k = 0 if V is None else V.shape[0]
        assert V is None or V.shape[1] == nt
        cost = C.c_double(0)
        res, g, diag = np.zeros(nr), np.zeros(nt), np.zeros(nt)
        HV = np.zeros((k, nt)) if k else None
        rc = self._l.orc_evaluate_products(self._h, _p(state), C.c_int(k), _p(V), C.byref(cost), _p(res), _p(g), _p(diag), _p(HV))
        if rc == -1:
            raise IndexError("oracle: time span out of range for trajectory (std::range_error)")
        if rc == -2:
            raise ValueError("oracle: logq of a non-unit quaternion (std::runtime_error)")
        assert rc == 0
        return {"cost": cost.value, "residuals": res, "g": g, "diag": diag, "HV": HV}

    def jacobian_csr(self, state):
        """The robustified Jacobian of every residual block as one scipy CSR matrix in tangent coordinates (rows family-major like `residuals`; only free scalars
        have entries) + the robustified residuals that go with it: dict(cost, residuals (raw), r (sqrt(rho') r), J).  OpenMP over the blocks (set_threads)."""
        import scipy.sparse as sp
        state = _d(state)
        assert state.size == self.state_size
        nt, nr = self.tangent_size, self.num_residuals
        cost = C.c_double(0)
        res, rs = np.zeros(nr), np.zeros(nr)
        ptr = np.zeros(nr + 1, np.int64)
        pc, pv = C.c_void_p(), C.c_void_p()
        rc = self._l.orc_jacobian_csr(self._h, _p(state), C.byref(cost), _p(res), _p(rs), _p(ptr), C.byref(pc), C.byref(pv))
        if rc == -1:
            raise IndexError("oracle: time span out of range for trajectory (std::range_error)")
        if rc == -2:
            raise ValueError("oracle: logq of a non-unit quaternion (std::runtime_error)")
        assert rc == 0
        nnz = int(ptr[-1])
        try:
            cols = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_int32)), shape=(max(nnz, 1),))[:nnz].copy()
            vals = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_double)), shape=(max(nnz, 1),))[:nnz].copy()
        finally:
            self._l.orc_free(pc); self._l.orc_free(pv)
        J = sp.csr_matrix((vals, cols, ptr), shape=(nr, nt))
        return {"cost": cost.value, "residuals": res, "r": rs, "J": J}

    def plus(self, state, delta):
        state, delta = _d(state), _d(delta)
        out = np.zeros_like(state)
        self._l.orc_plus(self._h, _p(state), _p(delta), _p(out))
        return out

    def eval_pose(self, state, t):
        state, t = _d(state), _d(np.atleast_1d(t))
        n = len(t)
        pos, quat, vel, acc, w = np.zeros((n, 3)), np.zeros((n, 4)), np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
        rc = self._l.orc_eval_pose(self._h, _p(state), C.c_int(n), _p(t), _p(pos), _p(quat), _p(vel), _p(acc), _p(w))
        if rc:
            raise IndexError("oracle: pose evaluation out of range")
        return {"pos": pos, "quat": quat, "vel": vel, "acc": acc, "angvel": w}


def ata_lower(A, threads=0):
    """Lower triangle of A^T A of a scipy CSR matrix as CSC (oracle/orc_sparse.cpp::orc_ata_lower, OpenMP over the columns); duplicates inside a row add."""
    import scipy.sparse as sp
    A = A.tocsr()
    n_rows, n_cols = A.shape
    ptr = np.ascontiguousarray(A.indptr, np.int64)
    cols = np.ascontiguousarray(A.indices, np.int32)
    vals = _d(A.data)
    cptr = np.zeros(n_cols + 1, np.int64)
    pr, pv = C.c_void_p(), C.c_void_p()
    rc = lib().orc_ata_lower(C.c_int64(n_rows), C.c_int32(n_cols), _p(ptr), _p(cols), _p(vals), C.c_int(threads), _p(cptr), C.byref(pr), C.byref(pv))
    assert rc == 0
    nnz = int(cptr[-1])
    try:
        rows = np.ctypeslib.as_array(C.cast(pr, C.POINTER(C.c_int32)), shape=(max(nnz, 1),))[:nnz].copy()
        v = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_double)), shape=(max(nnz, 1),))[:nnz].copy()
    finally:
        lib().orc_free(pr); lib().orc_free(pv)
    return sp.csc_matrix((v, rows, cptr), shape=(n_cols, n_cols))


def analytic_pass(o, state, threads=0):
    """CPU baseline mode (ii): one OpenMP pass over every gyro / accel / surfel / reprojection block with closed-form Jacobians and the block's J^T J / J^T r
    products (oracle/orc_analytic.cpp).  Returns (blocks, cost)."""
    state = _d(state)
    cost, chk = C.c_double(0), C.c_double(0)
    lib().orc_analytic_pass.restype = C.c_longlong
    n = lib().orc_analytic_pass(o._h, _p(state), C.c_int(threads), C.byref(cost), C.byref(chk))
    if n < 0:
        raise RuntimeError("orc_analytic_pass: residual error code %d" % (-n))
    return int(n), cost.value


POINT_XYZIT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad", "<f4"), ("intensity", "<f4"), ("pad2", "<f4"), ("timestamp", "<f8")])   # pcl_utils.h:39-44


def eval_lidar_pose(o, state, t):
    state, t = _d(state), _d(np.atleast_1d(t))
    n = len(t)
    q, p, ok = np.zeros((n, 4)), np.zeros((n, 3)), np.zeros(n, np.int32)
    lib().orc_eval_lidar_pose(o._h, _p(state), C.c_int(n), _p(t), _p(q), _p(p), _p(ok))
    return q, p, ok.astype(bool)


def undistort(o, state, raw, q_G_to_target, p_target_in_G, correct_position=True):
    raw = np.ascontiguousarray(raw, dtype=POINT_XYZIT)
    out = np.zeros((len(raw), 4), np.float32)
    rc = lib().orc_undistort(o._h, _p(_d(state)), C.c_int(len(raw)), _p(raw), _p(_d(q_G_to_target)), _p(_d(p_target_in_G)), C.c_int(1 if correct_position else 0), _p(out))
    assert rc == 0
    return out


def surfel_assoc_omp(scan_hw4, p4, box_min, box_max, radius=0.05, sel=2, threads=1):
    """surfel_assoc with the reference's OpenMP loop over planes (CPU timing)."""
    scan = np.ascontiguousarray(scan_hw4, dtype=np.float32)
    H, W = scan.shape[0], scan.shape[1]
    p4, box_min, box_max = _d(p4), _d(box_min), _d(box_max)
    flag = np.full(H * W, -1, np.int32)
    lib().orc_surfel_assoc_omp(C.c_int(H), C.c_int(W), _p(scan), C.c_int(len(p4)), _p(p4), _p(box_min), _p(box_max), C.c_double(radius), C.c_int(sel), _p(flag), C.c_int(threads))
    return flag.reshape(H, W)


def surfel_emit(flag_hw, scan_map_hw4, raw_hw):
    """SurfelPoint list of one associated scan (chronological order): dict of pt, pt_map, t, plane."""
    flag = np.ascontiguousarray(flag_hw, dtype=np.int32)
    H, W = flag.shape
    scan = np.ascontiguousarray(scan_map_hw4, dtype=np.float32)
    raw = np.ascontiguousarray(raw_hw, dtype=POINT_XYZIT)
    n = H * W
    pt, pm, ts, pl = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n), np.zeros(n, np.int32)
    k = lib().orc_surfel_emit(C.c_int(H), C.c_int(W), _p(flag), _p(scan), _p(raw), _p(pt), _p(pm), _p(ts), _p(pl))
    return dict(pt=pt[:k], pt_map=pm[:k], t=ts[:k], plane=pl[:k])


def landmark_assoc(o, state, q_LtoC_xyzw, t_LinC, map_time, p4, box_min, box_max, radius=0.05):
    """associateVisualPointsWithPlanes: surfel index per landmark of the oracle's problem (or -1)."""
    p4, box_min, box_max = _d(p4), _d(box_min), _d(box_max)
    out = np.full(max(o.n_landmarks, 1), -1, np.int32)
    rc = lib().orc_landmark_assoc(o._h, _p(_d(state)), _p(_d(q_LtoC_xyzw)), _p(_d(t_LinC)), C.c_double(map_time), C.c_int(len(p4)), _p(p4), _p(box_min), _p(box_max), C.c_double(radius), _p(out))
    assert rc == 0
    return out[:o.n_landmarks]


def dense_jacobian(jac_cols, jac_vals, n_tangent):
    """Scatter the fixed-width (cols, vals) rows into a dense (rows, n_tangent) matrix."""
    nr = jac_cols.shape[0]
    J = np.zeros((nr, n_tangent))
    rows = np.repeat(np.arange(nr), jac_cols.shape[1])
    c = jac_cols.ravel()
    m = c >= 0
    np.add.at(J, (rows[m], c[m]), jac_vals.ravel()[m])
    return J


# ---------------------------------------------------------------------------------------------------------
# upstream kernels (oracle/orc_upstream.cpp)
# ---------------------------------------------------------------------------------------------------------
RS_POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad", "<f4"), ("intensity", "u1"), ("pad2", "u1"), ("ring", "<u2"),
                     ("pad3", "<u4"), ("timestamp", "<f8")])   # RsPointXYZIRT, scanRegistration.cpp:57-66 (32 bytes)
assert RS_POINT.itemsize == 32


def scan_register(pts, n_rings, min_range):
    pts = np.ascontiguousarray(pts, dtype=RS_POINT)
    n = len(pts)
    out = dict(cloud=np.zeros((n, 4), np.float32), curvature=np.zeros(n, np.float32), label=np.zeros(n, np.int32), sort_ind=np.zeros(n, np.int32),
               picked=np.zeros(n, np.int32), scan_start=np.zeros(n_rings, np.int32), scan_end=np.zeros(n_rings, np.int32))
    lists = [np.zeros(max(n, 1), np.int32) for _ in range(4)]
    counts = np.zeros(4, np.int32)
    l = lib()
    l.orc_scan_register.restype = C.c_int
    m = l.orc_scan_register(C.c_int(n), _p(pts), C.c_int(n_rings), C.c_float(min_range), _p(out["cloud"]), _p(out["curvature"]), _p(out["label"]),
                            _p(out["sort_ind"]), _p(out["picked"]), _p(out["scan_start"]), _p(out["scan_end"]), *[_p(a) for a in lists], _p(counts))
    for k in ("cloud", "curvature", "label", "sort_ind", "picked"):
        out[k] = out[k][:m]
    out["n"] = m
    for name, a, c in zip(("sharp", "less_sharp", "flat", "less_flat"), lists, counts):
        out[name] = a[:c].copy()
    return out


def voxel_build(xyzi, leaf, min_pts=6, eig_mult=0.01):
    xyzi = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    n = len(xyzi)
    cap = max(n, 1)
    grid = np.zeros(12, np.int32)
    o = dict(leaf_key=np.zeros(cap, np.int32), leaf_n=np.zeros(cap, np.int32), mean=np.zeros((cap, 3)), cov=np.zeros((cap, 9)), icov=np.zeros((cap, 9)),
             evecs=np.zeros((cap, 9)), evals=np.zeros((cap, 3)), centroid=np.zeros((cap, 3), np.float32), offsets=np.zeros(cap + 1, np.int32),
             point_ids=np.zeros(cap, np.int32))
    l = lib()
    l.orc_voxel_build.restype = C.c_int
    nl = l.orc_voxel_build(C.c_int(n), _p(xyzi), C.c_float(leaf), C.c_int(min_pts), C.c_double(eig_mult), _p(grid), _p(o["leaf_key"]), _p(o["leaf_n"]),
                           _p(o["mean"]), _p(o["cov"]), _p(o["icov"]), _p(o["evecs"]), _p(o["evals"]), _p(o["centroid"]), _p(o["offsets"]), _p(o["point_ids"]), C.c_int(cap))
    assert nl >= 0
    for k in ("leaf_key", "leaf_n", "mean", "cov", "icov", "evecs", "evals", "centroid"):
        o[k] = o[k][:nl]
    o["offsets"] = o["offsets"][:nl + 1]
    o["grid"] = grid
    o["n_leaves"] = nl
    return o


def surfel_extract(xyzi, vox, p_lambda=0.7, dist_threshold=0.05, min_leaf_points=10, min_inliers=20):
    xyzi = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    nl = vox["n_leaves"]
    planes = np.zeros((max(nl, 1), 16))
    types = np.zeros(max(nl, 1), np.int32)
    l = lib()
    l.orc_surfel_extract.restype = C.c_int
    arrs = [np.ascontiguousarray(vox[k]) for k in ("leaf_n", "offsets", "point_ids", "mean", "evecs", "evals")]
    n = l.orc_surfel_extract(C.c_int(nl), _p(xyzi), *[_p(a) for a in arrs], C.c_double(p_lambda), C.c_double(dist_threshold), C.c_int(min_leaf_points),
                             C.c_int(min_inliers), _p(planes), _p(types), C.c_int(len(planes)))
    P = planes[:n]
    return dict(p4=P[:, 0:4].copy(), Pi=P[:, 4:7].copy(), box_min=P[:, 7:10].copy(), box_max=P[:, 10:13].copy(), leaf=P[:, 13].astype(np.int32),
                n_points=P[:, 14].astype(np.int32), n_inliers=P[:, 15].astype(np.int32), plane_type=types[:n].copy())


def ndt_derivatives(vox, leaf, src, trans, p6, outlier_ratio=0.55, compute_hessian=True, min_pts=6):
    src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1, 4)
    trans = np.ascontiguousarray(trans, dtype=np.float32).reshape(-1, 4)
    ids = voxel_lookup7(vox, trans, leaf, min_pts)
    score = C.c_double(0)
    g, H = np.zeros(6), np.zeros((6, 6))
    lib().orc_ndt_derivatives(C.c_int(len(src)), _p(src), _p(trans), _p(ids), _p(np.ascontiguousarray(vox["mean"])), _p(np.ascontiguousarray(vox["icov"])), _p(_d(p6)),
                              C.c_double(leaf), C.c_double(outlier_ratio), C.c_int(1 if compute_hessian else 0), C.byref(score), _p(g), _p(H))
    return score.value, g, H


def voxelgrid_xyzi(xyzi, leaf=0.2):
    xyzi = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    out = np.zeros((max(len(xyzi), 1), 4), np.float32)
    l = lib()
    l.orc_voxelgrid_xyzi.restype = C.c_int
    n = l.orc_voxelgrid_xyzi(C.c_int(len(xyzi)), _p(xyzi), C.c_float(leaf), _p(out))
    return out[:n]


def voxel_lookup7(vox, queries, leaf, min_pts=6):
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 4)
    ids = np.full((len(q), 7), -1, np.int32)
    lk, ln = np.ascontiguousarray(vox["leaf_key"]), np.ascontiguousarray(vox["leaf_n"])
    lib().orc_voxel_lookup7(C.c_int(len(q)), _p(q), C.c_float(leaf), C.c_int(min_pts), _p(vox["grid"]), C.c_int(vox["n_leaves"]), _p(lk), _p(ln), _p(ids))
    return ids


def surfel_assoc(scan_hw4, p4, box_min, box_max, radius=0.05, sel=2):
    scan = np.ascontiguousarray(scan_hw4, dtype=np.float32)
    H, W = scan.shape[0], scan.shape[1]
    p4, box_min, box_max = _d(p4), _d(box_min), _d(box_max)
    flag = np.full(H * W, -1, np.int32)
    lib().orc_surfel_assoc(C.c_int(H), C.c_int(W), _p(scan), C.c_int(len(p4)), _p(p4), _p(box_min), _p(box_max), C.c_double(radius), C.c_int(sel), _p(flag))
    return flag.reshape(H, W)
