"""ctypes binding of the C ABI in include/lvx.h (liblvx.so, HIP / gfx950).

Plumbing only: the product is the shared library.  There is NO CPU fallback — creating a context
without a HIP device raises, as does a missing liblvx.so.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, E_RANGE, E_NONUNIT_QUAT, E_ALLOC, E_HIP, E_RCCL, E_ARG, E_STATE, E_NODEVICE, E_NOTPD = 0, -1, -2, -3, -4, -5, -6, -7, -8, -9

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

EVAL_COST, EVAL_RESIDUALS, EVAL_NORMAL_EQ, EVAL_JACOBIAN = 1, 2, 4, 8
(FAM_GYRO, FAM_ACCEL, FAM_PRIOR, FAM_SURFEL, FAM_REPROJ, FAM_CAMSURF, KERNEL_FOLD, KERNEL_SOLVE, KERNEL_UPSTREAM, KERNEL_CLEAR, KERNEL_REP_JAC, KERNEL_REP_OBS,
 KERNEL_REP_REF, KERNEL_REP_CROSS, KERNEL_REP_LMROWS, KERNEL_REP_FUSED, KERNEL_FIXUP) = range(17)
KERNEL_NAMES = ["gyro", "accel", "prior", "surfel", "reproj", "camsurf", "fold", "solve", "upstream", "clear", "reproj_jac", "reproj_obs", "reproj_ref", "reproj_cross", "reproj_lmrows", "reproj_fused", "fixup"]
JAC_WIDTH = 64


class Pinhole(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("readout", C.c_double), ("fx", C.c_double), ("fy", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("k1", C.c_double), ("k2", C.c_double), ("p1", C.c_double),
                ("p2", C.c_double), ("k3", C.c_double)]


class Layout(C.Structure):
    _fields_ = [("n_knots", C.c_int32), ("n_landmarks", C.c_int32), ("n_tangent", C.c_int32), ("n_band", C.c_int32),
                ("bandwidth", C.c_int32), ("n_border", C.c_int32), ("border_ld", C.c_int32), ("n_hub_knots", C.c_int32), ("hub_knot0", C.c_int32),
                ("n_blocks", C.c_int64), ("n_residuals", C.c_int64), ("exact_fallback", C.c_int32), ("solver_fallbacks", C.c_int32), ("fallback_rows", C.c_int32), ("solver_separators", C.c_int32), ("solver_leaves", C.c_int32)]


class LmOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("jacobi_scaling", C.c_int32), ("verbose", C.c_int32)]


class LmSummary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("successful_steps", C.c_int32), ("termination", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("final_radius", C.c_double)]


LM_TERMINATION = ["no_convergence", "function_tolerance", "parameter_tolerance", "gradient_tolerance", "max_iterations", "failure", "min_trust_region_radius"]


class LvxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lvx error %d: %s" % (code, msg))
        self.code = code


def library_path():
    # LVX_LIB: an instrumented build of the same sources (tools/build_kt.sh: per-phase cycle counters), never another implementation
    return os.environ.get("LVX_LIB") or os.path.join(_HERE, "liblvx.so")


def lib():
    """Load liblvx.so; raises if it has not been built (python lvi-exc_amd/build.py)."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise FileNotFoundError("liblvx.so is missing: build it with `python lvi-exc_amd/build.py` (hipcc, gfx950)")
        _LIB = C.CDLL(path)
        _LIB.lvx_version.restype = C.c_char_p
        _LIB.lvx_last_error.restype = C.c_char_p
        _LIB.lvx_last_error.argtypes = [C.c_void_p]
    return _LIB


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One evaluator context bound to one GPU (thin wrapper over lvx_ctx)."""

    def __init__(self, device=0):
        self._l = lib()
        h = C.c_void_p()
        rc = self._l.lvx_create(C.byref(h), C.c_int(device), C.c_uint32(0))
        if rc != OK:
            raise LvxError(rc, "lvx_create failed (no HIP device?)" if rc == E_NODEVICE else "lvx_create failed")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._l.lvx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != OK:
            raise LvxError(rc, self._l.lvx_last_error(self._h).decode())

    # --- problem description (same argument meaning as oracle.Oracle) ---
    def set_spline(self, t0, dt, n_knots):
        self._ck(self._l.lvx_set_spline(self._h, C.c_double(t0), C.c_double(dt), C.c_int(n_knots)))

    def set_camera(self, rows, cols, readout, fx, fy, cx, cy, k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0):
        ph = Pinhole(rows, cols, readout, fx, fy, cx, cy, k1, k2, p1, p2, k3)
        self._ck(self._l.lvx_set_camera(self._h, C.byref(ph)))

    def set_imu(self, t, gyro, acc, w_g, w_a):
        t, gyro, acc = _d(t), _d(gyro), _d(acc)
        self._ck(self._l.lvx_set_imu(self._h, C.c_int(len(t)), _p(t), _p(gyro), _p(acc), C.c_double(w_g), C.c_double(w_a)))

    def set_orientation_prior(self, t, q_wxyz, w, enable=True):
        q = _d(q_wxyz)
        self._ck(self._l.lvx_set_orientation_prior(self._h, C.c_int(1 if enable else 0), C.c_double(t), _p(q), C.c_double(w)))

    def set_planes(self, pi3):
        pi3 = _d(pi3)
        self._ck(self._l.lvx_set_planes(self._h, C.c_int(len(pi3)), _p(pi3)))

    def set_surfel(self, pt, t, plane_id, t_map, huber, w):
        pt, t, plane_id = _d(pt), _d(t), _i(plane_id)
        self._ck(self._l.lvx_set_surfel(self._h, C.c_int(len(t)), _p(pt), _p(t), _p(plane_id), C.c_double(t_map), C.c_double(huber), C.c_double(w)))

    def set_landmarks(self, uv_ref, t0_ref):
        uv_ref, t0_ref = _d(uv_ref), _d(t0_ref)
        self._ck(self._l.lvx_set_landmarks(self._h, C.c_int(len(t0_ref)), _p(uv_ref), _p(t0_ref)))

    def set_reproj(self, lm, uv_obs, t0_obs, huber, w):
        lm, uv_obs, t0_obs = _i(lm), _d(uv_obs), _d(t0_obs)
        self._ck(self._l.lvx_set_reproj(self._h, C.c_int(len(lm)), _p(lm), _p(uv_obs), _p(t0_obs), C.c_double(huber), C.c_double(w)))

    def set_camsurf(self, lm, plane_id, t_map, huber, w):
        lm, plane_id = _i(lm), _i(plane_id)
        self._ck(self._l.lvx_set_camsurf(self._h, C.c_int(len(lm)), _p(lm), _p(plane_id), C.c_double(t_map), C.c_double(huber), C.c_double(w)))

    def set_locks(self, mask):
        self._ck(self._l.lvx_set_locks(self._h, C.c_uint32(mask)))

    def voxel_info(self):
        """Grid geometry and leaf count of the last voxel build (waits for an asynchronous lvx_voxel_build_d)."""
        info = VoxelInfo()
        self._ck(self._l.lvx_voxel_get_info(self._h, C.byref(info)))
        return dict(n_leaves=info.n_leaves, n_points=info.n_points, min_b=list(info.min_b), max_b=list(info.max_b), div_b=list(info.div_b), divb_mul=list(info.divb_mul))

    def set_switch(self, name, value=1):
        """Experiment / debug switch of this context, e.g. set_switch("FORCE_LEGACY", 1)."""
        self._ck(self._l.lvx_set_switch(self._h, C.c_char_p(name.encode()), C.c_int(int(value))))

    def set_so3_only(self, flag):
        """Solve #0 estimator (TrajectoryEstimator<UniformSO3SplineTrajectory>): R3 spline absent, gyro + prior only."""
        self._so3_only = bool(flag)

    @property
    def state_size(self):
        return self._l.lvx_state_size(self._h)

    @property
    def tangent_size(self):
        return self._l.lvx_tangent_size(self._h)

    def layout(self):
        lo = Layout()
        self._ck(self._l.lvx_get_layout(self._h, C.byref(lo)))
        return {k: getattr(lo, k) for k, _ in Layout._fields_}

    def family_rows(self):
        """First residual row of every family (FAM_* order) and the total, as lvx_get_family_rows."""
        r = (C.c_int64 * 7)()
        self._ck(self._l.lvx_get_family_rows(self._h, r))
        return [int(v) for v in r]

    # --- evaluation ---
    def evaluate(self, state, jac=False, normal_eq=False, dense=True, residuals=True):
        state = _d(state)
        assert state.size == self.state_size
        lo = self.layout()
        what = EVAL_COST | (EVAL_RESIDUALS if residuals else 0) | (EVAL_NORMAL_EQ if normal_eq else 0) | (EVAL_JACOBIAN if jac else 0)
        cost = C.c_double(0)
        res = np.zeros(lo["n_residuals"]) if residuals else None
        self._ck(self._l.lvx_evaluate(self._h, _p(state), C.c_uint32(what), C.byref(cost), _p(res)))
        out = {"cost": cost.value, "residuals": res}
        if jac:
            jc = np.full((lo["n_residuals"], JAC_WIDTH), -1, dtype=np.int32)
            jv = np.zeros((lo["n_residuals"], JAC_WIDTH))
            self._ck(self._l.lvx_get_jacobian(self._h, _p(jc), _p(jv)))
            out["jac_cols"], out["jac_vals"] = jc, jv
        if normal_eq and dense:
            nt = self.tangent_size
            H = np.zeros((nt, nt))
            g = np.zeros(nt)
            self._ck(self._l.lvx_get_normal_eq_dense(self._h, _p(H), _p(g)))
            out["H"], out["g"] = H, g
        return out

    def gradient(self):
        """(g = J^T r, diag(J^T J)) of the last normal-equation evaluation in the tangent layout (any problem size)."""
        g, d = np.zeros(self.tangent_size), np.zeros(self.tangent_size)
        self._ck(self._l.lvx_get_gradient(self._h, _p(g), _p(d)))
        return g, d

    def set_state(self, state):
        state = _d(state)
        assert state.size == self.state_size
        self._ck(self._l.lvx_set_state(self._h, _p(state)))

    def get_state(self):
        out = np.zeros(self.state_size)
        self._ck(self._l.lvx_get_state(self._h, _p(out)))
        return out

    def evaluate_resident(self, what=EVAL_COST | EVAL_NORMAL_EQ, want_cost=False):
        """Queue one evaluation of the resident state on the context's stream (asynchronous unless want_cost)."""
        if want_cost:
            cost = C.c_double(0)
            self._ck(self._l.lvx_evaluate_d(self._h, None, C.c_uint32(what), C.byref(cost)))
            return cost.value
        self._ck(self._l.lvx_evaluate_d(self._h, None, C.c_uint32(what), None))
        return None

    def set_stream(self, stream_handle):
        """Use a caller-owned HIP stream (int handle, e.g. torch.cuda.current_stream().cuda_stream); 0/None = own stream."""
        self._ck(self._l.lvx_set_stream(self._h, C.c_void_p(stream_handle or None)))

    def export_border(self, device_ptr):
        self._ck(self._l.lvx_export_border_d(self._h, C.c_void_p(device_ptr)))

    def normal_eq_checksum(self):
        out = (C.c_uint64 * 6)()
        self._ck(self._l.lvx_normal_eq_checksum(self._h, out))
        return tuple(int(v) for v in out)

    def synchronize(self):
        self._ck(self._l.lvx_synchronize(self._h))

    def set_profiling(self, on, only=None):
        """on: time every launch with HIP events; only=<kernel index>: just that kernel's launches (cheap enough for a timed region)."""
        self._ck(self._l.lvx_set_profiling(self._h, C.c_int((2 + int(only)) if (on and only is not None) else (1 if on else 0))))

    def kernel_ms(self):
        ms = np.zeros(len(KERNEL_NAMES))
        n = np.zeros(len(KERNEL_NAMES), dtype=np.int64)
        self._ck(self._l.lvx_get_kernel_ms(self._h, _p(ms), _p(n)))
        return ms, n

    def solve_step(self, radius, jacobi_scaling=True):
        """One damped solve on the normal equations of the last evaluate(normal_eq=True): returns (delta, model_cost_change)."""
        delta = np.zeros(self.tangent_size)
        mcc = C.c_double(0)
        self._ck(self._l.lvx_solve_step(self._h, C.c_double(radius), C.c_int(1 if jacobi_scaling else 0), _p(delta), C.byref(mcc)))
        return delta, mcc.value

    def lm_solve(self, state, max_iterations=50, **kw):
        """TrajectoryEstimator::Solve(max_iterations) equivalent.  Returns (state, summary dict with per-iteration history)."""
        opt = LmOptions()
        self._l.lvx_lm_default_options(C.byref(opt))
        opt.max_iterations = max_iterations
        for k, v in kw.items():
            setattr(opt, k, v)
        x = _d(state).copy()
        sm = LmSummary()
        self._ck(self._l.lvx_lm_solve(self._h, _p(x), C.byref(opt), C.byref(sm)))
        n = 4 * max_iterations + 8
        cost, rad, acc = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int32)
        k = self._l.lvx_lm_get_history(self._h, C.c_int(n), _p(cost), _p(rad), _p(acc))
        out = {f: getattr(sm, f) for f, _ in LmSummary._fields_}
        out["termination"] = LM_TERMINATION[sm.termination]
        out.update(cost_history=cost[:k], radius_history=rad[:k], accepted=acc[:k])
        return x, out

    # ---- joint solve over sequences that share the rig extrinsics (one context / GPU per sequence) ----
    @staticmethod
    def _hook(allreduce):
        """ctypes trampoline for lvx_allreduce_fn: allreduce(vec: np.ndarray[float64], op: 'sum' | 'max') reduces IN PLACE over all ranks."""
        def fn(_user, buf, n, op):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(n,)), "sum" if op == 0 else "max")
                return 0
            except Exception:   # noqa: BLE001 — must not unwind through C
                import traceback
                traceback.print_exc()
                return 1
        return ALLREDUCE_FN(fn)

    def solve_step_shared(self, radius, allreduce, jacobi_scaling=True):
        delta = np.zeros(self.tangent_size)
        mcc = C.c_double(0)
        cb = self._hook(allreduce)
        self._ck(self._l.lvx_solve_step_shared(self._h, C.c_double(radius), C.c_int(1 if jacobi_scaling else 0), cb, None, _p(delta), C.byref(mcc)))
        return delta, mcc.value

    def lm_solve_shared(self, state, allreduce, max_iterations=50, **kw):
        """LM on the JOINT problem of all ranks' sequences; every rank calls this with its own sequence loaded and the same options.
        allreduce = None: the RCCL communicator installed with rccl_init carries the reductions."""
        opt = LmOptions()
        self._l.lvx_lm_default_options(C.byref(opt))
        opt.max_iterations = max_iterations
        for k, v in kw.items():
            setattr(opt, k, v)
        x = _d(state).copy()
        sm = LmSummary()
        cb = self._hook(allreduce) if allreduce is not None else None
        self._ck(self._l.lvx_lm_solve_shared(self._h, _p(x), C.byref(opt), cb, None, C.byref(sm)))
        n = 4 * max_iterations + 8
        cost, rad, acc = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int32)
        k = self._l.lvx_lm_get_history(self._h, C.c_int(n), _p(cost), _p(rad), _p(acc))
        out = {f: getattr(sm, f) for f, _ in LmSummary._fields_}
        out["termination"] = LM_TERMINATION[sm.termination]
        out.update(cost_history=cost[:k], radius_history=rad[:k], accepted=acc[:k])
        return x, out

    # ---- RCCL transport of the joint solve (the reductions run as ncclAllReduce on the context's stream) ----
    def rccl_unique_id(self):
        buf = (C.c_char * 128)()
        self._ck(self._l.lvx_rccl_unique_id(self._h, buf))
        return bytes(buf)

    def rccl_init(self, unique_id, rank, world):
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        self._ck(self._l.lvx_rccl_init(self._h, buf, C.c_int(rank), C.c_int(world)))

    def rccl_allreduce(self, device_ptr, n, op=0):
        """In-place all-reduce of n doubles at device_ptr over the library's own communicator, on the context's stream (op 0 = sum, 1 = max)."""
        self._ck(self._l.lvx_rccl_allreduce_d(self._h, C.c_void_p(device_ptr), C.c_int(n), C.c_int(op)))

    def rccl_finalize(self):
        self._ck(self._l.lvx_rccl_finalize(self._h))

    def joint_shared_count(self):
        return int(self._l.lvx_joint_shared_count(self._h))

    def collective_count(self, reset=False):
        self._l.lvx_collective_count.restype = C.c_int64
        return int(self._l.lvx_collective_count(self._h, C.c_int(1 if reset else 0)))

    def plus(self, state, delta):
        state, delta = _d(state), _d(delta)
        out = np.zeros_like(state)
        self._ck(self._l.lvx_plus(self._h, _p(state), _p(delta), _p(out)))
        return out


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)


RS_POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad", "<f4"), ("intensity", "u1"), ("pad2", "u1"), ("ring", "<u2"),
                     ("pad3", "<u4"), ("timestamp", "<f8")])


class ScanRegOut(C.Structure):
    _fields_ = [("n", C.c_int32)] + [(k, C.c_void_p) for k in ("cloud", "curvature", "label", "sort_ind", "picked", "scan_start", "scan_end",
                                                                 "sharp", "less_sharp", "flat", "less_flat")] + [("counts", C.c_int32 * 4)]


class VoxelInfo(C.Structure):
    _fields_ = [("n_leaves", C.c_int32), ("n_points", C.c_int32), ("min_b", C.c_int32 * 3), ("max_b", C.c_int32 * 3), ("div_b", C.c_int32 * 3), ("divb_mul", C.c_int32 * 3)]


def scan_register(ctx, pts, n_rings, min_range):
    """A-LOAM scanRegistration core on the GPU; same result dict as oracle.scan_register."""
    pts = np.ascontiguousarray(pts, dtype=RS_POINT)
    n = len(pts)
    cap = max(n, 1)
    out = dict(cloud=np.zeros((cap, 4), np.float32), curvature=np.zeros(cap, np.float32), label=np.zeros(cap, np.int32), sort_ind=np.zeros(cap, np.int32),
               picked=np.zeros(cap, np.int32), scan_start=np.zeros(n_rings, np.int32), scan_end=np.zeros(n_rings, np.int32),
               sharp=np.zeros(cap, np.int32), less_sharp=np.zeros(cap, np.int32), flat=np.zeros(cap, np.int32), less_flat=np.zeros(cap, np.int32))
    so = ScanRegOut()
    for k in out:
        setattr(so, k, out[k].ctypes.data)
    ctx._ck(ctx._l.lvx_scan_register(ctx._h, C.c_int(n), _p(pts), C.c_int(n_rings), C.c_float(min_range), C.byref(so)))
    m = so.n
    for k in ("cloud", "curvature", "label", "sort_ind", "picked"):
        out[k] = out[k][:m]
    for i, k in enumerate(("sharp", "less_sharp", "flat", "less_flat")):
        out[k] = out[k][:so.counts[i]].copy()
    out["n"] = m
    return out


def scan_register_batch(ctx, sweeps, n_rings, min_range, fetch=True):
    """lvx_scan_register_batch: a list of RS_POINT arrays (one per sweep) in one call; returns one result dict per sweep (as scan_register).  fetch=False: only the
    counts come back (timing of the device work)."""
    sweeps = [np.ascontiguousarray(p, dtype=RS_POINT) for p in sweeps]
    S = len(sweeps)
    off = np.zeros(S + 1, np.int32)
    off[1:] = np.cumsum([len(p) for p in sweeps])
    allp = np.concatenate(sweeps) if S else np.zeros(0, RS_POINT)
    outs = (ScanRegOut * S)()
    res = []
    for s_, p in enumerate(sweeps):
        cap = max(len(p), 1)
        o = dict(scan_start=np.zeros(n_rings, np.int32), scan_end=np.zeros(n_rings, np.int32))
        if fetch:
            o.update(cloud=np.zeros((cap, 4), np.float32), curvature=np.zeros(cap, np.float32), label=np.zeros(cap, np.int32), sort_ind=np.zeros(cap, np.int32), picked=np.zeros(cap, np.int32),
                     sharp=np.zeros(cap, np.int32), less_sharp=np.zeros(cap, np.int32), flat=np.zeros(cap, np.int32), less_flat=np.zeros(cap, np.int32))
        for k in o:
            setattr(outs[s_], k, o[k].ctypes.data)
        res.append(o)
    ctx._ck(ctx._l.lvx_scan_register_batch(ctx._h, C.c_int(S), _p(off), _p(allp), C.c_int(n_rings), C.c_float(min_range), outs))
    for s_, o in enumerate(res):
        m = outs[s_].n
        o["n"] = m
        o["counts"] = list(outs[s_].counts)
        if fetch:
            for k in ("cloud", "curvature", "label", "sort_ind", "picked"):
                o[k] = o[k][:m]
            for i, k in enumerate(("sharp", "less_sharp", "flat", "less_flat")):
                o[k] = o[k][:outs[s_].counts[i]].copy()
    return res


def scan_register_batch_d(ctx, pts_d_ptr, offsets, n_rings, min_range):
    """lvx_scan_register_batch_d: points resident on the device (pointer), results stay there; returns (n_kept [S], counts [S, 4])."""
    off = np.ascontiguousarray(offsets, np.int32)
    S = len(off) - 1
    nk, cnt = np.zeros(S, np.int32), np.zeros((S, 4), np.int32)
    ctx._ck(ctx._l.lvx_scan_register_batch_d(ctx._h, C.c_int(S), _p(off), C.c_void_p(pts_d_ptr), C.c_int(n_rings), C.c_float(min_range), _p(nk), _p(cnt)))
    return nk, cnt


def scan_register_get(ctx, sweep, cap, n_rings):
    """Download one sweep of the context's last scan_register_batch(_d): result dict as scan_register."""
    cap = max(cap, 1)
    out = dict(cloud=np.zeros((cap, 4), np.float32), curvature=np.zeros(cap, np.float32), label=np.zeros(cap, np.int32), sort_ind=np.zeros(cap, np.int32),
               picked=np.zeros(cap, np.int32), scan_start=np.zeros(n_rings, np.int32), scan_end=np.zeros(n_rings, np.int32),
               sharp=np.zeros(cap, np.int32), less_sharp=np.zeros(cap, np.int32), flat=np.zeros(cap, np.int32), less_flat=np.zeros(cap, np.int32))
    so = ScanRegOut()
    for k in out:
        setattr(so, k, out[k].ctypes.data)
    ctx._ck(ctx._l.lvx_scan_register_get(ctx._h, C.c_int(sweep), C.byref(so)))
    m = so.n
    for k in ("cloud", "curvature", "label", "sort_ind", "picked"):
        out[k] = out[k][:m]
    for i, k in enumerate(("sharp", "less_sharp", "flat", "less_flat")):
        out[k] = out[k][:so.counts[i]].copy()
    out["n"] = m
    return out


def voxel_build(ctx, xyzi, leaf, min_pts=6, eig_mult=0.01, fetch=True):
    xyzi = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    info = VoxelInfo()
    ctx._ck(ctx._l.lvx_voxel_build(ctx._h, C.c_int(len(xyzi)), _p(xyzi), C.c_float(leaf), C.c_int(min_pts), C.c_double(eig_mult), C.byref(info)))
    nl, n = info.n_leaves, len(xyzi)
    o = dict(n_leaves=nl, grid=np.array(list(info.min_b) + list(info.max_b) + list(info.div_b) + list(info.divb_mul), np.int32))
    if fetch:
        o.update(leaf_key=np.zeros(nl, np.int32), leaf_n=np.zeros(nl, np.int32), mean=np.zeros((nl, 3)), cov=np.zeros((nl, 9)), icov=np.zeros((nl, 9)),
                 evecs=np.zeros((nl, 9)), evals=np.zeros((nl, 3)), centroid=np.zeros((nl, 3), np.float32), offsets=np.zeros(nl + 1, np.int32),
                 point_ids=np.zeros(max(n, 1), np.int32))
        ctx._ck(ctx._l.lvx_voxel_get(ctx._h, *[_p(o[k]) for k in ("leaf_key", "leaf_n", "mean", "cov", "icov", "evecs", "evals", "centroid", "offsets", "point_ids")]))
    return o


def voxel_lookup7(ctx, queries):
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 4)
    ids = np.full((len(q), 7), -1, np.int32)
    ctx._ck(ctx._l.lvx_voxel_lookup7(ctx._h, C.c_int(len(q)), _p(q), _p(ids)))
    return ids


def voxel_lookup1(ctx, queries):
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 4)
    ids = np.full(len(q), -1, np.int32)
    ctx._ck(ctx._l.lvx_voxel_lookup1(ctx._h, C.c_int(len(q)), _p(q), _p(ids)))
    return ids


def surfel_assoc(ctx, scan_hw4, p4, box_min, box_max, radius=0.05, sel=2):
    scan = np.ascontiguousarray(scan_hw4, dtype=np.float32)
    H, W = scan.shape[0], scan.shape[1]
    p4, box_min, box_max = _d(p4), _d(box_min), _d(box_max)
    flag = np.full(H * W, -1, np.int32)
    ctx._ck(ctx._l.lvx_surfel_assoc(ctx._h, C.c_int(H), C.c_int(W), _p(scan), C.c_int(len(p4)), _p(p4), _p(box_min), _p(box_max), C.c_double(radius), C.c_int(sel), _p(flag)))
    return flag.reshape(H, W)


def surfel_assoc_emit(ctx, scans_map, scans_raw, p4, box_min, box_max, radius=0.05, sel=2):
    """getAssociation for a batch of scans, on the device end to end: scans_map [S, H, W, 4] float32 (map frame), scans_raw [S, H, W] POINT_XYZIT.
    Returns (flags [S, H, W], dict(pt, pt_map, t, plane, counts)) — the SurfelPoint list of all scans in chronological order."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    sm = np.ascontiguousarray(scans_map, np.float32)
    S, H, W = sm.shape[0], sm.shape[1], sm.shape[2]
    raw = np.ascontiguousarray(scans_raw, dtype=POINT_XYZIT).reshape(S, H, W)
    sm_d = torch.from_numpy(sm).to(dev)
    raw_d = torch.from_numpy(raw.view(np.uint8).reshape(-1)).to(dev)
    pl_d = torch.from_numpy(np.concatenate([_d(p4).ravel(), _d(box_min).ravel(), _d(box_max).ravel()])).to(dev)
    flags_d = torch.empty(S * H * W, dtype=torch.int32, device=dev)
    l = ctx._l
    if len(p4):   # this call owns pl_d: one grid for all chunks, released before the table can be freed
        ctx._ck(l.lvx_surfel_map_prepare_d(ctx._h, C.c_int(len(p4)), C.c_void_p(pl_d.data_ptr())))
    try:
        ctx._ck(l.lvx_surfel_assoc_batch_d(ctx._h, C.c_int(S), C.c_int(H), C.c_int(W), C.c_void_p(sm_d.data_ptr()), C.c_int(len(p4)), C.c_void_p(pl_d.data_ptr()), C.c_double(radius), C.c_int(sel),
                                           C.c_void_p(flags_d.data_ptr())))
    finally:
        l.lvx_surfel_map_release(ctx._h)
    n = C.c_int32(0)
    counts = np.zeros(S, np.int32)
    ctx._ck(l.lvx_surfel_emit_d(ctx._h, C.c_int(S), C.c_int(H), C.c_int(W), C.c_void_p(flags_d.data_ptr()), C.c_void_p(sm_d.data_ptr()), C.c_void_p(raw_d.data_ptr()), C.c_int(0),
                                None, None, None, None, C.byref(n), _p(counts)))
    k = n.value
    pt_d = torch.empty(max(k, 1) * 3, dtype=torch.float64, device=dev); pm_d = torch.empty_like(pt_d)
    t_d = torch.empty(max(k, 1), dtype=torch.float64, device=dev); pid_d = torch.empty(max(k, 1), dtype=torch.int32, device=dev)
    if k:
        ctx._ck(l.lvx_surfel_emit_d(ctx._h, C.c_int(S), C.c_int(H), C.c_int(W), C.c_void_p(flags_d.data_ptr()), C.c_void_p(sm_d.data_ptr()), C.c_void_p(raw_d.data_ptr()), C.c_int(k),
                                    C.c_void_p(pt_d.data_ptr()), C.c_void_p(pm_d.data_ptr()), C.c_void_p(t_d.data_ptr()), C.c_void_p(pid_d.data_ptr()), C.byref(n), _p(counts)))
    ctx.synchronize()
    return flags_d.cpu().numpy().reshape(S, H, W), dict(pt=pt_d.cpu().numpy().reshape(-1, 3)[:k], pt_map=pm_d.cpu().numpy().reshape(-1, 3)[:k], t=t_d.cpu().numpy()[:k],
                                                          plane=pid_d.cpu().numpy()[:k], counts=counts)


def landmark_assoc(ctx, state, q_LtoC_xyzw, t_LinC, map_time, p4, box_min, box_max, radius=0.05):
    """associateVisualPointsWithPlanes: surfel index per landmark (or -1)."""
    p4, box_min, box_max = _d(p4), _d(box_min), _d(box_max)
    L = ctx.state_size - 7 * ctx.layout()["n_knots"] - 32
    out = np.full(max(L, 1), -1, np.int32)
    ctx._ck(ctx._l.lvx_landmark_assoc(ctx._h, _p(_d(state)), _p(_d(q_LtoC_xyzw)), _p(_d(t_LinC)), C.c_double(map_time), C.c_int(len(p4)), _p(p4), _p(box_min), _p(box_max), C.c_double(radius), _p(out)))
    return out[:L]


def upstream_bench(ctx, kind, arrays, reps=20):
    """Time one upstream kernel with its inputs RESIDENT on the device (torch tensors; `_d` entry points).  Returns seconds per call."""
    import torch
    l = ctx._l
    dev = torch.device("cuda", torch.cuda.current_device())
    if kind == "surfel_assoc":   # scan: [H, W, 4] (one scan) or [S, H, W, 4] (a batch per call)
        scan, p4, bmin, bmax = arrays
        scan = np.ascontiguousarray(scan, np.float32)
        if scan.ndim == 3:
            scan = scan[None]
        S, H, W = scan.shape[0], scan.shape[1], scan.shape[2]
        P = len(p4)
        scan_d = torch.from_numpy(scan).to(dev)
        pl_d = torch.from_numpy(np.concatenate([_d(p4).ravel(), _d(bmin).ravel(), _d(bmax).ravel()])).to(dev)
        flag_d = torch.empty(S * H * W, dtype=torch.int32, device=dev)
        call = lambda: ctx._ck(l.lvx_surfel_assoc_batch_d(ctx._h, C.c_int(S), C.c_int(H), C.c_int(W), C.c_void_p(scan_d.data_ptr()), C.c_int(P), C.c_void_p(pl_d.data_ptr()),
                                                          C.c_double(0.05), C.c_int(2), C.c_void_p(flag_d.data_ptr())))
    elif kind == "voxel_build":
        cloud, leaf = arrays
        n = len(cloud)
        cloud_d = torch.from_numpy(np.ascontiguousarray(cloud, np.float32)).to(dev)
        call = lambda: ctx._ck(l.lvx_voxel_build_d(ctx._h, C.c_int(n), C.c_void_p(cloud_d.data_ptr()), C.c_float(leaf), C.c_int(6), C.c_double(0.01)))
    elif kind == "voxel_lookup7":
        q = arrays
        n = len(q)
        q_d = torch.from_numpy(np.ascontiguousarray(q, np.float32)).to(dev)
        ids_d = torch.empty(n * 7, dtype=torch.int32, device=dev)
        call = lambda: ctx._ck(l.lvx_voxel_lookup7_d(ctx._h, C.c_int(n), C.c_void_p(q_d.data_ptr()), C.c_void_p(ids_d.data_ptr())))
    else:
        raise ValueError(kind)
    import time
    call(); ctx.synchronize()
    if kind == "voxel_build":   # the first build of a cloud sizes the context's cell table (grown on demand when somebody asks for the result): ask, then build once more
        ctx.voxel_info(); call(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps


def scan_less_flat_downsample(ctx, n_rings, max_out, leaf=0.2, sweep=0):
    """The published less-flat cloud of sweep `sweep` of the context's last scan_register / scan_register_batch: (xyzi [n][4], per-ring counts)."""
    out = np.zeros((max(max_out, 1), 4), np.float32)
    rc = np.zeros(n_rings, np.int32)
    n = C.c_int32(0)
    ctx._ck(ctx._l.lvx_scan_less_flat_downsample_sweep(ctx._h, C.c_int(sweep), C.c_float(leaf), C.c_int(max_out), _p(out), _p(rc), C.byref(n)))
    return out[:min(n.value, max_out)], rc, n.value


def ndt_derivatives(ctx, src, trans, p6, outlier_ratio=0.55, compute_hessian=True):
    """Score, gradient, Hessian of the NDT objective against the context's last voxel_build (computeDerivatives, DIRECT7)."""
    src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1, 4)
    trans = np.ascontiguousarray(trans, dtype=np.float32).reshape(-1, 4)
    score = C.c_double(0)
    g, H = np.zeros(6), np.zeros((6, 6))
    ctx._ck(ctx._l.lvx_ndt_derivatives(ctx._h, C.c_int(len(src)), _p(src), _p(trans), _p(_d(p6)), C.c_double(outlier_ratio), C.c_int(1 if compute_hessian else 0), C.byref(score), _p(g), _p(H)))
    return score.value, g, H


class NdtOptions(C.Structure):
    _fields_ = [("step_size", C.c_double), ("outlier_ratio", C.c_double), ("transformation_epsilon", C.c_double), ("max_iterations", C.c_int32), ("search", C.c_int32)]


class NdtResult(C.Structure):
    _fields_ = [("final_transformation", C.c_float * 16), ("p6", C.c_double * 6), ("score", C.c_double), ("trans_probability", C.c_double), ("iterations", C.c_int32),
                ("converged", C.c_int32), ("n_evaluations", C.c_int32), ("reserved", C.c_int32)]


def ndt_align(ctx, src, guess=None, search=7, want_aligned=False, **kw):
    """pclomp::NormalDistributionsTransform::align against the context's last voxel_build (setInputTarget at that resolution): returns a dict with the final 4 x 4
    transformation, the 6-vector, iteration / evaluation counts, the last score (and the aligned cloud)."""
    src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1, 4)
    opt = NdtOptions()
    ctx._ck(ctx._l.lvx_ndt_default_options(C.byref(opt)))
    opt.search = search
    for k, v in kw.items():
        setattr(opt, k, v)
    res = NdtResult()
    g = None if guess is None else np.ascontiguousarray(guess, dtype=np.float32).reshape(16)
    out = np.zeros_like(src) if want_aligned else None
    ctx._ck(ctx._l.lvx_ndt_align(ctx._h, C.c_int(len(src)), _p(src), None if g is None else _p(g), C.byref(opt), C.byref(res), None if out is None else _p(out)))
    r = dict(final_transformation=np.array(res.final_transformation, np.float32).reshape(4, 4), p=np.array(res.p6), score=res.score, trans_probability=res.trans_probability,
             iterations=res.iterations, converged=bool(res.converged), n_evaluations=res.n_evaluations)
    if want_aligned:
        r["aligned"] = out
    return r


def ndt_fitness(ctx, src, transform, tgt, max_range=float(np.finfo(np.float64).max)):
    """pcl::Registration::getFitnessScore: mean squared distance of the transformed source points to their nearest target points."""
    src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1, 4)
    tgt = np.ascontiguousarray(tgt, dtype=np.float32).reshape(-1, 4)
    T = np.ascontiguousarray(transform, dtype=np.float32).reshape(16)
    f = C.c_double(0)
    ctx._ck(ctx._l.lvx_ndt_fitness(ctx._h, C.c_int(len(src)), _p(src), _p(T), C.c_int(len(tgt)), _p(tgt), C.c_double(max_range), C.byref(f)))
    return f.value


def neighbor_cells_26():
    """pcl::getAllNeighborCellIndices(): the relative coordinates getNeighborhoodAtPoint(reference_point, neighbors) searches (DIRECT26; the centre cell is not one of them)."""
    half = [(i, j, -1) for i in (-1, 0, 1) for j in (-1, 0, 1)] + [(i, -1, 0) for i in (-1, 0, 1)] + [(-1, 0, 0)]
    h = np.array(half, np.int32)
    return np.concatenate([h, -h], axis=0)


def voxel_lookup_rel(ctx, queries, rel):
    """getNeighborhoodAtPoint(relative_coordinates, point): ids [nq, n_rel], -1 = no usable leaf at that displacement."""
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 4)
    rel = np.ascontiguousarray(rel, dtype=np.int32).reshape(-1, 3)
    ids = np.full((len(q), len(rel)), -1, np.int32)
    ctx._ck(ctx._l.lvx_voxel_lookup_rel(ctx._h, C.c_int(len(q)), _p(q), C.c_int(len(rel)), _p(rel), _p(ids)))
    return ids


SURFEL_PLANE = np.dtype([("p4", "<f8", 4), ("Pi", "<f8", 3), ("box_min", "<f8", 3), ("box_max", "<f8", 3), ("leaf", "<i4"), ("n_points", "<i4"), ("n_inliers", "<i4"),
                         ("plane_type", "<i4")])


def surfel_extract(ctx, max_planes, p_lambda=0.7, dist_threshold=0.05, min_leaf_points=10, min_inliers=20):
    """Surfel planes of the context's last voxel_build (setSurfelMap): structured array in voxel-key order."""
    out = np.zeros(max(max_planes, 1), dtype=SURFEL_PLANE)
    n = C.c_int32(0)
    ctx._ck(ctx._l.lvx_surfel_extract(ctx._h, C.c_double(p_lambda), C.c_double(dist_threshold), C.c_int(min_leaf_points), C.c_int(min_inliers), C.c_int(max_planes), _p(out), C.byref(n)))
    return out[:min(n.value, max_planes)], n.value


POINT_XYZIT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad", "<f4"), ("intensity", "<f4"), ("pad2", "<f4"), ("timestamp", "<f8")])


class AssocOptions(C.Structure):
    """lvx_assoc_options (include/lvx.h): the reference's DataAssociation parameters."""
    _fields_ = [("ndt_resolution", C.c_float), ("min_points_per_voxel", C.c_int32), ("min_covar_eigvalue_mult", C.c_double), ("plane_lambda", C.c_double), ("fit_threshold", C.c_double),
                ("min_leaf_points", C.c_int32), ("min_inliers", C.c_int32), ("radius", C.c_double), ("selected_per_ring", C.c_int32), ("reserved", C.c_int32)]


def assoc_default_options(ctx, **kw):
    o = AssocOptions()
    ctx._ck(ctx._l.lvx_assoc_default_options(C.byref(o)))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def set_scans(ctx, raw, H, W):
    """The dataset's organised raw scans [n_scans][H][W] (POINT_XYZIT), handed over once (lvx_set_scans)."""
    raw = np.ascontiguousarray(raw, dtype=POINT_XYZIT)
    assert raw.size % (H * W) == 0
    ctx._ck(ctx._l.lvx_set_scans(ctx._h, C.c_int(raw.size // (H * W)), C.c_int(H), C.c_int(W), _p(raw)))


def data_association(ctx, state, map_time, options=None):
    """One DataAssociation round (lvx_data_association): returns (number of surfels, number of SurfelPoints); the lists stay in the context."""
    npl, npt = C.c_int32(0), C.c_int32(0)
    ctx._ck(ctx._l.lvx_data_association(ctx._h, _p(_d(state)), C.c_double(map_time), C.byref(options) if options is not None else None, C.byref(npl), C.byref(npt)))
    return npl.value, npt.value


def data_association_poses(ctx, state, scan_t, poses16, has_pose=None, key_dist=0.2, key_angle_deg=5.0, options=None):
    """The first DataAssociation of a calibration, map from per-scan odometry poses (lvx_data_association_poses): (surfels, SurfelPoints, key-scan flags)."""
    scan_t, poses16 = _d(scan_t), _d(np.asarray(poses16, dtype=np.float64).reshape(-1, 16))
    assert len(poses16) == len(scan_t)
    hp = np.ascontiguousarray(has_pose, dtype=np.int32) if has_pose is not None else None
    key = np.zeros(len(scan_t), np.int32)
    npl, npt = C.c_int32(0), C.c_int32(0)
    ctx._ck(ctx._l.lvx_data_association_poses(ctx._h, _p(_d(state)), _p(scan_t), _p(poses16), _p(hp) if hp is not None else None, C.c_double(key_dist), C.c_double(key_angle_deg),
                                              C.byref(options) if options is not None else None, C.byref(npl), C.byref(npt), _p(key)))
    return npl.value, npt.value, key


def data_association_stats(ctx):
    """(rounds that took the one-stop chain, how many of those were repeated on the four-stop chain)."""
    a, b = C.c_int64(0), C.c_int64(0)
    ctx._ck(ctx._l.lvx_data_association_stats(ctx._h, C.byref(a), C.byref(b)))
    return a.value, b.value


def get_surfel_map(ctx, n_planes):
    out = np.zeros(max(n_planes, 1), dtype=SURFEL_PLANE)
    ctx._ck(ctx._l.lvx_get_surfel_map(ctx._h, C.c_int(n_planes), _p(out)))
    return out[:n_planes]


def get_surfel_points(ctx, n_points):
    pt, pm, t, pl = np.zeros((max(n_points, 1), 3)), np.zeros((max(n_points, 1), 3)), np.zeros(max(n_points, 1)), np.zeros(max(n_points, 1), np.int32)
    ctx._ck(ctx._l.lvx_get_surfel_points(ctx._h, C.c_int(n_points), _p(pt), _p(pm), _p(t), _p(pl)))
    return dict(pt=pt[:n_points], pt_map=pm[:n_points], t=t[:n_points], plane=pl[:n_points])


def get_scans_in_map(ctx, n_scans, H, W):
    out = np.zeros((n_scans, H, W, 4), np.float32)
    ctx._ck(ctx._l.lvx_get_scans_in_map(ctx._h, _p(out)))
    return out


def eval_lidar_pose(ctx, state, t):
    state, t = _d(state), _d(np.atleast_1d(t))
    n = len(t)
    q, p, ok = np.zeros((n, 4)), np.zeros((n, 3)), np.zeros(n, np.int32)
    ctx._ck(ctx._l.lvx_evaluate_lidar_pose(ctx._h, _p(state), C.c_int(n), _p(t), _p(q), _p(p), _p(ok)))
    return q, p, ok.astype(bool)


def undistort(ctx, state, raw, q_G_to_target, p_target_in_G, correct_position=True):
    raw = np.ascontiguousarray(raw, dtype=POINT_XYZIT)
    out = np.zeros((len(raw), 4), np.float32)
    ctx._ck(ctx._l.lvx_undistort_scan(ctx._h, _p(_d(state)), C.c_int(len(raw)), _p(raw), _p(_d(q_G_to_target)), _p(_d(p_target_in_G)), C.c_int(1 if correct_position else 0), _p(out)))
    return out


def load_problem(obj, P, locks=None):
    """Feed a synth.make_problem() dict into an lvx.Context or an oracle.Oracle (same setter names)."""
    obj.set_spline(P["t0"], P["dt"], P["n_knots"])
    c = P["camera"]
    obj.set_camera(c["rows"], c["cols"], c["readout"], c["fx"], c["fy"], c["cx"], c["cy"], c["k1"], c["k2"], c["p1"], c["p2"], c["k3"])
    obj.set_imu(P["t_imu"], P["gyro"], P["acc"], P["w_gyro"], P["w_acc"])
    obj.set_planes(P["planes"])
    obj.set_surfel(P["surf_pt"], P["surf_t"], P["surf_plane"], P["t_map"], P["huber_surf"], P["w_surf"])
    obj.set_landmarks(P["lm_uv"], P["lm_t0"])
    obj.set_reproj(P["rep_lm"], P["rep_uv"], P["rep_t0"], P["huber_rep"], P["w_rep"])
    obj.set_camsurf(P["cs_lm"], P["cs_plane"], P["t_map"], P["huber_cs"], P["w_cs"])
    if locks is not None:
        obj.set_locks(locks)
