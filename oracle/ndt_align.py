"""TEST INFRASTRUCTURE — CPU restatement (oracle) of ndt_omp's registration loop.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.

Follows /root/reference/src/ndt_omp/include/pclomp/ndt_omp_impl.hpp:
    NdtAligner.align            computeTransformation                       :81-171   (defaults of the constructor :46-76: step 0.1, outlier ratio 0.55,
                                                                                      transformation_epsilon 0.1, 35 iterations, DIRECT7)
    NdtAligner._derivatives     computeDerivatives                          :180-285  (oracle/orc_upstream.cpp orc_ndt_derivatives, float per-point arithmetic)
    NdtAligner._hessian         computeHessian / updateHessian              :540-645  (oracle/orc_ndt.cpp, double)
    _update_interval            updateIntervalMT                            :648-685
    _trial_value                trialValueSelectionMT                       :689-768
    NdtAligner._step_length     computeStepLengthMT                         :772-931
    NdtAligner.fitness          pcl::Registration::getFitnessScore          (apps/align.cpp:30; PCL, out of tree: mean squared nearest-neighbour distance)
and drives them exactly as /root/reference/src/ndt_omp/apps/align.cpp:15-33,60-69,85-103 does: both clouds through pcl::VoxelGrid(0.1 m), resolution 1.0,
identity guess, align() then getFitnessScore().

PINNED by the reference's own published outputs: src/ndt_omp/README.md:21 (DIRECT7 fitness 0.214205) and :26 (DIRECT1 0.208511) on the two scans the reference
ships (tests/golden/ndt_data_*.npz) — tests/test_ndt_align_oracle.py.
"""
import ctypes as C
import math

import numpy as np

from . import oracle as O

_p = O._p
_d = O._d

DIRECT1 = 1
DIRECT7 = 7
DIRECT26 = 26
KDTREE = 0       # radiusSearch over the leaf centroids (ndt_omp_impl.hpp:234-236; voxel_grid_covariance_omp.h:473-502): what pcl::NormalDistributionsTransform itself does
REL7 = np.array([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.int32)   # voxel_grid_covariance_omp_impl.hpp:427-434


def neighbor_cells_26():
    """pcl::getAllNeighborCellIndices() (pcl/filters/voxel_grid.h, out of tree): the 13 'half' neighbours — (i, j, -1) for i, j in -1..1, then (i, -1, 0), then
    (-1, 0, 0) — followed by their negatives; the centre cell is not part of it."""
    half = [(i, j, -1) for i in (-1, 0, 1) for j in (-1, 0, 1)] + [(i, -1, 0) for i in (-1, 0, 1)] + [(-1, 0, 0)]
    h = np.array(half, np.int32)
    return np.concatenate([h, -h], axis=0)


def rel_cells(search):
    return {DIRECT1: REL7[:1], DIRECT7: REL7, DIRECT26: neighbor_cells_26()}[search].copy()


def voxel_lookup_rel(vox, queries, leaf, rel, min_pts=6):
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 4)
    rel = np.ascontiguousarray(rel, dtype=np.int32).reshape(-1, 3)
    ids = np.full((len(q), len(rel)), -1, np.int32)
    lk, ln = np.ascontiguousarray(vox["leaf_key"]), np.ascontiguousarray(vox["leaf_n"])
    O.lib().orc_voxel_lookup_rel(C.c_int(len(q)), _p(q), C.c_float(leaf), C.c_int(min_pts), _p(vox["grid"]), C.c_int(vox["n_leaves"]), _p(lk), _p(ln), C.c_int(len(rel)), _p(rel),
                                 _p(ids))
    return ids


def ndt_matrix(p6):
    M = np.zeros(16, np.float32)
    O.lib().orc_ndt_matrix(_p(_d(p6)), _p(M))
    return M.reshape(4, 4)


def euler012(M):
    e = np.zeros(3, np.float32)
    O.lib().orc_ndt_euler012(_p(np.ascontiguousarray(M, np.float32)), _p(e))
    return e


def transform_cloud(cloud, M):
    cloud = np.ascontiguousarray(cloud, np.float32).reshape(-1, 4)
    out = np.empty_like(cloud)
    O.lib().orc_transform_cloud(C.c_int(len(cloud)), _p(cloud), _p(np.ascontiguousarray(M, np.float32)), _p(out))
    return out


def fitness(src, M, tgt, max_range=float(np.finfo(np.float64).max)):
    src = np.ascontiguousarray(src, np.float32).reshape(-1, 4)
    tgt = np.ascontiguousarray(tgt, np.float32).reshape(-1, 4)
    l = O.lib()
    l.orc_fitness.restype = C.c_double
    return l.orc_fitness(C.c_int(len(src)), _p(src), _p(np.ascontiguousarray(M, np.float32)), C.c_int(len(tgt)), _p(tgt), C.c_double(max_range), None)


def _update_interval(I, a_t, f_t, g_t):
    """updateIntervalMT (:648-685) on I = [a_l, f_l, g_l, a_u, f_u, g_u]; returns interval_converged."""
    a_l, f_l, g_l = I[0], I[1], I[2]
    if f_t > f_l:                                   # case U1 / a
        I[3], I[4], I[5] = a_t, f_t, g_t
        return False
    if g_t * (a_l - a_t) > 0:                       # case U2 / b
        I[0], I[1], I[2] = a_t, f_t, g_t
        return False
    if g_t * (a_l - a_t) < 0:                       # case U3 / c
        I[3], I[4], I[5] = a_l, f_l, g_l
        I[0], I[1], I[2] = a_t, f_t, g_t
        return False
    return True


def _sqrt(x):
    return math.sqrt(x) if x >= 0 else float("nan")     # std::sqrt of a negative double is NaN, not an exception


def _div(a, b):
    """IEEE double division as C++ does it: x / 0 is +-inf or NaN, never an exception."""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0:
            return float("nan")
        return math.copysign(float("inf"), a) * math.copysign(1.0, b)


def _trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, f_t, g_t):
    """trialValueSelectionMT (:689-768)."""
    if f_t > f_l:                                   # case 1
        z = _div(3 * (f_t - f_l), a_t - a_l) - g_t - g_l
        w = _sqrt(z * z - g_t * g_l)
        a_c = a_l + _div((a_t - a_l) * (w - g_l - z), g_t - g_l + 2 * w)
        a_q = a_l - _div(0.5 * (a_l - a_t) * g_l, g_l - _div(f_l - f_t, a_l - a_t))
        return a_c if abs(a_c - a_l) < abs(a_q - a_l) else 0.5 * (a_q + a_c)
    if g_t * g_l < 0:                               # case 2
        z = _div(3 * (f_t - f_l), a_t - a_l) - g_t - g_l
        w = _sqrt(z * z - g_t * g_l)
        a_c = a_l + _div((a_t - a_l) * (w - g_l - z), g_t - g_l + 2 * w)
        a_s = a_l - _div(a_l - a_t, g_l - g_t) * g_l
        return a_c if abs(a_c - a_t) >= abs(a_s - a_t) else a_s
    if abs(g_t) <= abs(g_l):                        # case 3
        z = _div(3 * (f_t - f_l), a_t - a_l) - g_t - g_l
        w = _sqrt(z * z - g_t * g_l)
        a_c = a_l + _div((a_t - a_l) * (w - g_l - z), g_t - g_l + 2 * w)
        a_s = a_l - _div(a_l - a_t, g_l - g_t) * g_l
        a_next = a_c if abs(a_c - a_t) < abs(a_s - a_t) else a_s
        return min(a_t + 0.66 * (a_u - a_t), a_next) if a_t > a_l else max(a_t + 0.66 * (a_u - a_t), a_next)
    z = _div(3 * (f_t - f_u), a_t - a_u) - g_t - g_u   # case 4
    w = _sqrt(z * z - g_t * g_u)
    return a_u + _div((a_t - a_u) * (w - g_u - z), g_t - g_u + 2 * w)


def _cmin(a, b):
    return b if b < a else a        # std::min(a, b)


def _cmax(a, b):
    return b if a < b else a        # std::max(a, b)


class NdtAligner:
    """pclomp::NormalDistributionsTransform: setInputTarget -> voxel covariance grid at `resolution` (ndt_omp.h:117-122, 275-282), then align()."""

    def __init__(self, target, resolution=1.0, search=DIRECT7, step_size=0.1, outlier_ratio=0.55, transformation_epsilon=0.1, max_iterations=35, min_pts=6):
        self.target = np.ascontiguousarray(target, np.float32).reshape(-1, 4)
        self.res, self.search, self.step_size, self.outlier_ratio = float(resolution), search, step_size, outlier_ratio
        self.eps, self.max_iterations, self.min_pts = transformation_epsilon, max_iterations, min_pts
        self.vox = O.voxel_build(self.target, np.float32(resolution), min_pts)
        self.rel = rel_cells(search) if search != KDTREE else None
        if search == KDTREE:
            # kdtree_ over voxel_centroids_: the float centroids of the leaves that had >= min_points points when the filter ran (voxel_grid_covariance_omp_impl.hpp:301-317;
            # leaves rejected afterwards by the eigenvalue test stay in the tree with nr_points = -1: none occurs on the demo clouds)
            from scipy.spatial import cKDTree
            self.kd_sel = np.flatnonzero(np.abs(self.vox["leaf_n"]) >= min_pts).astype(np.int32)
            self.kd = cKDTree(self.vox["centroid"][self.kd_sel].astype(np.float64))
        self.mean, self.icov = np.ascontiguousarray(self.vox["mean"]), np.ascontiguousarray(self.vox["icov"])
        self.n_eval = 0
        self.trace = []

    # -- the three evaluations of the loop ------------------------------------------------------------------------------------------------------
    def _neighbourhood(self, trans):
        """Leaf ids per transformed point, one row per point, -1 = empty slot: the cells of the direct searches, or the leaves whose centroid lies within `resolution` of the
        point (KDTREE: radiusSearch(x_trans_pt, resolution_, ...), sorted by distance as FLANN returns them)."""
        if self.search != KDTREE:
            return voxel_lookup_rel(self.vox, trans, np.float32(self.res), self.rel, self.min_pts)
        q = trans[:, :3].astype(np.float64)
        found = self.kd.query_ball_point(q, self.res)
        width = max(1, max(len(f) for f in found))
        ids = np.full((len(trans), width), -1, np.int32)
        cen = self.kd.data
        for i, f in enumerate(found):
            if f:
                f = np.asarray(f)
                order = np.argsort(((cen[f] - q[i]) ** 2).sum(1), kind="stable")
                ids[i, :len(f)] = self.kd_sel[f[order]]
        return ids

    def _derivatives(self, trans, p, compute_hessian=True):
        self.n_eval += 1
        score = C.c_double(0)
        g, H = np.zeros(6), np.zeros((6, 6))
        ids = self._neighbourhood(trans)
        O.lib().orc_ndt_derivatives_n(C.c_int(len(self.src)), _p(self.src), _p(trans), C.c_int(ids.shape[1]), _p(ids), _p(self.mean), _p(self.icov), _p(_d(p)), C.c_double(self.res),
                                      C.c_double(self.outlier_ratio), C.c_int(1 if compute_hessian else 0), C.byref(score), _p(g), _p(H))
        return score.value, g, H

    def _hessian(self, trans, p):
        ids = self._neighbourhood(trans)
        H = np.zeros((6, 6))
        O.lib().orc_ndt_hessian(C.c_int(len(self.src)), _p(self.src), _p(trans), C.c_int(ids.shape[1]), _p(ids), _p(self.mean), _p(self.icov), _p(_d(p)), C.c_double(self.res),
                                C.c_double(self.outlier_ratio), _p(H))
        return H

    # -- computeStepLengthMT (:772-931) ----------------------------------------------------------------------------------------------------------
    def _step_length(self, x, step_dir, step_init, step_max, step_min, st):
        """st: dict(score, grad, hess, trans, final) updated in place (the reference's by-reference arguments and members); step_dir may be reversed in place."""
        phi_0 = -st["score"]
        d_phi_0 = -float(st["grad"] @ step_dir)
        if d_phi_0 >= 0:
            if d_phi_0 == 0:
                return 0.0
            d_phi_0 *= -1
            step_dir *= -1
        max_step_iterations, step_iterations = 10, 0
        mu, nu = 1.e-4, 0.9
        a_l = a_u = 0.0
        f_l = phi_0 - phi_0 - mu * d_phi_0 * a_l          # auxilaryFunction_PsiMT (ndt_omp.h:430-433)
        g_l = d_phi_0 - mu * d_phi_0                      # auxilaryFunction_dPsiMT (:443-446)
        f_u = phi_0 - phi_0 - mu * d_phi_0 * a_u
        g_u = d_phi_0 - mu * d_phi_0
        I = [a_l, f_l, g_l, a_u, f_u, g_u]
        interval_converged, open_interval = (step_max - step_min) < 0, True
        a_t = _cmax(_cmin(step_init, step_max), step_min)
        x_t = x + step_dir * a_t
        st["final"] = ndt_matrix(x_t)
        st["trans"] = transform_cloud(self.src, st["final"])
        st["score"], st["grad"], st["hess"] = self._derivatives(st["trans"], x_t, True)
        phi_t = -st["score"]
        d_phi_t = -float(st["grad"] @ step_dir)
        psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t
        d_psi_t = d_phi_t - mu * d_phi_0
        while not interval_converged and step_iterations < max_step_iterations and not (psi_t <= 0 and d_phi_t <= -nu * d_phi_0):
            if open_interval:
                a_t = _trial_value(I[0], I[1], I[2], I[3], I[4], I[5], a_t, psi_t, d_psi_t)
            else:
                a_t = _trial_value(I[0], I[1], I[2], I[3], I[4], I[5], a_t, phi_t, d_phi_t)
            a_t = _cmax(_cmin(a_t, step_max), step_min)
            x_t = x + step_dir * a_t
            st["final"] = ndt_matrix(x_t)
            st["trans"] = transform_cloud(self.src, st["final"])
            st["score"], st["grad"], _ = self._derivatives(st["trans"], x_t, False)     # the Hessian argument is zeroed by the call (:187) and not recomputed
            st["hess"] = np.zeros((6, 6))
            phi_t = -st["score"]
            d_phi_t = -float(st["grad"] @ step_dir)
            psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t
            d_psi_t = d_phi_t - mu * d_phi_0
            if open_interval and (psi_t <= 0 and d_psi_t >= 0):
                open_interval = False
                I[1] = I[1] + phi_0 - mu * d_phi_0 * I[0]
                I[2] = I[2] + mu * d_phi_0
                I[4] = I[4] + phi_0 - mu * d_phi_0 * I[3]
                I[5] = I[5] + mu * d_phi_0
            if open_interval:
                interval_converged = _update_interval(I, a_t, psi_t, d_psi_t)
            else:
                interval_converged = _update_interval(I, a_t, phi_t, d_phi_t)
            step_iterations += 1
        if step_iterations:
            st["hess"] = self._hessian(st["trans"], x_t)
        st["mt_iterations"] = step_iterations
        return a_t

    # -- computeTransformation (:81-171) behind pcl::Registration::align --------------------------------------------------------------------------
    def align(self, source, guess=None):
        self.src = np.ascontiguousarray(source, np.float32).reshape(-1, 4)
        self.n_eval, self.trace = 0, []
        final = np.eye(4, dtype=np.float32)              # Registration::align: final_transformation_ = transformation_ = previous_transformation_ = Identity
        output = self.src.copy()
        if guess is not None and not np.array_equal(np.asarray(guess, np.float32), np.eye(4, dtype=np.float32)):
            final = np.ascontiguousarray(guess, np.float32).reshape(4, 4)
            output = transform_cloud(output, final)
        e = euler012(final)
        p = np.array([final[0, 3], final[1, 3], final[2, 3], e[0], e[1], e[2]], np.float64)
        st = dict(final=final, trans=output)
        st["score"], st["grad"], st["hess"] = self._derivatives(output, p, True)
        nr_iterations, converged = 0, False
        while not converged:
            # JacobiSVD(hessian, FullU | FullV).solve(-score_gradient) (:127-129): the minimum-norm least-squares solution; the rank decision is Eigen's default threshold
            U, s, Vt = np.linalg.svd(st["hess"])
            thr = max(s[0], 0.0) * np.finfo(np.float64).eps * 6 if len(s) else 0.0
            inv = np.array([1.0 / v if v > thr else 0.0 for v in s])
            delta_p = Vt.T @ (inv * (U.T @ (-st["grad"])))
            delta_p_norm = float(np.sqrt(delta_p @ delta_p))
            if delta_p_norm == 0 or delta_p_norm != delta_p_norm:
                converged = delta_p_norm == delta_p_norm
                break
            delta_p = delta_p / delta_p_norm
            delta_p_norm = self._step_length(p, delta_p, delta_p_norm, self.step_size, self.eps / 2, st)
            delta_p = delta_p * delta_p_norm
            p = p + delta_p
            self.trace.append(dict(step=delta_p_norm, score=st["score"], mt=st.get("mt_iterations", 0), p=p.copy()))
            if nr_iterations > self.max_iterations or (nr_iterations and abs(delta_p_norm) < self.eps):
                converged = True
            nr_iterations += 1
        self.final_transformation = st["final"]
        self.p = p
        self.nr_iterations = nr_iterations
        self.trans_probability = st["score"] / float(len(self.src))
        return st["final"]

    def fitness(self, max_range=float(np.finfo(np.float64).max)):
        return fitness(self.src, self.final_transformation, self.target, max_range)


def align_like_the_demo(target_raw, source_raw, search, leaf=0.1, resolution=1.0):
    """apps/align.cpp:60-69,85-103: VoxelGrid(0.1) both clouds, resolution 1.0, identity guess; returns (aligner, fitness)."""
    tgt, src = O.voxelgrid_xyzi(target_raw, leaf), O.voxelgrid_xyzi(source_raw, leaf)
    a = NdtAligner(tgt, resolution=resolution, search=search)
    a.align(src)
    return a, a.fitness()
