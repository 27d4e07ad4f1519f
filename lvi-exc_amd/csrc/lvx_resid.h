// lvx_resid.h — per-measurement residual + analytic local Jacobian (FP64) for the LVI-ExC solve.
//
// Each function evaluates ONE residual block of one measurement family and returns its weighted residual(s)
// and the Jacobian w.r.t. the block's LOCAL tangent columns (layout documented per function).  The reference
// computes these with ceres autodiff over the functors cited below; the oracle restates that literally, and
// tests compare the two.  Quaternion tangents are ceres::EigenQuaternionParameterization deltas.
//
// Citations into /root/reference/src/lvi_exc/thirdparty/Kontiki/include/kontiki/.
#pragma once
#include "lvx_math.h"

namespace lvx {

// state pointers (global memory or LDS) + master spline meta (trajectories/spline_base.h:31-39)
struct SplineRef {
  double t0, dt;
  int n;
  const double* r3;    // [n][3]
  const double* so3;   // [n][4] (x,y,z,w)
};

struct CamIntr {   // sensors/pinhole_camera.h:20-41 + camera.h:25-29
  double fx, fy, cx, cy, k1, k2, p1, p2, k3, readout;
  double inv_K11, inv_K13, inv_K22, inv_K23;
  int rows, cols, do_distortion;
};

struct ImuCal { double roll, pitch; v3 ba, bg; double tau; };      // sensors/imu.h, constant_bias_imu.h
struct SensorCal { quat q; v3 p; double tau; };                     // sensors/sensors.h:36-85

LVX_HD v3 load_v3(const double* p) { return mk(p[0], p[1], p[2]); }
LVX_HD quat load_q(const double* p) { quat q; q.x = p[0]; q.y = p[1]; q.z = p[2]; q.w = p[3]; return q; }

// ---------------------------------------------------------------------------------------------
// Segment construction for a residual's time spans (spline_base.h:380-426) + CheckTimeSpans
// (trajectory_estimator.h:102-127) and the segment dispatch of SplineView::Evaluate (spline_base.h:194-222).
// ---------------------------------------------------------------------------------------------
struct Segs { int nseg; int i1[2]; int n[2]; };

LVX_HD bool build_segments(const SplineRef& sp, const double spans[][2], int nspans, Segs* s) {
  if (sp.n < 4) return false;
  const double tmin = sp.t0, tmax = madd_2r((double)(sp.n - 3), sp.dt, sp.t0);
  double t1_prev = 0.0;
  // at most two segments: kept in scalars (a dynamically indexed i1[] / n[] would live in scratch memory on the GPU)
  int nseg = 0, i1a = 0, i1b = 0, na = 0, nb = 0;
  int cur_start = 0, cur_end = -1;
  for (int k = 0; k < nspans; ++k) {
    const double t1 = spans[k][0], t2 = spans[k][1];
    if ((t1 < tmin) || (t2 >= tmax)) return false;
    if (t1 > t2) return false;
    if (k > 0 && t1 < t1_prev) return false;
    t1_prev = t1;
    int i1 = (int)floor(quot_dt(t1 - sp.t0, sp.dt));
    const int i2 = (int)floor(quot_dt(t2 - sp.t0, sp.dt));
    if (i1 > cur_end) {
      if (nseg == 0) { i1a = i1; na = 0; } else { i1b = i1; nb = 0; }
      nseg += 1;
      cur_start = i1;
    } else {
      i1 = cur_end + 1;
    }
    const int add = (i2 + 4 > i1) ? (i2 + 4 - i1) : 0;
    if (nseg == 1) na += add; else nb += add;
    cur_end = cur_start + (nseg == 1 ? na : nb) - 1;
  }
  s->nseg = nseg; s->i1[0] = i1a; s->i1[1] = i1b; s->n[0] = na; s->n[1] = nb;
  return true;
}
LVX_HD bool seg_lookup(const SplineRef& sp, const Segs& s, double t, KnotRef* out) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {   // constant indices: the two segments stay in registers
    if (k >= s.nseg) break;
    const int i1 = k == 0 ? s.i1[0] : s.i1[1], n = k == 0 ? s.n[0] : s.n[1];
    const double t0s = madd_2r(sp.dt, (double)i1, sp.t0);
    const double tmax = madd_2r((double)(n - 3), sp.dt, t0s);
    double te = t;
    bool in = (te >= t0s) && (te < tmax);
    if (!in) { te = t - 0.00001; in = (te >= t0s) && (te < tmax); }
    if (in) {
      const double sc = quot_dt(te - t0s, sp.dt);
      const int il = (int)floor(sc);
      if ((n < 4) || (il < 0) || (il > (n - 4))) return false;
      out->i0 = i1 + il;
      out->u = sc - (double)il;
      return true;
    }
  }
  return false;
}

// Two point spans {t_a, t_a}, {t_b, t_b} (locked time offset: the hub time and a measurement time) and the lookup of te_b = t_b + offset:
// exactly what build_segments + seg_lookup give for them, without building the segments, when the two 4-knot segments do not merge
// (i1_b > i1_a + 3) and te_b does not fall into the first one.  Returns 0 (kb set as seg_lookup would), 1 (build_segments would fail),
// 2 (seg_lookup(te_b) would fail), or -1: merged / overlapping segments — the caller takes the generic path.
LVX_HD int two_point_lookup(const SplineRef& sp, double t_a, double t_b, double te_b, KnotRef* kb) {
  if (sp.n < 4) return 1;
  const double tmin = sp.t0, tmax = madd_2r((double)(sp.n - 3), sp.dt, sp.t0);
  if ((t_a < tmin) || (t_a >= tmax)) return 1;
  if ((t_b < tmin) || (t_b >= tmax)) return 1;
  if (t_b < t_a) return 1;
  const int ia = (int)floor(quot_dt(t_a - sp.t0, sp.dt));
  const int ib = (int)floor(quot_dt(t_b - sp.t0, sp.dt));
  if (ib <= ia + 3) return -1;
  {
    const double t0s = madd_2r(sp.dt, (double)ia, sp.t0), tm = madd_2r((double)(4 - 3), sp.dt, t0s);
    double te = te_b;
    bool in = (te >= t0s) && (te < tm);
    if (!in) { te = te_b - 0.00001; in = (te >= t0s) && (te < tm); }
    if (in) return -1;
  }
  const double t0s = madd_2r(sp.dt, (double)ib, sp.t0), tm = madd_2r((double)(4 - 3), sp.dt, t0s);
  double te = te_b;
  bool in = (te >= t0s) && (te < tm);
  if (!in) { te = te_b - 0.00001; in = (te >= t0s) && (te < tm); }
  if (!in) return 2;
  const double sc = quot_dt(te - t0s, sp.dt);
  const int il = (int)floor(sc);
  if (il != 0) return 2;
  kb->i0 = ib + il;
  kb->u = sc - (double)il;
  return 0;
}

LVX_HD void load_so3_cp(const SplineRef& sp, int i0, quat c[4]) {
  for (int j = 0; j < 4; ++j) c[j] = load_q(sp.so3 + 4 * (i0 + j));
}

// pose (position + orientation) with Jacobian pieces, used by surfel / reprojection / cam-surfel
struct PoseEval {
  KnotRef k;
  v3 p;
  v3 v;              // world-frame velocity, only with NEED_V (time-offset Jacobians)
  double Bp[4];
  So3Eval so3;       // so3.w_body only with NEED_V
};
// NEED_V: also the time derivative of the pose — d p / d t = v, q(t + e) = q (x) Exp(w_body e) — which is what a free sensor time
// offset differentiates through (the reference's Jets carry it through the spline time argument, sensors.h:36-85)
template <bool NEED_J, bool NEED_V = false, bool PRE = false>
LVX_HD bool pose_eval(const SplineRef& sp, const KnotRef& k, PoseEval* out, const So3Pre* pre = nullptr) {
  out->k = k;
  R3Basis b; r3_basis(k.u, sp.dt, &b);
  v3 p = mk(0, 0, 0), v = mk(0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const v3 cj = load_v3(sp.r3 + 3 * (k.i0 + j));
    out->Bp[j] = b.Bp[j]; p = p + b.Bp[j] * cj;
    if (NEED_V) v = v + b.Bv[j] * cj;
  }
  out->p = p;
  if (NEED_V) out->v = v;
  quat c[4]; load_so3_cp(sp, k.i0, c);
  if (PRE) return so3_eval_pre<NEED_V, NEED_J, false>(c, pre, k.u, sp.dt, &out->so3) == 0;   // pre: the entries of control-point pairs (i0, i0+1) .. (i0+2, i0+3)
  return so3_eval<NEED_V, NEED_J, false>(c, k.u, sp.dt, &out->so3);
}
// pose with the SO3 part from precomputed control-point pairs (pre[0..2] = pairs (i0, i0+1) .. (i0+2, i0+3)); returns so3_eval_pre's status
template <bool NEED_J>
LVX_HD int pose_eval_pre(const SplineRef& sp, const KnotRef& k, PoseEval* out, const So3Pre* pre) {
  out->k = k;
  R3Basis b; r3_basis(k.u, sp.dt, &b);
  v3 p = mk(0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) { out->Bp[j] = b.Bp[j]; p = p + b.Bp[j] * load_v3(sp.r3 + 3 * (k.i0 + j)); }
  out->p = p;
  quat c[4]; load_so3_cp(sp, k.i0, c);
  return so3_eval_pre<false, NEED_J, false>(c, pre, k.u, sp.dt, &out->so3);
}
// window of precomputed control-point-pair quantities handed to the residuals of the fused kernels: entry e belongs to the pair
// (k0 + e, k0 + e + 1); a row whose knot interval falls outside is reported as RES_OUTSIDE (the kernel then takes the exact fallback)
#ifdef LVX_KTIME
struct PreWin { const So3Pre* p; int k0, n; long long* kt; };
#define LVX_KT(pw, i) { const long long n_ = __builtin_amdgcn_s_memtime(); (pw)->kt[i] += n_ - (pw)->kt[15]; (pw)->kt[15] = n_; }
#else
struct PreWin { const So3Pre* p; int k0, n; };
#define LVX_KT(pw, i)
#endif
enum { RES_OUTSIDE = 16 };   // = LVX_ERR_FALLBACK
LVX_HD const So3Pre* pre_at(const PreWin& w, int i0) { return (i0 >= w.k0 && i0 + 2 < w.k0 + w.n) ? w.p + (i0 - w.k0) : nullptr; }

// error codes shared with include/lvx.h
enum { RES_OK = 0, RES_RANGE = 1, RES_NONUNIT = 2 };

// ---------------------------------------------------------------------------------------------
// Gyroscope (measurements/gyroscope_measurement.h:36-38 -> sensors/constant_bias_imu.h:57-61 -> imu.h:87-91)
//   r = w (omega_meas - (q^* omega_world + b_g)) at t + tau_imu
// local columns: [SO3 knot j: 3j..3j+2 (j=0..3) | b_g: 12..14]; R3 / roll / pitch / b_a columns are structurally zero.
// ---------------------------------------------------------------------------------------------
enum { GYRO_NC = 15, GYRO_NR = 3 };
template <bool NEED_J, bool PRE = false>
LVX_HD int gyro_residual(const SplineRef& sp, const ImuCal& imu, double t, v3 w_meas, double weight, int* i0, double r[3], double J[3][GYRO_NC], const PreWin* pw = nullptr) {
  KnotRef k;
  if (!knot_lookup(sp.t0, sp.dt, sp.n, t, t + imu.tau, &k)) return RES_RANGE;
  *i0 = k.i0;
  quat c[4]; load_so3_cp(sp, k.i0, c);
  So3Eval e;
  if (PRE) {
    const So3Pre* pre = pre_at(*pw, k.i0);
    if (!pre) return RES_OUTSIDE;
    const int bad = so3_eval_pre<true, NEED_J>(c, pre, k.u, sp.dt, &e);
    if (bad) return (bad & 1) ? RES_NONUNIT : RES_OUTSIDE;
  } else if (!so3_eval<true, NEED_J>(c, k.u, sp.dt, &e)) return RES_NONUNIT;
  const v3 pred = e.w_body + imu.bg;
  r[0] = weight * (w_meas.x - pred.x); r[1] = weight * (w_meas.y - pred.y); r[2] = weight * (w_meas.z - pred.z);
  if (NEED_J) {
    for (int kk = 0; kk < 4; ++kk)
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) J[a][3 * kk + b] = -weight * e.dw[kk].a[3 * a + b];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) J[a][12 + b] = (a == b) ? -weight : 0.0;
  }
  return RES_OK;
}

// ---------------------------------------------------------------------------------------------
// Accelerometer (measurements/accelerometer_measurement.h:39-41 -> constant_bias_imu.h:51-55 -> imu.h:61-70,95-101)
//   r = w (a_meas - (q^* (p'' + g(roll,pitch)) + b_a))
// local columns: [knot j: R3 6j..6j+2, SO3 6j+3..6j+5 | roll 24 | pitch 25 | b_a 26..28]
// ---------------------------------------------------------------------------------------------
enum { ACC_NC = 29, ACC_NR = 3 };
template <bool NEED_J, bool PRE = false>
LVX_HD int accel_residual(const SplineRef& sp, const ImuCal& imu, double t, v3 a_meas, double weight, int* i0, double r[3], double J[3][ACC_NC], const PreWin* pw = nullptr) {
  KnotRef k;
  if (!knot_lookup(sp.t0, sp.dt, sp.n, t, t + imu.tau, &k)) return RES_RANGE;
  *i0 = k.i0;
  R3Basis b; r3_basis(k.u, sp.dt, &b);
  v3 acc = mk(0, 0, 0);
  for (int j = 0; j < 4; ++j) acc = acc + b.Ba[j] * load_v3(sp.r3 + 3 * (k.i0 + j));
  quat c[4]; load_so3_cp(sp, k.i0, c);
  So3Eval e;
  if (PRE) {
    const So3Pre* pre = pre_at(*pw, k.i0);
    if (!pre) return RES_OUTSIDE;
    const int bad = so3_eval_pre<false, NEED_J>(c, pre, k.u, sp.dt, &e);
    if (bad) return (bad & 1) ? RES_NONUNIT : RES_OUTSIDE;
  } else if (!so3_eval<false, NEED_J>(c, k.u, sp.dt, &e)) return RES_NONUNIT;
  const double G = -9.79;   // imu.h:25
  const double cr = cos(imu.roll), sr = sin(imu.roll), cp = cos(imu.pitch), sp_ = sin(imu.pitch);
  const v3 g = mk(-sp_ * cr * G, sr * G, -cr * cp * G);
  const v3 y = acc + g;
  const v3 yb = qrot_inv(e.q, y);
  const v3 pred = yb + imu.ba;
  r[0] = weight * (a_meas.x - pred.x); r[1] = weight * (a_meas.y - pred.y); r[2] = weight * (a_meas.z - pred.z);
  if (NEED_J) {
    const m3 Rt = transpose(rotmat(e.q));
    const m3 S = skew(yb);
    for (int kk = 0; kk < 4; ++kk) {
      const m3 Mx = S * e.dxi[kk];      // d(R^T y)/d delta_k = [R^T y]x dxi_k
      for (int a = 0; a < 3; ++a)
        for (int bb = 0; bb < 3; ++bb) {
          J[a][6 * kk + bb] = -weight * b.Ba[kk] * Rt.a[3 * a + bb];
          J[a][6 * kk + 3 + bb] = -weight * Mx.a[3 * a + bb];
        }
    }
    const v3 dg_dr = Rt * mk(sp_ * sr * G, cr * G, sr * cp * G);
    const v3 dg_dp = Rt * mk(-cp * cr * G, 0.0, cr * sp_ * G);
    for (int a = 0; a < 3; ++a) {
      J[a][24] = -weight * comp(dg_dr, a);
      J[a][25] = -weight * comp(dg_dp, a);
      for (int bb = 0; bb < 3; ++bb) J[a][26 + bb] = (a == bb) ? -weight : 0.0;
    }
  }
  return RES_OK;
}

// plane in closest-point form Pi = d n (lidar_surfel_point.h:54-63)
LVX_HD void plane_nd(v3 Pi, v3* n, double* d) {
  const double pd = sqrt(Pi.x * Pi.x + Pi.y * Pi.y + Pi.z * Pi.z);
  *d = pd; *n = mk(Pi.x / pd, Pi.y / pd, Pi.z / pd);
}

// ---------------------------------------------------------------------------------------------
// LiDAR point-to-surfel (measurements/lidar_surfel_point.h:31-82)
//   p_I = q_L p_L + p_LI ; p_tmp = q0^* (qk p_I + pk - p0) ; p_M = q_L^* (p_tmp - p_LI) ; r = w (n.p_M - d)
// two pose evaluations: hub (t_map + tau_L) and k (t_k + tau_L).
// local columns: [hub knot j: 6j..6j+5 | k knot j: 24+6j.. | lidar theta 48..50 | lidar p 51..53]
// The hub evaluation is identical for every surfel residual and is passed in precomputed.
// ---------------------------------------------------------------------------------------------
enum { SURF_NC = 54, SURF_NR = 1 };   // + 1 column (lidar time offset) in the TAU variants

// shared tail of surfel / cam-surfel: given p_I (point in IMU frame at time k) and hub/k poses
struct PlaneChain { v3 nL, m, x, ptemp; double r_unweighted; };
LVX_HD void plane_chain(const PoseEval& h, const PoseEval& k, const SensorCal& lidar, v3 p_I, v3 Pi, PlaneChain* o) {
  v3 n; double d; plane_nd(Pi, &n, &d);
  const v3 s = qrot(k.so3.q, p_I) + k.p - h.p;
  o->ptemp = qrot_inv(h.so3.q, s);
  o->x = o->ptemp - lidar.p;
  const v3 pM = qrot_inv(lidar.q, o->x);
  o->r_unweighted = (n.x * pM.x + n.y * pM.y + n.z * pM.z) - d;
  o->nL = qrot(lidar.q, n);                                   // n^T R_L^T = (R_L n)^T
  o->m = qrot_inv(k.so3.q, qrot(h.so3.q, o->nL));            // Rk^T R0 nL
}
// gradients of a two-pose point-to-plane residual w.r.t. the poses: gp = d r / d p_k (= - d r / d p_0), gx0 = d r / d xi_0, gxk = d r / d xi_k
struct PlaneGrads { v3 gp, gx0, gxk; };
LVX_HD PlaneGrads plane_grads(const PoseEval& h, const PlaneChain& pc, v3 p_I, double w) {
  PlaneGrads g;
  g.gp = w * qrot(h.so3.q, pc.nL);
  g.gx0 = w * cross(pc.nL, pc.ptemp);
  g.gxk = w * cross(p_I, pc.m);
  return g;
}
// expand pose gradients to the 4 control points of one evaluation: position weight Bp[j] * gpos, rotation dxi[j]^T gxi
LVX_HD void pose_to_knots(const PoseEval& e, v3 gpos, v3 gxi, double* J24) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    J24[6 * j + 0] = e.Bp[j] * gpos.x; J24[6 * j + 1] = e.Bp[j] * gpos.y; J24[6 * j + 2] = e.Bp[j] * gpos.z;
    const v3 a = tmulv(e.so3.dxi[j], gxi);
    J24[6 * j + 3] = a.x; J24[6 * j + 4] = a.y; J24[6 * j + 5] = a.z;
  }
}
// knot columns of a two-pose point-to-plane residual: d r / d(hub knots), d r / d(k knots)
LVX_HD void plane_knot_jac(const PoseEval& h, const PoseEval& k, const PlaneChain& pc, v3 p_I, double w, double* Jhub, double* Jk) {
  const PlaneGrads g = plane_grads(h, pc, p_I, w);
  pose_to_knots(h, -g.gp, g.gx0, Jhub);
  pose_to_knots(k, g.gp, g.gxk, Jk);
}

// time-offset column of a two-pose point-to-plane residual: both poses move with tau
LVX_HD double plane_tau_jac(const PoseEval& h, const PoseEval& k, const PlaneGrads& g) {
  return dot(g.gp, k.v - h.v) + dot(g.gx0, h.so3.w_body) + dot(g.gxk, k.so3.w_body);
}

// TAU: the hub must have been evaluated with NEED_V; adds column SURF_NC = d r / d tau_lidar
template <bool NEED_J, bool TAU = false>
LVX_HD int surfel_residual(const SplineRef& sp, const PoseEval& hub, const Segs& segs, const SensorCal& lidar, double t_k, v3 p_L, v3 Pi,
                           double weight, int* i0_k, double r[1], double J[1][SURF_NC + (TAU ? 1 : 0)]) {
  KnotRef kr;
  if (!seg_lookup(sp, segs, t_k + lidar.tau, &kr)) return RES_RANGE;
  *i0_k = kr.i0;
  PoseEval k;
  if (!pose_eval<NEED_J, TAU>(sp, kr, &k)) return RES_NONUNIT;
  const v3 pLr = qrot(lidar.q, p_L);
  const v3 p_I = pLr + lidar.p;
  PlaneChain pc; plane_chain(hub, k, lidar, p_I, Pi, &pc);
  r[0] = weight * pc.r_unweighted;
  if (NEED_J) {
    plane_knot_jac(hub, k, pc, p_I, weight, &J[0][0], &J[0][24]);
    const v3 jq = (2.0 * weight) * (cross(pc.nL, pc.x) - cross(pc.m, pLr));
    const v3 jp = weight * (pc.m - pc.nL);
    J[0][48] = jq.x; J[0][49] = jq.y; J[0][50] = jq.z;
    J[0][51] = jp.x; J[0][52] = jp.y; J[0][53] = jp.z;
    if (TAU) J[0][SURF_NC] = plane_tau_jac(hub, k, plane_grads(hub, pc, p_I, weight));
  }
  return RES_OK;
}

// Same residual with the hub pose represented by 6 PSEUDO variables (d p_0 (3), xi_0 (3)) instead of 24 hub-knot columns:
// J_hub = g0^T M_hub with M_hub identical for every residual of a launch, so J^T J is assembled over g0 and folded back
// with M_hub afterwards (lvx_eval.hip: k_fold_border).  local columns: [k knot j: 6j.. (24) | g0 24..29 | lidar theta 30..32 | lidar p 33..35]
enum { SURFP_NC = 36 };
// pose value + what the reverse-mode knot gradients need (no 3x3 Jacobian blocks): used by the single-row residuals of the fused kernels
struct PoseVal { v3 p; double Bp[4]; quat c[4]; So3Val s; };
LVX_HD int pose_value_pre(const SplineRef& sp, const KnotRef& k, const So3Pre* pre, PoseVal* out) {   // 0 | 1 non-unit | 2 large angle (so3_value_pre)
  R3Basis b; r3_basis(k.u, sp.dt, &b);
  v3 p = mk(0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) { out->Bp[j] = b.Bp[j]; p = p + b.Bp[j] * load_v3(sp.r3 + 3 * (k.i0 + j)); }
  out->p = p;
  load_so3_cp(sp, k.i0, out->c);
  return so3_value_pre(out->c, pre, k.u, &out->s);
}
// knot columns of a scalar residual with pose gradients (gpos, gxi): position weights and the SO3 pull-back
LVX_HD void pose_pull_to_knots(const PoseVal& e, const So3Pre* pre, v3 gpos, v3 gxi, double* J24) {
  v3 y[4];
  so3_pullback_pre(e.c, pre, e.s, gxi, y);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    J24[6 * j + 0] = e.Bp[j] * gpos.x; J24[6 * j + 1] = e.Bp[j] * gpos.y; J24[6 * j + 2] = e.Bp[j] * gpos.z;
    J24[6 * j + 3] = y[j].x; J24[6 * j + 4] = y[j].y; J24[6 * j + 5] = y[j].z;
  }
}

// time derivative of the pose at a knot reference — world velocity and body angular velocity — for the fused kernels' time-offset column (the value path above
// carries no derivative): the R3 velocity basis and one more pass through the SO3 chain without Jacobians
template <bool PRE>
LVX_HD int pose_twist(const SplineRef& sp, const KnotRef& k, const So3Pre* pre, v3* v, v3* w_body) {
  R3Basis b; r3_basis(k.u, sp.dt, &b);
  v3 vv = mk(0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) vv = vv + b.Bv[j] * load_v3(sp.r3 + 3 * (k.i0 + j));
  *v = vv;
  quat c[4]; load_so3_cp(sp, k.i0, c);
  So3Eval e;
  if (PRE) { const int bad = so3_eval_pre<true, false, false>(c, pre, k.u, sp.dt, &e); if (bad) return bad; }
  else if (!so3_eval<true, false, false>(c, k.u, sp.dt, &e)) return 1;
  *w_body = e.w_body;
  return 0;
}
// TAU: one more column, J[0][SURFP_NC] = d r / d tau_lidar — both poses move with the offset (plane_tau_jac); the hub must carry v and w_body (pose_eval<.., NEED_V>)
template <bool PRE = false, bool TAU = false>
LVX_HD int surfel_residual_pseudo(const SplineRef& sp, const PoseEval& hub, const Segs& segs, const SensorCal& lidar, double t_k, v3 p_L, v3 Pi,
                                  double weight, int* i0_k, double r[1], double J[1][SURFP_NC + (TAU ? 1 : 0)], const PreWin* pw = nullptr, const KnotRef* kr_in = nullptr) {
  KnotRef kr;
  if (kr_in) kr = *kr_in;   // the caller has done the lookup (two_point_lookup)
  else if (!seg_lookup(sp, segs, t_k + lidar.tau, &kr)) return RES_RANGE;
  *i0_k = kr.i0;
  PoseEval k;
  PoseVal kv;
  const So3Pre* pre = PRE ? pre_at(*pw, kr.i0) : nullptr;
  if (PRE && !pre) return RES_OUTSIDE;
  LVX_KT(pw, 9)
  if (PRE) { const int bad = pose_value_pre(sp, kr, pre, &kv); if (bad) return (bad & 1) ? RES_NONUNIT : RES_OUTSIDE; k.p = kv.p; k.so3.q = kv.s.q; }
  else if (!pose_eval<true, false, false>(sp, kr, &k)) return RES_NONUNIT;
  LVX_KT(pw, 10)
  const v3 pLr = qrot(lidar.q, p_L);
  const v3 p_I = pLr + lidar.p;
  PlaneChain pc; plane_chain(hub, k, lidar, p_I, Pi, &pc);
  r[0] = weight * pc.r_unweighted;
  const PlaneGrads g = plane_grads(hub, pc, p_I, weight);
  LVX_KT(pw, 11)
  if (PRE) pose_pull_to_knots(kv, pre, g.gp, g.gxk, &J[0][0]); else pose_to_knots(k, g.gp, g.gxk, &J[0][0]);
  LVX_KT(pw, 12)
  J[0][24] = -g.gp.x; J[0][25] = -g.gp.y; J[0][26] = -g.gp.z; J[0][27] = g.gx0.x; J[0][28] = g.gx0.y; J[0][29] = g.gx0.z;
  const v3 jq = (2.0 * weight) * (cross(pc.nL, pc.x) - cross(pc.m, pLr));
  const v3 jp = weight * (pc.m - pc.nL);
  J[0][30] = jq.x; J[0][31] = jq.y; J[0][32] = jq.z; J[0][33] = jp.x; J[0][34] = jp.y; J[0][35] = jp.z;
  if (TAU) {
    v3 vk, wk;
    const int bad = pose_twist<PRE>(sp, kr, pre, &vk, &wk);
    if (bad) return (bad & 1) ? RES_NONUNIT : RES_OUTSIDE;
    J[0][SURFP_NC] = dot(g.gp, vk - hub.v) + dot(g.gx0, hub.so3.w_body) + dot(g.gxk, wk);
  }
  return RES_OK;
}
// M_hub (6 x 24): pseudo pose perturbation (d p_0, xi_0) per unit tangent of the 4 hub control points
LVX_HD void hub_matrix(const PoseEval& h, double M[6][24]) {
  for (int a = 0; a < 6; ++a) for (int c = 0; c < 24; ++c) M[a][c] = 0.0;
  for (int j = 0; j < 4; ++j) {
    for (int a = 0; a < 3; ++a) M[a][6 * j + a] = h.Bp[j];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) M[3 + a][6 * j + 3 + b] = h.so3.dxi[j].a[3 * a + b];
  }
}

// ---------------------------------------------------------------------------------------------
// Pinhole camera (sensors/pinhole_camera.h:96-238)
// ---------------------------------------------------------------------------------------------
LVX_HD void cam_distortion(const CamIntr& c, double x, double y, double* dx, double* dy, double D[4]) {
  const double mx2 = x * x, my2 = y * y, mxy = x * y, rho2 = mx2 + my2;
  const double rad = c.k1 * rho2 + c.k2 * rho2 * rho2 + c.k3 * rho2 * rho2 * rho2;
  *dx = x * rad + 2.0 * c.p1 * mxy + c.p2 * (rho2 + 2.0 * mx2);
  *dy = y * rad + 2.0 * c.p2 * mxy + c.p1 * (rho2 + 2.0 * my2);
  if (D) {
    const double drad = c.k1 + 2.0 * c.k2 * rho2 + 3.0 * c.k3 * rho2 * rho2;   // d rad / d rho2
    D[0] = rad + x * drad * 2.0 * x + 2.0 * c.p1 * y + c.p2 * (2.0 * x + 4.0 * x);
    D[1] = x * drad * 2.0 * y + 2.0 * c.p1 * x + c.p2 * (2.0 * y);
    D[2] = y * drad * 2.0 * x + 2.0 * c.p2 * y + c.p1 * (2.0 * x);
    D[3] = rad + y * drad * 2.0 * y + 2.0 * c.p2 * x + c.p1 * (2.0 * y + 4.0 * y);
  }
}
LVX_HD v3 cam_unproject(const CamIntr& c, double u, double v) {   // :113-124, :131-191
  if (c.do_distortion) {
    const double mx_d = c.inv_K11 * u + c.inv_K13, my_d = c.inv_K22 * v + c.inv_K23;
    double dx, dy;
    cam_distortion(c, mx_d, my_d, &dx, &dy, nullptr);
    double mx_u = mx_d - dx, my_u = my_d - dy;
    for (int i = 1; i < 8; ++i) { cam_distortion(c, mx_u, my_u, &dx, &dy, nullptr); mx_u = mx_d - dx; my_u = my_d - dy; }
    return mk(mx_u, my_u, 1.0);
  }
  // Matrix3::inverse() of K = [fx 0 cx; 0 fy cy; 0 0 1] (cofactor / determinant form)
  const double invdet = 1.0 / (c.fx * c.fy);
  return mk((c.fy * invdet) * u + ((-c.cx * c.fy) * invdet), (c.fx * invdet) * v + ((-(c.fx * c.cy)) * invdet), (c.fx * c.fy) * invdet);
}
// spaceToPlane (:217-238): y = K (p_u + d(p_u)); G = d y / d X (2x3)
LVX_HD void cam_project(const CamIntr& c, v3 X, double y[2], double G[2][3]) {
  const double z = 1e-32 + X.z;
  const double pu0 = X.x / z, pu1 = X.y / z;
  double pd0 = pu0, pd1 = pu1;
  double D[4] = {1.0, 0.0, 0.0, 1.0};
  if (c.do_distortion) {
    double dx, dy, Dd[4];
    cam_distortion(c, pu0, pu1, &dx, &dy, Dd);
    pd0 = pu0 + dx; pd1 = pu1 + dy;
    D[0] = 1.0 + Dd[0]; D[1] = Dd[1]; D[2] = Dd[2]; D[3] = 1.0 + Dd[3];
  }
  y[0] = c.fx * pd0 + c.cx;
  y[1] = c.fy * pd1 + c.cy;
  if (G) {
    // d p_u / d X = [1/z 0 -x/z^2; 0 1/z -y/z^2]
    const double iz = 1.0 / z;
    const double a0[3] = {iz, 0.0, -pu0 * iz}, a1[3] = {0.0, iz, -pu1 * iz};
    for (int k = 0; k < 3; ++k) {
      G[0][k] = c.fx * (D[0] * a0[k] + D[1] * a1[k]);
      G[1][k] = c.fy * (D[2] * a0[k] + D[3] * a1[k]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Rolling-shutter reprojection (measurements/static_rscamera_measurement.h:20-60, wiring :135-203)
// local columns: [ref knot j: 6j.. | obs knot j: 24+6j.. | cam theta 48..50 | cam p 51..53 | rho 54]
// ---------------------------------------------------------------------------------------------
enum { REP_NC = 55, REP_NR = 2 };
// gpre (fused path, !TAU): the pass's table of control-point-pair quantities, entry k = pair (k, k+1); the poses then take so3_eval_pre
// (small-angle polynomials; a pair beyond them returns RES_OUTSIDE and the pass falls back to the exact kernels)
template <bool NEED_J, bool TAU = false>
LVX_HD int reproj_residual(const SplineRef& sp, const CamIntr& ci, const SensorCal& cam, bool tau_locked, double max_time_offset,
                           double u_ref, double v_ref, double t0_ref, double u_obs, double v_obs, double t0_obs, double rho, double weight,
                           int* i0_ref, int* i0_obs, double r[2], double J[2][REP_NC + (TAU ? 1 : 0)], const So3Pre* gpre = nullptr) {
  // spans (:148-172): sorted (t0_ref, t0_obs), padded by the time-offset bound if free, then [-1e-3, readout + 1e-3]
  double t1, t2;
  if (t0_ref <= t0_obs) { t1 = t0_ref; t2 = t0_obs; } else { t1 = t0_obs; t2 = t0_ref; }
  if (!tau_locked) { t1 -= max_time_offset; t2 += max_time_offset; }
  const double margin = 1e-3;
  const double spans[2][2] = {{t1 - margin, t1 + ci.readout + margin}, {t2 - margin, t2 + ci.readout + margin}};
  Segs segs;
  if (!build_segments(sp, spans, 2, &segs)) return RES_RANGE;
  const double row_delta = ci.readout / (double)ci.rows;
  const double t_ref = madd_2r(v_ref, row_delta, t0_ref + cam.tau);   // (t0 + tau) + v * row_delta, three roundings like the reference's expression (:32-33)
  const double t_obs = madd_2r(v_obs, row_delta, t0_obs + cam.tau);
  KnotRef kr, ko;
  if (!seg_lookup(sp, segs, t_ref, &kr)) return RES_RANGE;
  if (!seg_lookup(sp, segs, t_obs, &ko)) return RES_RANGE;
  *i0_ref = kr.i0; *i0_obs = ko.i0;
  PoseEval er, eo;
  if (!TAU && gpre) {
    const int bad = pose_eval_pre<NEED_J>(sp, kr, &er, gpre + kr.i0) | pose_eval_pre<NEED_J>(sp, ko, &eo, gpre + ko.i0);
    if (bad) return (bad & 1) ? RES_NONUNIT : RES_OUTSIDE;
  } else {
    if (!pose_eval<NEED_J, TAU>(sp, kr, &er)) return RES_NONUNIT;
    if (!pose_eval<NEED_J, TAU>(sp, ko, &eo)) return RES_NONUNIT;
  }
  const v3 p_ct = qrot_inv(cam.q, -cam.p);
  const v3 yh = cam_unproject(ci, u_ref, v_ref);
  const v3 X_ref = qrot(cam.q, yh - rho * p_ct);
  const v3 X = qrot(er.so3.q, X_ref) + rho * er.p;
  const v3 X_obs = qrot_inv(eo.so3.q, X - rho * eo.p);
  const v3 X_c = qrot_inv(cam.q, X_obs) + rho * p_ct;
  double yhat[2], G[2][3];
  cam_project(ci, X_c, yhat, NEED_J ? G : nullptr);
  r[0] = weight * (u_obs - yhat[0]);
  r[1] = weight * (v_obs - yhat[1]);
  if (NEED_J) {
    const m3 RC = rotmat(cam.q), Rr = rotmat(er.so3.q), Ro = rotmat(eo.so3.q);
    // A = d X_c / d X_obs = RC^T ; chain matrices (3x3)
    const m3 RCt = transpose(RC);
    const m3 A_obs = RCt;                              // d Xc / d Xobs
    const m3 A_X = RCt * transpose(Ro);                // d Xc / d X
    const m3 A_Xref = A_X * Rr;                        // d Xc / d Xref
    const m3 dXc_dxo = A_obs * skew(X_obs);            // d Xobs / d xi_o = [Xobs]x
    const m3 dXc_dxr = -1.0 * (A_Xref * skew(X_ref));  // d X / d xi_r = -Rr [Xref]x
    const v3 RCyh = qrot(cam.q, yh);
    const v3 xo = X_obs - rho * cam.p;
    const m3 dXc_deps = RCt * skew(xo) - A_Xref * skew(RCyh);
    const m3 dXc_dpC = rho * (A_Xref - RCt);
    const v3 dXc_drho = RCt * (tmulv(Ro, (Rr * cam.p) + er.p - eo.p) - cam.p);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const v3 g = mk(-weight * G[a][0], -weight * G[a][1], -weight * G[a][2]);   // d r_a / d Xc
      const v3 gX = tmulv(A_X, g);               // d r / d X
      const v3 gxr = tmulv(dXc_dxr, g), gxo = tmulv(dXc_dxo, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double br = rho * er.Bp[j], bo = -rho * eo.Bp[j];
        J[a][6 * j + 0] = br * gX.x; J[a][6 * j + 1] = br * gX.y; J[a][6 * j + 2] = br * gX.z;
        const v3 sr = tmulv(er.so3.dxi[j], gxr);
        J[a][6 * j + 3] = sr.x; J[a][6 * j + 4] = sr.y; J[a][6 * j + 5] = sr.z;
        J[a][24 + 6 * j + 0] = bo * gX.x; J[a][24 + 6 * j + 1] = bo * gX.y; J[a][24 + 6 * j + 2] = bo * gX.z;
        const v3 so = tmulv(eo.so3.dxi[j], gxo);
        J[a][24 + 6 * j + 3] = so.x; J[a][24 + 6 * j + 4] = so.y; J[a][24 + 6 * j + 5] = so.z;
      }
      const v3 jq = 2.0 * tmulv(dXc_deps, g);
      const v3 jp = tmulv(dXc_dpC, g);
      J[a][48] = jq.x; J[a][49] = jq.y; J[a][50] = jq.z;
      J[a][51] = jp.x; J[a][52] = jp.y; J[a][53] = jp.z;
      J[a][54] = dot(g, dXc_drho);
      if (TAU) J[a][REP_NC] = rho * dot(gX, er.v - eo.v) + dot(gxr, er.so3.w_body) + dot(gxo, eo.so3.w_body);   // both views move with tau_cam
    }
  }
  return RES_OK;
}

// ---------------------------------------------------------------------------------------------
// Camera-landmark-to-surfel (measurements/camera_surfel_landmark.h:29-103): back-project the reference
// observation at depth 1/(rho + 1e-8) (rho read as a CONSTANT, :159-161), then point-to-plane as the surfel.
// local columns: [hub knot j: 6j.. | k knot j: 24+6j.. | cam theta 48..50 | cam p 51..53 | lidar theta 54..56 | lidar p 57..59]
// ---------------------------------------------------------------------------------------------
enum { CS_NC = 60, CS_NR = 1 };
template <bool NEED_J, bool TAU = false>
LVX_HD int camsurf_residual(const SplineRef& sp, const PoseEval& hub, const Segs& segs, const CamIntr& ci, const SensorCal& cam, const SensorCal& lidar,
                            double u_ref, double v_ref, double t0_ref, double rho, v3 Pi, double weight, int* i0_k, double r[1], double J[1][CS_NC + (TAU ? 1 : 0)]) {
  KnotRef kr;
  if (!seg_lookup(sp, segs, t0_ref + cam.tau, &kr)) return RES_RANGE;
  *i0_k = kr.i0;
  PoseEval k;
  if (!pose_eval<NEED_J, TAU>(sp, kr, &k)) return RES_NONUNIT;
  const double s = 1.0 / (rho + 1e-8);
  const v3 yu = cam_unproject(ci, u_ref, v_ref);
  const v3 yh = mk(yu.x * s, yu.y * s, yu.z * s);
  const v3 RCyh = qrot(cam.q, yh);
  const v3 p_I = RCyh + cam.p;
  PlaneChain pc; plane_chain(hub, k, lidar, p_I, Pi, &pc);
  r[0] = weight * pc.r_unweighted;
  if (NEED_J) {
    plane_knot_jac(hub, k, pc, p_I, weight, &J[0][0], &J[0][24]);
    const v3 jcq = (-2.0 * weight) * cross(pc.m, RCyh);
    const v3 jcp = weight * pc.m;
    const v3 jlq = (2.0 * weight) * cross(pc.nL, pc.x);
    const v3 jlp = (-weight) * pc.nL;
    J[0][48] = jcq.x; J[0][49] = jcq.y; J[0][50] = jcq.z; J[0][51] = jcp.x; J[0][52] = jcp.y; J[0][53] = jcp.z;
    J[0][54] = jlq.x; J[0][55] = jlq.y; J[0][56] = jlq.z; J[0][57] = jlp.x; J[0][58] = jlp.y; J[0][59] = jlp.z;
    if (TAU) J[0][CS_NC] = plane_tau_jac(hub, k, plane_grads(hub, pc, p_I, weight));   // tau_cam moves both the hub and the landmark's reference pose
  }
  return RES_OK;
}

// pseudo-hub variant: local columns [k knot j: 6j.. (24) | g0 24..29 | cam theta 30..32 | cam p 33..35 | lidar theta 36..38 | lidar p 39..41]
enum { CSP_NC = 42 };
// TAU: one more column, J[0][CSP_NC] = d r / d tau_cam (the hub and the landmark's reference pose both move with it)
template <bool PRE = false, bool TAU = false>
LVX_HD int camsurf_residual_pseudo(const SplineRef& sp, const PoseEval& hub, const Segs& segs, const CamIntr& ci, const SensorCal& cam, const SensorCal& lidar,
                                   double u_ref, double v_ref, double t0_ref, double rho, v3 Pi, double weight, int* i0_k, double r[1], double J[1][CSP_NC + (TAU ? 1 : 0)], const PreWin* pw = nullptr,
                                   const KnotRef* kr_in = nullptr) {
  KnotRef kr;
  if (kr_in) kr = *kr_in;   // the caller has done the lookup (two_point_lookup)
  else if (!seg_lookup(sp, segs, t0_ref + cam.tau, &kr)) return RES_RANGE;
  *i0_k = kr.i0;
  PoseEval k;
  PoseVal kv;
  const So3Pre* pre = PRE ? pre_at(*pw, kr.i0) : nullptr;
  if (PRE && !pre) return RES_OUTSIDE;
  if (PRE) { const int bad = pose_value_pre(sp, kr, pre, &kv); if (bad) return (bad & 1) ? RES_NONUNIT : RES_OUTSIDE; k.p = kv.p; k.so3.q = kv.s.q; }
  else if (!pose_eval<true, false, false>(sp, kr, &k)) return RES_NONUNIT;
  const double s = 1.0 / (rho + 1e-8);
  const v3 yu = cam_unproject(ci, u_ref, v_ref);
  const v3 yh = mk(yu.x * s, yu.y * s, yu.z * s);
  const v3 RCyh = qrot(cam.q, yh);
  const v3 p_I = RCyh + cam.p;
  PlaneChain pc; plane_chain(hub, k, lidar, p_I, Pi, &pc);
  r[0] = weight * pc.r_unweighted;
  const PlaneGrads g = plane_grads(hub, pc, p_I, weight);
  if (PRE) pose_pull_to_knots(kv, pre, g.gp, g.gxk, &J[0][0]); else pose_to_knots(k, g.gp, g.gxk, &J[0][0]);
  J[0][24] = -g.gp.x; J[0][25] = -g.gp.y; J[0][26] = -g.gp.z; J[0][27] = g.gx0.x; J[0][28] = g.gx0.y; J[0][29] = g.gx0.z;
  const v3 jcq = (-2.0 * weight) * cross(pc.m, RCyh);
  const v3 jcp = weight * pc.m;
  const v3 jlq = (2.0 * weight) * cross(pc.nL, pc.x);
  const v3 jlp = (-weight) * pc.nL;
  J[0][30] = jcq.x; J[0][31] = jcq.y; J[0][32] = jcq.z; J[0][33] = jcp.x; J[0][34] = jcp.y; J[0][35] = jcp.z;
  J[0][36] = jlq.x; J[0][37] = jlq.y; J[0][38] = jlq.z; J[0][39] = jlp.x; J[0][40] = jlp.y; J[0][41] = jlp.z;
  if (TAU) {
    v3 vk, wk;
    const int bad = pose_twist<PRE>(sp, kr, pre, &vk, &wk);
    if (bad) return (bad & 1) ? RES_NONUNIT : RES_OUTSIDE;
    J[0][CSP_NC] = dot(g.gp, vk - hub.v) + dot(g.gx0, hub.so3.w_body) + dot(g.gxk, wk);
  }
  return RES_OK;
}

// ---------------------------------------------------------------------------------------------
// Orientation prior (measurements/orientation_measurement.h:30-33): r = w * angularDistance(q_meas, q(t))
// Eigen angularDistance: d = q_meas * q^*, 2 atan2(|vec d|, |d.w|).  local columns: [SO3 knot j: 3j..3j+2]
// ---------------------------------------------------------------------------------------------
enum { PRI_NC = 12, PRI_NR = 1 };
template <bool NEED_J>
LVX_HD int prior_residual(const SplineRef& sp, double t, quat q_meas, double weight, int* i0, double r[1], double J[1][PRI_NC]) {
  KnotRef k;
  if (!knot_lookup(sp.t0, sp.dt, sp.n, t, t, &k)) return RES_RANGE;
  *i0 = k.i0;
  quat c[4]; load_so3_cp(sp, k.i0, c);
  So3Eval e;
  if (!so3_eval<false, NEED_J>(c, k.u, sp.dt, &e)) return RES_NONUNIT;
  const quat d = qmul(q_meas, qconj(e.q));
  const double vn = sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
  r[0] = weight * (2.0 * atan2(vn, fabs(d.w)));
  if (NEED_J) {
    // theta(d Exp(zeta)) ~ theta + axis . zeta, zeta = -R(q) xi ; axis taken with d.w >= 0
    // at vn == 0 the distance has a kink (|x|); autodiff yields 0/0 there.  LM reaches it exactly when the prior is the only gauge
    // constraint (Solve #0), so the zero sub-gradient is used instead of propagating NaN into J^T J
    const double sgn = d.w < 0.0 ? -1.0 : 1.0;
    const double ivn = vn > 0.0 ? 1.0 / vn : 0.0;
    const v3 axis = mk(sgn * d.x * ivn, sgn * d.y * ivn, sgn * d.z * ivn);
    // d * (q Exp(-xi) q^*) : world-frame increment -R xi applied on the right of d
    const v3 gxi = (-weight) * qrot_inv(e.q, axis);
    for (int j = 0; j < 4; ++j) {
      const v3 a = tmulv(e.dxi[j], gxi);
      J[0][3 * j + 0] = a.x; J[0][3 * j + 1] = a.y; J[0][3 * j + 2] = a.z;
    }
  }
  return RES_OK;
}

// ---------------------------------------------------------------------------------------------
// local column -> global tangent index.  Tangent layout (include/lvx.h):
//   knot k: 6k..6k+2 position, 6k+3..6k+5 rotation | calib base C = 6N: roll, pitch, b_a(3), b_g(3),
//   lidar theta(3) p(3) tau, cam theta(3) p(3) tau | rho_l at 6N + 22 + l
// ---------------------------------------------------------------------------------------------
LVX_HD int gyro_col(int c, int i0, int N) { return c < 12 ? 6 * (i0 + c / 3) + 3 + c % 3 : 6 * N + 5 + (c - 12); }
LVX_HD int acc_col(int c, int i0, int N) { return c < 24 ? 6 * (i0 + c / 6) + c % 6 : (c < 26 ? 6 * N + (c - 24) : 6 * N + 2 + (c - 26)); }
LVX_HD int surf_col(int c, int i0h, int i0k, int N) {
  return c < 24 ? 6 * (i0h + c / 6) + c % 6 : (c < 48 ? 6 * (i0k + (c - 24) / 6) + (c - 24) % 6 : 6 * N + 8 + (c - 48)); }   // c == 54: lidar tau
LVX_HD int rep_col(int c, int i0r, int i0o, int N, int lm) {
  return c < 24 ? 6 * (i0r + c / 6) + c % 6 : (c < 48 ? 6 * (i0o + (c - 24) / 6) + (c - 24) % 6 : (c < 54 ? 6 * N + 15 + (c - 48) : (c == 54 ? 6 * N + 22 + lm : 6 * N + 21))); }   // c == 55: cam tau
LVX_HD int cs_col(int c, int i0h, int i0k, int N) {
  return c < 24 ? 6 * (i0h + c / 6) + c % 6 : (c < 48 ? 6 * (i0k + (c - 24) / 6) + (c - 24) % 6 : (c < 54 ? 6 * N + 15 + (c - 48) : (c < 60 ? 6 * N + 8 + (c - 54) : 6 * N + 21))); }   // c == 60: cam tau
LVX_HD int pri_col(int c, int i0, int N) { return 6 * (i0 + c / 3) + 3 + c % 3; }

// lock mask (include/lvx.h LVX_LOCK_*) -> is global tangent index g constant?
LVX_HD bool tangent_locked(int g, int N, int n_lm, uint32_t locks) {
  if (g < 6 * N) { if (locks & 1u) return true; return (g % 6) < 3 && (locks & 2u); }
  const int c = g - 6 * N;
  if (c < 2) return false;                       // gravity roll / pitch: never constant (imu.h:135-141)
  if (c < 5) return (locks & (1u << 8)) != 0;    // b_a
  if (c < 8) return (locks & (1u << 9)) != 0;    // b_g
  if (c < 11) return (locks & (1u << 2)) != 0;   // lidar q
  if (c < 14) return (locks & (1u << 3)) != 0;   // lidar p
  if (c < 15) return (locks & (1u << 4)) != 0;   // lidar tau
  if (c < 18) return (locks & (1u << 5)) != 0;   // cam q
  if (c < 21) return (locks & (1u << 6)) != 0;   // cam p
  if (c < 22) return (locks & (1u << 7)) != 0;   // cam tau
  return (locks & (1u << 10)) != 0;              // landmarks
}

// Huber (ceres::HuberLoss + Corrector, restated): returns rho(s); *scale = sqrt(rho'(s)) applied to r and J
LVX_HD double huber_rho(double a, double s, double* scale) {
  const double b = a * a;
  if (a > 0.0 && s > b) { const double r = sqrt(s); *scale = sqrt(a / r); return 2.0 * a * r - b; }
  *scale = 1.0; return s;
}

}  // namespace lvx
