// tests/native/resid_host_check.cpp — TEST-ONLY host build of the device math (lvx_math.h / lvx_resid.h).
//
// Compiles the __host__ __device__ residual + analytic-Jacobian functions with g++ so that the CPU test
// suite (-m "not gpu") can compare them against the oracle's dual-number Jacobians without a GPU.  This is
// not a CPU fallback: nothing here is linked into liblvx.so and no product entry point can reach it.
// It reads its inputs from an oracle Problem (tests may use oracle/), and writes the same
// (residuals, jac_cols, jac_vals) layout as orc_evaluate.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../lvi-exc_amd/csrc/lvx_resid.h"
#include "../../oracle/orc_problem.hpp"

using namespace lvx;

namespace {
SensorCal sensor_from(const double* s) { SensorCal c; c.q = load_q(s); c.p = load_v3(s + 4); c.tau = s[7]; return c; }

template <int NR, int NC, class ColFn>
void emit(int row0, const double* r, const double (*J)[NC], ColFn colfn, int N, int nlm, uint32_t locks,
          double* residuals, int32_t* jac_cols, double* jac_vals) {
  for (int a = 0; a < NR; ++a) {
    if (residuals) residuals[row0 + a] = r[a];
    int32_t* c = jac_cols ? jac_cols + static_cast<size_t>(row0 + a) * ORC_MAX_COLS : nullptr;
    double* v = jac_vals ? jac_vals + static_cast<size_t>(row0 + a) * ORC_MAX_COLS : nullptr;
    for (int k = 0; k < ORC_MAX_COLS; ++k) { if (c) c[k] = -1; if (v) v[k] = 0.0; }
    for (int k = 0; k < NC; ++k) {
      const int g = colfn(k);
      const bool dead = tangent_locked(g, N, nlm, locks);
      if (c) c[k] = dead ? -1 : g;
      if (v) v[k] = dead ? 0.0 : J[a][k];
    }
  }
}
}  // namespace

extern "C" int hc_evaluate(const orc_problem* p, const double* state, double* cost, double* residuals, int32_t* jac_cols, double* jac_vals) {
  const int N = p->n_knots, L = p->n_landmarks;
  SplineRef sp{p->t0, p->dt, N, state, state + 3 * N};
  const double* si = state + 7 * N;
  ImuCal imu; imu.roll = si[8]; imu.pitch = si[9]; imu.ba = load_v3(si + 10); imu.bg = load_v3(si + 13); imu.tau = si[7];
  const SensorCal lidar = sensor_from(si + 16), cam = sensor_from(si + 24);
  const double* rho = si + 32;
  CamIntr ci; std::memset(&ci, 0, sizeof(ci));
  ci.fx = p->cam.fx; ci.fy = p->cam.fy; ci.cx = p->cam.cx; ci.cy = p->cam.cy; ci.k1 = p->cam.k1; ci.k2 = p->cam.k2; ci.p1 = p->cam.p1; ci.p2 = p->cam.p2; ci.k3 = p->cam.k3;
  ci.readout = p->cam.readout; ci.rows = p->cam.rows; ci.cols = p->cam.cols; ci.do_distortion = p->cam.do_distortion;
  ci.inv_K11 = p->cam.inv_K11; ci.inv_K13 = p->cam.inv_K13; ci.inv_K22 = p->cam.inv_K22; ci.inv_K23 = p->cam.inv_K23;
  const uint32_t locks = p->locks | (p->so3_only ? LVXO_LOCK_R3 : 0u);
  double total = 0.0;
  int row = 0;
  int err = 0;
  const int nI = static_cast<int>(p->imu_t.size());
  for (int i = 0; i < nI; ++i, row += 3) {
    double r[3], J[3][GYRO_NC]; int i0 = 0;
    int e = gyro_residual<true>(sp, imu, p->imu_t[i], load_v3(&p->imu_gyro[3 * i]), p->w_gyro, &i0, r, J);
    if (e) { err = e; continue; }
    total += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    emit<3, GYRO_NC>(row, r, J, [&](int c) { return gyro_col(c, i0, N); }, N, L, locks, residuals, jac_cols, jac_vals);
  }
  if (!p->so3_only) {
    for (int i = 0; i < nI; ++i, row += 3) {
      double r[3], J[3][ACC_NC]; int i0 = 0;
      int e = accel_residual<true>(sp, imu, p->imu_t[i], load_v3(&p->imu_acc[3 * i]), p->w_acc, &i0, r, J);
      if (e) { err = e; continue; }
      total += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
      emit<3, ACC_NC>(row, r, J, [&](int c) { return acc_col(c, i0, N); }, N, L, locks, residuals, jac_cols, jac_vals);
    }
  }
  if (p->has_prior) {
    double r[1], J[1][PRI_NC]; int i0 = 0;
    quat qm = mkq(p->prior_q[0], p->prior_q[1], p->prior_q[2], p->prior_q[3]);
    int e = prior_residual<true>(sp, p->prior_t, qm, p->prior_w, &i0, r, J);
    if (e) err = e;
    else { total += 0.5 * r[0] * r[0]; emit<1, PRI_NC>(row, r, J, [&](int c) { return pri_col(c, i0, N); }, N, L, locks, residuals, jac_cols, jac_vals); }
    row += 1;
  }
  auto two_pose_hub = [&](double tau, bool tau_locked, double t_first_k, Segs* /*unused*/, PoseEval* hub, int* i0h) -> int {
    (void)t_first_k; (void)tau_locked;
    KnotRef kh;
    if (!knot_lookup(sp.t0, sp.dt, sp.n, p->t_map, p->t_map + tau, &kh)) return RES_RANGE;
    *i0h = kh.i0;
    return pose_eval<true>(sp, kh, hub) ? RES_OK : RES_NONUNIT;
  };
  const int nS = static_cast<int>(p->surf_t.size());
  for (int i = 0; i < nS; ++i, row += 1) {
    // spans {{t_map, t_map}, {t_k, t_k}} (tau locked) -> segments per residual (hub segment may merge with the k segment)
    const double spans[2][2] = {{p->t_map, p->t_map}, {p->surf_t[i], p->surf_t[i]}};
    Segs segs;
    if (!build_segments(sp, spans, 2, &segs)) { err = RES_RANGE; continue; }
    KnotRef kh; PoseEval hub;
    if (!seg_lookup(sp, segs, p->t_map + lidar.tau, &kh)) { err = RES_RANGE; continue; }
    if (!pose_eval<true>(sp, kh, &hub)) { err = RES_NONUNIT; continue; }
    double r[1], J[1][SURF_NC]; int i0k = 0;
    int e = surfel_residual<true>(sp, hub, segs, lidar, p->surf_t[i], load_v3(&p->surf_pt[3 * i]), load_v3(&p->planes[3 * p->surf_plane[i]]), p->w_surf, &i0k, r, J);
    if (e) { err = e; continue; }
    double sc; total += 0.5 * huber_rho(p->huber_surf, r[0] * r[0], &sc);
    const int i0h = kh.i0;
    emit<1, SURF_NC>(row, r, J, [&](int c) { return surf_col(c, i0h, i0k, N); }, N, L, locks, residuals, jac_cols, jac_vals);
  }
  const int nR = static_cast<int>(p->rep_lm.size());
  for (int i = 0; i < nR; ++i, row += 2) {
    const int lm = p->rep_lm[i];
    double r[2], J[2][REP_NC]; int i0r = 0, i0o = 0;
    int e = reproj_residual<true>(sp, ci, cam, (locks & LVXO_LOCK_CAM_TAU) != 0, p->sensor_max_time_offset, p->lm_uv[2 * lm], p->lm_uv[2 * lm + 1], p->lm_t0[lm],
                                  p->rep_uv[2 * i], p->rep_uv[2 * i + 1], p->rep_t0[i], rho[lm], p->w_rep, &i0r, &i0o, r, J);
    if (e) { err = e; continue; }
    double sc; total += 0.5 * huber_rho(p->huber_rep, r[0] * r[0] + r[1] * r[1], &sc);
    emit<2, REP_NC>(row, r, J, [&](int c) { return rep_col(c, i0r, i0o, N, lm); }, N, L, locks, residuals, jac_cols, jac_vals);
  }
  const int nC = static_cast<int>(p->cs_lm.size());
  for (int i = 0; i < nC; ++i, row += 1) {
    const int lm = p->cs_lm[i];
    const double spans[2][2] = {{p->t_map, p->t_map}, {p->lm_t0[lm], p->lm_t0[lm]}};
    Segs segs;
    if (!build_segments(sp, spans, 2, &segs)) { err = RES_RANGE; continue; }
    KnotRef kh; PoseEval hub;
    if (!seg_lookup(sp, segs, p->t_map + cam.tau, &kh)) { err = RES_RANGE; continue; }
    if (!pose_eval<true>(sp, kh, &hub)) { err = RES_NONUNIT; continue; }
    double r[1], J[1][CS_NC]; int i0k = 0;
    int e = camsurf_residual<true>(sp, hub, segs, ci, cam, lidar, p->lm_uv[2 * lm], p->lm_uv[2 * lm + 1], p->lm_t0[lm], rho[lm],
                                   load_v3(&p->planes[3 * p->cs_plane[i]]), p->w_cs, &i0k, r, J);
    if (e) { err = e; continue; }
    double sc; total += 0.5 * huber_rho(p->huber_cs, r[0] * r[0], &sc);
    const int i0h = kh.i0;
    emit<1, CS_NC>(row, r, J, [&](int c) { return cs_col(c, i0h, i0k, N); }, N, L, locks, residuals, jac_cols, jac_vals);
  }
  (void)two_pose_hub;
  if (cost) *cost = total;
  return -err;
}

// so3_eval (reference arithmetic order) vs so3_eval_pre (u-independent part hoisted, double-angle forms): largest absolute difference
// of q, w_body, dxi and dw for the 4 control points cps[4][4] (x,y,z,w).  Returns 0 when both report the same unit-norm status.
extern "C" int hc_so3_pre_diff(const double* cps, double u, double dt, double* out4) {
  quat c[4];
  for (int j = 0; j < 4; ++j) c[j] = load_q(cps + 4 * j);
  So3Pre pre[3];
  for (int j = 0; j < 3; ++j) so3_pre(c[j], c[j + 1], &pre[j]);
  So3Eval a, b;
  const bool oka = so3_eval<true, true>(c, u, dt, &a);
  const int bad = so3_eval_pre<true, true>(c, pre, u, dt, &b);   // 0 | 1 non-unit | 2 angle beyond the small-angle polynomials
  if (bad & 2) return 2;
  const bool okb = bad == 0;
  auto mx = [](double x, double y) { return x > y ? x : y; };
  out4[0] = mx(mx(std::fabs(a.q.x - b.q.x), std::fabs(a.q.y - b.q.y)), mx(std::fabs(a.q.z - b.q.z), std::fabs(a.q.w - b.q.w)));
  out4[1] = mx(mx(std::fabs(a.w_body.x - b.w_body.x), std::fabs(a.w_body.y - b.w_body.y)), std::fabs(a.w_body.z - b.w_body.z));
  out4[2] = 0.0; out4[3] = 0.0;
  for (int k = 0; k < 4; ++k)
    for (int e = 0; e < 9; ++e) { out4[2] = mx(out4[2], std::fabs(a.dxi[k].a[e] - b.dxi[k].a[e])); out4[3] = mx(out4[3], std::fabs(a.dw[k].a[e] - b.dw[k].a[e])); }
  return oka == okb ? 0 : 1;
}

// reverse mode (so3_value_pre + so3_pullback_pre) vs the forward Jacobian blocks of so3_eval: max |dxi[k]^T g - y[k]| and |q - q_fwd|
extern "C" int hc_so3_pull_diff(const double* cps, double u, double dt, const double* g3, double* out2) {
  quat c[4];
  for (int j = 0; j < 4; ++j) c[j] = load_q(cps + 4 * j);
  So3Pre pre[3];
  for (int j = 0; j < 3; ++j) so3_pre(c[j], c[j + 1], &pre[j]);
  So3Eval a;
  const bool oka = so3_eval<false, true>(c, u, dt, &a);
  So3Val s;
  const int bad = so3_value_pre(c, pre, u, &s);   // 0 | 1 non-unit | 2 angle beyond the small-angle polynomials
  if (bad & 2) return 2;
  const bool okb = bad == 0;
  v3 y[4];
  const v3 g = mk(g3[0], g3[1], g3[2]);
  so3_pullback_pre(c, pre, s, g, y);
  auto mx = [](double x, double z) { return x > z ? x : z; };
  out2[0] = mx(mx(std::fabs(a.q.x - s.q.x), std::fabs(a.q.y - s.q.y)), mx(std::fabs(a.q.z - s.q.z), std::fabs(a.q.w - s.q.w)));
  out2[1] = 0.0;
  for (int k = 0; k < 4; ++k) { const v3 f = tmulv(a.dxi[k], g); out2[1] = mx(out2[1], mx(mx(std::fabs(f.x - y[k].x), std::fabs(f.y - y[k].y)), std::fabs(f.z - y[k].z))); }
  return oka == okb ? 0 : 1;
}

// reverse mode for the body angular velocity (so3_value_w_pre + so3_pullback_w_pre: the gyroscope rows of k_imu_rot) vs so3_eval_pre: max |w_body - w|, |dw[k]^T g - z[k]|
extern "C" int hc_so3_pullw_diff(const double* cps, double u, double dt, const double* g3, double* out2) {
  quat c[4];
  for (int j = 0; j < 4; ++j) c[j] = load_q(cps + 4 * j);
  So3Pre pre[3];
  for (int j = 0; j < 3; ++j) so3_pre(c[j], c[j + 1], &pre[j]);
  So3Eval a;
  const int bada = so3_eval_pre<true, true>(c, pre, u, dt, &a);
  So3ValW w;
  const int bad = so3_value_w_pre(c, pre, u, dt, &w);
  if (bad & 2) return 2;
  v3 z[4];
  const v3 g = mk(g3[0], g3[1], g3[2]);
  so3_pullback_w_pre(c, pre, w, g, z);
  auto mx = [](double x, double y) { return x > y ? x : y; };
  out2[0] = mx(mx(std::fabs(a.w_body.x - w.w_body.x), std::fabs(a.w_body.y - w.w_body.y)), std::fabs(a.w_body.z - w.w_body.z));
  out2[1] = 0.0;
  for (int k = 0; k < 4; ++k) { const v3 f = tmulv(a.dw[k], g); out2[1] = mx(out2[1], mx(mx(std::fabs(f.x - z[k].x), std::fabs(f.y - z[k].y)), std::fabs(f.z - z[k].z))); }
  return bada == bad ? 0 : 1;
}

// two_point_lookup (fast path of the locked-offset LiDAR rows) against build_segments + seg_lookup: 0 = same outcome (or the fast path
// defers to the generic one), 1 = different status, 2 = different knot reference; *code = the fast path's status
extern "C" int hc_two_point_check(double t0, double dt, int n, double t_a, double t_b, double tau, int* code) {
  const SplineRef sp{t0, dt, n, nullptr, nullptr};
  KnotRef kf{0, 0.0}, kg{0, 0.0};
  const int st = two_point_lookup(sp, t_a, t_b, t_b + tau, &kf);
  const double spans[2][2] = {{t_a, t_a}, {t_b, t_b}};
  Segs segs;
  int g;
  if (!build_segments(sp, spans, 2, &segs)) g = 1;
  else if (!seg_lookup(sp, segs, t_b + tau, &kg)) g = 2;
  else g = 0;
  *code = st;
  if (st == -1) return 0;
  if (st != g) return 1;
  if (st == 0 && (kf.i0 != kg.i0 || kf.u != kg.u)) return 2;
  return 0;
}
