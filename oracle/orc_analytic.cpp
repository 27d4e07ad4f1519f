// oracle/orc_analytic.cpp — TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE (see orc_core.hpp header).
//
// CPU baseline mode (ii) of BASELINE.md §3, "optimised CPU": the same residual blocks as orc_evaluate, but with CLOSED-FORM Jacobians on the group instead of
// stride-4 dual-number passes — a g++ host build of the residual functions the HIP kernels use (lvi-exc_amd/csrc/lvx_math.h / lvx_resid.h are
// __host__ __device__ headers) — plus every block's J^T J / J^T r products, OpenMP over the blocks as Ceres threads its evaluator
// (kontiki/trajectory_estimator.h:48-52).  Timing only: bench.py's cpu_baseline leg reports it beside mode (i); nothing in liblvx.so links or calls it.
// The products of a block are formed (the flops are spent) and folded into a per-thread checksum, not scattered into a global sparse matrix: the baseline
// prices evaluation + block products, the part the GPU pass fuses, not a CPU sparse assembly.
#include <cmath>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../lvi-exc_amd/csrc/lvx_resid.h"
#include "orc_problem.hpp"

using namespace lvx;

namespace {
SensorCal sensor_from(const double* s) { SensorCal c; c.q = load_q(s); c.p = load_v3(s + 4); c.tau = s[7]; return c; }
template <int NR, int NC> inline double block_products(const double* r, const double (*J)[NC], double sc) {
  double acc = 0.0;   // sum over the upper triangle of J^T J and over J^T r (robustified)
  for (int a = 0; a < NC; ++a) {
    double g = 0.0;
    for (int k = 0; k < NR; ++k) g += sc * J[k][a] * sc * r[k];
    acc += g;
    for (int b = a; b < NC; ++b) { double h = 0.0; for (int k = 0; k < NR; ++k) h += sc * J[k][a] * sc * J[k][b]; acc += h; }
  }
  return acc;
}
}  // namespace

// one pass over every block of the problem; returns blocks evaluated (negative: an error code of the first failing block); *cost, *checksum out
extern "C" long long orc_analytic_pass(const orc_problem* p, const double* state, int threads, double* cost, double* checksum) {
  const int N = p->n_knots;
  SplineRef sp{p->t0, p->dt, N, state, state + 3 * N};
  const double* si = state + 7 * N;
  ImuCal imu; imu.roll = si[8]; imu.pitch = si[9]; imu.ba = load_v3(si + 10); imu.bg = load_v3(si + 13); imu.tau = si[7];
  const SensorCal lidar = sensor_from(si + 16), cam = sensor_from(si + 24);
  const double* rho = si + 32;
  CamIntr ci; std::memset(&ci, 0, sizeof(ci));
  ci.fx = p->cam.fx; ci.fy = p->cam.fy; ci.cx = p->cam.cx; ci.cy = p->cam.cy; ci.k1 = p->cam.k1; ci.k2 = p->cam.k2; ci.p1 = p->cam.p1; ci.p2 = p->cam.p2; ci.k3 = p->cam.k3;
  ci.readout = p->cam.readout; ci.rows = p->cam.rows; ci.cols = p->cam.cols; ci.do_distortion = p->cam.do_distortion;
  ci.inv_K11 = p->cam.inv_K11; ci.inv_K13 = p->cam.inv_K13; ci.inv_K22 = p->cam.inv_K22; ci.inv_K23 = p->cam.inv_K23;
  const uint32_t locks = p->locks;
#ifdef _OPENMP
  const int nth = threads > 0 ? threads : omp_get_max_threads();
#else
  const int nth = 1; (void)threads;
#endif
  double total = 0.0, chk = 0.0;
  int err = 0;
  long long blocks = 0;
  const int nI = static_cast<int>(p->imu_t.size());
#pragma omp parallel for schedule(static) num_threads(nth) reduction(+ : total, chk, blocks)
  for (int i = 0; i < nI; ++i) {
    double r[3], J[3][GYRO_NC]; int i0 = 0;
    int e = gyro_residual<true>(sp, imu, p->imu_t[i], load_v3(&p->imu_gyro[3 * i]), p->w_gyro, &i0, r, J);
    if (e) { err = e; continue; }
    total += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]); chk += block_products<3, GYRO_NC>(r, J, 1.0); blocks += 1;
    double ra[3], Ja[3][ACC_NC];
    e = accel_residual<true>(sp, imu, p->imu_t[i], load_v3(&p->imu_acc[3 * i]), p->w_acc, &i0, ra, Ja);
    if (e) { err = e; continue; }
    total += 0.5 * (ra[0] * ra[0] + ra[1] * ra[1] + ra[2] * ra[2]); chk += block_products<3, ACC_NC>(ra, Ja, 1.0); blocks += 1;
  }
  const int nS = static_cast<int>(p->surf_t.size());
#pragma omp parallel for schedule(static) num_threads(nth) reduction(+ : total, chk, blocks)
  for (int i = 0; i < nS; ++i) {
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
    chk += block_products<1, SURF_NC>(r, J, sc); blocks += 1;
  }
  const int nR = static_cast<int>(p->rep_lm.size());
#pragma omp parallel for schedule(static) num_threads(nth) reduction(+ : total, chk, blocks)
  for (int i = 0; i < nR; ++i) {
    const int lm = p->rep_lm[i];
    double r[2], J[2][REP_NC]; int i0r = 0, i0o = 0;
    int e = reproj_residual<true>(sp, ci, cam, (locks & LVXO_LOCK_CAM_TAU) != 0, p->sensor_max_time_offset, p->lm_uv[2 * lm], p->lm_uv[2 * lm + 1], p->lm_t0[lm],
                                  p->rep_uv[2 * i], p->rep_uv[2 * i + 1], p->rep_t0[i], rho[lm], p->w_rep, &i0r, &i0o, r, J);
    if (e) { err = e; continue; }
    double sc; total += 0.5 * huber_rho(p->huber_rep, r[0] * r[0] + r[1] * r[1], &sc);
    chk += block_products<2, REP_NC>(r, J, sc); blocks += 1;
  }
  if (cost) *cost = total;
  if (checksum) *checksum = chk;
  return err ? -static_cast<long long>(err) : blocks;
}
