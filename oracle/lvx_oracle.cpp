// oracle/lvx_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see orc_core.hpp header).
//
// Problem layer of the CPU oracle: wires parameter blocks exactly as each Kontiki
// measurement's AddToEstimator does, evaluates residuals with doubles and Jacobians with
// stride-4 forward-mode dual passes (the cost profile of ceres::DynamicAutoDiffCostFunction),
// applies ceres::HuberLoss / EigenQuaternionParameterization semantics (restated; out-of-tree,
// PARITY UNPINNED) and assembles J^T J / J^T r.  C API (orc_*) for ctypes.
//
// Citations: K/ = /root/reference/src/lvi_exc/thirdparty/Kontiki/include/, L/ = /root/reference/src/lvi_exc/
#include "orc_core.hpp"
#include "orc_problem.hpp"

#include <algorithm>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

// ---------------------------------------------------------------------------
// Generic block evaluation with stride-4 dual passes
// ---------------------------------------------------------------------------
namespace {

constexpr int kStride = 4;  // ceres::DynamicAutoDiffCostFunction<F, Stride = 4>
using J4 = Jet<kStride>;

// Evaluate one residual block.  f(params<T>, res<T>) is the functor body.
// jac_amb[r][scalar] is filled for every scalar of non-constant blocks (ambient coordinates).
template <class F>
void eval_block(const std::vector<Block>& blocks, int nres, F&& f, double* res, std::vector<double>* jac_amb, int* total_scalars) {
  const int nb = static_cast<int>(blocks.size());
  // double pass
  {
    std::vector<const double*> pd(nb);
    for (int b = 0; b < nb; ++b) pd[b] = blocks[b].ptr;
    double r[4];
    f(pd.data(), r);
    for (int i = 0; i < nres; ++i) res[i] = r[i];
  }
  int ns = 0;
  for (auto& b : blocks) ns += b.size;
  if (total_scalars) *total_scalars = ns;
  if (!jac_amb) return;
  jac_amb->assign(static_cast<size_t>(nres) * ns, 0.0);
  // active scalars: those of non-constant blocks
  std::vector<int> active;
  {
    int off = 0;
    for (auto& b : blocks) { if (b.tan_base >= 0) for (int k = 0; k < b.size; ++k) active.push_back(off + k); off += b.size; }
  }
  // per-call temporaries on the heap, as the reference functor does for every pass
  std::vector<J4> store(ns);
  std::vector<const J4*> pj(nb);
  for (size_t p0 = 0; p0 < active.size(); p0 += kStride) {
    int off = 0;
    for (int b = 0; b < nb; ++b) {
      for (int k = 0; k < blocks[b].size; ++k) store[off + k] = J4(blocks[b].ptr[k]);
      pj[b] = &store[off];
      off += blocks[b].size;
    }
    const int np = std::min<int>(kStride, static_cast<int>(active.size() - p0));
    for (int k = 0; k < np; ++k) store[active[p0 + k]].v[k] = 1.0;
    J4 r[4];
    f(pj.data(), r);
    for (int i = 0; i < nres; ++i)
      for (int k = 0; k < np; ++k) (*jac_amb)[static_cast<size_t>(i) * ns + active[p0 + k]] = r[i].v[k];
  }
}

// EigenQuaternionParameterization::ComputeJacobian (ceres, restated): 4x3, storage x,y,z,w
inline void quat_plus_jacobian(const double* x, double P[4][3]) {
  P[0][0] = x[3];  P[0][1] = x[2];  P[0][2] = -x[1];
  P[1][0] = -x[2]; P[1][1] = x[3];  P[1][2] = x[0];
  P[2][0] = x[1];  P[2][1] = -x[0]; P[2][2] = x[3];
  P[3][0] = -x[0]; P[3][1] = -x[1]; P[3][2] = -x[2];
}

// ambient -> tangent; appends (col, value) pairs per row.  Columns of blocks sharing a tangent index are summed.
void to_local(const std::vector<Block>& blocks, int nres, const std::vector<double>& jac_amb, int ns, RowSet& out) {
  out.nres = nres;
  out.cols.clear();
  for (int r = 0; r < nres; ++r) out.vals[r].clear();
  int off = 0;
  for (auto& b : blocks) {
    if (b.tan_base >= 0) {
      const int tsize = b.is_quat ? 3 : b.size;
      // find or append columns
      int pos[4];
      for (int k = 0; k < tsize; ++k) {
        const int col = b.tan_base + k;
        auto it = std::find(out.cols.begin(), out.cols.end(), col);
        if (it == out.cols.end()) { out.cols.push_back(col); for (int r = 0; r < nres; ++r) out.vals[r].push_back(0.0); pos[k] = static_cast<int>(out.cols.size()) - 1; }
        else pos[k] = static_cast<int>(it - out.cols.begin());
      }
      for (int r = 0; r < nres; ++r) {
        const double* ja = &jac_amb[static_cast<size_t>(r) * ns + off];
        if (b.is_quat) {
          double P[4][3]; quat_plus_jacobian(b.ptr, P);
          for (int k = 0; k < 3; ++k) out.vals[r][pos[k]] += ja[0] * P[0][k] + ja[1] * P[1][k] + ja[2] * P[2][k] + ja[3] * P[3][k];
        } else {
          for (int k = 0; k < b.size; ++k) out.vals[r][pos[k]] += ja[k];
        }
      }
    }
    off += b.size;
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// Problem: block wiring (AddToEstimator restatements)
// ---------------------------------------------------------------------------
int Problem::state_size() const { return 3 * n_knots + 4 * n_knots + 16 + 8 + 8 + n_landmarks; }
int Problem::tangent_size() const { return 6 * n_knots + 22 + n_landmarks; }

// trajectory_estimator.h:102-127 CheckTimeSpans
void Problem::check_time_spans(const std::vector<std::pair<double, double>>& times) const {
  int i = 0; double t1_prev = 0;
  const double tmin = t0, tmax = t0 + (n_knots - 3) * dt;
  if (n_knots < 4) throw orc::range_error("Spline had too few control points");
  for (auto& ts : times) {
    const double t1 = ts.first, t2 = ts.second;
    if ((t1 < tmin) || (t2 >= tmax)) throw orc::range_error("Time span out of range for trajectory");
    if (t1 > t2) throw orc::range_error("At least one time span begins before it ends");
    else if ((i > 0) && (t1 < t1_prev)) throw orc::range_error("Time spans are not ordered");
    t1_prev = t1; i += 1;
  }
}

// trajectory_estimator.h:80-86 AddTrajectoryForTimes -> split_trajectory.h:116-123 (R3 first, then SO3)
void Problem::add_trajectory(const double* state, const std::vector<std::pair<double, double>>& times, SplitMeta& meta, std::vector<Block>& blocks) const {
  check_time_spans(times);
  const bool traj_const = (locks & LVXO_LOCK_TRAJ) != 0;
  if (!so3_only) {
    std::vector<int> knots;
    spline_add_to_problem(t0, dt, times, meta.r3, knots);
    for (int k : knots) blocks.push_back(Block{state + 3 * k, 3, (traj_const || (locks & LVXO_LOCK_R3)) ? -1 : 6 * k, false});
  }
  {
    std::vector<int> knots;
    spline_add_to_problem(t0, dt, times, meta.so3, knots);
    const double* so3 = state + 3 * n_knots;
    for (int k : knots) blocks.push_back(Block{so3 + 4 * k, 4, traj_const ? -1 : 6 * k + 3, true});
  }
}

// sensors.h:137-167 (+ imu.h:129-142, constant_bias_imu.h:100-119)
void Problem::add_imu(const double* state, std::vector<Block>& blocks) const {
  const double* s = state + 7 * n_knots;
  const int cb = 6 * n_knots;
  blocks.push_back(Block{s + 0, 4, -1, true});    // q_rel: relative_orientation_locked_ = true (never unlocked for the IMU)
  blocks.push_back(Block{s + 4, 3, -1, false});   // p_rel locked
  blocks.push_back(Block{s + 7, 1, -1, false});   // time offset locked (L/src/core/trajectory_manager_lvi.cpp never unlocks it)
  blocks.push_back(Block{s + 8, 1, cb + 0, false});   // gravity roll: never set constant (imu.h:135-137)
  blocks.push_back(Block{s + 9, 1, cb + 1, false});   // gravity pitch
  blocks.push_back(Block{s + 10, 3, (locks & LVXO_LOCK_ACC_BIAS) ? -1 : cb + 2, false});
  blocks.push_back(Block{s + 13, 3, (locks & LVXO_LOCK_GYRO_BIAS) ? -1 : cb + 5, false});
}
void Problem::add_lidar(const double* state, std::vector<Block>& blocks) const {
  const double* s = state + 7 * n_knots + 16;
  const int cb = 6 * n_knots + 8;
  blocks.push_back(Block{s + 0, 4, (locks & LVXO_LOCK_LIDAR_Q) ? -1 : cb + 0, true});
  blocks.push_back(Block{s + 4, 3, (locks & LVXO_LOCK_LIDAR_P) ? -1 : cb + 3, false});
  blocks.push_back(Block{s + 7, 1, (locks & LVXO_LOCK_LIDAR_TAU) ? -1 : cb + 6, false});
}
void Problem::add_camera(const double* state, std::vector<Block>& blocks) const {
  const double* s = state + 7 * n_knots + 24;
  const int cb = 6 * n_knots + 15;
  blocks.push_back(Block{s + 0, 4, (locks & LVXO_LOCK_CAM_Q) ? -1 : cb + 0, true});
  blocks.push_back(Block{s + 4, 3, (locks & LVXO_LOCK_CAM_P) ? -1 : cb + 3, false});
  blocks.push_back(Block{s + 7, 1, (locks & LVXO_LOCK_CAM_TAU) ? -1 : cb + 6, false});
}

static inline void span_for(double t, bool tau_locked, double max_off, double& tmin, double& tmax) {
  if (tau_locked) { tmin = t; tmax = t; } else { tmin = t - max_off; tmax = t + max_off; }
}

// One residual block of family fam, index i.  Fills res (raw weighted) and, if rows != nullptr, local Jacobian rows.
// Returns number of residuals.
int Problem::eval_one(int fam, int i, const double* state, double* res, RowSet* rows) const {
  std::vector<Block> blocks;
  SplitMeta meta;
  std::vector<double> jac;
  int ns = 0;
  int nres = 0;
  const bool so3o = so3_only;
  switch (fam) {
    case FAM_GYRO: {  // gyroscope_measurement.h:60-110
      const double t = imu_t[i];
      double tmin, tmax; span_for(t, true, imu_max_time_offset, tmin, tmax);  // IMU time offset is always locked
      add_trajectory(state, {{tmin, tmax}}, meta, blocks);
      const int ntraj = static_cast<int>(blocks.size());
      add_imu(state, blocks);
      const double* w = &imu_gyro[3 * i]; const double weight = w_gyro;
      nres = 3;
      auto f = [&](auto const* const* params, auto* r) {
        using T = std::decay_t<decltype(r[0])>;
        TrajView<T> traj{&meta, params, so3o};
        ImuView<T> imu; imu.p = params + ntraj;
        gyro_error<T>(imu, traj, t, w, weight, r);
      };
      eval_block(blocks, nres, f, res, rows ? &jac : nullptr, &ns);
      break;
    }
    case FAM_ACCEL: {  // accelerometer_measurement.h:64-115
      const double t = imu_t[i];
      double tmin, tmax; span_for(t, true, imu_max_time_offset, tmin, tmax);
      add_trajectory(state, {{tmin, tmax}}, meta, blocks);
      const int ntraj = static_cast<int>(blocks.size());
      add_imu(state, blocks);
      const double* a = &imu_acc[3 * i]; const double weight = w_acc;
      nres = 3;
      auto f = [&](auto const* const* params, auto* r) {
        using T = std::decay_t<decltype(r[0])>;
        TrajView<T> traj{&meta, params, so3o};
        ImuView<T> imu; imu.p = params + ntraj;
        accel_error<T>(imu, traj, t, a, weight, r);
      };
      eval_block(blocks, nres, f, res, rows ? &jac : nullptr, &ns);
      break;
    }
    case FAM_PRIOR: {  // orientation_measurement.h:44-80
      add_trajectory(state, {{prior_t, prior_t}}, meta, blocks);
      nres = 1;
      auto f = [&](auto const* const* params, auto* r) {
        using T = std::decay_t<decltype(r[0])>;
        TrajView<T> traj{&meta, params, so3o};
        r[0] = orientation_error<T>(traj, prior_t, prior_q, prior_w);
      };
      eval_block(blocks, nres, f, res, rows ? &jac : nullptr, &ns);
      break;
    }
    case FAM_SURFEL: {  // lidar_surfel_point.h:140-215
      const bool tl = (locks & LVXO_LOCK_LIDAR_TAU) != 0;
      double tmin, tmax, mmin, mmax;
      span_for(surf_t[i], tl, sensor_max_time_offset, tmin, tmax);
      span_for(t_map, tl, sensor_max_time_offset, mmin, mmax);
      add_trajectory(state, {{mmin, mmax}, {tmin, tmax}}, meta, blocks);
      const int ntraj = static_cast<int>(blocks.size());
      add_lidar(state, blocks);
      const double* plane = &planes[3 * surf_plane[i]];
      blocks.push_back(Block{plane, 3, -1, false});  // plane: locked_ = true (lidar_surfel_point.h:108,188-190)
      const double* pt = &surf_pt[3 * i]; const double ts = surf_t[i];
      nres = 1;
      auto f = [&](auto const* const* params, auto* r) {
        using T = std::decay_t<decltype(r[0])>;
        TrajView<T> traj{&meta, params, so3o};
        SensorView<T> lidar; lidar.p = params + ntraj;
        r[0] = surfel_error<T>(traj, lidar, params[ntraj + 3], pt, ts, t_map, w_surf);
      };
      eval_block(blocks, nres, f, res, rows ? &jac : nullptr, &ns);
      break;
    }
    case FAM_REPROJ: {  // static_rscamera_measurement.h:130-207
      const int lm = rep_lm[i];
      const double t0_ref = lm_t0[lm], t0_obs = rep_t0[i];
      double t1, t2;
      if (t0_ref <= t0_obs) { t1 = t0_ref; t2 = t0_obs; } else { t1 = t0_obs; t2 = t0_ref; }
      if (!(locks & LVXO_LOCK_CAM_TAU)) { t1 -= sensor_max_time_offset; t2 += sensor_max_time_offset; }
      const double margin = 1e-3;
      add_trajectory(state, {{t1 - margin, t1 + cam.readout + margin}, {t2 - margin, t2 + cam.readout + margin}}, meta, blocks);
      const int ntraj = static_cast<int>(blocks.size());
      add_camera(state, blocks);
      const double* rho = state + 7 * n_knots + 32 + lm;
      blocks.push_back(Block{rho, 1, (locks & LVXO_LOCK_LANDMARKS) ? -1 : 6 * n_knots + 22 + lm, false});
      const double* uvr = &lm_uv[2 * lm]; const double* uvo = &rep_uv[2 * i];
      nres = 2;
      auto f = [&](auto const* const* params, auto* r) {
        using T = std::decay_t<decltype(r[0])>;
        TrajView<T> traj{&meta, params, so3o};
        CameraView<T> camv; camv.p = params + ntraj; camv.meta = &cam;
        T inverse_depth = params[ntraj + 3][0];
        reproj_error<T>(traj, camv, inverse_depth, uvr, t0_ref, uvo, t0_obs, w_rep, r);
      };
      eval_block(blocks, nres, f, res, rows ? &jac : nullptr, &ns);
      break;
    }
    case FAM_CAMSURF: {  // camera_surfel_landmark.h:176-255
      const int lm = cs_lm[i];
      const bool tl = (locks & LVXO_LOCK_CAM_TAU) != 0;
      double tmin, tmax, mmin, mmax;
      span_for(lm_t0[lm], tl, sensor_max_time_offset, tmin, tmax);
      span_for(t_map, tl, sensor_max_time_offset, mmin, mmax);
      add_trajectory(state, {{mmin, mmax}, {tmin, tmax}}, meta, blocks);
      const int ntraj = static_cast<int>(blocks.size());
      add_camera(state, blocks);
      add_lidar(state, blocks);
      blocks.push_back(Block{&planes[3 * cs_plane[i]], 3, -1, false});
      const double* rho = state + 7 * n_knots + 32 + lm;
      blocks.push_back(Block{rho, 1, (locks & LVXO_LOCK_LANDMARKS) ? -1 : 6 * n_knots + 22 + lm, false});
      const double rho_const = *rho;  // functor reads measurement.landmark_->inverse_depth() as a constant (:159-161)
      const double* uvr = &lm_uv[2 * lm]; const double ts = lm_t0[lm];
      nres = 1;
      auto f = [&](auto const* const* params, auto* r) {
        using T = std::decay_t<decltype(r[0])>;
        TrajView<T> traj{&meta, params, so3o};
        CameraView<T> camv; camv.p = params + ntraj; camv.meta = &cam;
        SensorView<T> lidar; lidar.p = params + ntraj + 3;
        r[0] = camsurf_error<T>(traj, camv, lidar, params[ntraj + 6], rho_const, uvr, ts, t_map, w_cs);
      };
      eval_block(blocks, nres, f, res, rows ? &jac : nullptr, &ns);
      break;
    }
    default: throw std::runtime_error("bad family");
  }
  if (rows) to_local(blocks, nres, jac, ns, *rows);
  return nres;
}

int Problem::family_count(int fam) const {
  switch (fam) {
    case FAM_GYRO: return static_cast<int>(imu_t.size());
    case FAM_ACCEL: return so3_only ? 0 : static_cast<int>(imu_t.size());
    case FAM_PRIOR: return has_prior ? 1 : 0;
    case FAM_SURFEL: return static_cast<int>(surf_t.size());
    case FAM_REPROJ: return static_cast<int>(rep_lm.size());
    case FAM_CAMSURF: return static_cast<int>(cs_lm.size());
  }
  return 0;
}
int Problem::family_nres(int fam) { return (fam == FAM_GYRO || fam == FAM_ACCEL) ? 3 : (fam == FAM_REPROJ ? 2 : 1); }
double Problem::family_huber(int fam) const {  // <=0: no loss (nullptr loss function in AddResidualBlock)
  switch (fam) { case FAM_SURFEL: return huber_surf; case FAM_REPROJ: return huber_rep; case FAM_CAMSURF: return huber_cs; }
  return 0.0;
}
int Problem::num_residuals() const { int n = 0; for (int f = 0; f < NUM_FAM; ++f) n += family_count(f) * family_nres(f); return n; }
int Problem::num_blocks() const { int n = 0; for (int f = 0; f < NUM_FAM; ++f) n += family_count(f); return n; }

// ceres::HuberLoss + Corrector (restated): rho(s) = s (s<=a^2) else 2a sqrt(s) - a^2; rho'' <= 0 => scale r and J by sqrt(rho')
static inline void huber(double a, double s, double& rho, double& sqrt_rho1) {
  const double b = a * a;
  if (a > 0 && s > b) { const double r = std::sqrt(s); rho = 2.0 * a * r - b; sqrt_rho1 = std::sqrt(std::max(std::numeric_limits<double>::min(), a / r)); }
  else { rho = s; sqrt_rho1 = 1.0; }
}

// ---------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------
extern "C" {

orc_problem* orc_create() { return new Problem(); }
void orc_destroy(orc_problem* p) { delete p; }

int orc_set_spline(orc_problem* p, double t0, double dt, int n_knots) { p->t0 = t0; p->dt = dt; p->n_knots = n_knots; return 0; }
int orc_set_camera(orc_problem* p, int rows, int cols, double readout, double fx, double fy, double cx, double cy,
                   double k1, double k2, double p1, double p2, double k3) {
  p->cam.rows = rows; p->cam.cols = cols; p->cam.readout = readout; p->cam.fx = fx; p->cam.fy = fy; p->cam.cx = cx; p->cam.cy = cy;
  p->cam.k1 = k1; p->cam.k2 = k2; p->cam.p1 = p1; p->cam.p2 = p2; p->cam.k3 = k3; p->cam.finalize(); return 0; }
int orc_set_imu(orc_problem* p, int n, const double* t, const double* gyro3, const double* acc3, double w_g, double w_a) {
  p->imu_t.assign(t, t + n); p->imu_gyro.assign(gyro3, gyro3 + 3 * n); p->imu_acc.assign(acc3, acc3 + 3 * n); p->w_gyro = w_g; p->w_acc = w_a; return 0; }
int orc_set_orientation_prior(orc_problem* p, int enable, double t, const double* q_wxyz, double w) {
  p->has_prior = enable != 0; p->prior_t = t; if (q_wxyz) std::memcpy(p->prior_q, q_wxyz, 4 * sizeof(double)); p->prior_w = w; return 0; }
int orc_set_planes(orc_problem* p, int n, const double* pi3) { p->planes.assign(pi3, pi3 + 3 * n); return 0; }
int orc_set_surfel(orc_problem* p, int n, const double* pt3, const double* t, const int32_t* plane_id, double t_map, double huber_, double w) {
  p->surf_pt.assign(pt3, pt3 + 3 * n); p->surf_t.assign(t, t + n); p->surf_plane.assign(plane_id, plane_id + n);
  p->t_map = t_map; p->huber_surf = huber_; p->w_surf = w; return 0; }
int orc_set_landmarks(orc_problem* p, int n, const double* uv_ref2, const double* t0_ref) {
  p->n_landmarks = n; p->lm_uv.assign(uv_ref2, uv_ref2 + 2 * n); p->lm_t0.assign(t0_ref, t0_ref + n); return 0; }
int orc_set_reproj(orc_problem* p, int n, const int32_t* lm, const double* uv_obs2, const double* t0_obs, double huber_, double w) {
  p->rep_lm.assign(lm, lm + n); p->rep_uv.assign(uv_obs2, uv_obs2 + 2 * n); p->rep_t0.assign(t0_obs, t0_obs + n); p->huber_rep = huber_; p->w_rep = w; return 0; }
int orc_set_camsurf(orc_problem* p, int n, const int32_t* lm, const int32_t* plane_id, double t_map, double huber_, double w) {
  p->cs_lm.assign(lm, lm + n); p->cs_plane.assign(plane_id, plane_id + n); p->t_map = t_map; p->huber_cs = huber_; p->w_cs = w; return 0; }
int orc_set_locks(orc_problem* p, uint32_t mask) { p->locks = mask; return 0; }
int orc_set_so3_only(orc_problem* p, int flag) { p->so3_only = flag != 0; return 0; }
int orc_set_threads(orc_problem* p, int n) { p->threads = n; return 0; }
int orc_set_block_products(orc_problem* p, int on) { p->block_products = on != 0; p->products_checksum = 0.0; return 0; }
double orc_products_checksum(const orc_problem* p) { return p->products_checksum; }

int orc_state_size(const orc_problem* p) { return p->state_size(); }
int orc_tangent_size(const orc_problem* p) { return p->tangent_size(); }
int orc_num_residuals(const orc_problem* p) { return p->num_residuals(); }
int orc_num_blocks(const orc_problem* p) { return p->num_blocks(); }
int orc_max_cols() { return ORC_MAX_COLS; }

// Evaluate everything.  residuals: raw weighted residuals, family-major (gyro, accel, prior, surfel, reproj, camsurf).
// jac_cols/jac_vals (optional): per residual row ORC_MAX_COLS entries, cols = global tangent index or -1; raw (pre-loss).
// H/g (optional): dense tangent_size^2 / tangent_size, robustified (what Ceres would assemble).
// Returns 0, or -1 (range error), -2 (non-unit quaternion).
int orc_evaluate(const orc_problem* p, const double* state, double* cost, double* residuals,
                 int32_t* jac_cols, double* jac_vals, double* H, double* g) {
  const int nt = p->tangent_size();
  const bool want_jac = jac_cols || jac_vals || H || g;
  if (H) std::memset(H, 0, sizeof(double) * nt * nt);
  if (g) std::memset(g, 0, sizeof(double) * nt);
  double total = 0.0;
  int err = 0;
  int row0 = 0;
  for (int fam = 0; fam < NUM_FAM; ++fam) {
    const int cnt = p->family_count(fam);
    const int nr = Problem::family_nres(fam);
    const double a = p->family_huber(fam);
    double fam_cost = 0.0;
#ifdef _OPENMP
    const int nth = (H || g) ? 1 : (p->threads > 0 ? p->threads : omp_get_max_threads());
#pragma omp parallel for schedule(static) num_threads(nth) reduction(+ : fam_cost)
#endif
    for (int i = 0; i < cnt; ++i) {
      if (err) continue;
      try {
        double r[4];
        RowSet rows;
        p->eval_one(fam, i, state, r, want_jac ? &rows : nullptr);
        double s = 0; for (int k = 0; k < nr; ++k) s += r[k] * r[k];
        double rho, sr; huber(a, s, rho, sr);
        fam_cost += 0.5 * rho;
        const int row = row0 + i * nr;
        if (residuals) for (int k = 0; k < nr; ++k) residuals[row + k] = r[k];
        if (want_jac) {
          const int nc = static_cast<int>(rows.cols.size());
          if (nc > ORC_MAX_COLS) throw std::runtime_error("too many columns");
          for (int k = 0; k < nr; ++k) {
            if (jac_cols) { int32_t* c = jac_cols + static_cast<size_t>(row + k) * ORC_MAX_COLS; for (int j = 0; j < ORC_MAX_COLS; ++j) c[j] = j < nc ? rows.cols[j] : -1; }
            if (jac_vals) { double* v = jac_vals + static_cast<size_t>(row + k) * ORC_MAX_COLS; for (int j = 0; j < ORC_MAX_COLS; ++j) v[j] = j < nc ? rows.vals[k][j] : 0.0; }
          }
          if (H || g) {
            for (int k = 0; k < nr; ++k) {
              const double rk = sr * r[k];
              for (int a1 = 0; a1 < nc; ++a1) {
                const double ja = sr * rows.vals[k][a1];
                if (g) g[rows.cols[a1]] += ja * rk;
                if (H) for (int b1 = 0; b1 < nc; ++b1) H[static_cast<size_t>(rows.cols[a1]) * nt + rows.cols[b1]] += ja * sr * rows.vals[k][b1];
              }
            }
          }
        }
      } catch (const orc::range_error&) {
#ifdef _OPENMP
#pragma omp critical
#endif
        err = -1;
      } catch (const orc::nonunit_quat_error&) {
#ifdef _OPENMP
#pragma omp critical
#endif
        err = -2;
      } catch (const std::exception&) {
#ifdef _OPENMP
#pragma omp critical
#endif
        err = -3;
      }
    }
    total += fam_cost;
    row0 += cnt * nr;
  }
  if (cost) *cost = total;
  return err;
}

// Matrix-free products with the robustified Jacobian, for problems whose dense J^T J does not fit (config 4: 150 k tangent scalars): one OpenMP pass
// over the residual blocks (what one ceres::Problem::Evaluate visits, K/kontiki/trajectory_estimator.h:38-68) gives the raw residuals, the cost,
// g = J^T r, diag(J^T J) and, for n_vec given tangent vectors V[k], HV[k] = J^T (J V[k]) — J and r scaled by sqrt(rho') as the Corrector does.
// Thread-private accumulators, summed in thread order.  Any output may be NULL.  Returns as orc_evaluate.
int orc_evaluate_products(const orc_problem* p, const double* state, int n_vec, const double* V, double* cost, double* residuals, double* g, double* diag, double* HV) {
  const int nt = p->tangent_size();
#ifdef _OPENMP
  const int nth = p->threads > 0 ? p->threads : omp_get_max_threads();
#else
  const int nth = 1;
#endif
  const size_t per = static_cast<size_t>(2 + n_vec) * nt;
  std::vector<double> acc(per * nth, 0.0);
  double total = 0.0;
  int err = 0;
  int row0 = 0;
  for (int fam = 0; fam < NUM_FAM; ++fam) {
    const int cnt = p->family_count(fam);
    const int nr = Problem::family_nres(fam);
    const double a = p->family_huber(fam);
    double fam_cost = 0.0, fam_chk = 0.0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nth) reduction(+ : fam_cost, fam_chk)
#endif
    for (int i = 0; i < cnt; ++i) {
      if (err) continue;
#ifdef _OPENMP
      double* my = acc.data() + per * omp_get_thread_num();
#else
      double* my = acc.data();
#endif
      try {
        double r[4];
        RowSet rows;
        p->eval_one(fam, i, state, r, &rows);
        double s = 0; for (int k = 0; k < nr; ++k) s += r[k] * r[k];
        double rho, sr; huber(a, s, rho, sr);
        fam_cost += 0.5 * rho;
        const int row = row0 + i * nr;
        if (residuals) for (int k = 0; k < nr; ++k) residuals[row + k] = r[k];
        const int nc = static_cast<int>(rows.cols.size());
        if (p->block_products) {   // the block's dense J^T J (upper triangle): the flops are spent and folded into a checksum, as orc_analytic_pass does
          double chk = 0.0;
          for (int a1 = 0; a1 < nc; ++a1)
            for (int b1 = a1; b1 < nc; ++b1) { double h = 0.0; for (int k = 0; k < nr; ++k) h += sr * rows.vals[k][a1] * sr * rows.vals[k][b1]; chk += h; }
          fam_chk += chk;
        }
        for (int k = 0; k < nr; ++k) {
          const double rk = sr * r[k];
          for (int a1 = 0; a1 < nc; ++a1) { const double ja = sr * rows.vals[k][a1]; my[rows.cols[a1]] += ja * rk; my[nt + rows.cols[a1]] += ja * ja; }
          for (int v = 0; v < n_vec; ++v) {
            const double* x = V + static_cast<size_t>(v) * nt;
            double jx = 0; for (int a1 = 0; a1 < nc; ++a1) jx += sr * rows.vals[k][a1] * x[rows.cols[a1]];
            double* y = my + static_cast<size_t>(2 + v) * nt;
            for (int a1 = 0; a1 < nc; ++a1) y[rows.cols[a1]] += sr * rows.vals[k][a1] * jx;
          }
        }
      } catch (const orc::range_error&) {
#ifdef _OPENMP
#pragma omp critical
#endif
        err = -1;
      } catch (const orc::nonunit_quat_error&) {
#ifdef _OPENMP
#pragma omp critical
#endif
        err = -2;
      } catch (const std::exception&) {
#ifdef _OPENMP
#pragma omp critical
#endif
        err = -3;
      }
    }
    total += fam_cost;
    p->products_checksum += fam_chk;
    row0 += cnt * nr;
  }
  if (cost) *cost = total;
  for (int j = 0; j < nt; ++j) {
    double sg = 0, sd = 0;
    for (int th = 0; th < nth; ++th) { sg += acc[per * th + j]; sd += acc[per * th + nt + j]; }
    if (g) g[j] = sg;
    if (diag) diag[j] = sd;
  }
  if (HV) for (int v = 0; v < n_vec; ++v) for (int j = 0; j < nt; ++j) {
    double sy = 0; for (int th = 0; th < nth; ++th) sy += acc[per * th + static_cast<size_t>(2 + v) * nt + j];
    HV[static_cast<size_t>(v) * nt + j] = sy;
  }
  return err;
}

// x_plus = x (+) delta: Euclidean add for vectors, EigenQuaternionParameterization::Plus for quaternions
// (q_new = [sin|d|/|d| d, cos|d|] * q; ceres, restated).  delta in the tangent layout.
void orc_plus(const orc_problem* p, const double* state, const double* delta, double* out) {
  const int N = p->n_knots;
  std::memcpy(out, state, sizeof(double) * p->state_size());
  auto qplus = [](const double* x, const double* d, double* o) {
    const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    Quat<double> dq;
    if (nd > 0.0) { const double s = std::sin(nd) / nd; dq = Quat<double>(std::cos(nd), s * d[0], s * d[1], s * d[2]); }
    else { dq = Quat<double>(1.0, d[0], d[1], d[2]); }  // ceres: delta_q = (1, d) when |d| == 0 ... identity
    Quat<double> q = Quat<double>::from_coeffs(x);
    Quat<double> r = dq * q;
    o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
  };
  for (int k = 0; k < N; ++k) {
    for (int j = 0; j < 3; ++j) out[3 * k + j] = state[3 * k + j] + delta[6 * k + j];
    qplus(state + 3 * N + 4 * k, delta + 6 * k + 3, out + 3 * N + 4 * k);
  }
  const double* s = state + 7 * N; double* o = out + 7 * N; const double* d = delta + 6 * N;
  o[8] = s[8] + d[0]; o[9] = s[9] + d[1];
  for (int j = 0; j < 3; ++j) { o[10 + j] = s[10 + j] + d[2 + j]; o[13 + j] = s[13 + j] + d[5 + j]; }
  qplus(s + 16, d + 8, o + 16); for (int j = 0; j < 3; ++j) o[20 + j] = s[20 + j] + d[11 + j]; o[23] = s[23] + d[14];
  qplus(s + 24, d + 15, o + 24); for (int j = 0; j < 3; ++j) o[28 + j] = s[28 + j] + d[18 + j]; o[31] = s[31] + d[21];
  for (int l = 0; l < p->n_landmarks; ++l) o[32 + l] = s[32 + l] + d[22 + l];
  // ceres::ParameterBlock::Plus projects onto the box constraints the measurements set: inverse depth >= 0 (static_rscamera_measurement.h:185,
  // camera_surfel_landmark.h:232), |time offset| <= max_time_offset for a free offset (sensors.h:161-162)
  const double mto = p->sensor_max_time_offset;
  if (!(p->locks & LVXO_LOCK_LIDAR_TAU)) o[23] = std::min(std::max(o[23], -mto), mto);
  if (!(p->locks & LVXO_LOCK_CAM_TAU)) o[31] = std::min(std::max(o[31], -mto), mto);
  if (!(p->locks & LVXO_LOCK_LANDMARKS)) for (int l = 0; l < p->n_landmarks; ++l) o[32 + l] = std::max(o[32 + l], 0.0);
}

// Batch pose evaluation (position + orientation (x,y,z,w) [+ velocity, accel, angular velocity]) — used by KAT tests.
int orc_eval_pose(const orc_problem* p, const double* state, int n, const double* t, double* pos3, double* quat4, double* vel3, double* acc3, double* angvel3) {
  try {
    for (int i = 0; i < n; ++i) {
      std::vector<Block> blocks; SplitMeta meta;
      Problem q = *p; q.so3_only = false; q.locks = 0;
      q.add_trajectory(state, {{t[i], t[i]}}, meta, blocks);
      std::vector<const double*> pd(blocks.size()); for (size_t b = 0; b < blocks.size(); ++b) pd[b] = blocks[b].ptr;
      TrajView<double> traj{&meta, pd.data(), false};
      Eval<double> e;
      traj.Evaluate(t[i], EvalPosition | EvalVelocity | EvalAcceleration | EvalOrientation | EvalAngularVelocity, e);
      if (pos3) { pos3[3 * i] = e.position.x; pos3[3 * i + 1] = e.position.y; pos3[3 * i + 2] = e.position.z; }
      if (vel3) { vel3[3 * i] = e.velocity.x; vel3[3 * i + 1] = e.velocity.y; vel3[3 * i + 2] = e.velocity.z; }
      if (acc3) { acc3[3 * i] = e.acceleration.x; acc3[3 * i + 1] = e.acceleration.y; acc3[3 * i + 2] = e.acceleration.z; }
      if (quat4) { quat4[4 * i] = e.orientation.x; quat4[4 * i + 1] = e.orientation.y; quat4[4 * i + 2] = e.orientation.z; quat4[4 * i + 3] = e.orientation.w; }
      if (angvel3) { angvel3[3 * i] = e.angular_velocity.x; angvel3[3 * i + 1] = e.angular_velocity.y; angvel3[3 * i + 2] = e.angular_velocity.z; }
    }
  } catch (const orc::range_error&) { return -1; } catch (const orc::nonunit_quat_error&) { return -2; }
  return 0;
}

// TrajectoryManagerLVI::evaluateLidarPose (L/src/core/trajectory_manager_lvi.cpp:398-408): q_LtoG = q(t) q_LtoI, p_LinG = q(t) p_LinI + p(t);
// valid iff MinTime <= t + tau_L < MaxTime.  Concrete (single-segment) trajectory: SplineView::Evaluate over the whole spline.
static bool lidar_pose(const orc_problem* p, const double* state, double t, Quat<double>& q_LtoG, V3<double>& p_LinG) {
  const int N = p->n_knots;
  const double* sl = state + 7 * N + 16;
  const double tt = t + sl[7];
  const double tmin = p->t0, tmax = p->t0 + (N - 3) * p->dt;
  if (tmin > tt || tmax <= tt) return false;
  SplitMeta meta;
  meta.r3.segments.push_back(SegMeta{p->t0, p->dt, N});
  meta.so3.segments.push_back(SegMeta{p->t0, p->dt, N});
  std::vector<const double*> pd(2 * N);
  for (int k = 0; k < N; ++k) { pd[k] = state + 3 * k; pd[N + k] = state + 3 * N + 4 * k; }
  TrajView<double> traj{&meta, pd.data(), false};
  Eval<double> e;
  traj.Evaluate(tt, EvalOrientation | EvalPosition, e);
  const Quat<double> q_LtoI = Quat<double>::from_coeffs(sl);
  const V3<double> p_LinI(sl[4], sl[5], sl[6]);
  q_LtoG = e.orientation * q_LtoI;
  p_LinG = e.orientation * p_LinI + e.position;
  return true;
}
int orc_eval_lidar_pose(const orc_problem* p, const double* state, int n, const double* t, double* q_xyzw, double* pos, int32_t* valid) {
  try {
    for (int i = 0; i < n; ++i) {
      Quat<double> q; V3<double> pp;
      const bool ok = lidar_pose(p, state, t[i], q, pp);
      valid[i] = ok ? 1 : 0;
      if (ok) { q_xyzw[4 * i] = q.x; q_xyzw[4 * i + 1] = q.y; q_xyzw[4 * i + 2] = q.z; q_xyzw[4 * i + 3] = q.w; pos[3 * i] = pp.x; pos[3 * i + 1] = pp.y; pos[3 * i + 2] = pp.z; }
    }
  } catch (const orc::nonunit_quat_error&) { return -2; } catch (const orc::range_error&) { return -1; }
  return 0;
}
// ScanUndistortion::undistort (L/include/core/scan_undistortion.h:132-180).  raw: PointXYZIT {float x,y,z,pad; float intensity; (pad); double timestamp} (32 B);
// out: float xyzi per point (NaN point for NaN input; zeros where the pose is unavailable, like the default-constructed VPoint)
int orc_undistort(const orc_problem* p, const double* state, int n, const void* raw_v, const double* q_G_to_target_xyzw, const double* p_target_in_G, int correct_position, float* out) {
  struct PT { float x, y, z, pad; float intensity; float pad2; double timestamp; };
  const PT* raw = static_cast<const PT*>(raw_v);
  const Quat<double> qGt = Quat<double>::from_coeffs(q_G_to_target_xyzw);
  const V3<double> pT(p_target_in_G[0], p_target_in_G[1], p_target_in_G[2]);
  try {
    for (int i = 0; i < n; ++i) {
      float* o = out + 4 * i;
      o[0] = o[1] = o[2] = o[3] = 0.f;
      if (std::isnan(raw[i].x)) { o[0] = o[1] = o[2] = NAN; continue; }
      Quat<double> q_LktoG; V3<double> p_LkinG;
      if (!lidar_pose(p, state, raw[i].timestamp, q_LktoG, p_LkinG)) continue;
      const Quat<double> q_LktoL0 = qGt * q_LktoG;
      const V3<double> p_Lk(raw[i].x, raw[i].y, raw[i].z);
      V3<double> po = q_LktoL0 * p_Lk;
      if (correct_position) po = po + qGt * (p_LkinG - pT);
      o[0] = static_cast<float>(po.x); o[1] = static_cast<float>(po.y); o[2] = static_cast<float>(po.z); o[3] = raw[i].intensity;
    }
  } catch (const orc::nonunit_quat_error&) { return -2; } catch (const orc::range_error&) { return -1; }
  return 0;
}

// TrajectoryManagerLVI::evaluateCameraPose (L/src/core/trajectory_manager_lvi.cpp:430-440): q_CtoG = q(t) q_CtoI, p_CinG = q(t) p_CinI + p(t), valid iff
// MinTime <= t + tau_C < MaxTime
static bool camera_pose(const orc_problem* p, const double* state, double t, Quat<double>& q_CtoG, V3<double>& p_CinG) {
  const int N = p->n_knots;
  const double* sc = state + 7 * N + 24;
  const double tt = t + sc[7];
  const double tmin = p->t0, tmax = p->t0 + (N - 3) * p->dt;
  if (tmin > tt || tmax <= tt) return false;
  SplitMeta meta;
  meta.r3.segments.push_back(SegMeta{p->t0, p->dt, N});
  meta.so3.segments.push_back(SegMeta{p->t0, p->dt, N});
  std::vector<const double*> pd(2 * N);
  for (int k = 0; k < N; ++k) { pd[k] = state + 3 * k; pd[N + k] = state + 3 * N + 4 * k; }
  TrajView<double> traj{&meta, pd.data(), false};
  Eval<double> e;
  traj.Evaluate(tt, EvalOrientation | EvalPosition, e);
  const Quat<double> q_CtoI = Quat<double>::from_coeffs(sc);
  const V3<double> p_CinI(sc[4], sc[5], sc[6]);
  q_CtoG = e.orientation * q_CtoI;
  p_CinG = e.orientation * p_CinI + e.position;
  return true;
}
// SurfelAssociation::associateVisualPointsWithPlanes (L/src/core/surfel_association.cpp:161-214): every landmark of the problem against every
// surfel (ascending index: the last match stays); plane_of_landmark[l] = surfel index or -1.  planes as orc_surfel_assoc.
int orc_landmark_assoc(const orc_problem* p, const double* state, const double* q_LtoC_xyzw, const double* t_LinC, double map_time, int P, const double* p4, const double* bmin,
                       const double* bmax, double radius, int32_t* plane_of_landmark) {
  const int N = p->n_knots, L = p->n_landmarks;
  for (int l = 0; l < L; ++l) plane_of_landmark[l] = -1;
  try {
    Quat<double> q_CtoG; V3<double> p_CinG;
    if (!camera_pose(p, state, map_time, q_CtoG, p_CinG)) return 0;                       // :176-179
    const Quat<double> q_LtoC = Quat<double>::from_coeffs(q_LtoC_xyzw);
    const Quat<double> q_L0_G = q_CtoG * q_LtoC;
    const V3<double> t_L0_G = q_CtoG * V3<double>(t_LinC[0], t_LinC[1], t_LinC[2]) + p_CinG;
    const double n2 = q_L0_G.x * q_L0_G.x + q_L0_G.y * q_L0_G.y + q_L0_G.z * q_L0_G.z + q_L0_G.w * q_L0_G.w;
    Quat<double> q_inv = q_L0_G.conjugate();                                              // Eigen inverse(): conjugate / squaredNorm
    q_inv.x /= n2; q_inv.y /= n2; q_inv.z /= n2; q_inv.w /= n2;
    const double* sc = state + 7 * N + 24;
    CameraView<double> camv; const double* cp[3] = {sc, sc + 4, sc + 7}; camv.p = cp; camv.meta = &p->cam;
    for (int l = 0; l < L; ++l) {
      const double rho = state[7 * N + 32 + l];
      const double uv[2] = {p->lm_uv[2 * l], p->lm_uv[2 * l + 1]};
      V3<double> p3d_C = camv.Unproject(uv);
      if (rho < 0.05) continue;                                                           // beyond 20 m (:190)
      p3d_C = p3d_C / rho;
      if (!camera_pose(p, state, p->lm_t0[l], q_CtoG, p_CinG)) continue;
      const V3<double> p3d_G = q_CtoG * p3d_C + p_CinG;
      const V3<double> q = q_inv * (p3d_G - t_L0_G);
      for (int k = 0; k < P; ++k) {
        const double* lo = bmin + 3 * k; const double* hi = bmax + 3 * k; const double* pl = p4 + 4 * k;
        if (q.x > lo[0] && q.x < hi[0] && q.y > lo[1] && q.y < hi[1] && q.z > lo[2] && q.z < hi[2]) {
          double dst = q.x * pl[0] + q.y * pl[1] + q.z * pl[2] + pl[3];
          dst = dst > 0 ? dst : -dst;
          if (dst <= radius * 2) plane_of_landmark[l] = k;
        }
      }
    }
  } catch (const orc::nonunit_quat_error&) { return -2; } catch (const orc::range_error&) { return -1; }
  return 0;
}

}  // extern "C"
