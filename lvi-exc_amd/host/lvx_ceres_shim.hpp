// lvx_ceres_shim.hpp — keep ceres::Solve and the reference's Problem surface, replace only the evaluation.
//
// The reference's seam is one ceres::CostFunction per residual block, built in every measurement's AddToEstimator
// (kontiki/measurements/lidar_surfel_point.h:140-215, static_rscamera_measurement.h:135-203, camera_surfel_landmark.h:176-255,
// gyroscope_measurement.h:80-106, accelerometer_measurement.h, orientation_measurement.h:60-82) and evaluated block by block by ceres::Solve
// (kontiki/trajectory_estimator.h:38-68).  This header supplies drop-ins for exactly that surface:
//   * LvxEvaluationCallback : ceres::EvaluationCallback — ONE lvx_evaluate of the whole batch per iterate (options.evaluation_callback);
//   * LvxRowBlock : ceres::CostFunction — one per residual block, with the reference's parameter-block order and sizes, which hands out the
//     cached residual rows and the AMBIENT Jacobian blocks Ceres expects (quaternion blocks 4 wide);
//   * BlockLayout — which parameter blocks a block has: the segment construction of SplineEntity::AddToProblem (spline_base.h:380-424) and the
//     push order of SplitEntity / SensorEntity / ImuEntity / ConstantBiasImuEntity::AddToProblem (split_trajectory.h:116-122, sensors.h:137-167,
//     imu.h:129-142, constant_bias_imu.h:100-119);
//   * PackState / UnpackState between the Kontiki entities and the flat state vector of lvx.h.
// Tangent -> ambient: lvx produces Jacobians in the tangent of ceres::EigenQuaternionParameterization (3 columns per quaternion).  Ceres
// multiplies whatever a CostFunction returns by the manifold Jacobian P(q) (4 x 3, columns e_j (x) q, P^T P = I for unit q), so the block
// handed out is J_tangent P(q)^T: Ceres' own J P then reproduces the tangent rows exactly.
// Only <ceres/ceres.h> (CostFunction, EvaluationCallback) is needed; header-only; no HIP types.
#pragma once
#include <ceres/ceres.h>

#include <array>
#include <cmath>
#include <functional>
#include <stdexcept>
#include <utility>
#include <vector>

#include "../../include/lvx.h"

namespace lvx_host {

// one Ceres parameter block of a residual block
struct ParamBlock {
  int state_off;     // offset in the flat state vector (-1: not part of it — the surfel plane, always constant: lidar_surfel_point.h:108,188-190)
  int size;          // ambient size Ceres sees (4 for quaternions)
  int tangent_off;   // first tangent scalar (-1: the block has none: IMU q_rel / p_rel / tau stay constant in the reference, planes)
  bool quat;
};
struct BlockSpec { int family, index, num_residuals; std::vector<ParamBlock> params; };

// SplineEntity::AddToProblem (spline_base.h:380-424): segments (first knot, number of knots) for sorted time spans
inline std::vector<std::pair<int, int>> SegmentsForSpans(double t0, double dt, int n_knots, const std::vector<std::pair<double, double>>& spans) {
  const double tmax = t0 + (n_knots - 3) * dt;
  double prev = 0; bool first = true;
  for (const auto& s : spans) {   // TrajectoryEstimator::CheckTimeSpans (trajectory_estimator.h:102-127)
    if (s.first < t0 || s.second >= tmax) throw std::range_error("Time span out of range for trajectory");
    if (s.first > s.second) throw std::range_error("At least one time span begins before it ends");
    if (!first && s.first < prev) throw std::range_error("Time spans are not ordered");
    prev = s.first; first = false;
  }
  std::vector<std::pair<int, int>> segs;
  int cur_start = 0, cur_end = -1;
  for (const auto& s : spans) {
    int i1 = (int)std::floor((s.first - t0) / dt);
    const int i2 = (int)std::floor((s.second - t0) / dt);
    if (i1 > cur_end) { segs.push_back({i1, 0}); cur_start = i1; } else i1 = cur_end + 1;
    for (int i = i1; i < i2 + 4; ++i) segs.back().second += 1;
    cur_end = cur_start + segs.back().second - 1;
  }
  return segs;
}

// Parameter blocks of every block type in the order the reference pushes them (what `parameters[k]` means inside Evaluate)
class BlockLayout {
 public:
  BlockLayout(double t0, double dt, int n_knots, int n_landmarks, double readout, double sensor_max_time_offset, uint32_t locks)
      : t0_(t0), dt_(dt), N_(n_knots), L_(n_landmarks), readout_(readout), mto_(sensor_max_time_offset), locks_(locks) {}
  BlockSpec Gyro(int i, double t) const { BlockSpec b{LVX_FAM_GYRO, i, 3, {}}; traj(b, {{t, t}}); imu(b); return b; }
  BlockSpec Accel(int i, double t) const { BlockSpec b{LVX_FAM_ACCEL, i, 3, {}}; traj(b, {{t, t}}); imu(b); return b; }
  BlockSpec Prior(double t) const { BlockSpec b{LVX_FAM_PRIOR, 0, 1, {}}; traj(b, {{t, t}}); return b; }
  BlockSpec Surfel(int i, double t_map, double t) const {
    const double pad = (locks_ & LVX_LOCK_LIDAR_TAU) ? 0.0 : mto_;
    BlockSpec b{LVX_FAM_SURFEL, i, 1, {}}; traj(b, {{t_map - pad, t_map + pad}, {t - pad, t + pad}}); sensor(b, 16, 8); plane(b); return b;
  }
  BlockSpec Reproj(int i, double t0_ref, double t0_obs, int landmark) const {
    double t1 = std::min(t0_ref, t0_obs), t2 = std::max(t0_ref, t0_obs);
    if (!(locks_ & LVX_LOCK_CAM_TAU)) { t1 -= mto_; t2 += mto_; }
    const double m = 1e-3;                                             // static_rscamera_measurement.h:148-172
    BlockSpec b{LVX_FAM_REPROJ, i, 2, {}}; traj(b, {{t1 - m, t1 + readout_ + m}, {t2 - m, t2 + readout_ + m}}); sensor(b, 24, 15); rho(b, landmark); return b;
  }
  BlockSpec CamSurf(int i, double t_map, double t_ref, int landmark) const {
    const double pad = (locks_ & LVX_LOCK_CAM_TAU) ? 0.0 : mto_;
    BlockSpec b{LVX_FAM_CAMSURF, i, 1, {}}; traj(b, {{t_map - pad, t_map + pad}, {t_ref - pad, t_ref + pad}});
    sensor(b, 24, 15); sensor(b, 16, 8); plane(b); rho(b, landmark); return b;   // camera, lidar, plane, rho (camera_surfel_landmark.h:212-236)
  }
  int n_knots() const { return N_; }
  int n_landmarks() const { return L_; }

 private:
  void traj(BlockSpec& b, const std::vector<std::pair<double, double>>& spans) const {
    const auto segs = SegmentsForSpans(t0_, dt_, N_, spans);
    if (!(locks_ & LVX_LOCK_R3)) for (const auto& s : segs) for (int k = s.first; k < s.first + s.second; ++k) b.params.push_back({3 * k, 3, 6 * k, false});
    for (const auto& s : segs) for (int k = s.first; k < s.first + s.second; ++k) b.params.push_back({3 * N_ + 4 * k, 4, 6 * k + 3, true});
  }
  void sensor(BlockSpec& b, int so, int to) const {   // q_rel[4], p_rel[3], tau[1] (sensors.h:141-166)
    const int sb = 7 * N_, tb = 6 * N_;
    b.params.push_back({sb + so, 4, tb + to, true}); b.params.push_back({sb + so + 4, 3, tb + to + 3, false}); b.params.push_back({sb + so + 7, 1, tb + to + 6, false});
  }
  void imu(BlockSpec& b) const {                      // + roll, pitch (imu.h:135-141), b_a, b_g (constant_bias_imu.h:106-118)
    const int sb = 7 * N_, tb = 6 * N_;
    b.params.push_back({sb + 0, 4, -1, true}); b.params.push_back({sb + 4, 3, -1, false}); b.params.push_back({sb + 7, 1, -1, false});
    b.params.push_back({sb + 8, 1, tb + 0, false}); b.params.push_back({sb + 9, 1, tb + 1, false});
    b.params.push_back({sb + 10, 3, tb + 2, false}); b.params.push_back({sb + 13, 3, tb + 5, false});
  }
  void plane(BlockSpec& b) const { b.params.push_back({-1, 3, -1, false}); }
  void rho(BlockSpec& b, int l) const { b.params.push_back({7 * N_ + 32 + l, 1, 6 * N_ + 22 + l, false}); }
  double t0_, dt_; int N_, L_; double readout_, mto_; uint32_t locks_;
};

// ceres::Solver::Options::evaluation_callback: evaluates every block on the GPU once per iterate
class LvxEvaluationCallback : public ceres::EvaluationCallback {
 public:
  // pack(state): copy the parameter blocks Ceres has just updated into the flat state (PackState below, bound to the host's entities)
  LvxEvaluationCallback(lvx_ctx* ctx, std::function<void(double*)> pack) : ctx_(ctx), pack_(std::move(pack)) {
    lvx_layout lo;
    if (lvx_get_layout(ctx_, &lo) != LVX_OK) throw std::runtime_error(lvx_last_error(ctx_));
    state_.assign((size_t)lvx_state_size(ctx_), 0.0);
    residuals_.assign((size_t)lo.n_residuals, 0.0);
    jac_cols_.assign((size_t)lo.n_residuals * LVX_JAC_WIDTH, -1);
    jac_vals_.assign((size_t)lo.n_residuals * LVX_JAC_WIDTH, 0.0);
    if (lvx_get_family_rows(ctx_, row0_) != LVX_OK) throw std::runtime_error(lvx_last_error(ctx_));
  }
  void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) override {
    if (!new_evaluation_point && !(evaluate_jacobians && !have_jac_)) return;
    pack_(state_.data());
    double cost = 0;
    const int rc = lvx_evaluate(ctx_, state_.data(), LVX_EVAL_COST | LVX_EVAL_RESIDUALS | (evaluate_jacobians ? LVX_EVAL_JACOBIAN : 0u), &cost, residuals_.data());
    ok_ = rc == LVX_OK;   // a failed evaluation makes every block's Evaluate return false: Ceres rejects the step, as it does for a throwing functor
    if (ok_ && evaluate_jacobians) ok_ = lvx_get_jacobian(ctx_, jac_cols_.data(), jac_vals_.data()) == LVX_OK;
    have_jac_ = evaluate_jacobians && ok_;
  }
  bool ok() const { return ok_; }
  bool have_jacobians() const { return have_jac_; }
  const double* state() const { return state_.data(); }
  const double* residual_row(int family, int index, int nr) const { return residuals_.data() + row0_[family] + (int64_t)index * nr; }
  const int32_t* cols_row(int64_t row) const { return jac_cols_.data() + row * LVX_JAC_WIDTH; }
  const double* vals_row(int64_t row) const { return jac_vals_.data() + row * LVX_JAC_WIDTH; }
  int64_t first_row(int family, int index, int nr) const { return row0_[family] + (int64_t)index * nr; }

 private:
  lvx_ctx* ctx_;
  std::function<void(double*)> pack_;
  std::vector<double> state_, residuals_, jac_vals_;
  std::vector<int32_t> jac_cols_;
  int64_t row0_[LVX_NUM_FAM + 1] = {0};
  bool ok_ = false, have_jac_ = false;
};

// one residual block: problem.AddResidualBlock(new LvxRowBlock(cb, spec), loss, parameter_blocks_in_the_reference_order)
class LvxRowBlock : public ceres::CostFunction {
 public:
  LvxRowBlock(const LvxEvaluationCallback* cb, BlockSpec spec) : cb_(cb), spec_(std::move(spec)) {
    set_num_residuals(spec_.num_residuals);
    for (const auto& p : spec_.params) mutable_parameter_block_sizes()->push_back(p.size);
  }
  bool Evaluate(double const* const* /*parameters*/, double* residuals, double** jacobians) const override {
    if (!cb_->ok()) return false;
    const int nr = spec_.num_residuals;
    const double* r = cb_->residual_row(spec_.family, spec_.index, nr);
    for (int a = 0; a < nr; ++a) residuals[a] = r[a];
    if (!jacobians) return true;
    if (!cb_->have_jacobians()) return false;
    const int64_t row0 = cb_->first_row(spec_.family, spec_.index, nr);
    for (size_t k = 0; k < spec_.params.size(); ++k) {
      double* J = jacobians[k];
      if (!J) continue;                                  // constant block
      const ParamBlock& p = spec_.params[k];
      for (int e = 0; e < nr * p.size; ++e) J[e] = 0.0;
      if (p.tangent_off < 0) { if (p.state_off >= 0) return false; continue; }   // a block lvx treats as constant was left variable in the Problem
      const int nt = p.quat ? 3 : p.size;
      for (int a = 0; a < nr; ++a) {
        const int32_t* cols = cb_->cols_row(row0 + a); const double* vals = cb_->vals_row(row0 + a);
        double jt[4] = {0, 0, 0, 0};                     // the row's entries at this block's tangent scalars (a variable can appear through two poses: sum)
        for (int c = 0; c < LVX_JAC_WIDTH; ++c) { const int d = cols[c] - p.tangent_off; if (cols[c] >= 0 && d >= 0 && d < nt) jt[d] += vals[c]; }
        if (!p.quat) { for (int d = 0; d < nt; ++d) J[a * p.size + d] = jt[d]; continue; }
        // J_ambient = J_tangent P(q)^T, P(q)[:, j] = e_j (x) q (x, y, z, w): d (delta (+) q) / d delta at 0 of EigenQuaternionParameterization
        const double* q = cb_->state() + p.state_off;
        const double P[4][3] = {{q[3], q[2], -q[1]}, {-q[2], q[3], q[0]}, {q[1], -q[0], q[3]}, {-q[0], -q[1], -q[2]}};
        for (int c = 0; c < 4; ++c) J[a * 4 + c] = jt[0] * P[c][0] + jt[1] * P[c][1] + jt[2] * P[c][2];
      }
    }
    return true;
  }
  const BlockSpec& spec() const { return spec_; }

 private:
  const LvxEvaluationCallback* cb_;
  BlockSpec spec_;
};

// ---------------------------------------------------------------------------------------------------------
// Kontiki entities <-> flat state (layout in lvx.h).  Written against the accessors the reference's views expose: control points as
// Eigen::Map (spline_base.h:118-127: ControlPoint(i) / MutableControlPoint(i); .data() for R3, .coeffs().data() = x, y, z, w for SO3),
// sensors (sensors.h:36-85: relative_orientation(), relative_position(), time_offset()), IMU (imu.h:40-58 gravity_orientation_roll / pitch,
// constant_bias_imu.h:25-49 accelerometer_bias / gyroscope_bias), landmarks (sfm/landmark.h inverse_depth / set_inverse_depth).
// ---------------------------------------------------------------------------------------------------------
template <class Sensor> void PackSensor(const Sensor& s, double* o) {
  const auto q = s.relative_orientation(); const auto p = s.relative_position();
  for (int k = 0; k < 4; ++k) o[k] = q.coeffs().data()[k];
  for (int k = 0; k < 3; ++k) o[4 + k] = p.data()[k];
  o[7] = s.time_offset();
}
template <class Sensor> void UnpackSensor(const double* o, Sensor& s) {
  auto q = s.relative_orientation(); auto p = s.relative_position();
  for (int k = 0; k < 4; ++k) q.coeffs().data()[k] = o[k];
  for (int k = 0; k < 3; ++k) p.data()[k] = o[4 + k];
  s.set_time_offset(o[7]);
}
template <class Traj, class Imu, class Lidar, class Camera, class LandmarkPtrs>
std::vector<double> PackState(const Traj& traj, const Imu& imu, const Lidar& lidar, const Camera& cam, const LandmarkPtrs& landmarks) {
  const int N = (int)traj.R3Spline()->NumKnots(), L = (int)landmarks.size();
  std::vector<double> s((size_t)7 * N + 32 + L);
  for (int k = 0; k < N; ++k) {
    const auto p = traj.R3Spline()->ControlPoint(k); const auto q = traj.SO3Spline()->ControlPoint(k);
    for (int j = 0; j < 3; ++j) s[3 * k + j] = p.data()[j];
    for (int j = 0; j < 4; ++j) s[3 * N + 4 * k + j] = q.coeffs().data()[j];
  }
  double* b = s.data() + 7 * N;
  PackSensor(imu, b);
  b[8] = imu.gravity_orientation_roll(); b[9] = imu.gravity_orientation_pitch();
  { const auto ba = imu.accelerometer_bias(); const auto bg = imu.gyroscope_bias(); for (int j = 0; j < 3; ++j) { b[10 + j] = ba.data()[j]; b[13 + j] = bg.data()[j]; } }
  PackSensor(lidar, b + 16); PackSensor(cam, b + 24);
  for (int l = 0; l < L; ++l) b[32 + l] = landmarks[l]->inverse_depth();
  return s;
}
template <class Traj, class Imu, class Lidar, class Camera, class LandmarkPtrs>
void UnpackState(const std::vector<double>& s, Traj& traj, Imu& imu, Lidar& lidar, Camera& cam, LandmarkPtrs& landmarks) {
  const int N = (int)traj.R3Spline()->NumKnots(), L = (int)landmarks.size();
  if (s.size() != (size_t)7 * N + 32 + L) throw std::invalid_argument("state size does not match the entities");
  for (int k = 0; k < N; ++k) {
    auto p = traj.R3Spline()->MutableControlPoint(k); auto q = traj.SO3Spline()->MutableControlPoint(k);
    for (int j = 0; j < 3; ++j) p.data()[j] = s[3 * k + j];
    for (int j = 0; j < 4; ++j) q.coeffs().data()[j] = s[3 * N + 4 * k + j];
  }
  const double* b = s.data() + 7 * N;
  UnpackSensor(b, imu);
  imu.set_gravity_orientation_roll(b[8]); imu.set_gravity_orientation_pitch(b[9]);
  { auto ba = imu.accelerometer_bias(); auto bg = imu.gyroscope_bias(); for (int j = 0; j < 3; ++j) { ba.data()[j] = b[10 + j]; bg.data()[j] = b[13 + j]; } }
  UnpackSensor(b + 16, lidar); UnpackSensor(b + 24, cam);
  for (int l = 0; l < L; ++l) landmarks[l]->set_inverse_depth(b[32 + l]);
}

}  // namespace lvx_host
