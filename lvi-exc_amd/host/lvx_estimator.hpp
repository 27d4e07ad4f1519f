// lvx_estimator.hpp — C++ host mirror of the reference's estimator surface over the lvx C ABI (header-only, no HIP types).
//
// The reference's host builds its problem with
//     kontiki::TrajectoryEstimator<SplitTrajectory> est(traj);
//     est.AddMeasurement(std::make_shared<GyroMeasurement>(imu, t, w, weight)); ...          (trajectory_manager_lvi.cpp:464-606)
//     est.Solve(max_iterations, progress);                                                  (kontiki/trajectory_estimator.h:38-68)
// and reads the optimised values back from the entities it shares memory with.  This mirror keeps the same names and
// argument meaning — AddMeasurement(...) per measurement, Solve(max_iterations, progress) — but batches the measurements into
// the flat arrays lvx_set_* expects and runs the whole solve on the GPU.  Errors follow the reference: std::range_error for
// LVX_E_RANGE (spline_base.h:207-221, trajectory_estimator.h:102-127), std::runtime_error for LVX_E_NONUNIT_QUAT
// (quaternion_math.h:19-23) and everything else.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lvx.h"

namespace lvx_host {

struct GyroscopeMeasurement { double t; std::array<double, 3> w; };          // kontiki/measurements/gyroscope_measurement.h:19-20
struct AccelerometerMeasurement { double t; std::array<double, 3> a; };      // kontiki/measurements/accelerometer_measurement.h:22-23
struct LiDARSurfelPoint { std::array<double, 3> lidar_point; int32_t plane_id; double timestamp; };   // lidar_surfel_point.h:19-22
struct StaticRsCameraMeasurement { int32_t landmark_id; std::array<double, 2> uv; double view_t0; };   // static_rscamera_measurement.h:66-67
struct CameraSurfelLandmark { int32_t landmark_id; int32_t plane_id; };      // camera_surfel_landmark.h:19-27
struct OrientationMeasurement { double t; std::array<double, 4> q_wxyz; double weight; };   // orientation_measurement.h:23-27

struct Summary { lvx_lm_summary lm; std::string BriefReport() const {
  static const char* term[] = {"NO_CONVERGENCE", "CONVERGENCE (function tolerance)", "CONVERGENCE (parameter tolerance)", "CONVERGENCE (gradient tolerance)", "NO_CONVERGENCE (max iterations)", "FAILURE"};
  return "lvx LM: iterations " + std::to_string(lm.iterations) + ", initial cost " + std::to_string(lm.initial_cost) + ", final cost " + std::to_string(lm.final_cost) + ", " + term[lm.termination]; } };

// SplitTrajectory + sensors live in one flat state vector (layout in lvx.h); the estimator writes results back into it in place,
// as ceres does with the entities' parameter stores (dynamic_pstore.h:25-30).
class TrajectoryEstimator {
 public:
  TrajectoryEstimator(int device, double t0, double dt, int n_knots, std::vector<double>* state) : state_(state) {
    check(lvx_create(&ctx_, device, 0));
    check(lvx_set_spline(ctx_, t0, dt, n_knots));
  }
  ~TrajectoryEstimator() { lvx_destroy(ctx_); }
  TrajectoryEstimator(const TrajectoryEstimator&) = delete;
  TrajectoryEstimator& operator=(const TrajectoryEstimator&) = delete;

  void SetCamera(const lvx_pinhole& cam) { check(lvx_set_camera(ctx_, &cam)); }
  void SetPlanes(const std::vector<std::array<double, 3>>& closest_points) { check(lvx_set_planes(ctx_, (int)closest_points.size(), closest_points.empty() ? nullptr : closest_points[0].data())); }
  void SetLandmarks(const std::vector<std::array<double, 2>>& uv_ref, const std::vector<double>& t0_ref) { check(lvx_set_landmarks(ctx_, (int)t0_ref.size(), uv_ref.empty() ? nullptr : uv_ref[0].data(), t0_ref.data())); }
  void Lock(uint32_t lvx_lock_mask) { locks_ = lvx_lock_mask; }   // Lock*/LockRelativeOrientation/... of the entities

  // AddMeasurement overloads: same call shape as TrajectoryEstimator::AddMeasurement (trajectory_estimator.h:71-74)
  void AddMeasurement(const GyroscopeMeasurement& m, double weight) { gyro_.push_back(m); w_gyro_ = weight; }
  void AddMeasurement(const AccelerometerMeasurement& m, double weight) { acc_.push_back(m); w_acc_ = weight; }
  void AddMeasurement(const LiDARSurfelPoint& m, double map_time, double huber, double weight) { surf_.push_back(m); t_map_ = map_time; huber_surf_ = huber; w_surf_ = weight; }
  void AddMeasurement(const StaticRsCameraMeasurement& m, double huber, double weight) { rep_.push_back(m); huber_rep_ = huber; w_rep_ = weight; }
  void AddMeasurement(const CameraSurfelLandmark& m, double map_time, double huber, double weight) { cs_.push_back(m); t_map_ = map_time; huber_cs_ = huber; w_cs_ = weight; }
  void AddMeasurement(const OrientationMeasurement& m) { prior_ = m; has_prior_ = true; }

  // TrajectoryEstimator::Solve(max_iterations, progress, num_threads) — num_threads has no meaning on the GPU
  Summary Solve(int max_iterations = 30, bool progress = true, int /*num_threads*/ = -1) {
    upload();
    lvx_lm_options opt; lvx_lm_default_options(&opt);
    opt.max_iterations = max_iterations; opt.verbose = progress ? 1 : 0;
    Summary s{};
    check(lvx_lm_solve(ctx_, state_->data(), &opt, &s.lm));
    return s;
  }
  // one evaluation of every residual block (what ceres::Problem::Evaluate gives the reference's printErrorStatistics)
  double Evaluate(std::vector<double>* residuals = nullptr) {
    upload();
    lvx_layout lo; check(lvx_get_layout(ctx_, &lo));
    if (residuals) residuals->assign((size_t)lo.n_residuals, 0.0);
    double cost = 0;
    check(lvx_evaluate(ctx_, state_->data(), LVX_EVAL_COST | LVX_EVAL_RESIDUALS, &cost, residuals ? residuals->data() : nullptr));
    return cost;
  }
  lvx_ctx* context() { return ctx_; }

 private:
  void upload() {
    // gyro and accel blocks are added per IMU sample in the reference (addGyroscopeMeasurements / addAccelerometerMeasurement
    // iterate the same imu_data_), so they share timestamps
    if (!acc_.empty() && acc_.size() != gyro_.size()) throw std::runtime_error("gyro / accel measurement lists must come from the same IMU samples");
    std::vector<double> t(gyro_.size()), g(3 * gyro_.size()), a(3 * gyro_.size(), 0.0);
    for (size_t i = 0; i < gyro_.size(); ++i) { t[i] = gyro_[i].t; for (int k = 0; k < 3; ++k) { g[3 * i + k] = gyro_[i].w[k]; if (!acc_.empty()) a[3 * i + k] = acc_[i].a[k]; } }
    check(lvx_set_imu(ctx_, (int)t.size(), t.data(), g.data(), a.data(), w_gyro_, w_acc_));
    check(lvx_set_orientation_prior(ctx_, has_prior_ ? 1 : 0, prior_.t, prior_.q_wxyz.data(), prior_.weight));
    std::vector<double> pt(3 * surf_.size()), ts(surf_.size()); std::vector<int32_t> pid(surf_.size());
    for (size_t i = 0; i < surf_.size(); ++i) { ts[i] = surf_[i].timestamp; pid[i] = surf_[i].plane_id; for (int k = 0; k < 3; ++k) pt[3 * i + k] = surf_[i].lidar_point[k]; }
    check(lvx_set_surfel(ctx_, (int)ts.size(), pt.data(), ts.data(), pid.data(), t_map_, huber_surf_, w_surf_));
    std::vector<int32_t> lm(rep_.size()); std::vector<double> uv(2 * rep_.size()), t0(rep_.size());
    for (size_t i = 0; i < rep_.size(); ++i) { lm[i] = rep_[i].landmark_id; uv[2 * i] = rep_[i].uv[0]; uv[2 * i + 1] = rep_[i].uv[1]; t0[i] = rep_[i].view_t0; }
    check(lvx_set_reproj(ctx_, (int)lm.size(), lm.data(), uv.data(), t0.data(), huber_rep_, w_rep_));
    std::vector<int32_t> clm(cs_.size()), cpl(cs_.size());
    for (size_t i = 0; i < cs_.size(); ++i) { clm[i] = cs_[i].landmark_id; cpl[i] = cs_[i].plane_id; }
    check(lvx_set_camsurf(ctx_, (int)clm.size(), clm.data(), cpl.data(), t_map_, huber_cs_, w_cs_));
    check(lvx_set_locks(ctx_, locks_ | (acc_.empty() && !gyro_.empty() ? LVX_LOCK_R3 : 0u)));   // gyro-only estimator == SO3-only Solve #0
  }
  void check(int rc) {
    if (rc == LVX_OK) return;
    const std::string msg = ctx_ ? lvx_last_error(ctx_) : "lvx error";
    if (rc == LVX_E_RANGE) throw std::range_error(msg);
    throw std::runtime_error(msg + " (lvx error " + std::to_string(rc) + ")");
  }
  lvx_ctx* ctx_ = nullptr;
  std::vector<double>* state_;
  uint32_t locks_ = LVX_LOCK_LIDAR_TAU | LVX_LOCK_CAM_TAU;
  std::vector<GyroscopeMeasurement> gyro_; std::vector<AccelerometerMeasurement> acc_; std::vector<LiDARSurfelPoint> surf_;
  std::vector<StaticRsCameraMeasurement> rep_; std::vector<CameraSurfelLandmark> cs_;
  OrientationMeasurement prior_{}; bool has_prior_ = false;
  double w_gyro_ = 1, w_acc_ = 1, w_surf_ = 1, w_rep_ = 1, w_cs_ = 1, huber_surf_ = 5, huber_rep_ = 5, huber_cs_ = 5, t_map_ = 0;
};

}  // namespace lvx_host
