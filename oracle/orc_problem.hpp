// oracle/orc_problem.hpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see orc_core.hpp header; PARITY UNPINNED).
// Flat problem description shared by the oracle's evaluators.  Layouts mirror include/lvx.h.
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

#include "orc_core.hpp"

#define ORC_MAX_COLS 128

// lock bits — same values as LVX_LOCK_* in include/lvx.h (mirrors Lock*/IsLocked flags:
// K/kontiki/sensors/sensors.h:113-135, K/kontiki/trajectories/trajectory.h:149-155, constant_bias_imu.h:68-81)
enum : uint32_t {
  LVXO_LOCK_TRAJ = 1u << 0, LVXO_LOCK_R3 = 1u << 1,
  LVXO_LOCK_LIDAR_Q = 1u << 2, LVXO_LOCK_LIDAR_P = 1u << 3, LVXO_LOCK_LIDAR_TAU = 1u << 4,
  LVXO_LOCK_CAM_Q = 1u << 5, LVXO_LOCK_CAM_P = 1u << 6, LVXO_LOCK_CAM_TAU = 1u << 7,
  LVXO_LOCK_ACC_BIAS = 1u << 8, LVXO_LOCK_GYRO_BIAS = 1u << 9, LVXO_LOCK_LANDMARKS = 1u << 10,
};

enum { FAM_GYRO = 0, FAM_ACCEL = 1, FAM_PRIOR = 2, FAM_SURFEL = 3, FAM_REPROJ = 4, FAM_CAMSURF = 5, NUM_FAM = 6 };

namespace orc {

struct Block { const double* ptr; int size; int tan_base; bool is_quat; };  // tan_base < 0: constant block

struct RowSet { int nres = 0; std::vector<int> cols; std::vector<double> vals[4]; };

struct Problem {
  // spline (K/kontiki/trajectories/spline_base.h:31-39)
  double t0 = 0, dt = 1; int n_knots = 0;
  bool so3_only = false;  // Solve #0 runs on TrajectoryEstimator<UniformSO3SplineTrajectory> (L/src/core/trajectory_manager_lvi.cpp:43-62)
  uint32_t locks = LVXO_LOCK_LIDAR_TAU | LVXO_LOCK_CAM_TAU;
  int threads = 0;
  bool block_products = false;   // orc_evaluate_products also forms every block's J^T J upper triangle (CPU-baseline timing: what a Schur-based solver's assembly starts from)
  mutable double products_checksum = 0.0;
  double imu_max_time_offset = 0.01;       // sensors.h:109
  double sensor_max_time_offset = 0.001;   // L/include/core/trajectory_manager_lvi.h:118-119
  PinholeMeta cam;
  // IMU
  std::vector<double> imu_t, imu_gyro, imu_acc; double w_gyro = 1, w_acc = 1;
  // orientation prior
  bool has_prior = false; double prior_t = 0, prior_q[4] = {1, 0, 0, 0}, prior_w = 1;
  // surfels
  std::vector<double> planes;  // Pi[3] per plane (closest-point parametrisation)
  std::vector<double> surf_pt, surf_t; std::vector<int32_t> surf_plane; double t_map = 0, huber_surf = 5.0, w_surf = 1;
  // landmarks + reprojection
  int n_landmarks = 0; std::vector<double> lm_uv, lm_t0;
  std::vector<int32_t> rep_lm; std::vector<double> rep_uv, rep_t0; double huber_rep = 5.0, w_rep = 1;
  // camera-landmark-to-surfel
  std::vector<int32_t> cs_lm, cs_plane; double huber_cs = 5.0, w_cs = 1;

  int state_size() const;
  int tangent_size() const;
  int num_residuals() const;
  int num_blocks() const;
  int family_count(int fam) const;
  static int family_nres(int fam);
  double family_huber(int fam) const;
  void check_time_spans(const std::vector<std::pair<double, double>>& times) const;
  void add_trajectory(const double* state, const std::vector<std::pair<double, double>>& times, SplitMeta& meta, std::vector<Block>& blocks) const;
  void add_imu(const double* state, std::vector<Block>& blocks) const;
  void add_lidar(const double* state, std::vector<Block>& blocks) const;
  void add_camera(const double* state, std::vector<Block>& blocks) const;
  int eval_one(int fam, int i, const double* state, double* res, RowSet* rows) const;
};

}  // namespace orc

typedef orc::Problem orc_problem;
