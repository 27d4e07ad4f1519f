// lvx_calibrate.hpp — the reference's offline calibration schedule over the lvx C ABI (header-only, no HIP types).
//
// What it mirrors (src/lvi_exc/test/lvi_initialize_surfel_orb.cpp): LIinitializer's stage machine
//     Initialization()      -> TrajectoryManagerLVI::initialSO3TrajWithGyro            (trajectory_manager_lvi.cpp:43-62)        Solve #0
//     DataAssociation()     -> undistortScanInMap, ndt grid of the map cloud, setSurfelMap, getAssociation per scan,
//                              averageTimeDownSmaple                                   (:1169-1210)
//     BatchOptimization() / Refinement()  -> trajInitFromSurfel                        (:1212-1243, trajectory_manager_lvi.cpp:311-351)   Solve #1, #1', ...
//     LVI refinement        -> trajInitFromLVIdata(frames, surfels)                    (trajectory_manager_lvi.cpp:138-195)       Solve #2
//     camera refinement     -> associateVisualPointsWithPlanes + trajInitFromLVIdata(frames, surfels, lm_splane)  (:197-257)     Solve #3
// DataAssociation is ONE device-resident call (lvx_data_association: de-skew of every scan into the map frame, voxel grid of the map cloud, surfel
// extraction, association of every scan, chronological SurfelPoint emission); the raw scans are handed to the context once (lvx_set_scans).
// The FIRST DataAssociation (InitializationDone branch, :1175-1178) takes its map from per-scan odometry poses — LOAM's, ReadPoseGT (lvx_loaders.hpp) — exactly as the
// reference with using_loam: CalibrateInput::loam + scan_stamps select it (lvx_data_association_poses: rotation-only de-skew with the SO3 spline of Solve #0, key-scan
// map, surfel map, association), so a recorded dataset starts from the reference's initial state (identity rotations, zero positions) with nothing hand-fitted.
// Not mirrored: LiDAROdometry's own NDT scan-to-map registration (using_loam = false; lvx_ndt_align is its align()).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lvx.h"
#include "lvx_loaders.hpp"

namespace lvx_host {

struct CalibrateInput {
  double t0 = 0, dt = 0.02; int n_knots = 0;                       // spline meta
  lvx_pinhole camera{};
  std::vector<double> imu_t, gyro, acc;                            // [n], [n][3], [n][3]
  int H = 0, W = 0;                                                // organised scans
  std::vector<std::vector<lvx_point_xyzit>> scans;                 // raw LiDAR scans, [H * W] each (row-major h * W + w), per-point timestamps
  double map_time = 0;
  // first map from odometry poses (optional): the scans' header stamps and the LOAM pose file (ReadPoseGT); simulation = the reference's flag: a scan takes the pose with
  // EXACTLY its stamp (loam_poses_map_.find, :1283-1290) instead of the nearest of the poses idx - 5 .. idx + 4 (findAssociatedPose, :1246-1260)
  std::vector<double> scan_stamps; LoamPoses loam; bool simulation = true;
  // ORB results (lvx_loaders.hpp: LoadOrbResults): landmark table + observations
  std::vector<double> lm_uv, lm_t0; std::vector<int32_t> obs_landmark; std::vector<double> obs_uv, obs_t0;
};
struct CalibrateOptions {
  bool solve0_so3_from_gyro = false;     // Initialization()
  int refine_iterations = 2;             // DataAssociation + trajInitFromSurfel rounds (the reference runs batch + 2 refinements)
  bool lvi_stage = true;                 // trajInitFromLVIdata
  bool camera_surfel_stage = false;      // the third stage with camera-landmark-to-surfel blocks
  float ndt_resolution = 0.5f;           // lvi.yaml:26
  double plane_lambda = 0.7, fit_threshold = 0.05; int min_leaf_points = 10, min_inliers = 20;
  double first_map_plane_lambda = 0.6, key_scan_dist = 0.2, key_scan_angle_deg = 5.0;   // plane_lambda_ of the constructor (:127); checkKeyScan (lidar_odometry.cpp:121-122)
  double associated_radius = 0.05; int selected_per_ring = 2, downsample_step = 10;
  double w_gyro = 28, w_acc = 18, w_surfel = 10, w_cam = 5, w_cam_surfel = 30;   // SetCalibWeights (lvi_initialize_surfel_orb.cpp:904-928)
  bool opt_time_offset = false;
  bool keep_history = false, keep_clouds = false;   // AssociationRecord per DataAssociation round (tests, diagnostics)
  int verbose = 1;
};
struct StageReport { std::string name; lvx_lm_summary lm; int n_planes = 0, n_surfel_points = 0, n_cam_surfel = 0;
                     std::vector<double> cost_history, radius_history; std::vector<int32_t> accepted;   // per-iteration trace of the stage's solve (lvx_lm_get_history)
                     std::vector<double> state_in; };   // the state the stage's solve started from (CalibrateOptions::keep_history)
// what one DataAssociation round produced (kept when CalibrateOptions::keep_history): the state it ran at, the surfel map, the full SurfelPoint list and —
// keep_clouds — the de-skewed scans [n_scans][H][W][4]
struct AssociationRecord { std::vector<double> state; std::vector<lvx_surfel_plane> planes; std::vector<double> pt, pt_map, t; std::vector<int32_t> plane; std::vector<float> scans_in_map; };

class Calibrator {
 public:
  // `in` is copied: the calibrator may outlive the caller's object
  Calibrator(int device, const CalibrateInput& in, const CalibrateOptions& opt) : in_(in), opt_(opt) {
    check(lvx_create(&ctx_, device, 0));
    check(lvx_set_spline(ctx_, in_.t0, in_.dt, in_.n_knots));
    check(lvx_set_camera(ctx_, &in_.camera));
    check(lvx_set_landmarks(ctx_, (int)in_.lm_t0.size(), in_.lm_uv.data(), in_.lm_t0.data()));
    // LioDataset::get_scan_data: the organised raw scans go to the device once
    const size_t HW = (size_t)in_.H * in_.W;
    std::vector<lvx_point_xyzit> raw(in_.scans.size() * HW);
    for (size_t s = 0; s < in_.scans.size(); ++s) {
      if (in_.scans[s].size() != HW) throw std::invalid_argument("scan size does not match H x W");
      std::copy(in_.scans[s].begin(), in_.scans[s].end(), raw.begin() + s * HW);
    }
    if (!raw.empty()) check(lvx_set_scans(ctx_, (int)in_.scans.size(), in_.H, in_.W, raw.data()));
  }
  ~Calibrator() { lvx_destroy(ctx_); }
  Calibrator(const Calibrator&) = delete;
  Calibrator& operator=(const Calibrator&) = delete;

  // state: flat vector of lvx.h, in / out
  std::vector<StageReport> Run(std::vector<double>* state) {
    if ((int)state->size() != lvx_state_size(ctx_)) throw std::invalid_argument("state size does not match the problem");
    std::vector<StageReport> rep;
    if (opt_.solve0_so3_from_gyro) rep.push_back(Solve0(state));
    for (int it = 0; it < opt_.refine_iterations; ++it) {
      if (it == 0 && !in_.loam.all.empty()) FirstDataAssociation(*state); else
      DataAssociation(*state);
      rep.push_back(SolveSurfel(state, it == 0 ? "BatchOptimization" : "Refinement"));
    }
    if (opt_.lvi_stage) rep.push_back(SolveLVI(state, false));
    if (opt_.camera_surfel_stage) rep.push_back(SolveLVI(state, true));
    return rep;
  }
  // LIinitializer::CIoptimize (lvi_initialize_surfel_orb.cpp:519-537): camera-IMU calibration alone — initialSO3TrajWithGyro, then trajInitFromVisualFrames
  // (trajectory_manager_lvi.cpp:99-136: gyroscope + accelerometer + reprojection blocks, <= 200 iterations, LiDAR extrinsics locked)
  std::vector<StageReport> RunCameraImu(std::vector<double>* state) {
    if ((int)state->size() != lvx_state_size(ctx_)) throw std::invalid_argument("state size does not match the problem");
    std::vector<StageReport> rep;
    rep.push_back(Solve0(state));
    rep.push_back(SolveVisual(state));
    return rep;
  }
  const std::vector<lvx_surfel_plane>& planes() const { return planes_; }
  const std::vector<AssociationRecord>& associations() const { return assoc_history_; }
  const std::vector<int32_t>& key_scans() const { return key_scans_; }   // of the first-map association: 1 where the scan joined the key-scan map
  lvx_ctx* context() { return ctx_; }

 private:
  StageReport Solve0(std::vector<double>* state) {   // initialSO3TrajWithGyro: gyro blocks + one orientation prior at MinTime, SO3 spline only
    std::vector<double> zero(in_.acc.size(), 0.0);
    check(lvx_set_imu(ctx_, (int)in_.imu_t.size(), in_.imu_t.data(), in_.gyro.data(), zero.data(), opt_.w_gyro, opt_.w_acc));
    const double q0[4] = {std::cos(0.5e-4), 0, 0, std::sin(0.5e-4)};
    check(lvx_set_orientation_prior(ctx_, 1, in_.t0, q0, opt_.w_gyro));
    clear_families(true, true, true);
    check(lvx_set_locks(ctx_, StageLocks(Stage::SO3FromGyro, opt_.opt_time_offset)));
    StageReport r{"initialSO3TrajWithGyro", solve(state, 30)};
    attach_history(&r);
    check(lvx_set_orientation_prior(ctx_, 0, in_.t0, q0, opt_.w_gyro));
    return r;
  }
  bool pose_of_scan(int idx, double scan_t, double T[16]) const { return PoseOfScan(in_.loam, in_.simulation, idx, scan_t, T); }
  lvx_assoc_options assoc_options(double plane_lambda) const {
    lvx_assoc_options ao; lvx_assoc_default_options(&ao);
    ao.ndt_resolution = opt_.ndt_resolution; ao.plane_lambda = plane_lambda; ao.fit_threshold = opt_.fit_threshold; ao.min_leaf_points = opt_.min_leaf_points;
    ao.min_inliers = opt_.min_inliers; ao.radius = opt_.associated_radius; ao.selected_per_ring = opt_.selected_per_ring;
    return ao;
  }
  // DataAssociation of the InitializationDone branch (lvi_initialize_surfel_orb.cpp:1175-1178): Mapping() with the LOAM poses, undistortScanInMap(odom_data_map), setSurfelMap
  // on the key-scan map's NDT grid, getAssociation per scan
  void FirstDataAssociation(const std::vector<double>& state) {
    const size_t S = in_.scans.size();
    if (in_.scan_stamps.size() != S) throw std::invalid_argument("scan_stamps: one header stamp per scan is needed for the first map");
    std::vector<double> poses(S * 16, 0.0); std::vector<int32_t> has(S, 0);
    for (size_t s = 0; s < S; ++s) has[s] = pose_of_scan((int)s, in_.scan_stamps[s], &poses[16 * s]) ? 1 : 0;
    const lvx_assoc_options ao = assoc_options(opt_.first_map_plane_lambda);
    int32_t np = 0, n = 0;
    key_scans_.assign(S, 0);
    check(lvx_data_association_poses(ctx_, state.data(), in_.scan_stamps.data(), poses.data(), has.data(), opt_.key_scan_dist, opt_.key_scan_angle_deg, &ao, &np, &n, key_scans_.data()));
    fetch_association(state, np, n);
  }
  // DataAssociation of the refinement branch (lvi_initialize_surfel_orb.cpp:1180-1188, 1192-1201)
  void DataAssociation(const std::vector<double>& state) {
    const lvx_assoc_options ao = assoc_options(opt_.plane_lambda);
    int32_t np = 0, n = 0;
    check(lvx_data_association(ctx_, state.data(), in_.map_time, &ao, &np, &n));
    fetch_association(state, np, n);
  }
  void fetch_association(const std::vector<double>& state, int32_t np, int32_t n) {
    planes_.assign((size_t)np, lvx_surfel_plane{});
    if (np > 0) check(lvx_get_surfel_map(ctx_, np, planes_.data()));
    sp_pt_.assign((size_t)n * 3, 0.0); sp_map_.assign((size_t)n * 3, 0.0); sp_t_.assign((size_t)n, 0.0); sp_plane_.assign((size_t)n, 0);
    if (n > 0) check(lvx_get_surfel_points(ctx_, n, sp_pt_.data(), sp_map_.data(), sp_t_.data(), sp_plane_.data()));
    if (opt_.keep_history) {
      AssociationRecord r; r.state = state; r.planes = planes_; r.pt = sp_pt_; r.pt_map = sp_map_; r.t = sp_t_; r.plane = sp_plane_;
      if (opt_.keep_clouds) { r.scans_in_map.assign(in_.scans.size() * (size_t)in_.H * in_.W * 4, 0.f); check(lvx_get_scans_in_map(ctx_, r.scans_in_map.data())); }
      assoc_history_.push_back(std::move(r));
    }
    if (opt_.verbose) std::fprintf(stderr, "[lvx calibrate] association: %d surfels, %d surfel points (every %d-th is used)\n", np, n, opt_.downsample_step);
  }
  void set_surfels() {   // addSurfMeasurement over get_surfel_points() after averageTimeDownSmaple(step) (surfel_association.cpp:240-244)
    std::vector<double> Pi(planes_.size() * 3);
    for (size_t k = 0; k < planes_.size(); ++k) for (int a = 0; a < 3; ++a) Pi[3 * k + a] = planes_[k].Pi[a];
    check(lvx_set_planes(ctx_, (int)planes_.size(), Pi.data()));
    std::vector<double> pt, t; std::vector<int32_t> pid;
    for (size_t i = 0; i < sp_t_.size(); i += (size_t)std::max(1, opt_.downsample_step)) {
      if (sp_t_[i] < in_.map_time) continue;   // CheckTimeSpans: {map_time, t} must be ordered (trajectory_estimator.h:102-127) — the reference would throw
      pt.insert(pt.end(), sp_pt_.begin() + 3 * i, sp_pt_.begin() + 3 * i + 3); t.push_back(sp_t_[i]); pid.push_back(sp_plane_[i]);
    }
    n_surfel_used_ = (int)t.size();
    check(lvx_set_surfel(ctx_, (int)t.size(), pt.data(), t.data(), pid.data(), in_.map_time, 5.0, opt_.w_surfel));
  }
  StageReport SolveSurfel(std::vector<double>* state, const char* name) {   // trajInitFromSurfel
    check(lvx_set_imu(ctx_, (int)in_.imu_t.size(), in_.imu_t.data(), in_.gyro.data(), in_.acc.data(), opt_.w_gyro, opt_.w_acc));
    set_surfels();
    clear_families(false, true, true);
    check(lvx_set_locks(ctx_, StageLocks(Stage::TrajFromSurfel, opt_.opt_time_offset)));
    StageReport r{name, solve(state, 30)};
    attach_history(&r);
    r.n_planes = (int)planes_.size(); r.n_surfel_points = n_surfel_used_;
    return r;
  }
  StageReport SolveVisual(std::vector<double>* state) {   // trajInitFromVisualFrames
    check(lvx_set_imu(ctx_, (int)in_.imu_t.size(), in_.imu_t.data(), in_.gyro.data(), in_.acc.data(), opt_.w_gyro, opt_.w_acc));
    clear_families(true, false, true);
    check(lvx_set_reproj(ctx_, (int)in_.obs_landmark.size(), in_.obs_landmark.data(), in_.obs_uv.data(), in_.obs_t0.data(), /*huber*/ opt_.w_cam, /*weight*/ 1.0));   // argument swap of :525
    check(lvx_set_locks(ctx_, StageLocks(Stage::TrajFromVisualFrames, opt_.opt_time_offset)));
    StageReport r{"trajInitFromVisualFrames", solve(state, 200)};
    attach_history(&r);
    return r;
  }
  StageReport SolveLVI(std::vector<double>* state, bool camera_surfel) {   // trajInitFromLVIdata
    check(lvx_set_imu(ctx_, (int)in_.imu_t.size(), in_.imu_t.data(), in_.gyro.data(), in_.acc.data(), opt_.w_gyro, opt_.w_acc));
    set_surfels();
    check(lvx_set_reproj(ctx_, (int)in_.obs_landmark.size(), in_.obs_landmark.data(), in_.obs_uv.data(), in_.obs_t0.data(), /*huber*/ opt_.w_cam, /*weight*/ 1.0));   // argument swap of :525
    StageReport r{camera_surfel ? "trajInitFromLVIdata+lm_splane" : "trajInitFromLVIdata", {}};
    if (camera_surfel) {
      // associateVisualPointsWithPlanes with q_LtoC / t_LinC from the current extrinsics
      const int N = in_.n_knots; const double* sl = state->data() + 7 * N + 16; const double* sc = state->data() + 7 * N + 24;
      double qLC[4], tLC[3]; relative(sc, sl, qLC, tLC);
      const int L = (int)in_.lm_t0.size(), np = (int)planes_.size();
      std::vector<double> p4((size_t)np * 4), bmin((size_t)np * 3), bmax((size_t)np * 3);
      for (int k = 0; k < np; ++k) { for (int a = 0; a < 4; ++a) p4[4 * k + a] = planes_[k].p4[a]; for (int a = 0; a < 3; ++a) { bmin[3 * k + a] = planes_[k].box_min[a]; bmax[3 * k + a] = planes_[k].box_max[a]; } }
      std::vector<int32_t> pol((size_t)std::max(L, 1), -1);
      check(lvx_landmark_assoc(ctx_, state->data(), qLC, tLC, in_.map_time, np, p4.data(), bmin.data(), bmax.data(), opt_.associated_radius, pol.data()));
      std::vector<int32_t> lm, pl;
      for (int l = 0; l < L; ++l) if (pol[l] >= 0) { lm.push_back(l); pl.push_back(pol[l]); }
      check(lvx_set_camsurf(ctx_, (int)lm.size(), lm.data(), pl.data(), in_.map_time, 5.0, opt_.w_cam_surfel));
      check(lvx_set_locks(ctx_, StageLocks(Stage::TrajFromLVILandmarksOnly, opt_.opt_time_offset)));
      r.n_cam_surfel = (int)lm.size();
    } else {
      check(lvx_set_camsurf(ctx_, 0, nullptr, nullptr, in_.map_time, 5.0, opt_.w_cam_surfel));
      check(lvx_set_locks(ctx_, StageLocks(Stage::TrajFromLVI, opt_.opt_time_offset)));
    }
    r.lm = solve(state, 80);
    attach_history(&r);
    r.n_planes = (int)planes_.size(); r.n_surfel_points = n_surfel_used_;
    return r;
  }
  // LiDAR pose in the camera frame from the two sensor blocks (q_XtoI, p_XinI): q_LtoC = q_CtoI^-1 q_LtoI, t_LinC = q_CtoI^-1 (p_LinI - p_CinI)
  static void relative(const double* cam, const double* lidar, double q[4], double t[3]) {
    const double cx = -cam[0], cy = -cam[1], cz = -cam[2], cw = cam[3];
    const double lx = lidar[0], ly = lidar[1], lz = lidar[2], lw = lidar[3];
    q[0] = cw * lx + cx * lw + cy * lz - cz * ly; q[1] = cw * ly + cy * lw + cz * lx - cx * lz; q[2] = cw * lz + cz * lw + cx * ly - cy * lx; q[3] = cw * lw - cx * lx - cy * ly - cz * lz;
    const double d[3] = {lidar[4] - cam[4], lidar[5] - cam[5], lidar[6] - cam[6]};
    const double ux = 2 * (cy * d[2] - cz * d[1]), uy = 2 * (cz * d[0] - cx * d[2]), uz = 2 * (cx * d[1] - cy * d[0]);
    t[0] = d[0] + cw * ux + (cy * uz - cz * uy); t[1] = d[1] + cw * uy + (cz * ux - cx * uz); t[2] = d[2] + cw * uz + (cx * uy - cy * ux);
  }
  void clear_families(bool surfel, bool reproj, bool camsurf) {
    if (surfel) check(lvx_set_surfel(ctx_, 0, nullptr, nullptr, nullptr, in_.map_time, 5.0, opt_.w_surfel));
    if (reproj) check(lvx_set_reproj(ctx_, 0, nullptr, nullptr, nullptr, opt_.w_cam, 1.0));
    if (camsurf) check(lvx_set_camsurf(ctx_, 0, nullptr, nullptr, in_.map_time, 5.0, opt_.w_cam_surfel));
  }
  lvx_lm_summary solve(std::vector<double>* state, int max_it) {
    lvx_lm_options o; lvx_lm_default_options(&o); o.max_iterations = max_it; o.verbose = opt_.verbose > 1;
    lvx_lm_summary s{};
    if (opt_.keep_history) stage_state_in_ = *state;
    check(lvx_lm_solve(ctx_, state->data(), &o, &s));
    const int cap = 4 * max_it + 8;
    hist_cost_.assign((size_t)cap, 0.0); hist_radius_.assign((size_t)cap, 0.0); hist_acc_.assign((size_t)cap, 0);
    const int k = lvx_lm_get_history(ctx_, cap, hist_cost_.data(), hist_radius_.data(), hist_acc_.data());
    hist_cost_.resize((size_t)std::max(k, 0)); hist_radius_.resize(hist_cost_.size()); hist_acc_.resize(hist_cost_.size());
    if (opt_.verbose) std::fprintf(stderr, "[lvx calibrate] solve: %d iterations, cost %.6e -> %.6e, termination %d\n", s.iterations, s.initial_cost, s.final_cost, s.termination);
    return s;
  }
  void attach_history(StageReport* r) const { r->cost_history = hist_cost_; r->radius_history = hist_radius_; r->accepted = hist_acc_; r->state_in = stage_state_in_; }
  void check(int rc) {
    if (rc == LVX_OK) return;
    const std::string msg = ctx_ ? lvx_last_error(ctx_) : "lvx error";
    if (rc == LVX_E_RANGE) throw std::range_error(msg);
    throw std::runtime_error(msg + " (lvx error " + std::to_string(rc) + ")");
  }
  lvx_ctx* ctx_ = nullptr;
  const CalibrateInput in_;
  CalibrateOptions opt_;
  std::vector<AssociationRecord> assoc_history_;
  std::vector<double> hist_cost_, hist_radius_, stage_state_in_; std::vector<int32_t> hist_acc_;
  std::vector<lvx_surfel_plane> planes_;
  std::vector<int32_t> key_scans_;
  std::vector<double> sp_pt_, sp_map_, sp_t_; std::vector<int32_t> sp_plane_;
  int n_surfel_used_ = 0;
};

}  // namespace lvx_host
