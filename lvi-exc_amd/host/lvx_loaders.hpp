// lvx_loaders.hpp — text-format loaders of the reference's offline driver, header-only, no device code.
//
// The reference calibrates from files written by its own front ends:
//   * ORB-SLAM2 results  (writer: src/lvi_exc/test/write_orb_slam_results.cpp:131-184)
//         FramePose <stamp_ns> tx ty tz qx qy qz qw        camera centre + orientation per key frame
//         UV <stamp_ns> u v <landmark id> u v <id> ...      undistorted key points of the frame's map points
//         MapPoint <id> x y z <reference stamp_ns>          position in the REFERENCE key frame's camera frame
//   * A-LOAM poses       (writer: src/aloam/src/laserMapping.cpp:891-900)
//         <stamp_ns> x y z qw qx qy qz
// and LIinitializer::LoadOrbResults / ReadPoseGT (src/lvi_exc/test/lvi_initialize_surfel_orb.cpp:337-450, 453-516) turn them into
// kontiki::sfm views / landmarks / observations and pose lists.  These functions restate that parsing and hand back the flat arrays
// lvx_set_landmarks / lvx_set_reproj take (landmarks in id order, as landmark_db_ is a std::map; observations landmark-major with the
// views in stamp order, as views_db_ is a std::map).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/lvx.h"

namespace lvx_host {

inline std::vector<std::string> SplitString(const std::string& s, char sep) {   // same tokens as the reference helper: empty fields are dropped
  std::vector<std::string> out; std::string cur;
  for (char ch : s) { if (ch == sep) { if (!cur.empty()) out.push_back(cur); cur.clear(); } else if (ch != '\r') cur.push_back(ch); }
  if (!cur.empty()) out.push_back(cur);
  return out;
}

struct CameraFrame { int64_t stamp_ns; std::array<double, 3> t; std::array<double, 4> q_xyzw; };
struct OrbResults {
  std::vector<CameraFrame> frames;                                   // FramePose lines, file order (integration_frames_cam_)
  std::vector<int64_t> view_stamp_ns;                                // views_db_ keys, ascending
  std::vector<int64_t> landmark_id;                                  // landmark_db_ keys, ascending
  std::vector<std::array<double, 2>> uv_ref; std::vector<double> t0_ref, inverse_depth;
  std::vector<int32_t> obs_landmark;                                 // index into the landmark arrays
  std::vector<std::array<double, 2>> obs_uv; std::vector<double> obs_t0;
};

// border: LIinitializer::border_filter_uv_ ; cols / rows: camera_params_.col / .row
inline bool LoadOrbResults(const std::string& path, int cols, int rows, int border, OrbResults* out) {
  std::ifstream ifs(path);
  if (!ifs.is_open()) return false;
  std::map<int64_t, std::map<int64_t, std::array<double, 2>>> uv_points;   // frame -> landmark -> uv
  struct Lm { std::array<double, 2> uv; int64_t ref; double rho; };
  std::map<int64_t, Lm> landmark_db;
  std::string line;
  auto to_i64 = [](const std::string& s) { std::istringstream iss(s); int64_t v = 0; iss >> v; return v; };
  while (std::getline(ifs, line)) {
    const std::vector<std::string> tok = SplitString(line, ' ');
    if (tok.empty()) break;
    if (tok[0] == "FramePose") {
      if (tok.size() != 9) return false;
      CameraFrame f; f.stamp_ns = to_i64(tok[1]);
      for (int k = 0; k < 3; ++k) f.t[k] = std::atof(tok[2 + k].c_str());
      for (int k = 0; k < 4; ++k) f.q_xyzw[k] = std::atof(tok[5 + k].c_str());
      out->frames.push_back(f);
    } else if (tok[0] == "UV") {
      if ((tok.size() - 2) % 3 != 0) return false;
      const int64_t frameid = to_i64(tok[1]);
      std::map<int64_t, std::array<double, 2>> obs;
      for (size_t i = 0; i < (tok.size() - 2) / 3; ++i) {
        const size_t idx = 2 + 3 * i;
        obs[to_i64(tok[idx + 2])] = {std::atof(tok[idx].c_str()), std::atof(tok[idx + 1].c_str())};
      }
      uv_points[frameid] = obs;
    } else if (tok[0] == "MapPoint") {
      if (tok.size() != 6) return false;
      const int64_t lm_id = to_i64(tok[1]), ref_id = to_i64(tok[5]);
      const double z = std::atof(tok[4].c_str());
      const auto vref = uv_points.find(ref_id);                       // views_db_ and uv_points are filled by the same UV lines
      if (vref == uv_points.end()) continue;
      const auto it_uv = vref->second.find(lm_id);
      if (it_uv == vref->second.end()) continue;
      const std::array<double, 2> euv = it_uv->second;
      if (euv[0] < border || euv[1] < border || euv[0] > cols - border || euv[1] > rows - border) continue;   // key points near the image border are dropped
      if (landmark_db.find(lm_id) != landmark_db.end()) continue;
      landmark_db[lm_id] = Lm{euv, ref_id, 1.0 / (z + 1e-15)};
    }
  }
  for (const auto& v : uv_points) out->view_stamp_ns.push_back(v.first);
  for (const auto& kv : landmark_db) {
    const int32_t li = static_cast<int32_t>(out->landmark_id.size());
    out->landmark_id.push_back(kv.first);
    out->uv_ref.push_back(kv.second.uv);
    out->t0_ref.push_back(static_cast<double>(kv.second.ref) * 1e-9);
    out->inverse_depth.push_back(kv.second.rho);
    for (const auto& v : uv_points) {                                  // every other view that saw the landmark, in stamp order
      if (v.first == kv.second.ref) continue;
      const auto it = v.second.find(kv.first);
      if (it == v.second.end()) continue;
      out->obs_landmark.push_back(li); out->obs_uv.push_back(it->second); out->obs_t0.push_back(static_cast<double>(v.first) * 1e-9);
    }
  }
  return true;
}

struct PoseStamped { int64_t stamp_ns; std::array<double, 3> p; std::array<double, 4> q_wxyz; };
struct LoamPoses { std::vector<PoseStamped> all, key; };
// ReadPoseGT: every pose goes to `all` (loam_poses_); `key` (integration_frames_lidar_) keeps a pose when it rotated >= 5 deg or moved >= 0.1 m
// since the last kept one
inline bool ReadPoseGT(const std::string& path, LoamPoses* out) {
  std::ifstream ifs(path);
  if (!ifs.is_open()) return false;
  std::string line;
  while (std::getline(ifs, line)) {
    const std::vector<std::string> w = SplitString(line, ' ');
    if (w.size() != 8) break;
    PoseStamped ps; { std::istringstream iss(w[0]); iss >> ps.stamp_ns; }
    for (int k = 0; k < 3; ++k) ps.p[k] = std::atof(w[1 + k].c_str());
    for (int k = 0; k < 4; ++k) ps.q_wxyz[k] = std::atof(w[4 + k].c_str());
    out->all.push_back(ps);
    if (!out->key.empty()) {
      const PoseStamped& l = out->key.back();
      // Eigen::Quaternion::angularDistance of the UNNORMALISED file quaternions: 2 atan2(|vec(d)|, |d.w|), d = a * conj(b)
      const double* a = l.q_wxyz.data(); const double* b = ps.q_wxyz.data();
      const double dw = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
      const double dx = -a[0] * b[1] + a[1] * b[0] - a[2] * b[3] + a[3] * b[2];
      const double dy = -a[0] * b[2] + a[1] * b[3] + a[2] * b[0] - a[3] * b[1];
      const double dz = -a[0] * b[3] - a[1] * b[2] + a[2] * b[1] + a[3] * b[0];
      const double ang = 2.0 * std::atan2(std::sqrt(dx * dx + dy * dy + dz * dz), std::fabs(dw));
      const double dp = std::sqrt((l.p[0] - ps.p[0]) * (l.p[0] - ps.p[0]) + (l.p[1] - ps.p[1]) * (l.p[1] - ps.p[1]) + (l.p[2] - ps.p[2]) * (l.p[2] - ps.p[2]));
      if (ang * 180.0 / M_PI < 5.0 && dp < 0.1) continue;
    }
    out->key.push_back(ps);
  }
  return true;
}

// The pose Mapping() feeds LiDAROdometry with for scan `idx` (lvi_initialize_surfel_orb.cpp:1283-1295), as the row-major 4 x 4 Eigen builds from the file's numbers
// (Quaterniond(w, x, y, z).toRotationMatrix() of the UNNORMALISED quaternion).  simulation: the pose with exactly the scan's stamp — int64(scan_t * 1e9) looked up in
// loam_poses_map_, whose operator[] keeps the LAST pose of a stamp; false when there is none (the scan is skipped).  Otherwise findAssociatedPose (:1246-1260): the nearest in
// time of loam_poses_[idx - 5 .. idx + 4], the last pose of the file never looked at (`idx >= size() - 1`), and used even when it is further than 0.02 s away (the reference
// computes `ok` and does not look at it); false only when the window holds no pose at all (the reference would then feed an uninitialised matrix).
inline bool PoseOfScan(const LoamPoses& loam, bool simulation, int idx, double scan_t, double T[16]) {
  const PoseStamped* hit = nullptr;
  if (simulation) {
    const int64_t stamp = (int64_t)(scan_t * 1e9);
    for (const PoseStamped& ps : loam.all) if (ps.stamp_ns == stamp) hit = &ps;
  } else {
    double best = 1.7976931348623157e308;
    for (int i = -5; i < 5; ++i) {
      const long long k = (long long)i + idx;
      if (k < 0 || k >= (long long)loam.all.size() - 1) continue;
      const double d = std::fabs((double)loam.all[(size_t)k].stamp_ns * 1e-9 - scan_t);
      if (d < best) { best = d; hit = &loam.all[(size_t)k]; }
    }
  }
  if (!hit) return false;
  const double w = hit->q_wxyz[0], x = hit->q_wxyz[1], y = hit->q_wxyz[2], z = hit->q_wxyz[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c]; T[4 * r + 3] = hit->p[r]; }
  T[12] = T[13] = T[14] = 0; T[15] = 1;
  return true;
}

// Lock masks of the reference's solve stages (TrajectoryManagerLVI, src/lvi_exc/src/core/trajectory_manager_lvi.cpp): which Lock* calls each
// stage makes before building its estimator.  opt_time_offset = calib_param_manager->opt_time_offset (lvi.yaml:32).
enum class Stage { SO3FromGyro, TrajFromSurfel, TrajFromLVI, TrajFromLVILandmarksOnly, TrajFromVisualFrames };
inline uint32_t StageLocks(Stage s, bool opt_time_offset) {
  const uint32_t tau = opt_time_offset ? 0u : (LVX_LOCK_LIDAR_TAU | LVX_LOCK_CAM_TAU);
  switch (s) {
    case Stage::SO3FromGyro:              // initialSO3TrajWithGyro (:43-62): SO3 spline + one orientation prior, biases still locked
      return LVX_LOCK_R3 | LVX_LOCK_ACC_BIAS | LVX_LOCK_GYRO_BIAS | LVX_LOCK_LIDAR_TAU | LVX_LOCK_CAM_TAU;
    case Stage::TrajFromSurfel:           // trajInitFromSurfel (:311-351): lidar free, camera locked
      return LVX_LOCK_CAM_Q | LVX_LOCK_CAM_P | LVX_LOCK_CAM_TAU | LVX_LOCK_LANDMARKS | (opt_time_offset ? 0u : LVX_LOCK_LIDAR_TAU);
    case Stage::TrajFromLVI:              // trajInitFromLVIdata (:138-195) with the trajectory free: everything estimated
      return tau;
    case Stage::TrajFromLVILandmarksOnly: // same with `traj_->Lock(true)`: trajectory and lidar locked, camera + landmarks refined (:148-152, 208-212)
      return tau | LVX_LOCK_TRAJ | LVX_LOCK_LIDAR_Q | LVX_LOCK_LIDAR_P;
    case Stage::TrajFromVisualFrames:     // trajInitFromVisualFrames (:99-136; LIinitializer::CIoptimize): IMU + reprojection blocks only, LiDAR extrinsics locked, its offset untouched (locked)
      return LVX_LOCK_LIDAR_Q | LVX_LOCK_LIDAR_P | LVX_LOCK_LIDAR_TAU | (opt_time_offset ? 0u : LVX_LOCK_CAM_TAU);
  }
  return tau;
}

}  // namespace lvx_host
