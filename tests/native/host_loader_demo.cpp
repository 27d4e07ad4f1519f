// tests/native/host_loader_demo.cpp — TEST-ONLY: parse an ORB-SLAM result file and an A-LOAM pose file with lvx_loaders.hpp and dump the arrays as text
#include <cstdio>
#include "../../lvi-exc_amd/host/lvx_loaders.hpp"

int main(int argc, char** argv) {
  if (argc < 6) { std::fprintf(stderr, "usage: %s orb.txt poses.txt cols rows border\n", argv[0]); return 2; }
  lvx_host::OrbResults r;
  if (!lvx_host::LoadOrbResults(argv[1], std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]), &r)) return 3;
  std::printf("frames %zu views %zu landmarks %zu observations %zu\n", r.frames.size(), r.view_stamp_ns.size(), r.landmark_id.size(), r.obs_landmark.size());
  for (size_t i = 0; i < r.landmark_id.size(); ++i) std::printf("L %lld %.17g %.17g %.17g %.17g\n", (long long)r.landmark_id[i], r.uv_ref[i][0], r.uv_ref[i][1], r.t0_ref[i], r.inverse_depth[i]);
  for (size_t i = 0; i < r.obs_landmark.size(); ++i) std::printf("O %d %.17g %.17g %.17g\n", r.obs_landmark[i], r.obs_uv[i][0], r.obs_uv[i][1], r.obs_t0[i]);
  lvx_host::LoamPoses p;
  if (!lvx_host::ReadPoseGT(argv[2], &p)) return 4;
  std::printf("poses %zu key %zu\n", p.all.size(), p.key.size());
  for (const auto& k : p.key) std::printf("K %lld\n", (long long)k.stamp_ns);
  // the pose Mapping() associates with scan idx (PoseOfScan): scans are stamped like the poses (scan idx <-> pose idx), scan 3 a little late, scan 7 without a pose of its stamp
  for (int mode = 0; mode < 2; ++mode)
    for (size_t idx = 0; idx < p.all.size(); ++idx) {
      double t = (double)p.all[idx].stamp_ns * 1e-9;
      if (idx == 3) t += 0.004;
      if (idx == 7) t += 0.06;
      double T[16];
      const bool ok = lvx_host::PoseOfScan(p, mode == 0, (int)idx, t, T);
      std::printf("A %d %zu %d %.17g %.17g %.17g %.17g\n", mode, idx, ok ? 1 : 0, ok ? T[3] : 0.0, ok ? T[7] : 0.0, ok ? T[0] : 0.0, ok ? T[1] : 0.0);
    }
  std::printf("locks %u %u %u %u\n", lvx_host::StageLocks(lvx_host::Stage::SO3FromGyro, false), lvx_host::StageLocks(lvx_host::Stage::TrajFromSurfel, false),
              lvx_host::StageLocks(lvx_host::Stage::TrajFromLVI, false), lvx_host::StageLocks(lvx_host::Stage::TrajFromLVILandmarksOnly, true));
  return 0;
}
