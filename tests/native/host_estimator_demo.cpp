// tests/native/host_estimator_demo.cpp — TEST-ONLY driver of the C++ host mirror (lvi-exc_amd/host/lvx_estimator.hpp): builds a problem
// the way TrajectoryManagerLVI::trajInitFromLVIdata does (AddMeasurement per measurement, Lock flags, Solve(max_iterations)), from a flat
// binary file written by tests/test_host_estimator.py, and writes the optimised state + summary back.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../lvi-exc_amd/host/lvx_estimator.hpp"

namespace {
struct Reader {
  FILE* f;
  double d() { double v = 0; if (fread(&v, 8, 1, f) != 1) { std::fprintf(stderr, "short read\n"); std::exit(3); } return v; }
  int i() { return static_cast<int>(d()); }
  std::vector<double> vec() { const int n = i(); std::vector<double> v(n); for (double& x : v) x = d(); return v; }
};
}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]); return 2; }
  Reader r{std::fopen(argv[1], "rb")};
  if (!r.f) return 2;
  const double t0 = r.d(), dt = r.d(); const int n_knots = r.i(); const unsigned locks = static_cast<unsigned>(r.i()); const int max_it = r.i();
  lvx_pinhole cam{};
  cam.rows = r.i(); cam.cols = r.i(); cam.readout = r.d(); cam.fx = r.d(); cam.fy = r.d(); cam.cx = r.d(); cam.cy = r.d();
  cam.k1 = r.d(); cam.k2 = r.d(); cam.p1 = r.d(); cam.p2 = r.d(); cam.k3 = r.d();
  const double w_gyro = r.d(), w_acc = r.d(), t_map = r.d(), huber_surf = r.d(), w_surf = r.d(), huber_rep = r.d(), w_rep = r.d();
  std::vector<double> state = r.vec(), t_imu = r.vec(), gyro = r.vec(), acc = r.vec(), planes = r.vec(), surf_pt = r.vec(), surf_t = r.vec(), surf_plane = r.vec(),
                      lm_uv = r.vec(), lm_t0 = r.vec(), rep_lm = r.vec(), rep_uv = r.vec(), rep_t0 = r.vec();
  std::fclose(r.f);
  try {
    lvx_host::TrajectoryEstimator est(0, t0, dt, n_knots, &state);
    est.SetCamera(cam);
    est.Lock(locks);
    for (size_t i = 0; i < t_imu.size(); ++i) {
      est.AddMeasurement(lvx_host::GyroscopeMeasurement{t_imu[i], {gyro[3 * i], gyro[3 * i + 1], gyro[3 * i + 2]}}, w_gyro);
      est.AddMeasurement(lvx_host::AccelerometerMeasurement{t_imu[i], {acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]}}, w_acc);
    }
    std::vector<std::array<double, 3>> pl(planes.size() / 3);
    for (size_t i = 0; i < pl.size(); ++i) pl[i] = {planes[3 * i], planes[3 * i + 1], planes[3 * i + 2]};
    est.SetPlanes(pl);
    for (size_t i = 0; i < surf_t.size(); ++i)
      est.AddMeasurement(lvx_host::LiDARSurfelPoint{{surf_pt[3 * i], surf_pt[3 * i + 1], surf_pt[3 * i + 2]}, static_cast<int32_t>(surf_plane[i]), surf_t[i]}, t_map, huber_surf, w_surf);
    std::vector<std::array<double, 2>> uv(lm_t0.size());
    for (size_t i = 0; i < uv.size(); ++i) uv[i] = {lm_uv[2 * i], lm_uv[2 * i + 1]};
    est.SetLandmarks(uv, lm_t0);
    for (size_t i = 0; i < rep_t0.size(); ++i)
      est.AddMeasurement(lvx_host::StaticRsCameraMeasurement{static_cast<int32_t>(rep_lm[i]), {rep_uv[2 * i], rep_uv[2 * i + 1]}, rep_t0[i]}, huber_rep, w_rep);
    const lvx_host::Summary s = est.Solve(max_it, false);
    std::printf("%s\n", s.BriefReport().c_str());
    FILE* o = std::fopen(argv[2], "wb");
    const double hdr[5] = {static_cast<double>(s.lm.iterations), static_cast<double>(s.lm.termination), s.lm.initial_cost, s.lm.final_cost, static_cast<double>(s.lm.successful_steps)};
    std::fwrite(hdr, 8, 5, o);
    std::fwrite(state.data(), 8, state.size(), o);
    std::fclose(o);
  } catch (const std::range_error& e) { std::fprintf(stderr, "range_error: %s\n", e.what()); return 4;
  } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 5; }
  return 0;
}
