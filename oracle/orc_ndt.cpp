// TEST INFRASTRUCTURE — CPU restatement (oracle) of the pieces of ndt_omp's registration loop that orc_upstream.cpp does not hold yet.  Nothing in the product
// (lvi-exc_amd/, include/) may include, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
//
// Follows, line by line:
//   orc_voxel_lookup_rel   VoxelGridCovariance::getNeighborhoodAtPoint(relative_coordinates, ...)   /root/reference/src/ndt_omp/include/pclomp/voxel_grid_covariance_omp_impl.hpp:378-408
//   orc_ndt_matrix         the float transform built from a 6-vector (Translation * AngleAxis X * Y * Z)   /root/reference/src/ndt_omp/include/pclomp/ndt_omp_impl.hpp:826-829, 870-873
//   orc_ndt_hessian        computeHessian + updateHessian + the DOUBLE computePointDerivatives           ndt_omp_impl.hpp:540-645, 443-480 (tables of :329-336, 351-370)
//   orc_ndt_euler012       Eigen::Matrix3f::eulerAngles(0, 1, 2) of the guess                              ndt_omp_impl.hpp:103-111 (Eigen 3.3 algorithm, out of tree)
//   orc_transform_cloud    pcl::transformPointCloud (float 4x4 * float point; PCL <= 1.8 scalar form, out of tree)   ndt_omp_impl.hpp:832, 877
//   orc_fitness            pcl::Registration::getFitnessScore (mean squared nearest-neighbour distance, float L2; out of tree)   /root/reference/src/ndt_omp/apps/align.cpp:30
// The Newton / More-Thuente loop itself (computeTransformation :81-171, computeStepLengthMT :772-931, updateIntervalMT :648-685, trialValueSelectionMT :689-768)
// is oracle/ndt_align.py over these functions.
//
// PINNED: oracle/ndt_align.py reproduces the fitness scores the reference publishes for its own two scans (README.md:21,26 — DIRECT7 0.214205, DIRECT1 0.208511);
// tests/test_ndt_align_oracle.py holds that.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <vector>

extern "C" {

// ids[q][r] = leaf index (into the orc_voxel_build arrays) of the cell at displacement rel3[r] from the query's cell, or -1: outside the grid, no leaf, or fewer than
// min_pts points (:396-403).  The reference pushes the hits in this order and skips the misses; callers walk a row and skip the -1s.
void orc_voxel_lookup_rel(int nq, const float* xyzi, float leaf, int min_pts, const int32_t* grid, int n_leaves, const int32_t* leaf_key, const int32_t* leaf_n, int n_rel,
                          const int32_t* rel3, int32_t* ids) {
  std::map<int, int> key2leaf;
  for (int i = 0; i < n_leaves; ++i) key2leaf[leaf_key[i]] = i;
  for (int q = 0; q < nq; ++q) {
    const float* p = xyzi + 4 * q;
    const int ijk[3] = {static_cast<int>(std::floor(p[0] / leaf)), static_cast<int>(std::floor(p[1] / leaf)), static_cast<int>(std::floor(p[2] / leaf))};   // :383-385
    for (int r = 0; r < n_rel; ++r) {
      const int32_t* d = rel3 + 3 * r;
      int id = -1;
      bool in = true;
      for (int a = 0; a < 3; ++a) in = in && (grid[a] - ijk[a] <= d[a]) && (grid[3 + a] - ijk[a] >= d[a]);                                                  // :386-387, 396
      if (in) {
        const int key = (ijk[0] + d[0] - grid[0]) * grid[9] + (ijk[1] + d[1] - grid[1]) * grid[10] + (ijk[2] + d[2] - grid[2]) * grid[11];                  // :398
        auto it = key2leaf.find(key);
        if (it != key2leaf.end() && leaf_n[it->second] >= min_pts) id = it->second;                                                                          // :399
      }
      ids[static_cast<size_t>(n_rel) * q + r] = id;
    }
  }
}

// Eigen::AngleAxis<float>::toRotationMatrix for a unit axis (Eigen/src/Geometry/AngleAxis.h): diagonal = (1 - c) a_i a_i + c, off-diagonals tmp -+ s a_k
static void axis_rot(float angle, int axis, float R[3][3]) {
  const float s = std::sin(angle), c = std::cos(angle);
  float a[3] = {0, 0, 0}; a[axis] = 1.0f;
  const float sa[3] = {s * a[0], s * a[1], s * a[2]}, c1[3] = {(1.0f - c) * a[0], (1.0f - c) * a[1], (1.0f - c) * a[2]};
  float tmp;
  tmp = c1[0] * a[1]; R[0][1] = tmp - sa[2]; R[1][0] = tmp + sa[2];
  tmp = c1[0] * a[2]; R[0][2] = tmp + sa[1]; R[2][0] = tmp - sa[1];
  tmp = c1[1] * a[2]; R[1][2] = tmp - sa[0]; R[2][1] = tmp + sa[0];
  for (int i = 0; i < 3; ++i) R[i][i] = c1[i] * a[i] + c;
}
static void mul33(const float A[3][3], const float B[3][3], float C[3][3]) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[i][j] = (A[i][0] * B[0][j] + A[i][1] * B[1][j]) + A[i][2] * B[2][j];
}
// (Translation<float,3>(p0,p1,p2) * AngleAxis<float>(p3, X) * AngleAxis<float>(p4, Y) * AngleAxis<float>(p5, Z)).matrix(), row-major 4 x 4
void orc_ndt_matrix(const double* p6, float* M16) {
  float Rx[3][3], Ry[3][3], Rz[3][3], Rxy[3][3], R[3][3];
  axis_rot(static_cast<float>(p6[3]), 0, Rx); axis_rot(static_cast<float>(p6[4]), 1, Ry); axis_rot(static_cast<float>(p6[5]), 2, Rz);
  mul33(Rx, Ry, Rxy); mul33(Rxy, Rz, R);
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) M16[4 * i + j] = R[i][j]; M16[4 * i + 3] = static_cast<float>(p6[i]); }
  M16[12] = M16[13] = M16[14] = 0.0f; M16[15] = 1.0f;
}

// Eigen 3.3 MatrixBase::eulerAngles(0, 1, 2) on the float rotation block of a row-major 4 x 4 (i, j, k = 0, 1, 2; "odd" = 0): angles such that R = Rx(e0) Ry(e1) Rz(e2)
void orc_ndt_euler012(const float* M16, float* e3) {
  auto m = [&](int r, int c) { return M16[4 * r + c]; };
  const float pi = static_cast<float>(M_PI);
  float r0 = std::atan2(m(1, 2), m(2, 2));
  const float c2 = std::sqrt(m(0, 0) * m(0, 0) + m(0, 1) * m(0, 1));
  float r1;
  if (r0 > 0.0f) { r0 -= pi; r1 = std::atan2(-m(0, 2), -c2); }   // (!odd && res[0] > 0): res[0] > 0 here, so it is moved down by pi
  else r1 = std::atan2(-m(0, 2), c2);
  const float s1 = std::sin(r0), c1 = std::cos(r0);
  const float r2 = std::atan2(s1 * m(2, 0) - c1 * m(1, 0), c1 * m(1, 1) - s1 * m(2, 1));
  e3[0] = -r0; e3[1] = -r1; e3[2] = -r2;                           // if (!odd) res = -res
}

// pcl::transformPointCloud(cloud_in, cloud_out, Matrix4f) for a dense cloud: out = M(0..2, 0..2) * p + M(0..2, 3), summed left to right in float; intensity kept
void orc_transform_cloud(int n, const float* in_xyzi, const float* M16, float* out_xyzi) {
  for (int i = 0; i < n; ++i) {
    const float* p = in_xyzi + 4 * i; float* o = out_xyzi + 4 * i;
    for (int r = 0; r < 3; ++r) o[r] = ((M16[4 * r] * p[0] + M16[4 * r + 1] * p[1]) + M16[4 * r + 2] * p[2]) + M16[4 * r + 3];
    o[3] = p[3];
  }
}

// computeHessian (:540-609) with updateHessian (:613-644) and the double computePointDerivatives (:443-480): everything in double, points and cells in order.
// ids: n_rel leaf ids per transformed point (orc_voxel_lookup_rel / orc_voxel_lookup7 of trans), -1 = skipped.  p6 = the vector the LAST computeAngleDerivatives ran on
// (computeHessian does not recompute the angular tables: :561).
void orc_ndt_hessian(int n, const float* input_xyzi, const float* trans_xyzi, int n_rel, const int32_t* ids, const double* mean, const double* icov, const double* p6,
                     double resolution, double outlier_ratio, double* hess36) {
  const double gauss_c1 = 10.0 * (1 - outlier_ratio), gauss_c2 = outlier_ratio / std::pow(resolution, 3);
  const double gauss_d3 = -std::log(gauss_c2), gauss_d1 = -std::log(gauss_c1 + gauss_c2) - gauss_d3;
  const double gauss_d2 = -2 * std::log((-std::log(gauss_c1 * std::exp(-0.5) + gauss_c2) - gauss_d3) / gauss_d1);
  double cx, cy, cz, sx, sy, sz;
  if (std::fabs(p6[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p6[3]); sx = std::sin(p6[3]); }
  if (std::fabs(p6[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p6[4]); sy = std::sin(p6[4]); }
  if (std::fabs(p6[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p6[5]); sz = std::sin(p6[5]); }
  const double ja[8][3] = {{-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy}, {cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy}, {-sy * cz, sy * sz, cy},
                           {sx * cy * cz, -sx * cy * sz, sx * sy}, {-cx * cy * cz, cx * cy * sz, -cx * sy}, {-cy * sz, -cy * cz, 0}, {cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0},
                           {sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0}};
  const double ha[15][3] = {{-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy}, {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy}, {cx * cy * cz, -cx * cy * sz, cx * sy},
                            {sx * cy * cz, -sx * cy * sz, sx * sy}, {-sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0}, {cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0},
                            {-cy * cz, cy * sz, sy}, {-sx * sy * cz, sx * sy * sz, sx * cy}, {cx * sy * cz, -cx * sy * sz, -cx * cy}, {sy * sz, sy * cz, 0}, {-sx * cy * sz, -sx * cy * cz, 0},
                            {cx * cy * sz, cx * cy * cz, 0}, {-cy * cz, cy * sz, 0}, {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0}, {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0}};
  double H[36];
  for (int e = 0; e < 36; ++e) H[e] = 0.0;
  auto dot3 = [](const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; };
  for (int idx = 0; idx < n; ++idx) {
    const float* xi = input_xyzi + 4 * idx; const float* xt = trans_xyzi + 4 * idx;
    const double x[3] = {xi[0], xi[1], xi[2]};
    double pg[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
    pg[1][3] = dot3(x, ja[0]); pg[2][3] = dot3(x, ja[1]); pg[0][4] = dot3(x, ja[2]); pg[1][4] = dot3(x, ja[3]); pg[2][4] = dot3(x, ja[4]);
    pg[0][5] = dot3(x, ja[5]); pg[1][5] = dot3(x, ja[6]); pg[2][5] = dot3(x, ja[7]);
    double ph[6][3][6];   // ph[i][.][j] = block<3, 1>(3 i, j)
    for (int i = 0; i < 6; ++i) for (int k = 0; k < 3; ++k) for (int j = 0; j < 6; ++j) ph[i][k][j] = 0.0;
    const double a[3] = {0, dot3(x, ha[0]), dot3(x, ha[1])}, b[3] = {0, dot3(x, ha[2]), dot3(x, ha[3])}, c[3] = {0, dot3(x, ha[4]), dot3(x, ha[5])};
    const double d[3] = {dot3(x, ha[6]), dot3(x, ha[7]), dot3(x, ha[8])}, e[3] = {dot3(x, ha[9]), dot3(x, ha[10]), dot3(x, ha[11])}, f[3] = {dot3(x, ha[12]), dot3(x, ha[13]), dot3(x, ha[14])};
    for (int k = 0; k < 3; ++k) { ph[3][k][3] = a[k]; ph[4][k][3] = b[k]; ph[5][k][3] = c[k]; ph[3][k][4] = b[k]; ph[4][k][4] = d[k]; ph[5][k][4] = e[k]; ph[3][k][5] = c[k]; ph[4][k][5] = e[k]; ph[5][k][5] = f[k]; }
    for (int nb = 0; nb < n_rel; ++nb) {
      const int li = ids[static_cast<size_t>(n_rel) * idx + nb];
      if (li < 0) continue;
      const double xd[3] = {static_cast<double>(xt[0]) - mean[3 * li], static_cast<double>(xt[1]) - mean[3 * li + 1], static_cast<double>(xt[2]) - mean[3 * li + 2]};
      const double* ci = icov + 9 * li;
      double cxv[3];
      for (int r = 0; r < 3; ++r) cxv[r] = (ci[3 * r] * xd[0] + ci[3 * r + 1] * xd[1]) + ci[3 * r + 2] * xd[2];                      // c_inv * x_trans
      double e_x = gauss_d2 * std::exp(-gauss_d2 * dot3(xd, cxv) / 2);                                                                // :621
      if (e_x > 1 || e_x < 0 || e_x != e_x) continue;                                                                                  // :624
      e_x *= gauss_d1;                                                                                                                 // :628
      double cpg[6][3];   // c_inv * point_gradient.col(j)
      for (int j = 0; j < 6; ++j) for (int r = 0; r < 3; ++r) cpg[j][r] = (ci[3 * r] * pg[0][j] + ci[3 * r + 1] * pg[1][j]) + ci[3 * r + 2] * pg[2][j];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
          const double phv[3] = {ph[i][0][j], ph[i][1][j], ph[i][2][j]};
          double cph[3];
          for (int r = 0; r < 3; ++r) cph[r] = (ci[3 * r] * phv[0] + ci[3 * r + 1] * phv[1]) + ci[3 * r + 2] * phv[2];
          const double pgj[3] = {pg[0][j], pg[1][j], pg[2][j]};
          H[6 * i + j] += e_x * (-gauss_d2 * dot3(xd, cpg[i]) * dot3(xd, cpg[j]) + dot3(xd, cph) + dot3(pgj, cpg[i]));               // :638-640
        }
    }
  }
  for (int e = 0; e < 36; ++e) hess36[e] = H[e];
}

// getFitnessScore(max_range): source transformed by the final transformation, nearest target point by float squared L2 (FLANN L2_Simple: the three squared
// differences summed in order; an exact search returns the minimum of exactly these values), distances <= max_range summed in double, mean over the counted points.
double orc_fitness(int n_src, const float* src_xyzi, const float* M16, int n_tgt, const float* tgt_xyzi, double max_range, int32_t* nn_index) {
  std::vector<float> tr(static_cast<size_t>(std::max(n_src, 1)) * 4);
  orc_transform_cloud(n_src, src_xyzi, M16, tr.data());
  double sum = 0.0; int nr = 0;
  std::vector<float> best(static_cast<size_t>(std::max(n_src, 1)));
  std::vector<int32_t> bi(static_cast<size_t>(std::max(n_src, 1)), -1);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n_src; ++i) {
    const float* p = tr.data() + 4 * i;
    float b = std::numeric_limits<float>::max(); int32_t arg = -1;
    for (int j = 0; j < n_tgt; ++j) {
      const float* q = tgt_xyzi + 4 * j;
      const float d0 = p[0] - q[0], d1 = p[1] - q[1], d2 = p[2] - q[2];
      const float d = (d0 * d0 + d1 * d1) + d2 * d2;
      if (d < b) { b = d; arg = j; }
    }
    best[i] = b; bi[i] = arg;
  }
  for (int i = 0; i < n_src; ++i) { if (nn_index) nn_index[i] = bi[i]; if (n_tgt > 0 && best[i] <= max_range) { sum += best[i]; ++nr; } }
  return nr > 0 ? sum / nr : std::numeric_limits<double>::max();
}

}  // extern "C"
