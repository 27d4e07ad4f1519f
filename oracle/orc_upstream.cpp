// oracle/orc_upstream.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED (no reference tests/fixtures exist;
// PCL / Eigen pieces are restated from public semantics).
//
// CPU restatements of the upstream point-cloud kernels on the north-star path:
//   orc_scan_register    A-LOAM scanRegistration core        /root/reference/src/aloam/src/scanRegistration.cpp:101-131,199-447
//   orc_voxel_build      ndt_omp VoxelGridCovariance filter  /root/reference/src/ndt_omp/include/pclomp/voxel_grid_covariance_omp_impl.hpp:49-374
//   orc_voxel_lookup7    getNeighborhoodAtPoint7             same file :378-438
//   orc_surfel_assoc     SurfelAssociation::getAssociation   /root/reference/src/lvi_exc/src/core/surfel_association.cpp:111-159,296-331
// Built with -ffp-contract=off so float expressions round exactly as written (scanRegistration is compared bit-exactly).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <vector>

namespace {
struct RsPoint { float x, y, z, pad; uint8_t intensity; uint8_t pad2; uint16_t ring; uint32_t pad3; double timestamp; };   // 32 B, scanRegistration.cpp:57-66
static_assert(sizeof(RsPoint) == 32, "RsPointXYZIRT layout");
struct P4 { float x, y, z, i; };
}  // namespace

extern "C" {

// Outputs (all sized n_in unless noted): cloud[n][4] (x,y,z,intensity = ring + relTime), curvature, label, sort_ind, picked;
// scan_start/scan_end[n_rings]; lists sharp / less_sharp / flat / less_flat (indices into cloud, reference push order) + counts[4].
// Returns the number of points kept (cloudSize).
int orc_scan_register(int n_in, const void* pts_v, int n_rings, float min_range, float* cloud, float* curvature, int32_t* label, int32_t* sort_ind,
                      int32_t* picked, int32_t* scan_start, int32_t* scan_end, int32_t* sharp, int32_t* less_sharp, int32_t* flat, int32_t* less_flat, int32_t* counts) {
  const RsPoint* in = static_cast<const RsPoint*>(pts_v);
  if (n_in <= 0) { for (int k = 0; k < 4; ++k) counts[k] = 0; return 0; }
  const double start_point_time = in[0].timestamp;                                  // :161
  // removeClosedPointCloud (:101-131)
  std::vector<RsPoint> kept; kept.reserve(n_in);
  for (int i = 0; i < n_in; ++i) {
    const RsPoint& p = in[i];
    if (p.x * p.x + p.y * p.y + p.z * p.z < min_range * min_range) continue;
    if (std::isnan(p.x) || std::isnan(p.y) || std::isnan(p.z)) continue;
    kept.push_back(p);
  }
  const int cloudSize = static_cast<int>(kept.size());
  // ring bucketing (:199-279)
  std::vector<std::vector<P4>> scans(n_rings);
  for (int i = 0; i < cloudSize; ++i) {
    P4 q; q.x = kept[i].x; q.y = kept[i].y; q.z = kept[i].z;
    const double relTime = kept[i].timestamp - start_point_time;
    q.i = static_cast<float>(kept[i].ring + relTime);                                // point.intensity = ring + relTime (double -> float)
    scans[kept[i].ring].push_back(q);
  }
  std::vector<P4> lc; lc.reserve(cloudSize);
  for (int i = 0; i < n_rings; ++i) {                                                 // :284-290
    scan_start[i] = static_cast<int>(lc.size()) + 5;
    lc.insert(lc.end(), scans[i].begin(), scans[i].end());
    scan_end[i] = static_cast<int>(lc.size()) - 6;
  }
  for (int i = 0; i < cloudSize; ++i) { cloud[4 * i] = lc[i].x; cloud[4 * i + 1] = lc[i].y; cloud[4 * i + 2] = lc[i].z; cloud[4 * i + 3] = lc[i].i; curvature[i] = 0; label[i] = 0; sort_ind[i] = i; picked[i] = 0; }
  for (int i = 5; i < cloudSize - 5; i++) {                                           // :295-305
    float diffX = lc[i - 5].x + lc[i - 4].x + lc[i - 3].x + lc[i - 2].x + lc[i - 1].x - 10 * lc[i].x + lc[i + 1].x + lc[i + 2].x + lc[i + 3].x + lc[i + 4].x + lc[i + 5].x;
    float diffY = lc[i - 5].y + lc[i - 4].y + lc[i - 3].y + lc[i - 2].y + lc[i - 1].y - 10 * lc[i].y + lc[i + 1].y + lc[i + 2].y + lc[i + 3].y + lc[i + 4].y + lc[i + 5].y;
    float diffZ = lc[i - 5].z + lc[i - 4].z + lc[i - 3].z + lc[i - 2].z + lc[i - 1].z - 10 * lc[i].z + lc[i + 1].z + lc[i + 2].z + lc[i + 3].z + lc[i + 4].z + lc[i + 5].z;
    curvature[i] = diffX * diffX + diffY * diffY + diffZ * diffZ;
  }
  int ns = 0, nls = 0, nf = 0, nlf = 0;
  auto gap2 = [&](int a, int b) { const float dx = lc[a].x - lc[b].x, dy = lc[a].y - lc[b].y, dz = lc[a].z - lc[b].z; return dx * dx + dy * dy + dz * dz; };
  for (int i = 0; i < n_rings; i++) {                                                 // :316-447
    if (scan_end[i] - scan_start[i] < 6) continue;
    for (int j = 0; j < 6; j++) {
      const int sp = scan_start[i] + (scan_end[i] - scan_start[i]) * j / 6;
      const int ep = scan_start[i] + (scan_end[i] - scan_start[i]) * (j + 1) / 6 - 1;
      std::sort(sort_ind + sp, sort_ind + ep + 1, [&](int a, int b) { return curvature[a] < curvature[b]; });   // comp (:87)
      int largestPickedNum = 0;
      for (int k = ep; k >= sp; k--) {
        const int ind = sort_ind[k];
        if (picked[ind] == 0 && curvature[ind] > 0.1) {
          largestPickedNum++;
          if (largestPickedNum <= 2) { label[ind] = 2; sharp[ns++] = ind; less_sharp[nls++] = ind; }
          else if (largestPickedNum <= 20) { label[ind] = 1; less_sharp[nls++] = ind; }
          else break;
          picked[ind] = 1;
          for (int l = 1; l <= 5; l++) { if (gap2(ind + l, ind + l - 1) > 0.05) break; picked[ind + l] = 1; }
          for (int l = -1; l >= -5; l--) { if (gap2(ind + l, ind + l + 1) > 0.05) break; picked[ind + l] = 1; }
        }
      }
      int smallestPickedNum = 0;
      for (int k = sp; k <= ep; k++) {
        const int ind = sort_ind[k];
        if (picked[ind] == 0 && curvature[ind] < 0.1) {
          label[ind] = -1; flat[nf++] = ind;
          smallestPickedNum++;
          if (smallestPickedNum >= 4) break;                                          // before the 4th is marked picked (:394-403)
          picked[ind] = 1;
          for (int l = 1; l <= 5; l++) { if (gap2(ind + l, ind + l - 1) > 0.05) break; picked[ind + l] = 1; }
          for (int l = -1; l >= -5; l--) { if (gap2(ind + l, ind + l + 1) > 0.05) break; picked[ind + l] = 1; }
        }
      }
      for (int k = sp; k <= ep; k++) if (label[k] <= 0) less_flat[nlf++] = k;         // pre-VoxelGrid (:430-438)
    }
  }
  counts[0] = ns; counts[1] = nls; counts[2] = nf; counts[3] = nlf;
  return cloudSize;
}

// ----------------------------------------------------------------------------------------------------------
// symmetric 3x3 eigen decomposition (cyclic Jacobi, double), eigenvalues ascending, columns = eigenvectors.
// Stands in for Eigen::SelfAdjointEigenSolver<Matrix3d>::compute (out-of-tree): compare with tolerances / up to sign.
// ----------------------------------------------------------------------------------------------------------
static void eig3(const double A[9], double evals[3], double evecs[9]) {
  double a[3][3] = {{A[0], A[1], A[2]}, {A[3], A[4], A[5]}, {A[6], A[7], A[8]}};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
      if (a[p][q] == 0.0) continue;
      const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
      const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
      for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
      for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
      for (int k = 0; k < 3; ++k) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq; }
    }
  }
  int idx[3] = {0, 1, 2};
  std::sort(idx, idx + 3, [&](int i, int j) { return a[i][i] < a[j][j]; });
  for (int k = 0; k < 3; ++k) { evals[k] = a[idx[k]][idx[k]]; for (int r = 0; r < 3; ++r) evecs[3 * r + k] = v[r][idx[k]]; }
}
static bool inv3(const double M[9], double R[9]) {
  const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
  const double id = 1.0 / det;
  R[0] = c00 * id; R[1] = (M[2] * M[7] - M[1] * M[8]) * id; R[2] = (M[1] * M[5] - M[2] * M[4]) * id;
  R[3] = c01 * id; R[4] = (M[0] * M[8] - M[2] * M[6]) * id; R[5] = (M[2] * M[3] - M[0] * M[5]) * id;
  R[6] = c02 * id; R[7] = (M[1] * M[6] - M[0] * M[7]) * id; R[8] = (M[0] * M[4] - M[1] * M[3]) * id;
  return true;
}

// grid[0..2] = min_b, grid[3..5] = max_b, grid[6..8] = div_b, grid[9..11] = divb_mul.  Leaves are returned in std::map (ascending key)
// order: leaf_key, leaf_n (nr_points, -1 if rejected), mean[3], cov[9], icov[9], evecs[9], evals[3], centroid[3] (float), and the
// per-leaf point lists as offsets[n_leaves + 1] into point_ids (input order).  Returns the number of leaves (all, incl. < min_points).
int orc_voxel_build(int n, const float* xyzi, float leaf, int min_pts, double eig_mult, int32_t* grid, int32_t* leaf_key, int32_t* leaf_n, double* mean,
                    double* cov, double* icov, double* evecs, double* evals, float* centroid, int32_t* offsets, int32_t* point_ids, int max_leaves) {
  const float inv = 1.0f / leaf;                                                       // setLeafSize: inverse_leaf_size_ = 1 / leaf_size_ (float)
  float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
  float mx[3] = {-mn[0], -mn[1], -mn[2]};
  for (int i = 0; i < n; ++i) {                                                        // pcl::getMinMax3D (skips non-finite points)
    const float* p = xyzi + 4 * i;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); }
  }
  int min_b[3], max_b[3], div_b[3], mul[3];
  for (int k = 0; k < 3; ++k) { min_b[k] = static_cast<int>(std::floor(mn[k] * inv)); max_b[k] = static_cast<int>(std::floor(mx[k] * inv)); div_b[k] = max_b[k] - min_b[k] + 1; }   // :86-95
  mul[0] = 1; mul[1] = div_b[0]; mul[2] = div_b[0] * div_b[1];                                                                                                                          // :102
  for (int k = 0; k < 3; ++k) { grid[k] = min_b[k]; grid[3 + k] = max_b[k]; grid[6 + k] = div_b[k]; grid[9 + k] = mul[k]; }
  struct Leaf { int n = 0; double s[3] = {0, 0, 0}; double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; float cen[3] = {0, 0, 0}; std::vector<int> pts; };
  std::map<size_t, Leaf> leaves;
  for (int i = 0; i < n; ++i) {                                                        // :211-267
    const float* p = xyzi + 4 * i;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    const int ijk0 = static_cast<int>(std::floor(p[0] * inv) - static_cast<float>(min_b[0]));
    const int ijk1 = static_cast<int>(std::floor(p[1] * inv) - static_cast<float>(min_b[1]));
    const int ijk2 = static_cast<int>(std::floor(p[2] * inv) - static_cast<float>(min_b[2]));
    const int idx = ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2];
    Leaf& l = leaves[idx];
    const double x[3] = {p[0], p[1], p[2]};
    for (int a = 0; a < 3; ++a) { l.s[a] += x[a]; for (int b = 0; b < 3; ++b) l.c[3 * a + b] += x[a] * x[b]; l.cen[a] += p[a]; }
    ++l.n; l.pts.push_back(i);
  }
  int li = 0, po = 0;
  for (auto& kv : leaves) {                                                            // :286-371
    if (li >= max_leaves) return -1;
    Leaf& l = kv.second;
    leaf_key[li] = static_cast<int32_t>(kv.first);
    offsets[li] = po; for (int id : l.pts) point_ids[po++] = id;
    double* M = mean + 3 * li; double* Cv = cov + 9 * li; double* IC = icov + 9 * li; double* EV = evecs + 9 * li; double* EL = evals + 3 * li;
    for (int a = 0; a < 3; ++a) { centroid[3 * li + a] = l.cen[a] / static_cast<float>(l.n); M[a] = l.s[a] / l.n; }
    for (int a = 0; a < 9; ++a) { Cv[a] = (a % 4 == 0) ? 1.0 : 0.0; IC[a] = 0.0; EV[a] = (a % 4 == 0) ? 1.0 : 0.0; }
    for (int a = 0; a < 3; ++a) EL[a] = 0.0;
    int nr = l.n;
    if (l.n >= min_pts) {
      double C[9];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[3 * a + b] = (l.c[3 * a + b] - 2 * (l.s[a] * M[b])) / l.n + M[a] * M[b];   // :333
      for (int a = 0; a < 9; ++a) C[a] *= (l.n - 1.0) / l.n;                                                                                // :334
      double ev[3], V[9];
      eig3(C, ev, V);
      for (int a = 0; a < 9; ++a) EV[a] = V[a];
      if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) { nr = -1; for (int a = 0; a < 9; ++a) Cv[a] = C[a]; }
      else {
        const double min_ev = eig_mult * ev[2];
        if (ev[0] < min_ev) {
          ev[0] = min_ev; if (ev[1] < min_ev) ev[1] = min_ev;
          // cov = evecs * diag(ev) * evecs^-1  (evecs orthonormal: inverse == transpose up to rounding)
          for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[3 * a + b] = V[3 * a] * ev[0] * V[3 * b] + V[3 * a + 1] * ev[1] * V[3 * b + 1] + V[3 * a + 2] * ev[2] * V[3 * b + 2];
        }
        for (int a = 0; a < 3; ++a) EL[a] = ev[a];
        for (int a = 0; a < 9; ++a) Cv[a] = C[a];
        inv3(C, IC);
        double mxv = IC[0], mnv = IC[0]; for (int a = 1; a < 9; ++a) { mxv = std::max(mxv, IC[a]); mnv = std::min(mnv, IC[a]); }
        if (mxv == std::numeric_limits<float>::infinity() || mnv == -std::numeric_limits<float>::infinity()) nr = -1;
      }
    }
    leaf_n[li] = nr;
    ++li;
  }
  offsets[li] = po;
  return li;
}

// getNeighborhoodAtPoint7 (:378-438): ids7[q][k] = leaf index (into the orc_voxel_build leaf arrays) for displacement k of
// {0, +x, -x, +y, -y, +z, -z}, or -1 (out of grid / empty / fewer than min_pts points incl. rejected leaves with n = -1).
void orc_voxel_lookup7(int nq, const float* xyzi, float leaf, int min_pts, const int32_t* grid, int n_leaves, const int32_t* leaf_key, const int32_t* leaf_n, int32_t* ids7) {
  std::map<int, int> key2leaf;
  for (int i = 0; i < n_leaves; ++i) key2leaf[leaf_key[i]] = i;
  static const int disp[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  for (int q = 0; q < nq; ++q) {
    const float* p = xyzi + 4 * q;
    const int ijk[3] = {static_cast<int>(std::floor(p[0] / leaf)), static_cast<int>(std::floor(p[1] / leaf)), static_cast<int>(std::floor(p[2] / leaf))};
    for (int k = 0; k < 7; ++k) {
      int id = -1;
      bool in = true;
      for (int a = 0; a < 3; ++a) in = in && (grid[a] - ijk[a] <= disp[k][a]) && (grid[3 + a] - ijk[a] >= disp[k][a]);
      if (in) {
        const int key = (ijk[0] + disp[k][0] - grid[0]) * grid[9] + (ijk[1] + disp[k][1] - grid[1]) * grid[10] + (ijk[2] + disp[k][2] - grid[2]) * grid[11];
        auto it = key2leaf.find(key);
        if (it != key2leaf.end() && leaf_n[it->second] >= min_pts) id = it->second;
      }
      ids7[7 * q + k] = id;
    }
  }
}

// SurfelAssociation::getAssociation (:111-138) with the SERIAL plane loop (ascending plane_id; later planes overwrite).
// scan: organised H x W, float xyzi in the map frame.  planes: p4[P][4], box_min[P][3], box_max[P][3] (double).
void orc_surfel_assoc(int H, int W, const float* scan, int P, const double* p4, const double* bmin, const double* bmax, double radius, int sel, int32_t* flag) {
  for (int i = 0; i < H * W; ++i) flag[i] = -1;
  std::vector<int> mask;
  for (int pid = 0; pid < P; ++pid) {
    const double* pl = p4 + 4 * pid; const double* lo = bmin + 3 * pid; const double* hi = bmax + 3 * pid;
    for (int h = 0; h < H; ++h) {
      mask.clear();
      for (int w = 0; w < W; ++w) {                                                    // associateScanToSurfel (:305-331)
        const float* p = scan + 4 * (static_cast<size_t>(h) * W + w);
        if (!std::isnan(p[0]) && p[0] > lo[0] && p[0] < hi[0] && p[1] > lo[1] && p[1] < hi[1] && p[2] > lo[2] && p[2] < hi[2]) {
          const double px = p[0], py = p[1], pz = p[2];
          double dist = px * pl[0] + py * pl[1] + pz * pl[2] + pl[3];                   // pt.dot(normal) + d (:299-300)
          dist = dist > 0 ? dist : -dist;
          if (dist <= radius) mask.push_back(w);
        }
      }
      if (static_cast<int>(mask.size()) < sel * 2) continue;                            // :126-127
      int step = static_cast<int>(mask.size()) / (sel + 1);
      step = std::max(step, 1);
      for (int s = 0; s < sel; ++s) flag[h * W + mask[step * (s + 1) - 1]] = pid;
    }
  }
}

// SurfelAssociation::setSurfelMap (/root/reference/src/lvi_exc/src/core/surfel_association.cpp:50-86) with checkPlaneType (:246-266): walk the
// NDT leaves in std::map (voxel key) order; keep a leaf with nr_points >= min_leaf_points whose planarity 2 (l_mid - l_min) / (l_min + l_mid + l_max)
// of the (inflated) NDT eigenvalues reaches p_lambda; fit a plane; Pi = -d n; AABB of ALL the leaf's points (pcl::getMinMax3D, float).
// DEVIATION (documented): fitPlane (:268-294) is pcl RANSAC + optimizeCoefficients — random sampling, float PCA.  Here the plane search is
// deterministic: start from the leaf's own PCA plane (normal = eigenvector of the smallest eigenvalue through the mean), take the points
// closer than dist_threshold, refit by PCA of those inliers in double (what pcl's optimizeModelCoefficients + final selectWithinDistance
// do after the sampling stage), reselect; reject below min_inliers.  p4 is signed so that d <= 0.
// planes: per accepted leaf 16 doubles {p4[4], Pi[3], box_min[3], box_max[3], leaf, n_points, n_inliers} + plane_type in types[]
int orc_surfel_extract(int n_leaves, const float* xyzi, const int32_t* leaf_n, const int32_t* offsets, const int32_t* point_ids, const double* mean, const double* evecs,
                       const double* evals, double p_lambda, double dist_threshold, int min_leaf_points, int min_inliers, double* planes16, int32_t* types, int max_planes) {
  int np = 0;
  for (int li = 0; li < n_leaves; ++li) {
    const int n = leaf_n[li];
    if (n < min_leaf_points) continue;
    const double* ev = evals + 3 * li;
    // Eigen::sort_vec: indices sorted by DESCENDING value (std::sort on 3 elements)
    int ind[3] = {0, 1, 2};
    std::sort(ind, ind + 3, [&](int a, int b) { return ev[a] > ev[b]; });
    const double p = 2.0 * (ev[ind[1]] - ev[ind[2]]) / (ev[ind[2]] + ev[ind[1]] + ev[ind[0]]);
    if (p < p_lambda) continue;
    const double* V = evecs + 9 * li;   // row-major, eigenvector k = column k
    double nrm[3] = {V[0 + ind[2]], V[3 + ind[2]], V[6 + ind[2]]};
    double an[3] = {std::fabs(nrm[0]), std::fabs(nrm[1]), std::fabs(nrm[2])};
    int ti[3] = {0, 1, 2};
    std::sort(ti, ti + 3, [&](int a, int b) { return an[a] > an[b]; });
    const int plane_type = ti[2];
    const int o = offsets[li], cnt = offsets[li + 1] - offsets[li];
    double d = -(nrm[0] * mean[3 * li] + nrm[1] * mean[3 * li + 1] + nrm[2] * mean[3 * li + 2]);
    int nin = 0;
    for (int pass = 0; pass < 2; ++pass) {
      double sm[3] = {0, 0, 0}, cc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      nin = 0;
      for (int k = 0; k < cnt; ++k) {
        const float* q = xyzi + 4 * point_ids[o + k];
        const double x[3] = {q[0], q[1], q[2]};
        if (!(std::fabs(nrm[0] * x[0] + nrm[1] * x[1] + nrm[2] * x[2] + d) < dist_threshold)) continue;
        ++nin;
        for (int a = 0; a < 3; ++a) { sm[a] += x[a]; for (int b = 0; b < 3; ++b) cc[3 * a + b] += x[a] * x[b]; }
      }
      if (pass == 1 || nin < 3) break;
      double mu[3] = {sm[0] / nin, sm[1] / nin, sm[2] / nin}, C[9];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[3 * a + b] = cc[3 * a + b] / nin - mu[a] * mu[b];
      double e2[3], V2[9];
      eig3(C, e2, V2);                                 // ascending: column 0 = normal
      nrm[0] = V2[0]; nrm[1] = V2[3]; nrm[2] = V2[6];
      d = -(nrm[0] * mu[0] + nrm[1] * mu[1] + nrm[2] * mu[2]);
    }
    if (nin < min_inliers) continue;
    if (d > 0 || (d == 0 && (nrm[0] < 0 || (nrm[0] == 0 && (nrm[1] < 0 || (nrm[1] == 0 && nrm[2] < 0)))))) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; d = -d; }
    float bmin[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()}, bmax[3] = {-bmin[0], -bmin[1], -bmin[2]};
    for (int k = 0; k < cnt; ++k) { const float* q = xyzi + 4 * point_ids[o + k]; for (int a = 0; a < 3; ++a) { bmin[a] = std::min(bmin[a], q[a]); bmax[a] = std::max(bmax[a], q[a]); } }
    if (np < max_planes) {
      double* P = planes16 + 16 * np;
      P[0] = nrm[0]; P[1] = nrm[1]; P[2] = nrm[2]; P[3] = d;
      for (int a = 0; a < 3; ++a) { P[4 + a] = -d * nrm[a]; P[7 + a] = bmin[a]; P[10 + a] = bmax[a]; }
      P[13] = li; P[14] = n; P[15] = nin;
      types[np] = plane_type;
    }
    ++np;
  }
  return np;
}

// pclomp::NormalDistributionsTransform::computeDerivatives with DIRECT7 (/root/reference/src/ndt_omp/include/pclomp/ndt_omp_impl.hpp:180-285),
// computeAngleDerivatives (:289-383), computePointDerivatives float form (:387-430), updateDerivatives (:484-536), Gaussian constants (:63-67).
// The per-point arithmetic is FLOAT as in the reference (Eigen float matrices; sums here run left to right — Eigen's vectorised order is not
// restated, so agreement with the real library is to float rounding), the accumulation over cells and points is double, points in index order.
// ids7: orc_voxel_lookup7 of the TRANSFORMED points; leaf arrays from orc_voxel_build.
// n_rel leaf ids per point (the neighbourhood in the reference's push order, -1 = no leaf): 7 for DIRECT7, 1 for DIRECT1, 26 for DIRECT26, any width for a radius search
void orc_ndt_derivatives_n(int n, const float* input_xyzi, const float* trans_xyzi, int n_rel, const int32_t* ids7, const double* mean, const double* icov, const double* p6,
                           double resolution, double outlier_ratio, int compute_hessian, double* score_out, double* grad6, double* hess36) {
  const double gauss_c1 = 10.0 * (1 - outlier_ratio), gauss_c2 = outlier_ratio / std::pow(resolution, 3);
  const double gauss_d3 = -std::log(gauss_c2), gauss_d1 = -std::log(gauss_c1 + gauss_c2) - gauss_d3;
  const double gauss_d2 = -2 * std::log((-std::log(gauss_c1 * std::exp(-0.5) + gauss_c2) - gauss_d3) / gauss_d1);
  double cx, cy, cz, sx, sy, sz;
  if (std::fabs(p6[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p6[3]); sx = std::sin(p6[3]); }
  if (std::fabs(p6[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p6[4]); sy = std::sin(p6[4]); }
  if (std::fabs(p6[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p6[5]); sz = std::sin(p6[5]); }
  const float j_ang[8][3] = {{(float)(-sx * sz + cx * sy * cz), (float)(-sx * cz - cx * sy * sz), (float)(-cx * cy)}, {(float)(cx * sz + sx * sy * cz), (float)(cx * cz - sx * sy * sz), (float)(-sx * cy)},
                             {(float)(-sy * cz), (float)(sy * sz), (float)cy}, {(float)(sx * cy * cz), (float)(-sx * cy * sz), (float)(sx * sy)},
                             {(float)(-cx * cy * cz), (float)(cx * cy * sz), (float)(-cx * sy)}, {(float)(-cy * sz), (float)(-cy * cz), 0.0f},
                             {(float)(cx * cz - sx * sy * sz), (float)(-cx * sz - sx * sy * cz), 0.0f}, {(float)(sx * cz + cx * sy * sz), (float)(cx * sy * cz - sx * sz), 0.0f}};
  const float h_ang[15][3] = {{(float)(-cx * sz - sx * sy * cz), (float)(-cx * cz + sx * sy * sz), (float)(sx * cy)}, {(float)(-sx * sz + cx * sy * cz), (float)(-cx * sy * sz - sx * cz), (float)(-cx * cy)},
                              {(float)(cx * cy * cz), (float)(-cx * cy * sz), (float)(cx * sy)}, {(float)(sx * cy * cz), (float)(-sx * cy * sz), (float)(sx * sy)},
                              {(float)(-sx * cz - cx * sy * sz), (float)(sx * sz - cx * sy * cz), 0.0f}, {(float)(cx * cz - sx * sy * sz), (float)(-sx * sy * cz - cx * sz), 0.0f},
                              {(float)(-cy * cz), (float)(cy * sz), (float)sy}, {(float)(-sx * sy * cz), (float)(sx * sy * sz), (float)(sx * cy)}, {(float)(cx * sy * cz), (float)(-cx * sy * sz), (float)(-cx * cy)},
                              {(float)(sy * sz), (float)(sy * cz), 0.0f}, {(float)(-sx * cy * sz), (float)(-sx * cy * cz), 0.0f}, {(float)(cx * cy * sz), (float)(cx * cy * cz), 0.0f},
                              {(float)(-cy * cz), (float)(cy * sz), 0.0f}, {(float)(-cx * sz - sx * sy * cz), (float)(-cx * cz + sx * sy * sz), 0.0f}, {(float)(-sx * sz + cx * sy * cz), (float)(-cx * sy * sz - sx * cz), 0.0f}};
  double score = 0.0, G[6] = {0, 0, 0, 0, 0, 0}, H[36];
  for (int e = 0; e < 36; ++e) H[e] = 0.0;
  const float gd2 = (float)gauss_d2;
  for (int idx = 0; idx < n; ++idx) {
    const float* xi = input_xyzi + 4 * idx; const float* xt = trans_xyzi + 4 * idx;
    // point gradient (3 x 6) and point hessian blocks, float
    float pg[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
    float xj[8];
    for (int r = 0; r < 8; ++r) xj[r] = (j_ang[r][0] * xi[0] + j_ang[r][1] * xi[1]) + j_ang[r][2] * xi[2];
    pg[1][3] = xj[0]; pg[2][3] = xj[1]; pg[0][4] = xj[2]; pg[1][4] = xj[3]; pg[2][4] = xj[4]; pg[0][5] = xj[5]; pg[1][5] = xj[6]; pg[2][5] = xj[7];
    float ph[6][3][6];   // ph[i][.][j] = second derivative block (i, j)
    for (int i = 0; i < 6; ++i) for (int k = 0; k < 3; ++k) for (int j = 0; j < 6; ++j) ph[i][k][j] = 0.0f;
    if (compute_hessian) {
      float xh[15];
      for (int r = 0; r < 15; ++r) xh[r] = (h_ang[r][0] * xi[0] + h_ang[r][1] * xi[1]) + h_ang[r][2] * xi[2];
      const float a[3] = {0, xh[0], xh[1]}, b[3] = {0, xh[2], xh[3]}, c[3] = {0, xh[4], xh[5]}, d[3] = {xh[6], xh[7], xh[8]}, e[3] = {xh[9], xh[10], xh[11]}, f[3] = {xh[12], xh[13], xh[14]};
      for (int k = 0; k < 3; ++k) { ph[3][k][3] = a[k]; ph[4][k][3] = b[k]; ph[5][k][3] = c[k]; ph[3][k][4] = b[k]; ph[4][k][4] = d[k]; ph[5][k][4] = e[k]; ph[3][k][5] = c[k]; ph[4][k][5] = e[k]; ph[5][k][5] = f[k]; }
    }
    double score_pt = 0.0, g_pt[6] = {0, 0, 0, 0, 0, 0}, h_pt[36];
    for (int e = 0; e < 36; ++e) h_pt[e] = 0.0;
    for (int nb = 0; nb < n_rel; ++nb) {
      const int li = ids7[static_cast<size_t>(n_rel) * idx + nb];
      if (li < 0) continue;
      const double xd[3] = {(double)xt[0] - mean[3 * li], (double)xt[1] - mean[3 * li + 1], (double)xt[2] - mean[3 * li + 2]};   // x_trans -= cell->getMean() in double
      const float x4[3] = {(float)xd[0], (float)xd[1], (float)xd[2]};
      float ci[3][3];
      for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) ci[r][cc] = (float)icov[9 * li + 3 * r + cc];
      float xc[3];   // x^T C^-1
      for (int cc = 0; cc < 3; ++cc) xc[cc] = (x4[0] * ci[0][cc] + x4[1] * ci[1][cc]) + x4[2] * ci[2][cc];
      const float q = (x4[0] * xc[0] + x4[1] * xc[1]) + x4[2] * xc[2];
      float e_x = std::exp(-gd2 * q * 0.5f);
      const float score_inc = (float)(-gauss_d1 * e_x);
      e_x = gd2 * e_x;
      if (e_x > 1 || e_x < 0 || e_x != e_x) continue;
      e_x = (float)(e_x * gauss_d1);
      float cpg[3][6];   // C^-1 * point_gradient
      for (int r = 0; r < 3; ++r) for (int j = 0; j < 6; ++j) cpg[r][j] = (ci[r][0] * pg[0][j] + ci[r][1] * pg[1][j]) + ci[r][2] * pg[2][j];
      float xcpg[6];
      for (int j = 0; j < 6; ++j) xcpg[j] = (x4[0] * cpg[0][j] + x4[1] * cpg[1][j]) + x4[2] * cpg[2][j];
      for (int j = 0; j < 6; ++j) g_pt[j] += (double)(e_x * xcpg[j]);
      if (compute_hessian) {
        float pgcpg[6][6];   // point_gradient^T C^-1 point_gradient
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) pgcpg[i][j] = (pg[0][i] * cpg[0][j] + pg[1][i] * cpg[1][j]) + pg[2][i] * cpg[2][j];
        for (int i = 0; i < 6; ++i) {
          float v[6];
          for (int j = 0; j < 6; ++j) v[j] = (xc[0] * ph[i][0][j] + xc[1] * ph[i][1][j]) + xc[2] * ph[i][2][j];
          for (int j = 0; j < 6; ++j) h_pt[6 * i + j] += (double)(e_x * (-gd2 * xcpg[i] * xcpg[j] + v[j] + pgcpg[j][i]));
        }
      }
      score_pt += (double)score_inc;
    }
    score += score_pt;
    for (int j = 0; j < 6; ++j) G[j] += g_pt[j];
    for (int e = 0; e < 36; ++e) H[e] += h_pt[e];
  }
  *score_out = score;
  for (int j = 0; j < 6; ++j) grad6[j] = G[j];
  for (int e = 0; e < 36; ++e) hess36[e] = H[e];
}
void orc_ndt_derivatives(int n, const float* input_xyzi, const float* trans_xyzi, const int32_t* ids7, const double* mean, const double* icov, const double* p6,
                         double resolution, double outlier_ratio, int compute_hessian, double* score_out, double* grad6, double* hess36) {
  orc_ndt_derivatives_n(n, input_xyzi, trans_xyzi, 7, ids7, mean, icov, p6, resolution, outlier_ratio, compute_hessian, score_out, grad6, hess36);
}

// pcl::VoxelGrid<pcl::PointXYZI>::applyFilter as scanRegistration.cpp:440-444 uses it on one ring's less-flat points (leaf 0.2 m, all fields
// averaged, no minimum point count, no field filter): min / max of the points, voxel index floor(x * inv) - min_b per axis in float, sort by
// voxel index, centroid of every run in float.  pcl sorts with std::sort, whose order of EQUAL keys is implementation-defined; here equal keys
// keep their input order, so a centroid can differ from a pcl build in the last float digit (same points, different summation order).
// Returns the number of output points; out is xyzi per point.
int orc_voxelgrid_xyzi(int n, const float* xyzi, float leaf, float* out) {
  if (n <= 0) return 0;
  const float inv = 1.0f / leaf;
  float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()}, mx[3] = {-mn[0], -mn[1], -mn[2]};
  for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], xyzi[4 * i + k]); mx[k] = std::max(mx[k], xyzi[4 * i + k]); }
  int min_b[3], max_b[3], div_b[3];
  for (int k = 0; k < 3; ++k) { min_b[k] = static_cast<int>(std::floor(mn[k] * inv)); max_b[k] = static_cast<int>(std::floor(mx[k] * inv)); div_b[k] = max_b[k] - min_b[k] + 1; }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  std::vector<std::pair<int, int>> iv(n);
  for (int i = 0; i < n; ++i) {
    int idx = 0;
    for (int k = 0; k < 3; ++k) idx += static_cast<int>(std::floor(xyzi[4 * i + k] * inv) - static_cast<float>(min_b[k])) * mul[k];
    iv[i] = {idx, i};
  }
  std::stable_sort(iv.begin(), iv.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
  int no = 0;
  for (int a = 0; a < n;) {
    int b = a + 1;
    while (b < n && iv[b].first == iv[a].first) ++b;
    float c[4] = {0, 0, 0, 0};
    for (int k = a; k < b; ++k) for (int f = 0; f < 4; ++f) c[f] += xyzi[4 * iv[k].second + f];
    const float cnt = static_cast<float>(b - a);
    for (int f = 0; f < 4; ++f) out[4 * no + f] = c[f] / cnt;
    ++no; a = b;
  }
  return no;
}

// The chronological SurfelPoint emission of getAssociation (/root/reference/src/lvi_exc/src/core/surfel_association.cpp:141-158): w outer, h inner;
// a point needs a flag and a non-zero raw timestamp.  raw: PointXYZIT (32 B).  Returns the number of SurfelPoints written.
int orc_surfel_emit(int H, int W, const int32_t* flag, const float* scan_map, const void* raw_v, double* pt3, double* pt_map3, double* ts, int32_t* plane) {
  struct PT { float x, y, z, pad; float intensity; float pad2; double timestamp; };
  const PT* raw = static_cast<const PT*>(raw_v);
  int n = 0;
  for (int w = 0; w < W; ++w)
    for (int h = 0; h < H; ++h) {
      const size_t i = static_cast<size_t>(h) * W + w;
      if (flag[i] == -1 || 0 == raw[i].timestamp) continue;
      pt3[3 * n] = raw[i].x; pt3[3 * n + 1] = raw[i].y; pt3[3 * n + 2] = raw[i].z;
      pt_map3[3 * n] = scan_map[4 * i]; pt_map3[3 * n + 1] = scan_map[4 * i + 1]; pt_map3[3 * n + 2] = scan_map[4 * i + 2];
      plane[n] = flag[i]; ts[n] = raw[i].timestamp;
      ++n;
    }
  return n;
}

// The same with the reference's OpenMP loop over planes (surfel_association.cpp:122), for CPU timing: conflicts resolved as the serial loop does
// (highest plane id wins) so that the result does not depend on the thread schedule.
void orc_surfel_assoc_omp(int H, int W, const float* scan, int P, const double* p4, const double* bmin, const double* bmax, double radius, int sel, int32_t* flag, int threads) {
  for (int i = 0; i < H * W; ++i) flag[i] = -1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 8)
  for (int pid = 0; pid < P; ++pid) {
    std::vector<int> mask;
    const double* pl = p4 + 4 * pid; const double* lo = bmin + 3 * pid; const double* hi = bmax + 3 * pid;
    for (int h = 0; h < H; ++h) {
      mask.clear();
      for (int w = 0; w < W; ++w) {
        const float* p = scan + 4 * (static_cast<size_t>(h) * W + w);
        if (!std::isnan(p[0]) && p[0] > lo[0] && p[0] < hi[0] && p[1] > lo[1] && p[1] < hi[1] && p[2] > lo[2] && p[2] < hi[2]) {
          const double px = p[0], py = p[1], pz = p[2];
          double dist = px * pl[0] + py * pl[1] + pz * pl[2] + pl[3];
          dist = dist > 0 ? dist : -dist;
          if (dist <= radius) mask.push_back(w);
        }
      }
      if (static_cast<int>(mask.size()) < sel * 2) continue;
      int step = static_cast<int>(mask.size()) / (sel + 1);
      step = std::max(step, 1);
      for (int s = 0; s < sel; ++s) {
        int32_t* f = &flag[h * W + mask[step * (s + 1) - 1]];
        int32_t old = __atomic_load_n(f, __ATOMIC_RELAXED);
        while (old < pid && !__atomic_compare_exchange_n(f, &old, pid, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
      }
    }
  }
}

}  // extern "C"
