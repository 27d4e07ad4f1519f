// oracle/orc_sparse.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see orc_core.hpp header; PARITY UNPINNED).
//
// Sparse side of the CPU oracle, for problems whose dense J^T J does not fit (config 4: 155 k tangent scalars):
//   orc_jacobian_csr   the robustified Jacobian of every residual block as ONE generic CSR matrix in tangent coordinates — exactly the rows
//                      ceres::Problem::Evaluate hands to the linear solver (K/kontiki/trajectory_estimator.h:38-68: DynamicAutoDiffCostFunction rows x the
//                      manifold Jacobian, scaled by sqrt(rho') as ceres::Corrector does for a loss with rho'' <= 0) — produced by the same per-block
//                      stride-4 dual-number evaluation as orc_evaluate;
//   orc_ata_lower      the lower triangle of A^T A of any CSR matrix as CSC (Gustavson's row-merge on the transpose), OpenMP over the columns.
// Neither knows anything about knots, bands, borders or landmarks: oracle/lm_sparse.py builds the normal equations, the Schur complement of the
// landmark e-blocks and the fill-reducing ordering from the matrix alone, so the oracle's LM step shares no structure with the GPU solver
// (lvi-exc_amd/csrc/lvx_solver.hip, lvx_bcr.hip).
#include "orc_core.hpp"
#include "orc_problem.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

namespace {
void huber_scale(double a, double s, double& rho, double& sr) {   // ceres::HuberLoss + Corrector with rho'' <= 0 (restated; as lvx_oracle.cpp::huber)
  if (a <= 0.0 || s <= a * a) { rho = s; sr = 1.0; return; }
  const double r = std::sqrt(s);
  rho = 2.0 * a * r - a * a;
  sr = std::sqrt(a / r);
}
}  // namespace

extern "C" {

void orc_free(void* p) { std::free(p); }

// row_ptr[num_residuals + 1]; *cols_out / *vals_out are malloc'ed (orc_free).  residuals: raw weighted residuals (as orc_evaluate); rs: the
// robustified residuals sqrt(rho') r that go with the rows.  Returns 0, -1 (range error), -2 (non-unit quaternion), -3 (other).
int orc_jacobian_csr(const orc_problem* p, const double* state, double* cost, double* residuals, double* rs, int64_t* row_ptr, int32_t** cols_out, double** vals_out) {
#ifdef _OPENMP
  const int nth = p->threads > 0 ? p->threads : omp_get_max_threads();
#else
  const int nth = 1;
#endif
  const int nres_total = p->num_residuals();
  std::vector<int32_t> cnt(static_cast<size_t>(nres_total), 0);
  double total = 0.0;
  int err = 0;
  int row0 = 0;
  // per family and thread: the (cols, vals) of a contiguous range of blocks (static schedule), concatenated afterwards
  struct Part { std::vector<int32_t> c; std::vector<double> v; };
  std::vector<std::vector<Part>> parts(NUM_FAM, std::vector<Part>(nth));
  for (int fam = 0; fam < NUM_FAM; ++fam) {
    const int n = p->family_count(fam);
    const int nr = Problem::family_nres(fam);
    const double a = p->family_huber(fam);
    double fam_cost = 0.0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nth) reduction(+ : fam_cost)
#endif
    {
#ifdef _OPENMP
      const int th = omp_get_thread_num();
#else
      const int th = 0;
#endif
      const int lo = static_cast<int>(static_cast<long long>(n) * th / nth), hi = static_cast<int>(static_cast<long long>(n) * (th + 1) / nth);
      Part& part = parts[fam][th];
      for (int i = lo; i < hi && !err; ++i) {
        try {
          double r[4];
          RowSet rows;
          p->eval_one(fam, i, state, r, &rows);
          double s = 0; for (int k = 0; k < nr; ++k) s += r[k] * r[k];
          double rho, sr; huber_scale(a, s, rho, sr);
          fam_cost += 0.5 * rho;
          const int row = row0 + i * nr;
          const int nc = static_cast<int>(rows.cols.size());
          for (int k = 0; k < nr; ++k) {
            if (residuals) residuals[row + k] = r[k];
            if (rs) rs[row + k] = sr * r[k];
            cnt[row + k] = nc;
            for (int c = 0; c < nc; ++c) { part.c.push_back(rows.cols[c]); part.v.push_back(sr * rows.vals[k][c]); }
          }
        } catch (const orc::range_error&) {
          err = -1;
        } catch (const orc::nonunit_quat_error&) {
          err = -2;
        } catch (const std::exception&) {
          err = -3;
        }
      }
    }
    total += fam_cost;
    row0 += n * nr;
  }
  if (cost) *cost = total;
  if (err) { *cols_out = nullptr; *vals_out = nullptr; return err; }
  row_ptr[0] = 0;
  for (int r = 0; r < nres_total; ++r) row_ptr[r + 1] = row_ptr[r] + cnt[r];
  const int64_t nnz = row_ptr[nres_total];
  int32_t* C = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * std::max<int64_t>(nnz, 1)));
  double* V = static_cast<double*>(std::malloc(sizeof(double) * std::max<int64_t>(nnz, 1)));
  if (!C || !V) { std::free(C); std::free(V); return -3; }
  int64_t at = 0;
  for (int fam = 0; fam < NUM_FAM; ++fam)
    for (int th = 0; th < nth; ++th) {
      const Part& part = parts[fam][th];
      if (part.c.empty()) continue;
      std::memcpy(C + at, part.c.data(), sizeof(int32_t) * part.c.size());
      std::memcpy(V + at, part.v.data(), sizeof(double) * part.v.size());
      at += static_cast<int64_t>(part.c.size());
    }
  *cols_out = C; *vals_out = V;
  return at == nnz ? 0 : -3;
}

// Lower triangle (row >= column) of A^T A for a CSR matrix A (n_rows x n_cols, duplicate columns inside a row allowed: they add), as CSC with sorted
// row indices: col_ptr[n_cols + 1]; *rows_out / *vals_out malloc'ed.  Column j of the result = sum over the rows r that hold j of A[r][j] * (row r restricted
// to columns >= j): the transpose is formed first (counting sort), then one dense accumulator + touched list per thread.  Returns 0 or -3.
int orc_ata_lower(int64_t n_rows, int32_t n_cols, const int64_t* row_ptr, const int32_t* cols, const double* vals, int threads, int64_t* col_ptr, int32_t** rows_out, double** vals_out) {
#ifdef _OPENMP
  const int nth = threads > 0 ? threads : omp_get_max_threads();
#else
  const int nth = 1;
#endif
  const int64_t nnz = row_ptr[n_rows];
  // transpose: for every column the (row, value) pairs
  std::vector<int64_t> tp(static_cast<size_t>(n_cols) + 1, 0);
  for (int64_t e = 0; e < nnz; ++e) tp[cols[e] + 1]++;
  for (int32_t j = 0; j < n_cols; ++j) tp[j + 1] += tp[j];
  std::vector<int64_t> trow(static_cast<size_t>(std::max<int64_t>(nnz, 1)));
  std::vector<double> tval(static_cast<size_t>(std::max<int64_t>(nnz, 1)));
  {
    std::vector<int64_t> fill(tp.begin(), tp.end() - 1);
    for (int64_t r = 0; r < n_rows; ++r)
      for (int64_t e = row_ptr[r]; e < row_ptr[r + 1]; ++e) { const int64_t q = fill[cols[e]]++; trow[q] = r; tval[q] = vals[e]; }
  }
  std::vector<std::vector<int32_t>> out_r(static_cast<size_t>(n_cols));
  std::vector<std::vector<double>> out_v(static_cast<size_t>(n_cols));
#ifdef _OPENMP
#pragma omp parallel num_threads(nth)
#endif
  {
    std::vector<double> acc(static_cast<size_t>(n_cols), 0.0);
    std::vector<uint8_t> mark(static_cast<size_t>(n_cols), 0);
    std::vector<int32_t> touched;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
    for (int32_t j = 0; j < n_cols; ++j) {
      touched.clear();
      for (int64_t q = tp[j]; q < tp[j + 1]; ++q) {
        const int64_t r = trow[q];
        const double v = tval[q];
        for (int64_t e = row_ptr[r]; e < row_ptr[r + 1]; ++e) {
          const int32_t i = cols[e];
          if (i < j) continue;
          if (!mark[i]) { mark[i] = 1; touched.push_back(i); }
          acc[i] += v * vals[e];
        }
      }
      std::sort(touched.begin(), touched.end());
      out_r[j].assign(touched.begin(), touched.end());
      out_v[j].resize(touched.size());
      for (size_t k = 0; k < touched.size(); ++k) { out_v[j][k] = acc[touched[k]]; acc[touched[k]] = 0.0; mark[touched[k]] = 0; }
    }
  }
  col_ptr[0] = 0;
  for (int32_t j = 0; j < n_cols; ++j) col_ptr[j + 1] = col_ptr[j] + static_cast<int64_t>(out_r[j].size());
  const int64_t onz = col_ptr[n_cols];
  int32_t* R = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * std::max<int64_t>(onz, 1)));
  double* V = static_cast<double*>(std::malloc(sizeof(double) * std::max<int64_t>(onz, 1)));
  if (!R || !V) { std::free(R); std::free(V); return -3; }
  for (int32_t j = 0; j < n_cols; ++j) {
    if (out_r[j].empty()) continue;
    std::memcpy(R + col_ptr[j], out_r[j].data(), sizeof(int32_t) * out_r[j].size());
    std::memcpy(V + col_ptr[j], out_v[j].data(), sizeof(double) * out_v[j].size());
  }
  *rows_out = R; *vals_out = V;
  return 0;
}

}  // extern "C"
