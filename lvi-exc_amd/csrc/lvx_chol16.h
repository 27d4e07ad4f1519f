// lvx_chol16.h — Cholesky factor + inverse of a 16 x 16 SPD tile held in the MFMA accumulator layout, on the matrix cores (k_potrf_reg, tools/probes/chol16_probe.hip)
#pragma once
#include <hip/hip_runtime.h>
namespace lvx {
typedef double d4c __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane64(double v, int l) {   // v_readlane: the value of lane l as a wave-uniform scalar (l must be uniform)
  const long long u = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(u >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// 1 / sqrt(x): hardware estimate y0 and ONE third-order step, y = y0 (1 + r / 2 + 3 r^2 / 8), r = 1 - x y0^2 (error 5 r^3 / 16: below 2^-53 for any estimate better
// than 2^-19); four dependent operations instead of the eight of two Newton steps — this sits on the factorisation's column-to-column chain
__device__ __forceinline__ double rsqrt3_f64(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double t = x * y0, r = fma(-t, y0, 1.0);
  const double p = fma(r, 0.375, 0.5), yr = y0 * r;
  return fma(yr, p, y0);
}
// T: full symmetric 16 x 16 tile in the accumulator layout (row = (lane >> 4) + 4 reg, col = lane & 15) -> U (row cc valid from column cc on);
// Mres <- inv(L) = inv(U^T), zero above the diagonal.  Returns the 1-based first column with a non-positive pivot (0: none; such a pivot is replaced by 1).
//
// Outer-product Cholesky, ONE v_mfma_f64_16x16x4_f64 per column.  Row cc of the tile, masked to the 16 lanes that hold it, is at once the A operand (column cc,
// by symmetry) and the B operand (row cc) of the rank-1 update Z -= z_cc z_cc^T / d.  The inverse rides in the SAME tile: eliminating [[A, I], [I, 0]] puts
// inv(L)^T scaled by columns into the identity block, F[i][cc] = -T[i][cc] / d at step cc and the same row operations afterwards — and the slot (i, cc), i > cc,
// of the symmetric tile is free from step cc on (the update only needs row cc, whose mirror is the column).  With b[cc] = 2 d the update turns that slot into
// G[i][cc] = -T[i][cc] = d F[i][cc] (no cancellation), later steps apply a_i G[cc'][cc] to it like to any other column, and at the end
//   U[i][j] = Z[i][j] y_i (j >= i),   inv(L)[i][j] = Z[i][j] y_i / d_j (j < i),   inv(L)[i][i] = y_i,     y_i = 1 / sqrt(d_i).
// The pivot of column cc + 1 is formed from row cc BEFORE the update of column cc is issued (d' = Z[cc+1][cc+1] - Z[cc][cc+1]^2 / d), so its rsqrt chain
// runs in the shadow of that MFMA.  No branch inside the loop: a per-column branch (bad-pivot report) made every column a basic block of its own with the
// full MFMA hazard wait — 683 cycles per column against 127 (tools/probes/chol16_probe.hip).
__device__ __forceinline__ int chol16_mfma(d4c& T, d4c& Mres, int fk, int fi) {
  d4c Z = T, yrow = d4c{0.0, 0.0, 0.0, 0.0};
  double invdc = 0.0;          // 1 / d of this lane's column
  int badcol = 0;
  double d = readlane64(Z[0], 0);
#pragma unroll
  for (int cc = 0; cc < 16; ++cc) {
    const int kq = cc & 3, vq = cc >> 2;
    const bool pos = d > 0.0;
    badcol = (!pos && badcol == 0) ? cc + 1 : badcol;
    d = pos ? d : 1.0;
    const double y = rsqrt3_f64(d), nid = -(y * y), d2 = 2.0 * d;
    const bool mine = fk == kq;
    const double z = mine ? Z[vq] : 0.0;                      // lane (kq, j): Z[cc][j]
    if (cc < 15) {
      const double t = readlane64(Z[vq], kq * 16 + cc + 1), t1 = readlane64(Z[(cc + 1) >> 2], ((cc + 1) & 3) * 16 + cc + 1);
      d = fma(t * nid, t, t1);
    }
    yrow[vq] = mine ? y : yrow[vq];
    invdc = fi == cc ? -nid : invdc;
    if (cc < 15) {
      const double a = fi > cc ? z * nid : 0.0;               // A[i][kq] = -T[cc][i] / d, rows below cc only
      const double bb = (mine && fi == cc) ? d2 : z;         // B[kq][j] = Z[cc][j], 2 d at j = cc
      Z = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, Z, 0, 0, 0);
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = fk + 4 * v;
    const double zy = Z[v] * yrow[v];
    T[v] = fi >= row ? zy : 0.0;
    Mres[v] = fi < row ? zy * invdc : (fi == row ? yrow[v] : 0.0);
  }
  return badcol;
}
}  // namespace lvx
