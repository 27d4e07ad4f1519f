// chol16_probe.hip — where do the cycles of the 16 x 16 MFMA Cholesky (lvx_chol16.h) go?  One wavefront, s_memtime around variants of the column loop.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lvi-exc_amd/csrc tools/probes/chol16_probe.hip -o tools/probes/chol16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "lvx_chol16.h"
using namespace lvx;
// VAR 1: the unpacked form (separate identity tile F, two MFMAs per column, no branch); 2: the same with a per-column bad-pivot branch (what the first version did)
template <int VAR>
__device__ __forceinline__ void chol16_var(d4c& T, d4c& Mres, int fk, int fi, int lane, int* bad) {
  d4c F, U;
#pragma unroll
  for (int v = 0; v < 4; ++v) { F[v] = (fk + 4 * v == fi) ? 1.0 : 0.0; U[v] = 0.0; Mres[v] = 0.0; }
  double d = readlane64(T[0], 0);
#pragma unroll
  for (int cc = 0; cc < 16; ++cc) {
    const int kq = cc & 3, vq = cc >> 2;
    if (VAR == 2) { if (!(d > 0.0)) { if (lane == 0) atomicCAS(bad, 0, cc + 1); d = 1.0; } }
    const double y = rsqrt3_f64(d), nid = -(y * y);
    const bool mine = fk == kq;
    const double x = mine ? T[vq] : 0.0, f = mine ? F[vq] : 0.0;
    if (cc < 15) {
      const double t = readlane64(T[vq], kq * 16 + cc + 1), t1 = readlane64(T[(cc + 1) >> 2], ((cc + 1) & 3) * 16 + cc + 1);
      d = fma(t * nid, t, t1);
    }
    U[vq] = mine ? x * y : U[vq];
    Mres[vq] = mine ? f * y : Mres[vq];
    const double a = x * nid;
    if (cc < 15) {
      T = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x, T, 0, 0, 0);
      F = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f, F, 0, 0, 0);
    }
  }
  T = U;
}
template <int VAR> __global__ void k_probe(const double* A, double* out, long long* cyc, int* bad) {
  const int lane = threadIdx.x, fk = lane >> 4, fi = lane & 15;
  d4c T, M;
  for (int v = 0; v < 4; ++v) T[v] = A[(fk + 4 * v) * 16 + fi];
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (VAR == 0) { const int bc = chol16_mfma(T, M, fk, fi); if (bc && lane == 0) bad[0] = bc; } else chol16_var<VAR>(T, M, fk, fi, lane, bad);
  for (int v = 0; v < 4; ++v) { out[(fk + 4 * v) * 16 + fi] = T[v]; out[256 + (fk + 4 * v) * 16 + fi] = M[v]; }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
  std::vector<double> A(256), B(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) B[i * 16 + j] = sin(1.0 + i * 3.1 + j * 1.7);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = (i == j) ? 4.0 : 0.0; for (int k = 0; k < 16; ++k) s += B[i * 16 + k] * B[j * 16 + k]; A[i * 16 + j] = s; }
  double *dA, *dO; long long* dC; int* dB;
  hipMalloc(&dA, 256 * 8); hipMalloc(&dO, 512 * 8); hipMalloc(&dC, 64); hipMalloc(&dB, 4);
  hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice); hipMemset(dB, 0, 4);
  auto run = [&](auto kern, const char* name) {
    long long c = 0; std::vector<double> O(512);
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, dA, dO, dC, dB); hipDeviceSynchronize(); }
    hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost); hipMemcpy(O.data(), dO, 512 * 8, hipMemcpyDeviceToHost);
    // residual of U^T U - A (upper triangle rows) and of M L - I
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 16; ++i) for (int j = i; j < 16; ++j) { double s = 0; for (int k = 0; k <= i; ++k) s += O[k * 16 + i] * O[k * 16 + j]; e1 = fmax(e1, fabs(s - A[i * 16 + j])); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = j; k <= i; ++k) s += O[256 + i * 16 + k] * O[j * 16 + k]; e2 = fmax(e2, fabs(s - (i == j ? 1.0 : 0.0))); }
    printf("%-28s %6lld cycles (%5.1f per column)   |U^T U - A| %.2e   |M L - I| %.2e\n", name, c, c / 16.0, e1, e2);
  };
  run(k_probe<0>, "product (packed, 1 MFMA)");
  run(k_probe<1>, "unpacked, 2 MFMAs");
  run(k_probe<2>, "unpacked + per-column branch");
  return 0;
}
