// chol16_probe.hip — where do the cycles of the 16 x 16 MFMA Cholesky (lvx_chol16.h) go?  One wavefront, s_memtime around variants of the column loop.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lvi-exc_amd/csrc tools/probes/chol16_probe.hip -o tools/probes/chol16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <string>
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
// `chol16_probe check`: the product function on a family of matrices, machine-readable lines for tests/test_gpu_chol16.py:
//   CHECK <name> <|U^T U - A| / |A|> <|M L - I|> <bad column> <upper part of M>
static void check_mode() {
  double *dA, *dO; long long* dC; int* dB;
  hipMalloc(&dA, 256 * 8); hipMalloc(&dO, 512 * 8); hipMalloc(&dC, 64); hipMalloc(&dB, 4);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
  auto run = [&](const char* name, const std::vector<double>& A) {
    std::vector<double> O(512); int bad = 0;
    hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice); hipMemset(dB, 0, 4);
    hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(64), 0, 0, dA, dO, dC, dB); hipDeviceSynchronize();
    hipMemcpy(O.data(), dO, 512 * 8, hipMemcpyDeviceToHost); hipMemcpy(&bad, dB, 4, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, na = 0, up = 0;
    for (int i = 0; i < 256; ++i) na = fmax(na, fabs(A[i]));
    for (int i = 0; i < 16; ++i) for (int j = i; j < 16; ++j) { double s = 0; for (int k = 0; k <= i; ++k) s += O[k * 16 + i] * O[k * 16 + j]; e1 = fmax(e1, fabs(s - A[i * 16 + j])); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = j; k <= i; ++k) s += O[256 + i * 16 + k] * O[j * 16 + k]; e2 = fmax(e2, fabs(s - (i == j ? 1.0 : 0.0))); }
    for (int i = 0; i < 16; ++i) for (int j = i + 1; j < 16; ++j) up = fmax(up, fabs(O[256 + i * 16 + j]));
    printf("CHECK %s %.3e %.3e %d %.3e\n", name, e1 / na, e2, bad, up);
  };
  auto gram = [&](double diag, double scale) {
    std::vector<double> B(256), A(256);
    for (auto& x : B) x = rnd();
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = (i == j) ? diag : 0.0; for (int k = 0; k < 16; ++k) s += B[i * 16 + k] * B[j * 16 + k]; A[i * 16 + j] = scale * s; }
    return A;
  };
  for (int t = 0; t < 8; ++t) { char nm[32]; snprintf(nm, 32, "random%d", t); run(nm, gram(0.5, 1.0)); }
  run("jacobi_scaled", gram(1.0, 0.25));                       // what the solver feeds it: unit-ish diagonal
  run("large_scale", gram(0.5, 1e8));
  run("small_scale", gram(0.5, 1e-8));
  run("weak", gram(1e-6, 1.0));                                // smallest eigenvalue ~1e-6: the inverse amplifies
  { std::vector<double> I(256, 0.0); for (int i = 0; i < 16; ++i) I[i * 16 + i] = 1.0; run("identity", I); }
  { std::vector<double> A = gram(0.5, 1.0); for (int j = 0; j < 16; ++j) { A[5 * 16 + j] = 0.0; A[j * 16 + 5] = 0.0; } A[5 * 16 + 5] = -1.0; run("negative_pivot_at_6", A); }
}
int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "check") { check_mode(); return 0; }

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
