// mfma64_probe.hip — latency / throughput of v_mfma_f64_16x16x4_f64 on gfx950, alone and with 1 .. 8 wavefronts per workgroup (is the FP64 matrix pipe per SIMD?),
// and the cost of the register-resident tile update of k_potrf_reg (8 ds_read_b64 + 4 MFMA per tile).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma64_probe.hip -o tools/probes/mfma64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ long long now() { long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define PIN(x) asm volatile("" : "+v"(x))
template <int MODE>
__global__ void k(double* out, long long* cyc, int nact) {
  __shared__ double P[16 * 193];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, fk = lane >> 4, fi = lane & 15;
  for (int e = threadIdx.x; e < 16 * 193; e += blockDim.x) P[e] = 1e-3 * (e % 97);
  __syncthreads();
  double a = 1.0 + lane * 1e-3, bb = 0.5 - lane * 1e-3;
  d4 C0 = {0, 0, 0, 0}, C1 = C0, C2 = C0, C3 = C0;
  long long t0 = 0, t1 = 0;
  if (wv < nact) {
    PIN(a); PIN(bb);
    t0 = now();
    if (MODE == 0) {          // 64 dependent MFMAs (same accumulator)
#pragma unroll
      for (int i = 0; i < 64; ++i) C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, C0, 0, 0, 0);
    } else if (MODE == 1) {   // 64 MFMAs, 4 independent accumulators
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, C0, 0, 0, 0); C1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, C1, 0, 0, 0);
        C2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, C2, 0, 0, 0); C3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, C3, 0, 0, 0);
      }
    } else if (MODE == 2) {   // 16 tile updates as in k_potrf_reg: 8 LDS reads + 4 MFMAs each, 4 accumulators in turn
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        d4& C = (t & 3) == 0 ? C0 : ((t & 3) == 1 ? C1 : ((t & 3) == 2 ? C2 : C3));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) C = __builtin_amdgcn_mfma_f64_16x16x4f64(-P[(4 * ks + fk) * 193 + 16 * (t % 11) + fi], P[(4 * ks + fk) * 193 + 16 * ((t + 3) % 12) + fi], C, 0, 0, 0);
        asm volatile("" ::: "memory");
      }
    } else if (MODE == 3) {   // 64 MFMAs whose A operand comes from the previous result through one multiplication (the factorisation's hand-over)
#pragma unroll
      for (int i = 0; i < 64; ++i) { const double x = C0[i & 3] * 0.25 + a; C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, bb, C0, 0, 0, 0); }
    } else if (MODE == 4) {   // 256 dependent FMAs
#pragma unroll
      for (int i = 0; i < 256; ++i) a = fma(a, 0.999, bb);
      C0[0] = a;
    } else if (MODE == 5) {   // 64 dependent rsq
#pragma unroll
      for (int i = 0; i < 64; ++i) a = __builtin_amdgcn_rsq(a) + 1.0;
      C0[0] = a;
    }
    double z = C0[0] + C0[1] + C0[2] + C0[3] + C1[0] + C2[1] + C3[2];
    PIN(z);
    t1 = now();
    out[threadIdx.x] = z;
    if (lane == 0) cyc[wv] = t1 - t0;
  }
}
int main() {
  double* dO; long long* dC; hipMalloc(&dO, 512 * 8); hipMalloc(&dC, 64);
  auto run = [&](auto kern, const char* name, int per, int nops) {
    printf("%-52s", name);
    for (int nact : {1, 2, 4, 5, 8}) {
      long long c[8] = {0};
      for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(kern, dim3(1), dim3(512), 0, 0, dO, dC, nact); hipDeviceSynchronize(); }
      hipMemcpy(c, dC, 64, hipMemcpyDeviceToHost);
      long long mx = 0; for (int i = 0; i < nact; ++i) mx = c[i] > mx ? c[i] : mx;
      printf("  %dw: %6.1f", nact, (double)mx / nops);
    }
    printf("   cycles per %s\n", per == 0 ? "MFMA" : (per == 1 ? "tile" : "op"));
  };
  run(k<0>, "dependent MFMA chain (same accumulator)", 0, 64);
  run(k<1>, "4 independent accumulators", 0, 64);
  run(k<2>, "tile update (8 ds_read_b64 + 4 MFMA)", 1, 16);
  run(k<3>, "MFMA -> VALU -> MFMA chain", 0, 64);
  run(k<4>, "dependent fma_f64", 2, 256);
  run(k<5>, "dependent rsq_f64 + add", 2, 64);
  return 0;
}
