// micro-benchmark: FP64 global atomic-add throughput on gfx950 for several access patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_atomic(double* buf, size_t n, int mode, int iters, size_t stride_lines) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  for (int it = 0; it < iters; ++it) {
    size_t idx;
    if (mode == 0) idx = (gid + (size_t)it * nthreads) % n;                               // coalesced, every address once
    else if (mode == 1) idx = ((gid * 2654435761ull) + (size_t)it * 40503ull) % n;        // scattered
    else if (mode == 2) idx = ((gid / 64) * stride_lines * 8 + (gid % 8) * 8 + (gid % 64) / 8 + (size_t)it * 64) % n;  // 8 lanes per 64B line interleaved
    else if (mode == 3) idx = (gid % 4096 + (size_t)(it % 16) * 4096) % n;                // hot 512 KB region (heavy reuse of lines)
    else {                                                                                // band cross block: runs of RUN doubles, consecutive runs one band row (196 doubles) apart,
      const size_t RUN = (size_t)(mode - 4 + 1) * 8;                                      // blocks spread over a 240 MB band (modes 4.. : RUN = 8, 16, 24, 32, 48, 64)
      const size_t t = gid + (size_t)it * nthreads, run = t / RUN, off = t % RUN;
      const size_t blk = run / 25, row = run % 25;
      idx = ((blk * 7919ull) % 150000ull) * 196ull + row * 196ull + 100 + off;
      idx %= n;
    }
    atomicAdd(&buf[idx], 1.0);
  }
}
__global__ void k_store(double* buf, size_t n, int iters) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  for (int it = 0; it < iters; ++it) buf[(gid + (size_t)it * nthreads) % n] = 1.0;
}
int main() {
  const size_t n = 64ull << 20;  // 512 MB of doubles
  double* d; hipMalloc(&d, n * 8); hipMemset(d, 0, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 4096, threads = 256, iters = 64;
  const double total = (double)blocks * threads * iters;
  for (int mode = 0; mode < 12; ++mode) {
    k_atomic<<<blocks, threads>>>(d, n, mode, 4, 16); hipDeviceSynchronize();
    hipEventRecord(e0); k_atomic<<<blocks, threads>>>(d, n, mode, iters, 16); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("atomic mode %d: %.3f ms, %.1f G atomics/s, %.1f GB/s payload\n", mode, ms, total / ms / 1e6, total * 8 / ms / 1e6);
  }
  k_store<<<blocks, threads>>>(d, n, 4); hipDeviceSynchronize();
  hipEventRecord(e0); k_store<<<blocks, threads>>>(d, n, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("plain store: %.3f ms, %.1f G stores/s, %.1f GB/s\n", ms, total / ms / 1e6, total * 8 / ms / 1e6);
  return 0;
}
