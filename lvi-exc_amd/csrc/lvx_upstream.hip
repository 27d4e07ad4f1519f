// lvx_upstream.hip — upstream point-cloud kernels of the LVI-ExC pipeline (gfx950):
//   lvx_scan_register   A-LOAM scanRegistration core: range/NaN filter, ring bucketing, 11-tap float curvature, per-ring
//                       6-sector sort + greedy edge/plane pick     (reference: src/aloam/src/scanRegistration.cpp:101-131,199-447)
//   lvx_voxel_build     ndt_omp VoxelGridCovariance::applyFilter    (src/ndt_omp/include/pclomp/voxel_grid_covariance_omp_impl.hpp:49-374)
//   lvx_voxel_lookup7   getNeighborhoodAtPoint7                     (same file :378-438)
//   lvx_surfel_assoc    SurfelAssociation::getAssociation           (src/lvi_exc/src/core/surfel_association.cpp:111-138,296-331)
// These are HBM/latency-bound integer + float kernels (no MFMA).  Float expressions keep the reference's evaluation order and the
// library is built with -ffp-contract=off, so scan registration is bit-exact against the serial code.
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "lvx_ctx.h"
#include "lvx_stdsort.h"

namespace lvx {

struct RsPoint { float x, y, z, pad; uint8_t intensity; uint8_t pad2; uint16_t ring; uint32_t pad3; double timestamp; };
static_assert(sizeof(RsPoint) == 32, "RsPointXYZIRT layout (scanRegistration.cpp:57-66)");

// ------------------------------------------------------------------------------------------------------------------------
// scan registration
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sr_keep(const RsPoint& p, float thr) {   // removeClosedPointCloud (:101-131)
  if (p.x * p.x + p.y * p.y + p.z * p.z < thr * thr) return false;
  if (isnan(p.x) || isnan(p.y) || isnan(p.z)) return false;
  return true;
}
// A batch of sweeps on the device (lvx_scan_register_batch; one sweep = a batch of one): the point arrays of all sweeps are concatenated — sweep s owns
// [off[s], off[s + 1]) of every per-point array and uses SWEEP-LOCAL indices inside it, as the reference's globals do — the per-ring arrays are [S][n_rings][..].
// Sweeps are independent: every kernel takes the sweep from blockIdx.y, so 64 sweeps are one launch of 64 x 16 ring workgroups instead of 64 launches of 16.
struct SrBatch {
  const RsPoint* pts; const int* off; int n_rings; float thr; long long N;
  float4* cloud; float* curv; int* label; int* sort_ind; int* picked; int* src; int* lists; int* lflat_r;   // [N] each, lists [4][N]
  int* rc; int* ss; int* se; int* cnt; int* sharp_r; int* lsharp_r; int* flat_r; int* counts; int* err;     // [S][R], cnt [S][R][4], sharp_r [S][R][16], lsharp_r [S][R][128], flat_r [S][R][32], counts [S][4], err [S]
};
__device__ __forceinline__ int sr_kept(const SrBatch& B, int s) { int m = 0; for (int r = 0; r < B.n_rings; ++r) m += B.rc[s * B.n_rings + r]; return m; }
__global__ void k_sr_count(SrBatch B) {   // per-workgroup histogram in LDS: 28.8 k atomics on 16 addresses took 68 us
  const int s = blockIdx.y, n = B.off[s + 1] - B.off[s], n_rings = B.n_rings;
  const RsPoint* pts = B.pts + B.off[s]; const float thr = B.thr; int* ring_count = B.rc + s * n_rings;
  __shared__ int h[128];
  for (int t = threadIdx.x; t < 128; t += blockDim.x) h[t] = 0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const RsPoint p = pts[i]; if (sr_keep(p, thr) && p.ring < n_rings) { if (p.ring < 128) atomicAdd(&h[p.ring], 1); else atomicAdd(&ring_count[p.ring], 1); } }
  __syncthreads();
  for (int t = threadIdx.x; t < 128 && t < n_rings; t += blockDim.x) if (h[t]) atomicAdd(&ring_count[t], h[t]);
}
// one workgroup per ring: order-preserving compaction of the ring's points to src[ring_start + k]
__global__ __launch_bounds__(1024) void k_sr_bucket(SrBatch B) {
  const int s = blockIdx.y, n = B.off[s + 1] - B.off[s], n_rings = B.n_rings;
  const RsPoint* pts = B.pts + B.off[s]; const float thr = B.thr; const int* ring_count = B.rc + s * n_rings;
  int* src = B.src + B.off[s]; int* scan_start = B.ss + s * n_rings; int* scan_end = B.se + s * n_rings;
  (void)n_rings;
  const int r = blockIdx.x;
  __shared__ int wsum[16];
  __shared__ int base_s;
  int start = 0;
  for (int k = 0; k < r; ++k) start += ring_count[k];
  if (threadIdx.x == 0) { base_s = 0; scan_start[r] = start + 5; scan_end[r] = start + ring_count[r] - 6; }   // :284-290
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c0 = 0; c0 < n; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    bool f = false;
    if (i < n) { const RsPoint p = pts[i]; f = sr_keep(p, thr) && p.ring == r; }
    const unsigned long long m = __ballot(f);
    const int within = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int off = 0;
    for (int k = 0; k < wv; ++k) off += wsum[k];
    int tot = 0;
    for (int k = 0; k < 16; ++k) tot += wsum[k];
    const int base = base_s;
    if (f) src[start + base + off + within] = i;
    __syncthreads();
    if (threadIdx.x == 0) base_s = base + tot;
    __syncthreads();
  }
}
__global__ void k_sr_gather(SrBatch B) {
  const int s = blockIdx.y, o = B.off[s], m = sr_kept(B, s);
  const RsPoint* pts = B.pts + o; const int* src = B.src + o; float4* cloud = B.cloud + o; float* curv = B.curv + o; int* label = B.label + o; int* sort_ind = B.sort_ind + o; int* picked = B.picked + o;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const RsPoint p = pts[src[j]];
  const double rel = p.timestamp - pts[0].timestamp;                       // :161, :277
  cloud[j] = make_float4(p.x, p.y, p.z, (float)((double)p.ring + rel));     // intensity = ring + relTime (:278)
  curv[j] = 0.f; label[j] = 0; sort_ind[j] = j; picked[j] = 0;
}
// 11-tap curvature, float, reference evaluation order (:295-305); LDS tile with a halo of 5
__global__ __launch_bounds__(256) void k_sr_curv(SrBatch B) {
  const int s = blockIdx.y, o = B.off[s], m = sr_kept(B, s);
  const float4* cloud = B.cloud + o; float* curv = B.curv + o;
  __shared__ float sx[256 + 10], sy[256 + 10], sz[256 + 10];
  const int i0 = blockIdx.x * 256;
  if (i0 >= m) return;
  for (int t = threadIdx.x; t < 266; t += 256) {
    const int j = i0 - 5 + t;
    float4 p = make_float4(0, 0, 0, 0);
    if (j >= 0 && j < m) p = cloud[j];
    sx[t] = p.x; sy[t] = p.y; sz[t] = p.z;
  }
  __syncthreads();
  const int i = i0 + threadIdx.x;
  if (i < 5 || i >= m - 5) return;
  const int c = threadIdx.x + 5;
  const float dX = sx[c - 5] + sx[c - 4] + sx[c - 3] + sx[c - 2] + sx[c - 1] - 10 * sx[c] + sx[c + 1] + sx[c + 2] + sx[c + 3] + sx[c + 4] + sx[c + 5];
  const float dY = sy[c - 5] + sy[c - 4] + sy[c - 3] + sy[c - 2] + sy[c - 1] - 10 * sy[c] + sy[c + 1] + sy[c + 2] + sy[c + 3] + sy[c + 4] + sy[c + 5];
  const float dZ = sz[c - 5] + sz[c - 4] + sz[c - 3] + sz[c - 2] + sz[c - 1] - 10 * sz[c] + sz[c + 1] + sz[c + 2] + sz[c + 3] + sz[c + 4] + sz[c + 5];
  curv[i] = dX * dX + dY * dY + dZ * dZ;
}
#define SR_SEC_MAX 2048
// (curvature bits << 32 | point index), ordered by the curvature as the reference's comp does (scanRegistration.cpp:87): float compare, index ignored
struct SrKeyLess { __device__ __forceinline__ bool operator()(unsigned long long x, unsigned long long y) const { return __uint_as_float((unsigned)(x >> 32)) < __uint_as_float((unsigned)(y >> 32)); } };
__device__ __forceinline__ bool sr_key_tie(unsigned long long a, unsigned long long b) {   // equal curvatures (or a NaN, which no bit order represents): std::sort's order is not the parallel sort's
  const unsigned ha = (unsigned)(a >> 32), hb = (unsigned)(b >> 32);
  return ha == hb || ha > 0x7f800000u || hb > 0x7f800000u;
}
__device__ __forceinline__ float sr_gap2(const float4* c, int a, int b) {
  const float dx = c[a].x - c[b].x, dy = c[a].y - c[b].y, dz = c[a].z - c[b].z;
  return dx * dx + dy * dy + dz * dz;
}
// one workgroup per ring: six sectors in order (marks of one sector influence the next); per sector a bitonic sort of
// (curvature bits, index) in LDS, then the reference's serial greedy pick on one lane (:316-447).  The ring's points, curvatures, picked flags
// and labels live in LDS for the whole kernel: the serial pick is a chain of dependent reads, and from HBM each of them cost ~0.5 us (697 us
// per sweep); the less-flat list is compacted by the whole workgroup.  Rings longer than SR_RING_MAX take the global-memory path.
#define SR_RING_MAX 4096
struct SrRingLds { float x[SR_RING_MAX], y[SR_RING_MAX], z[SR_RING_MAX], c[SR_RING_MAX]; signed char lab[SR_RING_MAX]; unsigned char pk[SR_RING_MAX]; };
template <bool LDSR> struct SrView {
  const float4* cloud; const float* curv; int* label; int* picked; SrRingLds* L; int base;
  __device__ __forceinline__ float gap2(int a, int b) const {
    if (LDSR) { const float dx = L->x[a - base] - L->x[b - base], dy = L->y[a - base] - L->y[b - base], dz = L->z[a - base] - L->z[b - base]; return dx * dx + dy * dy + dz * dz; }
    return sr_gap2(cloud, a, b);
  }
  __device__ __forceinline__ float cv(int i) const { return LDSR ? L->c[i - base] : curv[i]; }
  __device__ __forceinline__ int pk(int i) const { return LDSR ? (int)L->pk[i - base] : picked[i]; }
  __device__ __forceinline__ void set_pk(int i) const { if (LDSR) L->pk[i - base] = 1; else picked[i] = 1; }
  __device__ __forceinline__ int lab(int i) const { return LDSR ? (int)L->lab[i - base] : label[i]; }
  __device__ __forceinline__ void set_lab(int i, int v) const { if (LDSR) L->lab[i - base] = (signed char)v; else label[i] = v; }
};
template <bool LDSR>
__device__ __forceinline__ void sr_classify_ring(const SrView<LDSR>& V, unsigned long long* key0, int kstride, int* cnt, int* wsum, int r, int s0, int e0, int* sort_ind,
                                                 int* sharp_r, int* lsharp_r, int* flat_r, int* lflat_r, int* err) {
  // kstride > 0: the six sectors were sorted beforehand, six wavefronts at once, into key0 + j * kstride (k_sr_classify); 0: sorted here, one after the other
  for (int j = 0; j < 6; ++j) {
    const int sp = s0 + (e0 - s0) * j / 6;
    const int ep = s0 + (e0 - s0) * (j + 1) / 6 - 1;
    const int len = ep - sp + 1;
    unsigned long long* key = key0 + (size_t)j * kstride;
    if (len > (kstride > 0 ? kstride : SR_SEC_MAX)) { if (threadIdx.x == 0) atomicOr(err, 8); break; }
    int np2 = 1; while (np2 < len) np2 <<= 1;
    if (kstride == 0) {
    for (int t = threadIdx.x; t < np2; t += 512)
      key[t] = t < len ? (((unsigned long long)__float_as_uint(V.cv(sp + t))) << 32) | (unsigned)(sp + t) : ~0ull;   // curvature >= 0: bit order == value order
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
      for (int jj = k >> 1; jj > 0; jj >>= 1) {
        for (int t = threadIdx.x; t < np2; t += 512) {
          const int x = t ^ jj;
          if (x > t) {
            const unsigned long long a = key[t], b = key[x];
            const bool up = (t & k) == 0;
            if ((a > b) == up) { key[t] = b; key[x] = a; }
          }
        }
        __syncthreads();
      }
    {   // equal curvatures in the sector: the reference's order is libstdc++ introsort's (lvx_stdsort.h) — one lane restates it from the identity order
      bool tie = false;
      for (int t = threadIdx.x; t + 1 < len; t += 512) tie |= sr_key_tie(key[t], key[t + 1]);
      if (__syncthreads_or(tie)) {
        for (int t = threadIdx.x; t < len; t += 512) key[t] = (((unsigned long long)__float_as_uint(V.cv(sp + t))) << 32) | (unsigned)(sp + t);
        __syncthreads();
        if (threadIdx.x == 0) libstdcxx_sort(key, len, SrKeyLess());
        __syncthreads();
      }
    }
    }
    for (int t = threadIdx.x; t < len; t += 512) sort_ind[sp + t] = (int)(key[t] & 0xffffffffu);
    if constexpr (LDSR) {
      // The reference's greedy pick is serial in its DECISIONS (a pick marks neighbours that later candidates must see) but not in its work: wavefront 0
      // examines 64 sorted candidates per ballot, takes the first unpicked one beyond the threshold, and evaluates the ten neighbour gaps of a pick in
      // parallel (the marks of each direction stop at its first gap > 0.05).  One lane doing all of it was a 200 us chain of dependent LDS reads per ring.
      if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
        auto mark_neighbours = [&](int ind) {   // :374-391 / :407-423: lanes 1..5 -> ind + l, lanes 9..13 -> ind - l
          bool brk = false; int tgt = -1;
          if (lane >= 1 && lane <= 5) { brk = V.gap2(ind + lane, ind + lane - 1) > 0.05; tgt = ind + lane; }
          else if (lane >= 9 && lane <= 13) { const int l = lane - 8; brk = V.gap2(ind - l, ind - l + 1) > 0.05; tgt = ind - l; }
          const unsigned long long mb = __ballot(brk);
          const unsigned plus = (unsigned)(mb >> 1) & 0x1fu, minus = (unsigned)(mb >> 9) & 0x1fu;
          const int fp = plus ? __ffs(plus) : 6, fm = minus ? __ffs(minus) : 6;     // 1-based l of the first break in each direction
          if (lane == 0) V.set_pk(ind);
          if (lane >= 1 && lane <= 5 && lane < fp) V.set_pk(tgt);
          if (lane >= 9 && lane <= 13 && lane - 8 < fm) V.set_pk(tgt);
          wsync();
        };
        int largest = 0;
        for (int k = ep; k >= sp;) {          // descending curvature
          const int kk = k - lane;
          int ind = -1; bool above = false, q = false;
          if (kk >= sp) { ind = (int)(key[kk - sp] & 0xffffffffu); above = V.cv(ind) > 0.1; q = above && V.pk(ind) == 0; }
          const unsigned long long mq = __ballot(q), ma = __ballot(above);
          if (!mq) { if (ma != ~0ull) break; k -= 64; continue; }   // nothing to pick among these 64; below the threshold nothing follows (sorted)
          const int f = __ffsll((long long)mq) - 1;
          const int pick = __shfl(ind, f);
          largest++;
          if (largest > 20) break;
          if (lane == 0) {
            if (largest <= 2) { V.set_lab(pick, 2); sharp_r[r * 16 + cnt[0]++] = pick; lsharp_r[r * 128 + cnt[1]++] = pick; }
            else { V.set_lab(pick, 1); lsharp_r[r * 128 + cnt[1]++] = pick; }
          }
          mark_neighbours(pick);
          k = k - f - 1;
        }
        int smallest = 0;
        for (int k = sp; k <= ep;) {          // ascending curvature
          const int kk = k + lane;
          int ind = -1; bool below = false, q = false;
          if (kk <= ep) { ind = (int)(key[kk - sp] & 0xffffffffu); below = V.cv(ind) < 0.1; q = below && V.pk(ind) == 0; }
          const unsigned long long mq = __ballot(q), ma = __ballot(below);
          if (!mq) { if (ma != ~0ull) break; k += 64; continue; }
          const int f = __ffsll((long long)mq) - 1;
          const int pick = __shfl(ind, f);
          if (lane == 0) { V.set_lab(pick, -1); flat_r[r * 32 + cnt[2]++] = pick; }
          smallest++;
          if (smallest >= 4) break;             // before it is marked (:394-403)
          mark_neighbours(pick);
          k = k + f + 1;
        }
        wsync();
      }
    } else
    if (threadIdx.x == 0) {   // the sorted indices are read from LDS (key[k - sp])
      int largest = 0;
      for (int k = ep; k >= sp; k--) {
        const int ind = (int)(key[k - sp] & 0xffffffffu);
        if (V.pk(ind) == 0 && V.cv(ind) > 0.1) {
          largest++;
          if (largest <= 2) { V.set_lab(ind, 2); sharp_r[r * 16 + cnt[0]++] = ind; lsharp_r[r * 128 + cnt[1]++] = ind; }
          else if (largest <= 20) { V.set_lab(ind, 1); lsharp_r[r * 128 + cnt[1]++] = ind; }
          else break;
          V.set_pk(ind);
          for (int l = 1; l <= 5; l++) { if (V.gap2(ind + l, ind + l - 1) > 0.05) break; V.set_pk(ind + l); }
          for (int l = -1; l >= -5; l--) { if (V.gap2(ind + l, ind + l + 1) > 0.05) break; V.set_pk(ind + l); }
        }
      }
      int smallest = 0;
      for (int k = sp; k <= ep; k++) {
        const int ind = (int)(key[k - sp] & 0xffffffffu);
        if (V.pk(ind) == 0 && V.cv(ind) < 0.1) {
          V.set_lab(ind, -1); flat_r[r * 32 + cnt[2]++] = ind;
          smallest++;
          if (smallest >= 4) break;
          V.set_pk(ind);
          for (int l = 1; l <= 5; l++) { if (V.gap2(ind + l, ind + l - 1) > 0.05) break; V.set_pk(ind + l); }
          for (int l = -1; l >= -5; l--) { if (V.gap2(ind + l, ind + l + 1) > 0.05) break; V.set_pk(ind + l); }
        }
      }
    }
    __syncthreads();
    // less-flat list of the sector: every point with label <= 0, in index order (:425-431) — ordered compaction by the workgroup
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c0 = sp; c0 <= ep; c0 += 512) {
      const int k = c0 + threadIdx.x;
      const bool f = k <= ep && V.lab(k) <= 0;
      const unsigned long long m = __ballot(f);
      if (lane == 0) wsum[wv] = __popcll(m);
      __syncthreads();
      int off = 0, tot = 0;
      for (int q = 0; q < 8; ++q) { if (q < wv) off += wsum[q]; tot += wsum[q]; }
      if (f) lflat_r[(s0 - 5) + cnt[3] + off + __popcll(m & ((1ull << lane) - 1ull))] = k;
      __syncthreads();
      if (threadIdx.x == 0) cnt[3] += tot;
      __syncthreads();
    }
  }
}
__global__ __launch_bounds__(512) void k_sr_classify(SrBatch B) {
  const int sw = blockIdx.y, o = B.off[sw], R = B.n_rings;
  const float4* cloud = B.cloud + o; const float* curv = B.curv + o; const int* scan_start = B.ss + sw * R; const int* scan_end = B.se + sw * R;
  int* label = B.label + o; int* sort_ind = B.sort_ind + o; int* picked = B.picked + o;
  int* sharp_r = B.sharp_r + (size_t)sw * R * 16; int* lsharp_r = B.lsharp_r + (size_t)sw * R * 128; int* flat_r = B.flat_r + (size_t)sw * R * 32; int* lflat_r = B.lflat_r + o;
  int* cnt_r = B.cnt + (size_t)sw * R * 4; int* err = B.err + sw;
#define SR_SEC_LDS 1024   // sector capacity of the LDS path: a ring of <= 4096 points has sectors of <= 683
  __shared__ unsigned long long key[6 * SR_SEC_LDS];   // (>= SR_SEC_MAX: the global-memory path sorts one sector at a time in its head)
  __shared__ int cnt[4], wsum[8];
  __shared__ SrRingLds ring;
  const int r = blockIdx.x;
  const int s0 = scan_start[r], e0 = scan_end[r];
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
  __syncthreads();
  if (e0 - s0 >= 6) {
    const int base = s0 - 5, np = e0 + 6 - base;   // the ring's points [base, base + np)
    if (np <= SR_RING_MAX) {
      for (int t = threadIdx.x; t < np; t += 512) { const float4 p = cloud[base + t]; ring.x[t] = p.x; ring.y[t] = p.y; ring.z[t] = p.z; ring.c[t] = curv[base + t]; ring.lab[t] = (signed char)label[base + t]; ring.pk[t] = (unsigned char)(picked[base + t] != 0); }
      __syncthreads();
      const SrView<true> V{cloud, curv, label, picked, &ring, base};
      {   // the six sector sorts at once: wavefront w sorts sector w on its own (bitonic, wavefront barriers only) — curvatures do not change during the picks
        const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (w < 6) {
          const int sp = s0 + (e0 - s0) * w / 6, ep = s0 + (e0 - s0) * (w + 1) / 6 - 1, len = ep - sp + 1;
          unsigned long long* kw = key + (size_t)w * SR_SEC_LDS;
          int np2 = 1; while (np2 < len) np2 <<= 1;
          if (len <= SR_SEC_LDS) {
            for (int t = lane; t < np2; t += 64) kw[t] = t < len ? (((unsigned long long)__float_as_uint(ring.c[sp + t - base])) << 32) | (unsigned)(sp + t) : ~0ull;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int k = 2; k <= np2; k <<= 1)
              for (int jj = k >> 1; jj > 0; jj >>= 1) {
                for (int t = lane; t < np2; t += 64) {
                  const int x = t ^ jj;
                  if (x > t) {
                    const unsigned long long a = kw[t], b = kw[x];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) { kw[t] = b; kw[x] = a; }
                  }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
              }
            // equal curvatures in the sector (quantised ranges, lattice coordinates): std::sort is unstable and the pick order of the tied points is whatever
            // libstdc++'s introsort leaves — lane 0 restates it from the identity order (lvx_stdsort.h); tie-free sectors keep the parallel result, which is unique
            bool tie = false;
            for (int t = lane; t + 1 < len; t += 64) tie |= sr_key_tie(kw[t], kw[t + 1]);
            if (__ballot(tie)) {
              for (int t = lane; t < len; t += 64) kw[t] = (((unsigned long long)__float_as_uint(ring.c[sp + t - base])) << 32) | (unsigned)(sp + t);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
              if (lane == 0) libstdcxx_sort(kw, len, SrKeyLess());
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
          }
        }
        __syncthreads();
      }
      sr_classify_ring<true>(V, key, SR_SEC_LDS, cnt, wsum, r, s0, e0, sort_ind, sharp_r, lsharp_r, flat_r, lflat_r, err);
      __syncthreads();
      for (int t = threadIdx.x; t < np; t += 512) { label[base + t] = ring.lab[t]; picked[base + t] = ring.pk[t]; }
    } else {
      const SrView<false> V{cloud, curv, label, picked, nullptr, base};
      sr_classify_ring<false>(V, key, 0, cnt, wsum, r, s0, e0, sort_ind, sharp_r, lsharp_r, flat_r, lflat_r, err);
    }
  }
  __syncthreads();
  if (threadIdx.x < 4) cnt_r[r * 4 + threadIdx.x] = cnt[threadIdx.x];
}
// ring-major concatenation of the per-ring lists (reference push order)
__global__ __launch_bounds__(256) void k_sr_compact(SrBatch B) {   // one workgroup per ring; every workgroup derives its own offsets
  const int sw = blockIdx.y, o = B.off[sw], n_rings = B.n_rings;
  const int* scan_start = B.ss + sw * n_rings; const int* cnt_r = B.cnt + (size_t)sw * n_rings * 4;
  const int* sharp_r = B.sharp_r + (size_t)sw * n_rings * 16; const int* lsharp_r = B.lsharp_r + (size_t)sw * n_rings * 128; const int* flat_r = B.flat_r + (size_t)sw * n_rings * 32; const int* lflat_r = B.lflat_r + o;
  int* sharp = B.lists + o; int* lsharp = B.lists + B.N + o; int* flat = B.lists + 2 * B.N + o; int* lflat = B.lists + 3 * B.N + o; int* counts = B.counts + 4 * sw;
  __shared__ int off[4];
  const int r = blockIdx.x;
  if (threadIdx.x < 4) {
    int a = 0, tot = 0;
    for (int q = 0; q < n_rings; ++q) { const int v = cnt_r[q * 4 + threadIdx.x]; if (q < r) a += v; tot += v; }
    off[threadIdx.x] = a;
    if (r == 0) counts[threadIdx.x] = tot;
  }
  __syncthreads();
  const int c0 = cnt_r[r * 4], c1 = cnt_r[r * 4 + 1], c2 = cnt_r[r * 4 + 2], c3 = cnt_r[r * 4 + 3];
  for (int t = threadIdx.x; t < c0; t += blockDim.x) sharp[off[0] + t] = sharp_r[r * 16 + t];
  for (int t = threadIdx.x; t < c1; t += blockDim.x) lsharp[off[1] + t] = lsharp_r[r * 128 + t];
  for (int t = threadIdx.x; t < c2; t += blockDim.x) flat[off[2] + t] = flat_r[r * 32 + t];
  for (int t = threadIdx.x; t < c3; t += blockDim.x) lflat[off[3] + t] = lflat_r[(scan_start[r] - 5) + t];
}

// pcl::VoxelGrid over one ring's less-flat points: one workgroup per ring; (voxel index, input position) pairs sorted in LDS (bitonic on a
// 64-bit composite key => equal voxel indices keep their input order), run heads sum their run in float, outputs land in the ring's own slice
#define SRV_CAP 4096
__global__ __launch_bounds__(256) void k_sr_voxelgrid(const float4* cloud, const int* scan_start, const int* cnt_r, const int* lflat_r, float leaf, float4* out_r, int* out_cnt, int* err) {
  __shared__ unsigned long long key[SRV_CAP];
  __shared__ float red[6][256];
  __shared__ int head_pos[SRV_CAP];
  __shared__ int nout_s;
  const int r = blockIdx.x, tid = threadIdx.x;
  const int n = cnt_r[4 * r + 3];
  const int* list = lflat_r + (scan_start[r] - 5);
  if (tid == 0) { out_cnt[r] = 0; nout_s = 0; }
  if (n <= 0) return;
  if (n > SRV_CAP) { if (tid == 0) atomicOr(err, 16); return; }
  float mn[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f}, mx[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
  for (int t = tid; t < n; t += 256) { const float4 q = cloud[list[t]]; mn[0] = fminf(mn[0], q.x); mn[1] = fminf(mn[1], q.y); mn[2] = fminf(mn[2], q.z); mx[0] = fmaxf(mx[0], q.x); mx[1] = fmaxf(mx[1], q.y); mx[2] = fmaxf(mx[2], q.z); }
  for (int k = 0; k < 3; ++k) { red[k][tid] = mn[k]; red[3 + k][tid] = mx[k]; }
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) { if (tid < s2) for (int k = 0; k < 3; ++k) { red[k][tid] = fminf(red[k][tid], red[k][tid + s2]); red[3 + k][tid] = fmaxf(red[3 + k][tid], red[3 + k][tid + s2]); } __syncthreads(); }
  const float inv = 1.0f / leaf;
  int min_b[3], mul[3];
  { int div_b[3]; for (int k = 0; k < 3; ++k) { min_b[k] = (int)floorf(red[k][0] * inv); div_b[k] = (int)floorf(red[3 + k][0] * inv) - min_b[k] + 1; } mul[0] = 1; mul[1] = div_b[0]; mul[2] = div_b[0] * div_b[1]; }
  int np2 = 1; while (np2 < n) np2 <<= 1;
  for (int t = tid; t < np2; t += 256) {
    unsigned long long kv = ~0ull;
    if (t < n) {
      const float4 q = cloud[list[t]];
      const int idx = (int)(floorf(q.x * inv) - (float)min_b[0]) * mul[0] + (int)(floorf(q.y * inv) - (float)min_b[1]) * mul[1] + (int)(floorf(q.z * inv) - (float)min_b[2]) * mul[2];
      kv = ((unsigned long long)(unsigned)idx << 32) | (unsigned)t;
    }
    key[t] = kv;
  }
  __syncthreads();
  for (int k2 = 2; k2 <= np2; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < np2; t += 256) {
        const int p = t ^ j;
        if (p > t) { const bool up = (t & k2) == 0; const unsigned long long a = key[t], b = key[p]; if ((a > b) == up) { key[t] = b; key[p] = a; } }
      }
      __syncthreads();
    }
  // run heads -> output positions (serial scan by one thread: a ring has a few thousand points at most)
  if (tid == 0) { int no = 0; for (int t = 0; t < n; ++t) { const bool head = t == 0 || (key[t] >> 32) != (key[t - 1] >> 32); if (head) { head_pos[t] = no; ++no; } else head_pos[t] = -1; } nout_s = no; out_cnt[r] = no; }
  __syncthreads();
  float4* out = out_r + (scan_start[r] - 5);
  for (int t = tid; t < n; t += 256) {
    if (head_pos[t] < 0) continue;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    int e = t;
    const unsigned vk = (unsigned)(key[t] >> 32);
    for (; e < n && (unsigned)(key[e] >> 32) == vk; ++e) { const float4 q = cloud[list[(unsigned)key[e]]]; c[0] += q.x; c[1] += q.y; c[2] += q.z; c[3] += q.w; }
    const float cnt = (float)(e - t);
    out[head_pos[t]] = make_float4(c[0] / cnt, c[1] / cnt, c[2] / cnt, c[3] / cnt);
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// voxel covariance grid
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }   // monotone float -> int
__host__ __device__ __forceinline__ float ord2f(int i) { const int j = i >= 0 ? i : i ^ 0x7fffffff; float f; memcpy(&f, &j, 4); return f; }
struct VxGrid { int min_b[3], max_b[3], div_b[3], mul[3]; float inv; };
// What a build learns about its cloud, computed ON THE DEVICE (k_vx_extent, k_vx_leaf) and mirrored to pinned host memory behind the last kernel: the host reads it
// when somebody needs it (vox_info), never in the middle of the kernel chain — round 3 stopped the chain twice (extents -> size of the cell table; leaf count -> strides of
// the leaf arrays).  overflow: 1 = the dense cell table (capacity cells_cap) is too small for this cloud's extents (the host grows it and builds again), 2 = more than
// 2^31 - 1 cells (the reference's "Leaf size is too small" error, voxel_grid_covariance_omp_impl.hpp:80-85).
struct VxInfo { VxGrid g; int n_leaves, overflow; long long cells; };
// extents + grid geometry (:86-95) in ONE launch: per-thread min / max over a grid-stride range, wavefront shuffles, one atomic per workgroup and bound; the LAST workgroup
// to finish (ticket) turns the extents into the grid geometry and leaves mm[] in its start state for the next build (mm = 3 x INT_MAX | 3 x INT_MIN | ticket 0, set when
// the buffer is allocated).  Was three launches of ~4.6 us each (k_vx_init, k_vx_minmax, k_vx_grid): a launch of a captured graph costs that much whatever it does.
__global__ __launch_bounds__(256) void k_vx_extent(const float4* p, int n, int* mm, float leaf, long long cells_cap, VxInfo* info, int* ghist) {
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  // at most 256 workgroups (every one ends in same-address traffic at the memory side: 976 of them took 80-90 us at 4 M points, 256 take 45), four loads in flight per thread
  const int stride = gridDim.x * blockDim.x;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  auto upd = [&](const float4 q) {
    if (!isfinite(q.x) || !isfinite(q.y) || !isfinite(q.z)) return;
    const int o[3] = {f2ord(q.x), f2ord(q.y), f2ord(q.z)};
    for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], o[a]); hi[a] = max(hi[a], o[a]); }
  };
  for (; (long long)i + 3ll * stride < n; i += 4 * stride) {
    const float4 q0 = p[i], q1 = p[i + stride], q2 = p[i + 2 * stride], q3 = p[i + 3 * stride];
    upd(q0); upd(q1); upd(q2); upd(q3);
  }
  for (; i < n; i += stride) upd(p[i]);
  for (int s = 32; s > 0; s >>= 1) for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], __shfl_xor(lo[a], s)); hi[a] = max(hi[a], __shfl_xor(hi[a], s)); }
  __shared__ int red[4][6];
  __shared__ int s_last;
  if ((threadIdx.x & 63) == 0) for (int a = 0; a < 3; ++a) { red[threadIdx.x >> 6][a] = lo[a]; red[threadIdx.x >> 6][3 + a] = hi[a]; }
  __syncthreads();
  // bounds (only where this workgroup improves them: most cannot any more), then the ticket.  The bounds are device-scope atomics (performed at the memory side, not in
  // this XCD's L2) and they have RETURNED before the ticket is drawn: no fence — an agent-scope release / acquire writes back and invalidates the XCD's L2 under the other
  // workgroups' streaming reads.
  __shared__ int ex[8];
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    int v = red[0][k];
    for (int w = 1; w < 4; ++w) v = k < 3 ? min(v, red[w][k]) : max(v, red[w][k]);
    const int cur = __hip_atomic_load(&mm[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int old = 0;
    if (k < 3) { if (v < cur) old = __hip_atomic_fetch_min(&mm[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else if (v > cur) old = __hip_atomic_fetch_max(&mm[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ex[k] = old;   // (the store waits for the atomic's return)
  }
  __syncthreads();
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&mm[6], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  if (ghist) for (int e = threadIdx.x; e < 4 * 1024; e += 256) ghist[e] = 0;   // digit histograms of the radix sort, 1 024 bins per pass at most (k_vx_keys adds to them)
  if (threadIdx.x < 7) ex[threadIdx.x] = atomicExch(&mm[threadIdx.x], threadIdx.x < 3 ? 0x7fffffff : (threadIdx.x < 6 ? (int)0x80000000 : 0));   // read at the memory side and reset
  __syncthreads();
  if (threadIdx.x != 0) return;
  int e[6];
  for (int k = 0; k < 6; ++k) e[k] = ex[k];
  VxGrid g;
  g.inv = 1.0f / leaf;
  const bool empty = e[0] == 0x7fffffff;
  for (int k = 0; k < 3; ++k) {
    g.min_b[k] = empty ? 0 : (int)floorf(ord2f(e[k]) * g.inv); g.max_b[k] = empty ? -1 : (int)floorf(ord2f(e[3 + k]) * g.inv);
    g.div_b[k] = g.max_b[k] - g.min_b[k] + 1;
  }
  const long long cells = (long long)g.div_b[0] * g.div_b[1] * g.div_b[2];
  g.mul[0] = 1; g.mul[1] = g.div_b[0]; g.mul[2] = g.div_b[0] * g.div_b[1];
  info->g = g; info->cells = cells; info->n_leaves = 0;
  info->overflow = cells > 2147483647LL ? 2 : (cells > cells_cap ? 1 : 0);
}
// keys: the voxel's linear index, or `invalid` (> every valid key, inside the sorted bit range) for non-finite points; the same launch empties the part of the dense cell
// table this cloud's extents span, resets the look-back states of k_vx_leaf (one per tile of sorted positions) and — for the own radix sort — counts the keys' 8-bit
// digits (LDS histogram per workgroup, one global atomic per non-empty bin) and clears the per-tile digit counts of the sort passes
__global__ __launch_bounds__(256) void k_vx_keys(const float4* p, int n, const VxInfo* info, unsigned invalid, unsigned* keys, int* vals, int* cells, unsigned long long* lb, int n_tiles,
                                                 int* ghist, unsigned* tcnt, int n_tcnt, int passes, int db) {
  __shared__ int lh[4][1024];   // per pass: 1 << db bins (db = 8, 9 or 10)
  const int nbins = 1 << db;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool ovf = info->overflow != 0;
  if (!ovf) { const long long nc = info->cells; for (long long e = i; e < nc; e += (long long)gridDim.x * blockDim.x) cells[e] = -1; }
  if (i < n_tiles) lb[i] = 0ull;
  if (tcnt) for (int e = i; e < n_tcnt; e += gridDim.x * blockDim.x) tcnt[e] = 0u;
  if (blockIdx.x * 1024 >= n) return;   // (workgroups that only clear)
  if (ghist) { for (int q = 0; q < passes; ++q) for (int e = threadIdx.x; e < nbins; e += 256) lh[q][e] = 0; __syncthreads(); }
  const VxGrid g = info->g;
#pragma unroll
  for (int u = 0; u < 4; ++u) {   // 1 024 points per workgroup: a quarter of the global histogram atomics of 256
    const int j = blockIdx.x * 1024 + 256 * u + threadIdx.x;
    if (j >= n) break;
    const float4 q = p[j];
    unsigned k = invalid;
    if (!ovf && isfinite(q.x) && isfinite(q.y) && isfinite(q.z)) {
      const int i0 = (int)(floorf(q.x * g.inv) - (float)g.min_b[0]);   // :220-222
      const int i1 = (int)(floorf(q.y * g.inv) - (float)g.min_b[1]);
      const int i2 = (int)(floorf(q.z * g.inv) - (float)g.min_b[2]);
      k = (unsigned)(i0 * g.mul[0] + i1 * g.mul[1] + i2 * g.mul[2]);
    }
    keys[j] = k; vals[j] = j;
    if (ghist) for (int qq = 0; qq < passes; ++qq) atomicAdd(&lh[qq][(k >> (db * qq)) & (unsigned)(nbins - 1)], 1);
  }
  if (ghist) {
    __syncthreads();
    for (int q = 0; q < passes; ++q) for (int e = threadIdx.x; e < nbins; e += 256) { const int v = lh[q][e]; if (v) atomicAdd(&ghist[1024 * q + e], v); }
  }
}
// One pass of a stable least-significant-digit radix sort (8-bit digit) in ONE launch ("onesweep"), for clouds of up to VX_OWN_SORT_MAX points — rocPRIM's radix sort takes
// ten launches at this size (histogram / scan / scatter per digit), 56 us of the 105 us build at 100 k points, each launch ~4.6 us whatever it does.  A workgroup owns a
// tile of 2 048 consecutive positions (striped over the threads: position = tile + 256 u + thread, so (u, wavefront, lane) order IS input order):
//   1. rank inside the wavefront by eight ballots per key (lanes with the same digit), the group's count goes to hist[u][wavefront][digit];
//   2. thread d turns the 32 groups of digit d into exclusive offsets and publishes the tile's count of d (flag | count, one word per (tile, digit));
//   3. first position of digit d in this tile = digits below d in the whole array (the histogram k_vx_keys built) + d's keys in the tiles before this one: a look-back
//      over the published counts, sixteen words at a time, until a tile that already knows its own inclusive sum;
//   4. scatter.
#ifndef VX_OWN_SORT_MAX
#define VX_OWN_SORT_MAX 1048576
#endif
// (measured: 400 k points 195 -> 134 us against rocPRIM, 4 M points 213 -> 284 us: rocPRIM there)
// DB = digit width (round 5b): 1 << DB threads per workgroup, a THREAD per digit value as before, the tile stays 1 024 positions — 8 bits: 256 threads x 4 keys, 9 bits:
// 512 x 2, 10 bits: 1 024 x 1.  A key range of up to 20 bits (a voxel grid of up to a million cells) sorts in TWO launches instead of three.
template <bool FIRST, int DB>
__global__ __launch_bounds__(1 << DB) void k_vx_sort_pass(const unsigned* __restrict__ kin, const int* __restrict__ vin, unsigned* __restrict__ kout, int* __restrict__ vout, int n, int shift,
                                                          const int* __restrict__ ghist, unsigned* tcnt) {
  constexpr int NT = 1 << DB, NW = NT / 64, KPT = 1024 / NT;
  __shared__ int hist[KPT * NW * NT];
  __shared__ int base[NT];
  __shared__ int wsum[NW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x;
  const int tile0 = b * NT * KPT;
  for (int e = tid; e < KPT * NW * NT; e += NT) hist[e] = 0;
  const int gh = ghist[tid];   // (requested with the keys: not behind the tile's own counting)
  unsigned key[KPT]; int val[KPT]; int rk[KPT];
#pragma unroll
  for (int u = 0; u < KPT; ++u) {
    const int idx = tile0 + NT * u + tid;
    key[u] = idx < n ? kin[idx] : 0xffffffffu;
    val[u] = FIRST ? idx : (idx < n ? vin[idx] : 0);
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int u = 0; u < KPT; ++u) {
    const bool valid = tile0 + NT * u + tid < n;
    const unsigned dg = (key[u] >> shift) & (unsigned)(NT - 1);
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < DB; ++bit) {
      const bool one = (dg >> bit) & 1u;
      const unsigned long long bal = __ballot(valid && one);
      m &= one ? bal : ~bal;
    }
    rk[u] = __popcll(m & lt);
    if (valid && rk[u] == 0) hist[(NW * u + wv) * NT + dg] = __popcll(m);
  }
  __syncthreads();
  int cnt = 0;
#pragma unroll 8
  for (int g = 0; g < NW * KPT; ++g) { const int t = hist[g * NT + tid]; hist[g * NT + tid] = cnt; cnt += t; }
  __hip_atomic_store(&tcnt[b * NT + tid], ((b == 0 ? 2u : 1u) << 30) | (unsigned)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // digits below mine in the whole array
  int incl = gh;
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int first = incl - gh;
  for (int w = 0; w < wv; ++w) first += wsum[w];
  // my digit's keys in the tiles before this one
  int acc = 0;
  constexpr int LBW = 16;   // words in flight per thread: a tile walks back LBW tiles per round trip while the inclusive sums advance towards it (measured per pass at 100 k keys: 8 and 16 words 9.5 us, 32 words 12.6 us)
  for (int t = b - 1; t >= 0; t -= LBW) {
    unsigned v[LBW];
#pragma unroll
    for (int j = 0; j < LBW; ++j) v[j] = t - j >= 0 ? __hip_atomic_load(&tcnt[(t - j) * NT + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2u << 30);
    bool done = false;
#pragma unroll
    for (int j = 0; j < LBW; ++j) {
      if (done) continue;
      while ((v[j] >> 30) == 0u) v[j] = __hip_atomic_load(&tcnt[(t - j) * NT + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += (int)(v[j] & 0x3fffffffu);
      if ((v[j] >> 30) == 2u) done = true;
    }
    if (done) break;
  }
  if (b > 0) __hip_atomic_store(&tcnt[b * NT + tid], (2u << 30) | (unsigned)(acc + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  base[tid] = first + acc;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < KPT; ++u) {
    if (tile0 + NT * u + tid >= n) continue;
    const unsigned dg = (key[u] >> shift) & (unsigned)(NT - 1);
    const int pos = base[dg] + hist[(NW * u + wv) * NT + dg] + rk[u];
    kout[pos] = key[u]; vout[pos] = val[u];
  }
}
// cyclic Jacobi, ascending eigenvalues (Eigen SelfAdjointEigenSolver stand-in).  EARLY: stop once the off-diagonal part is below 1e-18 of the diagonal (further sweeps
// rotate by angles that no longer change a double: the oracle's loop runs 2-3 more of them until the squares underflow) and take c from one rsqrt instead of sqrt + division.
template <bool EARLY>
__device__ void vx_eig3(const double A[9], double ev[3], double V[9]) {
  double a[3][3] = {{A[0], A[1], A[2]}, {A[3], A[4], A[5]}, {A[6], A[7], A[8]}};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off < 1e-300) break;
    if (EARLY && off <= 1e-36 * (a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2])) break;
    for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
      if (a[p][q] == 0.0) continue;
      const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = EARLY ? rsqrt(t * t + 1.0) : 1.0 / sqrt(t * t + 1.0), s = t * c;
      for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
      for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
      for (int k = 0; k < 3; ++k) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq; }
    }
  }
  int i0 = 0, i1 = 1, i2 = 2;
  if (a[i1][i1] < a[i0][i0]) { int t = i0; i0 = i1; i1 = t; }
  if (a[i2][i2] < a[i1][i1]) { int t = i1; i1 = i2; i2 = t; if (a[i1][i1] < a[i0][i0]) { int u = i0; i0 = i1; i1 = u; } }
  const int idx[3] = {i0, i1, i2};
  for (int k = 0; k < 3; ++k) { ev[k] = a[idx[k]][idx[k]]; for (int r = 0; r < 3; ++r) V[3 * r + k] = v[r][idx[k]]; }
}
// Leaves from the (stable-)sorted keys in ONE launch — run heads, leaf numbering, point lists, sums, eigen-solve, finalize (:286-371).  A workgroup owns a TILE of 256
// sorted positions and the leaves whose first point lies in it:
//   * head = position whose key differs from its predecessor's; the number of heads before the tile comes from a decoupled look-back over the tiles' published counts
//     (one 64-bit word per tile: flag | count, wavefront 0 looks at 64 predecessors per step) — this replaced rocPRIM's run-length encode + exclusive scan (four launches);
//   * the sorted ids and points of the tile's leaves are one contiguous range: all 256 threads fetch it (coalesced ids, one gather) into LDS —
//     the thread-per-leaf loop of round 3 paid two dependent HBM round trips per 8 points of its LONGEST leaf (35-50 % of that kernel);
//   * thread r < #heads then sums ITS leaf's points from LDS in input order: cov = (sum x x^T - 2 sum x mu^T) / n + mu mu^T cancels ~6 digits (coordinates of tens of
//     metres, spreads of centimetres), so only the reference's summation order reproduces it to 1e-12 of its own scale; the float centroid is compared bit for bit.
__device__ __forceinline__ unsigned long long vx_lb_load(const unsigned long long* a) { return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void vx_lb_store(unsigned long long* a, unsigned long long v) { __hip_atomic_store(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// PT = sorted positions per thread (tile = 256 PT): 1 for small clouds (many workgroups, the launch is a latency chain), 4 for large ones (a tile then holds ~140 heads:
// the eigen-solves run on two to three FULL wavefronts per workgroup instead of on half of one — at 4 M points the kernel is bound by resident eigen-solve wavefronts).
#define VX_LONG 128
template <int PT>
__global__ __launch_bounds__(256) void k_vx_leaf(const float4* __restrict__ p, const unsigned* __restrict__ skeys, const int* __restrict__ sorted_ids, int n_pts, unsigned invalid, unsigned long long* lb,
                                                 int min_pts, double eig_mult, VxInfo* info, int* grid, int* leaf_key, int* leaf_n, unsigned* counts, unsigned* offs,
                                                 double* mean, double* cov, double* icov, double* evecs, double* evals, float* centroid, VxInfo* h_info) {
  constexpr int TILE = 256 * PT;
  __shared__ int hpos[TILE + 1];
  __shared__ int wcnt[PT * 4 < 4 ? 4 : PT * 4];
  __shared__ int s_prefix, s_end;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x;
  const int tile0 = b * TILE;
#ifdef VXKT
  long long kt_[8]; int kti_ = 0; const long long rt0_ = wall_clock64();
#define VXKT_MARK() { long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); kt_[kti_++] = t_; }
#else
#define VXKT_MARK()
#endif
  VXKT_MARK()
  const bool ovf = info->overflow != 0;
  bool head[PT]; int rank[PT];
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    const int i = tile0 + 256 * u + tid;
    const unsigned key_i = i < n_pts ? skeys[i] : invalid;
    head[u] = !ovf && key_i != invalid && (i == 0 || skeys[i - 1] != key_i);
    const unsigned long long hm = __ballot(head[u]);
    if (lane == 0) wcnt[4 * u + wv] = __popcll(hm);
    rank[u] = __popcll(hm & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  int hb = 0;
#pragma unroll
  for (int e = 0; e < 4 * PT; ++e) hb += wcnt[e];
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    int r = rank[u];
    for (int e = 0; e < 4 * u + wv; ++e) r += wcnt[e];
    if (head[u]) hpos[r] = tile0 + 256 * u + tid;
  }
  // leaves before this tile: every tile publishes its own count as soon as it has it (one 64-bit word: flag | count) and sums ALL its predecessors' words — every
  // load in flight together, one round trip once they are out.  (Rounds 3-5a: decoupled look-back over inclusive prefixes, 64 predecessors per step — the last of 400
  // tiles of a 410 k-point cloud walked seven dependent steps, 14 us of the kernel's 71.)
  {
    if (tid == 0) vx_lb_store(&lb[b], (1ull << 32) | (unsigned)hb);
    int part = 0;
    for (int j0 = tid; j0 < b; j0 += 256 * 8) {
      unsigned long long v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = j0 + 256 * u < b ? vx_lb_load(&lb[j0 + 256 * u]) : (1ull << 32);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        while ((v[u] >> 32) == 0ull) v[u] = vx_lb_load(&lb[j0 + 256 * u]);
        part += (int)(unsigned)(v[u] & 0xffffffffull);
      }
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    __syncthreads();                     // (wcnt was read above by everybody)
    if (lane == 0) wcnt[wv] = part;
    __syncthreads();
    if (tid == 0) {
      const int prefix = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      s_prefix = prefix; s_end = -1;
      if (b == (int)gridDim.x - 1) { info->n_leaves = prefix + hb; VxInfo t = *info; t.n_leaves = prefix + hb; *h_info = t; }   // the host's mirror (pinned memory), read after a stream synchronisation
    }
  }
  __syncthreads();
  VXKT_MARK()
  if (hb == 0) return;
  // end of the tile's last leaf: the first position behind its head with another key
  {
    const int lasthead = hpos[hb - 1];
    const unsigned lastkey = skeys[lasthead];
    for (int pos = tile0 + ((lasthead - tile0) & ~255);; pos += 256) {   // (from inside the tile: the invalid-key run of the non-finite points may begin there)
      const int j = pos + tid;
      const bool stop = j > lasthead && (j >= n_pts || skeys[j] != lastkey);
      const unsigned long long sm = __ballot(stop);
      if (lane == 0) wcnt[wv] = sm ? __ffsll((long long)sm) - 1 : 64;
      __syncthreads();
      int e = -1;
      for (int w = 3; w >= 0; --w) if (wcnt[w] < 64) e = pos + 64 * w + wcnt[w];
      __syncthreads();
      if (e >= 0) { if (tid == 0) { s_end = min(e, n_pts); hpos[hb] = min(e, n_pts); } break; }
    }
  }
  __syncthreads();
  const int S = hpos[0], E = s_end;
  // finalize of :286-371 for one leaf from its sums
  auto finalize = [&](int r, int la, int n, const double* s, const double* c, const float* cen) {
    const int li = s_prefix + r;
    const unsigned key = skeys[la];
    leaf_key[li] = (int)key;
    grid[key] = li;
    counts[li] = (unsigned)n; offs[li] = (unsigned)la;
    double* M = mean + 3 * (size_t)li; double* Cv = cov + 9 * (size_t)li; double* IC = icov + 9 * (size_t)li; double* EV = evecs + 9 * (size_t)li; double* EL = evals + 3 * (size_t)li;
    double mu[3];
    for (int a = 0; a < 3; ++a) { centroid[3 * (size_t)li + a] = cen[a] / (float)n; mu[a] = s[a] / n; M[a] = mu[a]; }
    int nr = n;
    if (n < min_pts) {
      for (int a = 0; a < 3; ++a) EL[a] = 0.0;
      for (int a = 0; a < 9; ++a) { Cv[a] = (a % 4 == 0) ? 1.0 : 0.0; IC[a] = 0.0; EV[a] = (a % 4 == 0) ? 1.0 : 0.0; }
    } else {
      const double cs[9] = {c[0], c[1], c[2], c[1], c[3], c[4], c[2], c[4], c[5]};
      double C[9];
      for (int a = 0; a < 3; ++a) for (int bb = 0; bb < 3; ++bb) C[3 * a + bb] = (cs[3 * a + bb] - 2 * (s[a] * mu[bb])) / n + mu[a] * mu[bb];
      for (int a = 0; a < 9; ++a) C[a] *= (n - 1.0) / n;
      double ev[3], V[9];
      vx_eig3<true>(C, ev, V);
      for (int a = 0; a < 9; ++a) EV[a] = V[a];
      double el[3] = {0.0, 0.0, 0.0}, ic[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) nr = -1;
      else {
        const double min_ev = eig_mult * ev[2];
        if (ev[0] < min_ev) {
          ev[0] = min_ev; if (ev[1] < min_ev) ev[1] = min_ev;
          for (int a = 0; a < 3; ++a) for (int bb = 0; bb < 3; ++bb) C[3 * a + bb] = V[3 * a] * ev[0] * V[3 * bb] + V[3 * a + 1] * ev[1] * V[3 * bb + 1] + V[3 * a + 2] * ev[2] * V[3 * bb + 2];
        }
        for (int a = 0; a < 3; ++a) el[a] = ev[a];
        const double c00 = C[4] * C[8] - C[5] * C[7], c01 = C[5] * C[6] - C[3] * C[8], c02 = C[3] * C[7] - C[4] * C[6];
        const double id = 1.0 / (C[0] * c00 + C[1] * c01 + C[2] * c02);
        ic[0] = c00 * id; ic[1] = (C[2] * C[7] - C[1] * C[8]) * id; ic[2] = (C[1] * C[5] - C[2] * C[4]) * id;
        ic[3] = c01 * id; ic[4] = (C[0] * C[8] - C[2] * C[6]) * id; ic[5] = (C[2] * C[3] - C[0] * C[5]) * id;
        ic[6] = c02 * id; ic[7] = (C[1] * C[6] - C[0] * C[7]) * id; ic[8] = (C[0] * C[4] - C[1] * C[3]) * id;
        double mxv = ic[0], mnv = ic[0];
        for (int a = 1; a < 9; ++a) { mxv = fmax(mxv, ic[a]); mnv = fmin(mnv, ic[a]); }
        if (mxv == (double)INFINITY || mnv == -(double)INFINITY) nr = -1;
      }
      for (int a = 0; a < 3; ++a) EL[a] = el[a];
      for (int a = 0; a < 9; ++a) { Cv[a] = C[a]; IC[a] = ic[a]; }
    }
    leaf_n[li] = nr;
  };
  // SUMS.  A leaf needs twelve running sums — x y z | xx xy xz yy yz zz in double, x y z in float — each IN INPUT ORDER (the sort is stable): cov = (sum x x^T - 2 sum x
  // mu^T) / n + mu mu^T cancels ~6 digits, only the reference's summation order reproduces it to 1e-12 of its own scale, and the float centroid is compared bit for bit.
  // Twelve independent chains per leaf: a group of 16 lanes takes a leaf, lane j its chain j.  The workgroup walks its positions [S, E) in stages of 256: every thread
  // fetches one point (ids two stages ahead, points one stage ahead of the walk) and leaves its nine double terms (x * 1.0 and the products of two floats are exact) and
  // three float terms in LDS; then the groups walk the leaves that intersect the stage, 8 leaves per sub-round; a leaf that continues into the next stage carries its
  // partial sums in the LDS slot its final sums go to.  Thread-per-leaf finalize (eigen-solve) afterwards, 128 leaves at a time.
  // (Round 4: thread-per-leaf sums for leaves <= 128 points + one 16-lane group per workgroup for longer ones, 83 us on the map cloud of lvx_data_association — 4 072
  // leaves, 100 points on average, the longest 2 298 — of which 56 us were the long-leaf path: seven instructions per point and chain, leaf after leaf.)
  VXKT_MARK()
  {
    constexpr int LDD = 257;
    __shared__ double sD[9][LDD];
    __shared__ float sF[3][256];
    constexpr int LCH = 128;            // leaves per finalize chunk
    __shared__ double sumD[9][LCH];
    __shared__ float sumF[3][LCH];
    const int g = (tid >> 4) & 7, j = tid & 15;   // leaf slot of the sub-round, chain
    const bool fchain = wv >= 2;                   // (wave-uniform)
    const int jd = j < 9 ? j : 0, jf = j < 3 ? j : 0;
    for (int c0 = 0; c0 < hb; c0 += LCH) {          // leaves [c0, c1) of the tile
      const int c1 = min(hb, c0 + LCH);
      const int P0 = hpos[c0], P1 = hpos[c1];
      auto ld_id = [&](int pos) { return pos + tid < P1 ? sorted_ids[pos + tid] : -1; };
      auto ld_pt = [&](int id) { return id >= 0 ? p[id] : float4{0.f, 0.f, 0.f, 0.f}; };
      int id_nx = ld_id(P0);
      float4 q = ld_pt(id_nx);
      id_nx = ld_id(P0 + 256);
      int r_lo = c0;                                 // first leaf not yet complete
      for (int st0 = P0; st0 < P1; st0 += 256) {
        const int st1 = min(st0 + 256, P1);
        {
          const double x = q.x, y = q.y, z = q.z;
          sD[0][tid] = x; sD[1][tid] = y; sD[2][tid] = z;
          sD[3][tid] = x * x; sD[4][tid] = x * y; sD[5][tid] = x * z; sD[6][tid] = y * y; sD[7][tid] = y * z; sD[8][tid] = z * z;
          sF[0][tid] = q.x; sF[1][tid] = q.y; sF[2][tid] = q.z;
        }
        // leaves that begin before st1: r_lo .. r_hi (at most 256 begin inside a stage, plus the one that continues into it)
        const int cand = r_lo + tid;
        const int nbeg = __syncthreads_count(cand < c1 && hpos[cand] < st1);   // (also the barrier behind the LDS writes)
        const int r_hi = r_lo + nbeg - 1 + ((nbeg == 256 && r_lo + 256 < c1 && hpos[r_lo + 256] < st1) ? 1 : 0);
        q = ld_pt(id_nx);                            // next stage's points, next-but-one's ids: in flight during the walk
        id_nx = ld_id(st0 + 512);
        // the chain: ONE dependent add per point and lane.  Wavefronts 0-1 walk the nine double chains of leaf slots 0-7 (a 16-lane group per slot), wavefronts 2-3 the
        // three float chains of the same slots: an FP64 add occupies the SIMD for eight cycles, the float add rode in the same instruction stream for four more.  The
        // operands of the next eight points are fetched while a batch is added — two batches per trip in named registers, no copies that would wait for the loads just
        // issued.  (Plain loop: 50 cycles per point, 48 of the kernel's 71 us were the 2 298-point leaf's nine stages; pipelined, both chains in one wavefront: 30.)
        for (int rb = r_lo; rb <= r_hi; rb += 8) {
          const int r = rb + g;
          if (r > r_hi) continue;
          const int la = hpos[r], le = hpos[r + 1];
          const int k0 = max(la, st0) - st0, k1 = min(le, st1) - st0;
          constexpr int WB = 8;
          const int nfull = (k1 - k0) / WB;
          auto walk = [&](auto zero, const auto* src) {
            using T = decltype(zero);
            T acc = zero;
            T A[WB], B[WB];
            auto fetch = [&](T* D_, int k) {   // (past the leaf's end: loaded, never added — LDS reads do not fault)
#pragma unroll
              for (int u = 0; u < WB; ++u) D_[u] = src[k + u];
            };
            auto add = [&](const T* D_) {
#pragma unroll
              for (int u = 0; u < WB; ++u) acc += D_[u];
            };
            int i = 0;
            if (nfull > 0) fetch(A, k0);
            for (; i + 2 <= nfull; i += 2) {
              fetch(B, k0 + WB * (i + 1));
              add(A);
              fetch(A, k0 + WB * (i + 2));
              add(B);
            }
            if (i < nfull) add(A);
            for (int k = k0 + WB * nfull; k < k1; ++k) acc += src[k];
            return acc;
          };
          if (!fchain) {
            const double acc = walk(la < st0 ? sumD[jd][r - c0] : 0.0, &sD[jd][0]);
            if (j < 9) sumD[j][r - c0] = acc;
          } else {
            const float accf = walk(la < st0 ? sumF[jf][r - c0] : 0.0f, &sF[jf][0]);
            if (j < 3) sumF[j][r - c0] = accf;
          }
        }
        r_lo = hpos[r_hi + 1] <= st1 ? r_hi + 1 : r_hi;   // the last leaf of the stage may continue
        __syncthreads();
      }
      VXKT_MARK()
      if (c0 + tid < c1) {
        const int r = c0 + tid;
        double sm[3], cv[6]; float cen[3];
        for (int a = 0; a < 3; ++a) { sm[a] = sumD[a][tid]; cen[a] = sumF[a][tid]; }
        for (int a = 0; a < 6; ++a) cv[a] = sumD[3 + a][tid];
        finalize(r, hpos[r], hpos[r + 1] - hpos[r], sm, cv, cen);
      }
      __syncthreads();
      VXKT_MARK()
    }
  }
#ifdef VXKT
  { const long long rt1_ = wall_clock64();
    if (tid == 0) printf("VXKT b %d hb %d pts %d: heads %lld endsearch %lld sums %lld finalize %lld total %lld rt %lld %lld\n", b, hb, E - S, kt_[1] - kt_[0], kt_[2] - kt_[1], kt_[3] - kt_[2], kt_[4] - kt_[3], kt_[kti_ - 1] - kt_[0], rt0_, rt1_); }
#endif
}
// ------------------------------------------------------------------------------------------------------------------------
// NDT registration (pclomp::NormalDistributionsTransform, src/ndt_omp/include/pclomp/ndt_omp_impl.hpp): derivative evaluation, Hessian-only pass, fitness.
// ------------------------------------------------------------------------------------------------------------------------
// Displacement nb of a neighbourhood of NB cells, in the order the reference pushes them: NB <= 7 = {0, +x, -x, +y, -y, +z, -z} (voxel_grid_covariance_omp_impl.hpp:427-434,
// NB = 1 its first entry :440-446); NB = 26 = pcl::getAllNeighborCellIndices() (:411-419; pcl/filters/voxel_grid.h: (i, j, -1) for i, j = -1..1, (i, -1, 0), (-1, 0, 0),
// then their negatives).  Fully unrolled callers fold this into constants.
template <int NB> __device__ __forceinline__ int ndt_disp(int nb, int a) {
  if (NB <= 7) { const int d[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}}; return d[nb][a]; }
  const int h = nb % 13, sg = nb < 13 ? 1 : -1;
  const int v = h < 9 ? (a == 0 ? h / 3 - 1 : (a == 1 ? h % 3 - 1 : -1)) : (h < 12 ? (a == 0 ? h - 10 : (a == 1 ? -1 : 0)) : (a == 0 ? -1 : 0));
  return sg * v;
}
// pcl::transformPointCloud, dense cloud (PCL <= 1.8 scalar form): M(0..2, 0..2) * p + M(0..2, 3), summed left to right in float (the file is built without contraction)
struct NdtMat { float m[12]; };
__device__ __forceinline__ float4 ndt_xf(const NdtMat& M, const float4 p) {
  float4 o;
  o.x = ((M.m[0] * p.x + M.m[1] * p.y) + M.m[2] * p.z) + M.m[3];
  o.y = ((M.m[4] * p.x + M.m[5] * p.y) + M.m[6] * p.z) + M.m[7];
  o.z = ((M.m[8] * p.x + M.m[9] * p.y) + M.m[10] * p.z) + M.m[11];
  o.w = p.w;
  return o;
}
__global__ __launch_bounds__(256) void k_ndt_transform(const float4* src, int n, NdtMat M, float4* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ndt_xf(M, src[i]);
}
// Sum of NV doubles per thread over the whole launch, in a FIXED order (run-to-run identical): lanes by butterfly, the four wavefronts in order, workgroups in
// index order by the last workgroup to arrive (ticket).  part: [gridDim.x][NV]; ticket: one int, zero before the launch and left zero.
template <int NV>
__device__ __forceinline__ void ndt_block_reduce(double* acc, double* part, int* ticket, double* out) {
  __shared__ double red[4][NV];
  __shared__ int last;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    double v = acc[e];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[wv][e] = v;
  }
  __syncthreads();
  if (threadIdx.x < NV) part[(size_t)blockIdx.x * NV + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x < NV) {
    double v = 0.0;
    for (unsigned b = 0; b < gridDim.x; ++b) v += __builtin_nontemporal_load(&part[(size_t)b * NV + threadIdx.x]);
    out[threadIdx.x] = v;
  }
  if (threadIdx.x == 0) *ticket = 0;
}
// computeDerivatives (:180-285) with updateDerivatives (:484-536) and the float computePointDerivatives (:398-439): one thread per point, float arithmetic in the
// reference's order of operations, double accumulation.  XF: the transformed point is computed here from the source point (the loop of lvx_ndt_align); otherwise it is
// read from trn (lvx_ndt_derivatives: the caller transformed the cloud).
struct NdtConst { float j_ang[8][3]; float h_ang[15][3]; float gd2; double gauss_d1; };
template <int NB, bool XF>
__global__ __launch_bounds__(256) void k_ndt_derivatives(const float4* src, const float4* trn, NdtMat M, int n, float leaf, int min_pts, VxGrid g, const int* grid, const int* leaf_n,
                                                         const double* mean, const double* icov, NdtConst K, int compute_hessian, double* part, int* ticket, double* out43) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  double acc[43];
#pragma unroll
  for (int e = 0; e < 43; ++e) acc[e] = 0.0;
  if (idx < n) {
    const float4 xi = src[idx];
    const float4 xt = XF ? ndt_xf(M, xi) : trn[idx];
    float pg[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
    float xj[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) xj[r] = (K.j_ang[r][0] * xi.x + K.j_ang[r][1] * xi.y) + K.j_ang[r][2] * xi.z;
    pg[1][3] = xj[0]; pg[2][3] = xj[1]; pg[0][4] = xj[2]; pg[1][4] = xj[3]; pg[2][4] = xj[4]; pg[0][5] = xj[5]; pg[1][5] = xj[6]; pg[2][5] = xj[7];
    float hv[6][3];   // a, b, c, d, e, f of Equation 6.21
#pragma unroll
    for (int r = 0; r < 6; ++r) hv[r][0] = hv[r][1] = hv[r][2] = 0.0f;
    if (compute_hessian) {
      float xh[15];
#pragma unroll
      for (int r = 0; r < 15; ++r) xh[r] = (K.h_ang[r][0] * xi.x + K.h_ang[r][1] * xi.y) + K.h_ang[r][2] * xi.z;
      hv[0][1] = xh[0]; hv[0][2] = xh[1]; hv[1][1] = xh[2]; hv[1][2] = xh[3]; hv[2][1] = xh[4]; hv[2][2] = xh[5];
      hv[3][0] = xh[6]; hv[3][1] = xh[7]; hv[3][2] = xh[8]; hv[4][0] = xh[9]; hv[4][1] = xh[10]; hv[4][2] = xh[11]; hv[5][0] = xh[12]; hv[5][1] = xh[13]; hv[5][2] = xh[14];
    }
    // second-derivative block (i, j), i, j in 3..5: a b c / b d e / c e f
    const int hsel[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    const int ijk[3] = {(int)floorf(xt.x / leaf), (int)floorf(xt.y / leaf), (int)floorf(xt.z / leaf)};
    // the NB cell probes in flight, then the NB leaf counts (as in k_vx_lookup: probe by probe the loads queue up one behind the other); cells are still visited
    // in the reference's order
    int lis[NB]; bool use[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      bool in = true;
#pragma unroll
      for (int a = 0; a < 3; ++a) in = in & (g.min_b[a] - ijk[a] <= ndt_disp<NB>(nb, a)) & (g.max_b[a] - ijk[a] >= ndt_disp<NB>(nb, a));
      const int key = (ijk[0] + ndt_disp<NB>(nb, 0) - g.min_b[0]) * g.mul[0] + (ijk[1] + ndt_disp<NB>(nb, 1) - g.min_b[1]) * g.mul[1] + (ijk[2] + ndt_disp<NB>(nb, 2) - g.min_b[2]) * g.mul[2];
      const int l0 = grid[in ? key : 0];
      lis[nb] = in ? l0 : -1;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { const int cn = leaf_n[lis[nb] >= 0 ? lis[nb] : 0]; use[nb] = lis[nb] >= 0 && cn >= min_pts; }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (!use[nb]) continue;
      const int li = lis[nb];
      const double* mu = mean + 3 * (size_t)li; const double* ic = icov + 9 * (size_t)li;
      const float x4[3] = {(float)((double)xt.x - mu[0]), (float)((double)xt.y - mu[1]), (float)((double)xt.z - mu[2])};
      float ci[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) ci[r][cc] = (float)ic[3 * r + cc];
      float xc[3];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) xc[cc] = (x4[0] * ci[0][cc] + x4[1] * ci[1][cc]) + x4[2] * ci[2][cc];
      const float q = (x4[0] * xc[0] + x4[1] * xc[1]) + x4[2] * xc[2];
      float e_x = expf(-K.gd2 * q * 0.5f);
      const float score_inc = (float)(-K.gauss_d1 * e_x);
      e_x = K.gd2 * e_x;
      if (e_x > 1 || e_x < 0 || e_x != e_x) continue;
      e_x = (float)(e_x * K.gauss_d1);
      float cpg[3][6], xcpg[6];
#pragma unroll
      for (int r = 0; r < 3; ++r) for (int j = 0; j < 6; ++j) cpg[r][j] = (ci[r][0] * pg[0][j] + ci[r][1] * pg[1][j]) + ci[r][2] * pg[2][j];
#pragma unroll
      for (int j = 0; j < 6; ++j) { xcpg[j] = (x4[0] * cpg[0][j] + x4[1] * cpg[1][j]) + x4[2] * cpg[2][j]; acc[1 + j] += (double)(e_x * xcpg[j]); }
      if (compute_hessian) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const float pgcpg = (pg[0][j] * cpg[0][i] + pg[1][j] * cpg[1][i]) + pg[2][j] * cpg[2][i];   // (point_gradient^T C^-1 point_gradient)(j, i)
            float v = 0.0f;
            if (i >= 3 && j >= 3) { const float* h3 = hv[hsel[i - 3][j - 3]]; v = (xc[0] * h3[0] + xc[1] * h3[1]) + xc[2] * h3[2]; }
            acc[7 + 6 * i + j] += (double)(e_x * (-K.gd2 * xcpg[i] * xcpg[j] + v + pgcpg));
          }
      }
      acc[0] += (double)score_inc;
    }
  }
  ndt_block_reduce<43>(acc, part, ticket, out43);
}
// computeHessian (:540-609) with updateHessian (:613-644) and the DOUBLE computePointDerivatives (:443-480): run after a More-Thuente search that moved (:927-928).
// Everything in double; the angular tables are those of the last computeAngleDerivatives (the vector of the last derivative evaluation).
struct NdtConstD { double j_ang[8][3]; double h_ang[15][3]; double gd1, gd2; };
template <int NB>
__global__ __launch_bounds__(256) void k_ndt_hessian(const float4* src, NdtMat M, int n, float leaf, int min_pts, VxGrid g, const int* grid, const int* leaf_n, const double* mean,
                                                     const double* icov, NdtConstD K, double* part, int* ticket, double* out36) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  double acc[36];
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = 0.0;
  if (idx < n) {
    const float4 xi = src[idx];
    const float4 xt = ndt_xf(M, xi);
    const double x[3] = {(double)xi.x, (double)xi.y, (double)xi.z};
    auto dot3 = [](const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; };
    double pg[3][6] = {{1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 1, 0, 0, 0}};
    pg[1][3] = dot3(x, K.j_ang[0]); pg[2][3] = dot3(x, K.j_ang[1]); pg[0][4] = dot3(x, K.j_ang[2]); pg[1][4] = dot3(x, K.j_ang[3]); pg[2][4] = dot3(x, K.j_ang[4]);
    pg[0][5] = dot3(x, K.j_ang[5]); pg[1][5] = dot3(x, K.j_ang[6]); pg[2][5] = dot3(x, K.j_ang[7]);
    double hv[6][3];
    hv[0][0] = 0.0; hv[0][1] = dot3(x, K.h_ang[0]); hv[0][2] = dot3(x, K.h_ang[1]);
    hv[1][0] = 0.0; hv[1][1] = dot3(x, K.h_ang[2]); hv[1][2] = dot3(x, K.h_ang[3]);
    hv[2][0] = 0.0; hv[2][1] = dot3(x, K.h_ang[4]); hv[2][2] = dot3(x, K.h_ang[5]);
#pragma unroll
    for (int r = 0; r < 3; ++r) { hv[3][r] = dot3(x, K.h_ang[6 + r]); hv[4][r] = dot3(x, K.h_ang[9 + r]); hv[5][r] = dot3(x, K.h_ang[12 + r]); }
    const int hsel[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    const int ijk[3] = {(int)floorf(xt.x / leaf), (int)floorf(xt.y / leaf), (int)floorf(xt.z / leaf)};
    int lis[NB]; bool use[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      bool in = true;
#pragma unroll
      for (int a = 0; a < 3; ++a) in = in & (g.min_b[a] - ijk[a] <= ndt_disp<NB>(nb, a)) & (g.max_b[a] - ijk[a] >= ndt_disp<NB>(nb, a));
      const int key = (ijk[0] + ndt_disp<NB>(nb, 0) - g.min_b[0]) * g.mul[0] + (ijk[1] + ndt_disp<NB>(nb, 1) - g.min_b[1]) * g.mul[1] + (ijk[2] + ndt_disp<NB>(nb, 2) - g.min_b[2]) * g.mul[2];
      const int l0 = grid[in ? key : 0];
      lis[nb] = in ? l0 : -1;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { const int cn = leaf_n[lis[nb] >= 0 ? lis[nb] : 0]; use[nb] = lis[nb] >= 0 && cn >= min_pts; }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (!use[nb]) continue;
      const int li = lis[nb];
      const double* mu = mean + 3 * (size_t)li; const double* ci = icov + 9 * (size_t)li;
      const double xd[3] = {(double)xt.x - mu[0], (double)xt.y - mu[1], (double)xt.z - mu[2]};
      double cxv[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) cxv[r] = (ci[3 * r] * xd[0] + ci[3 * r + 1] * xd[1]) + ci[3 * r + 2] * xd[2];
      double e_x = K.gd2 * exp(-K.gd2 * dot3(xd, cxv) / 2);                                         // :621
      if (e_x > 1 || e_x < 0 || e_x != e_x) continue;                                               // :624
      e_x *= K.gd1;                                                                                 // :628
      double cpg[6][3], xg[6];   // c_inv * point_gradient.col(j), x_trans . (that)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int r = 0; r < 3; ++r) cpg[j][r] = (ci[3 * r] * pg[0][j] + ci[3 * r + 1] * pg[1][j]) + ci[3 * r + 2] * pg[2][j];
        xg[j] = dot3(xd, cpg[j]);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          double xh = 0.0;   // x_trans . (c_inv * point_hessian.block<3, 1>(3 i, j))
          if (i >= 3 && j >= 3) {
            const double* h3 = hv[hsel[i - 3][j - 3]];
            double cph[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) cph[r] = (ci[3 * r] * h3[0] + ci[3 * r + 1] * h3[1]) + ci[3 * r + 2] * h3[2];
            xh = dot3(xd, cph);
          }
          const double pgj[3] = {pg[0][j], pg[1][j], pg[2][j]};
          acc[6 * i + j] += e_x * (-K.gd2 * xg[i] * xg[j] + xh + dot3(pgj, cpg[i]));                 // :638-640
        }
    }
  }
  ndt_block_reduce<36>(acc, part, ticket, out36);
}
// pcl::Registration::getFitnessScore (align.cpp:30): squared distance of every transformed source point to its nearest target point — float differences, squares summed in
// order (the L2 of the reference's exact kd-tree search; the minimum over all targets is that search's result).  Targets are split over blockIdx.y and staged through LDS;
// the minimum of non-negative floats is the minimum of their bit patterns (atomicMin on unsigned; best[] starts at FLT_MAX).  The host sums best[] in index order.
__global__ __launch_bounds__(256) void k_ndt_fitness(const float4* src, int n_src, NdtMat M, const float4* tgt, int n_tgt, int tgt_per_y, unsigned* best) {
  __shared__ float4 tile[1024];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float4 p = ndt_xf(M, src[i < n_src ? i : 0]);
  const int t0 = blockIdx.y * tgt_per_y, t1 = min(n_tgt, t0 + tgt_per_y);
  float b = 3.402823466e+38f;
  for (int c0 = t0; c0 < t1; c0 += 1024) {
    const int m = min(1024, t1 - c0);
    __syncthreads();
    for (int k = threadIdx.x; k < m; k += 256) tile[k] = tgt[c0 + k];
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < m; ++k) {
      const float4 q = tile[k];
      const float d0 = p.x - q.x, d1 = p.y - q.y, d2 = p.z - q.z;
      const float d = (d0 * d0 + d1 * d1) + d2 * d2;
      b = d < b ? d : b;
    }
  }
  if (i < n_src && t1 > t0) atomicMin(&best[i], __float_as_uint(b));
}
// getNeighborhoodAtPoint(relative_coordinates, ...) (:378-408): one thread per (query, displacement); ids[q][r] = leaf or -1
__global__ __launch_bounds__(256) void k_vx_lookup_rel(const float4* q, int nq, float leaf, int min_pts, VxGrid g, const int* grid, const int* leaf_n, int n_rel, const int* rel3, int* ids) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)nq * n_rel) return;
  const int i = (int)(e / n_rel), r = (int)(e % n_rel);
  const float4 p = q[i];
  const int ijk[3] = {(int)floorf(p.x / leaf), (int)floorf(p.y / leaf), (int)floorf(p.z / leaf)};   // :383-385
  const int d[3] = {rel3[3 * r], rel3[3 * r + 1], rel3[3 * r + 2]};
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) ok = ok & (g.min_b[a] - ijk[a] <= d[a]) & (g.max_b[a] - ijk[a] >= d[a]);   // :386-387, 396
  int id = -1;
  if (ok) {
    const int li = grid[(ijk[0] + d[0] - g.min_b[0]) * g.mul[0] + (ijk[1] + d[1] - g.min_b[1]) * g.mul[1] + (ijk[2] + d[2] - g.min_b[2]) * g.mul[2]];   // :398
    if (li >= 0 && leaf_n[li] >= min_pts) id = li;                                                                                                       // :399
  }
  ids[e] = id;
}
// surfel map extraction: one WAVEFRONT per leaf (tens to hundreds of points each).  Round 3 had one thread per leaf walking its points three times with a dependent
// id -> point gather per point (1.5 ms for the 410 k-point map cloud, more than half of lvx_data_association); now the 64 lanes stride over the leaf's points, the inlier
// sums are reduced over the lanes in a fixed order (planes agree with the serial restatement to rounding, counts and boxes exactly) and lane 0 does the 3 x 3 work.
struct SurfelPlaneDev { double p4[4], Pi[3], bmin[3], bmax[3]; int leaf, n_points, n_inliers, plane_type; };
__device__ __forceinline__ double wave_sum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__global__ __launch_bounds__(256) void k_surfel_extract(const float4* __restrict__ p, const unsigned* counts, const unsigned* offs, const int* __restrict__ sorted_ids, int nl, const int* leaf_n, const double* mean,
                                                        const double* evecs, const double* evals, double p_lambda, double thr, int min_leaf, int min_inl, SurfelPlaneDev* out, int* flag, const VxInfo* nl_d) {
  const int lane = threadIdx.x & 63;
  const int li = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (nl_d) nl = min(nl, nl_d->n_leaves);   // launched over a capacity: the leaf count is still on the device (lvx_data_association without host stops)
  if (li >= nl) return;
  if (lane == 0) flag[li] = 0;
  const int n = leaf_n[li];
  if (n < min_leaf) return;
  const double* ev = evals + 3 * (size_t)li;
  int i0 = 0, i1 = 1, i2 = 2;   // descending by value (Eigen::sort_vec)
  if (ev[i1] > ev[i0]) { const int t = i0; i0 = i1; i1 = t; }
  if (ev[i2] > ev[i1]) { const int t = i1; i1 = i2; i2 = t; if (ev[i1] > ev[i0]) { const int u = i0; i0 = i1; i1 = u; } }
  const double pl = 2.0 * (ev[i1] - ev[i2]) / (ev[i2] + ev[i1] + ev[i0]);
  if (pl < p_lambda) return;
  const double* V = evecs + 9 * (size_t)li;
  double nrm[3] = {V[0 + i2], V[3 + i2], V[6 + i2]};
  const double an[3] = {fabs(nrm[0]), fabs(nrm[1]), fabs(nrm[2])};
  int t0 = 0, t1 = 1, t2 = 2;
  if (an[t1] > an[t0]) { const int t = t0; t0 = t1; t1 = t; }
  if (an[t2] > an[t1]) { const int t = t1; t1 = t2; t2 = t; if (an[t1] > an[t0]) { const int u = t0; t0 = t1; t1 = u; } }
  const int o = (int)offs[li]; int cnt = (int)counts[li];
  double d = -(nrm[0] * mean[3 * (size_t)li] + nrm[1] * mean[3 * (size_t)li + 1] + nrm[2] * mean[3 * (size_t)li + 2]);
  int nin = 0;
  float bmin[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f}, bmax[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
  for (int pass = 0; pass < 2; ++pass) {
    double sm[3] = {0, 0, 0}, cc[6] = {0, 0, 0, 0, 0, 0};
    int my = 0;
    // a lane's points k = lane, lane + 64, ... in that order (the sums keep their order), EX_U of them per trip: their ids were fetched with the previous trip's points,
    // their points leave together — one memory round trip per EX_U points of a lane.  (One point per trip was two dependent round trips each: the 2 298-point leaf of the
    // DataAssociation map cloud walked 36 of them per pass, 62-77 us for the launch.)
    constexpr int EX_U = 8;
    int idn[EX_U];
#pragma unroll
    for (int u = 0; u < EX_U; ++u) { const int kk = lane + 64 * u; idn[u] = kk < cnt ? sorted_ids[o + kk] : -1; }
    for (int k0 = lane; k0 < cnt; k0 += 64 * EX_U) {
      float4 qq[EX_U]; int idc[EX_U];
#pragma unroll
      for (int u = 0; u < EX_U; ++u) { idc[u] = idn[u]; qq[u] = p[max(idc[u], 0)]; }
#pragma unroll
      for (int u = 0; u < EX_U; ++u) { const int kk = k0 + 64 * (EX_U + u); idn[u] = kk < cnt ? sorted_ids[o + kk] : -1; }
#pragma unroll
      for (int u = 0; u < EX_U; ++u) {
        if (idc[u] < 0) continue;
        const float4 q = qq[u];
        const double x[3] = {q.x, q.y, q.z};
        if (pass == 0) {
          bmin[0] = fminf(bmin[0], q.x); bmin[1] = fminf(bmin[1], q.y); bmin[2] = fminf(bmin[2], q.z);
          bmax[0] = fmaxf(bmax[0], q.x); bmax[1] = fmaxf(bmax[1], q.y); bmax[2] = fmaxf(bmax[2], q.z);
        }
        if (!(fabs(nrm[0] * x[0] + nrm[1] * x[1] + nrm[2] * x[2] + d) < thr)) continue;
        ++my;
        if (pass == 0) {
          sm[0] += x[0]; sm[1] += x[1]; sm[2] += x[2];
          cc[0] += x[0] * x[0]; cc[1] += x[0] * x[1]; cc[2] += x[0] * x[2]; cc[3] += x[1] * x[1]; cc[4] += x[1] * x[2]; cc[5] += x[2] * x[2];
        }
      }
    }
    for (int s = 32; s > 0; s >>= 1) my += __shfl_xor(my, s);
    nin = my;
    if (pass == 1 || nin < 3) break;
    for (int a = 0; a < 3; ++a) sm[a] = wave_sum(sm[a]);
    for (int a = 0; a < 6; ++a) cc[a] = wave_sum(cc[a]);
    const double mu[3] = {sm[0] / nin, sm[1] / nin, sm[2] / nin};
    const double cs[9] = {cc[0], cc[1], cc[2], cc[1], cc[3], cc[4], cc[2], cc[4], cc[5]};
    double C[9];
    for (int a = 0; a < 3; ++a) for (int bb = 0; bb < 3; ++bb) C[3 * a + bb] = cs[3 * a + bb] / nin - mu[a] * mu[bb];
    // The refit needs ONE eigenvector of C, the smallest one's, and the leaf's own normal is within a few degrees of it: Rayleigh-quotient iteration from there
    // (y = adj(C - lambda I) n by cross products of the rows: no division, cubic convergence — three rounds reach the last bit), checked to be the SMALLEST eigenvalue
    // through the characteristic polynomial's other two roots; anything else (a leaf whose inliers turn the plane over) takes the full Jacobi solve as before.
    // (Every lane computes the same: the sums are wave-uniform after the butterfly.  The full solve was what the launch cost — 4 072 wave-level solves on 1 024 SIMDs:
    // 56 of 76 us with the sweeps run until the off-diagonal squares underflow as the oracle's loop does, 28 of 48 us stopping at 1e-18 of the diagonal.)
    bool rq_ok = false;
    {
      double nv[3] = {nrm[0], nrm[1], nrm[2]}, lam = 0.0;
      for (int it = 0; it < 4; ++it) {
        const double Cn[3] = {C[0] * nv[0] + C[1] * nv[1] + C[2] * nv[2], C[3] * nv[0] + C[4] * nv[1] + C[5] * nv[2], C[6] * nv[0] + C[7] * nv[1] + C[8] * nv[2]};
        lam = nv[0] * Cn[0] + nv[1] * Cn[1] + nv[2] * Cn[2];
        const double r0[3] = {C[0] - lam, C[1], C[2]}, r1[3] = {C[3], C[4] - lam, C[5]}, r2[3] = {C[6], C[7], C[8] - lam};
        const double a0[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};   // r1 x r2
        const double a1[3] = {r2[1] * r0[2] - r2[2] * r0[1], r2[2] * r0[0] - r2[0] * r0[2], r2[0] * r0[1] - r2[1] * r0[0]};   // r2 x r0
        const double a2[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};   // r0 x r1
        double y[3];
        for (int a = 0; a < 3; ++a) y[a] = a0[a] * nv[0] + a1[a] * nv[1] + a2[a] * nv[2];
        const double yy = y[0] * y[0] + y[1] * y[1] + y[2] * y[2];
        if (!(yy > 1e-290)) break;                       // C - lambda I is singular to the last bit: nv is the eigenvector
        const double sc = rsqrt(yy) * ((y[0] * nv[0] + y[1] * nv[1] + y[2] * nv[2]) < 0.0 ? -1.0 : 1.0);
        for (int a = 0; a < 3; ++a) nv[a] = y[a] * sc;
      }
      // is lambda the smallest eigenvalue?  the other two are the roots of x^2 - s x + p with s = tr C - lambda, p = (sum of the principal 2 x 2 minors) - lambda s
      const double tr = C[0] + C[4] + C[8], s2 = tr - lam;
      const double mn = (C[0] * C[4] - C[1] * C[3]) + (C[0] * C[8] - C[2] * C[6]) + (C[4] * C[8] - C[5] * C[7]);
      const double p2 = mn - lam * s2, disc = s2 * s2 - 4.0 * p2;
      const double small_other = 0.5 * (s2 - sqrt(disc > 0.0 ? disc : 0.0));
      const double nn = nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2];
      if (fabs(nn - 1.0) < 1e-12 && lam < small_other - 1e-9 * fabs(tr)) { nrm[0] = nv[0]; nrm[1] = nv[1]; nrm[2] = nv[2]; rq_ok = true; }
    }
    if (!rq_ok) {
      double e2[3], V2[9];
      vx_eig3<true>(C, e2, V2);
      nrm[0] = V2[0]; nrm[1] = V2[3]; nrm[2] = V2[6];
    }
    d = -(nrm[0] * mu[0] + nrm[1] * mu[1] + nrm[2] * mu[2]);
  }
  if (nin < min_inl) return;
  if (d > 0 || (d == 0 && (nrm[0] < 0 || (nrm[0] == 0 && (nrm[1] < 0 || (nrm[1] == 0 && nrm[2] < 0)))))) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; d = -d; }
  for (int s = 32; s > 0; s >>= 1) for (int a = 0; a < 3; ++a) { bmin[a] = fminf(bmin[a], __shfl_xor(bmin[a], s)); bmax[a] = fmaxf(bmax[a], __shfl_xor(bmax[a], s)); }
  if (lane != 0) return;
  SurfelPlaneDev& P = out[li];
  for (int a = 0; a < 3; ++a) { P.p4[a] = nrm[a]; P.Pi[a] = -d * nrm[a]; P.bmin[a] = bmin[a]; P.bmax[a] = bmax[a]; }
  P.p4[3] = d; P.leaf = li; P.n_points = n; P.n_inliers = nin; P.plane_type = t2;
  flag[li] = 1;
}
// accepted planes in leaf (= voxel key = std::map) order, compacted on the device: records first, then — behind them, 16-byte aligned — the association's plane table
// [p4 (4 P) | box min (3 P) | box max (3 P)]; one workgroup, ballot ranks + a running offset (the host used to fetch every leaf's record and flag, compact, rebuild the
// table and upload it again)
__global__ __launch_bounds__(1024) void k_surfel_compact(const SurfelPlaneDev* __restrict__ all, const int* __restrict__ flag, int nl, SurfelPlaneDev* recs, int* count, const VxInfo* nl_d) {
  __shared__ int wsum[16];
  __shared__ int running;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (nl_d) nl = min(nl, nl_d->n_leaves);
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  for (int l0 = 0; l0 < nl; l0 += 1024) {
    const int li = l0 + threadIdx.x;
    const bool keep = li < nl && flag[li] != 0;
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int before = running;
    for (int k = 0; k < wv; ++k) before += wsum[k];
    if (keep) recs[before + __popcll(m & ((1ull << lane) - 1ull))] = all[li];
    __syncthreads();
    if (threadIdx.x == 0) { int t = running; for (int k = 0; k < 16; ++k) t += wsum[k]; running = t; }
    __syncthreads();
  }
  const int P = running;
  if (threadIdx.x == 0) *count = P;
  __threadfence_block();
  __syncthreads();
  double* pl = (double*)((char*)recs + (((size_t)P * sizeof(SurfelPlaneDev) + 15) & ~(size_t)15));
  for (int k = threadIdx.x; k < P; k += 1024) {
    const SurfelPlaneDev r = recs[k];
    for (int a = 0; a < 4; ++a) pl[4 * (size_t)k + a] = r.p4[a];
    for (int a = 0; a < 3; ++a) { pl[4 * (size_t)P + 3 * (size_t)k + a] = r.bmin[a]; pl[7 * (size_t)P + 3 * (size_t)k + a] = r.bmax[a]; }
  }
}
// The same in up to 2 048 workgroups of 1 024 leaves (round 5b; one workgroup walked 5 600 flags in six trips of three barriers and copied every record itself, 30 us of
// a DataAssociation round, and a 4 M-point map has 556 k leaves): a thread takes four consecutive flags, the workgroup lists its accepted leaves in LDS and publishes its
// count as (epoch | count) — the epoch changes with every launch, the words are never cleared — and every workgroup sums ALL the published counts: those before it place
// its records, the total places the plane table behind the records.  The records are copied word by word by all threads (coalesced; a few registers per thread: every
// workgroup of the launch must be resident for the exchange of counts, 8 per CU x 256 CUs = SC_MAXB).
#define SC_MAXB 2048
__global__ __launch_bounds__(256) void k_surfel_compact_mb(const SurfelPlaneDev* __restrict__ all, const int* __restrict__ flag, int nl, SurfelPlaneDev* recs, int* count, const VxInfo* nl_d,
                                                           unsigned long long* pub, unsigned epoch) {
  static_assert(sizeof(SurfelPlaneDev) == 15 * 8, "record = 13 doubles + 4 ints");
  constexpr int RW = 15;
  __shared__ int idx[1024];
  __shared__ int wsum[4];
  __shared__ int s_before, s_total;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x, nb = gridDim.x;
  if (nl_d) nl = min(nl, nl_d->n_leaves);
  const int l0 = b * 1024 + 4 * tid;
  int f[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) f[u] = l0 + u < nl ? flag[l0 + u] : 0;
  int mine = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) mine += f[u] != 0 ? 1 : 0;
  int incl = mine;
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  if (lane == 63) wsum[wv] = incl;
  if (tid == 0) { s_before = 0; s_total = 0; }
  __syncthreads();
  int off = incl - mine, tot = 0;
  for (int k = 0; k < 4; ++k) { if (k < wv) off += wsum[k]; tot += wsum[k]; }
  if (tid == 0) __hip_atomic_store(&pub[b], ((unsigned long long)epoch << 32) | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int u = 0; u < 4; ++u) if (f[u] != 0) idx[off++] = l0 + u;
  int sb = 0, st = 0;
  for (int j = tid; j < nb; j += 256) {
    unsigned long long v;
    do { v = __hip_atomic_load(&pub[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned)(v >> 32) != epoch);
    const int cnt = (int)(unsigned)(v & 0xffffffffull);
    st += cnt; if (j < b) sb += cnt;
  }
  if (st) atomicAdd(&s_total, st);
  if (sb) atomicAdd(&s_before, sb);
  __syncthreads();
  const int P = s_total, k0 = s_before;
  if (b == 0 && tid == 0) *count = P;
  double* pl = (double*)((char*)recs + (((size_t)P * sizeof(SurfelPlaneDev) + 15) & ~(size_t)15));
  const double* src = (const double*)all; double* dst = (double*)recs;
  for (int e = tid; e < tot * RW; e += 256) {
    const int r = e / RW, w = e - RW * r, k = k0 + r;
    const double v = src[(size_t)idx[r] * RW + w];
    dst[(size_t)k * RW + w] = v;
    if (w < 4) pl[4 * (size_t)k + w] = v;                                   // p4
    else if (w >= 7 && w < 10) pl[4 * (size_t)P + 3 * (size_t)k + (w - 7)] = v;    // box min
    else if (w >= 10 && w < 13) pl[7 * (size_t)P + 3 * (size_t)k + (w - 10)] = v;  // box max
  }
}
// K = 7: getNeighborhoodAtPoint7 (:423-438), K = 1: getNeighborhoodAtPoint1 (:440-446, the cell of the point only)
template <int K>
__global__ void k_vx_lookup(const float4* q, int nq, float leaf, int min_pts, VxGrid g, const int* grid, const int* leaf_n, int* ids7) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const float4 p = q[i];
  const int ijk[3] = {(int)floorf(p.x / leaf), (int)floorf(p.y / leaf), (int)floorf(p.z / leaf)};   // :383-385
  const int disp[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  // all K cell probes in flight, then all K leaf counts: two memory round trips per query.  (Probe by probe — load the cell, wait, load its leaf's count, wait, store —
  // the compiler kept the 2 K loads of a query strictly one behind the other.)
  bool in[K]; int li[K], cn[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) ok = ok & (g.min_b[a] - ijk[a] <= disp[k][a]) & (g.max_b[a] - ijk[a] >= disp[k][a]);
    in[k] = ok;
    const int key = (ijk[0] + disp[k][0] - g.min_b[0]) * g.mul[0] + (ijk[1] + disp[k][1] - g.min_b[1]) * g.mul[1] + (ijk[2] + disp[k][2] - g.min_b[2]) * g.mul[2];
    li[k] = grid[ok ? key : 0];                  // (an unconditional load at a clamped address: selected afterwards)
  }
#pragma unroll
  for (int k = 0; k < K; ++k) { li[k] = in[k] ? li[k] : -1; cn[k] = leaf_n[li[k] >= 0 ? li[k] : 0]; }
#pragma unroll
  for (int k = 0; k < K; ++k) ids7[K * (size_t)i + k] = (li[k] >= 0 && cn[k] >= min_pts) ? li[k] : -1;
}

// ------------------------------------------------------------------------------------------------------------------------
// scan -> surfel association (getAssociation + associateScanToSurfel, surfel_association.cpp:111-138,296-331), S scans per launch.
// Phase 1 (k_assoc_hits): thread = scan point; it meets the surfels listed for its cell of a uniform grid over the surfels' boxes (below); a
//   hit (not NaN, strictly inside the AABB, |n.p + d| <= radius — double arithmetic on the float coordinates, as the reference) sets the
//   point's bit in the (scan, plane, ring) bitmask and bumps that ring's hit count.  16 B read per point, no write for the ~all misses.
//   (All pairs through an LDS plane table — 57.6 M exact tests per scan at P = 2000 — ran at the FP64 VALU limit: 16 us per scan.)
// Phase 2 (k_assoc_select): thread = (scan, plane, ring) with >= 2 sel hits: the evenly spaced ranks step (s + 1) - 1 of the ring's hit
//   list are found by popcount over the bitmask words; conflicts resolve as the reference's SERIAL plane loop (highest plane id wins:
//   atomicMax; the reference's OpenMP loop races here).
// The previous kernel walked a ring's W points twice in one thread per (plane, ring): 512 wavefronts, 0.37 ms per scan.
// ------------------------------------------------------------------------------------------------------------------------
#define SA_WMAX 4096
// A uniform grid over the surfels' AABBs (64 x 64 x 8 cells) lists, per cell, the surfels whose AABB reaches it; a point only meets the surfels of its
// own cell.  The cell index is the same monotone function of a coordinate for boxes and points, so a point strictly inside a box always finds that box in
// its cell's list: the candidate set is a superset of the reference's hits and the exact tests decide — results are identical to the all-pairs loop.
#define SA_GX 64
#define SA_GY 64
#define SA_GZ 8
#define SA_CELLS (SA_GX * SA_GY * SA_GZ)
struct AssocGrid { double g0[3], inv[3]; };
__device__ __forceinline__ int sa_cell(double v, double g0, double inv, int n) { const double f = floor((v - g0) * inv); return f < 0.0 ? 0 : (f >= (double)n ? n - 1 : (int)f); }
// one workgroup: bounds of all boxes -> grid geometry
__global__ __launch_bounds__(256) void k_assoc_grid_geom(int P, const double* planes10, AssocGrid* g, const int* P_d, int* cell_cnt) {
  __shared__ double lo[3][256], hi[3][256];
  for (int e = threadIdx.x; e < SA_CELLS; e += 256) cell_cnt[e] = 0;   // (k_assoc_grid_fill counts into them next)
  if (P_d) { const int Pr = *P_d; planes10 = (const double*)((const char*)planes10 + (((size_t)Pr * sizeof(SurfelPlaneDev) + 15) & ~(size_t)15)); P = min(Pr, P); }   // see k_assoc_grid_fill
  double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
  const size_t Ps = P_d ? (size_t)*P_d : (size_t)P;
  for (int k = threadIdx.x; k < P; k += 256)
    for (int a = 0; a < 3; ++a) { l[a] = fmin(l[a], planes10[4 * Ps + 3 * (size_t)k + a]); h[a] = fmax(h[a], planes10[7 * Ps + 3 * (size_t)k + a]); }
  for (int a = 0; a < 3; ++a) { lo[a][threadIdx.x] = l[a]; hi[a][threadIdx.x] = h[a]; }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) for (int a = 0; a < 3; ++a) { lo[a][threadIdx.x] = fmin(lo[a][threadIdx.x], lo[a][threadIdx.x + o]); hi[a][threadIdx.x] = fmax(hi[a][threadIdx.x], hi[a][threadIdx.x + o]); } __syncthreads(); }
  if (threadIdx.x < 3) {
    const int a = threadIdx.x, n = a == 0 ? SA_GX : (a == 1 ? SA_GY : SA_GZ);
    const double ext = hi[a][0] - lo[a][0];
    g->g0[a] = lo[a][0]; g->inv[a] = ext > 0.0 ? (double)n / ext : 0.0;
  }
}
// mode 0: count the cells every box reaches; mode 1: write the box into its cells' lists (cursor = running offsets; entries past list_cap are dropped — the host finds the
// true total behind the cursors and repeats the call with a larger list).  P_d != nullptr: the plane count is still on the device, P is the capacity the launch covers and
// planes10 the RECORD array k_surfel_compact filled — its plane table sits right behind the *P_d records, strided by *P_d.
__global__ void k_assoc_grid_fill(int P, const double* planes10, const AssocGrid* gp, int* cell_cnt, int* cursor, int* list, int mode, double* aos, const int* P_d, int list_cap,
                                  int* mirror = nullptr, const int* pose_valid = nullptr) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (mirror && k == 0) { mirror[0] = *pose_valid; mirror[1] = *P_d; mirror[2] = cursor[SA_CELLS]; }   // what the host of the one-stop chain looks at (pinned memory); cursor[SA_CELLS] = the scan's true total
  size_t Ps = (size_t)P;   // stride of the table
  if (P_d) { const int Pr = *P_d; planes10 = (const double*)((const char*)planes10 + (((size_t)Pr * sizeof(SurfelPlaneDev) + 15) & ~(size_t)15)); Ps = (size_t)Pr; P = min(Pr, P); }
  if (k >= P) return;
  const AssocGrid g = *gp;
  if (mode == 0 && aos) {   // one 64-byte record per surfel for the hit kernel: box min | box max as FLOATS | plane (the caller's table is three arrays: ten 8-byte loads from three places per candidate)
    // a scan point is a float: x > lo  <=>  x > (lo rounded DOWN to a float), x < hi  <=>  x < (hi rounded UP) — no float lies between a bound and its directed rounding,
    // so the float comparison decides exactly what the reference's double comparison decides (the bounds of a surfel map are floats to begin with)
    float* bf = (float*)(aos + 8 * (size_t)k);
    for (int a = 0; a < 3; ++a) {
      const double lo = planes10[4 * Ps + 3 * (size_t)k + a], hi = planes10[7 * Ps + 3 * (size_t)k + a];
      float fl = (float)lo, fh = (float)hi;
      if ((double)fl > lo) fl = nextafterf(fl, -INFINITY);
      if ((double)fh < hi) fh = nextafterf(fh, INFINITY);
      bf[a] = fl; bf[3 + a] = fh;
    }
    bf[6] = 0.f; bf[7] = 0.f;
    for (int a = 0; a < 4; ++a) aos[8 * (size_t)k + 4 + a] = planes10[4 * (size_t)k + a];
  }
  int c0[3], c1[3];
  const int nn[3] = {SA_GX, SA_GY, SA_GZ};
  for (int a = 0; a < 3; ++a) {
    const double lo = planes10[4 * Ps + 3 * (size_t)k + a], hi = planes10[7 * Ps + 3 * (size_t)k + a];
    if (!(lo < hi)) return;                          // an empty box holds no point
    c0[a] = sa_cell(lo, g.g0[a], g.inv[a], nn[a]); c1[a] = sa_cell(hi, g.g0[a], g.inv[a], nn[a]);
  }
  for (int z = c0[2]; z <= c1[2]; ++z) for (int y = c0[1]; y <= c1[1]; ++y) for (int x = c0[0]; x <= c1[0]; ++x) {
    const int cell = (z * SA_GY + y) * SA_GX + x;
    if (mode == 0) atomicAdd(&cell_cnt[cell], 1);
    else { const int pos = atomicAdd(&cursor[cell], 1); if (pos < list_cap) list[pos] = k; }
  }
}
// exclusive scan of the cell counts (one workgroup: each of its 16 wavefronts owns 2048 consecutive cells, read 64 at a time), total behind the last cell;
// cursor = copy of the offsets.  The offsets the hit kernel walks are clamped to list_cap (a list that turned out too small is never read past its end); the true
// total goes behind the cursors, where the host looks.
__global__ __launch_bounds__(1024) void k_assoc_grid_scan(const int* cnt, int* off, int* cursor, int list_cap) {
  __shared__ int wtot[16];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, base = wv * (SA_CELLS / 16);
  int v[SA_CELLS / 1024], tot = 0;
#pragma unroll
  for (int j = 0; j < SA_CELLS / 1024; ++j) { v[j] = cnt[base + 64 * j + lane]; tot += v[j]; }
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
  if (lane == 0) wtot[wv] = tot;
  __syncthreads();
  int running = 0;
  for (int k = 0; k < wv; ++k) running += wtot[k];
#pragma unroll
  for (int j = 0; j < SA_CELLS / 1024; ++j) {
    int inc = v[j];
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    const int ex = running + inc - v[j];
    off[base + 64 * j + lane] = min(ex, list_cap); cursor[base + 64 * j + lane] = ex;
    running += __shfl(inc, 63);
  }
  if (threadIdx.x == 1023) { off[SA_CELLS] = min(running, list_cap); cursor[SA_CELLS] = running; }
}
// all pairs: thread = scan point, the plane table of the block's chunk (256 planes) in LDS, read as broadcasts.  57.6 M exact tests per scan at P = 2000
// (FP64 VALU bound, 36 us) — cheaper than building the grid when only one or two scans are associated in a call.
#define SA_PC 256
// A hit = the point's bit in the (scan, plane, ring) mask + the mask word's bit in the ring's 64-bit occupancy word (round 5b; a hit count per ring until then).  Hits of one surfel come in runs along a ring (a box spans tens of consecutive columns): up
// to 32 lanes of a wavefront would OR the same mask word and bump the same counter, and same-address atomics of one instruction are served one after the other (round 4: the
// two atomics were 150 of k_assoc_hits' 200 us).  The lanes of a run — same (plane, ring, mask word); consecutive lanes are consecutive columns — hand their bits to the run's
// first lane: two ORs per run (neither returns a value: a returning add per run, to append first-hit rings to a work list, made the kernel 3.5x slower).
// Wave-uniform among the lanes that call it together.
__device__ __forceinline__ void sa_record_hit(bool hit, size_t ring, int w, int lane, int wpr, unsigned* bits, unsigned long long* occ, int oshift) {
  const size_t word = ring * wpr + (w >> 5);
  unsigned long long pend = __ballot(hit);
  while (pend) {
    const int ld = __ffsll((long long)pend) - 1;
    const size_t wd = __shfl(word, ld);
    const unsigned long long m = __ballot(hit && word == wd);        // the lanes of this run (ld among them)
    pend &= ~m;
    if (lane == ld) {
      atomicOr(&bits[word], (unsigned)((m >> ld) << (w & 31)));        // lane L holds column w + (L - ld) of the same 32-column word
      atomicOr(&occ[ring], 1ull << ((w >> 5) >> oshift));              // which of the ring's mask words hold anything (k_assoc_select reads only those)
    }
  }
}
__global__ __launch_bounds__(256) void k_assoc_hits_allpairs(const float4* __restrict__ scans, int H, int W, int P, const double* __restrict__ planes10, double radius,
                                                             unsigned* bits, unsigned long long* occ, int wpr, int oshift) {
  __shared__ double pl[10][SA_PC];
  const int p0 = blockIdx.y * SA_PC, np = min(SA_PC, P - p0), sc = blockIdx.z;
  for (int e = threadIdx.x; e < 10 * SA_PC; e += 256) {
    const int f = e / SA_PC, k = e % SA_PC;
    if (k < np) pl[f][k] = f < 4 ? planes10[4 * (size_t)(p0 + k) + f] : (f < 7 ? planes10[4 * (size_t)P + 3 * (size_t)(p0 + k) + (f - 4)] : planes10[7 * (size_t)P + 3 * (size_t)(p0 + k) + (f - 7)]);
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const float4 q = scans[(size_t)sc * H * W + i];
  if (isnan(q.x)) return;
  const double x = q.x, y = q.y, z = q.z;
  const int h = i / W, w = i - h * W, lane = threadIdx.x & 63;
  for (int k = 0; k < np; ++k) {
    const bool inside = (x > pl[4][k]) & (x < pl[7][k]) & (y > pl[5][k]) & (y < pl[8][k]) & (z > pl[6][k]) & (z < pl[9][k]);
    if (!__ballot(inside)) continue;                                   // wave-uniform skip
    double dist = inside ? x * pl[0][k] + y * pl[1][k] + z * pl[2][k] + pl[3][k] : 1e300;
    dist = dist > 0 ? dist : -dist;
    sa_record_hit(dist <= radius, ((size_t)sc * P + p0 + k) * H + h, w, lane, wpr, bits, occ, oshift);
  }
}
__global__ __launch_bounds__(256) void k_assoc_hits(const float4* __restrict__ scans, int H, int W, int P, const double* __restrict__ aos, double radius, const AssocGrid* gp,
                                                    const int* __restrict__ off, const int* __restrict__ list, unsigned* bits, unsigned long long* occ, int wpr, int oshift, int* flags_init) {
  const int i = blockIdx.x * 256 + threadIdx.x, sc = blockIdx.y, lane = threadIdx.x & 63;
  if (i >= H * W) return;
  if (flags_init) flags_init[(size_t)sc * H * W + i] = -1;   // "no surfel": k_assoc_select (the next launch) raises it — a 7 MB fill launch per call otherwise
  const float4 q = scans[(size_t)sc * H * W + i];
  const double x = q.x, y = q.y, z = q.z;
  const AssocGrid g = *gp;
  // NaN point, or outside the bounds of all boxes: no box can hold the point (sa_cell would clamp it into a border cell).  Tests combined with & (no short circuit): with &&
  // the compiler loads x, waits, branches, loads y z, waits ... — a memory round trip per condition, here and in the candidate loop below
  const double gx = (x - g.g0[0]) * g.inv[0], gy = (y - g.g0[1]) * g.inv[1], gz = (z - g.g0[2]) * g.inv[2];
  const bool in_grid = (!isnan(q.x)) & (gx >= 0.0) & (gx <= (double)SA_GX) & (gy >= 0.0) & (gy <= (double)SA_GY) & (gz >= 0.0) & (gz <= (double)SA_GZ);
  if (!in_grid) return;
  const int cell = (sa_cell(z, g.g0[2], g.inv[2], SA_GZ) * SA_GY + sa_cell(y, g.g0[1], g.inv[1], SA_GY)) * SA_GX + sa_cell(x, g.g0[0], g.inv[0], SA_GX);
  const int h = i / W, w = i - h * W;
  // SA_U candidates per trip: their list entries are loaded together (clamped addresses, the tail masked afterwards), then their SA_U x 4 record loads — two memory round
  // trips per SA_U candidates.  One candidate per iteration was two DEPENDENT round trips each (entry, then its record), 11 iterations for the longest list of an average
  // wavefront: 205 us for 64 scans x 2 000 surfels.
  constexpr int SA_U = 2;   // (round 5b, with 64-byte records: 1 and 2 candidates per trip 17.4-17.7 Gpts/s, 4: 16.8, 8: 14.7; with the 80-byte records and a count atomic per hit, 4 had been the best)
  const int e0 = off[cell], e1 = off[cell + 1];
  for (int e = e0; e < e1; e += SA_U) {
    int k[SA_U];
#pragma unroll
    for (int u = 0; u < SA_U; ++u) k[u] = list[min(e + u, e1 - 1)];
    float4 bx[SA_U][2]; double2 pq[SA_U][2];
#pragma unroll
    for (int u = 0; u < SA_U; ++u) {
      const float4* rec = (const float4*)(aos + 8 * (size_t)k[u]);   // four 16-byte loads of one 64-byte record: box (6 floats) | plane (4 doubles)
      bx[u][0] = rec[0]; bx[u][1] = rec[1];
      pq[u][0] = ((const double2*)rec)[2]; pq[u][1] = ((const double2*)rec)[3];
    }
#pragma unroll
    for (int u = 0; u < SA_U; ++u) {
      const float lo[3] = {bx[u][0].x, bx[u][0].y, bx[u][0].z}, hi[3] = {bx[u][0].w, bx[u][1].x, bx[u][1].y};
      const double pl[4] = {pq[u][0].x, pq[u][0].y, pq[u][1].x, pq[u][1].y};
      const bool inside = (q.x > lo[0]) & (q.x < hi[0]) & (q.y > lo[1]) & (q.y < hi[1]) & (q.z > lo[2]) & (q.z < hi[2]);   // float against float: exact, see k_assoc_grid_fill
      double dist = x * pl[0] + y * pl[1] + z * pl[2] + pl[3];
      dist = dist > 0 ? dist : -dist;
      sa_record_hit((e + u < e1) & inside & (dist <= radius), ((size_t)sc * P + k[u]) * H + h, w, lane, wpr, bits, occ, oshift);
    }
  }
}
// Ring selection (surfel_association.cpp:118-139): a (scan, surfel, ring) triple with c >= 2 sel hits keeps the hits of rank step, 2 step, ... (step = c / (sel + 1)) in
// column order.  One THREAD per ring: its occupancy word says which of the ring's mask words hold hits — a surfel's box spans tens of consecutive columns, two or three
// words of 57 — and only those are read (counted, walked for the target ranks) and returned to zero with the occupancy word: the work buffer is cleared once, when it is
// allocated, not 7 MB per scan and call.
// (Rounds 4-5a: a hit COUNT per ring and a wavefront per ring with hits reading all its words — 342 MB for 57 scans x 2 500 surfels, 34 us; round 3: one thread walking
// all 57 words serially, 99 us.  A work list of the rings with hits, appended by the hit kernel, was measured and dropped: the returning add per run costs the hit kernel
// far more than this scan.)
__global__ __launch_bounds__(256) void k_assoc_select(unsigned* __restrict__ bits, unsigned long long* __restrict__ occ, int S, int H, int W, int P, int wpr, int oshift, int sel, int* flags) {
  const size_t total = (size_t)S * P * H;
  const size_t rg = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (rg >= total) return;
  const unsigned long long o = occ[rg];
  if (o == 0ull) return;
  occ[rg] = 0ull;
  unsigned* bw = bits + rg * wpr;
  const int per = 1 << oshift;
  int c = 0;
  for (unsigned long long oo = o; oo; oo &= oo - 1)
    for (int q = 0; q < per; ++q) { const int k = ((__ffsll((long long)oo) - 1) << oshift) + q; if (k < wpr) c += __popc(bw[k]); }
  if (c >= sel * 2) {
    const int h = (int)(rg % H), pid = (int)((rg / H) % P), sc = (int)(rg / ((size_t)H * P));
    int step = c / (sel + 1);
    step = step > 1 ? step : 1;
    int seen = 0, s = 0;
    for (unsigned long long oo = o; oo && s < sel; oo &= oo - 1)
      for (int q = 0; q < per; ++q) {
        const int k = ((__ffsll((long long)oo) - 1) << oshift) + q;
        if (k >= wpr) continue;
        const unsigned word = bw[k];
        const int pc = __popc(word);
        while (s < sel) {
          const int target = step * (s + 1) - 1;       // the target-th hit of the ring, in column order
          if (target >= seen + pc) break;
          unsigned t = word;
          for (int r = target - seen; r > 0; --r) t &= t - 1;
          atomicMax(&flags[(size_t)sc * H * W + (size_t)h * W + 32 * k + (__ffs(t) - 1)], pid);
          ++s;
        }
        seen += pc;
      }
  }
  for (unsigned long long oo = o; oo; oo &= oo - 1)
    for (int q = 0; q < per; ++q) { const int k = ((__ffsll((long long)oo) - 1) << oshift) + q; if (k < wpr) bw[k] = 0u; }
}
// Chronological SurfelPoint emission (surfel_association.cpp:141-158): column-major (w outer, h inner), points with a flag and a non-zero raw
// timestamp.  One workgroup per scan; order-preserving compaction by ballots + a running offset.  mode 0: count only; mode 1: write behind the scans before this one
// (their counts of the mode-0 launch, summed here: no host hop between the two launches), nothing at all when the list would not fit max_out.
struct SurfelOut { double* pt; double* pt_map; double* t; int* plane; };
__global__ __launch_bounds__(1024) void k_assoc_emit(const int* __restrict__ flags, const float4* __restrict__ scans_map, const void* __restrict__ raw_v, int H, int W, int* counts, int n_scans, int max_out, int mode, SurfelOut o) {
  struct Raw { float x, y, z, pad; float intensity; float pad2; double timestamp; };
  const Raw* raw = (const Raw*)raw_v;
  __shared__ int wsum[16];
  __shared__ int running;
  const int sc = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = H * W;
  __shared__ int s_base, s_total;
  if (threadIdx.x == 0) { running = 0; s_base = 0; s_total = 0; }
  __syncthreads();
  if (mode) {
    int mine = 0, all = 0;
    for (int k = threadIdx.x; k < n_scans; k += blockDim.x) { const int ck = counts[k]; all += ck; if (k < sc) mine += ck; }
    if (mine) atomicAdd(&s_base, mine);
    if (all) atomicAdd(&s_total, all);
    __syncthreads();
    if (s_total > max_out) return;
  }
  const int base_out = s_base;
  for (int e0 = 0; e0 < n; e0 += 1024) {
    const int e = e0 + threadIdx.x;                 // position in the emission order: e = w * H + h
    bool keep = false; int idx = 0, pid = -1;
    if (e < n) { const int w = e / H, h = e - w * H; idx = h * W + w; pid = flags[(size_t)sc * n + idx]; keep = pid != -1 && raw[(size_t)sc * n + idx].timestamp != 0.0; }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int before = running;
    for (int k = 0; k < wv; ++k) before += wsum[k];
    if (keep && mode) {
      const int pos = base_out + before + __popcll(m & ((1ull << lane) - 1ull));
      const Raw r = raw[(size_t)sc * n + idx]; const float4 q = scans_map[(size_t)sc * n + idx];
      o.pt[3 * (size_t)pos] = r.x; o.pt[3 * (size_t)pos + 1] = r.y; o.pt[3 * (size_t)pos + 2] = r.z;
      o.pt_map[3 * (size_t)pos] = q.x; o.pt_map[3 * (size_t)pos + 1] = q.y; o.pt_map[3 * (size_t)pos + 2] = q.z;
      o.t[pos] = r.timestamp; o.plane[pos] = pid;
    }
    __syncthreads();
    if (threadIdx.x == 0) { int t = running; for (int k = 0; k < 16; ++k) t += wsum[k]; running = t; }
    __syncthreads();
  }
  if (threadIdx.x == 0 && !mode) counts[sc] = running;
}

// Count and write in ONE launch, a THREAD per column of the organised scan and a workgroup per 128 columns: the emission order is column-major (w outer, h inner), so a
// thread's H points are consecutive in the list and consecutive threads read consecutive flags / raw points of every ring.  A workgroup counts its columns, publishes
// (epoch | count) in a 64-bit word — the epoch changes with every launch, the words are never cleared — sums the words of every workgroup before it in (scan, column)
// order, and writes its points behind them while max_out allows; every workgroup must be resident (SE_MAXWG).  A list that does not fit is reported through the counts
// as before; what was written of it is unspecified (never past max_out).
// (Round 4-5a: a workgroup per scan, a thread per POSITION of the emission order, one launch to count and one to write: 2 x 17 us for 57 scans x 16 x 450 — the positions
// of a wavefront lie W apart, and one CU pulled the scan's 230 KB of timestamps alone: 7 us of each launch.)
#define SE_HMAX 128
#define SE_COLS 128
#define SE_MAXWG 2048
__global__ __launch_bounds__(SE_COLS) void k_assoc_emit_fused(const int* __restrict__ flags, const float4* __restrict__ scans_map, const void* __restrict__ raw_v, int H, int W, int* counts,
                                                              unsigned long long* pub, unsigned epoch, int max_out, int write, SurfelOut o) {
  struct Raw { float x, y, z, pad; float intensity; float pad2; double timestamp; };
  const int part = blockIdx.x, parts = gridDim.x, sc = blockIdx.y, seg = sc * parts + part;
  const Raw* __restrict__ raw = (const Raw*)raw_v + (size_t)sc * H * W;
  const int* __restrict__ fl = flags + (size_t)sc * H * W;
  const float4* __restrict__ sm = scans_map + (size_t)sc * H * W;
  __shared__ int wsum[SE_COLS / 64];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int w = part * SE_COLS + tid;
  unsigned long long km[2];                // keep bits of column w, ring h
#pragma unroll
  for (int half = 0; half < 2; ++half) {   // (static register indices: a run-time index would put km into scratch memory)
    unsigned long long acc = 0ull;
    for (int h0 = 64 * half; h0 < min(H, 64 * half + 64); h0 += 16) {   // sixteen rings' flags and timestamps in flight
      int pid[16]; double ts[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { const bool in = w < W && h0 + u < H; pid[u] = in ? fl[(size_t)(h0 + u) * W + w] : -1; }
#pragma unroll
      for (int u = 0; u < 16; ++u) { const bool in = w < W && h0 + u < H; ts[u] = in ? raw[(size_t)(h0 + u) * W + w].timestamp : 0.0; }
#pragma unroll
      for (int u = 0; u < 16; ++u) if (pid[u] != -1 && ts[u] != 0.0) acc |= 1ull << ((h0 + u) & 63);
    }
    km[half] = acc;
  }
  const int cnt = __popcll(km[0]) + __popcll(km[1]);
  int incl = cnt;
  for (int o_ = 1; o_ < 64; o_ <<= 1) { const int v = __shfl_up(incl, o_); if (lane >= o_) incl += v; }
  if (lane == 63) wsum[wv] = incl;
  if (tid == 0) s_base = 0;
  __syncthreads();
  int pos = incl - cnt, tot = 0;
  for (int k = 0; k < SE_COLS / 64; ++k) { const int v = wsum[k]; if (k < wv) pos += v; tot += v; }
  if (tid == 0) {
    atomicAdd(&counts[sc], tot);           // (cleared by the host before the launch)
    __hip_atomic_store(&pub[seg], ((unsigned long long)epoch << 32) | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!write) return;
  int before = 0;
  for (int k0 = tid; k0 < seg; k0 += SE_COLS * 8) {
    unsigned long long v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = k0 + SE_COLS * u < seg ? __hip_atomic_load(&pub[k0 + SE_COLS * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)epoch << 32);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      while ((unsigned)(v[u] >> 32) != epoch) v[u] = __hip_atomic_load(&pub[k0 + SE_COLS * u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      before += (int)(unsigned)(v[u] & 0xffffffffull);
    }
  }
  for (int s = 32; s > 0; s >>= 1) before += __shfl_xor(before, s);
  if (lane == 0 && before) atomicAdd(&s_base, before);
  __syncthreads();
  pos += s_base;
  double* __restrict__ o_pt = o.pt; double* __restrict__ o_pm = o.pt_map; double* __restrict__ o_t = o.t; int* __restrict__ o_pl = o.plane;
#pragma unroll
  for (int half = 0; half < 2; ++half)
    for (unsigned long long m = km[half]; m; m &= m - 1) {
      const int h = 64 * half + __ffsll((long long)m) - 1;
      const size_t at = (size_t)h * W + w;
      if (pos < max_out) {
        const Raw r = raw[at]; const float4 q = sm[at]; const int pid = fl[at];
        o_pt[3 * (size_t)pos] = r.x; o_pt[3 * (size_t)pos + 1] = r.y; o_pt[3 * (size_t)pos + 2] = r.z;
        o_pm[3 * (size_t)pos] = q.x; o_pm[3 * (size_t)pos + 1] = q.y; o_pm[3 * (size_t)pos + 2] = q.z;
        o_t[pos] = r.timestamp; o_pl[pos] = pid;
      }
      ++pos;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// batched lidar pose evaluation + scan de-skew: HBM-streaming reuse of the spline evaluator (no Jacobians)
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool lidar_pose_dev(const double* state, int N, double t0, double dt, double t, quat* q_LtoG, v3* p_LinG) {
  const double* sl = state + 7 * (size_t)N + 16;
  const double tt = t + sl[7];
  const double tmax = t0 + (double)(N - 3) * dt;
  if (t0 > tt || tmax <= tt) return false;                         // evaluateLidarPose range test (trajectory_manager_lvi.cpp:401-402)
  const double s = (tt - t0) / dt;
  const int i0 = (int)floor(s);
  if (N < 4 || i0 < 0 || i0 > N - 4) return false;
  const SplineRef sp{t0, dt, N, state, state + 3 * (size_t)N};
  KnotRef k; k.i0 = i0; k.u = s - (double)i0;
  PoseEval e;
  if (!pose_eval<false>(sp, k, &e)) return false;
  const quat qL = load_q(sl); const v3 pL = load_v3(sl + 4);
  *q_LtoG = qmul(e.so3.q, qL);
  *p_LinG = qrot(e.so3.q, pL) + e.p;
  return true;
}
// evaluateCameraPose (trajectory_manager_lvi.cpp:430-440)
__device__ __forceinline__ bool camera_pose_dev(const double* state, int N, double t0, double dt, double t, quat* q_CtoG, v3* p_CinG) {
  const double* sc = state + 7 * (size_t)N + 24;
  const double tt = t + sc[7];
  const double tmax = t0 + (double)(N - 3) * dt;
  if (t0 > tt || tmax <= tt) return false;
  const double s = (tt - t0) / dt;
  const int i0 = (int)floor(s);
  if (N < 4 || i0 < 0 || i0 > N - 4) return false;
  const SplineRef sp{t0, dt, N, state, state + 3 * (size_t)N};
  KnotRef k; k.i0 = i0; k.u = s - (double)i0;
  PoseEval e;
  if (!pose_eval<false>(sp, k, &e)) return false;
  const quat qC = load_q(sc); const v3 pC = load_v3(sc + 4);
  *q_CtoG = qmul(e.so3.q, qC);
  *p_CinG = qrot(e.so3.q, pC) + e.p;
  return true;
}
// associateVisualPointsWithPlanes (surfel_association.cpp:161-214): thread = landmark; its reference observation is back-projected at depth 1 / rho with the
// camera pose at the view's t0, moved into the LiDAR map frame (pose of the camera at the map time, q_LtoC / t_LinC), and tested against every surfel:
// strictly inside the AABB and within 2 radius of the plane; the last (highest) matching surfel stays, as in the reference's loop.
// (the plane table goes through LDS 256 planes at a time and the box tests are combined with &: a thread walking the caller's three arrays plane by plane made a memory
// round trip per condition, 2 000 planes long)
__global__ __launch_bounds__(256) void k_landmark_assoc(const double* state, int N, double t0, double dt, CamIntr cam, const double* lm_uv, const double* lm_t0, int L, double map_time,
                                                        quat q_LtoC, v3 t_LinC, int P, const double* planes10, double radius, int* out) {
  __shared__ double tab[256 * 10];   // box min | box max | plane of 256 planes
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  int res = -1;
  bool live = false;
  v3 q = mk(0, 0, 0);
  quat q_CtoG; v3 p_CinG;
  if (l < L && camera_pose_dev(state, N, t0, dt, map_time, &q_CtoG, &p_CinG)) {
    const quat q_L0_G = qmul(q_CtoG, q_LtoC);
    const v3 t_L0_G = qrot(q_CtoG, t_LinC) + p_CinG;
    const double n2 = q_L0_G.x * q_L0_G.x + q_L0_G.y * q_L0_G.y + q_L0_G.z * q_L0_G.z + q_L0_G.w * q_L0_G.w;
    const quat q_inv = mkq(q_L0_G.w / n2, -q_L0_G.x / n2, -q_L0_G.y / n2, -q_L0_G.z / n2);
    const double rho = state[7 * (size_t)N + 32 + l];
    if (!(rho < 0.05) && camera_pose_dev(state, N, t0, dt, lm_t0[l], &q_CtoG, &p_CinG)) {
      const v3 yu = cam_unproject(cam, lm_uv[2 * (size_t)l], lm_uv[2 * (size_t)l + 1]);
      const v3 p3d_C = mk(yu.x / rho, yu.y / rho, yu.z / rho);
      q = qrot(q_inv, (qrot(q_CtoG, p3d_C) + p_CinG) - t_L0_G);
      live = true;
    }
  }
  for (int k0 = 0; k0 < P; k0 += 256) {
    const int k = k0 + threadIdx.x;
    __syncthreads();
    if (k < P) {
      for (int a = 0; a < 3; ++a) { tab[10 * threadIdx.x + a] = planes10[4 * (size_t)P + 3 * (size_t)k + a]; tab[10 * threadIdx.x + 3 + a] = planes10[7 * (size_t)P + 3 * (size_t)k + a]; }
      for (int a = 0; a < 4; ++a) tab[10 * threadIdx.x + 6 + a] = planes10[4 * (size_t)k + a];
    }
    __syncthreads();
    if (!live) continue;
    const int m = min(256, P - k0);
    for (int j = 0; j < m; ++j) {   // ascending: the highest plane index wins, as the reference's loop
      const double* r = tab + 10 * j;
      const bool inside = (q.x > r[0]) & (q.x < r[3]) & (q.y > r[1]) & (q.y < r[4]) & (q.z > r[2]) & (q.z < r[5]);
      double dst = q.x * r[6] + q.y * r[7] + q.z * r[8] + r[9];
      dst = dst > 0 ? dst : -dst;
      if (inside & (dst <= radius * 2)) res = k0 + j;
    }
  }
  if (l < L) out[l] = res;
}
__global__ void k_lidar_pose(const double* state, int N, double t0, double dt, int n, const double* t, double* q4, double* p3, int* valid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  quat q; v3 p;
  const bool ok = lidar_pose_dev(state, N, t0, dt, t[i], &q, &p);
  valid[i] = ok ? 1 : 0;
  if (ok) { q4[4 * (size_t)i] = q.x; q4[4 * (size_t)i + 1] = q.y; q4[4 * (size_t)i + 2] = q.z; q4[4 * (size_t)i + 3] = q.w; p3[3 * (size_t)i] = p.x; p3[3 * (size_t)i + 1] = p.y; p3[3 * (size_t)i + 2] = p.z; }
}
struct PointXYZIT { float x, y, z, pad; float intensity; float pad2; double timestamp; };
static_assert(sizeof(PointXYZIT) == 32, "PointXYZIT layout (pcl_utils.h:39-44)");
// pose_d != null: the target frame's pose (q_L0_to_G x y z w | p | valid flag behind them) is still on the device (k_lidar_pose of the same stream): no host hop between the two
// kernels; an invalid pose (map time outside the trajectory) gives NaN points, the host reports it at its next synchronisation
__global__ void k_undistort(const double* state, int N, double t0, double dt, int n, const PointXYZIT* raw, quat qGt, v3 pT, int correct_position, float4* out, const double* pose_d = nullptr, const int* pose_ok = nullptr, int* flags_init = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (flags_init) flags_init[i] = -1;   // lvx_data_association: the association's "no surfel" start state rides with the de-skew (a 1.6 MB fill launch otherwise)
  const PointXYZIT r = raw[i];
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pose_d) { qGt = quat{-pose_d[0], -pose_d[1], -pose_d[2], pose_d[3]}; pT = mk(pose_d[4], pose_d[5], pose_d[6]); }
  if (isnan(r.x) || (pose_ok && !*pose_ok)) { o.x = o.y = o.z = NAN; out[i] = o; return; }
  quat q; v3 p;
  if (lidar_pose_dev(state, N, t0, dt, r.timestamp, &q, &p)) {
    const quat qk0 = qmul(qGt, q);
    v3 po = qrot(qk0, mk((double)r.x, (double)r.y, (double)r.z));
    if (correct_position) po = po + qrot(qGt, p - pT);
    o = make_float4((float)po.x, (float)po.y, (float)po.z, r.intensity);
  }
  out[i] = o;
}

// First map of a calibration (LIinitializer::Mapping + undistortScanInMap(odom_data_map), lvi_initialize_surfel_orb.cpp:1262-1300, scan_undistortion.h:40-57,95-116):
// every point is de-skewed ROTATION-ONLY into its scan's own start frame with the SO3 spline (undistort(..., correct_position = false): q_G_to_target = the LiDAR
// orientation at the scan's stamp, conjugated) — rounded to float, the VPoint of scan_data_ — and then moved with the scan's odometry pose as pcl::transformPointCloud does
// with a Matrix4d (double arithmetic on the float coordinates, rounded to float; non-finite points stay as they are).  Scans without a pose or whose stamp lies outside
// the spline are absent from scan_data_in_map_: their points become NaN here (no association, no map point).
__global__ void k_deskew_pose(const double* state, int N, double t0, double dt, int HW, int n, const PointXYZIT* raw, const double* q_scan, const int* present, const double* pose16, float4* out, int* flags_init) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flags_init[i] = -1;
  const int s = i / HW;
  const PointXYZIT r = raw[i];
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!present[s] || isnan(r.x)) { o.x = o.y = o.z = NAN; out[i] = o; return; }
  quat q; v3 p;
  if (lidar_pose_dev(state, N, t0, dt, r.timestamp, &q, &p)) {   // (a point whose own stamp lies outside the spline keeps the resize()'d zeros, as in the reference)
    const quat qGt = quat{-q_scan[4 * s], -q_scan[4 * s + 1], -q_scan[4 * s + 2], q_scan[4 * s + 3]};
    const v3 po = qrot(qmul(qGt, q), mk((double)r.x, (double)r.y, (double)r.z));
    o = make_float4((float)po.x, (float)po.y, (float)po.z, r.intensity);
  }
  const double* T = pose16 + 16 * (size_t)s;
  const double x = (double)o.x, y = (double)o.y, z = (double)o.z;
  out[i] = make_float4((float)(T[0] * x + T[1] * y + T[2] * z + T[3]), (float)(T[4] * x + T[5] * y + T[6] * z + T[7]), (float)(T[8] * x + T[9] * y + T[10] * z + T[11]), o.w);
}
// the key-scan map (LiDAROdometry::updateKeyScan: *map_cloud_ += scan_in_target): scans key[0], key[1], ... of the scans in the map frame, concatenated
__global__ void k_gather_scans(const float4* scans, const int* key, int HW, int n, float4* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = scans[(size_t)key[i / HW] * HW + i % HW];
}

}  // namespace lvx

using namespace lvx;

extern "C" {

// Scan registration of a batch of sweeps: six launches (count, bucket, gather, curvature, classify, compact; grid.y = sweep).  No host round trip in between: the
// kept-point count of a sweep is re-derived from its ring counts inside the kernels, the grids are sized by the input counts.  pts_d != null: the points are already
// on the device; otherwise pts (host) is uploaded into the scratch buffer.  Results stay in the scratch buffer (d_up[0]); sr_* describe it.
struct SrLayout { size_t pts, cloud, curv, label, sort, pick, src, lists, lflat, off, rc, ss, se, cnt, sharp, lsharp, flat, counts, err, end; };
static SrLayout sr_layout(long long N, int S, int n_rings, bool with_pts) {
  SrLayout L; size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 15) & ~(size_t)15; return at; };
  const size_t SR = (size_t)S * n_rings;
  L.pts = take(with_pts ? (size_t)N * 32 : 0); L.cloud = take((size_t)N * 16); L.curv = take((size_t)N * 4); L.label = take((size_t)N * 4); L.sort = take((size_t)N * 4); L.pick = take((size_t)N * 4);
  L.src = take((size_t)N * 4); L.lists = take((size_t)N * 16); L.lflat = take((size_t)N * 4); L.off = take((size_t)(S + 1) * 4); L.rc = take(SR * 4); L.ss = take(SR * 4); L.se = take(SR * 4);
  L.cnt = take(SR * 16); L.sharp = take(SR * 64); L.lsharp = take(SR * 512); L.flat = take(SR * 128); L.counts = take((size_t)S * 16); L.err = take((size_t)S * 4); L.end = o + 64;
  return L;
}
static int scan_register_launch(lvx_ctx* c, int S, const int32_t* off_h, const lvx_rs_point* pts_h, const lvx_rs_point* pts_d, int n_rings, float min_range) {
  hipStream_t st = c->stream;
  const long long N = off_h[S];
  int nmax = 0;
  for (int s = 0; s < S; ++s) nmax = std::max(nmax, off_h[s + 1] - off_h[s]);
  c->sr_S = 0;
  if (N == 0) { c->sr_S = S; c->sr_rings = n_rings; c->sr_N = 0; c->sr_off.assign(off_h, off_h + S + 1); c->sr_m.assign(S, 0); c->sr_counts.assign((size_t)S * 4, 0); return LVX_OK; }
  int rc;
  const SrLayout L = sr_layout(N, S, n_rings, pts_d == nullptr);
  if ((rc = dev_alloc(c, c->d_up[0], L.end))) return rc;
  char* base = (char*)c->d_up[0].p;
  const size_t SR = (size_t)S * n_rings;
  SrBatch B;
  B.pts = pts_d ? (const RsPoint*)pts_d : (const RsPoint*)(base + L.pts); B.off = (const int*)(base + L.off); B.n_rings = n_rings; B.thr = min_range; B.N = N;
  B.cloud = (float4*)(base + L.cloud); B.curv = (float*)(base + L.curv); B.label = (int*)(base + L.label); B.sort_ind = (int*)(base + L.sort); B.picked = (int*)(base + L.pick); B.src = (int*)(base + L.src);
  B.lists = (int*)(base + L.lists); B.lflat_r = (int*)(base + L.lflat); B.rc = (int*)(base + L.rc); B.ss = (int*)(base + L.ss); B.se = (int*)(base + L.se); B.cnt = (int*)(base + L.cnt);
  B.sharp_r = (int*)(base + L.sharp); B.lsharp_r = (int*)(base + L.lsharp); B.flat_r = (int*)(base + L.flat); B.counts = (int*)(base + L.counts); B.err = (int*)(base + L.err);
  ProfScope ps(c, LVX_KERNEL_UPSTREAM);
  if (!pts_d) LVX_HIP(c, hipMemcpyAsync(base + L.pts, pts_h, (size_t)N * 32, hipMemcpyHostToDevice, st));
  // The small words of a call — sweep offsets up, ring counts / list counts / error words back — go through the context's pinned words when they fit (a single sweep, a few):
  // hipMemcpyAsync to or from pageable memory is staged and returns when the copy is done, so the three "asynchronous" reads behind the kernels were three host stops of
  // ~20 us each on top of the final one (a third of a single sweep's 258 us).  Larger batches keep the pageable path: per sweep the stops are noise there.
  const size_t pin_need = (size_t)(S + 1) + SR + (size_t)S * 4 + (size_t)S;   // ints
  if (!c->pin) { LVX_HIP(c, hipHostMalloc((void**)&c->pin, 128 * 8, hipHostMallocDefault)); std::memset(c->pin, 0, 128 * 8); }
  int* pin = pin_need <= 256 ? (int*)c->pin : nullptr;
  if (pin) { std::memcpy(pin, off_h, (size_t)(S + 1) * 4); LVX_HIP(c, hipMemcpyAsync(base + L.off, pin, (size_t)(S + 1) * 4, hipMemcpyHostToDevice, st)); }
  else LVX_HIP(c, hipMemcpyAsync(base + L.off, off_h, (size_t)(S + 1) * 4, hipMemcpyHostToDevice, st));
  LVX_HIP(c, hipMemsetAsync(base + L.rc, 0, SR * 4, st));
  LVX_HIP(c, hipMemsetAsync(base + L.counts, 0, (L.err - L.counts) + (size_t)S * 4, st));   // counts and err are adjacent
  const unsigned gx = (unsigned)((nmax + 255) / 256);
  hipLaunchKernelGGL(k_sr_count, dim3(gx, S), dim3(256), 0, st, B);
  hipLaunchKernelGGL(k_sr_bucket, dim3(n_rings, S), dim3(1024), 0, st, B);
  hipLaunchKernelGGL(k_sr_gather, dim3(gx, S), dim3(256), 0, st, B);
  hipLaunchKernelGGL(k_sr_curv, dim3(gx, S), dim3(256), 0, st, B);
  hipLaunchKernelGGL(k_sr_classify, dim3(n_rings, S), dim3(512), 0, st, B);
  hipLaunchKernelGGL(k_sr_compact, dim3(n_rings, S), dim3(256), 0, st, B);
  LVX_HIP(c, hipGetLastError());
  std::vector<int> hrc(SR), herr(S);
  c->sr_counts.assign((size_t)S * 4, 0);
  if (pin) {
    int* prc = pin + (S + 1); int* pcnt = prc + SR; int* perr = pcnt + (size_t)S * 4;
    LVX_HIP(c, hipMemcpyAsync(prc, B.rc, SR * 4, hipMemcpyDeviceToHost, st));
    LVX_HIP(c, hipMemcpyAsync(pcnt, B.counts, (size_t)S * 16, hipMemcpyDeviceToHost, st));
    LVX_HIP(c, hipMemcpyAsync(perr, B.err, (size_t)S * 4, hipMemcpyDeviceToHost, st));
    LVX_HIP(c, hipStreamSynchronize(st));
    std::memcpy(hrc.data(), prc, SR * 4); std::memcpy(c->sr_counts.data(), pcnt, (size_t)S * 16); std::memcpy(herr.data(), perr, (size_t)S * 4);
  } else {
  LVX_HIP(c, hipMemcpyAsync(hrc.data(), B.rc, SR * 4, hipMemcpyDeviceToHost, st));
  LVX_HIP(c, hipMemcpyAsync(c->sr_counts.data(), B.counts, (size_t)S * 16, hipMemcpyDeviceToHost, st));
  LVX_HIP(c, hipMemcpyAsync(herr.data(), B.err, (size_t)S * 4, hipMemcpyDeviceToHost, st));
  LVX_HIP(c, hipStreamSynchronize(st));
  }
  c->sr_S = S; c->sr_rings = n_rings; c->sr_N = N; c->sr_off.assign(off_h, off_h + S + 1); c->sr_m.assign(S, 0); c->sr_batch_off = {L.cloud, L.lflat, L.ss, L.cnt};
  c->sr_lay = {L.cloud, L.curv, L.label, L.sort, L.pick, L.lists, L.ss, L.se};
  for (int s = 0; s < S; ++s) { int m = 0; for (int r = 0; r < n_rings; ++r) m += hrc[(size_t)s * n_rings + r]; c->sr_m[s] = m; }
  for (int s = 0; s < S; ++s) if (herr[s] & 8) return fail(c, LVX_E_ARG, "scan sector longer than the LDS sort capacity");
  return LVX_OK;
}
// download: everything the caller asked for — concatenated arrays into host staging (one copy per array over all sweeps), then a copy per sweep
static int scan_register_fetch(lvx_ctx* c, int s0, int S, lvx_scanreg_out* outs) {
  hipStream_t st = c->stream;
  const int n_rings = c->sr_rings;
  const size_t R = (size_t)n_rings;
  for (int s = 0; s < S; ++s) { outs[s].n = c->sr_m[s0 + s]; for (int k = 0; k < 4; ++k) outs[s].counts[k] = c->sr_counts[(size_t)(s0 + s) * 4 + k]; }
  if (c->sr_N == 0) return LVX_OK;
  const size_t lo = (size_t)c->sr_off[s0], hi = (size_t)c->sr_off[s0 + S], n = hi - lo, N = (size_t)c->sr_N;
  char* base = (char*)c->d_up[0].p;
  bool want[9] = {false};   // cloud curvature label sort_ind picked | sharp less_sharp flat less_flat
  bool want_ss = false, want_se = false;
  for (int s = 0; s < S; ++s) { const lvx_scanreg_out& q = outs[s]; const void* pp[9] = {q.cloud, q.curvature, q.label, q.sort_ind, q.picked, q.sharp, q.less_sharp, q.flat, q.less_flat};
    for (int k = 0; k < 9; ++k) want[k] = want[k] || pp[k]; want_ss = want_ss || q.scan_start; want_se = want_se || q.scan_end; }
  std::vector<float> hcloud(want[0] ? n * 4 : 0), hcurv(want[1] ? n : 0);
  std::vector<int> hint[7], hss(want_ss ? (size_t)S * R : 0), hse(want_se ? (size_t)S * R : 0);
  const int* lists = (const int*)(base + c->sr_lay[5]);
  const void* dsrc[9] = {(const float4*)(base + c->sr_lay[0]) + lo, (const float*)(base + c->sr_lay[1]) + lo, (const int*)(base + c->sr_lay[2]) + lo, (const int*)(base + c->sr_lay[3]) + lo,
                         (const int*)(base + c->sr_lay[4]) + lo, lists + lo, lists + N + lo, lists + 2 * N + lo, lists + 3 * N + lo};
  if (n > 0) {
    if (want[0]) LVX_HIP(c, hipMemcpyAsync(hcloud.data(), dsrc[0], n * 16, hipMemcpyDeviceToHost, st));
    if (want[1]) LVX_HIP(c, hipMemcpyAsync(hcurv.data(), dsrc[1], n * 4, hipMemcpyDeviceToHost, st));
    for (int k = 2; k < 9; ++k) if (want[k]) { hint[k - 2].resize(n); LVX_HIP(c, hipMemcpyAsync(hint[k - 2].data(), dsrc[k], n * 4, hipMemcpyDeviceToHost, st)); }
  }
  if (want_ss) LVX_HIP(c, hipMemcpyAsync(hss.data(), (const int*)(base + c->sr_lay[6]) + (size_t)s0 * R, (size_t)S * R * 4, hipMemcpyDeviceToHost, st));
  if (want_se) LVX_HIP(c, hipMemcpyAsync(hse.data(), (const int*)(base + c->sr_lay[7]) + (size_t)s0 * R, (size_t)S * R * 4, hipMemcpyDeviceToHost, st));
  LVX_HIP(c, hipStreamSynchronize(st));
  for (int s = 0; s < S; ++s) {
    lvx_scanreg_out& q = outs[s];
    const size_t o0 = (size_t)c->sr_off[s0 + s] - lo;
    const int m = q.n;
    if (q.scan_start) std::memcpy(q.scan_start, &hss[(size_t)s * R], R * 4);
    if (q.scan_end) std::memcpy(q.scan_end, &hse[(size_t)s * R], R * 4);
    if (m > 0) {
      if (q.cloud) std::memcpy(q.cloud, &hcloud[o0 * 4], (size_t)m * 16);
      if (q.curvature) std::memcpy(q.curvature, &hcurv[o0], (size_t)m * 4);
      int32_t* ip[3] = {q.label, q.sort_ind, q.picked};
      for (int k = 0; k < 3; ++k) if (ip[k]) std::memcpy(ip[k], &hint[k][o0], (size_t)m * 4);
    }
    int32_t* lp[4] = {q.sharp, q.less_sharp, q.flat, q.less_flat};
    for (int k = 0; k < 4; ++k) if (lp[k] && q.counts[k] > 0) std::memcpy(lp[k], &hint[3 + k][o0], (size_t)q.counts[k] * 4);
  }
  return LVX_OK;
}
static int scan_register_batch(lvx_ctx* c, int S, const int32_t* off_h, const lvx_rs_point* pts, int n_rings, float min_range, lvx_scanreg_out* outs) {
  for (int s = 0; s < S; ++s) { std::memset(outs[s].counts, 0, sizeof(outs[s].counts)); outs[s].n = 0; }
  int rc = scan_register_launch(c, S, off_h, pts, nullptr, n_rings, min_range);
  if (rc) return rc;
  return scan_register_fetch(c, 0, S, outs);
}
int lvx_scan_register(lvx_ctx* c, int n, const lvx_rs_point* pts, int n_rings, float min_range, lvx_scanreg_out* out) {
  if (!c || !out || n < 0 || n_rings <= 0 || n_rings > 1024 || (n > 0 && !pts)) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  const int32_t off[2] = {0, n};
  return scan_register_batch(c, 1, off, pts, n_rings, min_range, out);
}
int lvx_scan_register_batch_d(lvx_ctx* c, int n_sweeps, const int32_t* sweep_offsets, const lvx_rs_point* pts_d, int n_rings, float min_range, int32_t* n_kept, int32_t* counts4) {
  if (!c || n_sweeps <= 0 || !sweep_offsets || n_rings <= 0 || n_rings > 1024 || sweep_offsets[0] != 0) return LVX_E_ARG;
  for (int s = 0; s < n_sweeps; ++s) if (sweep_offsets[s + 1] < sweep_offsets[s]) return LVX_E_ARG;
  if (sweep_offsets[n_sweeps] > 0 && !pts_d) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  int rc = scan_register_launch(c, n_sweeps, sweep_offsets, nullptr, pts_d, n_rings, min_range);
  if (rc) return rc;
  for (int s = 0; s < n_sweeps; ++s) { if (n_kept) n_kept[s] = c->sr_m[s]; if (counts4) for (int k = 0; k < 4; ++k) counts4[4 * s + k] = c->sr_counts[(size_t)s * 4 + k]; }
  return LVX_OK;
}
int lvx_scan_register_get(lvx_ctx* c, int sweep, lvx_scanreg_out* out) {
  if (!c || !out) return LVX_E_ARG;
  if (c->sr_S <= 0) return fail(c, LVX_E_STATE, "lvx_scan_register has not been called");
  if (sweep < 0 || sweep >= c->sr_S) return fail(c, LVX_E_ARG, "sweep index outside the last batch");
  LVX_HIP(c, hipSetDevice(c->device));
  return scan_register_fetch(c, sweep, 1, out);
}
int lvx_scan_register_batch(lvx_ctx* c, int n_sweeps, const int32_t* sweep_offsets, const lvx_rs_point* pts, int n_rings, float min_range, lvx_scanreg_out* outs) {
  if (!c || !outs || n_sweeps <= 0 || !sweep_offsets || n_rings <= 0 || n_rings > 1024 || sweep_offsets[0] != 0) return LVX_E_ARG;
  for (int s = 0; s < n_sweeps; ++s) if (sweep_offsets[s + 1] < sweep_offsets[s]) return LVX_E_ARG;
  if (sweep_offsets[n_sweeps] > 0 && !pts) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  return scan_register_batch(c, n_sweeps, sweep_offsets, pts, n_rings, min_range, outs);
}

// voxel grid kept on the device in the context.  The whole build is ONE chain of launches without a host hop (k_vx_extent: extents + grid geometry -> k_vx_keys: cell-table
// clear + keys -> rocPRIM radix sort -> k_vx_leaf: run heads, leaf numbering, sums, eigen-solve), captured once per (cloud buffer, size, parameters) into a HIP graph and
// replayed: a small cloud's build is bound by the NUMBER of launches (~4.6 us each in a replayed graph), not by HBM.  Leaf arrays are strided by the CAPACITY (the point count: a leaf holds at least one point), the dense cell table by its own capacity.
static int vox_info(lvx_ctx* c);
static int voxel_enqueue(lvx_ctx* c, const float4* d_pts, int n, float leaf, int min_pts, double eig_mult) {
  hipStream_t st = c->stream;
  lvx_ctx::Voxels& V = c->vox;
  int* d_mm = (int*)V.misc.p;
  VxInfo* d_info = (VxInfo*)((char*)V.misc.p + 64);
  unsigned* k_in = (unsigned*)V.keys.p; unsigned* k_out = k_in + n;
  int* v_in = (int*)V.vals.p; int* v_out = v_in + n;
  unsigned* counts = (unsigned*)V.runs.p + n; unsigned* offs = counts + n;   // ([0, n) unused since the run-length encode left)
  unsigned long long* lbs = (unsigned long long*)((unsigned*)V.runs.p + (((size_t)n * 3 + 1) & ~(size_t)1));   // look-back states of k_vx_leaf, one per tile
  const size_t cap = (size_t)V.cap;
  const int n_tiles = (n + 255) / 256;
  // own radix sort up to VX_OWN_SORT_MAX points (one launch per 8-bit digit), rocPRIM above; either way the sorted keys / ids end in the second halves of the buffers
  const bool own_sort = n <= VX_OWN_SORT_MAX;
  // digit width: the narrowest of 8 / 9 / 10 bits that covers the key range in the fewest passes (<= 16 bits: 2 x 8, <= 18: 2 x 9, <= 20: 2 x 10, <= 24: 3 x 8, ...)
  int db = 8, passes = (V.sort_bits + 7) / 8;
  for (int d = 9; d <= 10; ++d) if ((V.sort_bits + d - 1) / d < passes) { db = d; passes = (V.sort_bits + d - 1) / d; }
  const int nbins = 1 << db, sort_tiles = (n + 1023) / 1024;
  int* ghist = own_sort ? (int*)V.tmp.p : nullptr;
  unsigned* tcnt = own_sort ? (unsigned*)V.tmp.p + 4 * 1024 : nullptr;
  hipLaunchKernelGGL(k_vx_extent, dim3((unsigned)std::min(std::max(n / 4096, 64), 256)), dim3(256), 0, st, d_pts, n, d_mm, leaf, (long long)V.cells_cap, d_info, ghist);
  const unsigned invalid = (unsigned)((1ull << V.sort_bits) - 1ull);   // sort_bits <= 31 (cells_cap is clamped), the 64-bit shift keeps even 32 defined
  unsigned* k_first = (own_sort && passes % 2 == 0) ? k_out : k_in;   // an even number of passes starts in the second buffer
  hipLaunchKernelGGL(k_vx_keys, dim3((unsigned)std::max((n + 1023) / 1024, 512)), dim3(256), 0, st, d_pts, n, (const VxInfo*)d_info, invalid, k_first, v_in, (int*)V.cells.p, lbs, n_tiles,
                     ghist, tcnt, passes * sort_tiles * nbins, passes, db);
  if (own_sort) {
    unsigned* ka = k_first; unsigned* kb = k_first == k_in ? k_out : k_in;
    int* va = k_first == k_in ? v_in : v_out; int* vb = k_first == k_in ? v_out : v_in;
    for (int q = 0; q < passes; ++q) {
      const int* gh_q = (const int*)ghist + 1024 * q; unsigned* tc_q = tcnt + (size_t)q * sort_tiles * nbins;
      auto pass = [&](auto first, auto width) {
        constexpr bool F = decltype(first)::value; constexpr int W = decltype(width)::value;
        hipLaunchKernelGGL((k_vx_sort_pass<F, W>), dim3((unsigned)sort_tiles), dim3(1 << W), 0, st, (const unsigned*)ka, (const int*)va, kb, vb, n, db * q, gh_q, tc_q);
      };
      auto by_width = [&](auto first) {
        if (db == 8) pass(first, std::integral_constant<int, 8>{}); else if (db == 9) pass(first, std::integral_constant<int, 9>{}); else pass(first, std::integral_constant<int, 10>{});
      };
      if (q == 0) by_width(std::true_type{}); else by_width(std::false_type{});
      std::swap(ka, kb); std::swap(va, vb);
    }
  } else {
    size_t t1 = V.tmp_bytes[0];
    LVX_HIP(c, rocprim::radix_sort_pairs(V.tmp.p, t1, k_in, k_out, v_in, v_out, (size_t)n, 0, (unsigned)V.sort_bits, st));   // stable: input order kept inside a leaf
  }
  int* lk = (int*)V.leaf_i.p; int* ln = lk + cap;
  double* mean = (double*)V.leaf_d.p; double* cov = mean + 3 * cap; double* icov = cov + 9 * cap; double* evecs = icov + 9 * cap; double* evals = evecs + 9 * cap;
  // tile = 256 PT sorted positions per workgroup, PT chosen so that the launch is ONE round of resident workgroups (4 per CU at 37 KB of LDS = 1 024) as long as the
  // cloud allows: the workgroup is a latency chain (keys -> counts of all predecessors -> points -> sums -> eigen-solve) and a second round repeats it
  auto launch_leaf = [&](auto pt) {
    constexpr int PT = decltype(pt)::value;
    hipLaunchKernelGGL(k_vx_leaf<PT>, dim3((unsigned)((n + 256 * PT - 1) / (256 * PT))), dim3(256), 0, st, d_pts, (const unsigned*)k_out, (const int*)v_out, n, invalid, lbs, min_pts, eig_mult, d_info, (int*)V.cells.p, lk, ln,
                       counts, offs, mean, cov, icov, evecs, evals, (float*)V.leaf_f.p, (VxInfo*)V.h_info);
  };
  if (n > 524288) launch_leaf(std::integral_constant<int, 4>{});
  else if (n > 262144) launch_leaf(std::integral_constant<int, 2>{});
  else launch_leaf(std::integral_constant<int, 1>{});
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}
static int voxel_build_device(lvx_ctx* c, const float4* d_pts, int n, float leaf, int min_pts, double eig_mult) {
  hipStream_t st = c->stream;
  int rc;
  lvx_ctx::Voxels& V = c->vox;
  V.leaf = leaf; V.min_pts = min_pts; V.eig_mult = eig_mult; V.n_points = n; V.n_leaves = 0; V.pending = false; V.d_pts = d_pts;
  std::memset(&V.grid, 0, sizeof(V.grid));
  if (n == 0) return LVX_OK;
  if (!V.h_info) LVX_HIP(c, hipHostMalloc(&V.h_info, sizeof(VxInfo), hipHostMallocDefault));
  if (V.cells_cap < (1 << 18) - 1) V.cells_cap = (1 << 18) - 1;   // 262 143 cells (1 MB; 18-bit keys = two 9-bit sort passes): a 32 m x 32 m x 32 m map at 0.5 m; grown on demand (vox_info)
  int bits = 1; while ((1ll << bits) - 1 < (long long)V.cells_cap && bits < 31) ++bits;   // keys < cells <= capacity <= 2^31 - 1, the invalid key = 2^bits - 1 >= capacity above them
  V.sort_bits = bits;
  if (!V.misc.p) {   // extents in their start state (k_vx_extent's last workgroup restores it after every build)
    if ((rc = dev_alloc(c, V.misc, 64 + sizeof(VxInfo)))) return rc;
    const int mm0[7] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000, 0};
    LVX_HIP(c, hipMemcpyAsync(V.misc.p, mm0, sizeof(mm0), hipMemcpyHostToDevice, st));
    LVX_HIP(c, hipStreamSynchronize(st));
  }
  if ((rc = dev_alloc(c, V.keys, (size_t)n * 4 * 2))) return rc;
  if ((rc = dev_alloc(c, V.vals, (size_t)n * 4 * 2))) return rc;
  if ((rc = dev_alloc(c, V.runs, ((size_t)n * 3 + 2 + 2 * (((size_t)n + 255) / 256) + 8) * 4))) return rc;
  if ((rc = dev_alloc(c, V.cells, (size_t)V.cells_cap * 4))) return rc;
  V.cap = n;
  if ((rc = dev_alloc(c, V.leaf_i, (size_t)n * 2 * 4))) return rc;
  if ((rc = dev_alloc(c, V.leaf_d, (size_t)n * 33 * 8))) return rc;
  if ((rc = dev_alloc(c, V.leaf_f, (size_t)n * 3 * 4))) return rc;
  {
    unsigned* k_in = (unsigned*)V.keys.p; int* v_in = (int*)V.vals.p;
    size_t t1 = 0;
    LVX_HIP(c, rocprim::radix_sort_pairs(nullptr, t1, k_in, k_in + n, v_in, v_in + n, (size_t)n, 0, (unsigned)V.sort_bits, st));
    V.tmp_bytes[0] = t1;
    const size_t own = (size_t)4 * 1024 * 4 + (size_t)4 * ((VX_OWN_SORT_MAX + 1023) / 1024) * 1024 * 4;   // digit histograms + per-tile digit counts of the own sort (up to 4 passes x 1 024 bins)
    if ((rc = dev_alloc(c, V.tmp, std::max(t1 + 16, own)))) return rc;
  }
  // replay the captured chain when nothing it was captured with has changed
  uint64_t lb = 0, eb = 0; std::memcpy(&lb, &leaf, 4); std::memcpy(&eb, &eig_mult, 8);
  const std::array<uint64_t, 16> key{(uint64_t)(uintptr_t)d_pts, (uint64_t)n, lb, (uint64_t)min_pts, eb, (uint64_t)(uintptr_t)V.keys.p, (uint64_t)(uintptr_t)V.vals.p, (uint64_t)(uintptr_t)V.runs.p,
                                     (uint64_t)(uintptr_t)V.cells.p, (uint64_t)(uintptr_t)V.tmp.p, (uint64_t)(uintptr_t)V.leaf_i.p, (uint64_t)(uintptr_t)V.leaf_d.p, (uint64_t)(uintptr_t)V.leaf_f.p,
                                     (uint64_t)(uintptr_t)V.misc.p, (uint64_t)V.cells_cap, (uint64_t)(uintptr_t)st};
  if (V.graph && key != V.graph_key) { (void)hipGraphExecDestroy((hipGraphExec_t)V.graph); V.graph = nullptr; }
  if (!V.graph && !V.graph_failed && !c->sw.no_graph && !c->profiling) {
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) == hipSuccess) {
      rc = voxel_enqueue(c, d_pts, n, leaf, min_pts, eig_mult);
      const hipError_t ce = hipStreamEndCapture(st, &graph);
      hipGraphExec_t exec = nullptr;
      if (!rc && ce == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) { V.graph = exec; V.graph_key = key; }
      else { V.graph_failed = true; (void)hipGetLastError(); }   // (rocPRIM could not be captured on this stack: plain launches from now on)
      if (graph) (void)hipGraphDestroy(graph);
    } else { V.graph_failed = true; (void)hipGetLastError(); }
  }
  if (V.graph) LVX_HIP(c, hipGraphLaunch((hipGraphExec_t)V.graph, st));
  else if ((rc = voxel_enqueue(c, d_pts, n, leaf, min_pts, eig_mult))) return rc;
  V.pending = true;
  return LVX_OK;
}
// The host's view of the last build (leaf count, grid geometry): waits for the chain once, when somebody asks; a cell table that was too small is grown and the build repeated.
static int vox_info(lvx_ctx* c) {
  lvx_ctx::Voxels& V = c->vox;
  for (int round = 0; V.pending; ++round) {
    LVX_HIP(c, hipStreamSynchronize(c->stream));
    V.pending = false;
    VxInfo inf; std::memcpy(&inf, V.h_info, sizeof(inf));
    if (inf.overflow == 2) return fail(c, LVX_E_ARG, "Leaf size is too small for the input dataset. Integer indices would overflow.");   // :80-85
    if (inf.overflow == 1) {
      if (round > 0) return fail(c, LVX_E_ALLOC, "voxel cell table could not be grown");
      V.cells_cap = std::min<long long>(inf.cells + inf.cells / 4, 2147483647LL);   // cells <= 2^31 - 1 (overflow 2 above): keys <= 2^31 - 2 stay below the invalid key of a 31-bit sort
      const int rc = voxel_build_device(c, (const float4*)V.d_pts, V.n_points, V.leaf, V.min_pts, V.eig_mult);
      if (rc) return rc;
      continue;
    }
    std::memcpy(&V.grid, &inf.g, sizeof(inf.g));
    V.n_leaves = inf.n_leaves;
  }
  return LVX_OK;
}

int lvx_voxel_build(lvx_ctx* c, int n, const float* xyzi4, float leaf, int min_pts, double eig_mult, lvx_voxel_info* info) {
  if (!c || n < 0 || !(leaf > 0) || (n > 0 && !xyzi4)) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  int rc;
  if ((rc = upload(c, c->d_up[1], xyzi4, (size_t)n * 16))) return rc;
  { ProfScope ps(c, LVX_KERNEL_UPSTREAM);
    if ((rc = voxel_build_device(c, (const float4*)c->d_up[1].p, n, leaf, min_pts, eig_mult))) return rc; }
  if ((rc = vox_info(c))) return rc;
  return info ? lvx_voxel_get_info(c, info) : LVX_OK;
}
int lvx_voxel_get_info(lvx_ctx* c, lvx_voxel_info* info) {
  if (!c || !info) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  const int rc = vox_info(c);
  if (rc) return rc;
  VxGrid g; std::memcpy(&g, &c->vox.grid, sizeof(g));
  info->n_leaves = c->vox.n_leaves; info->n_points = c->vox.n_points;
  for (int k = 0; k < 3; ++k) { info->min_b[k] = g.min_b[k]; info->max_b[k] = g.max_b[k]; info->div_b[k] = g.div_b[k]; info->divb_mul[k] = g.mul[k]; }
  return LVX_OK;
}
int lvx_voxel_build_d(lvx_ctx* c, int n, const float* xyzi4_d, float leaf, int min_pts, double eig_mult) {
  if (!c || n < 0 || !(leaf > 0) || (n > 0 && !xyzi4_d)) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  ProfScope ps(c, LVX_KERNEL_UPSTREAM);
  return voxel_build_device(c, (const float4*)xyzi4_d, n, leaf, min_pts, eig_mult);   // asynchronous: nothing comes back to the host until a consumer asks (vox_info)
}
int lvx_voxel_get(lvx_ctx* c, int32_t* leaf_key, int32_t* leaf_n, double* mean, double* cov, double* icov, double* evecs, double* evals, float* centroid,
                  int32_t* offsets, int32_t* point_ids) {
  if (!c) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  { const int rc = vox_info(c); if (rc) return rc; }
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  const lvx_ctx::Voxels& V = c->vox;
  const size_t nl = (size_t)V.n_leaves, n = (size_t)V.n_points, cap = (size_t)V.cap;
  if (nl == 0) { if (offsets) offsets[0] = 0; return LVX_OK; }
  const int* lk = (const int*)V.leaf_i.p;
  const double* d = (const double*)V.leaf_d.p;
  if (leaf_key) LVX_HIP(c, hipMemcpy(leaf_key, lk, nl * 4, hipMemcpyDeviceToHost));
  if (leaf_n) LVX_HIP(c, hipMemcpy(leaf_n, lk + cap, nl * 4, hipMemcpyDeviceToHost));
  if (mean) LVX_HIP(c, hipMemcpy(mean, d, nl * 24, hipMemcpyDeviceToHost));
  if (cov) LVX_HIP(c, hipMemcpy(cov, d + 3 * cap, nl * 72, hipMemcpyDeviceToHost));
  if (icov) LVX_HIP(c, hipMemcpy(icov, d + 12 * cap, nl * 72, hipMemcpyDeviceToHost));
  if (evecs) LVX_HIP(c, hipMemcpy(evecs, d + 21 * cap, nl * 72, hipMemcpyDeviceToHost));
  if (evals) LVX_HIP(c, hipMemcpy(evals, d + 30 * cap, nl * 24, hipMemcpyDeviceToHost));
  if (centroid) LVX_HIP(c, hipMemcpy(centroid, V.leaf_f.p, nl * 12, hipMemcpyDeviceToHost));
  if (offsets) {
    const unsigned* offs = (const unsigned*)V.runs.p + 2 * n;
    const unsigned* counts = (const unsigned*)V.runs.p + n;
    LVX_HIP(c, hipMemcpy(offsets, offs, nl * 4, hipMemcpyDeviceToHost));
    unsigned lc = 0; LVX_HIP(c, hipMemcpy(&lc, counts + (nl - 1), 4, hipMemcpyDeviceToHost));
    offsets[nl] = offsets[nl - 1] + (int)lc;
  }
  if (point_ids && n > 0) LVX_HIP(c, hipMemcpy(point_ids, (const int*)V.vals.p + n, n * 4, hipMemcpyDeviceToHost));
  return LVX_OK;
}
static int lookup_device(lvx_ctx* c, const float4* q_d, int nq, int* ids_d, int K) {
  { const int rc = vox_info(c); if (rc) return rc; }
  const lvx_ctx::Voxels& V = c->vox;
  if (!V.cells.p || V.n_leaves == 0) { LVX_HIP(c, hipMemsetAsync(ids_d, 0xff, (size_t)nq * 4 * K, c->stream)); return LVX_OK; }
  VxGrid g; std::memcpy(&g, &V.grid, sizeof(g));
  if (K == 7) hipLaunchKernelGGL(k_vx_lookup<7>, dim3((nq + 255) / 256), dim3(256), 0, c->stream, q_d, nq, V.leaf, V.min_pts, g, (const int*)V.cells.p, (const int*)V.leaf_i.p + V.cap, ids_d);
  else hipLaunchKernelGGL(k_vx_lookup<1>, dim3((nq + 255) / 256), dim3(256), 0, c->stream, q_d, nq, V.leaf, V.min_pts, g, (const int*)V.cells.p, (const int*)V.leaf_i.p + V.cap, ids_d);
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}
static int lookup_host(lvx_ctx* c, int nq, const float* xyzi4, int32_t* leaf_ids, int K) {
  if (!c || nq < 0 || (nq > 0 && (!xyzi4 || !leaf_ids))) return LVX_E_ARG;
  if (nq == 0) return LVX_OK;
  LVX_HIP(c, hipSetDevice(c->device));
  int rc;
  if ((rc = upload(c, c->d_up[2], xyzi4, (size_t)nq * 16))) return rc;
  if ((rc = dev_alloc(c, c->d_up[3], (size_t)nq * 4 * K))) return rc;
  { ProfScope ps(c, LVX_KERNEL_UPSTREAM);
    if ((rc = lookup_device(c, (const float4*)c->d_up[2].p, nq, (int*)c->d_up[3].p, K))) return rc; }
  LVX_HIP(c, hipMemcpyAsync(leaf_ids, c->d_up[3].p, (size_t)nq * 4 * K, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}
int lvx_voxel_lookup7(lvx_ctx* c, int nq, const float* xyzi4, int32_t* leaf_ids7) { return lookup_host(c, nq, xyzi4, leaf_ids7, 7); }
int lvx_voxel_lookup1(lvx_ctx* c, int nq, const float* xyzi4, int32_t* leaf_ids1) { return lookup_host(c, nq, xyzi4, leaf_ids1, 1); }
int lvx_voxel_lookup7_d(lvx_ctx* c, int nq, const float* xyzi4_d, int32_t* leaf_ids7_d) {
  if (!c || nq <= 0 || !xyzi4_d || !leaf_ids7_d) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  ProfScope ps(c, LVX_KERNEL_UPSTREAM);
  return lookup_device(c, (const float4*)xyzi4_d, nq, leaf_ids7_d, 7);
}
int lvx_voxel_lookup1_d(lvx_ctx* c, int nq, const float* xyzi4_d, int32_t* leaf_ids1_d) {
  if (!c || nq <= 0 || !xyzi4_d || !leaf_ids1_d) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  ProfScope ps(c, LVX_KERNEL_UPSTREAM);
  return lookup_device(c, (const float4*)xyzi4_d, nq, leaf_ids1_d, 1);
}

// S scans [S][H][W] against one plane table; flags [S][H * W].  Work buffers (bitmasks 4 P H ceil(W / 32) bytes per scan, counts) live in d_assoc[3] — a buffer
// nothing else writes, because it is cleared only when (re)allocated or re-shaped (k_assoc_select leaves it zero).
// geometry, per-cell counts, offsets and lists of the association grid (d_assoc[0], d_assoc[1]); have = it is already built for this table
static int assoc_grid_build(lvx_ctx* c, int P, const double* planes_d, bool have) {
  if (have) return LVX_OK;
  c->assoc_map_ready = false;   // the grid buffers are about to hold another table's grid
  int rc;
  const size_t grid_bytes = (sizeof(AssocGrid) + (size_t)(3 * SA_CELLS + 8) * 4 + 15) & ~(size_t)15;
  if ((rc = dev_alloc(c, c->d_assoc[0], grid_bytes + (size_t)P * 80))) return rc;
  AssocGrid* gd = (AssocGrid*)c->d_assoc[0].p; int* ccnt = (int*)(gd + 1); int* coff = ccnt + SA_CELLS; int* ccur = coff + SA_CELLS + 1;
  double* aos = (double*)((char*)c->d_assoc[0].p + grid_bytes);
  hipLaunchKernelGGL(k_assoc_grid_geom, dim3(1), dim3(256), 0, c->stream, P, planes_d, gd, (const int*)nullptr, ccnt);
  hipLaunchKernelGGL(k_assoc_grid_fill, dim3((unsigned)((P + 127) / 128)), dim3(128), 0, c->stream, P, planes_d, (const AssocGrid*)gd, ccnt, ccur, (int*)nullptr, 0, aos, (const int*)nullptr, 0x7fffffff);
  hipLaunchKernelGGL(k_assoc_grid_scan, dim3(1), dim3(1024), 0, c->stream, (const int*)ccnt, coff, ccur, 0x7fffffff);
  int total = 0;
  LVX_HIP(c, hipMemcpyAsync(&total, coff + SA_CELLS, 4, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  c->assoc_list_total = total;
  if ((rc = dev_alloc(c, c->d_assoc[1], (size_t)std::max(total, 1) * 4))) return rc;
  int* clist = (int*)c->d_assoc[1].p;
  hipLaunchKernelGGL(k_assoc_grid_fill, dim3((unsigned)((P + 127) / 128)), dim3(128), 0, c->stream, P, planes_d, (const AssocGrid*)gd, ccnt, ccur, clist, 1, (double*)nullptr, (const int*)nullptr, 0x7fffffff);
  return LVX_OK;
}
static int assoc_device(lvx_ctx* c, const float4* scans_d, int S, int H, int W, int P, const double* planes_d, double radius, int sel, int* flags_d) {
  if (P <= 0 || H <= 0 || W <= 0 || S <= 2) LVX_HIP(c, hipMemsetAsync(flags_d, 0xff, (size_t)std::max(S, 0) * H * W * 4, c->stream));   // (S > 2: k_assoc_hits writes the start state itself)
  if (P <= 0 || H <= 0 || W <= 0 || S <= 0) return LVX_OK;
  const int wpr = (W + 31) / 32;
  const size_t rings = (size_t)S * P * H, bits_bytes = (rings * wpr * 4 + 7) & ~(size_t)7, bytes = bits_bytes + rings * 8;
  const int oshift = wpr > 64 ? 1 : 0;   // (W <= SA_WMAX = 4096: at most 128 mask words per ring, two per occupancy bit)
  int rc;
  if (!c->d_assoc[3].p || c->d_assoc[3].bytes < bytes || c->assoc_rings != rings || c->assoc_wpr != wpr) {   // (re)allocated or re-shaped: clear once; k_assoc_select leaves it clean
    if ((rc = dev_alloc(c, c->d_assoc[3], bytes))) return rc;
    LVX_HIP(c, hipMemsetAsync(c->d_assoc[3].p, 0, c->d_assoc[3].bytes, c->stream));
    c->assoc_rings = rings; c->assoc_wpr = wpr;
  }
  unsigned* bits = (unsigned*)c->d_assoc[3].p; unsigned long long* occ = (unsigned long long*)((char*)c->d_assoc[3].p + bits_bytes);
  if (S <= 2) {   // a scan or two: all pairs beat the grid build
    hipLaunchKernelGGL(k_assoc_hits_allpairs, dim3((unsigned)((H * W + 255) / 256), (unsigned)((P + SA_PC - 1) / SA_PC), (unsigned)S), dim3(256), 0, c->stream, scans_d, H, W, P, planes_d, radius, bits, occ, wpr, oshift);
    hipLaunchKernelGGL(k_assoc_select, dim3((unsigned)((rings + 255) / 256)), dim3(256), 0, c->stream, bits, occ, S, H, W, P, wpr, oshift, sel, flags_d);
    LVX_HIP(c, hipGetLastError());
    return LVX_OK;
  }
  // the surfel grid (depends on the plane table only): built here, or once by lvx_surfel_map_prepare_d for every later call with that table
  if ((rc = assoc_grid_build(c, P, planes_d, c->assoc_map_ready && c->assoc_map_planes == planes_d && c->assoc_map_P == P))) return rc;
  AssocGrid* gd = (AssocGrid*)c->d_assoc[0].p; int* ccnt = (int*)(gd + 1); int* coff = ccnt + SA_CELLS; int* clist = (int*)c->d_assoc[1].p;
  (void)ccnt;
  const double* aos = (const double*)((const char*)c->d_assoc[0].p + ((sizeof(AssocGrid) + (size_t)(3 * SA_CELLS + 8) * 4 + 15) & ~(size_t)15));
  hipLaunchKernelGGL(k_assoc_hits, dim3((unsigned)((H * W + 255) / 256), (unsigned)S), dim3(256), 0, c->stream, scans_d, H, W, P, aos, radius, (const AssocGrid*)gd, (const int*)coff, (const int*)clist,
                     bits, occ, wpr, oshift, flags_d);
  hipLaunchKernelGGL(k_assoc_select, dim3((unsigned)((rings + 255) / 256)), dim3(256), 0, c->stream, bits, occ, S, H, W, P, wpr, oshift, sel, flags_d);
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}
int lvx_surfel_assoc(lvx_ctx* c, int H, int W, const float* scan_map_xyzi4, int n_planes, const double* plane_p4, const double* box_min3, const double* box_max3,
                     double radius, int sel_per_ring, int32_t* plane_of_point) {
  if (!c || H < 0 || W < 0 || W > SA_WMAX || n_planes < 0 || !plane_of_point || ((size_t)H * W > 0 && !scan_map_xyzi4)) return c ? fail(c, LVX_E_ARG, "bad surfel_assoc arguments (W <= 4096)") : LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  int rc;
  const size_t npt = (size_t)H * W;
  if ((rc = upload(c, c->d_up[4], scan_map_xyzi4, npt * 16))) return rc;
  std::vector<double> pl((size_t)n_planes * 10);
  if (n_planes > 0) { std::memcpy(pl.data(), plane_p4, (size_t)n_planes * 32); std::memcpy(pl.data() + 4 * (size_t)n_planes, box_min3, (size_t)n_planes * 24); std::memcpy(pl.data() + 7 * (size_t)n_planes, box_max3, (size_t)n_planes * 24); }
  if ((rc = upload(c, c->d_up[5], pl.data(), pl.size() * 8))) return rc;
  if (c->assoc_map_planes == (const double*)c->d_up[5].p) lvx_surfel_map_release(c);
  LVX_HIP(c, hipStreamSynchronize(c->stream));   // pl dies with this frame
  if ((rc = dev_alloc(c, c->d_up[6], npt * 4))) return rc;
  { ProfScope ps(c, LVX_KERNEL_UPSTREAM);
    if ((rc = assoc_device(c, (const float4*)c->d_up[4].p, 1, H, W, n_planes, (const double*)c->d_up[5].p, radius, sel_per_ring, (int*)c->d_up[6].p))) return rc; }
  if (npt) LVX_HIP(c, hipMemcpyAsync(plane_of_point, c->d_up[6].p, npt * 4, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}
int lvx_surfel_assoc_d(lvx_ctx* c, int H, int W, const float* scan_d, int n_planes, const double* planes10_d, double radius, int sel_per_ring, int32_t* plane_of_point_d) {
  return lvx_surfel_assoc_batch_d(c, 1, H, W, scan_d, n_planes, planes10_d, radius, sel_per_ring, plane_of_point_d);
}
int lvx_surfel_map_prepare_d(lvx_ctx* c, int n_planes, const double* planes10_d) {
  if (!c || n_planes < 0 || (n_planes > 0 && !planes10_d)) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  c->assoc_map_ready = false;
  if (n_planes == 0) return LVX_OK;
  int rc = assoc_grid_build(c, n_planes, planes10_d, false);
  if (rc) return rc;
  c->assoc_map_planes = planes10_d; c->assoc_map_P = n_planes; c->assoc_map_ready = true;
  return LVX_OK;
}
int lvx_surfel_map_release(lvx_ctx* c) {
  if (!c) return LVX_E_ARG;
  c->assoc_map_ready = false; c->assoc_map_planes = nullptr; c->assoc_map_P = 0;
  return LVX_OK;
}
int lvx_surfel_assoc_batch_d(lvx_ctx* c, int n_scans, int H, int W, const float* scans_d, int n_planes, const double* planes10_d, double radius, int sel_per_ring, int32_t* plane_of_point_d) {
  if (!c || n_scans <= 0 || H <= 0 || W <= 0 || W > SA_WMAX || n_planes < 0 || !scans_d || !plane_of_point_d || (n_planes > 0 && !planes10_d)) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  ProfScope ps(c, LVX_KERNEL_UPSTREAM);
  const int chunk = 64;   // scans per launch: bounds the bitmask work buffer (7 MB per scan at P = 2000, 16 x 1800)
  for (int s0 = 0; s0 < n_scans; s0 += chunk) {
    const int ns = std::min(chunk, n_scans - s0);
    int rc = assoc_device(c, (const float4*)scans_d + (size_t)s0 * H * W, ns, H, W, n_planes, planes10_d, radius, sel_per_ring, plane_of_point_d + (size_t)s0 * H * W);
    if (rc) return rc;
  }
  return LVX_OK;
}
// the publication words of the single-launch compactions (lvx_ctx::d_pub): allocated and cleared once per context
// Kernels whose workgroups wait for words published by EVERY other workgroup of the launch (k_surfel_compact_mb, k_assoc_emit_fused) terminate only if the whole grid is
// resident at once.  The bound is asked of the runtime for THIS device (occupancy API x compute units; a partition or a smaller part has fewer) and taken with a margin of
// one workgroup per CU (the API is a block high near a register-file edge: MI355X_MICROARCH.md), never above the size of the publication table; larger launches take the
// single-workgroup / two-launch path.  (the bound was a hard-coded 8 x 256 CUs.)
static int coresident_bound(lvx_ctx* c, const void* kernel, int block, int table) {
  int per_cu = 0, ncu = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) ncu = prop.multiProcessorCount;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess) per_cu = 0;
  per_cu = std::min(per_cu, 8) - 1;
  return std::max(0, std::min(table, per_cu * ncu));
}
static int pub_words(lvx_ctx* c) {
  if (c->d_pub.p) return LVX_OK;
  int rc = dev_alloc(c, c->d_pub, (SE_MAXWG + SC_MAXB) * 8); if (rc) return rc;
  LVX_HIP(c, hipMemsetAsync(c->d_pub.p, 0, c->d_pub.bytes, c->stream));
  return LVX_OK;
}
// SurfelPoint lists of S associated scans, concatenated in scan order, every scan in the reference's chronological (column-major) order
int lvx_surfel_emit_d(lvx_ctx* c, int n_scans, int H, int W, const int32_t* flags_d, const float* scans_map_d, const lvx_point_xyzit* scans_raw_d, int max_out,
                      double* pt3_d, double* pt_map3_d, double* t_d, int32_t* plane_d, int32_t* n_out, int32_t* per_scan_counts) {
  if (!c || n_scans <= 0 || H <= 0 || W <= 0 || !flags_d || !scans_map_d || !scans_raw_d || !n_out || max_out < 0) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  int rc = dev_alloc(c, c->d_assoc[2], (size_t)n_scans * 4 + 16); if (rc) return rc;
  int* cnt_d = (int*)c->d_assoc[2].p;
  if ((rc = pub_words(c))) return rc;
  ProfScope ps(c, LVX_KERNEL_UPSTREAM);
  SurfelOut o{pt3_d, pt_map3_d, t_d, plane_d};
  const bool have_out = pt3_d && pt_map3_d && t_d && plane_d && max_out > 0;
  const int parts = (W + SE_COLS - 1) / SE_COLS;
  if (c->coresident_emit < 0) c->coresident_emit = coresident_bound(c, (const void*)k_assoc_emit_fused, SE_COLS, SE_MAXWG);
  if ((long long)n_scans * parts <= c->coresident_emit && H <= SE_HMAX) {   // one launch: count, publish, write behind the workgroups before (k_assoc_emit_fused); every workgroup resident
    if (++c->emit_epoch == 0u) c->emit_epoch = 1u;
    LVX_HIP(c, hipMemsetAsync(cnt_d, 0, (size_t)n_scans * 4, c->stream));
    hipLaunchKernelGGL(k_assoc_emit_fused, dim3((unsigned)parts, (unsigned)n_scans), dim3(SE_COLS), 0, c->stream, flags_d, (const float4*)scans_map_d, (const void*)scans_raw_d, H, W, cnt_d,
                       (unsigned long long*)c->d_pub.p, c->emit_epoch, max_out, have_out ? 1 : 0, o);
  } else {
    hipLaunchKernelGGL(k_assoc_emit, dim3((unsigned)n_scans), dim3(1024), 0, c->stream, flags_d, (const float4*)scans_map_d, (const void*)scans_raw_d, H, W, cnt_d, n_scans, 0, 0, o);
    if (have_out)   // written only if the whole list fits max_out (decided on the device from the counts): one host synchronisation per call
      hipLaunchKernelGGL(k_assoc_emit, dim3((unsigned)n_scans), dim3(1024), 0, c->stream, flags_d, (const float4*)scans_map_d, (const void*)scans_raw_d, H, W, cnt_d, n_scans, max_out, 1, o);
  }
  std::vector<int> cnt((size_t)n_scans);
  LVX_HIP(c, hipMemcpyAsync(cnt.data(), cnt_d, (size_t)n_scans * 4, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipGetLastError());
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  int total = 0;
  for (int s = 0; s < n_scans; ++s) { total += cnt[s]; if (per_scan_counts) per_scan_counts[s] = cnt[s]; }
  *n_out = total;
  if (total > max_out || total == 0) return LVX_OK;   // the caller sizes the outputs from *n_out and calls again
  if (!have_out) return LVX_E_ARG;
  return LVX_OK;
}

// host-buffer convenience over the two device entry points above: S organised scans (map frame + raw) in, flags and the SurfelPoint list out
int lvx_surfel_assoc_emit(lvx_ctx* c, int n_scans, int H, int W, const float* scans_map_xyzi4, const lvx_point_xyzit* scans_raw, int n_planes, const double* plane_p4, const double* box_min3,
                          const double* box_max3, double radius, int sel_per_ring, int32_t* plane_of_point, int max_out, double* pt3, double* pt_map3, double* t, int32_t* plane, int32_t* n_out) {
  if (!c || n_scans <= 0 || H <= 0 || W <= 0 || W > SA_WMAX || n_planes < 0 || !scans_map_xyzi4 || !scans_raw || !n_out || (n_planes > 0 && (!plane_p4 || !box_min3 || !box_max3))) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  const size_t npt = (size_t)n_scans * H * W;
  int rc;
  std::vector<double> pl((size_t)std::max(n_planes, 1) * 10, 0.0);
  if (n_planes > 0) { std::memcpy(pl.data(), plane_p4, (size_t)n_planes * 32); std::memcpy(pl.data() + 4 * (size_t)n_planes, box_min3, (size_t)n_planes * 24); std::memcpy(pl.data() + 7 * (size_t)n_planes, box_max3, (size_t)n_planes * 24); }
  if ((rc = upload(c, c->d_up[4], scans_map_xyzi4, npt * 16))) return rc;
  if ((rc = upload(c, c->d_up[2], scans_raw, npt * 32))) return rc;
  if ((rc = upload(c, c->d_up[5], pl.data(), pl.size() * 8))) return rc;
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  if ((rc = dev_alloc(c, c->d_up[6], npt * 4))) return rc;
  // the context-owned table was just overwritten: whatever grid was prepared for this address is stale.  One grid for all chunks of this call.
  lvx_surfel_map_release(c);
  if (n_scans > 2 && n_planes > 0 && (rc = lvx_surfel_map_prepare_d(c, n_planes, (const double*)c->d_up[5].p))) return rc;
  rc = lvx_surfel_assoc_batch_d(c, n_scans, H, W, (const float*)c->d_up[4].p, n_planes, (const double*)c->d_up[5].p, radius, sel_per_ring, (int32_t*)c->d_up[6].p);
  lvx_surfel_map_release(c);
  if (rc) return rc;
  if (plane_of_point) LVX_HIP(c, hipMemcpyAsync(plane_of_point, c->d_up[6].p, npt * 4, hipMemcpyDeviceToHost, c->stream));
  int32_t total = 0;
  if ((rc = lvx_surfel_emit_d(c, n_scans, H, W, (const int32_t*)c->d_up[6].p, (const float*)c->d_up[4].p, (const lvx_point_xyzit*)c->d_up[2].p, 0, nullptr, nullptr, nullptr, nullptr, &total, nullptr))) return rc;
  *n_out = total;
  if (total == 0 || total > max_out || !pt3 || !pt_map3 || !t || !plane) { LVX_HIP(c, hipStreamSynchronize(c->stream)); return LVX_OK; }
  if ((rc = dev_alloc(c, c->d_up[3], (size_t)total * (24 + 24 + 8 + 4) + 64))) return rc;
  double* d_pt = (double*)c->d_up[3].p; double* d_pm = d_pt + 3 * (size_t)total; double* d_t = d_pm + 3 * (size_t)total; int32_t* d_pl = (int32_t*)(d_t + total);
  if ((rc = lvx_surfel_emit_d(c, n_scans, H, W, (const int32_t*)c->d_up[6].p, (const float*)c->d_up[4].p, (const lvx_point_xyzit*)c->d_up[2].p, total, d_pt, d_pm, d_t, d_pl, &total, nullptr))) return rc;
  LVX_HIP(c, hipMemcpyAsync(pt3, d_pt, (size_t)total * 24, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipMemcpyAsync(pt_map3, d_pm, (size_t)total * 24, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipMemcpyAsync(t, d_t, (size_t)total * 8, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipMemcpyAsync(plane, d_pl, (size_t)total * 4, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}

int lvx_landmark_assoc(lvx_ctx* c, const double* state, const double* q_LtoC_xyzw, const double* t_LinC3, double map_time, int n_planes, const double* plane_p4, const double* box_min3,
                       const double* box_max3, double radius, int32_t* plane_of_landmark) {
  if (!c || !state || !q_LtoC_xyzw || !t_LinC3 || n_planes < 0 || (c->L > 0 && !plane_of_landmark) || (n_planes > 0 && (!plane_p4 || !box_min3 || !box_max3))) return LVX_E_ARG;
  if (!c->have_spline) return fail(c, LVX_E_STATE, "lvx_set_spline has not been called");
  if (c->L == 0) return LVX_OK;
  LVX_HIP(c, hipSetDevice(c->device));
  int rc;
  const int L = c->L;
  std::vector<double> pl((size_t)std::max(n_planes, 1) * 10, 0.0);
  if (n_planes > 0) { std::memcpy(pl.data(), plane_p4, (size_t)n_planes * 32); std::memcpy(pl.data() + 4 * (size_t)n_planes, box_min3, (size_t)n_planes * 24); std::memcpy(pl.data() + 7 * (size_t)n_planes, box_max3, (size_t)n_planes * 24); }
  if ((rc = upload(c, c->d_up[5], pl.data(), pl.size() * 8))) return rc;
  if ((rc = upload(c, c->d_up[2], state, (size_t)lvx_state_size(c) * 8))) return rc;
  if ((rc = upload(c, c->d_up[3], c->lm_uv.data(), (size_t)L * 16))) return rc;
  if ((rc = upload(c, c->d_up[4], c->lm_t0.data(), (size_t)L * 8))) return rc;
  if ((rc = dev_alloc(c, c->d_up[6], (size_t)L * 4))) return rc;
  { ProfScope ps(c, LVX_KERNEL_UPSTREAM);
    hipLaunchKernelGGL(k_landmark_assoc, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, c->stream, (const double*)c->d_up[2].p, c->N, c->t0, c->dt, c->cam, (const double*)c->d_up[3].p,
                       (const double*)c->d_up[4].p, L, map_time, mkq(q_LtoC_xyzw[3], q_LtoC_xyzw[0], q_LtoC_xyzw[1], q_LtoC_xyzw[2]), mk(t_LinC3[0], t_LinC3[1], t_LinC3[2]), n_planes,
                       (const double*)c->d_up[5].p, radius, (int*)c->d_up[6].p); }
  LVX_HIP(c, hipGetLastError());
  LVX_HIP(c, hipMemcpyAsync(plane_of_landmark, c->d_up[6].p, (size_t)L * 4, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}

int lvx_evaluate_lidar_pose(lvx_ctx* c, const double* state, int n, const double* t, double* q4, double* p3, int32_t* valid) {
  if (!c || !state || n < 0 || (n > 0 && (!t || !q4 || !p3 || !valid))) return LVX_E_ARG;
  if (!c->have_spline) return fail(c, LVX_E_STATE, "lvx_set_spline has not been called");
  if (n == 0) return LVX_OK;
  LVX_HIP(c, hipSetDevice(c->device));
  int rc;
  if ((rc = upload(c, c->d_up[7], state, (size_t)lvx_state_size(c) * 8))) return rc;
  if ((rc = upload(c, c->d_up[2], t, (size_t)n * 8))) return rc;
  if ((rc = dev_alloc(c, c->d_up[3], (size_t)n * (32 + 24 + 4)))) return rc;
  double* dq = (double*)c->d_up[3].p; double* dp = dq + 4 * (size_t)n; int* dv = (int*)(dp + 3 * (size_t)n);
  { ProfScope ps(c, LVX_KERNEL_UPSTREAM);
    hipLaunchKernelGGL(k_lidar_pose, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const double*)c->d_up[7].p, c->N, c->t0, c->dt, n, (const double*)c->d_up[2].p, dq, dp, dv); }
  LVX_HIP(c, hipGetLastError());
  LVX_HIP(c, hipMemcpyAsync(q4, dq, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipMemcpyAsync(p3, dp, (size_t)n * 24, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipMemcpyAsync(valid, dv, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}
int lvx_undistort_scan(lvx_ctx* c, const double* state, int n, const lvx_point_xyzit* raw, const double* qGt, const double* pT, int correct_position, float* out) {
  if (!c || !state || n < 0 || !qGt || !pT || (n > 0 && (!raw || !out))) return LVX_E_ARG;
  if (!c->have_spline) return fail(c, LVX_E_STATE, "lvx_set_spline has not been called");
  if (n == 0) return LVX_OK;
  LVX_HIP(c, hipSetDevice(c->device));
  int rc;
  if ((rc = upload(c, c->d_up[7], state, (size_t)lvx_state_size(c) * 8))) return rc;
  if ((rc = upload(c, c->d_up[2], raw, (size_t)n * 32))) return rc;
  if ((rc = dev_alloc(c, c->d_up[3], (size_t)n * 16))) return rc;
  { ProfScope ps(c, LVX_KERNEL_UPSTREAM);
    hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const double*)c->d_up[7].p, c->N, c->t0, c->dt, n, (const PointXYZIT*)c->d_up[2].p,
                       load_q(qGt), load_v3(pT), correct_position, (float4*)c->d_up[3].p); }
  LVX_HIP(c, hipGetLastError());
  LVX_HIP(c, hipMemcpyAsync(out, c->d_up[3].p, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}

// compaction of the accepted planes: [records | plane table] in recs, the count in d_cnt (device)
static int surfel_compact_launch(lvx_ctx* c, const SurfelPlaneDev* all, const int* flag, int nl, SurfelPlaneDev* recs, int* d_cnt, const VxInfo* info) {
  const int nb = (nl + 1023) / 1024;
  if (c->coresident_compact < 0) c->coresident_compact = coresident_bound(c, (const void*)k_surfel_compact_mb, 256, SC_MAXB);
  if (nb > c->coresident_compact) {   // (every workgroup of the multi-block kernel must be resident on THIS device)
    hipLaunchKernelGGL(k_surfel_compact, dim3(1), dim3(1024), 0, c->stream, all, flag, nl, recs, d_cnt, info);
    return LVX_OK;
  }
  int rc = pub_words(c); if (rc) return rc;
  if (++c->compact_epoch == 0u) c->compact_epoch = 1u;
  hipLaunchKernelGGL(k_surfel_compact_mb, dim3((unsigned)std::max(nb, 1)), dim3(256), 0, c->stream, all, flag, nl, recs, d_cnt, info, (unsigned long long*)c->d_pub.p + SE_MAXWG, c->compact_epoch);
  return LVX_OK;
}
// setSurfelMap over the leaves of the context's voxel grid: the accepted planes in voxel-key (std::map) order
static int surfel_extract_device(lvx_ctx* c, double p_lambda, double dist_threshold, int min_leaf_points, int min_inliers, std::vector<SurfelPlaneDev>& out, DevBuf& dst) {
  out.clear();
  { const int rc0 = vox_info(c); if (rc0) return rc0; }
  const lvx_ctx::Voxels& V = c->vox;
  const int nl = V.n_leaves, n = V.n_points; const size_t cap = (size_t)V.cap;
  if (nl == 0) return LVX_OK;
  if (!V.d_pts) return fail(c, LVX_E_STATE, "lvx_voxel_build has not been called");
  int rc;
  if ((rc = dev_alloc(c, c->d_up[4], (size_t)nl * sizeof(SurfelPlaneDev)))) return rc;
  if ((rc = dev_alloc(c, c->d_up[5], (size_t)nl * 4))) return rc;
  if (c->assoc_map_planes == (const double*)c->d_up[5].p) lvx_surfel_map_release(c);
  const unsigned* counts = (const unsigned*)V.runs.p + n; const unsigned* offs = (const unsigned*)V.runs.p + 2 * (size_t)n;
  const int* lk = (const int*)V.leaf_i.p; const double* d = (const double*)V.leaf_d.p;
  { ProfScope ps(c, LVX_KERNEL_UPSTREAM);
    hipLaunchKernelGGL(k_surfel_extract, dim3((nl + 3) / 4), dim3(256), 0, c->stream, (const float4*)V.d_pts, counts, offs, (const int*)V.vals.p + n, nl, lk + cap, d, d + 21 * cap,
                       d + 30 * cap, p_lambda, dist_threshold, min_leaf_points, min_inliers, (SurfelPlaneDev*)c->d_up[4].p, (int*)c->d_up[5].p, (const VxInfo*)nullptr); }
  LVX_HIP(c, hipGetLastError());
  // compaction on the device: [records | plane table] in dst, the count comes back (4 bytes), then the records
  if ((rc = dev_alloc(c, dst, (size_t)nl * (sizeof(SurfelPlaneDev) + 80) + 64))) return rc;
  int* d_cnt = (int*)((char*)dst.p + dst.bytes - 16);
  if ((rc = surfel_compact_launch(c, (const SurfelPlaneDev*)c->d_up[4].p, (const int*)c->d_up[5].p, nl, (SurfelPlaneDev*)dst.p, d_cnt, nullptr))) return rc;
  int P = 0;
  LVX_HIP(c, hipMemcpyAsync(&P, d_cnt, 4, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  out.resize((size_t)P);
  if (P > 0) LVX_HIP(c, hipMemcpy(out.data(), dst.p, (size_t)P * sizeof(SurfelPlaneDev), hipMemcpyDeviceToHost));
  return LVX_OK;
}
int lvx_surfel_extract(lvx_ctx* c, double p_lambda, double dist_threshold, int min_leaf_points, int min_inliers, int max_planes, lvx_surfel_plane* planes, int32_t* n_planes) {
  static_assert(sizeof(lvx_surfel_plane) == sizeof(SurfelPlaneDev), "plane record layout");
  if (!c || !n_planes || max_planes < 0 || (max_planes > 0 && !planes)) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  *n_planes = 0;
  std::vector<SurfelPlaneDev> acc;
  int rc = surfel_extract_device(c, p_lambda, dist_threshold, min_leaf_points, min_inliers, acc, c->d_up[6]);
  if (rc) return rc;
  const int np = (int)acc.size();
  if (np > 0 && max_planes > 0) std::memcpy(planes, acc.data(), (size_t)std::min(np, max_planes) * sizeof(SurfelPlaneDev));
  *n_planes = np;
  return LVX_OK;
}

// ---- LIinitializer::DataAssociation, refinement branch (src/lvi_exc/test/lvi_initialize_surfel_orb.cpp:1180-1201), device-resident -------------------------------
int lvx_set_scans(lvx_ctx* c, int n_scans, int H, int W, const lvx_point_xyzit* raw) {
  if (!c || n_scans < 0 || H < 0 || W < 0 || W > SA_WMAX || ((size_t)n_scans * H * W > 0 && !raw)) return c ? fail(c, LVX_E_ARG, "bad lvx_set_scans arguments (W <= 4096)") : LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  c->da_S = n_scans; c->da_H = H; c->da_W = W; c->da_planes.clear(); c->da_points = 0; c->da_planes_pending = 0; c->da_cap_nl = c->da_cap_P = c->da_cap_list = 0;
  int rc = upload(c, c->d_da[0], raw, (size_t)n_scans * H * W * 32);
  if (rc) return rc;
  LVX_HIP(c, hipStreamSynchronize(c->stream));   // the caller's buffer may go away
  return LVX_OK;
}
int lvx_assoc_default_options(lvx_assoc_options* o) {
  if (!o) return LVX_E_ARG;
  o->ndt_resolution = 0.5f; o->min_points_per_voxel = 6; o->min_covar_eigvalue_mult = 0.01; o->plane_lambda = 0.7; o->fit_threshold = 0.05; o->min_leaf_points = 10; o->min_inliers = 20;
  o->radius = 0.05; o->selected_per_ring = 2; o->reserved = 0;
  return LVX_OK;
}
// ---- lvx_data_association without host stops (round 5) ---------------------------------------------------------------------------------------------------
// The synchronous chain below waits for the device four times: the leaf count of the voxel grid (launch size of the plane extraction), the plane count (shape of the
// association), the length of the association grid's lists (their allocation), the SurfelPoint counts.  A calibration calls lvx_data_association once per refinement
// round on the SAME scans with a slightly moved trajectory, so the three intermediate counts barely change: the speculative chain launches everything over CAPACITIES
// learned from the previous call (kernels read the true counts from device memory and clamp to the capacity), waits ONCE at the end, and looks at what the device
// mirrored into pinned memory — a count above its capacity (or a cell table that was too small) discards the attempt and the synchronous chain runs.
static int da_capacity(int have, int need, int slack) { return (have >= need && (long long)have <= 4ll * need + 4ll * slack) ? have : need + need / 8 + slack; }
// returns LVX_OK, an error, or 1 = a capacity was exceeded (nothing of the attempt is kept)
static int da_speculative(lvx_ctx* c, const lvx_assoc_options& o, int S, int H, int W, size_t npt, const int* dv, int32_t* n_planes, int32_t* n_points) {
  hipStream_t st = c->stream;
  int rc;
  lvx_ctx::Voxels& V = c->vox;
  if (!c->da_pinned) LVX_HIP(c, hipHostMalloc((void**)&c->da_pinned, 64, hipHostMallocDefault));
  const int nl_cap = std::min<long long>(c->da_cap_nl, (long long)npt), P_cap = std::min(c->da_cap_P, nl_cap), list_cap = c->da_cap_list;
  if ((rc = voxel_build_device(c, (const float4*)c->d_da[4].p, (int)npt, o.ndt_resolution, o.min_points_per_voxel, o.min_covar_eigvalue_mult))) return rc;
  const VxInfo* d_info = (const VxInfo*)((const char*)V.misc.p + 64);
  const int n = V.n_points; const size_t cap = (size_t)V.cap;
  // setSurfelMap over the first min(leaves, nl_cap) leaves
  if ((rc = dev_alloc(c, c->d_up[4], (size_t)nl_cap * sizeof(SurfelPlaneDev)))) return rc;
  if ((rc = dev_alloc(c, c->d_up[5], (size_t)nl_cap * 4))) return rc;
  DevBuf& dst = c->d_da[5];
  if ((rc = dev_alloc(c, dst, (size_t)nl_cap * (sizeof(SurfelPlaneDev) + 80) + 64))) return rc;
  int* d_cnt = (int*)((char*)dst.p + dst.bytes - 16);
  { const unsigned* counts = (const unsigned*)V.runs.p + n; const unsigned* offs = (const unsigned*)V.runs.p + 2 * (size_t)n;
    const int* lk = (const int*)V.leaf_i.p; const double* d = (const double*)V.leaf_d.p;
    hipLaunchKernelGGL(k_surfel_extract, dim3((nl_cap + 3) / 4), dim3(256), 0, st, (const float4*)V.d_pts, counts, offs, (const int*)V.vals.p + n, nl_cap, lk + cap, d, d + 21 * cap, d + 30 * cap,
                       o.plane_lambda, o.fit_threshold, o.min_leaf_points, o.min_inliers, (SurfelPlaneDev*)c->d_up[4].p, (int*)c->d_up[5].p, d_info);
    if ((rc = surfel_compact_launch(c, (const SurfelPlaneDev*)c->d_up[4].p, (const int*)c->d_up[5].p, nl_cap, (SurfelPlaneDev*)dst.p, d_cnt, d_info))) return rc; }
  // the association grid of min(planes, P_cap) planes, lists of list_cap entries
  const size_t grid_bytes = (sizeof(AssocGrid) + (size_t)(3 * SA_CELLS + 8) * 4 + 15) & ~(size_t)15;
  c->assoc_map_ready = false;
  if ((rc = dev_alloc(c, c->d_assoc[0], grid_bytes + (size_t)P_cap * 80))) return rc;
  if ((rc = dev_alloc(c, c->d_assoc[1], (size_t)std::max(list_cap, 1) * 4))) return rc;
  AssocGrid* gd = (AssocGrid*)c->d_assoc[0].p; int* ccnt = (int*)(gd + 1); int* coff = ccnt + SA_CELLS; int* ccur = coff + SA_CELLS + 1; int* clist = (int*)c->d_assoc[1].p;
  double* aos = (double*)((char*)c->d_assoc[0].p + grid_bytes);
  const double* recs = (const double*)dst.p;
  hipLaunchKernelGGL(k_assoc_grid_geom, dim3(1), dim3(256), 0, st, P_cap, recs, gd, (const int*)d_cnt, ccnt);
  hipLaunchKernelGGL(k_assoc_grid_fill, dim3((unsigned)((P_cap + 127) / 128)), dim3(128), 0, st, P_cap, recs, (const AssocGrid*)gd, ccnt, ccur, (int*)nullptr, 0, aos, (const int*)d_cnt, list_cap);
  hipLaunchKernelGGL(k_assoc_grid_scan, dim3(1), dim3(1024), 0, st, (const int*)ccnt, coff, ccur, list_cap);
  hipLaunchKernelGGL(k_assoc_grid_fill, dim3((unsigned)((P_cap + 127) / 128)), dim3(128), 0, st, P_cap, recs, (const AssocGrid*)gd, ccnt, ccur, clist, 1, (double*)nullptr, (const int*)d_cnt, list_cap,
                     c->da_pinned, dv);
  // flags of every scan: rings strided by the plane CAPACITY (rings of planes that do not exist stay empty)
  const int wpr = (W + 31) / 32, chunk = 64;   // (flags: d_da[6], set to -1 by the de-skew kernel)
  { const size_t rings = (size_t)std::min(chunk, S) * P_cap * H, bits_bytes = (rings * wpr * 4 + 7) & ~(size_t)7, bytes = bits_bytes + rings * 8;
    const int oshift = wpr > 64 ? 1 : 0;
    if (!c->d_assoc[3].p || c->d_assoc[3].bytes < bytes || c->assoc_rings != rings || c->assoc_wpr != wpr) {
      if ((rc = dev_alloc(c, c->d_assoc[3], bytes))) return rc;
      LVX_HIP(c, hipMemsetAsync(c->d_assoc[3].p, 0, c->d_assoc[3].bytes, st));
      c->assoc_rings = rings; c->assoc_wpr = wpr;
    }
    unsigned* bits = (unsigned*)c->d_assoc[3].p; unsigned long long* occ = (unsigned long long*)((char*)c->d_assoc[3].p + bits_bytes);
    for (int s0 = 0; s0 < S; s0 += chunk) {
      const int ns = std::min(chunk, S - s0);
      const float4* sc = (const float4*)c->d_da[4].p + (size_t)s0 * H * W;
      hipLaunchKernelGGL(k_assoc_hits, dim3((unsigned)((H * W + 255) / 256), (unsigned)ns), dim3(256), 0, st, sc, H, W, P_cap, (const double*)aos, o.radius, (const AssocGrid*)gd, (const int*)coff, (const int*)clist, bits, occ, wpr, oshift, (int*)nullptr);
      hipLaunchKernelGGL(k_assoc_select, dim3((unsigned)(((size_t)ns * P_cap * H + 255) / 256)), dim3(256), 0, st, bits, occ, ns, H, W, P_cap, wpr, oshift, o.selected_per_ring, (int*)c->d_da[6].p + (size_t)s0 * H * W);
    } }
  LVX_HIP(c, hipGetLastError());
  // SurfelPoint lists (capacity-strided), then THE host stop
  int32_t total = 0;
  if ((rc = dev_alloc(c, c->d_da[7], npt * (24 + 24 + 8 + 4) + 64))) return rc;
  { double* d_pt = (double*)c->d_da[7].p; double* d_pm = d_pt + 3 * npt; double* d_t = d_pm + 3 * npt; int32_t* d_pl = (int32_t*)(d_t + npt);
    if ((rc = lvx_surfel_emit_d(c, S, H, W, (const int32_t*)c->d_da[6].p, (const float*)c->d_da[4].p, (const lvx_point_xyzit*)c->d_da[0].p, (int)npt, d_pt, d_pm, d_t, d_pl, &total, nullptr))) return rc; }
  V.pending = false;
  VxInfo inf; std::memcpy(&inf, V.h_info, sizeof(inf));
  const int hv = c->da_pinned[0], P = c->da_pinned[1], ltot = c->da_pinned[2];
  if (inf.overflow == 2) return fail(c, LVX_E_ARG, "Leaf size is too small for the input dataset. Integer indices would overflow.");
  if (inf.overflow == 1 || inf.n_leaves > nl_cap || P > P_cap || ltot > list_cap) {
    if (inf.overflow != 1) { c->da_cap_nl = da_capacity(c->da_cap_nl, inf.n_leaves, 1024); if (inf.n_leaves <= nl_cap) c->da_cap_P = da_capacity(c->da_cap_P, P, 128); if (inf.n_leaves <= nl_cap && P <= P_cap) c->da_cap_list = da_capacity(c->da_cap_list, ltot, 4096); }
    return 1;
  }
  std::memcpy(&V.grid, &inf.g, sizeof(inf.g));
  V.n_leaves = inf.n_leaves;
  if (!hv) return fail(c, LVX_E_RANGE, "map time outside the trajectory");
  c->da_cap_nl = da_capacity(c->da_cap_nl, inf.n_leaves, 1024); c->da_cap_P = da_capacity(c->da_cap_P, P, 128); c->da_cap_list = da_capacity(c->da_cap_list, ltot, 4096);
  c->da_planes.clear(); c->da_planes_pending = P;   // fetched from d_da[5] when lvx_get_surfel_map asks
  if (n_planes) *n_planes = P;
  c->da_points = P > 0 ? total : 0;
  if (n_points) *n_points = c->da_points;
  return LVX_OK;
}
// steps 2-4 of a DataAssociation round, four host stops: voxel grid of the MAP cloud (the scans in the map frame for a refinement round, the key-scan map for the first
// one), surfel map, association of every scan in d_da[4], SurfelPoint emission.  dv: device flag of the map pose (null: nothing to check)
static int da_sync_chain(lvx_ctx* c, const lvx_assoc_options& o, const float4* map_pts, size_t map_n, int S, int H, int W, size_t npt, const int* dv, int32_t* n_planes, int32_t* n_points) {
  int rc;
  hipStream_t st = c->stream; (void)st;
  // 2. LiDAROdometry::ndtInit(resolution) + setInputTarget(map_cloud): the voxel covariance grid of the map cloud
  if ((rc = voxel_build_device(c, map_pts, (int)map_n, o.ndt_resolution, o.min_points_per_voxel, o.min_covar_eigvalue_mult))) return rc;
  // 3. SurfelAssociation::setSurfelMap
  std::vector<SurfelPlaneDev> acc;
  lvx_surfel_map_release(c);
  if ((rc = surfel_extract_device(c, o.plane_lambda, o.fit_threshold, o.min_leaf_points, o.min_inliers, acc, c->d_da[5]))) return rc;   // records + the association's plane table stay in d_da[5]
  if (dv) { int hv = 0; LVX_HIP(c, hipMemcpy(&hv, dv, 4, hipMemcpyDeviceToHost)); if (!hv) return fail(c, LVX_E_RANGE, "map time outside the trajectory"); }   // (the stream has been waited for above)
  const int P = (int)acc.size();
  c->da_planes.resize((size_t)P);
  if (P > 0) std::memcpy(c->da_planes.data(), acc.data(), (size_t)P * sizeof(SurfelPlaneDev));
  if (n_planes) *n_planes = P;
  c->da_cap_nl = da_capacity(c->da_cap_nl, c->vox.n_leaves, 1024); c->da_cap_P = da_capacity(c->da_cap_P, P, 128);
  if (P == 0) return LVX_OK;
  const double* planes_d = (const double*)((const char*)c->d_da[5].p + (((size_t)P * sizeof(SurfelPlaneDev) + 15) & ~(size_t)15));
  // 4. getAssociation for every scan: flags (one surfel grid for all scans), then the chronological SurfelPoint lists, scans concatenated
  if ((rc = dev_alloc(c, c->d_da[6], npt * 4))) return rc;
  if (S > 2 && (rc = lvx_surfel_map_prepare_d(c, P, planes_d))) return rc;
  if (S > 2) c->da_cap_list = da_capacity(c->da_cap_list, c->assoc_list_total, 4096);
  rc = lvx_surfel_assoc_batch_d(c, S, H, W, (const float*)c->d_da[4].p, P, planes_d, o.radius, o.selected_per_ring, (int32_t*)c->d_da[6].p);
  lvx_surfel_map_release(c);
  if (rc) return rc;
  // outputs strided by the CAPACITY (every scan point could be a SurfelPoint): count + write in one call, one host synchronisation
  int32_t total = 0;
  if ((rc = dev_alloc(c, c->d_da[7], npt * (24 + 24 + 8 + 4) + 64))) return rc;
  { double* d_pt = (double*)c->d_da[7].p; double* d_pm = d_pt + 3 * npt; double* d_t = d_pm + 3 * npt; int32_t* d_pl = (int32_t*)(d_t + npt);
    if ((rc = lvx_surfel_emit_d(c, S, H, W, (const int32_t*)c->d_da[6].p, (const float*)c->d_da[4].p, (const lvx_point_xyzit*)c->d_da[0].p, (int)npt, d_pt, d_pm, d_t, d_pl, &total, nullptr))) return rc; }
  c->da_points = total;
  if (n_points) *n_points = total;
  return LVX_OK;
}
int lvx_data_association(lvx_ctx* c, const double* state, double map_time, const lvx_assoc_options* opt_in, int32_t* n_planes, int32_t* n_points) {
  if (!c || !state) return LVX_E_ARG;
  if (!c->have_spline) return fail(c, LVX_E_STATE, "lvx_set_spline has not been called");
  if (c->da_S <= 0 || (size_t)c->da_H * c->da_W == 0) return fail(c, LVX_E_STATE, "lvx_set_scans has not been called");
  lvx_assoc_options o; lvx_assoc_default_options(&o); if (opt_in) o = *opt_in;
  LVX_HIP(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const int S = c->da_S, H = c->da_H, W = c->da_W;
  const size_t npt = (size_t)S * H * W;
  if (npt > 2147483647ull) return fail(c, LVX_E_ARG, "too many scan points for one map cloud");
  int rc;
  ProfScope ps(c, LVX_KERNEL_UPSTREAM);
  c->da_planes.clear(); c->da_points = 0; c->da_planes_pending = 0;
  if (n_planes) *n_planes = 0;
  if (n_points) *n_points = 0;
  // 1. ScanUndistortion::undistortScanInMap (scan_undistortion.h:59-74): the LiDAR pose at the map time, then EVERY point of EVERY scan moved with the pose at its own
  //    timestamp into that frame — one launch over the whole recording; map_cloud_ = the scans concatenated in order
  if ((rc = upload(c, c->d_da[1], state, (size_t)lvx_state_size(c) * 8))) return rc;
  if ((rc = upload(c, c->d_da[2], &map_time, 8))) return rc;
  if ((rc = dev_alloc(c, c->d_da[3], 64 + 8))) return rc;
  double* dq = (double*)c->d_da[3].p; double* dp = dq + 4; int* dv = (int*)(dp + 3);
  hipLaunchKernelGGL(k_lidar_pose, dim3(1), dim3(1), 0, st, (const double*)c->d_da[1].p, c->N, c->t0, c->dt, 1, (const double*)c->d_da[2].p, dq, dp, dv);
  if ((rc = dev_alloc(c, c->d_da[4], npt * 16))) return rc;
  if ((rc = dev_alloc(c, c->d_da[6], npt * 4))) return rc;
  hipLaunchKernelGGL(k_undistort, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, st, (const double*)c->d_da[1].p, c->N, c->t0, c->dt, (int)npt, (const PointXYZIT*)c->d_da[0].p,
                     quat{0, 0, 0, 1}, mk(0, 0, 0), 1, (float4*)c->d_da[4].p, (const double*)dq, (const int*)dv, (int*)c->d_da[6].p);   // the pose stays on the device (q_L0_to_G.conjugate() is taken there)
  c->vox.d_pts = c->d_da[4].p;
  lvx_surfel_map_release(c);
  // steps 2-4 over the capacities of the previous call, one host stop (S > 2: one or two scans take the all-pairs kernel, whose launch shape needs the plane count)
  if (S > 2 && c->da_cap_nl > 0 && c->da_cap_P > 0 && c->da_cap_list > 0 && !c->sw.da_sync) {
    c->da_spec_runs++;
    rc = da_speculative(c, o, S, H, W, npt, dv, n_planes, n_points);
    if (rc <= 0) return rc;
    c->da_spec_misses++;
    c->da_planes.clear(); c->da_points = 0; c->da_planes_pending = 0;
  }
  return da_sync_chain(c, o, (const float4*)c->d_da[4].p, npt, S, H, W, npt, dv, n_planes, n_points);
}
// The FIRST DataAssociation of a calibration: the map comes from per-scan odometry poses (LOAM's: ReadPoseGT, lvi_initialize_surfel_orb.cpp:458-516), not from the spline —
// the InitializationDone branch (:1175-1178): Mapping() (:1262-1300) = undistortScan() + LiDAROdometry::feedScan(t, scan, pose, update_map, using_loam = true) per scan
// (src/core/lidar_odometry.cpp:45-74: the pose is taken as given; updateKeyScan / checkKeyScan :89-128: a key scan is the first one, or one further than key_dist from the
// last key scan, or turned by more than key_angle_deg in yaw, pitch or roll — it joins the key-scan map, the NDT target), undistortScanInMap(odom_data_map)
// (scan_undistortion.h:95-116), setSurfelMap(lidar_odom->getNDTPtr(), map_time) over the voxel grid of the KEY-SCAN map, getAssociation of every scan.
static void r2ypr_deg(const double* T, double ypr[3]) {   // mathutils::R2ypr (include/utils/math_utils.h:192-207), T row-major 4 x 4
  const double n0 = T[0], n1 = T[4], n2 = T[8], o0 = T[1], o1 = T[5], a0 = T[2], a1 = T[6];
  const double y = std::atan2(n1, n0);
  const double p = std::atan2(-n2, n0 * std::cos(y) + n1 * std::sin(y));
  const double r = std::atan2(a0 * std::sin(y) - a1 * std::cos(y), -o0 * std::sin(y) + o1 * std::cos(y));
  ypr[0] = y / M_PI * 180.0; ypr[1] = p / M_PI * 180.0; ypr[2] = r / M_PI * 180.0;
}
int lvx_data_association_poses(lvx_ctx* c, const double* state, const double* scan_t, const double* pose16, const int32_t* has_pose, double key_dist, double key_angle_deg,
                               const lvx_assoc_options* opt_in, int32_t* n_planes, int32_t* n_points, int32_t* key_scan) {
  if (!c || !state || !scan_t || !pose16) return LVX_E_ARG;
  if (!c->have_spline) return fail(c, LVX_E_STATE, "lvx_set_spline has not been called");
  if (c->da_S <= 0 || (size_t)c->da_H * c->da_W == 0) return fail(c, LVX_E_STATE, "lvx_set_scans has not been called");
  lvx_assoc_options o; lvx_assoc_default_options(&o); if (opt_in) o = *opt_in;
  LVX_HIP(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const int S = c->da_S, H = c->da_H, W = c->da_W, HW = H * W;
  const size_t npt = (size_t)S * HW;
  if (npt > 2147483647ull) return fail(c, LVX_E_ARG, "too many scan points for one map cloud");
  int rc;
  ProfScope ps(c, LVX_KERNEL_UPSTREAM);
  c->da_planes.clear(); c->da_points = 0; c->da_planes_pending = 0;
  if (n_planes) *n_planes = 0;
  if (n_points) *n_points = 0;
  // per-scan block on the device: [t S | q 4S | p 3S | pose 16S] doubles, then [valid S | present S | key S] ints
  const size_t nd = (size_t)S * (1 + 4 + 3 + 16);
  if ((rc = dev_alloc(c, c->d_da_aux, nd * 8 + (size_t)S * 12 + 64))) return rc;
  double* d_t = (double*)c->d_da_aux.p; double* d_q = d_t + S; double* d_p = d_q + 4 * (size_t)S; double* d_pose = d_p + 3 * (size_t)S;
  int* d_valid = (int*)(d_pose + 16 * (size_t)S); int* d_present = d_valid + S; int* d_key = d_present + S;
  if ((rc = upload(c, c->d_da[1], state, (size_t)lvx_state_size(c) * 8))) return rc;
  LVX_HIP(c, hipMemcpyAsync(d_t, scan_t, (size_t)S * 8, hipMemcpyHostToDevice, st));
  LVX_HIP(c, hipMemcpyAsync(d_pose, pose16, (size_t)S * 128, hipMemcpyHostToDevice, st));
  // ScanUndistortion::undistortScan: the target frame of scan s is the LiDAR orientation at its stamp; a stamp outside the spline drops the scan ("pass")
  hipLaunchKernelGGL(k_lidar_pose, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, (const double*)c->d_da[1].p, c->N, c->t0, c->dt, S, (const double*)d_t, d_q, d_p, d_valid);
  std::vector<int> present((size_t)S), key;
  LVX_HIP(c, hipMemcpyAsync(present.data(), d_valid, (size_t)S * 4, hipMemcpyDeviceToHost, st));
  LVX_HIP(c, hipStreamSynchronize(st));
  // Mapping(): scans in order; feedScan needs the scan's pose (loam_poses_map_.find(stamp)); checkKeyScan against the LAST KEY scan
  double pos_last[3] = {0, 0, 0}, ypr_last[3] = {0, 0, 0};
  for (int s = 0; s < S; ++s) {
    present[s] = (present[s] && (!has_pose || has_pose[s])) ? 1 : 0;
    if (key_scan) key_scan[s] = 0;
    if (!present[s]) continue;
    const double* T = pose16 + 16 * (size_t)s;
    const double dx = T[3] - pos_last[0], dy = T[7] - pos_last[1], dz = T[11] - pos_last[2];
    const double dist = std::sqrt(dx * dx + dy * dy + dz * dz);
    double ypr[3]; r2ypr_deg(T, ypr);
    bool turned = false;
    for (int a = 0; a < 3; ++a) {
      double d = ypr[a] - ypr_last[a];
      if (d > 180) d -= 360;           // LiDAROdometry::normalize_angle (include/core/lidar_odometry.h:95-102)
      if (d < -180) d += 360;
      if (std::fabs(d) > key_angle_deg) turned = true;
    }
    if (key.empty() || dist > key_dist || turned) {
      pos_last[0] = T[3]; pos_last[1] = T[7]; pos_last[2] = T[11]; ypr_last[0] = ypr[0]; ypr_last[1] = ypr[1]; ypr_last[2] = ypr[2];
      key.push_back(s);
      if (key_scan) key_scan[s] = 1;
    }
  }
  if (key.empty()) return LVX_OK;   // no scan with a pose inside the spline: an empty map
  LVX_HIP(c, hipMemcpyAsync(d_present, present.data(), (size_t)S * 4, hipMemcpyHostToDevice, st));
  LVX_HIP(c, hipMemcpyAsync(d_key, key.data(), key.size() * 4, hipMemcpyHostToDevice, st));
  if ((rc = dev_alloc(c, c->d_da[4], npt * 16))) return rc;
  if ((rc = dev_alloc(c, c->d_da[6], npt * 4))) return rc;
  hipLaunchKernelGGL(k_deskew_pose, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, st, (const double*)c->d_da[1].p, c->N, c->t0, c->dt, HW, (int)npt, (const PointXYZIT*)c->d_da[0].p,
                     (const double*)d_q, (const int*)d_present, (const double*)d_pose, (float4*)c->d_da[4].p, (int*)c->d_da[6].p);
  const size_t nmap = key.size() * (size_t)HW;
  if ((rc = dev_alloc(c, c->d_da_key, nmap * 16))) return rc;
  hipLaunchKernelGGL(k_gather_scans, dim3((unsigned)((nmap + 255) / 256)), dim3(256), 0, st, (const float4*)c->d_da[4].p, (const int*)d_key, HW, (int)nmap, (float4*)c->d_da_key.p);
  LVX_HIP(c, hipGetLastError());
  LVX_HIP(c, hipStreamSynchronize(st));   // (the host vectors above go out of scope)
  lvx_surfel_map_release(c);
  return da_sync_chain(c, o, (const float4*)c->d_da_key.p, nmap, S, H, W, npt, nullptr, n_planes, n_points);
}
int lvx_data_association_stats(lvx_ctx* c, int64_t* one_stop_rounds, int64_t* repeated_rounds) {
  if (!c) return LVX_E_ARG;
  if (one_stop_rounds) *one_stop_rounds = c->da_spec_runs;
  if (repeated_rounds) *repeated_rounds = c->da_spec_misses;
  return LVX_OK;
}
int lvx_get_surfel_map(lvx_ctx* c, int max_planes, lvx_surfel_plane* planes) {
  if (!c || max_planes < 0 || (max_planes > 0 && !planes)) return LVX_E_ARG;
  if (c->da_planes_pending > 0) {   // the last lvx_data_association left its records on the device (no host stop for them): fetch them now, once
    LVX_HIP(c, hipSetDevice(c->device));
    c->da_planes.resize((size_t)c->da_planes_pending);
    LVX_HIP(c, hipMemcpyAsync(c->da_planes.data(), c->d_da[5].p, (size_t)c->da_planes_pending * sizeof(lvx_surfel_plane), hipMemcpyDeviceToHost, c->stream));
    LVX_HIP(c, hipStreamSynchronize(c->stream));
    c->da_planes_pending = 0;
  }
  const size_t n = std::min((size_t)max_planes, c->da_planes.size());
  if (n) std::memcpy(planes, c->da_planes.data(), n * sizeof(lvx_surfel_plane));
  return LVX_OK;
}
int lvx_get_surfel_points(lvx_ctx* c, int max_points, double* pt3, double* pt_map3, double* t, int32_t* plane) {
  if (!c || max_points < 0) return LVX_E_ARG;
  const size_t total = (size_t)c->da_points, n = std::min((size_t)max_points, total);
  if (n == 0) return LVX_OK;
  LVX_HIP(c, hipSetDevice(c->device));
  const size_t cap = (size_t)c->da_S * c->da_H * c->da_W;   // the lists are strided by the capacity (lvx_data_association)
  const double* d_pt = (const double*)c->d_da[7].p; const double* d_pm = d_pt + 3 * cap; const double* d_t = d_pm + 3 * cap; const int32_t* d_pl = (const int32_t*)(d_t + cap);
  if (pt3) LVX_HIP(c, hipMemcpyAsync(pt3, d_pt, n * 24, hipMemcpyDeviceToHost, c->stream));
  if (pt_map3) LVX_HIP(c, hipMemcpyAsync(pt_map3, d_pm, n * 24, hipMemcpyDeviceToHost, c->stream));
  if (t) LVX_HIP(c, hipMemcpyAsync(t, d_t, n * 8, hipMemcpyDeviceToHost, c->stream));
  if (plane) LVX_HIP(c, hipMemcpyAsync(plane, d_pl, n * 4, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}
int lvx_get_scans_in_map(lvx_ctx* c, float* xyzi4) {
  if (!c || !xyzi4) return LVX_E_ARG;
  const size_t npt = (size_t)c->da_S * c->da_H * c->da_W;
  if (npt == 0 || !c->d_da[4].p) return fail(c, LVX_E_STATE, "lvx_data_association has not been called");
  LVX_HIP(c, hipSetDevice(c->device));
  LVX_HIP(c, hipMemcpyAsync(xyzi4, c->d_da[4].p, npt * 16, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}

}  // extern "C"
// ---- NDT registration: host side ---------------------------------------------------------------------------------------------------------------------
namespace {
struct NdtTables { NdtConst f; NdtConstD d; };
// Gaussian fitting parameters (eq. 6.8 [Magnusson 2009]; ndt_omp_impl.hpp:86-93) and computeAngleDerivatives (:289-394) at the transform vector p6
NdtTables ndt_tables(const double* p6, double res, double outlier_ratio) {
  NdtTables T;
  const double gauss_c1 = 10.0 * (1 - outlier_ratio), gauss_c2 = outlier_ratio / std::pow(res, 3);
  const double gauss_d3 = -std::log(gauss_c2);
  const double gauss_d1 = -std::log(gauss_c1 + gauss_c2) - gauss_d3;
  const double gauss_d2 = -2 * std::log((-std::log(gauss_c1 * std::exp(-0.5) + gauss_c2) - gauss_d3) / gauss_d1);
  T.f.gauss_d1 = gauss_d1; T.f.gd2 = (float)gauss_d2; T.d.gd1 = gauss_d1; T.d.gd2 = gauss_d2;
  double cx, cy, cz, sx, sy, sz;
  if (std::fabs(p6[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = std::cos(p6[3]); sx = std::sin(p6[3]); }
  if (std::fabs(p6[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = std::cos(p6[4]); sy = std::sin(p6[4]); }
  if (std::fabs(p6[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = std::cos(p6[5]); sz = std::sin(p6[5]); }
  const double ja[8][3] = {{-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy}, {cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy}, {-sy * cz, sy * sz, cy}, {sx * cy * cz, -sx * cy * sz, sx * sy},
                           {-cx * cy * cz, cx * cy * sz, -cx * sy}, {-cy * sz, -cy * cz, 0}, {cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0}, {sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0}};
  const double ha[15][3] = {{-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy}, {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy}, {cx * cy * cz, -cx * cy * sz, cx * sy},
                            {sx * cy * cz, -sx * cy * sz, sx * sy}, {-sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0}, {cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0}, {-cy * cz, cy * sz, sy},
                            {-sx * sy * cz, sx * sy * sz, sx * cy}, {cx * sy * cz, -cx * sy * sz, -cx * cy}, {sy * sz, sy * cz, 0}, {-sx * cy * sz, -sx * cy * cz, 0}, {cx * cy * sz, cx * cy * cz, 0},
                            {-cy * cz, cy * sz, 0}, {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0}, {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0}};
  for (int r = 0; r < 8; ++r) for (int k = 0; k < 3; ++k) { T.f.j_ang[r][k] = (float)ja[r][k]; T.d.j_ang[r][k] = ja[r][k]; }
  for (int r = 0; r < 15; ++r) for (int k = 0; k < 3; ++k) { T.f.h_ang[r][k] = (float)ha[r][k]; T.d.h_ang[r][k] = ha[r][k]; }
  return T;
}
// Eigen::AngleAxis<float>::toRotationMatrix about a coordinate axis, and the float transform of a 6-vector:
// (Translation<float,3>(p0, p1, p2) * AngleAxis<float>(p3, X) * AngleAxis<float>(p4, Y) * AngleAxis<float>(p5, Z)).matrix()   (ndt_omp_impl.hpp:826-829)
void ndt_axis_rot(float angle, int axis, float R[3][3]) {
  const float s = std::sin(angle), c = std::cos(angle);
  float a[3] = {0, 0, 0}; a[axis] = 1.0f;
  const float sa[3] = {s * a[0], s * a[1], s * a[2]}, c1[3] = {(1.0f - c) * a[0], (1.0f - c) * a[1], (1.0f - c) * a[2]};
  float t;
  t = c1[0] * a[1]; R[0][1] = t - sa[2]; R[1][0] = t + sa[2];
  t = c1[0] * a[2]; R[0][2] = t + sa[1]; R[2][0] = t - sa[1];
  t = c1[1] * a[2]; R[1][2] = t - sa[0]; R[2][1] = t + sa[0];
  for (int i = 0; i < 3; ++i) R[i][i] = c1[i] * a[i] + c;
}
void ndt_matrix(const double* p6, float* M16) {
  float Rx[3][3], Ry[3][3], Rz[3][3], A[3][3], R[3][3];
  ndt_axis_rot((float)p6[3], 0, Rx); ndt_axis_rot((float)p6[4], 1, Ry); ndt_axis_rot((float)p6[5], 2, Rz);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = (Rx[i][0] * Ry[0][j] + Rx[i][1] * Ry[1][j]) + Rx[i][2] * Ry[2][j];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = (A[i][0] * Rz[0][j] + A[i][1] * Rz[1][j]) + A[i][2] * Rz[2][j];
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) M16[4 * i + j] = R[i][j]; M16[4 * i + 3] = (float)p6[i]; }
  M16[12] = M16[13] = M16[14] = 0.0f; M16[15] = 1.0f;
}
// Matrix3f::eulerAngles(0, 1, 2) of the guess (ndt_omp_impl.hpp:109; Eigen 3.3's algorithm): R = Rx(e0) Ry(e1) Rz(e2)
void ndt_euler012(const float* M16, float* e3) {
  auto m = [&](int r, int c) { return M16[4 * r + c]; };
  const float pi = (float)M_PI;
  float r0 = std::atan2(m(1, 2), m(2, 2));
  const float c2 = std::sqrt(m(0, 0) * m(0, 0) + m(0, 1) * m(0, 1));
  float r1;
  if (r0 > 0.0f) { r0 -= pi; r1 = std::atan2(-m(0, 2), -c2); } else r1 = std::atan2(-m(0, 2), c2);
  const float s1 = std::sin(r0), c1 = std::cos(r0);
  const float r2 = std::atan2(s1 * m(2, 0) - c1 * m(1, 0), c1 * m(1, 1) - s1 * m(2, 1));
  e3[0] = -r0; e3[1] = -r1; e3[2] = -r2;
}
NdtMat ndt_mat12(const float* M16) { NdtMat M; for (int i = 0; i < 12; ++i) M.m[i] = M16[i]; return M; }
// JacobiSVD<Matrix<double, 6, 6>>(H, FullU | FullV).solve(b) (:127-129): minimum-norm least-squares solution.  One-sided Jacobi (Hestenes): H V = U S with
// orthogonal columns; singular values below eps * 6 * max are treated as zero (Eigen's default threshold).
void ndt_svd_solve(const double* H36, const double* b6, double* x6) {
  double A[6][6], V[6][6];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { A[i][j] = H36[6 * i + j]; V[i][j] = i == j ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 5; ++p) for (int q = p + 1; q < 6; ++q) {
      double al = 0, be = 0, ga = 0;
      for (int k = 0; k < 6; ++k) { al += A[k][p] * A[k][p]; be += A[k][q] * A[k][q]; ga += A[k][p] * A[k][q]; }
      if (ga == 0.0 || al == 0.0 || be == 0.0) continue;
      off = std::max(off, std::fabs(ga) / std::sqrt(al * be));
      const double zeta = (be - al) / (2.0 * ga);
      const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
      const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
      for (int k = 0; k < 6; ++k) { const double ap = A[k][p], aq = A[k][q]; A[k][p] = cs * ap - sn * aq; A[k][q] = sn * ap + cs * aq; }
      for (int k = 0; k < 6; ++k) { const double vp = V[k][p], vq = V[k][q]; V[k][p] = cs * vp - sn * vq; V[k][q] = sn * vp + cs * vq; }
    }
    if (off < 1e-15) break;
  }
  double sv[6], smax = 0.0;
  for (int j = 0; j < 6; ++j) { double s2 = 0; for (int k = 0; k < 6; ++k) s2 += A[k][j] * A[k][j]; sv[j] = std::sqrt(s2); smax = std::max(smax, sv[j]); }
  const double thr = smax * 6.0 * 2.220446049250313e-16;
  for (int i = 0; i < 6; ++i) x6[i] = 0.0;
  for (int j = 0; j < 6; ++j) {
    if (!(sv[j] > thr)) continue;
    double ub = 0; for (int k = 0; k < 6; ++k) ub += A[k][j] * b6[k];   // (U^T b)_j * s_j
    const double w = ub / (sv[j] * sv[j]);
    for (int i = 0; i < 6; ++i) x6[i] += V[i][j] * w;
  }
}
struct NdtDev {   // device work area of one registration: the partial sums of the reductions, their ticket, the result; d_up[2] = source cloud
  double* part; int* ticket; double* out; int blocks;
};
int ndt_workspace(lvx_ctx* c, int n, NdtDev* W) {
  const int blocks = std::max(1, (n + 255) / 256);
  int rc;
  if ((rc = dev_alloc(c, c->d_up[6], ((size_t)blocks * 43 + 64) * 8))) return rc;
  W->blocks = blocks; W->out = (double*)c->d_up[6].p; W->ticket = (int*)(W->out + 48); W->part = W->out + 64;
  LVX_HIP(c, hipMemsetAsync(W->out, 0, 64 * 8, c->stream));
  return LVX_OK;
}
struct NdtGridArgs { VxGrid g; const int* cells; const int* leaf_n; const double* mean; const double* icov; float leaf; int min_pts; };
NdtGridArgs ndt_grid(const lvx_ctx* c) {
  const lvx_ctx::Voxels& V = c->vox;
  NdtGridArgs G; std::memcpy(&G.g, &V.grid, sizeof(G.g));
  const size_t nl = (size_t)V.cap;   // leaf arrays are strided by the capacity
  G.cells = (const int*)V.cells.p; G.leaf_n = (const int*)V.leaf_i.p + nl; G.mean = (const double*)V.leaf_d.p; G.icov = G.mean + 12 * nl; G.leaf = V.leaf; G.min_pts = V.min_pts;
  return G;
}
// one computeDerivatives on the device: score, gradient, Hessian at p6 (transform M16 applied in the kernel, or the caller's transformed cloud trn_d)
int ndt_eval(lvx_ctx* c, const NdtDev& W, int n, const float4* src_d, const float4* trn_d, const float* M16, const double* p6, double outlier_ratio, int search, int compute_hessian, double* h43) {
  const NdtGridArgs G = ndt_grid(c);
  const NdtTables T = ndt_tables(p6, (double)G.leaf, outlier_ratio);
  const NdtMat M = ndt_mat12(M16);
  const dim3 grid((unsigned)W.blocks), blk(256);
  { ProfScope ps(c, LVX_KERNEL_UPSTREAM);
#define LVX_NDT_LAUNCH(NB, XF) hipLaunchKernelGGL((k_ndt_derivatives<NB, XF>), grid, blk, 0, c->stream, src_d, trn_d, M, n, G.leaf, G.min_pts, G.g, G.cells, G.leaf_n, G.mean, G.icov, T.f, compute_hessian, W.part, W.ticket, W.out)
    if (trn_d) { if (search == 1) LVX_NDT_LAUNCH(1, false); else if (search == 26) LVX_NDT_LAUNCH(26, false); else LVX_NDT_LAUNCH(7, false); }
    else { if (search == 1) LVX_NDT_LAUNCH(1, true); else if (search == 26) LVX_NDT_LAUNCH(26, true); else LVX_NDT_LAUNCH(7, true); }
#undef LVX_NDT_LAUNCH
  }
  LVX_HIP(c, hipGetLastError());
  LVX_HIP(c, hipMemcpyAsync(h43, W.out, 43 * 8, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}
int ndt_hessian(lvx_ctx* c, const NdtDev& W, int n, const float4* src_d, const float* M16, const double* p6, double outlier_ratio, int search, double* h36) {
  const NdtGridArgs G = ndt_grid(c);
  const NdtTables T = ndt_tables(p6, (double)G.leaf, outlier_ratio);
  const NdtMat M = ndt_mat12(M16);
  const dim3 grid((unsigned)W.blocks), blk(256);
  { ProfScope ps(c, LVX_KERNEL_UPSTREAM);
#define LVX_NDT_LAUNCH(NB) hipLaunchKernelGGL((k_ndt_hessian<NB>), grid, blk, 0, c->stream, src_d, M, n, G.leaf, G.min_pts, G.g, G.cells, G.leaf_n, G.mean, G.icov, T.d, W.part, W.ticket, W.out)
    if (search == 1) LVX_NDT_LAUNCH(1); else if (search == 26) LVX_NDT_LAUNCH(26); else LVX_NDT_LAUNCH(7);
#undef LVX_NDT_LAUNCH
  }
  LVX_HIP(c, hipGetLastError());
  LVX_HIP(c, hipMemcpyAsync(h36, W.out, 36 * 8, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}
// updateIntervalMT (:648-685)
bool ndt_update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
  if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  return true;
}
// trialValueSelectionMT (:689-768)
double ndt_trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return std::fabs(a_c - a_l) < std::fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * g_l < 0) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return std::fabs(a_c - a_t) >= std::fabs(a_s - a_t) ? a_c : a_s;
  }
  if (std::fabs(g_t) <= std::fabs(g_l)) {
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = std::sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double a_next = std::fabs(a_c - a_t) < std::fabs(a_s - a_t) ? a_c : a_s;
    return a_t > a_l ? std::min(a_t + 0.66 * (a_u - a_t), a_next) : std::max(a_t + 0.66 * (a_u - a_t), a_next);
  }
  const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u, w = std::sqrt(z * z - g_t * g_u);
  return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}
struct NdtState { double score, grad[6], hess[36]; float final[16]; int mt_iterations, n_eval; };
double dot6(const double* a, const double* b) { double s = 0; for (int i = 0; i < 6; ++i) s += a[i] * b[i]; return s; }
// computeStepLengthMT (:772-931); returns the step length through *a_out
int ndt_step_length(lvx_ctx* c, const NdtDev& W, int n, const float4* src_d, const lvx_ndt_options& o, const double* x, double* step_dir, double step_init, double step_max, double step_min,
                    NdtState& st, double* a_out) {
  const double phi_0 = -st.score;
  double d_phi_0 = -dot6(st.grad, step_dir);
  st.mt_iterations = 0;
  if (d_phi_0 >= 0) {
    if (d_phi_0 == 0) { *a_out = 0; return LVX_OK; }
    d_phi_0 *= -1;
    for (int i = 0; i < 6; ++i) step_dir[i] *= -1;
  }
  const int max_step_iterations = 10;
  int step_iterations = 0;
  const double mu = 1.e-4, nu = 0.9;
  double a_l = 0, a_u = 0;
  double f_l = phi_0 - phi_0 - mu * d_phi_0 * a_l, g_l = d_phi_0 - mu * d_phi_0;   // auxilaryFunction_PsiMT / _dPsiMT (ndt_omp.h:430-446)
  double f_u = phi_0 - phi_0 - mu * d_phi_0 * a_u, g_u = d_phi_0 - mu * d_phi_0;
  bool interval_converged = (step_max - step_min) < 0, open_interval = true;
  double a_t = step_init;
  a_t = std::min(a_t, step_max);
  a_t = std::max(a_t, step_min);
  double x_t[6], h43[43];
  for (int i = 0; i < 6; ++i) x_t[i] = x[i] + step_dir[i] * a_t;
  ndt_matrix(x_t, st.final);
  int rc;
  if ((rc = ndt_eval(c, W, n, src_d, nullptr, st.final, x_t, o.outlier_ratio, o.search, 1, h43))) return rc;
  ++st.n_eval;
  st.score = h43[0]; std::memcpy(st.grad, h43 + 1, 48); std::memcpy(st.hess, h43 + 7, 288);
  double phi_t = -st.score, d_phi_t = -dot6(st.grad, step_dir);
  double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t, d_psi_t = d_phi_t - mu * d_phi_0;
  while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
    a_t = open_interval ? ndt_trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t) : ndt_trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    a_t = std::min(a_t, step_max);
    a_t = std::max(a_t, step_min);
    for (int i = 0; i < 6; ++i) x_t[i] = x[i] + step_dir[i] * a_t;
    ndt_matrix(x_t, st.final);
    if ((rc = ndt_eval(c, W, n, src_d, nullptr, st.final, x_t, o.outlier_ratio, o.search, 0, h43))) return rc;   // score and gradient only; the Hessian comes back zero (:187)
    ++st.n_eval;
    st.score = h43[0]; std::memcpy(st.grad, h43 + 1, 48); std::memcpy(st.hess, h43 + 7, 288);
    phi_t = -st.score; d_phi_t = -dot6(st.grad, step_dir);
    psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t; d_psi_t = d_phi_t - mu * d_phi_0;
    if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
      open_interval = false;
      f_l = f_l + phi_0 - mu * d_phi_0 * a_l; g_l = g_l + mu * d_phi_0;
      f_u = f_u + phi_0 - mu * d_phi_0 * a_u; g_u = g_u + mu * d_phi_0;
    }
    interval_converged = open_interval ? ndt_update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t) : ndt_update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    step_iterations++;
  }
  if (step_iterations) { if ((rc = ndt_hessian(c, W, n, src_d, st.final, x_t, o.outlier_ratio, o.search, st.hess))) return rc; }   // :927-928
  st.mt_iterations = step_iterations;
  *a_out = a_t;
  return LVX_OK;
}
}  // namespace

extern "C" {

int lvx_ndt_default_options(lvx_ndt_options* o) {
  if (!o) return LVX_E_ARG;
  o->step_size = 0.1; o->outlier_ratio = 0.55; o->transformation_epsilon = 0.1; o->max_iterations = 35; o->search = 7;   // ndt_omp_impl.hpp:46-76
  return LVX_OK;
}

int lvx_ndt_derivatives(lvx_ctx* c, int n, const float* input_xyzi4, const float* trans_xyzi4, const double* p6, double outlier_ratio, int compute_hessian,
                        double* score, double* gradient6, double* hessian36) {
  if (!c || n < 0 || !p6 || !score || !gradient6 || (compute_hessian && !hessian36) || (n > 0 && (!input_xyzi4 || !trans_xyzi4))) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  { const int rc0 = vox_info(c); if (rc0) return rc0; }
  const lvx_ctx::Voxels& V = c->vox;
  *score = 0.0;
  for (int j = 0; j < 6; ++j) gradient6[j] = 0.0;
  if (hessian36) for (int e = 0; e < 36; ++e) hessian36[e] = 0.0;
  if (n == 0 || V.n_leaves == 0) return LVX_OK;
  int rc;
  if ((rc = upload(c, c->d_up[2], input_xyzi4, (size_t)n * 16))) return rc;
  if ((rc = upload(c, c->d_up[3], trans_xyzi4, (size_t)n * 16))) return rc;
  NdtDev W;
  if ((rc = ndt_workspace(c, n, &W))) return rc;
  const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double h[43];
  if ((rc = ndt_eval(c, W, n, (const float4*)c->d_up[2].p, (const float4*)c->d_up[3].p, I16, p6, outlier_ratio, 7, compute_hessian, h))) return rc;
  *score = h[0];
  for (int j = 0; j < 6; ++j) gradient6[j] = h[1 + j];
  if (hessian36) for (int e = 0; e < 36; ++e) hessian36[e] = h[7 + e];
  return LVX_OK;
}

// pcl::Registration::align -> pclomp::NormalDistributionsTransform::computeTransformation (ndt_omp_impl.hpp:81-171) against the voxel grid of the last lvx_voxel_build
int lvx_ndt_align(lvx_ctx* c, int n, const float* src_xyzi4, const float* guess16, const lvx_ndt_options* opt, lvx_ndt_result* res, float* aligned_xyzi4) {
  if (!c || n < 0 || !res || (n > 0 && !src_xyzi4)) return LVX_E_ARG;
  lvx_ndt_options o; lvx_ndt_default_options(&o); if (opt) o = *opt;
  if (o.search != 1 && o.search != 7 && o.search != 26) return fail(c, LVX_E_ARG, "lvx_ndt_align: search must be 1 (DIRECT1), 7 (DIRECT7) or 26 (DIRECT26)");
  LVX_HIP(c, hipSetDevice(c->device));
  { const int rc0 = vox_info(c); if (rc0) return rc0; }
  if (c->vox.n_leaves == 0 || !c->vox.cells.p) return fail(c, LVX_E_STATE, "lvx_ndt_align needs a target: call lvx_voxel_build first");
  std::memset(res, 0, sizeof(*res));
  if (n == 0) {   // no points: gradient and Hessian are zero, the Newton step has norm 0 and the loop leaves converged before its first iteration (:134-139)
    for (int i = 0; i < 4; ++i) res->final_transformation[5 * i] = 1.0f;
    if (guess16) std::memcpy(res->final_transformation, guess16, 64);
    float e0[3]; ndt_euler012(res->final_transformation, e0);
    res->p6[0] = res->final_transformation[3]; res->p6[1] = res->final_transformation[7]; res->p6[2] = res->final_transformation[11]; res->p6[3] = e0[0]; res->p6[4] = e0[1]; res->p6[5] = e0[2];
    res->converged = 1;
    return LVX_OK;
  }
  int rc;
  if ((rc = upload(c, c->d_up[2], src_xyzi4, (size_t)n * 16))) return rc;
  const float4* src_d = (const float4*)c->d_up[2].p;
  NdtDev W;
  if ((rc = ndt_workspace(c, n, &W))) return rc;
  NdtState st{};
  const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::memcpy(st.final, I16, sizeof(I16));                                    // Registration::align: final_transformation_ = Identity
  if (guess16 && std::memcmp(guess16, I16, sizeof(I16)) != 0) std::memcpy(st.final, guess16, sizeof(I16));   // :95-101
  float e3[3]; ndt_euler012(st.final, e3);
  double p[6] = {st.final[3], st.final[7], st.final[11], e3[0], e3[1], e3[2]}, h43[43];                      // :106-111
  if ((rc = ndt_eval(c, W, n, src_d, nullptr, st.final, p, o.outlier_ratio, o.search, 1, h43))) return rc;   // :119
  st.n_eval = 1;
  st.score = h43[0]; std::memcpy(st.grad, h43 + 1, 48); std::memcpy(st.hess, h43 + 7, 288);
  int nr_iterations = 0; bool converged = false;
  while (!converged) {
    double delta_p[6], mg[6];
    for (int i = 0; i < 6; ++i) mg[i] = -st.grad[i];
    ndt_svd_solve(st.hess, mg, delta_p);                                                                     // :127-129
    double delta_p_norm = std::sqrt(dot6(delta_p, delta_p));
    if (delta_p_norm == 0 || delta_p_norm != delta_p_norm) { converged = delta_p_norm == delta_p_norm; break; }   // :134-139
    for (int i = 0; i < 6; ++i) delta_p[i] /= delta_p_norm;
    if ((rc = ndt_step_length(c, W, n, src_d, o, p, delta_p, delta_p_norm, o.step_size, o.transformation_epsilon / 2, st, &delta_p_norm))) return rc;
    for (int i = 0; i < 6; ++i) p[i] += delta_p[i] * delta_p_norm;                                           // :143, 152
    if (nr_iterations > o.max_iterations || (nr_iterations && std::fabs(delta_p_norm) < o.transformation_epsilon)) converged = true;   // :158-162
    nr_iterations++;
  }
  std::memcpy(res->final_transformation, st.final, sizeof(st.final));
  for (int i = 0; i < 6; ++i) res->p6[i] = p[i];
  res->iterations = nr_iterations; res->converged = converged ? 1 : 0; res->n_evaluations = st.n_eval;
  res->score = st.score; res->trans_probability = n > 0 ? st.score / (double)n : 0.0;                       // :170
  if (aligned_xyzi4 && n > 0) {   // the output cloud of align(): the source under the final transformation (:832, 877)
    if ((rc = dev_alloc(c, c->d_up[3], (size_t)n * 16))) return rc;
    hipLaunchKernelGGL(k_ndt_transform, dim3((n + 255) / 256), dim3(256), 0, c->stream, src_d, n, ndt_mat12(st.final), (float4*)c->d_up[3].p);
    LVX_HIP(c, hipGetLastError());
    LVX_HIP(c, hipMemcpyAsync(aligned_xyzi4, c->d_up[3].p, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
    LVX_HIP(c, hipStreamSynchronize(c->stream));
  }
  return LVX_OK;
}

// pcl::Registration::getFitnessScore(max_range) (apps/align.cpp:30)
int lvx_ndt_fitness(lvx_ctx* c, int n_src, const float* src_xyzi4, const float* transform16, int n_tgt, const float* tgt_xyzi4, double max_range, double* fitness) {
  if (!c || !fitness || !transform16 || n_src < 0 || n_tgt < 0 || (n_src > 0 && !src_xyzi4) || (n_tgt > 0 && !tgt_xyzi4)) return LVX_E_ARG;
  *fitness = std::numeric_limits<double>::max();
  if (n_src == 0 || n_tgt == 0) return LVX_OK;
  LVX_HIP(c, hipSetDevice(c->device));
  int rc;
  if ((rc = upload(c, c->d_up[2], src_xyzi4, (size_t)n_src * 16))) return rc;
  if ((rc = upload(c, c->d_up[3], tgt_xyzi4, (size_t)n_tgt * 16))) return rc;
  if ((rc = dev_alloc(c, c->d_up[6], (size_t)n_src * 4))) return rc;
  const int bx = (n_src + 255) / 256;
  int ny = std::max(1, std::min(64, 2048 / std::max(bx, 1)));           // ~2 k workgroups: the target cloud split into ny slices per source tile
  const int per_y = (((n_tgt + ny - 1) / ny) + 1023) / 1024 * 1024;
  ny = (n_tgt + per_y - 1) / per_y;
  { ProfScope ps(c, LVX_KERNEL_UPSTREAM);
    LVX_HIP(c, hipMemsetD32Async((hipDeviceptr_t)c->d_up[6].p, 0x7f7fffff, (size_t)n_src, c->stream));   // FLT_MAX
    hipLaunchKernelGGL(k_ndt_fitness, dim3((unsigned)bx, (unsigned)ny), dim3(256), 0, c->stream, (const float4*)c->d_up[2].p, n_src, ndt_mat12(transform16), (const float4*)c->d_up[3].p, n_tgt, per_y,
                       (unsigned*)c->d_up[6].p); }
  LVX_HIP(c, hipGetLastError());
  std::vector<float> best((size_t)n_src);
  LVX_HIP(c, hipMemcpyAsync(best.data(), c->d_up[6].p, (size_t)n_src * 4, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  double sum = 0.0; int nr = 0;
  for (int i = 0; i < n_src; ++i) if (best[i] <= max_range) { sum += best[i]; ++nr; }
  if (nr > 0) *fitness = sum / nr;
  return LVX_OK;
}

// VoxelGridCovariance::getNeighborhoodAtPoint(relative_coordinates, reference_point, neighbors) (voxel_grid_covariance_omp_impl.hpp:378-408)
int lvx_voxel_lookup_rel(lvx_ctx* c, int nq, const float* xyzi4, int n_rel, const int32_t* rel3, int32_t* leaf_ids) {
  if (!c || nq < 0 || n_rel < 0 || (nq > 0 && n_rel > 0 && (!xyzi4 || !rel3 || !leaf_ids))) return LVX_E_ARG;
  if (nq == 0 || n_rel == 0) return LVX_OK;
  LVX_HIP(c, hipSetDevice(c->device));
  { const int rc0 = vox_info(c); if (rc0) return rc0; }
  const lvx_ctx::Voxels& V = c->vox;
  const size_t ne = (size_t)nq * n_rel;
  if (!V.cells.p || V.n_leaves == 0) { for (size_t e = 0; e < ne; ++e) leaf_ids[e] = -1; return LVX_OK; }
  int rc;
  if ((rc = upload(c, c->d_up[2], xyzi4, (size_t)nq * 16))) return rc;
  if ((rc = upload(c, c->d_up[6], rel3, (size_t)n_rel * 12))) return rc;
  if ((rc = dev_alloc(c, c->d_up[3], ne * 4))) return rc;
  const NdtGridArgs G = ndt_grid(c);
  { ProfScope ps(c, LVX_KERNEL_UPSTREAM);
    hipLaunchKernelGGL(k_vx_lookup_rel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, c->stream, (const float4*)c->d_up[2].p, nq, G.leaf, G.min_pts, G.g, G.cells, G.leaf_n, n_rel,
                       (const int*)c->d_up[6].p, (int*)c->d_up[3].p); }
  LVX_HIP(c, hipGetLastError());
  LVX_HIP(c, hipMemcpyAsync(leaf_ids, c->d_up[3].p, ne * 4, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}

int lvx_scan_less_flat_downsample_sweep(lvx_ctx* c, int sweep, float leaf, int max_out, float* out_xyzi4, int32_t* ring_counts, int32_t* n_out) {
  if (!c || !n_out || !(leaf > 0) || max_out < 0 || (max_out > 0 && !out_xyzi4)) return LVX_E_ARG;
  *n_out = 0;
  if (c->sr_S <= 0 || !c->d_up[0].p) return fail(c, LVX_E_STATE, "lvx_scan_register has not been called");
  if (sweep < 0 || sweep >= c->sr_S) return fail(c, LVX_E_ARG, "sweep index outside the last batch");
  LVX_HIP(c, hipSetDevice(c->device));
  const int n_rings = c->sr_rings, m = c->sr_m[sweep];
  const size_t o0 = (size_t)c->sr_off[sweep];
  const int n = c->sr_off[sweep + 1] - c->sr_off[sweep];
  if (ring_counts) for (int r = 0; r < n_rings; ++r) ring_counts[r] = 0;
  if (m == 0) return LVX_OK;
  char* base = (char*)c->d_up[0].p;   // scratch layout of the batch (scan_register_batch)
  const float4* d_cloud = (const float4*)(base + c->sr_batch_off[0]) + o0; const int* d_lflat_r = (const int*)(base + c->sr_batch_off[1]) + o0;
  const int* d_ss = (const int*)(base + c->sr_batch_off[2]) + (size_t)sweep * n_rings; const int* d_cnt = (const int*)(base + c->sr_batch_off[3]) + (size_t)sweep * n_rings * 4;
  int rc;
  if ((rc = dev_alloc(c, c->d_up[4], (size_t)n * 16))) return rc;
  if ((rc = dev_alloc(c, c->d_up[5], (size_t)(n_rings + 1) * 4))) return rc;
  if (c->assoc_map_planes == (const double*)c->d_up[5].p) lvx_surfel_map_release(c);
  int* d_oc = (int*)c->d_up[5].p; int* d_err = d_oc + n_rings;
  LVX_HIP(c, hipMemsetAsync(d_err, 0, 4, c->stream));
  { ProfScope ps(c, LVX_KERNEL_UPSTREAM);
    hipLaunchKernelGGL(k_sr_voxelgrid, dim3(n_rings), dim3(256), 0, c->stream, d_cloud, d_ss, d_cnt, d_lflat_r, leaf, (float4*)c->d_up[4].p, d_oc, d_err); }
  LVX_HIP(c, hipGetLastError());
  std::vector<int> oc(n_rings + 1), ss(n_rings);
  LVX_HIP(c, hipMemcpyAsync(oc.data(), d_oc, (size_t)(n_rings + 1) * 4, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipMemcpyAsync(ss.data(), d_ss, (size_t)n_rings * 4, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  if (oc[n_rings] & 16) return fail(c, LVX_E_ARG, "more less-flat points in one ring than the LDS sort capacity (4096)");
  int total = 0;
  for (int r = 0; r < n_rings; ++r) {
    if (ring_counts) ring_counts[r] = oc[r];
    const int room = std::max(0, std::min(oc[r], max_out - total));
    if (room > 0) LVX_HIP(c, hipMemcpy(out_xyzi4 + 4 * (size_t)total, (const float4*)c->d_up[4].p + (ss[r] - 5), (size_t)room * 16, hipMemcpyDeviceToHost));
    total += oc[r];
  }
  *n_out = total;
  return LVX_OK;
}
int lvx_scan_less_flat_downsample(lvx_ctx* c, float leaf, int max_out, float* out_xyzi4, int32_t* ring_counts, int32_t* n_out) {
  return lvx_scan_less_flat_downsample_sweep(c, 0, leaf, max_out, out_xyzi4, ring_counts, n_out);
}

}  // extern "C"
