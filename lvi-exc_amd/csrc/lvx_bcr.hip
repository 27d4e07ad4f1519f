// lvx_bcr.hip — block cyclic reduction (nested-dissection Cholesky) of the banded SPD part of the LM system.
//
// The band (n_band x n_band, half-bandwidth bw) is viewed as block tridiagonal with block size b >= bw; the chain of
// n_blk blocks is eliminated in log2(n_blk) levels: at level l the blocks j = 2^l - 1 + 2^(l+1) k are eliminated in parallel,
//     C_j C_j^T = D_j,   X+_k = A_{j+s,j} C_j^-T,   Y_k = C_j^-1 A_{j,j-s},
//     D_{j+s} -= X+ X+^T,   D_{j-s} -= Y^T Y,   A_{j+s,j-s} = -X+ Y        (s = 2^l)
// which replaces a 155 k-long sequential dependency chain by ~10 rounds of BATCHED dense b x b operations — own MFMA kernels for blocks up to 208 wide (Cholesky,
// triangular solves, Schur updates: below); wider blocks go to rocSOLVER potrf_strided_batched / rocBLAS gemm_strided_batched, loaded on demand (vendor_blas).
// This is what Ceres' SPARSE_SCHUR + sparse Cholesky does for the reference (kontiki/trajectory_estimator.h:44), restructured for a GPU.
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "lvx_ctx.h"
#include "lvx_chol16.h"

namespace lvx {

#define LVX_BLAS(ctx, expr)                                                                                     \
  do {                                                                                                          \
    rocblas_status s_ = (expr);                                                                                 \
    if (s_ != rocblas_status_success) return fail(ctx, LVX_E_HIP, std::string(#expr) + ": rocblas status " + std::to_string((int)s_)); \
  } while (0)

// rocBLAS / rocSOLVER serve ONLY bands wider than 208 (free time offsets with very wide co-visibility; config 4 and every reference stage run on the kernels below).  They are
// loaded the first time such a band is factorised (dlopen, as librccl is): liblvx.so does not link them, a host without them keeps everything but that case
// (the library hard-linked librocblas / librocsolver and through them hipblaslt / rocroller for a path the headline never takes).
struct VendorBlas {
  void* hb = nullptr; void* hs = nullptr; bool tried = false;
  decltype(&rocblas_create_handle) create_handle = nullptr;
  decltype(&rocblas_destroy_handle) destroy_handle = nullptr;
  decltype(&rocblas_set_stream) set_stream = nullptr;
  decltype(&rocblas_set_pointer_mode) set_pointer_mode = nullptr;
  decltype(&rocblas_dgemm_strided_batched) dgemm_strided_batched = nullptr;
  decltype(&rocsolver_dpotrf_strided_batched) dpotrf_strided_batched = nullptr;
  bool ok() const { return create_handle && destroy_handle && set_stream && set_pointer_mode && dgemm_strided_batched && dpotrf_strided_batched; }
};
static VendorBlas g_vb;
static std::mutex g_vb_mu;
static int vendor_blas(lvx_ctx* c) {
  std::lock_guard<std::mutex> lk(g_vb_mu);
  if (!g_vb.tried) {
    g_vb.tried = true;
    for (const char* n : {"librocblas.so", "librocblas.so.5", "librocblas.so.4", "/opt/rocm/lib/librocblas.so"}) if ((g_vb.hb = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    for (const char* n : {"librocsolver.so", "librocsolver.so.0", "/opt/rocm/lib/librocsolver.so"}) if ((g_vb.hs = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (g_vb.hb && g_vb.hs) {
      g_vb.create_handle = (decltype(g_vb.create_handle))dlsym(g_vb.hb, "rocblas_create_handle");
      g_vb.destroy_handle = (decltype(g_vb.destroy_handle))dlsym(g_vb.hb, "rocblas_destroy_handle");
      g_vb.set_stream = (decltype(g_vb.set_stream))dlsym(g_vb.hb, "rocblas_set_stream");
      g_vb.set_pointer_mode = (decltype(g_vb.set_pointer_mode))dlsym(g_vb.hb, "rocblas_set_pointer_mode");
      g_vb.dgemm_strided_batched = (decltype(g_vb.dgemm_strided_batched))dlsym(g_vb.hb, "rocblas_dgemm_strided_batched");
      g_vb.dpotrf_strided_batched = (decltype(g_vb.dpotrf_strided_batched))dlsym(g_vb.hs, "rocsolver_dpotrf_strided_batched");
    }
  }
  if (!g_vb.ok()) return fail(c, LVX_E_STATE, "band wider than 208 needs rocBLAS / rocSOLVER, which could not be loaded (librocblas.so, librocsolver.so)");
  return LVX_OK;
}

// band (scaled + damped) -> dense blocks.  D_i lower triangle (column-major b x b), G0_i = A_{i+1,i}; padding blocks are identity / zero
__global__ __launch_bounds__(256) void k_bcr_build(const double* __restrict__ Hb, const double* __restrict__ scale, const double* __restrict__ lmd, double inv_radius,
                                                   int nb, int bw, int b, int nblk, double* D, double* G0) {
  // blockIdx.x = column cc of the block, blockIdx.y = block i, blockIdx.z = 0: D_i, 1: G0_i = A_{i+1,i}; threads = rows.  (One thread per element of a flat index
  // spent most of its time in three 64-bit divisions per element: 240 us for 430 MB.)
  const int cc = blockIdx.x * 4 + (threadIdx.x >> 6), i = blockIdx.y;   // a wavefront per column, four columns per workgroup
  if (cc >= b) return;
  const bool isG = blockIdx.z != 0;
  const size_t bb = (size_t)b * b;
  const long long c = (long long)i * b + cc;
  const long long r0 = (long long)(isG ? i + 1 : i) * b;
  double* out = (isG ? G0 : D) + (size_t)i * bb + (size_t)cc * b;   // column-major
  const double* col = c < nb ? Hb + (size_t)c * (bw + 1) : nullptr;
  const double sc = c < nb ? scale[c] : 0.0;
  for (int rr = threadIdx.x & 63; rr < b; rr += 64) {
    const long long r = r0 + rr;
    double v = 0.0;
    if (!isG && rr < cc) continue;   // the strict upper triangle of D is never read (Cholesky, triangular solves and Schur updates work on the lower one): 108 MB less to write
    if (!isG) {
      if (rr >= cc) {
        if (r < nb) { const long long d = r - c; if (d <= bw) { const double hv = col[d]; v = hv * scale[r] * sc; if (d == 0) v = hv == 0.0 ? 1.0 : v + lmd[c] * inv_radius; } }   // untouched variable (zero row, zero gradient): any pivot gives y = 0; use 1 instead of 1e-6/radius
        else if (r == c) v = 1.0;
      }
    } else if (r < nb && c < nb) { const long long d = r - c; if (d <= bw) v = col[d] * scale[r] * sc; }
    out[rr] = v;
  }
}
__global__ void k_bcr_pad_identity(double* D, int b, int first, int nblk) {   // D_i = I for the padding blocks i in [first, nblk)
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x, bb = (size_t)b * b;
  if (e >= (size_t)(nblk - first) * bb) return;
  const size_t r = (e % bb) % b, cc = (e % bb) / b;
  D[(size_t)first * bb + e] = r == cc ? 1.0 : 0.0;
}
__global__ void k_bcr_info(const int* info, int n, int* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && info[i] != 0) atomicMax(out, 2000000000 + i);   // reported ahead of the dense border pivot codes (1e9 + k)
}

// rocBLAS handles are pooled per device: creating one costs ~17 ms, and the stage driver makes a new context per stage (34 ms of a 170 ms two-stage solve).  A context
// takes a handle at its first solve and hands it back when it is destroyed; contexts that live at the same time (one per thread in the joint solve) hold different ones.
#define LVX_BLAS_H(ctx, h) do { if (!(h)) { const int rh_ = bcr_handle((ctx), &(h)); if (rh_) return rh_; } } while (0)
static int bcr_handle(lvx_ctx* c, rocblas_handle* h);
static std::mutex g_blas_mu;
static std::vector<std::pair<int, rocblas_handle>> g_blas_free;   // (device, handle)
static int bcr_handle(lvx_ctx* c, rocblas_handle* h) {
  if (!c->blas) {
    rocblas_handle hh = nullptr;
    { std::lock_guard<std::mutex> lk(g_blas_mu);
      for (size_t i = 0; i < g_blas_free.size(); ++i) if (g_blas_free[i].first == c->device) { hh = g_blas_free[i].second; g_blas_free.erase(g_blas_free.begin() + (long)i); break; } }
    if (!hh) { const int rv_ = vendor_blas(c); if (rv_) return rv_; LVX_BLAS(c, g_vb.create_handle(&hh)); }
    c->blas = hh;
  }
  *h = (rocblas_handle)c->blas;
  { const int rv_ = vendor_blas(c); if (rv_) return rv_; }
  LVX_BLAS(c, g_vb.set_stream(*h, c->stream));
  LVX_BLAS(c, g_vb.set_pointer_mode(*h, rocblas_pointer_mode_host));
  return LVX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Batched multi-vector triangular solve with a dense lower factor L (b x b, column-major): in place L w = v (TRANS = false) or
// L^T w = v (TRANS = true) for `nvec` vectors per batch element.  Vector k, element i lives at V[k * sv + i * se].
// A workgroup handles 64 vectors of one batch element, each of its 4 wavefronts 16 of them; L streams through LDS in 16-column panels
// (register double-buffered).  Per panel: the 16 x 16 diagonal triangle by substitution, everything else a rank-16 update on the FP64
// matrix cores (v_mfma_f64_16x16x4_f64):
//   forward : W[rows below, vectors] -= L[rows below, panel] w[panel, vectors]     one 16 x 16 tile per 16 rows below the panel
//   backward: v[panel, vectors]      -= L[rows below, panel]^T w[rows below]       accumulator tiles, k-steps over the rows below
// (rocBLAS' strided-batched TRSM turns into thousands of tiny launches at b ~ 200, and explicit inverses of the blocks lose positive
// definiteness of the Schur complements on weakly constrained problems, so this step is hand-written; no inverse is formed.)
// ---------------------------------------------------------------------------------------------------------
typedef double d4 __attribute__((ext_vector_type(4)));
// The 64 vectors of a workgroup never touch LDS: a wavefront keeps its 16 vectors as NT accumulator tiles
// (tile t = rows 16 t .. 16 t + 15, MFMA C layout: col = lane & 15 = vector, row = (lane >> 4) + 4 reg).  The B fragment of k-step ks of a
// solved tile is exactly its register ks (row (lane >> 4) + 4 ks), so the rank-16 updates read their right-hand operand from registers; only
// the 16-column panel of L (shared by the 4 wavefronts) lives in LDS (27 KB; the predecessor kept the vectors in LDS, 135 KB).  NT is a compile-time bound on ceil(b / 16); everything is unrolled so that no
// tile is indexed dynamically.
// blockIdx.z = 1, 2 select further, independent problem sets with the same b (p2, p3): the X+ and Y solves of a BCR level and the forward substitution of
// the right-hand sides against the same factors share a launch — on the lower levels each of them is one latency-bound wave of workgroups (68 us), side by
// side they cost it once
// DINV: the Cholesky kernel left the INVERSES of the factor's 16 x 16 diagonal triangles (k_potrf_batched: LI[block][panel][16][16], row-major, zero above the
// diagonal): a panel's triangular solve is then x = inv(L_pp) w — four MFMAs — instead of a 16-step substitution chain of LDS reads, lane broadcasts and FMAs (2.4 k of
// the ~5 k cycles a panel costs).  Only the 16 x 16 triangles are inverted: they are the Cholesky factors of diagonal blocks of a Jacobi-scaled SPD matrix, far from
// the conditioning of a whole b x b block (inverting THOSE lost positive definiteness of the Schur complements).
struct TrsmSet { const double* Lm; long long strideL; double* V; long long se, sv, strideV; int nvec, batch, batch0; const double* LI; long long strideLI; };   // batch0: batch count of the FIRST set (the grid covers the larger of the two)
template <bool TRANS, int NT, bool DINV>
__global__ __launch_bounds__(256, 2) void k_trsm_reg(const double* __restrict__ Lm, int b, long long strideL, double* V, long long se, long long sv, long long strideV, int nvec, const double* __restrict__ LIm, long long strideLI,
                                                     TrsmSet p2, TrsmSet p3) {
  if (blockIdx.z == 2) { Lm = p3.Lm; strideL = p3.strideL; V = p3.V; se = p3.se; sv = p3.sv; strideV = p3.strideV; nvec = p3.nvec; LIm = p3.LI; strideLI = p3.strideLI; if ((int)blockIdx.y >= p3.batch) return; }
  else if (blockIdx.z == 1) { Lm = p2.Lm; strideL = p2.strideL; V = p2.V; se = p2.se; sv = p2.sv; strideV = p2.strideV; nvec = p2.nvec; LIm = p2.LI; strideLI = p2.strideLI; if ((int)blockIdx.y >= p2.batch) return; }
  else if ((int)blockIdx.y >= p2.batch0) return;
  if ((int)(blockIdx.x * 64) >= nvec) return;
  extern __shared__ double lds[];
  constexpr int bp = 16 * NT, PS = bp | 1;
  double* P = lds;                         // [16][PS]
  double* dinv = P + 16 * PS;              // [16]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fk = lane >> 4, fi = lane & 15;
  const int vec = blockIdx.x * 64 + wv * 16 + fi;
  const bool vact = vec < nvec;
  const double* L = Lm + (size_t)blockIdx.y * strideL;
  const double* LI = DINV ? LIm + (size_t)blockIdx.y * strideLI : nullptr;
  double* v0 = V + (size_t)blockIdx.y * strideV + (size_t)vec * sv;
  d4 W[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) { const int i = 16 * t + fk + 4 * v; W[t][v] = (vact && i < b) ? v0[(size_t)i * se] : 0.0; }
  constexpr int NPRE = (16 * bp + 255) / 256;
  double pre[NPRE];
  double pre_li = 0.0;   // DINV: one entry of inv(L_pp) per thread on its way to LDS (Mi[16][17]); a lane's A-operand entries are [fi][4 ks + fk] (forward) / [4 ks + fk][fi] (transposed)
  double* Mi = dinv + 16;
  auto fetch = [&](int k0) {
    if (DINV) pre_li = LI[(size_t)(k0 >> 4) * 256 + tid];
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
      const int e = tid + 256 * j;
      double v = 0.0;
      if (e < 16 * bp) {
        const int kk = e / bp, i = e - kk * bp, k = k0 + kk;
        if (k < b) { if (i >= k && i < b) v = L[(size_t)k * b + i]; } else if (i == k) v = 1.0;
      }
      pre[j] = v;
    }
  };
  auto commit = [&](int k0) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
      const int e = tid + 256 * j;
      if (e < 16 * bp) {
        const int kk = e / bp, i = e - kk * bp;
        P[kk * PS + i] = pre[j];
        if (!DINV && i == k0 + kk) dinv[kk] = 1.0 / pre[j];
      }
    }
    if (DINV) Mi[(tid >> 4) * 17 + (tid & 15)] = pre_li;
    __syncthreads();
  };
  if (!TRANS) {
    fetch(0);
#pragma clang loop unroll(full)
    for (int p = 0; p < NT; ++p) {
      const int k0 = 16 * p;
      if (k0 < b) {                         // uniform
        commit(k0);
        if (k0 + 16 < b) fetch(k0 + 16);
        if (DINV) {   // w_p <- inv(L_pp) w_p on the matrix cores
          d4 X = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) X = __builtin_amdgcn_mfma_f64_16x16x4f64(Mi[fi * 17 + 4 * ks + fk], W[p][ks], X, 0, 0, 0);
          W[p] = X;
        } else
        {   // 16 x 16 triangular solve in the accumulator layout: the 16 rows of a vector sit in 4 lanes (16 apart) x 4 registers; the
            // solved entry is broadcast to the vector's other 3 lanes and every lane updates its own 4 rows
          double lq[4];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
#pragma unroll
            for (int v = 0; v < 4; ++v) lq[v] = P[q * PS + k0 + fk + 4 * v];           // L[k0 + row][k0 + q], rows of this lane
            const double cand = W[p][q >> 2] * dinv[q];
            const double xq = __shfl(cand, ((q & 3) << 4) + fi);
#pragma unroll
            for (int v = 0; v < 4; ++v) { const int r = fk + 4 * v; W[p][v] = r == q ? xq : (r > q ? W[p][v] - lq[v] * xq : W[p][v]); }
          }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int t = p + 1; t < NT; ++t)
            W[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-P[(ks * 4 + fk) * PS + 16 * t + fi], W[p][ks], W[t], 0, 0, 0);
      }
    }
  } else {
    const int plast = (b - 1) >> 4;
    fetch(16 * plast);
#pragma clang loop unroll(full)
    for (int p = NT - 1; p >= 0; --p) {
      const int k0 = 16 * p;
      if (p <= plast) {                     // uniform
        commit(k0);
        if (p > 0) fetch(k0 - 16);
        d4 Ca = W[p], Cb = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = p + 1; t < NT; ++t)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const double av = -P[fi * PS + 16 * t + 4 * ks + fk];
            if (((t - p) & 1) != 0) Ca = __builtin_amdgcn_mfma_f64_16x16x4f64(av, W[t][ks], Ca, 0, 0, 0);
            else Cb = __builtin_amdgcn_mfma_f64_16x16x4f64(av, W[t][ks], Cb, 0, 0, 0);
          }
        if (DINV) {   // w_p <- inv(L_pp)^T (v_p - sum)
          d4 X, Y = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int v = 0; v < 4; ++v) X[v] = Ca[v] + Cb[v];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(Mi[(4 * ks + fk) * 17 + fi], X[ks], Y, 0, 0, 0);
          W[p] = Y;
        } else
        {   // transposed 16 x 16 triangular solve, rows 15 .. 0, in the accumulator layout
          d4 X;
#pragma unroll
          for (int v = 0; v < 4; ++v) X[v] = Ca[v] + Cb[v];
          double lq[4];
#pragma unroll
          for (int q = 15; q >= 0; --q) {
#pragma unroll
            for (int v = 0; v < 4; ++v) lq[v] = P[(fk + 4 * v) * PS + k0 + q];         // L[k0 + q][k0 + row]
            const double cand = X[q >> 2] * dinv[q];
            const double xq = __shfl(cand, ((q & 3) << 4) + fi);
#pragma unroll
            for (int v = 0; v < 4; ++v) { const int r = fk + 4 * v; X[v] = r == q ? xq : (r < q ? X[v] - lq[v] * xq : X[v]); }
          }
          W[p] = X;
        }
      }
    }
  }
  if (vact) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) { const int i = 16 * t + fk + 4 * v; if (i < b) v0[(size_t)i * se] = W[t][v]; }
  }
}
template <bool TRANS, int NT, bool DINV>
static int launch_trsm_reg(lvx_ctx* c, const double* L, int b, long long strideL, double* V, long long se, long long sv, long long strideV, int nvec, int batch, const double* LI, long long strideLI,
                           const TrsmSet* second, const TrsmSet* third) {
  const size_t lds = ((size_t)16 * ((16 * NT) | 1) + 16 + 16 * 17) * 8;
  LVX_HIP(c, hipFuncSetAttribute((const void*)k_trsm_reg<TRANS, NT, DINV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  TrsmSet p2{}, p3{}; p2.batch0 = batch; unsigned gx = (unsigned)((nvec + 63) / 64), gy = (unsigned)batch, gz = 1;
  auto live = [](const TrsmSet* t) { return t && t->batch > 0 && t->nvec > 0; };
  if (live(second)) { p2 = *second; p2.batch0 = batch; gz = 2; }
  if (live(third)) { p3 = *third; gz = 3; }      // an empty second set with a live third: blocks of z = 1 exit at once (batch 0)
  for (const TrsmSet* t : {&p2, &p3}) if (t->batch > 0 && t->nvec > 0) { gx = std::max(gx, (unsigned)((t->nvec + 63) / 64)); gy = std::max(gy, (unsigned)t->batch); }
  hipLaunchKernelGGL((k_trsm_reg<TRANS, NT, DINV>), dim3(gx, gy, gz), dim3(256), lds, c->stream, L, b, strideL, V, se, sv, strideV, nvec, LI, strideLI, p2, p3);
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Forward solve with the factor RESIDENT IN LDS (the default when it fits: b <= 192 and the diagonal-triangle inverses exist).  k_trsm_reg streams L through a
// 16-column LDS panel: two block barriers per panel, one trip to global memory per panel that the ~1.7 k cycles of a panel's MFMAs do not cover (43 us for a single
// block = 12 panels x 3.5 us), and every 64 vectors load the factor again.  Here a workgroup of 8 wavefronts (128 vectors) loads the sub-diagonal part of L once
// (the 16 x 16 diagonal triangles are never needed: their inverses come from the Cholesky kernel, four values per lane and panel straight from global memory), and
// after ONE barrier every wavefront runs its 12 panels on its own: 4 dependent MFMAs for the panel (x_p = inv(L_pp) w_p), then one rank-16 update per tile below.
// Vectors that are contiguous along their elements (Y columns, right-hand sides) are loaded with the lanes along the elements and transposed through a per-wave LDS
// tile (they were 8-byte gathers at 1440-byte stride).  LDS: sum over panels of 16 x (b_pad - 16 (p + 1)) doubles = 118 KB at b = 180, + 17 KB of transpose tiles.
// ---------------------------------------------------------------------------------------------------------
// The three problem sets of a BCR level that share the factor C_k: the rows of X+_k, the columns of Y_k (k >= 1) and the right-hand sides of block k.  A set is
// `nvec` vectors, element i of vector v at V[v * sv + i * se] (+ batch stride).  blockIdx.y = block k, blockIdx.x = split: the block's 16-vector groups (X+ first,
// then Y, then the right-hand sides) are cut into gridDim.x contiguous shares, and a workgroup runs its share 8 groups (one per wavefront) at a time against the
// factor it loaded ONCE — on the wide levels two shares per block: one workgroup per (block, 128 vectors) loaded the factor four times per block and ran
// load -> solve -> store back to back with nothing to overlap them (33 us for 17 us of MFMAs); here the wavefronts drift apart after the first pass and one's
// loads and stores run under the other's MFMAs.
struct TrsmVecs { double* V; long long se, sv, strideV; int nvec; int first, cnt; };   // block k's entry of the set is e = k - first (Y: first = 1), present if 0 <= e < cnt
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void k_trsm_lds(const double* __restrict__ Lm, int b, long long strideL, const double* __restrict__ LIm, long long strideLI, TrsmVecs s0, TrsmVecs s1, TrsmVecs s2, int nblk) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, fk = lane >> 4, fi = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = blockIdx.y;
  const int ntb = (b + 15) >> 4, bp = 16 * ntb;
  double* Tw = lds + wv * (16 * 17);                  // this wavefront's transpose tile
  double* Lp = lds + NW * 16 * 17;           // panels: [16][R_p | 1], R_p = bp - 16 (p + 1)
  const double* L = Lm + (size_t)k * strideL;
  const double* LI = LIm + (size_t)k * strideLI;
  // this block's 16-vector groups
  const int g0n = (k - s0.first >= 0 && k - s0.first < s0.cnt && s0.nvec > 0) ? (s0.nvec + 15) >> 4 : 0;
  const int g1n = (k - s1.first >= 0 && k - s1.first < s1.cnt && s1.nvec > 0) ? (s1.nvec + 15) >> 4 : 0;
  const int g2n = (k - s2.first >= 0 && k - s2.first < s2.cnt && s2.nvec > 0) ? (s2.nvec + 15) >> 4 : 0;
  const int ng = g0n + g1n + g2n;
  const int per = (ng + (int)gridDim.x - 1) / (int)gridDim.x;
  const int gbeg = blockIdx.x * per, gend = min(ng, gbeg + per);
  if (gbeg >= gend) return;
  // 1. the factor's sub-diagonal panels -> LDS.  Thread (kk = tid >> 5, i = tid & 31 + 32 q): 256-byte runs along a column.  ALL loads first, then all LDS stores:
  // panel by panel (load, wait, store) was eleven trips to memory in a row
  constexpr int KSTEP = 2 * NW < 16 ? 2 * NW : 16;     // columns of a panel the workgroup's threads cover per trip (4 wavefronts: 8, two trips)
#pragma unroll
  for (int kh = 0; kh < 16; kh += KSTEP) {
    const int kk = kh + (tid >> 5), i5 = tid & 31;
    constexpr int QMAX = (16 * (NT - 1) + 31) / 32;
    double v[NT > 1 ? NT - 1 : 1][QMAX];
#pragma unroll
    for (int p = 0; p + 1 < NT; ++p) {
      const int col = 16 * p + kk;
#pragma unroll
      for (int q = 0; q < (16 * (NT - 1 - p) + 31) / 32; ++q) {
        const int i = 16 * (p + 1) + i5 + 32 * q;
        v[p][q] = (p < ntb && i < b && col < b && kk < kh + KSTEP) ? L[(size_t)col * b + i] : 0.0;   // (wavefronts 8 .. 11 of a 12-wavefront workgroup sit this out)
      }
    }
    int off = 0;
#pragma unroll
    for (int p = 0; p + 1 < NT; ++p) {
      const int R = bp - 16 * (p + 1), PS = R | 1;
      if (p < ntb && R > 0) {   // uniform
#pragma unroll
        for (int q = 0; q < (16 * (NT - 1 - p) + 31) / 32; ++q) {
          const int r = i5 + 32 * q;
          if (r < R && kk < kh + KSTEP) Lp[off + kk * PS + r] = v[p][q];
        }
        off += 16 * PS;
      }
    }
  }
  __syncthreads();
  // every wavefront on its own from here on (measured and dropped: the next group's vectors requested before the current solve and picked up after it —
  // 96 more live registers, spills, 468 instead of 445 us on the widest level)
  const int g = gbeg + wv;     // one pass: the launcher gives a workgroup at most 8 groups (a loop over passes here cost 110 more registers and spilled)
  if (g < gend) {
    const TrsmVecs& S = g < g0n ? s0 : (g < g0n + g1n ? s1 : s2);
    const int gl = g < g0n ? g : (g < g0n + g1n ? g - g0n : g - g0n - g1n);
    const int vbase = 16 * gl, nvec = S.nvec;
    const long long se = S.se, sv = S.sv;
    double* Vb = S.V + (size_t)(k - S.first) * S.strideV;
    auto wave_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    // 2. 16 vectors as accumulator tiles (tile t: rows 16 t .. 16 t + 15; col = lane & 15 = vector, row = (lane >> 4) + 4 reg)
    const bool along_elems = se == 1 && sv != 1;
    d4 W[NT];
    // every load of the 16 vectors first (tile by tile — load, transpose through LDS, next tile — was twelve trips to memory in a row for the element-contiguous sets)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int vec = along_elems ? vbase + fk + 4 * v : vbase + fi, i = along_elems ? 16 * t + fi : 16 * t + fk + 4 * v;
        W[t][v] = (vec < nvec && i < b) ? Vb[(size_t)vec * sv + (size_t)i * se] : 0.0;
      }
    if (along_elems) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int v = 0; v < 4; ++v) Tw[(fk + 4 * v) * 17 + fi] = W[t][v];          // [vector][element]
        wave_sync();
#pragma unroll
        for (int v = 0; v < 4; ++v) W[t][v] = Tw[fi * 17 + fk + 4 * v];
        wave_sync();
      }
    }
#ifndef LVX_TRSM_NO_COMPUTE
    // 3. the solve
    double mi[4], mn[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) mi[ks] = LI[fi * 16 + 4 * ks + fk];
    int off = 0;
#pragma unroll
    for (int p = 0; p < NT; ++p) {
      if (p < ntb) {   // uniform
        if (p + 1 < ntb) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) mn[ks] = LI[(size_t)(p + 1) * 256 + fi * 16 + 4 * ks + fk];
        }
        d4 X = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) X = __builtin_amdgcn_mfma_f64_16x16x4f64(mi[ks], W[p][ks], X, 0, 0, 0);
        W[p] = X;
        const int R = bp - 16 * (p + 1), PS = R | 1;
        const double* P = Lp + off;
#pragma unroll
        for (int t = p + 1; t < NT; ++t) {
          if (t < ntb) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) W[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-P[(4 * ks + fk) * PS + 16 * (t - p - 1) + fi], X[ks], W[t], 0, 0, 0);
          }
        }
        if (R > 0) off += 16 * PS;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) mi[ks] = mn[ks];
      }
    }
#endif
    // 4. back
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (along_elems) {
#pragma unroll
        for (int v = 0; v < 4; ++v) Tw[fi * 17 + fk + 4 * v] = W[t][v];            // [vector][element]
        wave_sync();
#pragma unroll
        for (int v = 0; v < 4; ++v) { const int vec = vbase + fk + 4 * v, i = 16 * t + fi; if (vec < nvec && i < b) Vb[(size_t)vec * sv + i] = Tw[(fk + 4 * v) * 17 + fi]; }
        wave_sync();
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) { const int vec = vbase + fi, i = 16 * t + fk + 4 * v; if (vec < nvec && i < b) Vb[(size_t)vec * sv + (size_t)i * se] = W[t][v]; }
      }
    }
  }
}
static size_t trsm_lds_bytes(int b, int NW) {
  const int ntb = (b + 15) / 16, bp = 16 * ntb;
  size_t d = (size_t)NW * 16 * 17;
  for (int p = 0; p < ntb; ++p) { const int R = bp - 16 * (p + 1); if (R > 0) d += (size_t)16 * (R | 1); }
  return d * 8;
}
static bool trsm_lds_ok(const lvx_ctx*, int b) { return b <= 192 && trsm_lds_bytes(b, 8) <= 160 * 1024; }
// the sets in the TrsmSet convention of trsv_batched (first set: batch entries 0 .. batch - 1 against L + e strideL; a set whose factor pointer starts one block
// further — the Y solves — belongs to block e + 1)
template <int NT, int NW>
static int launch_trsm_lds_nw(lvx_ctx* c, const double* L, int b, long long strideL, double* V, long long se, long long sv, long long strideV, int nvec, int batch, const double* LI, long long strideLI,
                           const TrsmSet* second, const TrsmSet* third) {
  const size_t lds = trsm_lds_bytes(b, NW);
  LVX_HIP(c, hipFuncSetAttribute((const void*)k_trsm_lds<NT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto live = [](const TrsmSet* t) { return t && t->batch > 0 && t->nvec > 0; };
  TrsmVecs v[3] = {{V, se, sv, strideV, nvec, 0, batch}, {nullptr, 0, 0, 0, 0, 0, 0}, {nullptr, 0, 0, 0, 0, 0, 0}};
  int nblk = batch, ng = (nvec + 15) / 16;
  int slot = 1;
  for (const TrsmSet* t : {second, third}) {
    if (live(t)) {
      const int first = strideL ? (int)((t->Lm - L) / strideL) : 0;     // which block this set's first entry solves against
      v[slot] = TrsmVecs{t->V, t->se, t->sv, t->strideV, t->nvec, first, t->batch};
      nblk = std::max(nblk, first + t->batch);
      ng += (t->nvec + 15) / 16;
    }
    ++slot;
  }
  // shares per block: one pass of the 8 wavefronts per workgroup (measured: fewer, longer shares — the factor loaded once or twice per block instead of four times —
  // are no faster, 445 vs 428 us on the widest level: with one workgroup per CU its load, solve and store phases do not overlap either way)
  int splits = std::max(1, (ng + NW - 1) / NW);
  hipLaunchKernelGGL((k_trsm_lds<NT, NW>), dim3((unsigned)splits, (unsigned)nblk), dim3(64 * NW), lds, c->stream, L, b, strideL, LI, strideLI, v[0], v[1], v[2], nblk);
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}

// 12 wavefronts per workgroup (three per SIMD: a wavefront's 4-MFMA panel chains hide behind two others' updates; 140 registers fit) when the LDS holds their
// transpose tiles next to the factor, 8 otherwise
template <int NT>
static int launch_trsm_lds(lvx_ctx* c, const double* L, int b, long long strideL, double* V, long long se, long long sv, long long strideV, int nvec, int batch, const double* LI, long long strideLI,
                           const TrsmSet* second, const TrsmSet* third) {
  // narrow levels: 4 wavefronts per workgroup (one per SIMD: a block's 28 groups on 7 CUs instead of 3 — a 12-wavefront workgroup is 25 us of MFMAs however few blocks there are)
  {
    int ng = (nvec + 15) / 16, nblk = batch;
    for (const TrsmSet* t : {second, third}) if (t && t->batch > 0 && t->nvec > 0) { ng += (t->nvec + 15) / 16; nblk = std::max(nblk, t->batch + 1); }
    if ((long long)nblk * ((ng + 3) / 4) <= 256)
      return launch_trsm_lds_nw<NT, 4>(c, L, b, strideL, V, se, sv, strideV, nvec, batch, LI, strideLI, second, third);
  }
  if (trsm_lds_bytes(b, 12) <= 160 * 1024) return launch_trsm_lds_nw<NT, 12>(c, L, b, strideL, V, se, sv, strideV, nvec, batch, LI, strideLI, second, third);
  return launch_trsm_lds_nw<NT, 8>(c, L, b, strideL, V, se, sv, strideV, nvec, batch, LI, strideLI, second, third);
}

// LI / strideLI: the diagonal-triangle inverses of the factors (c->bcr_linv: the own Cholesky kernel produced them), or null
template <bool TRANS>
static int trsv_batched(lvx_ctx* c, const double* L, int b, long long strideL, double* V, long long se, long long sv, long long strideV, int nvec, int batch, const double* LI, long long strideLI,
                        const TrsmSet* second = nullptr, const TrsmSet* third = nullptr) {
  if (batch <= 0 || nvec <= 0) {   // the first set is empty: promote the next live one
    for (const TrsmSet* t : {second, third}) if (t && t->batch > 0 && t->nvec > 0)
      return trsv_batched<TRANS>(c, t->Lm, b, t->strideL, t->V, t->se, t->sv, t->strideV, t->nvec, t->batch, t->LI, t->strideLI, t == second ? third : nullptr);
    return LVX_OK;
  }
  if (LI && !TRANS && trsm_lds_ok(c, b)) {   // every set of the launch has its inverses (bcr_linv is per plan)
    if (b <= 64) return launch_trsm_lds<4>(c, L, b, strideL, V, se, sv, strideV, nvec, batch, LI, strideLI, second, third);
    if (b <= 128) return launch_trsm_lds<8>(c, L, b, strideL, V, se, sv, strideV, nvec, batch, LI, strideLI, second, third);
    return launch_trsm_lds<12>(c, L, b, strideL, V, se, sv, strideV, nvec, batch, LI, strideLI, second, third);
  }
  if (LI) {
    if (b <= 128) return launch_trsm_reg<TRANS, 8, true>(c, L, b, strideL, V, se, sv, strideV, nvec, batch, LI, strideLI, second, third);
    if (b <= 208) return launch_trsm_reg<TRANS, 13, true>(c, L, b, strideL, V, se, sv, strideV, nvec, batch, LI, strideLI, second, third);
  }
  if (b <= 128) return launch_trsm_reg<TRANS, 8, false>(c, L, b, strideL, V, se, sv, strideV, nvec, batch, nullptr, 0, second, third);
  if (b <= 208) return launch_trsm_reg<TRANS, 13, false>(c, L, b, strideL, V, se, sv, strideV, nvec, batch, nullptr, 0, second, third);
  if (b <= 256) return launch_trsm_reg<TRANS, 16, false>(c, L, b, strideL, V, se, sv, strideV, nvec, batch, nullptr, 0, second, third);
  return fail(c, LVX_E_ARG, "block size too large for the batched triangular solve (half-bandwidth > 256)");
}

// ---------------------------------------------------------------------------------------------------------
// Batched Cholesky of the b x b diagonal blocks (lower, column-major in place), one workgroup per block with the whole lower triangle
// resident in LDS (packed rows: b (b + 1) / 2 doubles = 154 KB at b = 196, so b <= 201).  Right-looking with 16-column panels:
//   1. the 16 x 16 diagonal block by 16 lanes of wavefront 0 (lane = row, columns in registers, cross-lane broadcasts),
//   2. the rows below: one thread per row, forward substitution against the 16 x 16 factor,
//   3. the trailing update A22 -= L21 L21^T as 16 x 16 tiles on v_mfma_f64_16x16x4_f64, tiles dealt round-robin to the 4 wavefronts.
// rocSOLVER's strided-batched potrf runs each level as several dozen launches of small kernels (~5 ms per solve at config 4).
// info[batch] = 1-based column of the first non-positive pivot (the factorisation continues with pivot 1 so that nothing turns into NaN).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tri(int i, int j) { return ((i * (i + 1)) >> 1) + j; }
__device__ __forceinline__ double readlane_f64(double v, int l) {   // v_readlane: the value of lane l as a wave-uniform scalar (l must be uniform)
  const long long u = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(u >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double rsqrt_f64(double x) {             // hardware estimate + two Newton steps: full double precision for x in the normal range
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}
// ---------------------------------------------------------------------------------------------------------
// Register-resident batched Cholesky (the default for b <= 208).  The LDS-resident kernel above spends its time on LDS round trips of the trailing
// matrix (every tile update reads and rewrites its 16 x 16 accumulator) and on a 16-column chain of ~90 instructions per column in the diagonal block.
// Here the whole (transposed) factor lives in REGISTERS: U = L^T is cut into 16 x 16 tiles (tj, ti), tj <= ti, in the MFMA accumulator layout
// (row = (lane >> 4) + 4 reg, col = lane & 15), dealt round-robin to the 8 wavefronts of the workgroup (78 tiles at b <= 192: 10 tiles = 80 VGPRs per lane).
// Per 16-row panel k:
//   1. diagonal tile: outer-product Cholesky ON THE MATRIX CORES.  The tile is kept as a full symmetric matrix; row cc of it, masked to the 16 lanes that
//      hold it, is at once the A operand (column cc, by symmetry) and the B operand (row cc) of v_mfma_f64_16x16x4_f64, so one MFMA is the whole rank-1 update
//      T -= t_cc t_cc^T / d and a second one carries the identity along, F -= t_cc f_cc^T / d, whose row cc, scaled by 1 / sqrt(d), is row cc of inv(L_kk)
//      (Cholesky of [[A, I], [I, 0]]).  ~12 instructions per column instead of ~90; the chain is pivot -> rsqrt -> MFMA.
//   2. row panel U[k, ti] = inv(L_kk) A[k, ti]: the register tile is the B operand as it stands (its rows are the contraction index), inv(L_kk) comes from LDS;
//      the solved tiles go to an LDS row panel [16][16 NT] — the only part of the matrix that ever touches LDS.
//   3. trailing update A[tj, ti] -= U[k, tj]^T U[k, ti]: both operands from the LDS panel, the accumulator never leaves its registers.
// The owner of the NEXT diagonal tile updates it first and factorises it while the other wavefronts finish the trailing update.
// Output as before: L column-major lower in place, inv(L_kk) per panel to LIm (k_trsm_reg<.., DINV>, k_bcr_back_level), info = first bad pivot.
// ---------------------------------------------------------------------------------------------------------
template <int K, int N, class Fn> __device__ __forceinline__ void static_for(Fn&& f) {
  if constexpr (K < N) { f(std::integral_constant<int, K>{}); static_for<K + 1, N>(f); }
}
#define POTRF_REG_NW 8
#ifdef LVX_POTRF_KT
__device__ __forceinline__ long long pkt_now() { long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define PKT(i) { const long long n_ = pkt_now(); pkt_[i] += n_ - pkt0_; pkt0_ = n_; }
#else
#define PKT(i)
#endif
template <int NT>
__global__ __launch_bounds__(64 * POTRF_REG_NW) void k_potrf_reg(double* Dm, int b, long long strideD, int* info, double* LIm, long long strideLI) {
  // wavefront 0 factorises the diagonal tiles and does nothing else: that chain is what every panel waits for; wavefront 4 (same SIMD) keeps the waiting diagonal
  // tiles up to date; the NT (NT - 1) / 2 off-diagonal tiles are dealt to the other six.  The panel loop is a RUN-TIME loop (the fully unrolled version executed every
  // instruction once, 160 KB of code: instruction fetch, not arithmetic, set its pace — the 16 x 16 factorisation took 4.7 k cycles inside it and 0.8 k with a
  // warm instruction cache), so the diagonal wavefront keeps its waiting tiles in LDS in the accumulator layout (every lane touches only its own four words of a
  // tile: no synchronisation, dynamic tile index); the off-diagonal tiles sit in statically indexed registers.
  constexpr int NCW = 6, NTO = NT * (NT - 1) / 2, NSO = (NTO + NCW - 1) / NCW, PS = (16 * NT) | 1;
  __shared__ double Pn[2][16 * PS];    // solved row panel U[k, :], double-buffered over k
  __shared__ double Mi[2][16 * 17];    // inv(L_kk)
  __shared__ double Dg[NT][4][64];     // diagonal tiles not yet factorised (wavefront 0 only)
  __shared__ int bad;
  const int tid = threadIdx.x, lane = tid & 63, fk = lane >> 4, fi = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool fw = wv == 0;
  const int cw = (wv & 3) == 0 ? -1 : (wv < 4 ? wv - 1 : wv - 2);   // compute wavefront 0 .. 5
  double* D = Dm + (size_t)blockIdx.x * strideD;
  double* LI = LIm ? LIm + (size_t)blockIdx.x * strideLI : nullptr;
  if (tid == 0) bad = 0;
#ifdef LVX_POTRF_KT
  long long pkt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long pkt0_ = pkt_now(); const long long pkts_ = pkt0_;
#endif
  d4 S[NSO];
  int ti_[NSO], tj_[NSO];
#pragma unroll
  for (int s = 0; s < NSO; ++s) {
    const int t = s * NCW + cw;
    int ti = 1; while ((ti * (ti + 1)) >> 1 <= t) ++ti;          // off-diagonal tile t = ti (ti - 1) / 2 + tj, tj < ti
    const bool ok = cw >= 0 && t < NTO;
    ti_[s] = ok ? ti : -1; tj_[s] = ok ? t - ((ti * (ti - 1)) >> 1) : NT + 1;
  }
  // every tile straight from global memory into its registers: unconditional loads (clamped offsets, selected afterwards), all in flight together; diagonal
  // tiles are mirrored, padding is the identity.  No barrier: wavefront 0 starts on tile (0, 0) as soon as it is back.
  auto load_elem = [&](int ti, int tj, int v) {
    const int cc = 16 * tj + fk + 4 * v, r = 16 * ti + fi;
    const int lo = min(cc, r), hi = max(cc, r);
    const bool ok = ti >= 0 && hi < b;
    const double x = D[ok ? lo * b + hi : 0];
    return ok ? x : ((ti >= 0 && lo == hi) ? 1.0 : 0.0);
  };
  d4 Tcur = d4{0.0, 0.0, 0.0, 0.0}, Mcur = d4{0.0, 0.0, 0.0, 0.0};
  if (fw) {
    d4 G[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
      for (int v = 0; v < 4; ++v) G[s][v] = load_elem(s, s, v);
    Tcur = G[0];
#pragma unroll
    for (int s = 1; s < NT; ++s)
#pragma unroll
      for (int v = 0; v < 4; ++v) Dg[s][v][lane] = G[s][v];
  } else {
#pragma unroll
    for (int s = 0; s < NSO; ++s)
#pragma unroll
      for (int v = 0; v < 4; ++v) S[s][v] = load_elem(ti_[s], tj_[s], v);
  }
  PKT(0)
  // a finished tile goes back at once (L column-major lower: U (row cc, col r) = L[r][cc]); the stores drain under the rest of the factorisation
  auto store_tile = [&](const d4& X, int ti, int tj) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int cc = 16 * tj + fk + 4 * v, r = 16 * ti + fi;
      if (r < b && cc <= r) D[cc * b + r] = X[v];
    }
  };
  auto factor_diag = [&](d4& T, int k) {
    d4 Mres;
#ifdef LVX_POTRF_KT
    asm volatile("" : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]));
    const long long c0_ = pkt_now();
#endif
    const int bc = chol16_mfma(T, Mres, fk, fi);
    if (lane == 0 && bc > 0 && 16 * k + bc <= b) atomicCAS(&bad, 0, 16 * k + bc);   // first non-positive pivot (replaced by 1: nothing turns into NaN)
#ifdef LVX_POTRF_KT
    asm volatile("" : "+v"(T[0]), "+v"(Mres[0]), "+v"(Mres[3]), "+v"(T[3]));
    pkt_[7] += pkt_now() - c0_;
#endif
    double* mi = Mi[k & 1];
#pragma unroll
    for (int v = 0; v < 4; ++v) mi[(fk + 4 * v) * 17 + fi] = Mres[v];
    return Mres;
  };
  // what the panel solves do not wait for: the factor's inverse and the diagonal tile itself go to global memory behind the barrier
  auto publish_diag = [&](const d4& T, const d4& Mres, int k) {
    if (LI && 16 * k < b) {   // ceil(b / 16) panels per block; the tiles beyond are identity padding
#pragma unroll
      for (int v = 0; v < 4; ++v) LI[k * 256 + (fk + 4 * v) * 16 + fi] = Mres[v];
    }
    store_tile(T, k, k);
  };
  if (fw) Mcur = factor_diag(Tcur, 0);
  PKT(1)
  __syncthreads();
  PKT(2)
#pragma unroll 1
  for (int k = 0; k + 1 < NT; ++k) {
    const double* mi = Mi[k & 1];
    double* pn = Pn[k & 1];
    if (fw) publish_diag(Tcur, Mcur, k);
    else {
      // 2. row panel: U[k, ti] = inv(L_kk) A[k, ti]
#pragma unroll
      for (int s = 0; s < NSO; ++s) {
        if (tj_[s] == k) {    // wave-uniform
          d4 X = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) X = __builtin_amdgcn_mfma_f64_16x16x4f64(mi[fi * 17 + 4 * ks + fk], S[s][ks], X, 0, 0, 0);
          S[s] = X;
#pragma unroll
          for (int v = 0; v < 4; ++v) pn[(fk + 4 * v) * PS + 16 * ti_[s] + fi] = X[v];
        }
      }
    }
    PKT(3)
    __syncthreads();
    PKT(4)
    // 3. trailing update; wavefront 0: the next diagonal tile, its factorisation, then its other diagonal tiles
    auto update = [&](d4& C, int ti, int tj) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) C = __builtin_amdgcn_mfma_f64_16x16x4f64(-pn[(4 * ks + fk) * PS + 16 * tj + fi], pn[(4 * ks + fk) * PS + 16 * ti + fi], C, 0, 0, 0);
      asm volatile("" ::: "memory");   // the operand loads of the next tile stay behind this one's (all of a wavefront's tiles hoisted at once: spills)
    };
    if (fw) {
#pragma unroll
      for (int v = 0; v < 4; ++v) Tcur[v] = Dg[k + 1][v][lane];
      update(Tcur, k + 1, k + 1);
      Mcur = factor_diag(Tcur, k + 1);   // the chain everybody waits for
      PKT(1)
    } else if (wv == 4) {
      // the diagonal wavefront's SIMD-mate keeps the OTHER waiting diagonal tiles up to date (they sat behind the factorisation on wavefront 0: up to ten tile
      // updates of ~0.5 k cycles before the barrier); its MFMAs fill the gaps of the factorisation's chain (one MFMA per ~140 cycles).  Two tiles at a time: the
      // 4-MFMA chains of a tile are dependent, two tiles' chains interleave.
      for (int s = k + 2; s < NT; s += 2) {
        const bool two = s + 1 < NT;
        d4 C0, C1;
#pragma unroll
        for (int v = 0; v < 4; ++v) { C0[v] = Dg[s][v][lane]; C1[v] = two ? Dg[s + 1][v][lane] : 0.0; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-pn[(4 * ks + fk) * PS + 16 * s + fi], pn[(4 * ks + fk) * PS + 16 * s + fi], C0, 0, 0, 0);
          if (two) C1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-pn[(4 * ks + fk) * PS + 16 * (s + 1) + fi], pn[(4 * ks + fk) * PS + 16 * (s + 1) + fi], C1, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) { Dg[s][v][lane] = C0[v]; if (two) Dg[s + 1][v][lane] = C1[v]; }
      }
    } else {
#pragma unroll
      for (int s = 0; s < NSO; ++s) {
        if (tj_[s] == k) store_tile(S[s], ti_[s], k);        // the solved panel tile, off the barrier-to-barrier path
        else if (tj_[s] > k && ti_[s] >= 0) update(S[s], ti_[s], tj_[s]);
      }
    }
    PKT(5)
    __syncthreads();
    PKT(2)
  }
  if (fw) publish_diag(Tcur, Mcur, NT - 1);
  if (tid == 0) info[blockIdx.x] = bad;
#ifdef LVX_POTRF_KT
  PKT(6)
  if (lane == 0 && blockIdx.x == 0) printf("PKT b %d wv %d: load %lld factor %lld syncA %lld panel %lld syncB %lld trail %lld store %lld chol16 %lld total %lld\n", b, wv, pkt_[0], pkt_[1], pkt_[2], pkt_[3], pkt_[4], pkt_[5], pkt_[6], pkt_[7], pkt_now() - pkts_);
#endif
}
static bool potrf_reg_ok(const lvx_ctx*, int b) { return b <= 208; }
template <int NT> static void launch_potrf_reg(lvx_ctx* c, double* D, int b, long long strideD, int* info, int batch, double* LI, long long strideLI) {
  hipLaunchKernelGGL(k_potrf_reg<NT>, dim3((unsigned)batch), dim3(64 * POTRF_REG_NW), 0, c->stream, D, b, strideD, info, LI, strideLI);
}
// potrf of `batch` blocks: the register-resident kernel up to b = 208 (it also leaves the inverses of the diagonal triangles for the solves), rocSOLVER beyond
static bool potrf_own(const lvx_ctx* c, int b) { return potrf_reg_ok(c, b); }
static int potrf_batched(lvx_ctx* c, rocblas_handle& h, double* D, int b, long long strideD, int* info, int batch, double* LI, long long strideLI) {
  if (!potrf_own(c, b)) {
    { LVX_BLAS_H(c, h); LVX_BLAS(c, g_vb.dpotrf_strided_batched(h, rocblas_fill_lower, b, D, b, (rocblas_stride)strideD, info, batch)); }
    return LVX_OK;
  }
  if (b <= 64) launch_potrf_reg<4>(c, D, b, strideD, info, batch, LI, strideLI);
  else if (b <= 128) launch_potrf_reg<8>(c, D, b, strideD, info, batch, LI, strideLI);
  else if (b <= 192) launch_potrf_reg<12>(c, D, b, strideD, info, batch, LI, strideLI);
  else launch_potrf_reg<13>(c, D, b, strideD, info, batch, LI, strideLI);
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}


// ---------------------------------------------------------------------------------------------------------
// One level of the backward sweep for a SINGLE right-hand side (the LM step itself), fused:  z_j <- C_j^-T (z_j - X+_k^T z_{j+s} - Y_k z_{j-s}).
// Three library launches per level did this (two batched GEMMs with one column, a 64-vector triangular solve with one live vector: ~110 us per level,
// latency of 13 panel steps each).  Here a workgroup owns block j: both matrix-vector products stream their 180 x 180 operands from HBM with 45 loads in flight
// per lane, the factor's lower triangle is staged in LDS packed by COLUMNS (the transposed solve walks columns), and a panel's triangle is applied as
// x_p = inv(L_pp)^T (...) from the inverses the Cholesky kernel left — 12 short steps instead of 180.  (An earlier fused attempt kept the substitution chain: 78 us.)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int coff(int b, int i) { return i * b - ((i * (i - 1)) >> 1); }   // start of column i (rows i .. b - 1) in the column-packed lower triangle
#define BACK_NT 512
__global__ __launch_bounds__(BACK_NT) void k_bcr_back_level(const double* __restrict__ Dj, long long sD, const double* __restrict__ LIj, long long sLI, const double* __restrict__ Gl, long long sG,
                                                        double* Z, long long zj_off, long long zr_off, long long sZ, int b, int n2, long long bb, int det) {
  extern __shared__ double sh[];
  const int k = blockIdx.x;
  if (k >= n2) return;
  constexpr int NW = BACK_NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ntri = (b * (b + 1)) >> 1;
  double* T = sh;                 // column-packed lower triangle of C_j
  double* v = T + ntri;           // [b] right-hand side, then the solution
  double* zr = v + b;             // [b] z_{j+s}
  double* zl = zr + b;            // [b] z_{j-s}
  double* sp = zl + b;            // [16] partial sums of a panel
  double* Mi = sp + 16;           // [16][17] inv(L_pp)
  const double* L = Dj + (size_t)k * sD;
  const double* LI = LIj + (size_t)k * sLI;
  const double* Xp = Gl ? Gl + (size_t)k * sG : nullptr;      // Gl == null: a block without neighbours (the last one of the chain): the transposed solve only
  const double* Y = (Gl && k >= 1) ? Gl + (size_t)(k - 1) * sG + bb : nullptr;
  double* zj = Z + zj_off + (size_t)k * sZ;
  const double* zrg = Gl ? Z + zr_off + (size_t)k * sZ : nullptr;
  const double* zlg = (Gl && k >= 1) ? Z + zr_off + (size_t)(k - 1) * sZ : nullptr;
  for (int i = tid; i < b; i += BACK_NT) { v[i] = zj[i]; zr[i] = zrg ? zrg[i] : 0.0; zl[i] = zlg ? zlg[i] : 0.0; }
  __syncthreads();
  // One sweep over the three operands, a wavefront per column and four columns of EACH operand (up to 48 loads per lane) in flight per trip — one column at a time
  // waited for every load: 95 us for a single block.
  //   factor: column i, rows i .. b - 1 -> LDS (contiguous in global memory and in LDS);
  //   v -= X+^T z_{j+s}: entry i is the dot product of COLUMN i of X+ with z_{j+s};
  //   v -= Y z_{j-s}: lane = row (coalesced over rows for a fixed column), the columns dealt over the wavefronts, partial sums added in LDS.
  double ya[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i0 = wv * 4; i0 < b; i0 += NW * 4) {
    double tl[4][4], tx[4][4], ty[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + q, r = lane + 64 * u;
        tl[q][u] = (i < b && i + r < b) ? L[(size_t)i * b + i + r] : 0.0;
        tx[q][u] = (Xp && i < b && r < b) ? Xp[(size_t)i * b + r] : 0.0;
        ty[q][u] = (Y && i < b && r < b) ? Y[(size_t)i * b + r] : 0.0;
      }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q;
      double a = 0.0;
      const double zc = i < b ? zl[i] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = lane + 64 * u;
        if (i < b && i + r < b) T[coff(b, i) + r] = tl[q][u];
        if (r < b) a += tx[q][u] * zr[r];
        ya[u] += ty[q][u] * zc;
      }
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
      if (lane == 0 && i < b) atomicAdd(&v[i], -a);
    }
  }
  if (Y && det) {   // deterministic mode: the wavefronts add their partial sums one after the other, behind the X+ terms
    for (int w_ = 0; w_ < NW; ++w_) {
      __syncthreads();
      if (wv == w_) for (int u = 0; u < 4; ++u) { const int r = lane + 64 * u; if (r < b) v[r] -= ya[u]; }
    }
  } else if (Y) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int r = lane + 64 * u; if (r < b) atomicAdd(&v[r], -ya[u]); }   // LDS, one adder per wavefront and entry
  }
  // transposed solve, panels from the last to the first
  const int np = (b + 15) >> 4;
  double mi_next = tid < 256 ? LI[(size_t)(np - 1) * 256 + tid] : 0.0;
  __syncthreads();
  // Right-looking: once a panel's 16 unknowns are out, every EARLIER entry takes its share at once — v[c] -= sum_i L[k0 + i][c] x[k0 + i], a 16-term product per thread from
  // 16 contiguous words of the column-packed factor — instead of each panel first reducing over all the rows below it (64-lane reductions, two columns per wavefront:
  // ~4 k cycles per panel); two barriers per panel remain.
  for (int p = np - 1; p >= 0; --p) {
    const int k0 = 16 * p, nk = min(16, b - k0);
    if (tid < 256) { Mi[(tid >> 4) * 17 + (tid & 15)] = mi_next; if (p > 0) mi_next = LI[(size_t)(p - 1) * 256 + tid]; }
    __syncthreads();                 // Mi in place; v[k0 ..] carries every later panel's update
    if (tid < 16) {                  // x_p = inv(L_pp)^T w
      double x = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) x += Mi[q * 17 + tid] * (q < nk ? v[k0 + q] : 0.0);
      sp[tid] = tid < nk ? x : 0.0;
    }
    __syncthreads();
    if (tid < 16 && tid < nk) v[k0 + tid] = sp[tid];
    if (tid < k0) {
      const double* col = T + coff(b, tid) + (k0 - tid);      // L[k0 .. k0 + 15][tid]
      double a = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) a += (i < nk ? col[i] : 0.0) * sp[i];
      v[tid] -= a;
    }
  }
  __syncthreads();
  for (int i = tid; i < b; i += BACK_NT) zj[i] = v[i];
}

// ---------------------------------------------------------------------------------------------------------
// Schur updates of a BCR level in ONE launch (five rocBLAS strided-batched GEMMs before: 64 x 64 Tensile tiles, a full GEMM where only the lower triangle of D is
// read, 8 - 10 us each on the narrow levels):
//   D_{j+s} -= X+_k X+_k^T + Y_{k+1}^T Y_{k+1}   (both updates of a diagonal block by the same workgroup: one read-modify-write, LOWER tiles only)
//   A_{j+s,j-s} = -X+_k Y_k                         (next level's coupling)
//   b_{j+s} -= X+_k y_k + Y_{k+1}^T y_{k+1}         (right-hand sides riding along)
// One workgroup of 12 wavefronts per (block, output): the output's 16 x 16 tiles are dealt round-robin to the wavefronts into statically indexed accumulators
// (D: 78 lower tiles = 7 per wavefront, coupling: 144 = 12, right-hand sides: 48 = 4), the K loop stages 16-deep chunks of the row panel [16][192] and the column
// panel in LDS, double-buffered (one barrier per chunk; the next chunk's global loads are issued before the MFMAs of the current one).  X X^T and Y^T Y take both
// operands from the row panel.  Every MFMA reads its two operands from LDS (no register reuse across tiles: 64 B per clock and CU, half the LDS rate) — what matters
// is that three wavefronts per SIMD keep the matrix pipe (64 cycles per FP64 MFMA) busy across the barrier.  The MFMA computes the TRANSPOSED tile (A operand = column
// panel, B operand = row panel) so that a lane's accumulator entries are contiguous along the rows of the column-major outputs.
// (First version: 64-column output panels, 8 wavefronts with 3 x 2 register-blocked tiles, single-buffered with two barriers per chunk: level with rocBLAS at 32 % of the
// matrix pipe's rate.)
// ---------------------------------------------------------------------------------------------------------
struct SchurArgs {
  const double* Gl; long long sG, bb;       // X+_k = Gl + k sG, Y_k = Gl + (k - 1) sG + bb (k >= 1)
  double* Dr; long long sD;                 // D_{j+s} of eliminated k
  double* Gn;                               // next level's couplings, block k - 1 <- -X+_k Y_k
  const double* Zj; double* Zr; long long sZ; int ldz, nrhs;   // right-hand sides (null: none)
  int b, n2;
};
#define SCHUR_NW 12
// TYPE 0: D (lower tiles), 1: coupling, 2: right-hand sides.  NTB: 16 x 16 tiles per block side (b <= 16 NTB).  ZT: column tiles of the right-hand sides (nrhs <= 16 ZT)
template <int NTB, int TYPE, int ZT, int SPLIT>
__device__ __forceinline__ void schur_body(const SchurArgs& a, double* lds) {
  constexpr int NT = SCHUR_NW * 64, PR = 16 * NTB, PSA = PR | 1, CW = TYPE == 2 ? 16 * ZT : PR, PSB = CW | 1, KC = 16;
  constexpr int NTILE = TYPE == 0 ? NTB * (NTB + 1) / 2 : (TYPE == 1 ? NTB * NTB : NTB * ZT), NSLOT = (NTILE + SCHUR_NW * SPLIT - 1) / (SCHUR_NW * SPLIT);   // SPLIT workgroups (blockIdx.z) share an output on the narrow levels
  constexpr int BUF = KC * PSA + (TYPE == 0 ? 0 : KC * PSB);
  const int b = a.b, k = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, fk = lane >> 4, fi = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double* X = a.Gl + (size_t)k * a.sG;
  const double* Yk = k >= 1 ? a.Gl + (size_t)(k - 1) * a.sG + a.bb : nullptr;
  const double* Yn = k + 1 < a.n2 ? a.Gl + (size_t)k * a.sG + a.bb : nullptr;
  // products: row source (ptr, transposed?), column source (ptr, ld; always contiguous along K), K = b
  const double* rsrc[2] = {X, Yn}; const bool rtr[2] = {false, true};
  const double* csrc[2] = {nullptr, nullptr}; int cld = b, np = 1, ncol = b;
  double* C; int ldc; bool sub;
  if (TYPE == 0) { np = Yn ? 2 : 1; C = a.Dr + (size_t)k * a.sD; ldc = b; sub = true; }
  else if (TYPE == 1) { csrc[0] = Yk; C = a.Gn + (size_t)(k - 1) * a.bb; ldc = b; sub = false; }
  else { csrc[0] = a.Zj + (size_t)k * a.sZ; csrc[1] = a.Zj + (size_t)(k + 1) * a.sZ; cld = a.ldz; np = Yn ? 2 : 1; C = a.Zr + (size_t)k * a.sZ; ldc = a.ldz; sub = true; ncol = a.nrhs; }
  // this wavefront's tiles: slot s <-> tile t = s * 12 + wv -> (rt, ct)
  int rt_[NSLOT], ct_[NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int t = min((s * SPLIT + (int)blockIdx.z) * SCHUR_NW + wv, NTILE - 1);      // the last slot's spare wavefronts recompute the last tile (and do not store it)
    if (TYPE == 0) { int r = 0; while (((r + 1) * (r + 2)) >> 1 <= t) ++r; rt_[s] = r; ct_[s] = t - ((r * (r + 1)) >> 1); }
    else if (TYPE == 1) { rt_[s] = t / NTB; ct_[s] = t % NTB; }
    else { rt_[s] = t / ZT; ct_[s] = t % ZT; }
  }
  d4 acc[NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) acc[s] = d4{0.0, 0.0, 0.0, 0.0};
  // Operand loads are 16 bytes wide: PAIRS of elements (b is a multiple of 4, chunks start at multiples of 16: a pair never straddles K = b or row b).
  // row panel pair e: natural source (X: contiguous along the rows): i = 2 (e % (PR / 2)), kk = e / (PR / 2); transposed source (Y^T: contiguous along K): kk = 2 (e % 8), i = e / 8.
  // column panel: always contiguous along K (Y_k columns, right-hand sides): kk = 2 (e % 8), j = e / 8.
  // Everything that does not depend on the chunk is formed once (the first version rebuilt 64-bit addresses and bounds behind three branches per load, inside the K loop:
  // ~40 instructions per load, as long as the chunk's MFMAs): element offset of (index, k = 0), whether the index exists, kk; per chunk only the clamped k is added.
  typedef double d2 __attribute__((ext_vector_type(2)));
  constexpr int NPA = (PR * KC / 2 + NT - 1) / NT, NPB = TYPE == 0 ? 0 : (CW * KC / 2 + NT - 1) / NT;
  d2 pa[NPA], pb[NPB > 0 ? NPB : 1];
  int ra_off[2][NPA], ra_kk[2][NPA], ra_lds[2][NPA]; bool ra_ok[2][NPA];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
      const int e = tid + NT * q;
      const bool tr = rtr[p];
      const int i = tr ? e >> 3 : 2 * (e % (PR / 2)), kk = tr ? 2 * (e & 7) : e / (PR / 2);
      ra_ok[p][q] = e < PR * KC / 2 && i < b;
      ra_off[p][q] = ra_ok[p][q] ? (tr ? i * b : i) : 0;
      ra_kk[p][q] = kk;
      ra_lds[p][q] = e < PR * KC / 2 ? kk * PSA + i : -1;
    }
  int cb_off[NPB > 0 ? NPB : 1], cb_lds[NPB > 0 ? NPB : 1]; bool cb_ok[NPB > 0 ? NPB : 1];
  const int ckk = 2 * (tid & 7);      // NT is a multiple of 8: the same kk for every q
#pragma unroll
  for (int q = 0; q < NPB; ++q) {
    const int e = tid + NT * q, j = e >> 3;
    cb_ok[q] = e < CW * KC / 2 && j < ncol;
    cb_off[q] = cb_ok[q] ? j * cld : 0;
    cb_lds[q] = e < CW * KC / 2 ? ckk * PSB + j : -1;
  }
  // fetch only ISSUES the loads (clamped, always valid addresses) and notes which values count; the values are looked at in commit, after the chunk's MFMAs —
  // with the select next to the load the compiler waited for every load on the spot (s_waitcnt vmcnt(0) eight times per chunk, nothing overlapped)
  unsigned pmask = 0;
  auto fetch = [&](int p, int k0) {
    const double* rs = rsrc[p];
    const int kstr = rtr[p] ? 1 : b;
    pmask = 0;
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
      const int kq = k0 + ra_kk[p][q];
      pa[q] = *(const d2*)(rs + ra_off[p][q] + min(kq, rtr[p] ? b - 2 : b - 1) * kstr);   // transposed source: the pair is (k, k + 1); natural: (i, i + 1) of column k
      pmask |= (ra_ok[p][q] && kq < b) ? (1u << q) : 0u;
    }
    if (TYPE != 0) {
      const double* cs = csrc[p];
      const int kq = k0 + ckk, kc = min(kq, b - 2);
#pragma unroll
      for (int q = 0; q < NPB; ++q) {
        pb[q] = *(const d2*)(cs + cb_off[q] + kc);
        pmask |= (cb_ok[q] && kq < b) ? (1u << (16 + q)) : 0u;
      }
    }
  };
  auto commit = [&](int p, double* buf) {
    double* As = buf; double* Bs = buf + KC * PSA;
    const int step = rtr[p] ? PSA : 1;     // the pair's second element: next k (transposed source) or next row (natural)
#pragma unroll
    for (int q = 0; q < NPA; ++q) if (ra_lds[p][q] >= 0) {
      const bool on = (pmask >> q) & 1u;
      As[ra_lds[p][q]] = on ? pa[q][0] : 0.0; As[ra_lds[p][q] + step] = on ? pa[q][1] : 0.0;
    }
    if (TYPE != 0) {
#pragma unroll
      for (int q = 0; q < NPB; ++q) if (cb_lds[q] >= 0) {
        const bool on = (pmask >> (16 + q)) & 1u;
        Bs[cb_lds[q]] = on ? pb[q][0] : 0.0; Bs[cb_lds[q] + PSB] = on ? pb[q][1] : 0.0;
      }
    }
  };
  const int nchunk = (b + KC - 1) / KC, total = np * nchunk;
  fetch(0, 0);
  commit(0, lds);
  for (int it = 0; it < total; ++it) {
    __syncthreads();                 // chunk `it` is in its buffer; everybody is done with the other buffer (chunk it - 1)
    const bool more = it + 1 < total;
    if (more) { const int pn = it + 1 >= nchunk ? 1 : 0; fetch(pn, (it + 1 - pn * nchunk) * KC); }
    const double* As = lds + (it & 1) * BUF;
    const double* Bs = TYPE == 0 ? As : As + KC * PSA;
    constexpr int cps = TYPE == 0 ? PSA : PSB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int s = 0; s < NSLOT; ++s)
        acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bs[(4 * ks + fk) * cps + 16 * ct_[s] + fi], As[(4 * ks + fk) * PSA + 16 * rt_[s] + fi], acc[s], 0, 0, 0);
    if (more) commit(it + 1 >= nchunk ? 1 : 0, lds + ((it + 1) & 1) * BUF);
  }
  // accumulator (reg v, lane (fk, fi)) = sum for column j = 16 ct + fk + 4 v, row i = 16 rt + fi
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    if ((s * SPLIT + (int)blockIdx.z) * SCHUR_NW + wv >= NTILE) continue;
    const int i = 16 * rt_[s] + fi;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int j = 16 * ct_[s] + fk + 4 * v;
      if (i < b && j < ncol && (TYPE != 0 || i >= j)) {
        double* dst = C + (size_t)j * ldc + i;
        *dst = sub ? *dst - acc[s][v] : -acc[s][v];
      }
    }
  }
}
template <int NTB, int ZT, int SPLIT>
__global__ __launch_bounds__(64 * SCHUR_NW) void k_bcr_schur(SchurArgs a) {
  extern __shared__ double lds[];
  const int type = blockIdx.x;
  if (type == 0) schur_body<NTB, 0, ZT, SPLIT>(a, lds);
  else if (type == 1) { if (blockIdx.y >= 1) schur_body<NTB, 1, ZT, SPLIT>(a, lds); }
  else schur_body<NTB, 2, ZT, SPLIT>(a, lds);
}
// LVX_BCR_OWN_SCHUR: 1 own kernel on every level, 0 rocBLAS on every level, default (-1): own kernel
static bool schur_own(const lvx_ctx*, int b, int nrhs) { return b <= 208 && nrhs <= 64; }
template <int NTB, int ZT, int SPLIT> static int launch_schur_s(lvx_ctx* c, const SchurArgs& a) {
  constexpr int PR = 16 * NTB, PSA = PR | 1;
  const size_t lds = (size_t)2 * (16 * PSA + 16 * PSA) * 8;     // the coupling product is the largest: row panel + a full column panel, two buffers
  LVX_HIP(c, hipFuncSetAttribute((const void*)k_bcr_schur<NTB, ZT, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_bcr_schur<NTB, ZT, SPLIT>), dim3(a.Zj ? 3u : 2u, (unsigned)a.n2, (unsigned)SPLIT), dim3(64 * SCHUR_NW), lds, c->stream, a);
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}
// one workgroup per (block, output) where that fills the chip; two on the narrow levels (a single block's D update is 47 us of MFMAs on one CU)
template <int NTB, int ZT> static int launch_schur(lvx_ctx* c, const SchurArgs& a) {
  if (3 * a.n2 >= 512) return launch_schur_s<NTB, ZT, 1>(c, a);   // measured at config 4: 209 blocks 231 us whole / 280 us split, 104 blocks 150 us whole (312 workgroups on 256 CUs: two rounds for 1.2 rounds of work)
  if (12 * a.n2 <= 256) return launch_schur_s<NTB, ZT, 4>(c, a);   // the narrowest levels: four workgroups per output (a quarter of the tiles each, every one streams the row panel: bandwidth is not what these levels lack)
  return launch_schur_s<NTB, ZT, 2>(c, a);
}
static int schur_level(lvx_ctx* c, const SchurArgs& a) {
  if (a.b <= 64) return launch_schur<4, 4>(c, a);
  if (a.b <= 128) return launch_schur<8, 4>(c, a);
  if (a.b <= 192) return launch_schur<12, 4>(c, a);
  return launch_schur<13, 4>(c, a);
}

int bcr_plan(lvx_ctx* c) {
  const int b = std::max(16, ((c->bw + 3) / 4) * 4);
  int nblk = 1;
  while ((long long)nblk * b < c->nb) nblk <<= 1;
  nblk = std::max(nblk, 2);
  c->bcr_b = b; c->bcr_nblk = nblk; c->bcr_nreal = (c->nb + b - 1) / b;
  int rc;
  const size_t bb = (size_t)b * b;
  const size_t guard = 1;
  if ((rc = dev_alloc(c, c->d_bcrD, guard * (size_t)nblk * bb * 8))) return rc;       // diagonal blocks -> Cholesky factors C_j
  if ((rc = dev_alloc(c, c->d_bcrG, guard * (size_t)2 * nblk * bb * 8))) return rc;   // couplings per level -> X+ (even slots) / Y (odd slots)
  if ((rc = dev_alloc(c, c->d_bcrInfo, guard * (size_t)(2 * nblk + 8) * 4))) return rc;
  c->bcr_linv = potrf_own(c, b) && b <= 208;   // inverses of the factors' 16 x 16 diagonal triangles: [nblk][ceil(b / 16)][16][16]
  if (c->bcr_linv && (rc = dev_alloc(c, c->d_bcrLinv, (size_t)nblk * ((b + 15) / 16) * 256 * 8))) return rc;
  // couplings of the levels above 0 that involve a padding block are never computed (level_batch) and must read as zero
  LVX_HIP(c, hipMemsetAsync(c->d_bcrG.p, 0, (size_t)2 * nblk * bb * 8, c->stream));
  if (c->bcr_nreal < nblk) {   // padding blocks: identity, decoupled — nothing ever changes them (potrf(I) = I, updates with zero couplings)
    const size_t npad = (size_t)(nblk - c->bcr_nreal) * bb;
    hipLaunchKernelGGL(k_bcr_pad_identity, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, c->stream, (double*)c->d_bcrD.p, b, c->bcr_nreal, nblk);
  }
  return LVX_OK;
}
// start of level l's blocks inside the per-level array (level l holds nblk >> l blocks)
static inline size_t g_off(int nblk, int l, size_t bb) { size_t o = 0; for (int k = 0; k < l; ++k) o += (size_t)(nblk >> k) * bb; return o; }

// Z != null: the forward substitution of the right-hand sides Z [ldz x nrhs] (bcr_forward) rides along — its triangular solves against C_j join the X+ / Y
// launch of the level, its updates follow the level's GEMMs
// Blocks eliminated at level l: j = s - 1 + 2 s k.  The chain is padded to a power of two with identity blocks that couple to nothing: only the k with
// j < nreal need any work (834 of 1024 blocks at config 4: 19 % of every batched launch of the wide levels)
static inline int level_batch(int nblk, int nreal, int l) {
  const int s = 1 << l, full = (nblk >> l) / 2;
  if (nreal <= s - 1) return 0;
  return std::min(full, (nreal - (s - 1) + 2 * s - 1) / (2 * s));
}

// A chain of nblk (power of two; nreal live) b x b blocks: diagonal blocks D (-> factors), per-level couplings G, the factors' diagonal-triangle inverses LI (or null), pivot codes.
// The band itself (bcr_factor) and the separator system of the leaves + separators elimination (lvx_nd.h) are both such chains.
struct BcrChain { int b, nblk, nreal; double *D, *G, *LI; int* info; };
static BcrChain ctx_chain(lvx_ctx* c) {
  return BcrChain{c->bcr_b, c->bcr_nblk, c->bcr_nreal, (double*)c->d_bcrD.p, (double*)c->d_bcrG.p, c->bcr_linv ? (double*)c->d_bcrLinv.p : nullptr, (int*)c->d_bcrInfo.p};
}
static int chain_factor(lvx_ctx* c, const BcrChain& ch, int* info_out_d, double* Z, int ldz, int nrhs);
int bcr_factor(lvx_ctx* c, const double* scale, const double* lmd, double inv_radius, int* info_out_d, double* Z, int ldz, int nrhs) {
  const BcrChain ch = ctx_chain(c);
  const int nfill = std::min(ch.nblk, std::max(ch.nreal, 1));
  hipLaunchKernelGGL(k_bcr_build, dim3((unsigned)((ch.b + 3) / 4), (unsigned)nfill, 2), dim3(256), 0, c->stream, c->p_Hs ? c->p_Hs : (const double*)c->d_Hb.p, scale, lmd, inv_radius, c->nb, c->bw, ch.b, nfill, ch.D, ch.G);
  return chain_factor(c, ch, info_out_d, Z, ldz, nrhs);
}
static int chain_factor(lvx_ctx* c, const BcrChain& ch, int* info_out_d, double* Z, int ldz, int nrhs) {
  rocblas_handle h = nullptr; int rc = LVX_OK;   // (the handle is fetched only where a library call is really made: blocks wider than 208)
  const int b = ch.b, nblk = ch.nblk, nreal = ch.nreal;
  const size_t bb = (size_t)b * b;
  double* D = ch.D; double* G = ch.G; int* info = ch.info;
  double* LI = ch.LI;
  const size_t liS = (size_t)((b + 15) / 16) * 256;
  hipStream_t st = c->stream;
  LVX_HIP(c, hipMemsetAsync(info, 0, (size_t)(2 * nblk + 8) * 4, st));
  const double one = 1.0, mone = -1.0, zero = 0.0;
  int L = 0; while ((1 << L) < nblk) ++L;
  int info_pos = 0;
  for (int l = 0; l < L; ++l) {
    const int s = 1 << l, n2 = level_batch(nblk, nreal, l);
    if (n2 <= 0) continue;
    const long long sD = (long long)2 * s * bb, sG = (long long)2 * bb;
    double* Dj = D + (size_t)(s - 1) * bb;
    double* Dr = D + (size_t)(2 * s - 1) * bb;
    double* Gl = G + g_off(nblk, l, bb);
    double* Gn = G + g_off(nblk, l + 1, bb);
    double* LIj = LI ? LI + (size_t)(s - 1) * liS : nullptr;
    const long long sLI = (long long)2 * s * liS;
    if ((rc = potrf_batched(c, h, Dj, b, sD, info + info_pos, n2, LIj, sLI))) return rc;
    info_pos += n2;
    // X+_k = G[2k] C_k^-T : every ROW x of G[2k] solves C x^T = g^T
    // Y_k = C_k^-1 G[2k-1], k = 1..n2-1 : every COLUMN — in the same launch
    const TrsmSet ysolve{Dj + sD, sD, Gl + bb, 1, b, sG, b, n2 - 1, 0, LIj ? LIj + sLI : nullptr, sLI};
    const long long sZ = (long long)2 * s * b;
    double* Zj = Z ? Z + (size_t)(s - 1) * b : nullptr;
    double* Zr = Z ? Z + (size_t)(2 * s - 1) * b : nullptr;
    const TrsmSet rsolve{Dj, sD, Zj, 1, ldz, sZ, Z ? nrhs : 0, Z ? n2 : 0, 0, LIj, sLI};                    // y_j = C_j^-1 b_j
    if ((rc = trsv_batched<false>(c, Dj, b, sD, Gl, /*se*/ b, /*sv*/ 1, sG, b, n2, LIj, sLI, &ysolve, &rsolve))) return rc;
    if (schur_own(c, b, Z ? nrhs : 0)) {   // every Schur update of the level in one launch
      const SchurArgs sa{Gl, sG, (long long)bb, Dr, sD, Gn, Zj, Zr, sZ, ldz, Z ? nrhs : 0, b, n2};
      if ((rc = schur_level(c, sa))) return rc;
      continue;
    }
    // D_{j+s} -= X+ X+^T
    // (full GEMM instead of SYRK: rocBLAS' batched SYRK runs as many small launches; the upper triangle of D is never read)
    { LVX_BLAS_H(c, h); LVX_BLAS(c, g_vb.dgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_transpose, b, b, b, &mone, Gl, b, sG, Gl, b, sG, &one, Dr, b, sD, n2)); }
    if (n2 > 1) {
      // D_{j-s} -= Y^T Y   (left neighbour of eliminated k is the right neighbour of eliminated k-1)
      { LVX_BLAS_H(c, h); LVX_BLAS(c, g_vb.dgemm_strided_batched(h, rocblas_operation_transpose, rocblas_operation_none, b, b, b, &mone, Gl + bb, b, sG, Gl + bb, b, sG, &one, Dr, b, sD, n2 - 1)); }
      // next level's coupling A_{j+s,j-s} = -X+_k Y_k
      { LVX_BLAS_H(c, h); LVX_BLAS(c, g_vb.dgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_none, b, b, b, &mone, Gl + 2 * bb, b, sG, Gl + bb, b, sG, &zero, Gn, b, (rocblas_stride)bb, n2 - 1)); }
    }
    if (Z) {   // b_{j+s} -= X+ y_j,  b_{j-s} -= Y^T y_j
      { LVX_BLAS_H(c, h); LVX_BLAS(c, g_vb.dgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_none, b, nrhs, b, &mone, Gl, b, sG, Zj, ldz, sZ, &one, Zr, ldz, sZ, n2)); }
      if (n2 > 1)
        { LVX_BLAS_H(c, h); LVX_BLAS(c, g_vb.dgemm_strided_batched(h, rocblas_operation_transpose, rocblas_operation_none, b, nrhs, b, &mone, Gl + bb, b, sG, Zj + sZ, ldz, sZ, &one, Zr, ldz, sZ, n2 - 1)); }
    }
  }
  double* LIlast = LI ? LI + (size_t)(nblk - 1) * liS : nullptr;
  if ((rc = potrf_batched(c, h, D + (size_t)(nblk - 1) * bb, b, (long long)bb, info + info_pos, 1, LIlast, 0))) return rc;
  if (Z && (rc = trsv_batched<false>(c, D + (size_t)(nblk - 1) * bb, b, 0, Z + (size_t)(nblk - 1) * b, 1, ldz, 0, nrhs, 1, LIlast, 0))) return rc;
  info_pos += 1;
  hipLaunchKernelGGL(k_bcr_info, dim3((info_pos + 255) / 256), dim3(256), 0, st, (const int*)info, info_pos, info_out_d);
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}

// Zin: column-major [ldz x nrhs] right-hand sides; in place Zin <- L^-1 Zin (Zy aliases Zin; kept in the signature for the caller's bookkeeping)
int bcr_forward(lvx_ctx* c, double* Zin, double* Zy, int ldz, int nrhs) {
  rocblas_handle h = nullptr; int rc = LVX_OK;   // (the handle is fetched only where a library call is really made: blocks wider than 208)
  (void)Zy;
  double* Z = Zin;
  const int b = c->bcr_b, nblk = c->bcr_nblk;
  const size_t bb = (size_t)b * b;
  double* D = (double*)c->d_bcrD.p; double* G = (double*)c->d_bcrG.p;
  const double* LI = c->bcr_linv ? (const double*)c->d_bcrLinv.p : nullptr;
  const size_t liS = (size_t)((b + 15) / 16) * 256;
  const double one = 1.0, mone = -1.0;
  int L = 0; while ((1 << L) < nblk) ++L;
  for (int l = 0; l < L; ++l) {
    const int s = 1 << l, n2 = level_batch(nblk, c->bcr_nreal, l);
    if (n2 <= 0) continue;
    const long long sD = (long long)2 * s * bb, sG = (long long)2 * bb, sZ = (long long)2 * s * b;
    double* Dj = D + (size_t)(s - 1) * bb;
    double* Gl = G + g_off(nblk, l, bb);
    double* Zj = Z + (size_t)(s - 1) * b;
    double* Zr = Z + (size_t)(2 * s - 1) * b;
    if ((rc = trsv_batched<false>(c, Dj, b, sD, Zj, 1, ldz, sZ, nrhs, n2, LI ? LI + (size_t)(s - 1) * liS : nullptr, (long long)2 * s * liS))) return rc;                      // y_j = C_j^-1 b_j
    { LVX_BLAS_H(c, h); LVX_BLAS(c, g_vb.dgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_none, b, nrhs, b, &mone, Gl, b, sG, Zj, ldz, sZ, &one, Zr, ldz, sZ, n2)); }
    if (n2 > 1)
      { LVX_BLAS_H(c, h); LVX_BLAS(c, g_vb.dgemm_strided_batched(h, rocblas_operation_transpose, rocblas_operation_none, b, nrhs, b, &mone, Gl + bb, b, sG, Zj + sZ, ldz, sZ, &one, Zr, ldz, sZ, n2 - 1)); }
  }
  return trsv_batched<false>(c, D + (size_t)(nblk - 1) * bb, b, 0, Z + (size_t)(nblk - 1) * b, 1, ldz, 0, nrhs, 1, LI ? LI + (size_t)(nblk - 1) * liS : nullptr, 0);
}
// in place Zy <- L^-T Zy (Zx aliases Zy)
static int chain_backward(lvx_ctx* c, const BcrChain& ch, double* Z, int ldz, int nrhs);
int bcr_backward(lvx_ctx* c, double* Zy, double* Zx, int ldz, int nrhs) { (void)Zx; return chain_backward(c, ctx_chain(c), Zy, ldz, nrhs); }
static int chain_backward(lvx_ctx* c, const BcrChain& ch, double* Z, int ldz, int nrhs) {
  rocblas_handle h = nullptr; int rc = LVX_OK;   // (the handle is fetched only where a library call is really made: blocks wider than 208)
  const int b = ch.b, nblk = ch.nblk, nreal = ch.nreal;
  const size_t bb = (size_t)b * b;
  double* D = ch.D; double* G = ch.G;
  const double* LI = ch.LI;
  const size_t liS = (size_t)((b + 15) / 16) * 256;
  const double one = 1.0, mone = -1.0;
  int L = 0; while ((1 << L) < nblk) ++L;
  const size_t lds_fused = ((size_t)b * (b + 1) / 2 + 3 * (size_t)b + 16 + 16 * 17) * 8;
  const bool fused = nrhs == 1 && LI && lds_fused <= 160 * 1024;
  if (fused) LVX_HIP(c, hipFuncSetAttribute((const void*)k_bcr_back_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fused));
  if (fused) {   // the last block of the chain: the same kernel without neighbours (the 64-vector streaming solve took 63 us for this one vector)
    hipLaunchKernelGGL(k_bcr_back_level, dim3(1), dim3(BACK_NT), lds_fused, c->stream, (const double*)(D + (size_t)(nblk - 1) * bb), 0ll, LI + (size_t)(nblk - 1) * liS, 0ll, (const double*)nullptr, 0ll,
                       Z, (long long)(nblk - 1) * b, 0ll, 0ll, b, 1, (long long)bb, c->sw.deterministic);
    LVX_HIP(c, hipGetLastError());
  } else if ((rc = trsv_batched<true>(c, D + (size_t)(nblk - 1) * bb, b, 0, Z + (size_t)(nblk - 1) * b, 1, ldz, 0, nrhs, 1, LI ? LI + (size_t)(nblk - 1) * liS : nullptr, 0))) return rc;
  for (int l = L - 1; l >= 0; --l) {
    const int s = 1 << l, n2 = level_batch(nblk, nreal, l);
    if (n2 <= 0) continue;
    const long long sD = (long long)2 * s * bb, sG = (long long)2 * bb, sZ = (long long)2 * s * b;
    double* Dj = D + (size_t)(s - 1) * bb;
    double* Gl = G + g_off(nblk, l, bb);
    double* Zj = Z + (size_t)(s - 1) * b;
    double* Zr = Z + (size_t)(2 * s - 1) * b;
    if (fused) {   // one launch per level: both matrix-vector products and the transposed solve of every eliminated block
      hipLaunchKernelGGL(k_bcr_back_level, dim3((unsigned)n2), dim3(BACK_NT), lds_fused, c->stream, (const double*)Dj, sD, LI + (size_t)(s - 1) * liS, (long long)2 * s * liS, (const double*)Gl, sG,
                         Z, (long long)(s - 1) * b, (long long)(2 * s - 1) * b, sZ, b, n2, (long long)bb, c->sw.deterministic);
      LVX_HIP(c, hipGetLastError());
      continue;
    }
    { LVX_BLAS_H(c, h); LVX_BLAS(c, g_vb.dgemm_strided_batched(h, rocblas_operation_transpose, rocblas_operation_none, b, nrhs, b, &mone, Gl, b, sG, Zr, ldz, sZ, &one, Zj, ldz, sZ, n2)); }
    if (n2 > 1)
      { LVX_BLAS_H(c, h); LVX_BLAS(c, g_vb.dgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_none, b, nrhs, b, &mone, Gl + bb, b, sG, Zr, ldz, sZ, &one, Zj + sZ, ldz, sZ, n2 - 1)); }
    if ((rc = trsv_batched<true>(c, Dj, b, sD, Zj, 1, ldz, sZ, nrhs, n2, LI ? LI + (size_t)(s - 1) * liS : nullptr, (long long)2 * s * liS))) return rc;
  }
  return LVX_OK;
}
// M (n x n, column-major, both triangles) = Z^T Z for the tall-skinny Z [ldz x n] (column-major, ldz = nblk * b rows; n = border + 1 <= 80).  Split-K by hand on the matrix
// cores: a workgroup takes GRAM_ROWS rows, stages 64 of them at a time in LDS as P[row][col] (coalesced loads along the rows, odd row stride), and every 4 panel rows are
// one k-step of each upper tile pair — the fragment of column tile c at lane l, P[4 ks + (l >> 4)][16 c + (l & 15)], is the A operand of Z^T and the B operand of Z, as in
// the evaluation kernels; the tile pairs are dealt to the four wavefronts.  Partial Grams per workgroup, then k_sum_partials.  Replaced rocBLAS' strided-batched GEMM
// (one small GEMM per row block, 54 us): the last library call of a solve with blocks up to 208.  The layouts of this library have 22 (no hub) or 52 border variables: n = 23 / 53,
// the 3- and 4-tile instances; the 5-tile one and the library path beyond 80 columns are there for completeness.
#define GRAM_ROWS 256
#define GRAM_NT 5
#define GRAM_LDP (16 * GRAM_NT + 1)
__host__ __device__ constexpr int gram_ci(int nt, int t) { int ci = 0; while (ci < nt && t >= nt - ci) { t -= nt - ci; ++ci; } return ci; }
__host__ __device__ constexpr int gram_cj(int nt, int t) { int ci = 0; while (ci < nt && t >= nt - ci) { t -= nt - ci; ++ci; } return ci + t; }
// the 16 k-steps of a staged panel for wavefront WV: its tile pairs are WV, WV + 4, ... of the NT (NT + 1) / 2 upper pairs — compile-time, so every column-tile fragment is
// read from LDS once per k-step and stays in a named register
template <int NT, int WV> __device__ __forceinline__ void gram_steps(const double* P, int lane, d4* D) {
  constexpr int NP = NT * (NT + 1) / 2;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const double* src = P + (4 * ks + (lane >> 4)) * GRAM_LDP + (lane & 15);
    double f[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) f[c] = src[16 * c];
#pragma unroll
    for (int q = 0; q < (NP - WV + 3) / 4; ++q) D[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[gram_ci(NT, WV + 4 * q)], f[gram_cj(NT, WV + 4 * q)], D[q], 0, 0, 0);
  }
}
template <int NT>
__global__ __launch_bounds__(256) void k_gram_mfma(const double* __restrict__ Z, int ldz, int m, int n, double* __restrict__ part, int nz) {   // nz != 0: Z row-major [m][nz]
  __shared__ double P[64 * GRAM_LDP];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NP = NT * (NT + 1) / 2, MAXP = (NP + 3) / 4;
  const int r_begin = blockIdx.x * GRAM_ROWS, r_end = min(m, r_begin + GRAM_ROWS);
  d4 D[MAXP];
#pragma unroll
  for (int q = 0; q < MAXP; ++q) D[q] = d4{0.0, 0.0, 0.0, 0.0};
  // the next 64 rows are in flight (registers) while the matrix cores work on the staged ones
  double v[4 * NT];
  auto fetch = [&](int r0) {
    if (nz) {   // row-major: consecutive threads take consecutive columns of a row (element tid + 256 u of the 64 x 16 NT stage)
#pragma unroll
      for (int u = 0; u < 4 * NT; ++u) { const int e = tid + 256 * u, row = e / (16 * NT), j = e % (16 * NT), i = r0 + row; v[u] = (j < n && i < r_end) ? Z[(size_t)i * nz + j] : 0.0; }
      return;
    }
    const int i = r0 + lane;
#pragma unroll
    for (int u = 0; u < 4 * NT; ++u) { const int j = wv + 4 * u; v[u] = (j < n && i < r_end) ? Z[(size_t)i + (size_t)j * ldz] : 0.0; }
  };
  fetch(r_begin);
  for (int r0 = r_begin; r0 < r_end; r0 += 64) {
    if (nz) {
#pragma unroll
      for (int u = 0; u < 4 * NT; ++u) { const int e = tid + 256 * u; P[(e / (16 * NT)) * GRAM_LDP + e % (16 * NT)] = v[u]; }
    } else {
#pragma unroll
    for (int u = 0; u < 4 * NT; ++u) P[lane * GRAM_LDP + wv + 4 * u] = v[u];
    }
    __syncthreads();
    if (r0 + 64 < r_end) fetch(r0 + 64);
    switch (wv) {
      case 0: gram_steps<NT, 0>(P, lane, D); break;
      case 1: gram_steps<NT, 1>(P, lane, D); break;
      case 2: gram_steps<NT, 2>(P, lane, D); break;
      default: gram_steps<NT, 3>(P, lane, D); break;
    }
    __syncthreads();
  }
  // partial Gram in TILE layout [pair][register][lane]: coalesced stores (matrix layout scattered 16 four-double segments per instruction: the epilogue of 587
  // workgroups cost more than their products); k_sum_tiles puts the sums where they belong
  double* out = part + (size_t)blockIdx.x * NP * 256;
#pragma unroll
  for (int q = 0; q < MAXP; ++q) {
    const int t = wv + 4 * q;
    if (t >= NP) break;
#pragma unroll
    for (int vv = 0; vv < 4; ++vv) out[(t * 4 + vv) * 64 + lane] = D[q][vv];
  }
}
__global__ void k_sum_tiles(const double* P, int nt, int nparts, int per, int n, double* M) {
  const int np = nt * (nt + 1) / 2, e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= np * 256) return;
  const int p0 = blockIdx.y * per, p1 = min(nparts, p0 + per);
  double s = 0.0;
  for (int p = p0; p < p1; ++p) s += P[(size_t)p * np * 256 + e];
  if (s == 0.0) return;
  const int t = e >> 8, vv = (e >> 6) & 3, lane = e & 63, ci = gram_ci(nt, t), cj = gram_cj(nt, t);
  const int row = 16 * ci + (lane >> 4) + 4 * vv, col = 16 * cj + (lane & 15);
  if (row >= n || col >= n) return;
  atomicAdd(&M[row + (size_t)col * n], s);
  if (ci != cj) atomicAdd(&M[col + (size_t)row * n], s);
}
__global__ void k_sum_partials(const double* P, int nn, int nparts, int per, double* M) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nn) return;
  const int p0 = blockIdx.y * per, p1 = min(nparts, p0 + per);
  double s = 0.0;
  for (int p = p0; p < p1; ++p) s += P[(size_t)p * nn + e];
  if (s != 0.0) atomicAdd(&M[e], s);
}
int bcr_gram(lvx_ctx* c, const double* Z, int ldz, int n, double* M, int row_major_nz) {
  const int b = c->bcr_b, nblk = c->bcr_nblk;
  const size_t nn = (size_t)n * n;
  const int m = row_major_nz ? c->nb : ldz;   // rows of Z: nblk * b for the chain of the band (rows past the band are zero); the band's own for the leaves + separators elimination (nothing writes the rows behind them)
  int rc;
  const double one = 1.0, zero = 0.0;
  LVX_HIP(c, hipMemsetAsync(M, 0, nn * 8, c->stream));
  const int per = c->sw.deterministic ? (1 << 30) : 16;   // deterministic mode: one adder per entry, the partial Grams in index order
  if (n <= 16 * GRAM_NT) {
    const int nparts = (m + GRAM_ROWS - 1) / GRAM_ROWS, nt = n <= 48 ? 3 : (n <= 64 ? 4 : 5), np = nt * (nt + 1) / 2;
    if ((rc = dev_alloc(c, c->d_Y2, (size_t)nparts * np * 256 * 8))) return rc;
    if (nt == 3) hipLaunchKernelGGL(k_gram_mfma<3>, dim3((unsigned)nparts), dim3(256), 0, c->stream, Z, ldz, m, n, (double*)c->d_Y2.p, row_major_nz);
    else if (nt == 4) hipLaunchKernelGGL(k_gram_mfma<4>, dim3((unsigned)nparts), dim3(256), 0, c->stream, Z, ldz, m, n, (double*)c->d_Y2.p, row_major_nz);
    else hipLaunchKernelGGL(k_gram_mfma<5>, dim3((unsigned)nparts), dim3(256), 0, c->stream, Z, ldz, m, n, (double*)c->d_Y2.p, row_major_nz);
    hipLaunchKernelGGL(k_sum_tiles, dim3((unsigned)((np * 256 + 255) / 256), (unsigned)((nparts + (long long)per - 1) / per)), dim3(256), 0, c->stream, (const double*)c->d_Y2.p, nt, nparts, per, n, M);
  } else {   // (more than 80 border columns: one small library GEMM per row block)
    if (row_major_nz) return fail(c, LVX_E_STATE, "row-major right-hand sides with more than 80 columns");
    rocblas_handle h; if ((rc = bcr_handle(c, &h))) return rc;
    if ((rc = dev_alloc(c, c->d_Y2, (size_t)nblk * nn * 8))) return rc;
    LVX_BLAS(c, g_vb.dgemm_strided_batched(h, rocblas_operation_transpose, rocblas_operation_none, n, n, b, &one, Z, ldz, (rocblas_stride)b, Z, ldz, (rocblas_stride)b,
                                              &zero, (double*)c->d_Y2.p, n, (rocblas_stride)nn, nblk));
    hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((nn + 255) / 256), (unsigned)((nblk + (long long)per - 1) / per)), dim3(256), 0, c->stream, (const double*)c->d_Y2.p, (int)nn, nblk, per, M);
  }
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}

#include "lvx_nd.h"

void bcr_destroy(lvx_ctx* c) {
  nd_destroy(c);
  if (!c->blas) return;
  std::lock_guard<std::mutex> lk(g_blas_mu);
  if (g_blas_free.size() < 16) g_blas_free.emplace_back(c->device, (rocblas_handle)c->blas); else if (g_vb.destroy_handle) (void)g_vb.destroy_handle((rocblas_handle)c->blas);
  c->blas = nullptr;
}

}  // namespace lvx
