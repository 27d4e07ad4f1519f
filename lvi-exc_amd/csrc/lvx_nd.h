// lvx_nd.h — leaves + separators elimination of the band (included at the end of lvx_bcr.hip, inside namespace lvx).
//
// The uniform chain of lvx_bcr.hip cuts the band into blocks as wide as its WIDEST column (b = 180 at config 4: the 50 co-visibility windows of the camera) and pays
// b^3 per block everywhere — but 94 % of the columns couple only to the 4 neighbouring knots (IMU, LiDAR: reach 23, 29 with a free time offset).  Here the band is cut
// by its column profile (lvx_ctx::h_colhi, ensure_layout) instead:
//
//     leaf | separator | leaf | separator | ... | leaf          separator = the <= 32 columns everything to its left can reach: the leaves on its two sides do not couple
//
//   * narrow leaves (a few hundred columns of reach <= 32): band Cholesky in 16 x 16 tiles, U = L^T with three tiles per tile column, ONE wavefront per leaf walks the
//     chain (k_nd_factor); the right-hand sides — the coupling to the left separator (32 columns), to the right one (last rows only) and the solver's own columns — are
//     solved against it by a workgroup per leaf, a wavefront per 16 columns, and the products the separators need are summed on the way (k_nd_solve);
//   * dense leaves (a wide run of <= 192 columns: one co-visibility window): the register-resident Cholesky and the LDS triangular solves of the chain kernels,
//     batched over the runs, on a second stream beside the narrow leaves;
//   * the separators: a block tridiagonal system with 32 x 32 blocks, D_k -= W_R^T W_R (leaf on the left) + W_L^T W_L (leaf on the right), A_{k+1,k} = -W_R^T W_L —
//     block cyclic reduction again, but with level kernels of its own for this block size (k_c32_level: one launch per level; k_c32_back).
// Backward: separators first (k_c32_back, level by level), then every leaf on its own, x_I = U^-1 (y_I - W_L x_left - W_R x_right).
// Every tile lives in the MFMA accumulator layout (row = (lane >> 4) + 4 reg, col = lane & 15) and is stored as it stands (index = reg * 64 + lane: 512-byte coalesced
// rows); such a tile is the A operand of its TRANSPOSE and the B operand of itself, so U^T Y, W^T Y and inv(L) X (from the stored TRANSPOSE of the triangle's inverse)
// need no data movement between the products.
// What the solver sees: Z = L^-1 [B^T | -g_b] S (in the elimination order: a row permutation the Gram Z^T Z does not see) as OUTPUT — the right-hand sides themselves are never
// built: every kernel forms its entries from the border rows where it loads them (nd_rhs) —, ROW-major [nd_ldz rows][nd_nz] (bcr_gram's row-major staging,
// k_sub_border_rm in lvx_solver.hip), then one vector back.
// Applies when the profile allows it (nd_plan); otherwise, and as the reference in the tests (switch SOLVER_ND = -1), the uniform chain runs.
#pragma once

struct NdLeaf { int c0, m, nt, sl, sr, cl, wl, cr, wr, toff, pr0, slot, dense; };   // columns [c0, c0 + m); separators left / right (-1: none), their first column and width; first
                                                                                   // tile column in the tile storage; first tile row the right separator couples to; index among its kind
struct NdSep { int c0, w, lf, rt; };                                                // columns [c0, c0 + w), w <= 32; leaf on the left / right (-1: none)
#define ND_WS 32
#define ND_DENSE_MAX 192   // columns of a dense leaf: what the LDS-resident triangular solves and the fused backward kernel of the chain kernels hold (k_trsm_lds, k_bcr_back_level)
// LDS-only synchronisation: __syncthreads() also waits for every outstanding GLOBAL load and store (s_waitcnt vmcnt(0)) — the prefetched tiles and the results on their way
// out, once per tile row: 5 - 10 us each.  ND_WAVE_LDS: one wavefront, its own LDS writes before its reads; ND_LDS_BARRIER: workgroup.
// ND_KEEP4: a prefetched tile stays a LOADED value until here — without it the compiler forms next iteration's products (scaling, the negated MFMA operand) right behind the
// loads and waits for them at the top of the loop, i.e. no prefetch at all
#define ND_KEEP4(x) asm volatile("" : "+v"((x)[0]), "+v"((x)[1]), "+v"((x)[2]), "+v"((x)[3]))
#define ND_KEEP1(x) asm volatile("" : "+v"(x))
#define ND_WAVE_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define ND_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
struct NdPlan {
  bool active = false;
  int epoch = -1, mode = 0, nrhs = 0, zt = 0, nc = 0, ldz = 0;
  int nleaf = 0, nnar = 0, nden = 0, nsep = 0, ntile = 0, bd = 0, maxnt = 0, nblk2 = 0;
  std::vector<NdLeaf> leaves; std::vector<NdSep> seps;
  DevBuf leaf, sep, nar, den, U, WL, WR, GO, Dc, Rc, LIc, info, D2, G2, F2, Z2, Y2, info2, zb2, tc, Fd;   // Fd: tiles of the dense border's factor (k_dense_tiles)
  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_trsm = nullptr, ev_rc = nullptr;
};
struct NdArgs {
  const NdLeaf* leaf; const NdSep* sep; const int* nar; const int* den;
  const double* Hs; const double* scale; const double* lmd; double ir; int ld, npre, nb;
  double *U, *WL, *WR, *GO; int nc, nrhs;
  const double* Bs; const double* gbs; int nbd;   // the right-hand sides are formed where they are loaded (nd_rhs): [B^T | -g_b] scaled — no array of them is built first
  double* Z; int ldz, nz;   // Z row-major [ldz rows][nz]: a 16 x 16 tile is four 128-byte rows per load (column-major: sixteen 32-byte pieces — measured: NOT what bounded k_nd_solve; kept, the Gram and the border subtraction read rows as well)
  int* info; double* trash;   // trash: 64 words nobody reads (the target of masked stores that stay unconditional)
};

__device__ __forceinline__ d4 nd_mm(const d4& A, const d4& B, d4 C) {    // C + A^T B
#pragma unroll
  for (int r = 0; r < 4; ++r) C = __builtin_amdgcn_mfma_f64_16x16x4f64(A[r], B[r], C, 0, 0, 0);
  return C;
}
__device__ __forceinline__ d4 nd_mmn(const d4& A, const d4& B, d4 C) {   // C - A^T B
#pragma unroll
  for (int r = 0; r < 4; ++r) C = __builtin_amdgcn_mfma_f64_16x16x4f64(-A[r], B[r], C, 0, 0, 0);
  return C;
}
// entry (band position row, column col) of the solver's right-hand sides: columns 0 .. nbd - 1 the border rows (S B^T S), column nbd the band's own -S g_b
__device__ __forceinline__ double nd_rhs(const NdArgs& a, int row, int col) {
  if (col < a.nbd) return a.Bs[(size_t)col * a.nb + row] * a.scale[a.nb + col] * a.scale[row];
  return col == a.nbd ? -a.gbs[row] * a.scale[row] : 0.0;
}
// Entries of tile (rows 16 pr .., columns 16 pc ..), pr <= pc, of a narrow leaf as they lie in the band storage (the diagonal tile mirrored).  nd_leaf_raw only LOADS
// (clamped index, unconditional: a conditional load becomes a branch with its own s_waitcnt, and thirty of them in a row were 8 us per tile column); nd_leaf_ok says
// which entries count — inside the leaf and within the npre entries a narrow column holds (k_clear leaves the rest of its storage alone).  The mask is applied where
// the tile is USED, one iteration after the load: a select right behind the load is a use, and the wait for it sat at the top of the loop.
__device__ __forceinline__ bool nd_leaf_ok(const NdArgs& a, int m, int pr, int pc, int q, int j, int r) {
  const int R = 16 * pr + q + 4 * r, Cc = 16 * pc + j, hi = max(R, Cc), d = hi - min(R, Cc);
  return pr >= 0 && hi < m && d < a.npre;
}
__device__ __forceinline__ d4 nd_leaf_raw(const NdArgs& a, int c0, int m, int pr, int pc, int q, int j) {
  d4 X;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int R = 16 * pr + q + 4 * r, Cc = 16 * pc + j, lo = min(R, Cc), d = max(R, Cc) - lo;
    X[r] = a.Hs[nd_leaf_ok(a, m, pr, pc, q, j, r) ? (size_t)(c0 + lo) * a.ld + d : 0];
  }
  return X;
}
__device__ __forceinline__ d4 nd_rows4(const double* v, int c0, int m, int p, int q) {   // v [c0 + 16 p + q + 4 r] (clamped: whatever lies past the leaf multiplies a masked entry)
  d4 X;
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int R = 16 * p + q + 4 * r; X[r] = v[(p >= 0 && R < m) ? c0 + R : 0]; }
  return X;
}

// One wavefront per narrow leaf: U (p-2, p) = inv(L_{p-2}) A (p-2, p),  U (p-1, p) = inv(L_{p-1}) (A (p-1, p) - U (p-2, p-1)^T U (p-2, p)),
// U (p, p) = chol (A (p, p) - U (p-2, p)^T U (p-2, p) - U (p-1, p)^T U (p-1, p)); per tile column [U (p-2, p) | U (p-1, p) | U (p, p) | inv(L_p)^T] goes to a.U.
// A = S H S + diag(lmd) / radius as k_bcr_build forms it (an untouched variable — zero diagonal — gets the pivot 1: any pivot gives y = 0).
// Four leaves per workgroup, a wavefront each (no workgroup barrier anywhere): 109 workgroups sit on 109 CUs and leave the others EMPTY for the dense leaves' Cholesky, whose
// workgroups need a whole CU's register file (a wavefront per workgroup spread the 436 over every CU: k_potrf_reg<12> then waited for CUs to drain).
__global__ __launch_bounds__(256) void k_nd_factor(NdArgs a, int nnar) {
  __shared__ double tr4[4][16 * 17];
  const int wv_ = threadIdx.x >> 6, leaf_i = blockIdx.x * 4 + wv_;
  if (leaf_i >= nnar) return;
  double* tr = tr4[wv_];
  const NdLeaf lf = a.leaf[a.nar[leaf_i]];
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  const int c0 = lf.c0, m = lf.m;
  const d4 zero4 = d4{0.0, 0.0, 0.0, 0.0};
  d4 U1p = zero4, LIT1 = zero4, LIT2 = zero4;
  // raw tiles, row scales of the three tile rows, column scale / damping of tile column p
  d4 H0 = zero4, H1 = zero4, H2 = nd_leaf_raw(a, c0, m, 0, 0, q, j), S0 = zero4, S1 = zero4, S2 = nd_rows4(a.scale, c0, m, 0, q);
  auto col1 = [&](const double* v, int p) { return v[16 * p + j < m ? c0 + 16 * p + j : 0]; };
  double sc = col1(a.scale, 0), lc = col1(a.lmd, 0);
  int bad = 0;
  double* Ut = a.U + (size_t)lf.toff * 1024;
  for (int p = 0; p < lf.nt; ++p) {
    // the next tile column's loads fly under this one's chain (past the last one: clamped, unused)
    const d4 N0 = nd_leaf_raw(a, c0, m, p - 1, p + 1, q, j), N1 = nd_leaf_raw(a, c0, m, p, p + 1, q, j), N2 = nd_leaf_raw(a, c0, m, p + 1, p + 1, q, j), SN = nd_rows4(a.scale, c0, m, p + 1, q);
    const double scn = col1(a.scale, p + 1), lcn = col1(a.lmd, p + 1);
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks these loads to the end of the iteration: their latency then sits on the chain)
    d4 A0, A1, T;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      A0[r] = nd_leaf_ok(a, m, p - 2, p, q, j, r) ? H0[r] * S0[r] * sc : 0.0;
      A1[r] = nd_leaf_ok(a, m, p - 1, p, q, j, r) ? H1[r] * S1[r] * sc : 0.0;
      const int R = 16 * p + q + 4 * r, Cc = 16 * p + j;
      double v = nd_leaf_ok(a, m, p, p, q, j, r) ? H2[r] * S2[r] * sc : 0.0;
      if (R == Cc) v = R < m ? (H2[r] == 0.0 ? 1.0 : v + lc * a.ir) : 1.0;
      T[r] = v;
    }
    d4 U0 = zero4, U1 = zero4;
    if (p >= 2) { U0 = nd_mm(LIT2, A0, zero4); A1 = nd_mmn(U1p, U0, A1); T = nd_mmn(U0, U0, T); }
    if (p >= 1) { U1 = nd_mm(LIT1, A1, zero4); T = nd_mmn(U1, U1, T); }
    d4 Mres;
    const int bc = chol16_mfma(T, Mres, q, j);
    bad = (bc > 0 && bad == 0 && 16 * p + bc <= m) ? 16 * p + bc : bad;
#pragma unroll
    for (int r = 0; r < 4; ++r) tr[(q + 4 * r) * 17 + j] = Mres[r];
    ND_WAVE_LDS();
    d4 LIT;
#pragma unroll
    for (int r = 0; r < 4; ++r) LIT[r] = tr[j * 17 + q + 4 * r];
    ND_WAVE_LDS();
    double* out = Ut + (size_t)p * 1024;
#pragma unroll
    for (int r = 0; r < 4; ++r) { out[r * 64 + lane] = U0[r]; out[256 + r * 64 + lane] = U1[r]; out[512 + r * 64 + lane] = T[r]; out[768 + r * 64 + lane] = LIT[r]; }
    d4 K0 = N0, K1 = N1, K2 = N2, KS = SN; double k1 = scn, k2 = lcn;
    ND_KEEP4(K0); ND_KEEP4(K1); ND_KEEP4(K2); ND_KEEP4(KS); ND_KEEP1(k1); ND_KEEP1(k2);
    U1p = U1; LIT2 = LIT1; LIT1 = LIT; H0 = K0; H1 = K1; H2 = K2; S0 = S1; S1 = S2; S2 = KS; sc = k1; lc = k2;
  }
  if (lane == 0) a.info[leaf_i] = bad;
}

// A workgroup per narrow leaf, a wavefront per 16 right-hand sides: wavefronts 0, 1 the coupling to the left separator (W_L, dense: the forward substitution fills it
// down the whole leaf), 2 .. 1 + ZT the solver's columns, the last two the coupling to the right separator (W_R: nothing above tile row pr0).
//   Y_p = inv(L_p) (X_p - U (p-2, p)^T Y_{p-2} - U (p-1, p)^T Y_{p-1})
// and, with the W tiles of the row passed through LDS, what the separators need: GO [64 x nc] = [W_L | W_R]^T [W_L | W_R | Y].
// ROLE 0: W_L, 1: the solver's columns, 2: W_R — compile-time per wavefront (with a run-time role every role-dependent load became a scalar branch with its own wait)
template <int ROLE>
__device__ __forceinline__ void nd_solve_role(const NdArgs& a, const NdLeaf& lf, int li, int t, int lane, double (*xch)[4][256], double (*ubuf)[768]) {
  const int q = lane >> 4, j = lane & 15, col = 16 * t + j;
  const d4 zero4 = d4{0.0, 0.0, 0.0, 0.0};
  const double* Ut = a.U + (size_t)lf.toff * 1024;
  const bool colok = ROLE == 0 ? (lf.sl >= 0 && col < lf.wl) : (ROLE == 2 ? (lf.sr >= 0 && col < lf.wr) : col < a.nrhs);
  const double scol = ROLE == 1 ? 1.0 : a.scale[colok ? (ROLE == 0 ? lf.cl : lf.cr) + col : 0];
  // The solver's columns (ROLE 1): loads only (clamped index, unconditional), the mask where the tile is used — as in k_nd_factor.  The couplings (ROLE 0 / 2) are
  // non-zero in the first two / the last three tile rows only: those tiles are formed BEFORE the loop (their scattered loads with a wait each sat in every iteration
  // and, through the barrier, in every wavefront's: 60 % of the kernel).
  auto rhs_ok = [&](int p, int r) {
    const int R = 16 * p + q + 4 * r;
    const bool in = p < lf.nt && R < lf.m && colok;
    if (ROLE == 1) return in;
    const int d = ROLE == 0 ? lf.c0 + R - (lf.cl + col) : lf.cr + col - (lf.c0 + R);
    return in && d < a.npre && d >= 0;
  };
  // (ROLE 1) raw entry of [B^T | -g_b] and the row's scale; the column's scale (and sign) is scol1
  const double scol1 = ROLE == 1 ? (col < a.nbd ? a.scale[a.nb + min(col, a.nbd - 1)] : -1.0) : 0.0;
  auto rhs_raw = [&](int p, d4& X, d4& S) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int R = 16 * p + q + 4 * r;
      const bool ok = rhs_ok(p, r);
      X[r] = col < a.nbd ? a.Bs[ok ? (size_t)col * a.nb + lf.c0 + R : 0] : a.gbs[ok ? lf.c0 + R : 0];
      S[r] = a.scale[ok ? lf.c0 + R : 0];
    }
  };
  auto coupling_tile = [&](int p) {
    d4 X;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int R = 16 * p + q + 4 * r;
      const bool ok = rhs_ok(p, r);
      const int d = ROLE == 0 ? lf.c0 + R - (lf.cl + col) : lf.cr + col - (lf.c0 + R);
      const double hv = a.Hs[ok ? (ROLE == 0 ? (size_t)(lf.cl + col) * a.ld + d : (size_t)(lf.c0 + R) * a.ld + d) : 0], sr = a.scale[ok ? lf.c0 + R : 0];
      X[r] = ok ? hv * sr * scol : 0.0;
    }
    return X;
  };
  d4 C0 = zero4, C1 = zero4, C2 = zero4;
  const int cbase = ROLE == 0 ? 0 : lf.pr0;
  if (ROLE != 1) { C0 = coupling_tile(cbase); C1 = coupling_tile(cbase + 1); if (ROLE == 2) C2 = coupling_tile(cbase + 2); }
  // The three tiles of tile column p ([U (p-2, p) | U (p-1, p) | . | inv(L_p)^T]) reach the wavefronts through LDS: the workgroup's first 384 threads carry two words each
  // of column p + 1 from global memory (requested one iteration ahead) into ubuf[(p + 1) & 1] before the barrier of iteration p; every wavefront reads ubuf[p & 1] as MFMA
  // operands.  (Each wavefront loading the tiles itself: eight times the traffic and 48 registers of prefetch.)
  const int tid = threadIdx.x;
  auto u_fetch = [&](int p, double& w0, double& w1) {   // words 2 tid, 2 tid + 1 of [U0 | U1 | LIT] of tile column p
    const int pc = min(p, lf.nt - 1), e = 2 * tid, tile = e >> 8, which = tile == 2 ? 3 : tile;
    const double* src = Ut + (size_t)pc * 1024 + which * 256 + (e & 255);
    const bool ok = tid < 384;
    w0 = src[ok ? 0 : -(e & 255) - which * 256]; w1 = src[ok ? 1 : -(e & 255) - which * 256];
  };
  auto u_put = [&](int p, double w0, double w1) { if (tid < 384) { ubuf[p & 1][2 * tid] = w0; ubuf[p & 1][2 * tid + 1] = w1; } };
  auto u_tile = [&](int p, int tile) {
    d4 X;
#pragma unroll
    for (int r = 0; r < 4; ++r) X[r] = ubuf[p & 1][tile * 256 + r * 64 + lane];
    return X;
  };
  d4 Y1 = zero4, Y2 = zero4, GL0 = zero4, GL1 = zero4, GR0 = zero4, GR1 = zero4;
#ifdef LVX_ND_KT
  long long kt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long kt0_ = __builtin_amdgcn_s_memtime(); const long long kts_ = kt0_;
#define NKT(i) { const long long n_ = __builtin_amdgcn_s_memtime(); kt_[i] += n_ - kt0_; kt0_ = n_; }
#define NKT_USE(x) { const int u_ = __builtin_amdgcn_readfirstlane(__double2hiint(x)); asm volatile("" :: "s"(u_)); }
#else
#define NKT(i)
#define NKT_USE(x)
#endif
  d4 X = zero4, XS = zero4;
  if (ROLE == 1) rhs_raw(0, X, XS);
  double w0, w1;
  u_fetch(0, w0, w1); u_put(0, w0, w1);
  u_fetch(1, w0, w1);
  ND_LDS_BARRIER();
  for (int p = 0; p < lf.nt; ++p) {
    d4 Xn = zero4, XSn = zero4;
    if (ROLE == 1) rhs_raw(p + 1, Xn, XSn);
    double w0n, w1n;
    u_fetch(p + 2, w0n, w1n);
    __builtin_amdgcn_sched_barrier(0);   // (keep the next row's loads up here, ahead of this row's products)
    NKT(0)
    d4 Y = zero4;
    if (ROLE != 2 || p >= lf.pr0) {
      d4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = ROLE == 1 ? (rhs_ok(p, r) ? X[r] * XS[r] * scol1 : 0.0) : (p == cbase ? C0[r] : (p == cbase + 1 ? C1[r] : (p == cbase + 2 ? C2[r] : 0.0)));
      if (p >= 2) acc = nd_mmn(u_tile(p, 0), Y2, acc);
      if (p >= 1) acc = nd_mmn(u_tile(p, 1), Y1, acc);
      Y = nd_mm(u_tile(p, 2), acc, zero4);
      NKT_USE(Y[0]); NKT(1)
      if (ROLE == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a.WL[((size_t)lf.toff * 16 + 16 * p + q + 4 * r) * 32 + col] = Y[r];
      } else if (ROLE == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // unconditional (a masked entry goes to the lane's trash word): four exec-masked branches otherwise
          const int R = 16 * p + q + 4 * r;
          double* dst = (colok && R < lf.m) ? a.Z + (size_t)(lf.c0 + R) * a.nz + col : a.trash + lane;
          *dst = Y[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) a.WR[((size_t)lf.slot * 48 + 16 * (p - lf.pr0) + q + 4 * r) * 32 + col] = Y[r];
      }
    }
    if (ROLE != 1) {
      double* slot = xch[p & 1][ROLE == 0 ? t : 2 + t];
#pragma unroll
      for (int r = 0; r < 4; ++r) slot[r * 64 + lane] = Y[r];
    }
    u_put(p + 1, w0, w1);
    NKT(2)
    ND_LDS_BARRIER();
    NKT(3)
    if (ROLE != 2) {
      d4 W;
#pragma unroll
      for (int r = 0; r < 4; ++r) W[r] = xch[p & 1][0][r * 64 + lane];
      GL0 = nd_mm(W, Y, GL0);
#pragma unroll
      for (int r = 0; r < 4; ++r) W[r] = xch[p & 1][1][r * 64 + lane];
      GL1 = nd_mm(W, Y, GL1);
    }
    if (p >= lf.pr0) {
      d4 W;
#pragma unroll
      for (int r = 0; r < 4; ++r) W[r] = xch[p & 1][2][r * 64 + lane];
      GR0 = nd_mm(W, Y, GR0);
#pragma unroll
      for (int r = 0; r < 4; ++r) W[r] = xch[p & 1][3][r * 64 + lane];
      GR1 = nd_mm(W, Y, GR1);
    }
    NKT_USE(GL1[0]); NKT_USE(GR1[0]); NKT(4)
    d4 KX = Xn, KS = XSn;
    if (ROLE == 1) { ND_KEEP4(KX); ND_KEEP4(KS); }
    ND_KEEP1(w0n); ND_KEEP1(w1n);
    Y2 = Y1; Y1 = Y; X = KX; XS = KS; w0 = w0n; w1 = w1n;
#ifdef LVX_ND_KT
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#endif
    NKT(5)
  }
#ifdef LVX_ND_KT
  if (lane == 0 && (blockIdx.x == 7 || blockIdx.x == 300) && t == 0) printf("NKT blk %d role %d nt %d: issue %lld chain %lld store %lld barrier %lld gram %lld wait %lld total %lld\n", (int)blockIdx.x, ROLE, lf.nt, kt_[0], kt_[1], kt_[2], kt_[3], kt_[4], kt_[5], __builtin_amdgcn_s_memtime() - kts_);
#endif
  double* go = a.GO + (size_t)li * 64 * a.nc + (ROLE == 0 ? col : (ROLE == 1 ? 64 + col : 32 + col));
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = q + 4 * r;
    if (ROLE != 2) { go[(size_t)row * a.nc] = GL0[r]; go[(size_t)(16 + row) * a.nc] = GL1[r]; }
    go[(size_t)(32 + row) * a.nc] = GR0[r]; go[(size_t)(48 + row) * a.nc] = GR1[r];
  }
}
template <int ZT>
__global__ __launch_bounds__(64 * (4 + ZT), 4) void k_nd_solve(NdArgs a) {   // (<= 128 registers: two workgroups per CU fill each other's barrier waits)
  __shared__ double xch[2][4][256];
  __shared__ double ubuf[2][768];
  const int li = a.nar[blockIdx.x];
  const NdLeaf lf = a.leaf[li];
  const int tid = threadIdx.x, lane = tid & 63;
  // two workgroups share a CU (a wavefront's SIMD is its index mod 4): the W_L wavefronts carry twice the products of the others, so every other workgroup deals its
  // roles two places on — SIMDs 0, 1 get the heavy pair of one workgroup, SIMDs 2, 3 that of the other
  const int wv = (__builtin_amdgcn_readfirstlane(tid >> 6) + (((blockIdx.x >> 8) & 1) ? 2 : 0)) % (4 + ZT);
  if (wv < 2) nd_solve_role<0>(a, lf, li, wv, lane, xch, ubuf);
  else if (wv < 2 + ZT) nd_solve_role<1>(a, lf, li, wv - 2, lane, xch, ubuf);
  else nd_solve_role<2>(a, lf, li, wv - 2 - ZT, lane, xch, ubuf);
}

// dense leaves: D_c (bd x bd, lower, column-major; identity padding) and the right-hand sides R_c [bd x nc] = [A_L | A_R | Z rows], column-major
__global__ __launch_bounds__(256) void k_nd_cbuild(NdArgs a, double* Dc, double* Rc, int bd, int bw, int which) {   // which 0: D_c, 1: R_c
  const NdLeaf lf = a.leaf[a.den[blockIdx.x]];
  double* D = Dc + (size_t)blockIdx.x * bd * bd;
  double* R = Rc + (size_t)blockIdx.x * bd * a.nc;
  const int cbeg = which == 1 ? bd : 0, ncol = which == 0 ? bd : bd + a.nc;   // which 0: D_c, 1: R_c, 2: both
  for (int cc = cbeg + blockIdx.y * 4 + (threadIdx.x >> 6); cc < ncol; cc += gridDim.y * 4) {
    const int lane = threadIdx.x & 63;
    if (cc < bd) {
      const double sc = cc < lf.m ? a.scale[lf.c0 + cc] : 0.0;
      for (int rr = cc + lane; rr < bd; rr += 64) {
        double v = rr == cc ? 1.0 : 0.0;
        if (rr < lf.m) {
          const int d = rr - cc;
          const double hv = d <= bw ? a.Hs[(size_t)(lf.c0 + cc) * a.ld + d] : 0.0;
          v = hv * a.scale[lf.c0 + rr] * sc;
          if (d == 0) v = hv == 0.0 ? 1.0 : v + a.lmd[lf.c0 + cc] * a.ir;
        }
        D[(size_t)cc * bd + rr] = v;
      }
    } else {
      const int c = cc - bd;
      for (int i = lane; i < bd; i += 64) {
        double v = 0.0;
        if (i < lf.m) {
          if (c < 32) {
            const int g = lf.cl + c, d = lf.c0 + i - g;
            if (lf.sl >= 0 && c < lf.wl && d < a.npre) v = a.Hs[(size_t)g * a.ld + d] * a.scale[lf.c0 + i] * a.scale[g];
          } else if (c < 64) {
            const int gs = lf.cr + c - 32, d = gs - (lf.c0 + i);
            if (lf.sr >= 0 && c - 32 < lf.wr && d <= bw) v = a.Hs[(size_t)(lf.c0 + i) * a.ld + d] * a.scale[lf.c0 + i] * a.scale[gs];
          } else if (c - 64 < a.nrhs) v = nd_rhs(a, lf.c0 + i, c - 64);
        }
        R[(size_t)c * bd + i] = v;
      }
    }
  }
}
// GO of a dense leaf from its solved right-hand sides R_c [bd x nc] (column-major), and its rows of Z back in place.  64 rows at a time through LDS (a column of R_c is
// contiguous: coalesced loads, row-major stage), a wavefront per 16 columns: G [64 x 16] += W^T X with both fragments read from the stage as in k_gram_mfma.
// (A wavefront loading its tiles straight from the column-major array: sixteen 32-byte pieces per load, one exposed round trip per tile row — 62 us for 50 leaves.)
#define CG_LDP 145   // >= 64 + 16 * 5, odd
template <int ZT>
__global__ __launch_bounds__(64 * (4 + ZT)) void k_nd_cgram(NdArgs a, const double* Rc, int bd) {
  __shared__ double S[64 * CG_LDP];
  const int li = a.den[blockIdx.x];
  const NdLeaf lf = a.leaf[li];
  const double* R = Rc + (size_t)blockIdx.x * bd * a.nc;
  constexpr int NW = 4 + ZT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int t = __builtin_amdgcn_readfirstlane(tid >> 6);
  d4 G[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) G[g] = d4{0.0, 0.0, 0.0, 0.0};
  for (int r0 = 0; r0 < bd; r0 += 64) {
    for (int c = t; c < a.nc; c += NW) S[lane * CG_LDP + c] = r0 + lane < bd ? R[(size_t)c * bd + r0 + lane] : 0.0;
    __syncthreads();
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      const double* src = S + (4 * ks + (lane >> 4)) * CG_LDP + (lane & 15);
      const double x = src[16 * t];
#pragma unroll
      for (int g = 0; g < 4; ++g) G[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(src[16 * g], x, G[g], 0, 0, 0);
    }
    // the solved right-hand sides of these rows back to the solver's (row-major) array
    for (int e = tid; e < 64 * a.nz; e += 64 * NW) {
      const int row = e / a.nz, z = e % a.nz, i = r0 + row;
      if (i < lf.m && z < a.nrhs) a.Z[(size_t)(lf.c0 + i) * a.nz + z] = S[row * CG_LDP + 64 + z];
    }
    __syncthreads();
  }
  double* go = a.GO + (size_t)li * 64 * a.nc + 16 * t + (lane & 15);
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) go[(size_t)(16 * g + (lane >> 4) + 4 * r) * a.nc] = G[g][r];
}

// separator k of the reduced chain: D2_k = H (S_k, S_k) - W_R^T W_R (left leaf) - W_L^T W_L (right leaf), A_{k+1,k} = -W_R^T W_L (right leaf), its rows of the right-hand sides
__global__ __launch_bounds__(256) void k_nd_assemble(NdArgs a, double* D2, double* G2, double* Z2, int ldz2, int nsep) {
  const int k = blockIdx.x;
  const NdSep sp = a.sep[k];
  const double* GoL = sp.lf >= 0 ? a.GO + (size_t)sp.lf * 64 * a.nc : nullptr;
  const double* GoR = sp.rt >= 0 ? a.GO + (size_t)sp.rt * 64 * a.nc : nullptr;
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int i = e & 31, c = e >> 5;
    double v = 0.0;
    if (i >= c) {
      if (i < sp.w) {
        const int d = i - c;
        const double hv = d < a.npre ? a.Hs[(size_t)(sp.c0 + c) * a.ld + d] : 0.0;
        v = hv * a.scale[sp.c0 + i] * a.scale[sp.c0 + c];
        if (d == 0) v = hv == 0.0 ? 1.0 : v + a.lmd[sp.c0 + c] * a.ir;
        if (GoL) v -= GoL[(size_t)(32 + i) * a.nc + 32 + c];
        if (GoR) v -= GoR[(size_t)i * a.nc + c];
      } else v = i == c ? 1.0 : 0.0;
    }
    D2[(size_t)k * 1024 + e] = v;
    G2[(size_t)k * 1024 + e] = (GoR && k + 1 < nsep) ? -GoR[(size_t)(32 + i) * a.nc + c] : 0.0;
  }
  for (int e = threadIdx.x; e < 32 * a.nrhs; e += 256) {
    const int i = e & 31, z = e >> 5;
    double v = 0.0;
    if (i < sp.w) {
      v = nd_rhs(a, sp.c0 + i, z);
      if (GoL) v -= GoL[(size_t)(32 + i) * a.nc + 64 + z];
      if (GoR) v -= GoR[(size_t)i * a.nc + 64 + z];
    }
    Z2[((size_t)k * 32 + i) * ldz2 + z] = v;   // (ldz2: columns of the row-major separator right-hand sides)
  }
}
__global__ __launch_bounds__(256) void k_nd_scatter(NdArgs a, const double* Z2, int ldz2) {
  const NdSep sp = a.sep[blockIdx.x];
  for (int e = threadIdx.x; e < 32 * a.nrhs; e += 256) {
    const int i = e & 31, z = e >> 5;
    if (i < sp.w) a.Z[(size_t)(sp.c0 + i) * a.nz + z] = Z2[((size_t)blockIdx.x * 32 + i) * ldz2 + z];
  }
}
__global__ void k_nd_gather1(const NdSep* sep, int nsep, int nblk2, const double* zb, double* zb2) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nblk2 * 32) return;
  const int k = e >> 5, i = e & 31;
  double v = 0.0;
  if (k < nsep) { const NdSep sp = sep[k]; if (i < sp.w) v = zb[sp.c0 + i]; }
  zb2[e] = v;
}
// backward of the narrow leaves (blocks < nnar), and the separators' unknowns back to their rows (the blocks after them)
__global__ __launch_bounds__(64) void k_nd_back(NdArgs a, int nnar, const double* zb2, double* zb) {
  extern __shared__ double sh[];
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= nnar) {
    const int k = blockIdx.x - nnar;
    const NdSep sp = a.sep[k];
    if (lane < sp.w) zb[sp.c0 + lane] = zb2[k * 32 + lane];
    return;
  }
  const NdLeaf lf = a.leaf[a.nar[blockIdx.x]];
  double* xs = sh;            // [64] unknowns of the left / right separator
  double* vv = sh + 64;       // [16]
  double* tv = sh + 80;       // [16 nt]
  xs[lane] = lane < 32 ? (lf.sl >= 0 ? zb2[lf.sl * 32 + lane] : 0.0) : (lf.sr >= 0 ? zb2[lf.sr * 32 + lane - 32] : 0.0);
  __syncthreads();
  for (int i = lane; i < 16 * lf.nt; i += 64) {
    double tt = i < lf.m ? zb[lf.c0 + i] : 0.0;
    if (lf.sl >= 0) {
      const double* w = a.WL + ((size_t)lf.toff * 16 + i) * 32;
#pragma unroll 8
      for (int c = 0; c < 32; ++c) tt -= w[c] * xs[c];
    }
    if (lf.sr >= 0 && i >= 16 * lf.pr0) {
      const double* w = a.WR + ((size_t)lf.slot * 48 + i - 16 * lf.pr0) * 32;
#pragma unroll 8
      for (int c = 0; c < 32; ++c) tt -= w[c] * xs[32 + c];
    }
    tv[i] = tt;
  }
  __syncthreads();
  const double* Ut = a.U + (size_t)lf.toff * 1024;
  const int part = lane >> 4, i = lane & 15;
  const int eo = (i >> 2) * 64 + (i & 3) * 16 + 4 * part;   // element (row i, columns 4 part ..) of a stored tile
  double ua[4], ub[4], lt[4];
  auto fetch = [&](int p) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ua[u] = p + 1 < lf.nt ? Ut[(size_t)(p + 1) * 1024 + 256 + eo + u] : 0.0;   // U (p, p + 1)
      ub[u] = p + 2 < lf.nt ? Ut[(size_t)(p + 2) * 1024 + eo + u] : 0.0;         // U (p, p + 2)
      lt[u] = Ut[(size_t)p * 1024 + 768 + eo + u];                               // inv(L_p)^T
    }
  };
  fetch(lf.nt - 1);
  for (int p = lf.nt - 1; p >= 0; --p) {
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (p + 1 < lf.nt) s += ua[u] * tv[16 * (p + 1) + 4 * part + u];
      if (p + 2 < lf.nt) s += ub[u] * tv[16 * (p + 2) + 4 * part + u];
    }
    double l4[4] = {lt[0], lt[1], lt[2], lt[3]};
    if (p > 0) fetch(p - 1);
    s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
    if (part == 0) vv[i] = tv[16 * p + i] - s;
    ND_WAVE_LDS();
    double x = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) x += l4[u] * vv[4 * part + u];
    x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
    ND_WAVE_LDS();
    if (part == 0) tv[16 * p + i] = x;
    ND_WAVE_LDS();
  }
  for (int k = lane; k < lf.m; k += 64) zb[lf.c0 + k] = tv[k];
}
// dense leaves, backward: t = y - W_L x_left - W_R x_right (k_bcr_back_level without neighbours then solves C^T x = t), and the result back to its rows
__global__ __launch_bounds__(256) void k_nd_cprep(NdArgs a, const double* Rc, int bd, const double* zb2, const double* zb, double* tc) {
  __shared__ double xs[64];
  const NdLeaf lf = a.leaf[a.den[blockIdx.x]];
  const double* R = Rc + (size_t)blockIdx.x * bd * a.nc;
  if (threadIdx.x < 64) { const int l = threadIdx.x; xs[l] = l < 32 ? (lf.sl >= 0 ? zb2[lf.sl * 32 + l] : 0.0) : (lf.sr >= 0 ? zb2[lf.sr * 32 + l - 32] : 0.0); }
  __syncthreads();
  for (int i = threadIdx.x; i < bd; i += 256) {
    double tt = 0.0;
    if (i < lf.m) {
      tt = zb[lf.c0 + i];
      for (int c = 0; c < 64; ++c) tt -= R[(size_t)c * bd + i] * xs[c];
    }
    tc[(size_t)blockIdx.x * bd + i] = tt;
  }
}
__global__ __launch_bounds__(256) void k_nd_cscatter(NdArgs a, int bd, const double* tc, double* zb) {
  const NdLeaf lf = a.leaf[a.den[blockIdx.x]];
  for (int i = threadIdx.x; i < lf.m; i += 256) zb[lf.c0 + i] = tc[(size_t)blockIdx.x * bd + i];
}

// ---------------------------------------------------------------------------------------------------------
// The separator chain: block cyclic reduction with b = 32, ONE launch per level (the general level kernels of lvx_bcr.hip — Cholesky, triangular solves, Schur updates:
// three launches of kernels shaped for b ~ 180 — cost 13 + 8 + 9 us per level on these 32 x 32 blocks, 0.3 ms per solve for the nine levels).
// Level l: the workgroup of the REMAINING block r = j + s (s = 2^l) factorises its left eliminated neighbour j (and keeps it: factor, solved couplings and right-hand
// sides for the backward sweep) and, once more, its right one j + 2 s (which the workgroup of r + 2 s keeps) — two 32 x 32 factorisations instead of a grid-wide
// dependency —, solves their couplings and right-hand sides and applies
//     D_r -= X+ X+^T + Y_R^T Y_R,   A (r, r - 2 s) = -X+ Y_L,   z_r -= X+ y_L + Y_R^T y_R.
// Tiles in the accumulator layout through LDS as in the leaves; Xt = X+^T = inv(C_L) A (r, j)^T is what is solved and kept.  Right-hand sides row-major [nblk * 32][nz];
// the solved ones go to a SECOND array (the right neighbour's workgroup still reads the unsolved rows).  Couplings per level in the layout of the general chain (g_off).
// F [nblk][13][256]: per eliminated block U00, U01, U11, inv(L0)^T, inv(L1)^T, Xt (4 tiles: 5 + 2 rowtile + coltile), Y_L (9 + ...).
// ---------------------------------------------------------------------------------------------------------
struct C32 { double* D; double* G; double* Z; double* Y; double* F; int* info; int nblk, nz, zt; };
__device__ __forceinline__ size_t c32_goff(int nblk, int l) { size_t o = 0; for (int k = 0; k < l; ++k) o += (size_t)(nblk >> k) * 1024; return o; }
// Cholesky of a 32 x 32 block (column-major lower in global memory) by one wavefront: tiles [U00 | U01 | U11 | inv(L0)^T | inv(L1)^T] to LDS (as they stand) and, if
// keep, to global memory; returns the 1-based first bad pivot (0: none).  ONE copy of the 16 x 16 factorisation's ~800 instructions, run twice by a loop and shared by
// every call (noinline): executed once from a cold instruction cache it costs 4.7 k cycles against 0.8 k warm (docs/HISTORY.md, round 3) — two inlined copies per call
// made this function 15 k cycles, more than half of a level's kernel.
__device__ __attribute__((noinline)) int c32_factor(const double* D, double (*fac)[256], double* keep, double* tr, int lane) {
  const int q = lane >> 4, j = lane & 15;
  const d4 zero4 = d4{0.0, 0.0, 0.0, 0.0};
  d4 T, A01, T11;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = q + 4 * r, lo = min(row, j), hi = max(row, j);
    T[r] = D[lo * 32 + hi]; A01[r] = D[row * 32 + 16 + j]; T11[r] = D[(16 + lo) * 32 + 16 + hi];
  }
  d4 U01 = zero4;
  int bad = 0;
#ifdef LVX_ND_KT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long fk[6]; int fki = 0; fk[fki++] = __builtin_amdgcn_s_memtime();
#endif
#pragma nounroll
  for (int p = 0; p < 2; ++p) {
    d4 M;
    const int b = chol16_mfma(T, M, q, j);
#ifdef LVX_ND_KT
    { const int u_ = __builtin_amdgcn_readfirstlane(__double2hiint(M[0])); asm volatile("" :: "s"(u_)); fk[fki++] = __builtin_amdgcn_s_memtime(); }
#endif
    bad = bad ? bad : (b ? 16 * p + b : 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) tr[(q + 4 * r) * 17 + j] = M[r];
    ND_WAVE_LDS();
    d4 LIT;
#pragma unroll
    for (int r = 0; r < 4; ++r) LIT[r] = tr[j * 17 + q + 4 * r];
    ND_WAVE_LDS();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = r * 64 + lane;
      fac[2 * p][e] = T[r]; fac[3 + p][e] = LIT[r];            // U00 / U11, inv(L0)^T / inv(L1)^T
      if (keep) { keep[2 * p * 256 + e] = T[r]; keep[(3 + p) * 256 + e] = LIT[r]; }
    }
    if (p == 0) {
      U01 = nd_mm(LIT, A01, zero4);
      T = nd_mmn(U01, U01, T11);
#pragma unroll
      for (int r = 0; r < 4; ++r) { fac[1][r * 64 + lane] = U01[r]; if (keep) keep[256 + r * 64 + lane] = U01[r]; }
    }
#ifdef LVX_ND_KT
    { const int u_ = __builtin_amdgcn_readfirstlane(__double2hiint(T[0])); asm volatile("" :: "s"(u_)); fk[fki++] = __builtin_amdgcn_s_memtime(); }
#endif
  }
#ifdef LVX_ND_KT
  if (lane == 0 && blockIdx.x == 0 && keep) printf("FKT chol0 %lld rest0 %lld chol1 %lld rest1 %lld\n", fk[1] - fk[0], fk[2] - fk[1], fk[3] - fk[2], fk[4] - fk[3]);
#endif
  return bad;
}
__device__ __forceinline__ d4 c32_lds(const double* t, int lane) { d4 X;
#pragma unroll
  for (int r = 0; r < 4; ++r) X[r] = t[r * 64 + lane];
  return X; }
// V = inv(C) R for a 32 x 16 column tile R = [R0; R1]: V0 = inv(L0) R0, V1 = inv(L1) (R1 - U01^T V0)
__device__ __forceinline__ void c32_solve(const double (*fac)[256], const d4& R0, const d4& R1, d4& V0, d4& V1, int lane) {
  const d4 zero4 = d4{0.0, 0.0, 0.0, 0.0};
  V0 = nd_mm(c32_lds(fac[3], lane), R0, zero4);
  V1 = nd_mm(c32_lds(fac[4], lane), nd_mmn(c32_lds(fac[1], lane), V0, R1), zero4);
}
#define C32_MAXT 16
__global__ __launch_bounds__(512) void k_c32_level(C32 c, int l, int last) {
  extern __shared__ double sh[];
  double (*facL)[256] = (double (*)[256])sh;                 // [5][256]
  double (*facR)[256] = (double (*)[256])(sh + 5 * 256);     // [5][256]
  double (*V)[2][256] = (double (*)[2][256])(sh + 10 * 256); // [tasks][2][256]
  double* tr = sh + 10 * 256 + C32_MAXT * 512;               // [8][16 * 17]
  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = blockIdx.x, s = 1 << l, count = c.nblk >> l;
  const int jL = s - 1 + 2 * s * k, r = jL + s, jR = r + s;
  const bool hasR = 2 * k + 2 < count, hasYL = k >= 1;
  const double* Gl = c.G + c32_goff(c.nblk, l);
  double* Gn = c.G + c32_goff(c.nblk, l + 1);
  const int zt = c.zt, nz = c.nz;
  double* FL = c.F + (size_t)jL * 13 * 256;
  const d4 zero4 = d4{0.0, 0.0, 0.0, 0.0};
  // every global load of the kernel is issued up front — the right-hand sides of the (up to two) solves of this wavefront and the tiles its (up to two) outputs update —
  // so the three phases share ONE memory round trip (three, one per phase: 16 us per level)
  const int ntL = 4 + zt, ntask = hasR ? ntL + 2 + zt : ntL;   // tasks: 0, 1 Xt | 2, 3 Y_L | 4 .. 3 + zt y_L | then Y_R (2), y_R (zt)
  auto task_load = [&](int t, d4& R0, d4& R1) {
    R0 = zero4; R1 = zero4;
    if (t >= ntask) return;
    const bool right = t >= ntL;
    const int tt = right ? t - ntL + 2 : t;
    const int kind = tt < 2 ? 0 : (tt < 4 ? 1 : 2), ct = kind == 0 ? tt : (kind == 1 ? tt - 2 : tt - 4);
    if (kind == 0) {
      const double* g = Gl + (size_t)(2 * k) * 1024;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) { R0[rr] = g[(q + 4 * rr) * 32 + 16 * ct + j]; R1[rr] = g[(16 + q + 4 * rr) * 32 + 16 * ct + j]; }
    } else if (kind == 1) {
      if (right || hasYL) {
        const double* g = Gl + (size_t)(right ? 2 * k + 1 : 2 * k - 1) * 1024;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { R0[rr] = g[(16 * ct + j) * 32 + q + 4 * rr]; R1[rr] = g[(16 * ct + j) * 32 + 16 + q + 4 * rr]; }
      }
    } else {
      const double* z = c.Z + (size_t)(right ? jR : jL) * 32 * nz + 16 * ct + j;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) { R0[rr] = z[(size_t)(q + 4 * rr) * nz]; R1[rr] = z[(size_t)(16 + q + 4 * rr) * nz]; }
    }
  };
  // outputs: 0 .. 2 D_r tiles (0,0), (1,0), (1,1) | 3 .. 6 coupling (ti, tc) | 7 .. z_r tiles (ti, ct)
  const int nout = 7 + 2 * zt;
  auto out_ptr = [&](int o, int rr) -> double* {
    if (o < 3) { const int ti = o == 0 ? 0 : 1, tc = o == 2 ? 1 : 0; return c.D + (size_t)r * 1024 + (16 * tc + j) * 32 + 16 * ti + q + 4 * rr; }
    if (o < 7) { const int ti = (o - 3) >> 1, tc = (o - 3) & 1; return Gn + (size_t)(hasYL ? k - 1 : 0) * 1024 + (16 * tc + j) * 32 + 16 * ti + q + 4 * rr; }
    const int ti = (o - 7) / zt, ct = (o - 7) % zt;
    return c.Z + ((size_t)r * 32 + 16 * ti + q + 4 * rr) * nz + 16 * ct + j;
  };
  auto out_load = [&](int o) {
    d4 X = zero4;
    if (o < nout && (o < 3 || o >= 7)) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) X[rr] = *out_ptr(o, rr);
    }
    return X;
  };
#ifdef LVX_ND_KT
  const long long ck0 = __builtin_amdgcn_s_memtime();
#endif
  d4 Ra0, Ra1, Rb0, Rb1;
  task_load(wv, Ra0, Ra1); task_load(wv + 8, Rb0, Rb1);
  d4 Oa = out_load(wv), Ob = out_load(wv + 8);
  if (wv == 0) { const int bad = c32_factor(c.D + (size_t)jL * 1024, facL, FL, tr, lane); if (lane == 0) c.info[jL] = bad; }
  else if (wv == 1 && hasR) (void)c32_factor(c.D + (size_t)jR * 1024, facR, nullptr, tr + 16 * 17, lane);
#ifdef LVX_ND_KT
  const long long ck1 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
#ifdef LVX_ND_KT
  const long long ck2 = __builtin_amdgcn_s_memtime();
#endif
  // solves
  for (int t = wv, it = 0; t < ntask; t += 8, ++it) {
    const bool right = t >= ntL;
    const int tt = right ? t - ntL + 2 : t;            // 0, 1 Xt | 2, 3 Y | 4 .. y
    const int kind = tt < 2 ? 0 : (tt < 4 ? 1 : 2), ct = kind == 0 ? tt : (kind == 1 ? tt - 2 : tt - 4);
    d4 V0, V1;
    c32_solve(right ? facR : facL, it == 0 ? Ra0 : Rb0, it == 0 ? Ra1 : Rb1, V0, V1, lane);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) { V[t][0][rr * 64 + lane] = V0[rr]; V[t][1][rr * 64 + lane] = V1[rr]; }
    if (!right) {   // what the backward sweep needs of the eliminated block j_L
      if (kind < 2) {
        double* f = FL + (size_t)(kind == 0 ? 5 : 9) * 256;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { f[(0 + ct) * 256 + rr * 64 + lane] = V0[rr]; f[(2 + ct) * 256 + rr * 64 + lane] = V1[rr]; }
      } else {
        double* y = c.Y + (size_t)jL * 32 * nz + 16 * ct + j;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { y[(size_t)(q + 4 * rr) * nz] = V0[rr]; y[(size_t)(16 + q + 4 * rr) * nz] = V1[rr]; }
      }
    }
  }
#ifdef LVX_ND_KT
  const long long ck3 = __builtin_amdgcn_s_memtime();
#endif
  ND_LDS_BARRIER();
#ifdef LVX_ND_KT
  const long long ck4 = __builtin_amdgcn_s_memtime();
#endif
  // products
  for (int o = wv, it = 0; o < nout; o += 8, ++it) {
    d4 acc = zero4;
    const d4 old = it == 0 ? Oa : (it == 1 ? Ob : out_load(o));   // (a third round only with five column tiles of right-hand sides)
    if (o < 3) {
      const int ti = o == 0 ? 0 : 1, tc = o == 2 ? 1 : 0;
#pragma unroll
      for (int ka = 0; ka < 2; ++ka) {
        acc = nd_mm(c32_lds(V[ti][ka], lane), c32_lds(V[tc][ka], lane), acc);
        if (hasR) acc = nd_mm(c32_lds(V[ntL + ti][ka], lane), c32_lds(V[ntL + tc][ka], lane), acc);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) { const int row = 16 * ti + q + 4 * rr, col = 16 * tc + j; if (row >= col) *out_ptr(o, rr) = old[rr] - acc[rr]; }
    } else if (o < 7) {
      if (hasYL) {
        const int ti = (o - 3) >> 1, tc = (o - 3) & 1;
#pragma unroll
        for (int ka = 0; ka < 2; ++ka) acc = nd_mm(c32_lds(V[ti][ka], lane), c32_lds(V[2 + tc][ka], lane), acc);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) *out_ptr(o, rr) = -acc[rr];
      }
    } else {
      const int ct = (o - 7) % zt, ti = (o - 7) / zt;
#pragma unroll
      for (int ka = 0; ka < 2; ++ka) {
        acc = nd_mm(c32_lds(V[ti][ka], lane), c32_lds(V[4 + ct][ka], lane), acc);
        if (hasR) acc = nd_mm(c32_lds(V[ntL + ti][ka], lane), c32_lds(V[ntL + 2 + ct][ka], lane), acc);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) *out_ptr(o, rr) = old[rr] - acc[rr];
    }
  }
#ifdef LVX_ND_KT
  if (lane == 0 && k == 0 && (wv == 0 || wv == 3) && l < 3) printf("C32KT l %d wv %d: to-factor-end %lld barrier %lld solves %lld barrier %lld products %lld\n", l, wv, ck1 - ck0, ck2 - ck1, ck3 - ck2, ck4 - ck3, (long long)__builtin_amdgcn_s_memtime() - ck4);
#endif
  if (!last) return;
  // the last level leaves ONE block: factor it and solve its right-hand sides here
  __syncthreads();
  double* FR = c.F + (size_t)r * 13 * 256;
  if (wv == 0) { const int bad = c32_factor(c.D + (size_t)r * 1024, facL, FR, tr, lane); if (lane == 0) c.info[r] = bad; }
  __syncthreads();
  for (int ct = wv; ct < zt; ct += 8) {
    const double* z = c.Z + (size_t)r * 32 * nz + 16 * ct + j;
    d4 R0, R1, V0, V1;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) { R0[rr] = z[(size_t)(q + 4 * rr) * nz]; R1[rr] = z[(size_t)(16 + q + 4 * rr) * nz]; }
    c32_solve(facL, R0, R1, V0, V1, lane);
    double* y = c.Y + (size_t)r * 32 * nz + 16 * ct + j;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) { y[(size_t)(q + 4 * rr) * nz] = V0[rr]; y[(size_t)(16 + q + 4 * rr) * nz] = V1[rr]; }
  }
}
// backward sweep of a level for ONE vector, a wavefront per eliminated block: x_j = U^-1 (x_j - Xt x_r - Y_L x_{j - s}); top: the last block first
__device__ __forceinline__ double c32_mv(const double* tile, const double* x16, int lane) {   // (tile . x) [lane & 15], the tile as it stands in memory
  const int part = lane >> 4, i = lane & 15;
  const double* t = tile + (i >> 2) * 64 + (i & 3) * 16 + 4 * part;
  double s = t[0] * x16[4 * part] + t[1] * x16[4 * part + 1] + t[2] * x16[4 * part + 2] + t[3] * x16[4 * part + 3];
  s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
  return s;
}
__device__ __forceinline__ void c32_usolve(const double* F, double* v, int lane) {   // v [32] (LDS) <- U^-1 v
  const int i = lane & 15;
  const double x1 = c32_mv(F + 4 * 256, v + 16, lane);
  ND_WAVE_LDS();
  if (lane < 16) v[16 + i] = x1;
  ND_WAVE_LDS();
  const double t0 = v[i] - c32_mv(F + 1 * 256, v + 16, lane);
  ND_WAVE_LDS();
  if (lane < 16) v[i] = t0;
  ND_WAVE_LDS();
  const double x0 = c32_mv(F + 3 * 256, v, lane);
  ND_WAVE_LDS();
  if (lane < 16) v[i] = x0;
  ND_WAVE_LDS();
}
__device__ __forceinline__ void c32_back_block(const C32& c, int l, int k, int top, double* zb, double* xr, double* xl, double* v, int lane) {
  const int i = lane & 15;
  const int s = 1 << l;
  const int jL = s - 1 + 2 * s * k, r = jL + s;
  const double* F = c.F + (size_t)jL * 13 * 256;
  if (top) {
    if (lane < 32) v[lane] = zb[r * 32 + lane];
    ND_WAVE_LDS();
    c32_usolve(c.F + (size_t)r * 13 * 256, v, lane);
    if (lane < 32) zb[r * 32 + lane] = v[lane];
    ND_WAVE_LDS();
  }
  if (lane < 32) { xr[lane] = top ? v[lane] : zb[r * 32 + lane]; xl[lane] = k >= 1 ? zb[(jL - s) * 32 + lane] : 0.0; }
  ND_WAVE_LDS();
  if (lane < 32) v[lane] = zb[jL * 32 + lane];
  ND_WAVE_LDS();
  // v -= Xt x_r + Y_L x_l   (tiles 5 + 2 a + b: rows 16 a .., columns 16 b ..)
  double a0 = c32_mv(F + 5 * 256, xr, lane) + c32_mv(F + 6 * 256, xr + 16, lane);
  double a1 = c32_mv(F + 7 * 256, xr, lane) + c32_mv(F + 8 * 256, xr + 16, lane);
  if (k >= 1) {
    a0 += c32_mv(F + 9 * 256, xl, lane) + c32_mv(F + 10 * 256, xl + 16, lane);
    a1 += c32_mv(F + 11 * 256, xl, lane) + c32_mv(F + 12 * 256, xl + 16, lane);
  }
  ND_WAVE_LDS();
  if (lane < 16) { v[i] -= a0; v[16 + i] -= a1; }
  ND_WAVE_LDS();
  c32_usolve(F, v, lane);
  if (lane < 32) zb[jL * 32 + lane] = v[lane];
}
__global__ __launch_bounds__(64) void k_c32_back(C32 c, int l, int top, double* zb) {
  __shared__ double xr[32], xl[32], v[32];
  c32_back_block(c, l, blockIdx.x, top, zb, xr, xl, v, threadIdx.x);
}
// the levels with at most eight eliminated blocks (the top of the tree: 8, 4, 2, 1 — or fewer) in ONE launch: a wavefront per block, a workgroup barrier between the
// levels (four launches of 4 us with as much again between them otherwise)
__global__ __launch_bounds__(512) void k_c32_back_top(C32 c, int l_hi, int l_lo, double* zb) {
  __shared__ double xr[8][32], xl[8][32], v[8][32];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int l = l_hi; l >= l_lo; --l) {
    const int n2 = c.nblk >> (l + 1);
    if (wv < n2) c32_back_block(c, l, wv, l == l_hi ? 1 : 0, zb, xr[wv], xl[wv], v[wv], lane);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// The dense border system (n <= 64: hub knots + calibration scalars of a single sequence) on the same tiles: Cholesky and forward substitution by ONE wavefront
// (k_dense_tiles), backward substitution by one (k_dense_tiles_back).  The LDS-resident column-by-column kernel of lvx_solver.hip (k_dense_partial: three workgroup
// barriers per column, 50 us at n = 52) stays for the joint solve, which eliminates only the private part of the border.
// S: row-major lower triangle; F [NT * NT + NT][256]: U (p, q), q >= p, at p * NT + q, inv(L_p)^T at NT * NT + p — tiles as they stand.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double nd_colsum(const d4& T, const double* x16, int lane) {   // (T^T x) [lane & 15] from a register tile
  const int q = lane >> 4;
  double s = T[0] * x16[q] + T[1] * x16[q + 4] + T[2] * x16[q + 8] + T[3] * x16[q + 12];
  s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
  return s;
}
template <int NT>
__global__ __launch_bounds__(64) void k_dense_tiles(const double* __restrict__ S, double* rhs, int n, int* info, double* F) {
  __shared__ double tr[16 * 17], yv[16 * NT], vt[16];
  const int lane = threadIdx.x, q = lane >> 4, j = lane & 15;
  const d4 zero4 = d4{0.0, 0.0, 0.0, 0.0};
  d4 U[NT][NT], LIT[NT];
#pragma unroll
  for (int p = 0; p < NT; ++p)
#pragma unroll
    for (int c = p; c < NT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * p + q + 4 * r, col = 16 * c + j, lo = min(row, col), hi = max(row, col);
        const double v = S[hi < n ? (size_t)hi * n + lo : 0];
        U[p][c][r] = hi < n ? v : (row == col ? 1.0 : 0.0);
      }
  for (int e = lane; e < 16 * NT; e += 64) yv[e] = e < n ? rhs[e] : 0.0;
  int bad = 0;
#pragma unroll
  for (int p = 0; p < NT; ++p) {
    d4 T = U[p][p];
#pragma unroll
    for (int r2 = 0; r2 < p; ++r2) T = nd_mmn(U[r2][p], U[r2][p], T);
    d4 M;
    const int b = chol16_mfma(T, M, q, j);
    bad = (bad == 0 && b > 0 && 16 * p + b <= n) ? 16 * p + b : bad;
    U[p][p] = T;
#pragma unroll
    for (int r = 0; r < 4; ++r) tr[(q + 4 * r) * 17 + j] = M[r];
    ND_WAVE_LDS();
#pragma unroll
    for (int r = 0; r < 4; ++r) LIT[p][r] = tr[j * 17 + q + 4 * r];
    ND_WAVE_LDS();
#pragma unroll
    for (int c = p + 1; c < NT; ++c) {
      d4 X = U[p][c];
#pragma unroll
      for (int r2 = 0; r2 < p; ++r2) X = nd_mmn(U[r2][p], U[r2][c], X);
      U[p][c] = nd_mm(LIT[p], X, zero4);
    }
  }
#pragma unroll
  for (int p = 0; p < NT; ++p) {
#pragma unroll
    for (int c = p; c < NT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) F[(size_t)(p * NT + c) * 256 + r * 64 + lane] = U[p][c][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) F[(size_t)(NT * NT + p) * 256 + r * 64 + lane] = LIT[p][r];
  }
  // forward substitution: y_p = inv(L_p) (r_p - sum_{c < p} U (c, p)^T y_c)
  ND_WAVE_LDS();
#pragma unroll
  for (int p = 0; p < NT; ++p) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < p; ++c) s += nd_colsum(U[c][p], yv + 16 * c, lane);
    const double v = yv[16 * p + j] - s;
    ND_WAVE_LDS();
    if (q == 0) vt[j] = v;
    ND_WAVE_LDS();
    const double y = nd_colsum(LIT[p], vt, lane);
    ND_WAVE_LDS();
    if (q == 0) yv[16 * p + j] = y;
    ND_WAVE_LDS();
  }
  for (int e = lane; e < n; e += 64) rhs[e] = yv[e];
  if (lane == 0 && bad && info[1] == 0) { info[1] = bad; info[2] = 0; info[3] = 0; }
}
template <int NT>
__global__ __launch_bounds__(64) void k_dense_tiles_back(const double* __restrict__ F, double* rhs, int n) {
  __shared__ double yv[16 * NT], vt[16];
  const int lane = threadIdx.x, i = lane & 15;
  for (int e = lane; e < 16 * NT; e += 64) yv[e] = e < n ? rhs[e] : 0.0;
  ND_WAVE_LDS();
#pragma unroll
  for (int p = NT - 1; p >= 0; --p) {
    double s = 0.0;
#pragma unroll
    for (int c = p + 1; c < NT; ++c) s += c32_mv(F + (size_t)(p * NT + c) * 256, yv + 16 * c, lane);
    const double v = yv[16 * p + i] - s;
    ND_WAVE_LDS();
    if (lane < 16) vt[i] = v;
    ND_WAVE_LDS();
    const double x = c32_mv(F + (size_t)(NT * NT + p) * 256, vt, lane);
    ND_WAVE_LDS();
    if (lane < 16) yv[16 * p + i] = x;
    ND_WAVE_LDS();
  }
  for (int e = lane; e < n; e += 64) rhs[e] = yv[e];
}

// ---- the plan: leaves and separators from the column profile ----
static NdPlan* nd_get(lvx_ctx* c) { if (!c->nd) c->nd = new NdPlan; return (NdPlan*)c->nd; }
bool nd_active(const lvx_ctx* c) { return c->nd && ((const NdPlan*)c->nd)->active; }
int nd_ldz(const lvx_ctx* c) { return c->nd ? ((const NdPlan*)c->nd)->ldz : 0; }
int nd_nz(const lvx_ctx* c) { return c->nd ? ((const NdPlan*)c->nd)->nc - 64 : 0; }
void nd_counts(const lvx_ctx* c, int* separators, int* leaves) { const bool on = nd_active(c); *separators = on ? ((const NdPlan*)c->nd)->nsep : 0; *leaves = on ? ((const NdPlan*)c->nd)->nleaf : 0; }
void nd_destroy(lvx_ctx* c) {
  NdPlan* P = (NdPlan*)c->nd;
  if (!P) return;
  for (DevBuf* b : {&P->leaf, &P->sep, &P->nar, &P->den, &P->U, &P->WL, &P->WR, &P->GO, &P->Dc, &P->Rc, &P->LIc, &P->info, &P->D2, &P->G2, &P->F2, &P->Z2, &P->Y2, &P->info2, &P->zb2, &P->tc, &P->Fd})
    if (b->p) (void)hipFree(b->p);
  if (P->side) (void)hipStreamDestroy(P->side);
  if (P->ev_fork) (void)hipEventDestroy(P->ev_fork);
  if (P->ev_join) (void)hipEventDestroy(P->ev_join);
  if (P->ev_trsm) (void)hipEventDestroy(P->ev_trsm);
  if (P->ev_rc) (void)hipEventDestroy(P->ev_rc);
  delete P; c->nd = nullptr;
}
// SOLVER_ND: -1 never, 0 when the profile suits (a band of >= 8192 columns, at most half of them in wide runs), 1 whenever the profile allows it (tests on small problems).
// LVX_ND_LEAF: columns per narrow leaf (default: the smallest multiple of 16 >= 128 that keeps the separator chain at <= 511 blocks).
#define ND_NO(code) do { if (std::getenv("LVX_ND_DEBUG")) fprintf(stderr, "[lvx nd] no plan: reason %d (nb %d, bw %d, near %d, nrhs %d)\n", code, c->nb, c->bw, c->bw_near, nrhs); return LVX_OK; } while (0)
int nd_plan(lvx_ctx* c, int nrhs) {
  NdPlan* P = nd_get(c);
  const int mode = c->sw.solver_nd;
  if (P->epoch == c->layout_epoch && P->mode == mode && P->nrhs == nrhs) return LVX_OK;
  P->epoch = c->layout_epoch; P->mode = mode; P->nrhs = nrhs; P->active = false;
  const int nb = c->nb, W = c->bw_near + 1;
  if (mode < 0 || nb <= 0 || nrhs > 80 || W > ND_WS || (int)c->h_colhi.size() != nb) ND_NO(2);
  if (mode == 0 && nb < 8192) ND_NO(3);
  const std::vector<int>& hi = c->h_colhi; const std::vector<uint8_t>& full = c->h_colfull;
  std::vector<std::pair<int, int>> runs;   // wide runs [a, e)
  int nwide = 0;
  for (int j = 0; j < nb;) { if (!full[j]) { ++j; continue; } int e = j; while (e < nb && full[e]) ++e; runs.emplace_back(j, e); nwide += e - j; j = e; }
  if (mode == 0 && 2 * nwide > nb) ND_NO(4);
  struct El { int kind, c0, m; };   // 0 narrow leaf, 1 dense leaf, 2 separator
  auto build = [&](int M, std::vector<El>& els) -> bool {
    els.clear();
    auto stretch = [&](int a0, int e0, bool sf, bool sl) -> bool {   // narrow columns [a0, e0): [sep] leaf sep leaf ... leaf [sep]
      const int len = e0 - a0;
      if (len == 0) return true;
      if (sf && sl && len >= W && len <= ND_WS) { els.push_back({2, a0, len}); return true; }
      int nl = std::max(1, (len + (M + W) / 2) / (M + W));
      for (; nl >= 1; --nl) { const int ns = nl - 1 + (sf ? 1 : 0) + (sl ? 1 : 0); if (len - ns * W >= nl * W) break; }
      if (nl < 1) return false;
      const int ns = nl - 1 + (sf ? 1 : 0) + (sl ? 1 : 0), cols = len - ns * W, base = cols / nl, rem = cols % nl;
      int pos = a0;
      if (sf) { els.push_back({2, pos, W}); pos += W; }
      for (int i = 0; i < nl; ++i) {
        const int m = base + (i < rem ? 1 : 0);
        els.push_back({0, pos, m}); pos += m;
        if (i + 1 < nl || sl) { els.push_back({2, pos, W}); pos += W; }
      }
      return pos == e0;
    };
    int cur = 0;
    for (size_t r = 0; r <= runs.size(); ++r) {
      const int a0 = r < runs.size() ? runs[r].first : nb;
      if (a0 > cur || r == runs.size()) { if (a0 > cur && !stretch(cur, a0, r > 0, r < runs.size())) return false; }
      else if (r > 0) return false;   // (cannot happen: runs are maximal)
      if (r < runs.size()) { if (runs[r].second - runs[r].first > ND_DENSE_MAX) return false; els.push_back({1, runs[r].first, runs[r].second - runs[r].first}); cur = runs[r].second; }
    }
    return true;
  };
  std::vector<El> els;
  int M = 0;
  if (const char* e = std::getenv("LVX_ND_LEAF")) M = std::max(32, atoi(e) / 16 * 16);
  if (M > 0) { if (!build(M, els)) ND_NO(5); }
  else {
    bool ok = false;
    for (M = 128; M <= 4096; M += 16) {
      if (!build(M, els)) ND_NO(6);
      int ns = 0; for (const El& e : els) ns += e.kind == 2;
      if (ns <= 511) { ok = true; break; }
    }
    if (!ok) ND_NO(7);
  }
  // leaves / separators and their neighbours
  P->leaves.clear(); P->seps.clear();
  std::vector<int> nar, den;
  int toff = 0, maxnt = 0, bd = 16;
  for (size_t i = 0; i < els.size(); ++i) {
    const El& e = els[i];
    if (e.kind == 2) {
      NdSep s{e.c0, e.m, -1, -1};
      if (i > 0) { if (els[i - 1].kind == 2) ND_NO(8); s.lf = (int)P->leaves.size() - 1; }
      if (i + 1 < els.size()) { if (els[i + 1].kind == 2) ND_NO(9); s.rt = (int)P->leaves.size(); }
      P->seps.push_back(s);
    } else {
      NdLeaf l{};
      l.c0 = e.c0; l.m = e.m; l.nt = (e.m + 15) / 16; l.dense = e.kind;
      l.sl = (i > 0 && els[i - 1].kind == 2) ? (int)P->seps.size() - 1 : -1;
      l.sr = (i + 1 < els.size() && els[i + 1].kind == 2) ? (int)P->seps.size() : -1;
      if (i > 0 && els[i - 1].kind != 2) ND_NO(10);           // two leaves side by side: no separator between them
      l.cl = l.sl >= 0 ? els[i - 1].c0 : 0; l.wl = l.sl >= 0 ? els[i - 1].m : 0;
      l.cr = l.sr >= 0 ? els[i + 1].c0 : 0; l.wr = l.sr >= 0 ? els[i + 1].m : 0;
      l.pr0 = std::max(0, e.m - c->bw_near) / 16;
      if (e.kind == 0) { l.toff = toff; toff += l.nt; l.slot = (int)nar.size(); nar.push_back((int)P->leaves.size()); maxnt = std::max(maxnt, l.nt); if (l.nt - l.pr0 > 3) ND_NO(11); }
      else { l.toff = 0; l.slot = (int)den.size(); den.push_back((int)P->leaves.size()); bd = std::max(bd, (e.m + 15) / 16 * 16); }
      P->leaves.push_back(l);
    }
  }
  // the profile must agree: nothing left of a separator reaches past it, nothing up to its end reaches the next one; narrow leaves are narrow
  {
    std::vector<int> pm(nb);
    int run = -1;
    for (int j = 0; j < nb; ++j) { run = std::max(run, hi[j]); pm[j] = run; }
    for (size_t k = 0; k < P->seps.size(); ++k) {
      const NdSep& s = P->seps[k];
      if (s.w > ND_WS || s.w < 1) ND_NO(12);
      if (s.c0 > 0 && pm[s.c0 - 1] >= s.c0 + s.w) ND_NO(13);
      if (k + 1 < P->seps.size() && pm[s.c0 + s.w - 1] >= P->seps[k + 1].c0) ND_NO(14);
      for (int j = s.c0; j < s.c0 + s.w; ++j) if (full[j]) ND_NO(15);
    }
    for (const NdLeaf& l : P->leaves) {
      if (l.dense) { if (l.m > ND_DENSE_MAX) ND_NO(16); continue; }
      for (int j = l.c0; j < l.c0 + l.m; ++j) if (full[j] || hi[j] - j > c->bw_near) ND_NO(17);
    }
  }
  if (P->seps.empty()) ND_NO(18);
  const int nnar_ = (int)nar.size(), nden_ = (int)den.size();   // a single leaf: nothing to gain
  P->nleaf = (int)P->leaves.size(); P->nnar = nnar_; P->nden = nden_; P->nsep = (int)P->seps.size(); P->ntile = toff; P->bd = bd; P->maxnt = maxnt;
  P->zt = (nrhs + 15) / 16; if (P->zt < 2) P->zt = 2;
  P->nc = 64 + 16 * P->zt; P->ldz = (nb + 63) / 64 * 64;
  int nblk2 = 2; while (nblk2 < P->nsep) nblk2 <<= 1;
  P->nblk2 = nblk2;
  int rc;
  if ((rc = upload_tmp(c, P->leaf, P->leaves.data(), P->leaves.size() * sizeof(NdLeaf)))) return rc;
  if ((rc = upload_tmp(c, P->sep, P->seps.data(), P->seps.size() * sizeof(NdSep)))) return rc;
  if (nar.empty()) nar.push_back(0);
  if (den.empty()) den.push_back(0);
  if ((rc = upload_tmp(c, P->nar, nar.data(), nar.size() * 4))) return rc;
  if ((rc = upload_tmp(c, P->den, den.data(), den.size() * 4))) return rc;
  if ((rc = dev_alloc(c, P->U, (size_t)std::max(toff, 1) * 1024 * 8))) return rc;
  if ((rc = dev_alloc(c, P->WL, (size_t)std::max(toff, 1) * 16 * 32 * 8))) return rc;
  if ((rc = dev_alloc(c, P->WR, (size_t)std::max(P->nnar, 1) * 48 * 32 * 8))) return rc;
  if ((rc = dev_alloc(c, P->GO, (size_t)P->nleaf * 64 * P->nc * 8))) return rc;
  if ((rc = dev_alloc(c, P->Dc, (size_t)std::max(P->nden, 1) * bd * bd * 8))) return rc;
  if ((rc = dev_alloc(c, P->Rc, (size_t)std::max(P->nden, 1) * bd * P->nc * 8))) return rc;
  if ((rc = dev_alloc(c, P->LIc, (size_t)std::max(P->nden, 1) * (bd / 16) * 256 * 8))) return rc;
  if ((rc = dev_alloc(c, P->tc, (size_t)std::max(P->nden, 1) * bd * 8))) return rc;
  if ((rc = dev_alloc(c, P->info, (size_t)(P->nleaf + 8) * 4))) return rc;
  if ((rc = dev_alloc(c, P->D2, (size_t)nblk2 * 1024 * 8))) return rc;
  if ((rc = dev_alloc(c, P->G2, (size_t)2 * nblk2 * 1024 * 8))) return rc;
  const int nz = P->nc - 64;
  if ((rc = dev_alloc(c, P->F2, (size_t)nblk2 * 13 * 256 * 8))) return rc;
  if ((rc = dev_alloc(c, P->Z2, (size_t)nblk2 * 32 * nz * 8))) return rc;
  if ((rc = dev_alloc(c, P->Y2, (size_t)nblk2 * 32 * nz * 8))) return rc;
  if ((rc = dev_alloc(c, P->zb2, ((size_t)nblk2 * 32 + 64) * 8))) return rc;   // (+ 64 trash words: NdArgs::trash)
  if ((rc = dev_alloc(c, P->info2, (size_t)(2 * nblk2 + 8) * 4))) return rc;
  LVX_HIP(c, hipMemsetAsync(P->G2.p, 0, (size_t)2 * nblk2 * 1024 * 8, c->stream));
  LVX_HIP(c, hipMemsetAsync(P->Z2.p, 0, (size_t)nblk2 * 32 * nz * 8, c->stream));
  if (P->nsep < nblk2) {
    const size_t npad = (size_t)(nblk2 - P->nsep) * 1024;
    hipLaunchKernelGGL(k_bcr_pad_identity, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, c->stream, (double*)P->D2.p, 32, P->nsep, nblk2);
  }
  if (!P->side) {
    LVX_HIP(c, hipStreamCreateWithFlags(&P->side, hipStreamNonBlocking));   // (a high-priority stream for the few long dense-leaf workgroups made EVERY kernel of the solve slower: 1.15 -> 1.65 ms per step)
    LVX_HIP(c, hipEventCreateWithFlags(&P->ev_fork, hipEventDisableTiming));
    LVX_HIP(c, hipEventCreateWithFlags(&P->ev_join, hipEventDisableTiming));
    LVX_HIP(c, hipEventCreateWithFlags(&P->ev_trsm, hipEventDisableTiming));
    LVX_HIP(c, hipEventCreateWithFlags(&P->ev_rc, hipEventDisableTiming));
  }
  P->active = true;
  return LVX_OK;
}
static NdArgs nd_args(lvx_ctx* c, NdPlan* P, const double* scale, const double* lmd, double ir, double* Z, int ldz, const double* Bs = nullptr, const double* gbs = nullptr) {
  NdArgs a{};
  a.leaf = (const NdLeaf*)P->leaf.p; a.sep = (const NdSep*)P->sep.p; a.nar = (const int*)P->nar.p; a.den = (const int*)P->den.p;
  a.Hs = c->p_Hs ? c->p_Hs : (const double*)c->d_Hb.p; a.scale = scale; a.lmd = lmd; a.ir = ir; a.ld = c->bw + 1; a.npre = c->clear_npre; a.nb = c->nb;
  a.U = (double*)P->U.p; a.WL = (double*)P->WL.p; a.WR = (double*)P->WR.p; a.GO = (double*)P->GO.p; a.nc = P->nc; a.nrhs = P->nrhs;
  a.Bs = Bs; a.gbs = gbs; a.nbd = P->nrhs - 1;
  a.Z = Z; a.ldz = ldz; a.nz = P->nc - 64; a.info = (int*)P->info.p; a.trash = (double*)P->zb2.p + (size_t)P->nblk2 * 32;
  return a;
}
static C32 nd_c32(NdPlan* P) { return C32{(double*)P->D2.p, (double*)P->G2.p, (double*)P->Z2.p, (double*)P->Y2.p, (double*)P->F2.p, (int*)P->info2.p, P->nblk2, P->nc - 64, P->zt}; }
static size_t c32_lds_bytes() { return (size_t)(10 * 256 + C32_MAXT * 512 + 8 * 16 * 17) * 8; }

// The dense leaves, on the side stream, from their matrices to their products for the separators (nothing here waits for the narrow leaves; the right-hand sides are
// formed from the border rows where they are loaded: nd_rhs).  k_potrf_reg<12> needs whole CUs: k_nd_factor's four-leaf workgroups leave it more than half of them.
int nd_dense_start(lvx_ctx* c, const double* scale, const double* lmd, double inv_radius, const double* Bs, const double* gbs, double* Z, int ldz) {
  NdPlan* P = (NdPlan*)c->nd;
  if (!P || !P->active || ldz != P->ldz) return fail(c, LVX_E_STATE, "nd_dense_start without a matching plan");
  if (P->nden == 0) return LVX_OK;
  hipStream_t st = c->stream;
  NdArgs ad = nd_args(c, P, scale, lmd, inv_radius, Z, ldz, Bs, gbs);
  ad.info = ad.info + P->nnar;
  const int bd = P->bd, zt = P->zt;
  LVX_HIP(c, hipEventRecord(P->ev_fork, st));
  LVX_HIP(c, hipStreamWaitEvent(P->side, P->ev_fork, 0));
  // matrices on the side stream (the Cholesky, 60 us, starts behind them at once); right-hand sides on the caller's stream, ahead of the narrow leaves' factorisation
  // (which has 25 us to spare against the dense leaves' chain) — the triangular solve waits for both
  hipLaunchKernelGGL(k_nd_cbuild, dim3((unsigned)P->nden, (unsigned)((bd + 3) / 4)), dim3(256), 0, P->side, ad, (double*)P->Dc.p, (double*)P->Rc.p, bd, c->bw, 0);   // a wavefront per column
  hipLaunchKernelGGL(k_nd_cbuild, dim3((unsigned)P->nden, (unsigned)((P->nc + 3) / 4)), dim3(256), 0, st, ad, (double*)P->Dc.p, (double*)P->Rc.p, bd, c->bw, 1);
  LVX_HIP(c, hipEventRecord(P->ev_rc, st));
  c->stream = P->side;
  rocblas_handle h = nullptr;
  const long long sD = (long long)bd * bd, sLI = (long long)(bd / 16) * 256;
  int rc = potrf_batched(c, h, (double*)P->Dc.p, bd, sD, ad.info, P->nden, (double*)P->LIc.p, sLI);
  // (the streaming solve k_trsm_reg, which could share CUs with k_nd_solve, is slower still: 136 us beside it)
  if (!rc) LVX_HIP(c, hipStreamWaitEvent(P->side, P->ev_rc, 0));
  if (!rc) rc = trsv_batched<false>(c, (const double*)P->Dc.p, bd, sD, (double*)P->Rc.p, 1, bd, (long long)bd * P->nc, P->nc, P->nden, (const double*)P->LIc.p, sLI);
  if (!rc) LVX_HIP(c, hipEventRecord(P->ev_trsm, P->side));   // k_nd_solve starts behind it: the LDS-resident solve (130 KB per workgroup) cannot share a CU with its workgroups and waited for CUs to drain (102 us against 38)
  if (!rc) {
    if (zt == 2) hipLaunchKernelGGL(k_nd_cgram<2>, dim3((unsigned)P->nden), dim3(64 * 6), 0, P->side, ad, (const double*)P->Rc.p, bd);
    else if (zt == 3) hipLaunchKernelGGL(k_nd_cgram<3>, dim3((unsigned)P->nden), dim3(64 * 7), 0, P->side, ad, (const double*)P->Rc.p, bd);
    else if (zt == 4) hipLaunchKernelGGL(k_nd_cgram<4>, dim3((unsigned)P->nden), dim3(64 * 8), 0, P->side, ad, (const double*)P->Rc.p, bd);
    else hipLaunchKernelGGL(k_nd_cgram<5>, dim3((unsigned)P->nden), dim3(64 * 9), 0, P->side, ad, (const double*)P->Rc.p, bd);
  }
  c->stream = st;
  if (rc) return rc;
  LVX_HIP(c, hipEventRecord(P->ev_join, P->side));
  return LVX_OK;
}
// factor + Z <- L^-1 [B^T | -g_b] S (in the elimination order; Z is written, never read before); nd_dense_start has run
int nd_factor(lvx_ctx* c, const double* scale, const double* lmd, double inv_radius, int* info_out_d, const double* Bs, const double* gbs, double* Z, int ldz, int nrhs) {
  NdPlan* P = (NdPlan*)c->nd;
  if (!P || !P->active || nrhs != P->nrhs || ldz != P->ldz) return fail(c, LVX_E_STATE, "nd_factor without a matching plan");
  hipStream_t st = c->stream;
  NdArgs a = nd_args(c, P, scale, lmd, inv_radius, Z, ldz, Bs, gbs);
  const int zt = P->zt;
  if (P->nnar > 0) {
    hipLaunchKernelGGL(k_nd_factor, dim3((unsigned)((P->nnar + 3) / 4)), dim3(256), 0, st, a, P->nnar);
    if (P->nden > 0) LVX_HIP(c, hipStreamWaitEvent(st, P->ev_trsm, 0));
    if (zt == 2) hipLaunchKernelGGL(k_nd_solve<2>, dim3((unsigned)P->nnar), dim3(64 * 6), 0, st, a);
    else if (zt == 3) hipLaunchKernelGGL(k_nd_solve<3>, dim3((unsigned)P->nnar), dim3(64 * 7), 0, st, a);
    else if (zt == 4) hipLaunchKernelGGL(k_nd_solve<4>, dim3((unsigned)P->nnar), dim3(64 * 8), 0, st, a);
    else hipLaunchKernelGGL(k_nd_solve<5>, dim3((unsigned)P->nnar), dim3(64 * 9), 0, st, a);
  }
  if (P->nden > 0) LVX_HIP(c, hipStreamWaitEvent(st, P->ev_join, 0));
  const int nz = P->nc - 64;
  hipLaunchKernelGGL(k_nd_assemble, dim3((unsigned)P->nsep), dim3(256), 0, st, a, (double*)P->D2.p, (double*)P->G2.p, (double*)P->Z2.p, nz, P->nsep);
  hipLaunchKernelGGL(k_bcr_info, dim3((unsigned)((P->nleaf + 255) / 256)), dim3(256), 0, st, (const int*)P->info.p, P->nleaf, info_out_d);
  {   // the separator chain, one launch per level
    const C32 ch = nd_c32(P);
    LVX_HIP(c, hipFuncSetAttribute((const void*)k_c32_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c32_lds_bytes()));
    int L = 0; while ((1 << L) < P->nblk2) ++L;
    for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_c32_level, dim3((unsigned)(P->nblk2 >> (l + 1))), dim3(512), c32_lds_bytes(), st, ch, l, l == L - 1 ? 1 : 0);
    hipLaunchKernelGGL(k_bcr_info, dim3((unsigned)((P->nblk2 + 255) / 256)), dim3(256), 0, st, (const int*)P->info2.p, P->nblk2, info_out_d);
  }
  hipLaunchKernelGGL(k_nd_scatter, dim3((unsigned)P->nsep), dim3(256), 0, st, a, (const double*)P->Y2.p, nz);
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}
// zb <- L^-T zb (one vector, in place)
int nd_backward(lvx_ctx* c, double* zb) {
  NdPlan* P = (NdPlan*)c->nd;
  if (!P || !P->active) return fail(c, LVX_E_STATE, "nd_backward without a plan");
  hipStream_t st = c->stream;
  int rc;
  NdArgs a = nd_args(c, P, nullptr, nullptr, 0.0, nullptr, P->ldz);
  double* zb2 = (double*)P->zb2.p;
  hipLaunchKernelGGL(k_nd_gather1, dim3((unsigned)((P->nblk2 * 32 + 255) / 256)), dim3(256), 0, st, a.sep, P->nsep, P->nblk2, (const double*)zb, zb2);
  {
    const C32 ch = nd_c32(P);
    int L = 0; while ((1 << L) < P->nblk2) ++L;
    int l_lo = L - 1; while (l_lo > 0 && (P->nblk2 >> l_lo) <= 8) --l_lo;      // levels l_lo .. L - 1 have <= 8 eliminated blocks each
    hipLaunchKernelGGL(k_c32_back_top, dim3(1), dim3(512), 0, st, ch, L - 1, l_lo, zb2);
    for (int l = l_lo - 1; l >= 0; --l) hipLaunchKernelGGL(k_c32_back, dim3((unsigned)(P->nblk2 >> (l + 1))), dim3(64), 0, st, ch, l, 0, zb2);
  }
  if (P->nden > 0) {
    const int bd = P->bd;
    LVX_HIP(c, hipEventRecord(P->ev_fork, st));
    LVX_HIP(c, hipStreamWaitEvent(P->side, P->ev_fork, 0));
    hipLaunchKernelGGL(k_nd_cprep, dim3((unsigned)P->nden), dim3(256), 0, P->side, a, (const double*)P->Rc.p, bd, (const double*)zb2, (const double*)zb, (double*)P->tc.p);
    const size_t lds_fused = ((size_t)bd * (bd + 1) / 2 + 3 * (size_t)bd + 16 + 16 * 17) * 8;
    if (lds_fused > 160 * 1024) return fail(c, LVX_E_STATE, "dense leaf too wide for the fused backward kernel");
    LVX_HIP(c, hipFuncSetAttribute((const void*)k_bcr_back_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fused));
    hipLaunchKernelGGL(k_bcr_back_level, dim3((unsigned)P->nden), dim3(BACK_NT), lds_fused, P->side, (const double*)P->Dc.p, (long long)bd * bd, (const double*)P->LIc.p, (long long)(bd / 16) * 256,
                       (const double*)nullptr, 0ll, (double*)P->tc.p, 0ll, 0ll, (long long)bd, bd, P->nden, (long long)bd * bd, c->sw.deterministic);
    hipLaunchKernelGGL(k_nd_cscatter, dim3((unsigned)P->nden), dim3(256), 0, P->side, a, bd, (const double*)P->tc.p, zb);
    LVX_HIP(c, hipEventRecord(P->ev_join, P->side));
  }
  const size_t lds = (size_t)(80 + 16 * std::max(P->maxnt, 1)) * 8;
  if (lds > 64 * 1024) LVX_HIP(c, hipFuncSetAttribute((const void*)k_nd_back, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_nd_back, dim3((unsigned)(P->nnar + P->nsep)), dim3(64), lds, st, a, P->nnar, (const double*)zb2, zb);
  if (P->nden > 0) LVX_HIP(c, hipStreamWaitEvent(st, P->ev_join, 0));
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}

// dense border of a single sequence (n <= 64): see k_dense_tiles
bool dense_tiles_ok(const lvx_ctx* c, int n) { return c->ns == 0 && n >= 1 && n <= 64; }
int dense_tiles_factor(lvx_ctx* c, const double* S, double* rhs, int n, int* info) {
  NdPlan* P = nd_get(c);
  int rc = dev_alloc(c, P->Fd, (size_t)20 * 256 * 8); if (rc) return rc;
  double* F = (double*)P->Fd.p;
  const int nt = (n + 15) / 16;
  if (nt == 1) hipLaunchKernelGGL(k_dense_tiles<1>, dim3(1), dim3(64), 0, c->stream, S, rhs, n, info, F);
  else if (nt == 2) hipLaunchKernelGGL(k_dense_tiles<2>, dim3(1), dim3(64), 0, c->stream, S, rhs, n, info, F);
  else if (nt == 3) hipLaunchKernelGGL(k_dense_tiles<3>, dim3(1), dim3(64), 0, c->stream, S, rhs, n, info, F);
  else hipLaunchKernelGGL(k_dense_tiles<4>, dim3(1), dim3(64), 0, c->stream, S, rhs, n, info, F);
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}
int dense_tiles_back(lvx_ctx* c, double* rhs, int n) {
  NdPlan* P = nd_get(c);
  if (!P->Fd.p) return fail(c, LVX_E_STATE, "dense_tiles_back without a factorisation");
  const double* F = (const double*)P->Fd.p;
  const int nt = (n + 15) / 16;
  if (nt == 1) hipLaunchKernelGGL(k_dense_tiles_back<1>, dim3(1), dim3(64), 0, c->stream, F, rhs, n);
  else if (nt == 2) hipLaunchKernelGGL(k_dense_tiles_back<2>, dim3(1), dim3(64), 0, c->stream, F, rhs, n);
  else if (nt == 3) hipLaunchKernelGGL(k_dense_tiles_back<3>, dim3(1), dim3(64), 0, c->stream, F, rhs, n);
  else hipLaunchKernelGGL(k_dense_tiles_back<4>, dim3(1), dim3(64), 0, c->stream, F, rhs, n);
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}
