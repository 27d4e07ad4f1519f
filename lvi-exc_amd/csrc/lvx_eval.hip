// lvx_eval.hip — residual + analytic Jacobian + J^T J / J^T r assembly kernels (gfx950, FP64), the layout of a problem on the device (ensure_layout) and the pass
// that evaluates it (run_evaluate); the C ABI entry points over it are in lvx_api.hip.  The fused kernels (k_family_mfma, k_imu_own, the reprojection chain) are
// described where they stand and in DESIGN.md 3.1; k_family is the per-segment kernel (exact fallback, orientation prior, debug Jacobian): a workgroup is ONE wavefront,
// phase 1 lane = measurement (residual + local Jacobian rows in registers, Huber scaling, rows transposed into an LDS tile), phase 2 lane = column pair, one f64 atomic per
// entry and run of rows with equal control points.  The reference does all of this with one DynamicAutoDiffCostFunction::Evaluate per block on CPU threads and Ceres'
// block-sparse J^T J (kontiki/trajectory_estimator.h:38-68).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <numeric>
#include <cstdlib>

#include <thread>
#include "lvx_ctx.h"

namespace lvx {

#define LVX_PW 4   // wavefronts per workgroup of k_family
struct Keys { int k0, k1, lm; };
LVX_HD bool same(const Keys& a, const Keys& b) { return a.k0 == b.k0 && a.k1 == b.k1 && a.lm == b.lm; }

struct Cal { ImuCal imu; SensorCal lidar, cam; const double* rho; };
__device__ __forceinline__ Cal load_cal(const DevCommon& cm) {
  const double* s = cm.state + 7 * (size_t)cm.N;
  Cal c;
  c.imu.tau = s[7]; c.imu.roll = s[8]; c.imu.pitch = s[9]; c.imu.ba = load_v3(s + 10); c.imu.bg = load_v3(s + 13);
  c.lidar.q = load_q(s + 16); c.lidar.p = load_v3(s + 20); c.lidar.tau = s[23];
  c.cam.q = load_q(s + 24); c.cam.p = load_v3(s + 28); c.cam.tau = s[31];
  c.rho = s + 32;
  return c;
}

struct HubShared { PoseEval A; int ok; double M[6][24]; };   // M: hub_matrix(A), filled by the pass's prepass for the fold kernel

// ---------------------------------------------------------------------------------------------------------
// family policies
// ---------------------------------------------------------------------------------------------------------
struct GyroFam {
  enum { NC = GYRO_NC, NR = GYRO_NR, USES_HUB = 0 };
  int n; const double* t; const double* m3; const int* perm; double weight, huber;
  __device__ void make_hub(const DevCommon&, const SplineRef&, const Cal&, HubShared*) const {}
  __device__ int eval(const DevCommon& cm, const SplineRef& sp, const Cal& cal, const HubShared&, int si, double r[NR], double (*J)[NC], Keys& k) const {
    k.k1 = 0; k.lm = 0;
    return gyro_residual<true>(sp, cal.imu, t[si], load_v3(m3 + 3 * (size_t)si), weight, &k.k0, r, J);
  }
  __device__ int col(int c, const Keys& k, int N) const { return gyro_col(c, k.k0, N); }
};
struct AccelFam {
  enum { NC = ACC_NC, NR = ACC_NR, USES_HUB = 0 };
  int n; const double* t; const double* m3; const int* perm; double weight, huber;
  __device__ void make_hub(const DevCommon&, const SplineRef&, const Cal&, HubShared*) const {}
  __device__ int eval(const DevCommon& cm, const SplineRef& sp, const Cal& cal, const HubShared&, int si, double r[NR], double (*J)[NC], Keys& k) const {
    k.k1 = 0; k.lm = 0;
    return accel_residual<true>(sp, cal.imu, t[si], load_v3(m3 + 3 * (size_t)si), weight, &k.k0, r, J);
  }
  __device__ int col(int c, const Keys& k, int N) const { return acc_col(c, k.k0, N); }
};
struct PriorFam {
  enum { NC = PRI_NC, NR = PRI_NR, USES_HUB = 0 };
  int n; double t; quat q; const int* perm; double weight, huber;
  __device__ void make_hub(const DevCommon&, const SplineRef&, const Cal&, HubShared*) const {}
  __device__ int eval(const DevCommon& cm, const SplineRef& sp, const Cal&, const HubShared&, int, double r[NR], double (*J)[NC], Keys& k) const {
    k.k1 = 0; k.lm = 0;
    return prior_residual<true>(sp, t, q, weight, &k.k0, r, J);
  }
  __device__ int col(int c, const Keys& k, int N) const { return pri_col(c, k.k0, N); }
};

LVX_HD void hub_spans(double t_map, bool tau_locked, double mto, double sp1[1][2]) {
  if (tau_locked) { sp1[0][0] = t_map; sp1[0][1] = t_map; } else { sp1[0][0] = t_map - mto; sp1[0][1] = t_map + mto; }
}

template <bool TAU> struct SurfFamT {
  enum { NC = SURF_NC + (TAU ? 1 : 0), NR = SURF_NR, USES_HUB = 1 };
  int n; const double* t; const double* pt; const int* plane; const int* perm; const double* planes; double t_map, weight, huber;
  __device__ void make_hub(const DevCommon& cm, const SplineRef& sp, const Cal& cal, HubShared* h) const {
    double s1[1][2]; hub_spans(t_map, (cm.locks & LVX_LOCK_LIDAR_TAU) != 0, cm.sensor_mto, s1);
    Segs sg; KnotRef kh; h->ok = 0;
    if (!build_segments(sp, s1, 1, &sg)) return;
    if (!seg_lookup(sp, sg, t_map + cal.lidar.tau, &kh)) return;
    if (!pose_eval<true, TAU>(sp, kh, &h->A)) { h->ok = -RES_NONUNIT; return; }
    h->ok = 1;
  }
  __device__ int eval(const DevCommon& cm, const SplineRef& sp, const Cal& cal, const HubShared& hub, int si, double r[NR], double (*J)[NC], Keys& k) const {
    const bool tl = (cm.locks & LVX_LOCK_LIDAR_TAU) != 0;
    const double tk = t[si];
    const double pad = tl ? 0.0 : cm.sensor_mto;
    const double spans[2][2] = {{t_map - pad, t_map + pad}, {tk - pad, tk + pad}};
    Segs segs;
    if (!build_segments(sp, spans, 2, &segs)) return RES_RANGE;
    KnotRef kh;
    if (!seg_lookup(sp, segs, t_map + cal.lidar.tau, &kh)) return RES_RANGE;
    const PoseEval* hp = &hub.A;
    PoseEval own;
    if (hub.ok != 1 || kh.i0 != hub.A.k.i0 || kh.u != hub.A.k.u) {   // merged-segment corner (spline_base.h:196-203)
      if (!pose_eval<true, TAU>(sp, kh, &own)) return RES_NONUNIT;
      hp = &own;
    }
    k.k0 = kh.i0; k.lm = 0;
    const int pid = plane[si];
    return surfel_residual<true, TAU>(sp, *hp, segs, cal.lidar, tk, load_v3(pt + 3 * (size_t)si), load_v3(planes + 3 * (size_t)pid), weight, &k.k1, r, J);
  }
  __device__ int col(int c, const Keys& k, int N) const { return surf_col(c, k.k0, k.k1, N); }
};
using SurfFam = SurfFamT<false>;
template <bool TAU> struct CamSurfFamT {
  enum { NC = CS_NC + (TAU ? 1 : 0), NR = CS_NR, USES_HUB = 1 };
  int n; const int* lm; const int* plane; const int* perm; const double* planes; const double* lm_uv; const double* lm_t0; double t_map, weight, huber;
  __device__ void make_hub(const DevCommon& cm, const SplineRef& sp, const Cal& cal, HubShared* h) const {
    double s1[1][2]; hub_spans(t_map, (cm.locks & LVX_LOCK_CAM_TAU) != 0, cm.sensor_mto, s1);
    Segs sg; KnotRef kh; h->ok = 0;
    if (!build_segments(sp, s1, 1, &sg)) return;
    if (!seg_lookup(sp, sg, t_map + cal.cam.tau, &kh)) return;
    if (!pose_eval<true, TAU>(sp, kh, &h->A)) { h->ok = -RES_NONUNIT; return; }
    h->ok = 1;
  }
  __device__ int eval(const DevCommon& cm, const SplineRef& sp, const Cal& cal, const HubShared& hub, int si, double r[NR], double (*J)[NC], Keys& k) const {
    const bool tl = (cm.locks & LVX_LOCK_CAM_TAU) != 0;
    const int l = lm[si];
    const double tk = lm_t0[l];
    const double pad = tl ? 0.0 : cm.sensor_mto;
    const double spans[2][2] = {{t_map - pad, t_map + pad}, {tk - pad, tk + pad}};
    Segs segs;
    if (!build_segments(sp, spans, 2, &segs)) return RES_RANGE;
    KnotRef kh;
    if (!seg_lookup(sp, segs, t_map + cal.cam.tau, &kh)) return RES_RANGE;
    const PoseEval* hp = &hub.A;
    PoseEval own;
    if (hub.ok != 1 || kh.i0 != hub.A.k.i0 || kh.u != hub.A.k.u) {
      if (!pose_eval<true, TAU>(sp, kh, &own)) return RES_NONUNIT;
      hp = &own;
    }
    k.k0 = kh.i0; k.lm = 0;
    return camsurf_residual<true, TAU>(sp, *hp, segs, cm.cam, cal.cam, cal.lidar, lm_uv[2 * l], lm_uv[2 * l + 1], tk, cal.rho[l],
                                  load_v3(planes + 3 * (size_t)plane[si]), weight, &k.k1, r, J);
  }
  __device__ int col(int c, const Keys& k, int N) const { return cs_col(c, k.k0, k.k1, N); }
};
using CamSurfFam = CamSurfFamT<false>;
template <bool TAU> struct ReprojFamT {
  enum { NC = REP_NC + (TAU ? 1 : 0), NR = REP_NR, USES_HUB = 0 };
  int n; const int* lm; const double* uv; const double* t0o; const int* perm; const double* lm_uv; const double* lm_t0; double weight, huber;
  __device__ void make_hub(const DevCommon&, const SplineRef&, const Cal&, HubShared*) const {}
  __device__ int eval(const DevCommon& cm, const SplineRef& sp, const Cal& cal, const HubShared&, int si, double r[NR], double (*J)[NC], Keys& k) const {
    const int l = lm[si];
    k.lm = l;
    return reproj_residual<true, TAU>(sp, cm.cam, cal.cam, (cm.locks & LVX_LOCK_CAM_TAU) != 0, cm.sensor_mto, lm_uv[2 * l], lm_uv[2 * l + 1], lm_t0[l],
                                 uv[2 * (size_t)si], uv[2 * (size_t)si + 1], t0o[si], cal.rho[l], weight, &k.k0, &k.k1, r, J);
  }
  // fused path only (k_reproj_jac): SO3 parts from the pass's control-point-pair table
  __device__ __forceinline__ int eval_pre(const DevCommon& cm, const SplineRef& sp, const Cal& cal, int si, double r[NR], double (*J)[NC], Keys& k) const {
    const int l = lm[si];
    k.lm = l;
    return reproj_residual<true, TAU>(sp, cm.cam, cal.cam, (cm.locks & LVX_LOCK_CAM_TAU) != 0, cm.sensor_mto, lm_uv[2 * l], lm_uv[2 * l + 1], lm_t0[l],
                                 uv[2 * (size_t)si], uv[2 * (size_t)si + 1], t0o[si], cal.rho[l], weight, &k.k0, &k.k1, r, J, cm.pre);
  }
  __device__ int col(int c, const Keys& k, int N) const { return rep_col(c, k.k0, k.k1, N, k.lm); }
};
using ReprojFam = ReprojFamT<false>;

// ---------------------------------------------------------------------------------------------------------
// routing of one normal-equation entry into the structured storage
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void add_H(const DevCommon& cm, int pa, int pb, double v, int rep) {
  if (pa >= LVX_LM_BASE || pb >= LVX_LM_BASE) {   // an entry of a landmark's row
    const int l = (pa >= LVX_LM_BASE ? pa : pb) - LVX_LM_BASE, o = pa >= LVX_LM_BASE ? pb : pa;
    double* row = cm.lmH + (size_t)l * cm.lm_ls;
    if (o >= LVX_LM_BASE) atomicAdd(&row[cm.lm_wl + cm.nbd], v);
    else if (o >= 0) { const int k = o - cm.lm_p0[l]; if (k < 0 || k >= cm.lm_wl) { atomicOr(cm.err, 4); return; } atomicAdd(&row[k], v); }
    else atomicAdd(&row[cm.lm_wl + (-1 - o)], v);
    return;
  }
  if (pa >= 0 && pb >= 0) {
    const int i = pa > pb ? pa : pb, j = pa > pb ? pb : pa, d = i - j;
    if (d > cm.bw) { atomicOr(cm.err, 4); return; }
    atomicAdd(&cm.Hb[(size_t)j * (cm.bw + 1) + d], v);
  } else if (pa < 0 && pb < 0) {
    const int ba = -1 - pa, bb = -1 - pb;
    const int i = ba > bb ? ba : bb, j = ba > bb ? bb : ba;
    atomicAdd(&cm.C[(size_t)rep * cm.nbd * cm.nbd + (size_t)i * cm.nbd + j], v);
  } else {
    const int b = pa < 0 ? -1 - pa : -1 - pb, p = pa < 0 ? pb : pa;
    atomicAdd(&cm.Bd[(size_t)b * cm.nb + p], v);
  }
}
__device__ __forceinline__ void add_g(const DevCommon& cm, int pc, double v, int rep) {
  if (pc >= LVX_LM_BASE) atomicAdd(&cm.lmH[(size_t)(pc - LVX_LM_BASE) * cm.lm_ls + cm.lm_wl + cm.nbd + 1], v);
  else if (pc >= 0) atomicAdd(&cm.gb[pc], v);
  else atomicAdd(&cm.gc[(size_t)rep * cm.nbd + (-1 - pc)], v);
}

// a row a fused kernel cannot take exactly (control-point pair beyond the small-angle polynomials, merged map-time segment corner, interval outside the chunk's window):
// onto the family's fallback list — the exact per-segment kernel evaluates the listed rows right behind the fused kernel — or, without lists / beyond their capacity,
// the whole pass is redone by the per-segment kernels
#define LVX_FB_CAP 4096
__device__ __forceinline__ void fallback_row(const DevCommon& cm, int fam, int si) {
  const int i = atomicAdd(&cm.err[4 + fam], 1);   // counted either way: the host learns which families need a list
  if (cm.fb_list && ((cm.fb_cap >> 16) >> fam & 1) && i < (cm.fb_cap & 0xffff)) { cm.fb_list[(size_t)fam * (cm.fb_cap & 0xffff) + i] = si; return; }
  atomicOr(cm.err, LVX_ERR_FALLBACK);
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// PW = wavefronts per workgroup: wave 0 evaluates the 64 measurements (phase 1), all PW waves share the pair work of phase 2
template <class F, int PW>
__global__ __launch_bounds__(64 * PW) void k_family(F fam, DevCommon cm, const uint16_t* __restrict__ pairs, long long row0, const int* __restrict__ rows = nullptr, const int* __restrict__ nrows = nullptr) {
  constexpr int NC = F::NC, NR = F::NR, TS = NR * 64 + 1, NP = NC * (NC + 1) / 2;
  __shared__ double Jt[NC * TS];
  __shared__ double rs[NR * 64];
  __shared__ int cpos[64 * NC];
  __shared__ int seg_l0[64], seg_ok[64];
  __shared__ Keys seg_keys[64];
  __shared__ HubShared hub;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int rep = blockIdx.x % cm.nrep;
  __shared__ int nseg_s;
  const int nlist = rows ? min(*nrows, cm.fb_cap & 0xffff) : 0;   // rows != null: the rows of a fallback list instead of all of them
  if (rows && (int)blockIdx.x * 64 >= nlist) return;
  if (wave == 0) {
  const int idx = blockIdx.x * 64 + lane;
  const bool in = rows ? idx < nlist : idx < fam.n;
  const int si = rows ? (in ? rows[idx] : 0) : idx;
  const SplineRef sp{cm.t0, cm.dt, cm.N, cm.state, cm.state + 3 * (size_t)cm.N};
  const Cal cal = load_cal(cm);
  if (F::USES_HUB) {
    if (lane == 0) fam.make_hub(cm, sp, cal, &hub);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  double r[NR];
  double J[NR][NC];
  Keys key{-1, -1, -1};
  bool valid = false;
  if (in) {
    const int status = fam.eval(cm, sp, cal, hub, si, r, J, key);
    valid = status == RES_OK;
    if (!valid) { atomicOr(cm.err, status); key = Keys{-1, -1, -1}; }
  }
  double mycost = 0.0;
  if (valid) {
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < NR; ++a) s += r[a] * r[a];
    double scale;
    mycost = 0.5 * huber_rho(fam.huber, s, &scale);
    const long long orow = row0 + (long long)fam.perm[si] * NR;
    if (cm.residuals) {
#pragma unroll
      for (int a = 0; a < NR; ++a) cm.residuals[orow + a] = r[a];
    }
    if (cm.jcols) {
#pragma unroll
      for (int a = 0; a < NR; ++a) {
        int32_t* jc = cm.jcols + (orow + a) * LVX_JAC_WIDTH;
        double* jv = cm.jvals + (orow + a) * LVX_JAC_WIDTH;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int g = fam.col(c, key, cm.N);
          const bool dead = tangent_locked(g, cm.N, cm.L, cm.locks);
          jc[c] = dead ? -1 : g;
          jv[c] = dead ? 0.0 : J[a][c];
        }
        for (int c = NC; c < LVX_JAC_WIDTH; ++c) { jc[c] = -1; jv[c] = 0.0; }
      }
    }
    if (scale != 1.0) {
#pragma unroll
      for (int a = 0; a < NR; ++a) {
        r[a] *= scale;
#pragma unroll
        for (int c = 0; c < NC; ++c) J[a][c] *= scale;
      }
    }
  }
  mycost = wave_sum(mycost);
  if (lane == 0) atomicAdd(&cm.cost[rep], mycost);
  if (cm.what & LVX_EVAL_NORMAL_EQ) {
#pragma unroll
    for (int a = 0; a < NR; ++a) {
      rs[lane * NR + a] = valid ? r[a] : 0.0;
#pragma unroll
      for (int c = 0; c < NC; ++c) Jt[c * TS + lane * NR + a] = valid ? J[a][c] : 0.0;
    }
    // segment heads: first lane of every run of equal keys
    Keys pk;
    pk.k0 = __shfl_up(key.k0, 1); pk.k1 = __shfl_up(key.k1, 1); pk.lm = __shfl_up(key.lm, 1);
    const bool head = (lane == 0) || !same(pk, key);
    const unsigned long long heads = __ballot(head);
    const int myseg = __popcll(heads & ((2ull << lane) - 1ull)) - 1;
    if (head) { seg_l0[myseg] = lane; seg_ok[myseg] = valid ? 1 : 0; seg_keys[myseg] = key; }
    if (lane == 0) nseg_s = __popcll(heads);
  }
  }   // wave 0
  if (!(cm.what & LVX_EVAL_NORMAL_EQ)) return;
  __syncthreads();
  const int nseg = nseg_s;
  const int tid = threadIdx.x;
  constexpr int NT = 64 * PW;
  // column positions of every segment, computed once (no barriers inside the segment loop below)
  for (int e = tid; e < nseg * NC; e += NT) { const int sg = e / NC, c = e % NC; cpos[e] = seg_ok[sg] ? cm.ord[fam.col(c, seg_keys[sg], cm.N)] : LVX_DEAD; }
  // this thread's pair codes, reused for every segment
  constexpr int ITER = (NP + NC + NT - 1) / NT;
  unsigned code[ITER];
#pragma unroll
  for (int i = 0; i < ITER; ++i) { const int p = tid + NT * i; code[i] = p < NP ? pairs[p] : 0u; }
  __syncthreads();
  for (int sg = 0; sg < nseg; ++sg) {
    if (!seg_ok[sg]) continue;
    const int l0 = seg_l0[sg];
    const int l1 = sg + 1 < nseg ? seg_l0[sg + 1] : 64;
    const int rb = l0 * NR, re = l1 * NR;
    const int* cp = &cpos[sg * NC];
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int p = tid + NT * i;
      if (p < NP) {
        const int a = code[i] & 0xff, b = code[i] >> 8;
        const int pa = cp[a], pb = cp[b];
        if (pa == LVX_DEAD || pb == LVX_DEAD) continue;
        const double* ja = &Jt[a * TS];
        const double* jb = &Jt[b * TS];
        double acc = 0.0;
        for (int row = rb; row < re; ++row) acc += ja[row] * jb[row];
        if (a != b && pa == pb) acc *= 2.0;
        add_H(cm, pa, pb, acc, rep);
      } else if (p < NP + NC) {
        const int c = p - NP;
        const int pc = cp[c];
        if (pc == LVX_DEAD) continue;
        const double* jc = &Jt[c * TS];
        double acc = 0.0;
        for (int row = rb; row < re; ++row) acc += jc[row] * rs[row];
        add_g(cm, pc, acc, rep);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Fast assembly path (k_family_mfma below).  A workgroup OWNS a range of LVX_CHUNK_R knot intervals: its 4 wavefronts evaluate the
// range's measurements in batches of 256 and accumulate J^T J into workgroup-shared LDS accumulators (band window, border rows,
// dense border).  One flush of the accumulators to HBM per workgroup replaces the per-segment global atomics of k_family.
// The pose at t_map, common to every surfel / cam-surfel residual, enters through 6 pseudo variables (d p_0, xi_0):
// J_hub = g0^T M_hub, folded back onto the hub control points by k_fold_border.
// ---------------------------------------------------------------------------------------------------------
// u-independent SO3 quantities of every control-point pair (k, k+1) — log, |Omega|, J_r^-1 — once per pass instead of once per workgroup
// (the fused kernels copy their CR + 4 entries into LDS) and once per row (reprojection Jacobian kernel): k_state_prepass below
__device__ void hub_eval_thread(const DevCommon& cm, double t_map, int want_surf, int want_cs, HubShared* hubs, int s) {
  if (s > 1) return;
  if ((s == 0 && !want_surf) || (s == 1 && !want_cs)) { hubs[s].ok = 0; return; }
  const SplineRef sp{cm.t0, cm.dt, cm.N, cm.state, cm.state + 3 * (size_t)cm.N};
  const Cal cal = load_cal(cm);
  const bool tl = s == 0 ? (cm.locks & LVX_LOCK_LIDAR_TAU) != 0 : (cm.locks & LVX_LOCK_CAM_TAU) != 0;
  const double tau = s == 0 ? cal.lidar.tau : cal.cam.tau;
  double s1[1][2]; hub_spans(t_map, tl, cm.sensor_mto, s1);
  Segs sg; KnotRef kh; hubs[s].ok = 0;
  if (!build_segments(sp, s1, 1, &sg)) return;
  if (!seg_lookup(sp, sg, t_map + tau, &kh)) return;
  if (!(tl ? pose_eval<true>(sp, kh, &hubs[s].A) : pose_eval<true, true>(sp, kh, &hubs[s].A))) { hubs[s].ok = -RES_NONUNIT; return; }   // a free offset differentiates through the hub's velocity / angular velocity
  hub_matrix(hubs[s].A, hubs[s].M);   // once here instead of by one thread of each of the fold's ~600 workgroups
  hubs[s].ok = 1;
}
// everything that depends on the state only: blocks [0, nblk_tab) fill the control-point-pair table, the next block evaluates the shared
// t_map poses; run by the first blocks of the pass's clear kernel (k_clear)
__device__ __forceinline__ void state_prepass_block(const DevCommon& cm, So3Pre* tab, int nblk_tab, double t_map, int want_surf, int want_cs, HubShared* hubs, int blk) {
  if (blk >= nblk_tab) { hub_eval_thread(cm, t_map, want_surf, want_cs, hubs, threadIdx.x); return; }
  const int k = blk * blockDim.x + threadIdx.x;
  if (k >= cm.N) return;
  const double* so3 = cm.state + 3 * (size_t)cm.N;
  So3Pre e;
  if (k + 1 < cm.N) so3_pre(load_q(so3 + 4 * (size_t)k), load_q(so3 + 4 * (size_t)(k + 1)), &e);
  else { e.Om = mk(0, 0, 0); e.on = 0.0; e.Jri = m3_identity(); e.c3 = 1.0 / 12.0; e.ok = 1; }
  tab[k] = e;
}

// per-row extras of the MFMA path: window id (rows with wid in [w, w + WS) share a window; default = knot interval), and for the
// reprojection families the knot interval of the OTHER pose and the landmark
struct Aux { int wid, xk, lm; const PreWin* pw; };   // pw: the workgroup's precomputed control-point-pair table (lvx_math.h: So3Pre)
// traits: NK knot columns (4 knots x KPK, at offset LVO of the knot's 6 tangent scalars) | NG global columns; WS knot intervals per MFMA window; GL lanes per panel;
// SKIP_GG: global x global and the global gradient are assembled by another pass; SECONDARY: no cost / residual output
struct GyroAcc {
  enum { PMAJ = 1 };
  enum { NK = 12, NG = 3, NR = 3, HUB = -1, KPK = 3, LVO = 3, WS = 1, GL = 32, SKIP_GG = 0, SECONDARY = 0, LB = 64, NCP = 15, USE_PRE = 1, OCC = 1, FAM = LVX_FAM_GYRO };
  __device__ static constexpr int jm(int c) { return c; }   // KPK columns per knot at offset LVO of its 6 tangent scalars; WS = knot intervals per MFMA window; GL = lanes per panel
  int n; const double* t; const double* m3; const int* perm; double weight, huber;
  __device__ int eval(const DevCommon& cm, const SplineRef& sp, const Cal& cal, const HubShared*, int si, double r[NR], double (*J)[NCP], int& key, Aux& aux) const {
    return gyro_residual<true, true>(sp, cal.imu, t[si], load_v3(m3 + 3 * (size_t)si), weight, &key, r, J, aux.pw);
  }
  __device__ static int klv(int c) { return 6 * (c / 3) + 3 + c % 3; }
  __device__ static int gcol(int g, int N, int nt) { return 6 * N + 5 + g; }
};
// TAU (free LiDAR time offset, the reference's opt_time_offset_ stages: trajectory_manager_lvi.cpp:159-165, sensors.h:70-85): one more global column, d r / d tau_L =
// g_p (v_k - v_0) + g_xi0 w_0 + g_xik w_k — the pose gradients the row already has, contracted with the spline's velocity and body angular velocity at both poses —
// and padded spans through the generic segment branch.  The locked instantiation is unchanged.
template <bool TAU> struct SurfAccT {
  enum { NK = 24, NG = 12 + (TAU ? 1 : 0), NR = 1, HUB = 0, KPK = 6, LVO = 0, WS = 2, GL = 16, SKIP_GG = 0, SECONDARY = 0, LB = 64, NCP = 36 + (TAU ? 1 : 0), USE_PRE = 1, OCC = 2, FAM = LVX_FAM_SURFEL };   // reverse-mode Jacobian: fits two wavefronts per SIMD with a few spills, and two workgroups per CU hide its latencies
  __device__ static constexpr int jm(int c) { return c; }
  int n; const double* t; const double* pt; const double* rowpl; const int* perm; double t_map, weight, huber;   // rowpl: the row's plane (gathered at layout time: no dependent load)
  // raw inputs of a row, loaded one batch ahead of their use: the HBM latency hides behind the previous batch's assembly
  struct Row { double t; v3 p, Pi; };
  enum { PREFETCH = 1 };
  __device__ Row load(int si) const { return Row{t[si], load_v3(pt + 3 * (size_t)si), load_v3(rowpl + 3 * (size_t)si)}; }
  __device__ int eval_row(const DevCommon& cm, const SplineRef& sp, const Cal& cal, const HubShared* hub, const Row& row, double r[NR], double (*J)[NCP], int& key, Aux& aux) const {
    const bool tl = (cm.locks & LVX_LOCK_LIDAR_TAU) != 0;
    const double tk = row.t;
    Segs segs;
    if (tl) {   // locked offset: two point spans — the segment bookkeeping reduces to two interval indices unless the segments merge
      KnotRef kr;
      const int st = two_point_lookup(sp, t_map, tk, tk + cal.lidar.tau, &kr);
      if (st >= 0) {
        if (st == 1) return RES_RANGE;
        if (hub->ok != 1) return hub->ok < 0 ? RES_NONUNIT : RES_RANGE;   // the hub's own lookup in its segment is the generic path's first lookup
        if (st == 2) return RES_RANGE;
        LVX_KT(aux.pw, 8)
        return surfel_residual_pseudo<true, TAU>(sp, hub->A, segs, cal.lidar, tk, row.p, row.Pi, weight, &key, r, J, aux.pw, &kr);
      }
    }
    const double pad = tl ? 0.0 : cm.sensor_mto;
    const double spans[2][2] = {{t_map - pad, t_map + pad}, {tk - pad, tk + pad}};
    if (!build_segments(sp, spans, 2, &segs)) return RES_RANGE;
    KnotRef kh;
    if (!seg_lookup(sp, segs, t_map + cal.lidar.tau, &kh)) return RES_RANGE;
    if (hub->ok != 1) return hub->ok < 0 ? RES_NONUNIT : RES_RANGE;
    if (kh.i0 != hub->A.k.i0 || kh.u != hub->A.k.u) return LVX_ERR_FALLBACK;   // merged-segment corner: only the legacy kernel is exact
    LVX_KT(aux.pw, 8)
    return surfel_residual_pseudo<true, TAU>(sp, hub->A, segs, cal.lidar, tk, row.p, row.Pi, weight, &key, r, J, aux.pw);
  }
  __device__ static int klv(int c) { return c; }
  __device__ static int gcol(int g, int N, int nt) { return g < 6 ? nt + g : 6 * N + 8 + (g - 6); }   // g = 12 (TAU): 6 N + 14, the LiDAR time offset
};
using SurfAcc = SurfAccT<false>;
template <bool TAU> struct CamSurfAccT {
  enum { NK = 24, NG = 18 + (TAU ? 1 : 0), NR = 1, HUB = 1, KPK = 6, LVO = 0, WS = 1, GL = 16, SKIP_GG = 0, SECONDARY = 0, LB = 64, NCP = 42 + (TAU ? 1 : 0), USE_PRE = 1, OCC = 2, FAM = LVX_FAM_CAMSURF };
  __device__ static constexpr int jm(int c) { return c; }
  int n; const int* lm; const int* plane; const int* perm; const double* planes; const double* lm_uv; const double* lm_t0; double t_map, weight, huber;
  __device__ int eval(const DevCommon& cm, const SplineRef& sp, const Cal& cal, const HubShared* hub, int si, double r[NR], double (*J)[NCP], int& key, Aux& aux) const {
    const bool tl = (cm.locks & LVX_LOCK_CAM_TAU) != 0;
    const int l = lm[si];
    const double tk = lm_t0[l];
    Segs segs;
    if (tl) {   // locked offset: two point spans (see SurfAcc::eval_row)
      KnotRef kr;
      const int st = two_point_lookup(sp, t_map, tk, tk + cal.cam.tau, &kr);
      if (st >= 0) {
        if (st == 1) return RES_RANGE;
        if (hub->ok != 1) return hub->ok < 0 ? RES_NONUNIT : RES_RANGE;
        if (st == 2) return RES_RANGE;
        return camsurf_residual_pseudo<true, TAU>(sp, hub->A, segs, cm.cam, cal.cam, cal.lidar, lm_uv[2 * l], lm_uv[2 * l + 1], tk, cal.rho[l],
                                                  load_v3(planes + 3 * (size_t)plane[si]), weight, &key, r, J, aux.pw, &kr);
      }
    }
    const double pad = tl ? 0.0 : cm.sensor_mto;
    const double spans[2][2] = {{t_map - pad, t_map + pad}, {tk - pad, tk + pad}};
    if (!build_segments(sp, spans, 2, &segs)) return RES_RANGE;
    KnotRef kh;
    if (!seg_lookup(sp, segs, t_map + cal.cam.tau, &kh)) return RES_RANGE;
    if (hub->ok != 1) return hub->ok < 0 ? RES_NONUNIT : RES_RANGE;
    if (kh.i0 != hub->A.k.i0 || kh.u != hub->A.k.u) return LVX_ERR_FALLBACK;
    return camsurf_residual_pseudo<true, TAU>(sp, hub->A, segs, cm.cam, cal.cam, cal.lidar, lm_uv[2 * l], lm_uv[2 * l + 1], tk, cal.rho[l],
                                              load_v3(planes + 3 * (size_t)plane[si]), weight, &key, r, J, aux.pw);
  }
  __device__ static int klv(int c) { return c; }
  __device__ static int gcol(int g, int N, int nt) { return g < 6 ? nt + 6 + g : (g < 12 ? 6 * N + 15 + (g - 6) : (g < 18 ? 6 * N + 8 + (g - 12) : 6 * N + 21)); }   // g = 18 (TAU): the camera time offset
};
using CamSurfAcc = CamSurfAccT<false>;

// Rolling-shutter reprojection on the fast path.  k_reproj_jac evaluates residual + Jacobian of every block once and stores the Huber-scaled
// rows; three assembly kernels read them back, all in ONE row order — sorted by (observation window, reference window, landmark), a window
// being 4 aligned knot intervals — so every pass reads the rows coalesced:
//   RepSideAcc<1>: [obs knots | camera] x same through the chunk accumulators (views of one frame share them);
//   RepSideAcc<0>: [ref knots | camera] x same + camera x camera + the gradient of all three;
//   k_reproj_cross: what is unique to a (reference frame, observation frame) pair or to a landmark — the cross block ref knots x obs knots,
//              accumulated per (reference window, observation window) GROUP on the matrix cores (a 42 x 42 tile: 7 knots on either side) and
//              added to HBM once per group, and the landmark's own row rho x [ref | obs | camera | rho | gradient] (56 entries per block).
// With ORB-like tracks (>= 100 observations per frame, co-visible frame pairs) a group holds tens of blocks and the cross terms cost
// ~1.8 k atomics per group instead of 576 per block; with one block per frame pair it degenerates to the per-block scatter (576 + 56 atomics
// per block, bound by the atomic rate: a CU retires one FP64 atomic lane every ~3.75 cycles, tools/probes/atomic_bw.hip).
struct RepJac { const double* J; const double* r; const int* k; int n; };   // k_reproj_jac output: J[(a * RJ + c) * n + i] with RJ = REP_NC (+ 1: the camera time-offset column 55 when it is free), r[a * n + i], k[i] = ref interval, k[n + i] = obs interval (-1: skipped)
// TAU: the camera time offset is free — one more global column (the row's column 55, tangent 6 N + 21) rides with the camera block
template <int SIDE, bool TAU = false> struct RepSideAcc {   // SIDE 0: the reference view's pose, 1: the observation's
  enum { PMAJ = 1 };
  enum { NK = 24, NG = 6 + (TAU ? 1 : 0), NR = 2, HUB = -1, KPK = 6, LVO = 0, WS = 1, GL = 16, SKIP_GG = SIDE, SECONDARY = 1, LB = 16, NCP = 31 + (TAU ? 1 : 0), USE_PRE = 0, OCC = 1, FAM = LVX_FAM_REPROJ };
  __device__ static constexpr int jm(int c) { return c; }   // [knots of this side | camera]
  static constexpr int RJ = REP_NC + (TAU ? 1 : 0);
  int n; RepJac jac;
  double huber;
  __device__ int eval(const DevCommon&, const SplineRef&, const Cal&, const HubShared*, int si, double r[NR], double (*J)[NCP], int& key, Aux& aux) const {
    const int k = jac.k[(size_t)SIDE * jac.n + si];
    if (jac.k[(size_t)jac.n + si] < 0) return -1;
    key = k; aux.wid = k;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      r[a] = jac.r[(size_t)a * jac.n + si];
#pragma unroll
      for (int c = 0; c < 24; ++c) J[a][c] = jac.J[(size_t)(a * RJ + 24 * SIDE + c) * jac.n + si];
#pragma unroll
      for (int c = 0; c < 6; ++c) J[a][24 + c] = jac.J[(size_t)(a * RJ + 48 + c) * jac.n + si];
      if (TAU) J[a][30] = jac.J[(size_t)(a * RJ + 55) * jac.n + si];
    }
    return RES_OK;
  }
  __device__ static int gcol(int g, int N, int nt) { return 6 * N + 15 + g; }   // g = 6 (TAU): 6 N + 21
};

// Reprojection phase 1 on its own: residual + Jacobian of every block (observation order), Huber-scaled, to HBM (0.9 KB per block);
// cost and residual output happen here.  The two-pose rolling-shutter residual needs > 512 registers when it shares a kernel with the
// assembly, so the MFMA passes read the rows back instead (2 x 45 MB at config 4, nothing against their atomics).
template <bool TAU>
__global__ __launch_bounds__(64) void k_reproj_jac(ReprojFamT<TAU> fam, DevCommon cm, double* Jb, double* rb, int* kb, long long row0, double* Trec) {
  constexpr int RJ = REP_NC + (TAU ? 1 : 0), RW = 56 + (TAU ? 1 : 0);
  constexpr int RS = RW | 1;           // odd LDS stride: a lane writes its record's words one instruction at a time
  extern __shared__ double trec_s[];   // [64 * RS] when there are records to stage (Trec != null), nothing otherwise (29 KB of static LDS capped every launch at 5 workgroups per CU)
  const int lane = threadIdx.x, si = blockIdx.x * 64 + lane, n = fam.n;
  const int rep = blockIdx.x % cm.nrep;
  double mycost = 0.0;
  if (si < n) {
    const SplineRef sp{cm.t0, cm.dt, cm.N, cm.state, cm.state + 3 * (size_t)cm.N};
    const Cal cal = load_cal(cm);
    HubShared none;
    double r[2], J[2][RJ];
    Keys key{-1, -1, -1};
    const int status = fam.eval_pre(cm, sp, cal, si, r, J, key);
    if (status != RES_OK) { if (status == RES_OUTSIDE) fallback_row(cm, LVX_FAM_REPROJ, si); else atomicOr(cm.err, status); kb[si] = -1; kb[n + si] = -1; }
    else {
      double scale;
      mycost = 0.5 * huber_rho(fam.huber, r[0] * r[0] + r[1] * r[1], &scale);
      if (cm.residuals) { const long long orow = row0 + (long long)fam.perm[si] * 2; cm.residuals[orow] = r[0]; cm.residuals[orow + 1] = r[1]; }
      kb[si] = key.k0; kb[n + si] = key.k1;
      if (cm.what & LVX_EVAL_NORMAL_EQ) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          rb[(size_t)a * n + si] = r[a] * scale;
#pragma unroll
          for (int c = 0; c < RJ; ++c) Jb[(size_t)(a * RJ + c) * n + si] = J[a][c] * scale;
        }
        if (Trec) {   // the landmark's own row of this block: rho x [ref knots 24 | obs knots 24 | camera 6 | rho | gradient (| camera time offset)] — one contiguous record per
          // block for k_reproj_lmrows (round 5b: formed here, where the rows sit in registers; the cross-term kernel read them back for it: 11 of its 58 us).  Through LDS:
          // the records of the wavefront's 64 blocks are contiguous in memory, a lane's own record is not a coalesced store.
          double* rec = trec_s + lane * RS;
          const double j0 = J[0][54] * scale, j1 = J[1][54] * scale;
#pragma unroll
          for (int c = 0; c < 54; ++c) rec[c] = j0 * (J[0][c] * scale) + j1 * (J[1][c] * scale);
          rec[54] = j0 * j0 + j1 * j1;
          rec[55] = j0 * (r[0] * scale) + j1 * (r[1] * scale);
          if (TAU) rec[56] = j0 * (J[0][55] * scale) + j1 * (J[1][55] * scale);
        }
      }
    }
  }
  if (Trec && (cm.what & LVX_EVAL_NORMAL_EQ)) {   // (records of skipped blocks are never read: k_reproj_lmrows looks at the block's interval first)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nrec = min(64, n - (int)blockIdx.x * 64);
    double* dst = Trec + (size_t)blockIdx.x * 64 * RW;
    for (int e = lane; e < nrec * RW; e += 64) { const int r_ = e / RW; dst[e] = trec_s[r_ * RS + (e - r_ * RW)]; }
  }
  mycost = wave_sum(mycost);
  if (lane == 0) atomicAdd(&cm.cost[rep], mycost);
}

// Reprojection cross terms (see the comment above RepSideAcc).  A wavefront owns one GROUP = the blocks whose reference interval lies in
// the aligned window [4 w0, 4 w0 + 4) and whose observation interval lies in [4 w1, 4 w1 + 4): both sides have a 7-knot (42-column) local
// space.  Eight blocks at a time, 8 lanes per block (side, residual row, half of the 24 knot columns) load the materialised rows into two
// LDS panels; T = J_ref^T J_obs accumulates in 3 x 3 MFMA tiles over the whole group and leaves with one atomic per non-zero entry.
// The landmark's own row (rho x everything, 56 entries per block) is formed from the same registers with one lane exchange.
typedef double d4 __attribute__((ext_vector_type(4)));
struct RepCross { RepJac jac; const int* lm; const int* goff; const int* gw; int ng; double* T; const int* det_list; };   // det_list: deterministic mode — block b works on group det_list[b] alone   // T[i][56]: the landmark-row products of block i (k_reproj_lmrows)   // gw[g] = w0, gw[ng + g] = w1
// TAU (free camera time offset): rows of RJ = 56 entries, landmark records of RW = 57 (rho x tau last).
// STRAYS.  The groups are fixed at layout time from the view times at tau = 0; with a non-zero offset (free, or locked at a non-zero value) a view within |tau| of a knot
// lands in the neighbouring interval and — one row in ~80 at the 1 ms bound — outside its 4-interval window.  Such a block stays out of the group's panels and adds
// its 24 x 24 cross products one by one (576 atomics; rare), so the pass never has to fall back to the per-segment kernels for it.
// RX_NW wavefronts per workgroup: 51 KB of LDS, three workgroups per CU = 2 304 resident wavefronts — config 4 has 2 108 groups, and with four wavefronts per workgroup
// (68 KB, two per CU, 2 048 resident) the last 15 workgroups ran as a second round behind the first.
#ifndef RX_NW
#define RX_NW 3
#endif
template <bool TAU>
__global__ __launch_bounds__(64 * RX_NW) void k_reproj_cross(RepCross rc, DevCommon cm) {
  constexpr int LDP = 49, BR = 16;   // panel: 16 rows (8 blocks x 2 residual rows) x 48 columns, odd stride
  constexpr int RJ = REP_NC + (TAU ? 1 : 0), RW = 56 + (TAU ? 1 : 0);
  __shared__ double pan[RX_NW][2][BR * LDP];
  __shared__ double sbuf[RX_NW][2][48];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rep = blockIdx.x % cm.nrep;
  if (rc.det_list && wv != 0) return;   // (the kernel has wavefront barriers only)
  const int part = lane & 7, b = lane >> 3, side = part >> 2, a = (part >> 1) & 1, h = part & 1;
  const int n = rc.jac.n, N = cm.N;
  double* Pr = pan[wv][0];
  double* Po = pan[wv][1];
  const int frag_off = (lane >> 4) * LDP + (lane & 15);
  const bool lm_free = !(cm.locks & LVX_LOCK_LANDMARKS);
  for (int g = rc.det_list ? rc.det_list[blockIdx.x] : blockIdx.x * RX_NW + wv; g < rc.ng; g += rc.det_list ? rc.ng : gridDim.x * RX_NW) {
    const int m0 = rc.goff[g], m1 = rc.goff[g + 1], w0 = rc.gw[g], w1 = rc.gw[rc.ng + g];
    d4 D[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) D[t] = d4{0.0, 0.0, 0.0, 0.0};
    for (int base = m0; base < m1; base += 8) {
      const int i = base + b;
      const bool in = i < m1;
      // intervals AND rows requested together (clamped address, selected afterwards: the rows of a skipped block were never written) — the rows behind the test of the
      // interval were a second, dependent memory round trip per eight blocks
      const int ic = min(i, m1 - 1);
      const int k0r = rc.jac.k[ic], k1r = rc.jac.k[(size_t)n + ic];
      double v[12];
      {
        const double* src = rc.jac.J + (size_t)(a * RJ + 24 * side + 12 * h) * n + ic;
#pragma unroll
        for (int c = 0; c < 12; ++c) v[c] = src[(size_t)c * n];
      }
      const int k0 = in ? k0r : -1, k1 = in ? k1r : -1;
      const bool live = in && k1 >= 0;
      const int o0 = k0 - 4 * w0, o1 = k1 - 4 * w1;
      const bool stray = live && (o0 < 0 || o0 > 3 || o1 < 0 || o1 > 3);   // the camera time offset moved a view out of its window
      const bool valid = live && !stray;
#pragma unroll
      for (int c = 0; c < 12; ++c) v[c] = live ? v[c] : 0.0;
      for (int e = lane; e < 2 * BR * LDP; e += 64) Pr[e] = 0.0;   // both panels are contiguous
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      {
        double* dst = (side ? Po : Pr) + (2 * b + a) * LDP + 6 * (side ? o1 : o0) + 12 * h;
        if (valid) {
#pragma unroll
          for (int c = 0; c < 12; ++c) dst[c] = v[c];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int nks = (2 * min(8, m1 - base) + 3) >> 2;
      for (int ks = 0; ks < nks; ++ks) {
        double fr[3], fo[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { fr[c] = Pr[ks * 4 * LDP + frag_off + c * 16]; fo[c] = Po[ks * 4 * LDP + frag_off + c * 16]; }
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
          for (int cj = 0; cj < 3; ++cj) D[ci * 3 + cj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[ci], fo[cj], D[ci * 3 + cj], 0, 0, 0);
      }
      {   // strays: the 24 x 24 cross block of each, entry by entry
        unsigned long long sm = __ballot(stray && part == 0);
        while (sm) {
          const int sl = __ffsll((long long)sm) - 1; sm &= sm - 1;
          const int sb = sl >> 3;
          const int sk0 = __shfl(k0, sl), sk1 = __shfl(k1, sl);
          __builtin_amdgcn_wave_barrier();
          if (b == sb) {
#pragma unroll
            for (int c = 0; c < 12; ++c) sbuf[wv][a][24 * side + 12 * h + c] = v[c];
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          for (int e2 = lane; e2 < 576; e2 += 64) {
            const int x = e2 / 24, y = e2 % 24;
            const double val = sbuf[wv][0][x] * sbuf[wv][0][24 + y] + sbuf[wv][1][x] * sbuf[wv][1][24 + y];
            if (val == 0.0) continue;
            const int pa = cm.ord[6 * (sk0 + x / 6) + x % 6], pb = cm.ord[6 * (sk1 + y / 6) + y % 6];
            if (pa == LVX_DEAD || pb == LVX_DEAD) continue;
            add_H(cm, pa, pb, pa == pb ? 2.0 * val : val, rep);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // T[x][y], x in the reference window, y in the observation window: one atomic per non-zero entry.  The same variable can sit in both
    // windows (views closer than 7 knots): both orders of a pair land on the same stored entry, the diagonal takes both.
    int pcol[3], prow[12];
#pragma unroll
    for (int cj = 0; cj < 3; ++cj) { const int y = cj * 16 + (lane & 15); pcol[cj] = y < 42 ? cm.ord[24 * w1 + y] : LVX_DEAD; }
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int vv = 0; vv < 4; ++vv) { const int x = ci * 16 + (lane >> 4) + 4 * vv; prow[ci * 4 + vv] = x < 42 ? cm.ord[24 * w0 + x] : LVX_DEAD; }
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int cj = 0; cj < 3; ++cj)
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) {
          const double val = D[ci * 3 + cj][vv];
          const int pa = prow[ci * 4 + vv], pb = pcol[cj];
          if (val == 0.0 || pa == LVX_DEAD || pb == LVX_DEAD) continue;
          add_H(cm, pa, pb, pa == pb ? 2.0 * val : val, rep);
        }
  }
}

// Landmark rows from the per-block records of k_reproj_cross: a wavefront OWNS landmark l — it sums the records of the landmark's blocks into
// an LDS image of the row [band couplings | border couplings | H_ll | g_l] and stores the whole row (no atomics, nothing to clear beforehand).
struct RepLmRows { const double* T; const int* k; int n; const int* ptr; const int* rows; int L; };   // blocks of landmark l: rows[ptr[l] .. ptr[l + 1])
template <bool TAU>
__global__ __launch_bounds__(256) void k_reproj_lmrows(RepLmRows q, DevCommon cm) {
  constexpr int RW = 56 + (TAU ? 1 : 0);   // record: [ref knots 24 | obs knots 24 | camera 6 | H_ll | g_l (| camera time offset)]
  extern __shared__ double rowbuf[];   // 4 x [lm_ls]
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l = blockIdx.x * 4 + wv;
  if (l >= q.L) return;
  double* rb = rowbuf + (size_t)wv * cm.lm_ls;
  for (int k = lane; k < cm.lm_ls; k += 64) rb[k] = 0.0;
  const int j0 = q.ptr[l], j1 = q.ptr[l + 1], p0 = cm.lm_p0[l], N = cm.N;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int jb = j0; jb < j1; jb += 4) {   // 4 blocks in flight
    double val[4]; int pc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      val[u] = 0.0; pc[u] = LVX_DEAD;
      if (jb + u < j1 && lane < RW) {
        const int i = q.rows[jb + u];
        const int kr = q.k[i], ko = q.k[(size_t)q.n + i];
        if (ko >= 0) {
          val[u] = q.T[(size_t)i * RW + lane];
          const int c = lane < 24 ? lane : lane - 24;
          pc[u] = lane < 48 ? cm.ord[6 * ((lane < 24 ? kr : ko) + c / 6) + c % 6] : (lane < 54 ? cm.ord[6 * N + 15 + (lane - 48)] : (lane < 56 ? LVX_LM_BASE : cm.ord[6 * N + 21]));   // 54: H_ll, 55: g_l, 56: camera time offset
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (pc[u] == LVX_DEAD || val[u] == 0.0) continue;
      int slot;
      if (lane == 54 || lane == 55) slot = cm.lm_wl + cm.nbd + (lane - 54);
      else if (pc[u] >= 0) { slot = pc[u] - p0; if (slot < 0 || slot >= cm.lm_wl) { atomicOr(cm.err, 4); continue; } }
      else slot = cm.lm_wl + (-1 - pc[u]);
      atomicAdd(&rb[slot], val[u]);   // LDS: the same variable can appear through both poses of a block
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  double* row = cm.lmH + (size_t)l * cm.lm_ls;
  for (int k = lane; k < cm.lm_ls; k += 64) row[k] = rb[k];
}

// ---------------------------------------------------------------------------------------------------------
// Fused reprojection kernel (round 6): residual + Jacobian + J^T J / J^T r + landmark rows of every block in ONE launch; no Jacobian row ever goes to HBM
// (the five-launch chain above writes 0.9 KB of rows and a 448-byte record per block and reads them back three times: 176 MB per pass for 3 MB of inputs).
//   * A WAVEFRONT owns a group = up to 64 blocks of one (reference window, observation window) pair, a window being 4 aligned knot intervals (7 knots, 42 columns) —
//     the unit k_reproj_cross already owned; the rolling shutter spreads the views of one frame over 3.3 intervals, so exact (interval, interval) pairs hold ~3 blocks
//     each and were measured hopeless (26 M atomics, 0.6 ms), windows hold tens.
//   * lane = block: the two-pose residual takes the whole 512-register budget (one wavefront per SIMD), its rows stay in registers.
//   * 16 lanes at a time write their two rows into a 32 x 96 LDS panel with the column space  [ref knots 42 | camera 6 || obs knots 42 | tau | r | rho | 0 0 0]  = six
//     16-column MFMA tiles; the 21 upper tile pairs accumulate over the whole group in registers and leave through LDS with ONE atomic per non-zero entry per group:
//     reference side, observation side, cross block, camera block and gradient together (the chain flushed the sides per 256-row chunk and the cross block per 32 blocks).
//   * the landmark's own row (rho x everything: 57 products per block) is formed from the panel, (block, column) per lane, and added to the landmark's row in HBM
//     (cleared by k_clear on this path: a landmark's views sit in different groups, nobody owns its row).
//   * windows are taken from the intervals the evaluation RETURNS: a camera time offset that moves a view out of its group's window makes that block a group of its own.
// Code size matters as much as instruction count: at one wavefront per SIMD nothing hides an instruction-cache miss.  The first version — scatter and landmark loops
// fully unrolled over register arrays, ~40 k instructions executed once per group — spent 260 us in the scatter and 170 in the landmark loop; the loops below read the
// panel / the dumped tiles back from LDS with a few dozen instructions per trip.
// MEASURED SLOWER than the chain at config 4 (218 vs ~120 us) and therefore opt-in (switch REP_FUSED = 1): evaluation 63 us (two rounds, 573 spilled registers next to the
// 21 accumulator tiles), MFMA 30, landmark atomics 51, scatter 69 — at one wavefront per SIMD the 6 M atomics retire at ~50 G/s, the chain's cross kernel (6-9 wavefronts
// per CU) at twice that.  What it does achieve: one launch, no Jacobian rows in HBM.
// ---------------------------------------------------------------------------------------------------------
struct RepFused { const int* g_start; const int* g_count; int ng; };   // group g: blocks [g_start[g], g_start[g] + g_count[g]) of the device row order, largest groups first
#define RF_LDP 97                      // panel row stride (doubles)
#define RF_WBUF (32 * RF_LDP + 48)     // per-wavefront LDS (doubles): the 32 x 96 panel, reused for 11 accumulator tiles at a time at the end of a group; + 96 column positions (int)
template <bool TAU>
__global__ __launch_bounds__(256, 1) void k_reproj_fused(ReprojFamT<TAU> fam, RepFused q, DevCommon cm, long long row0) {
  constexpr int RJ = REP_NC + (TAU ? 1 : 0), LDP = RF_LDP, TCOL = 90, RCOL = 91, LCOL = 92;   // panel columns: camera time offset, residual, the landmark's own (rho)
  __shared__ double smf[4 * RF_WBUF];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int N = cm.N;
  const bool want_ne = (cm.what & LVX_EVAL_NORMAL_EQ) != 0;
  const bool lm_free = want_ne && cm.L > 0 && !(cm.locks & LVX_LOCK_LANDMARKS);
  double* P = smf + wv * RF_WBUF;
  int* posL = (int*)(P + 32 * LDP);
  for (int g = blockIdx.x * 4 + wv; g < q.ng; g += gridDim.x * 4) {
    const int rep = g % cm.nrep;
    const int si = q.g_start[g] + lane;
    const bool in = lane < q.g_count[g];
    double r[2], J[2][RJ];
    Keys key{-1, -1, -1};
    bool valid = false;
    double mycost = 0.0;
    if (in) {
      const SplineRef sp{cm.t0, cm.dt, cm.N, cm.state, cm.state + 3 * (size_t)cm.N};
      const Cal cal = load_cal(cm);
      const int status = fam.eval_pre(cm, sp, cal, si, r, J, key);
      if (status == RES_OUTSIDE) fallback_row(cm, LVX_FAM_REPROJ, si);
      else if (status != RES_OK) atomicOr(cm.err, status);
      else {
        valid = true;
        double scale;
        mycost = 0.5 * huber_rho(fam.huber, r[0] * r[0] + r[1] * r[1], &scale);
        if (cm.residuals) { const long long orow = row0 + (long long)fam.perm[si] * 2; cm.residuals[orow] = r[0]; cm.residuals[orow + 1] = r[1]; }
        if (scale != 1.0) {
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            r[a] *= scale;
#pragma unroll
            for (int c = 0; c < RJ; ++c) J[a][c] *= scale;
          }
        }
      }
    }
    mycost = wave_sum(mycost);
    if (lane == 0 && mycost != 0.0) atomicAdd(&cm.cost[rep], mycost);
#ifdef RF_EVAL_ONLY
    if (cm.N > 0) continue;
#endif
    if (!want_ne) continue;
    const int my_w0 = valid ? key.k0 >> 2 : -1, my_w1 = valid ? key.k1 >> 2 : -1;
    const int my_o0 = 6 * (key.k0 & 3), my_o1 = 48 + 6 * (key.k1 & 3);   // first panel column of this block's reference / observation knots
    const int my_p0 = (valid && lm_free) ? cm.lm_p0[key.lm] : 0;
    unsigned long long rem = __ballot(valid);
    while (rem) {                                        // one (reference window, observation window) pair per trip — the group's own; a second trip only for strays
      const int lf = __ffsll((long long)rem) - 1;
      const int w0 = __builtin_amdgcn_readfirstlane(__shfl(my_w0, lf)), w1 = __builtin_amdgcn_readfirstlane(__shfl(my_w1, lf));
      const unsigned long long wm = __ballot(valid && my_w0 == w0 && my_w1 == w1);
      rem &= ~wm;
      for (int c = lane; c < 96; c += 64) {              // band / border position of panel column c for this pair of windows
        int tg = -1;
        if (c < 42) { if (4 * w0 + c / 6 < N) tg = 24 * w0 + c; }
        else if (c < 48) tg = 6 * N + 15 + (c - 42);
        else if (c < 90) { if (4 * w1 + (c - 48) / 6 < N) tg = 24 * w1 + (c - 48); }
        else if (c == TCOL && TAU) tg = 6 * N + 21;
        posL[c] = tg >= 0 ? cm.ord[tg] : LVX_DEAD;
      }
      d4 D[21];
#pragma unroll
      for (int t = 0; t < 21; ++t) D[t] = d4{0.0, 0.0, 0.0, 0.0};
      unsigned long long todo = wm;
      while (todo) {                                     // 16 lanes of the pair at a time through the panel
        const int l0 = __ffsll((long long)todo) - 1;
        const int ws = min(l0, 48);                      // the panel always maps 16 existing lanes
        const unsigned long long chunk = todo & (0xffffull << ws);
        todo &= ~chunk;
        for (int e = lane; e < 32 * LDP; e += 64) P[e] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if ((chunk >> lane) & 1ull) {
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            double* prow_ = P + ((lane - ws) * 2 + a) * LDP;
#pragma unroll
            for (int c = 0; c < 24; ++c) prow_[my_o0 + c] = J[a][c];
#pragma unroll
            for (int c = 0; c < 6; ++c) prow_[42 + c] = J[a][48 + c];
#pragma unroll
            for (int c = 0; c < 24; ++c) prow_[my_o1 + c] = J[a][24 + c];
            if (TAU) prow_[TCOL] = J[a][RJ - 1];
            prow_[RCOL] = r[a];
            prow_[LCOL] = J[a][54];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifndef RF_NO_LM
        if (lm_free) {   // landmark rows: rho x [ref | camera | obs | tau | r | rho] of the chunk's blocks, (block, column) per lane
          for (int e0 = 0; e0 < 16 * (LCOL + 1); e0 += 64) {   // (every lane takes every trip: the lane exchange below reads active lanes only)
            const int e = e0 + lane;
            const int li = min(e / (LCOL + 1), 15), c = e - li * (LCOL + 1);
            const int src = ws + li;
            const int p0 = __shfl(my_p0, src), l = __shfl(key.lm, src);
            if (e >= 16 * (LCOL + 1) || !((chunk >> src) & 1ull)) continue;
            const double* ra = P + (2 * li) * LDP;
            const double v = ra[LCOL] * ra[c] + ra[LDP + LCOL] * ra[LDP + c];
            if (v == 0.0) continue;
            int slot;
            if (c == RCOL) slot = cm.lm_wl + cm.nbd + 1;
            else if (c == LCOL) slot = cm.lm_wl + cm.nbd;
            else {
              const int p = posL[c];
              if (p == LVX_DEAD) continue;
              if (p >= 0) { slot = p - p0; if (slot < 0 || slot >= cm.lm_wl) { atomicOr(cm.err, 4); continue; } }
              else slot = cm.lm_wl + (-1 - p);
            }
            atomicAdd(&cm.lmH[(size_t)l * cm.lm_ls + slot], v);
          }
        }
#endif
        const int lhi = 63 - __clzll((long long)chunk);
        const int ks0 = (l0 - ws) >> 1, ks1 = (2 * (lhi - ws + 1) + 3) >> 2;   // k-steps (4 panel rows each) that hold rows of the chunk; rows of other lanes are zero
#ifndef RF_NO_MFMA
        for (int ks = ks0; ks < ks1; ++ks) {
          double f[6];
          const double* src = P + (4 * ks + (lane >> 4)) * LDP + (lane & 15);
#pragma unroll
          for (int c = 0; c < 6; ++c) f[c] = src[c * 16];
          int t = 0;
#pragma unroll
          for (int ci = 0; ci < 6; ++ci)
#pragma unroll
            for (int cj = ci; cj < 6; ++cj, ++t) D[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[ci], f[cj], D[t], 0, 0, 0);
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                  // the next chunk / the tile dump overwrites these rows
      }
      // the pair's accumulator tiles -> LDS (11 at a time, tile at P + 256 (t mod 11), row-major 16 x 16) -> one atomic per non-zero entry of the upper triangle;
      // the residual column is the gradient, the rho column has no position (its products are the landmark rows above)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int t = 11 * h; t < (h ? 21 : 11); ++t)
#pragma unroll
          for (int v = 0; v < 4; ++v) P[(t - 11 * h) * 256 + ((lane >> 4) + 4 * v) * 16 + (lane & 15)] = D[t][v];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifndef RF_NO_FLUSH
        for (int ci = 0, t = 0; ci < 6; ++ci)
          for (int cj = ci; cj < 6; ++cj, ++t) {
            if (t / 11 != h) continue;
            for (int e = lane; e < 256; e += 64) {
              const int row = ci * 16 + (e >> 4), col = cj * 16 + (e & 15);
              const double val = P[(t - 11 * h) * 256 + e];
              if (val == 0.0 || col < row) continue;
              const int pa = posL[row], pb = posL[col];
              if (row == RCOL) { if (col != RCOL && pb != LVX_DEAD) add_g(cm, pb, val, rep); continue; }
              if (col == RCOL) { if (pa != LVX_DEAD) add_g(cm, pa, val, rep); continue; }
              if (pa == LVX_DEAD || pb == LVX_DEAD) continue;
              add_H(cm, pa, pb, (pa == pb && row != col) ? 2.0 * val : val, rep);   // the same variable through both poses (views closer than 7 knots)
            }
          }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

#define ACC_BW 24

// ---------------------------------------------------------------------------------------------------------
// MFMA assembly path (the production fast path).  A workgroup owns CR knot intervals and
// accumulates into LDS — but J^T J is a tall-skinny FP64 matrix product on the matrix cores:
//   * rows are sorted by knot interval, so a run of rows whose intervals fall into a WINDOW of WS consecutive intervals shares
//     one local column space  [ (WS+3) knots x KPK | NG globals | residual ]  (<= 48 columns = 3 MFMA column tiles);
//   * GL lanes at a time write their rows, shifted to their knot offset inside the window and zero elsewhere, into a per-wave
//     LDS panel P[GL*NR][LDP]; every 4 panel rows are one k-step of v_mfma_f64_16x16x4_f64 for each upper-triangular tile pair
//     (operand of tile c at lane l = P[4 ks + (l >> 4)][16 c + (l & 15)]; it is both the A fragment of J^T and the B fragment of J);
//   * the residual rides along as one more column, so J^T r falls out of the same products;
//   * at the end of a window the accumulator tiles (D: col = lane & 15, row = (lane >> 4) + 4 reg) are added to the workgroup's
//     LDS accumulators, which are flushed to HBM once per workgroup.
// ---------------------------------------------------------------------------------------------------------

// families with F::Row / F::load / F::eval_row have their raw row inputs loaded one batch ahead
struct NoRow {};
template <class F, class = void> struct RowOf { using type = NoRow; static constexpr bool prefetch = false; };
template <class F> struct RowOf<F, std::void_t<typename F::Row>> { using type = typename F::Row; static constexpr bool prefetch = true; };

// F::PMAJ (families with WS == 1: every row of a window has the same local columns, no shift): the wavefront writes the rows of GL lanes into its
// panel ONCE and then walks the windows inside the panel — a window is the set of panel rows whose lanes share a knot interval; its k-steps
// run over that row span and mask the rows of other windows.  With the window-major loop a panel held one window (8 IMU samples: 8 of 64
// lanes wrote, 90 ds_write instructions per 8 samples); here all GL lanes write at once.  The end-of-window scatter takes its LDS targets from
// registers (window-independent for WS == 1: computed once per lane before the batch loop) instead of re-deriving the column classes.
template <class F, class = void> struct PanelMajor { static constexpr bool on = false; };
template <class F> struct PanelMajor<F, std::enable_if_t<(F::PMAJ > 0)>> { static constexpr bool on = true; static_assert(F::WS == 1, "panel-major needs shift-free windows"); };

template <class F> struct MfmaGeom {
  static constexpr int NKL = (F::WS + 3) * F::KPK;           // knot columns of a window
  static constexpr int NCL = NKL + F::NG + 1;                // + globals + residual
  static constexpr int NT = (NCL + 15) / 16;                 // 16-column tiles
  static constexpr int LDP = (NT * 16) | 1;                  // odd row stride: conflict-free row writes and fragment reads
  static constexpr int PR = F::GL * F::NR;                   // panel rows
  static_assert(PR % 4 == 0, "panel rows must be a multiple of the MFMA k-step");
  static constexpr int NTP = NT * (NT + 1) / 2;
};
template <class F> size_t mfma_lds_bytes(int cr) {
  const int LV = (cr + 5) * 6;
  return (size_t)(LV * ACC_BW + F::NG * LV + F::NG * F::NG + LV + F::NG + 64 + 4 * MfmaGeom<F>::PR * MfmaGeom<F>::LDP) * 8 + (F::USE_PRE ? (size_t)(cr + 4) * sizeof(So3Pre) : 0) +
         (size_t)(LV + F::NG) * 4 + 64;   // + 64 dummy slots of the panel-major families' branch-free scatter
}

// CR = knot intervals per workgroup, chosen per problem by the host (pick_chunk) so that the workgroup count fills whole rounds of the CUs
template <class F, int OCC>
__global__ __launch_bounds__(256, OCC) void k_family_mfma(F fam, DevCommon cm, const int* __restrict__ chunk_off, long long row0, int CR, int var_nch, const int* __restrict__ det_list) {
  using G = MfmaGeom<F>;
  constexpr int NK = F::NK, NG = F::NG, NC = F::NCP, NR = F::NR, KPK = F::KPK, WS = F::WS, GL = F::GL, LB = F::LB;
  constexpr int NKL = G::NKL, NT = G::NT, LDP = G::LDP, PR = G::PR;
  const int ACC_LV = (CR + 5) * 6;
  extern __shared__ double sm[];
  double* acc_band = sm;                              // [ACC_LV][ACC_BW]
  double* acc_bd = acc_band + ACC_LV * ACC_BW;        // [NG][ACC_LV]
  double* acc_gg = acc_bd + NG * ACC_LV;              // [NG][NG]
  double* acc_gk = acc_gg + NG * NG;                  // [ACC_LV]
  double* acc_gG = acc_gk + ACC_LV;                   // [NG]
  const int dummy = (int)(acc_gG + NG - sm);          // 64 doubles: where a lane adds +0.0 when its tile position has no accumulator entry (branch-free scatter)
  double* panels = acc_gG + NG + 64;                  // 4 x [PR][LDP]
  So3Pre* pre_tab = (So3Pre*)(panels + 4 * PR * LDP); // [CR + 4] control-point pairs (k_lo + e, k_lo + e + 1), families with USE_PRE
  int* kpos = (int*)(pre_tab + (F::USE_PRE ? CR + 4 : 0));   // [ACC_LV]
  int* gpos = kpos + ACC_LV;                          // [NG]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ch = det_list ? det_list[blockIdx.x] : blockIdx.x;   // deterministic mode: the chunks of one colour (disjoint knot ranges), one wavefront each
  const int nwv = det_list ? 1 : 4;
  const int m0 = chunk_off[ch], m1 = chunk_off[ch + 1];
  if (m0 >= m1) return;
#ifdef LVX_KTIME
  long long kt_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long kt0_ = __builtin_amdgcn_s_memtime(), kts_ = kt0_;
#define KT(i) { const long long n_ = __builtin_amdgcn_s_memtime(); kt_[i] += n_ - kt0_; kt0_ = n_; }
#else
#define KT(i)
#endif
  // var_nch == 0: chunk c covers the intervals [c CR, (c + 1) CR); else: chunks of equal row count, their first interval in the table behind the offsets
  const int k_lo = (var_nch ? chunk_off[var_nch + 1 + ch] : ch * CR) - 1;
  const int nt = 6 * cm.N + 22 + cm.L;
  const bool want_ne = (cm.what & LVX_EVAL_NORMAL_EQ) != 0;
  // the accumulator rows this chunk can touch: its own knot span (+ 1 interval either side for a time offset, + 3 knots), not the launch-wide maximum CR — the
  // reprojection side passes allow 48 intervals per chunk and typically hold one camera frame (4): clearing and flushing 318 rows for 60 was a third of the kernel
  const int LVU = var_nch ? min(ACC_LV, (chunk_off[2 * var_nch + 1 + ch] + 6) * 6) : ACC_LV;
  for (int e = tid; e < LVU * ACC_BW; e += 256) acc_band[e] = 0.0;
  for (int g = 0; g < NG; ++g) for (int e = tid; e < LVU; e += 256) acc_bd[g * ACC_LV + e] = 0.0;
  for (int e = tid; e < NG * NG + LVU; e += 256) { if (e < NG * NG) acc_gg[e] = 0.0; else acc_gk[e - NG * NG] = 0.0; }
  if (tid < NG) acc_gG[tid] = 0.0;
  for (int e = tid; e < 4 * PR * LDP; e += 256) panels[e] = 0.0;
  for (int e = tid; e < LVU; e += 256) { const int k = k_lo + e / 6; kpos[e] = (k >= 0 && k < cm.N) ? cm.ord[6 * k + e % 6] : LVX_DEAD; }
  if (tid < NG) gpos[tid] = cm.ord[F::gcol(tid, cm.N, nt)];
  const SplineRef sp{cm.t0, cm.dt, cm.N, cm.state, cm.state + 3 * (size_t)cm.N};
  if (F::USE_PRE && tid < CR + 4) {   // this chunk's slice of the pass's control-point-pair table
    const int ka = k_lo + tid;
    if (ka >= 0 && ka + 1 < cm.N) pre_tab[tid] = cm.pre[ka];
    else { pre_tab[tid].Om = mk(0, 0, 0); pre_tab[tid].on = 0.0; pre_tab[tid].Jri = m3_identity(); pre_tab[tid].c3 = 1.0 / 12.0; pre_tab[tid].ok = 1; }
  }
#ifdef LVX_KTIME
  const PreWin pwin{pre_tab, k_lo, CR + 4, kt_};
#else
  const PreWin pwin{pre_tab, k_lo, CR + 4};
#endif
  const Cal cal = load_cal(cm);
  const HubShared* hub = F::HUB >= 0 ? ((const HubShared*)cm.hubs) + F::HUB : nullptr;
  double* P = panels + wv * (PR * LDP);
  const int rep = ch % cm.nrep;
  // class of a window-local column: >= 0 knot scalar (offset inside the window, in units of tangent scalars), -1-g global g, -100 residual, -200 padding
  auto cls = [](int lc) { return lc < NKL ? 6 * (lc / KPK) + F::LVO + lc % KPK : (lc < NKL + NG ? -1 - (lc - NKL) : (lc == NKL + NG ? -100 : -200)); };
  const int frag_off = (lane >> 4) * LDP + (lane & 15);
  double mycost = 0.0;
  // panel-major families: where accumulator register (tile pair t, v) of THIS lane goes at the end of a window — LDS index (doubles, relative
  // to sm) | multiplier of the window base (0: none, 1: wb, 2: wb * ACC_BW) << 16 | largest accumulator row touched (relative to wb) << 18, or -1
  int rtab[PanelMajor<F>::on ? G::NTP * 4 : 1];
  if constexpr (PanelMajor<F>::on) {
    int t = 0;
#pragma unroll
    for (int ci = 0; ci < NT; ++ci)
#pragma unroll
      for (int cj = ci; cj < NT; ++cj, ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = ci * 16 + (lane >> 4) + 4 * v, col = cj * 16 + (lane & 15);
          int ent = -1;
          if (!(ci == cj && col < row)) {
            const int ra = cls(row), cb = cls(col);
            if (ra >= 0) {
              if (cb >= 0) { const int d = cb - ra; if (d >= 0 && d < ACC_BW) ent = ((int)(acc_band - sm) + ra * ACC_BW + d) | (2 << 16) | (cb << 18); }
              else if (cb > -100) ent = ((int)(acc_bd - sm) + (-1 - cb) * ACC_LV + ra) | (1 << 16) | (ra << 18);
              else if (cb == -100) ent = ((int)(acc_gk - sm) + ra) | (1 << 16) | (ra << 18);
            } else if (ra > -100 && !F::SKIP_GG) {
              if (cb > -100 && cb < 0) ent = (int)(acc_gg - sm) + (-1 - ra) * NG + (-1 - cb);
              else if (cb == -100) ent = (int)(acc_gG - sm) + (-1 - ra);
            }
          }
          rtab[t * 4 + v] = ent;
        }
  }
  typename RowOf<F>::type nxt{};
  if constexpr (RowOf<F>::prefetch) { const int s0 = m0 + wv * LB + lane; if (wv < nwv && lane < LB && s0 < m1) nxt = fam.load(s0); }
  __syncthreads();
  KT(0)
  for (int base = wv < nwv ? m0 + wv * LB : m1; base < m1; base += nwv * LB) {   // LB rows per wave batch: sparse families spread their rows over the 4 waves
    const int si = base + lane;
    const bool in = lane < LB && si < m1;
    const typename RowOf<F>::type cur = nxt;
    double r[NR];
    double J[NR][NC];
    int key = -1;
    Aux aux{-1, 0, 0, &pwin};
    bool valid = false;
#ifdef LVX_KTIME
    kt_[15] = __builtin_amdgcn_s_memtime();
#endif
    if (in) {
      int status;
      if constexpr (RowOf<F>::prefetch) status = fam.eval_row(cm, sp, cal, hub, cur, r, J, key, aux);
      else status = fam.eval(cm, sp, cal, hub, si, r, J, key, aux);
      if (aux.wid < 0) aux.wid = key;
      valid = status == RES_OK;
      if (valid && (key < k_lo || key - k_lo > CR + 1 || 6 * (key - k_lo + 4) > LVU)) { valid = false; fallback_row(cm, F::FAM, si); }
      else if (!valid && status == LVX_ERR_FALLBACK) fallback_row(cm, F::FAM, si);
      else if (!valid && status > 0) atomicOr(cm.err, status);   // status < 0: row skipped (reported by the kernel that produced it)
    }
    if constexpr (RowOf<F>::prefetch) { const int sn = si + nwv * LB; if (lane < LB && sn < m1) nxt = fam.load(sn); }   // next batch's rows: in flight during this batch's assembly
    if (valid) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < NR; ++a) s += r[a] * r[a];
      double scale;
      const double rho_s = huber_rho(fam.huber, s, &scale);
      if constexpr (!F::SECONDARY) {
        mycost += 0.5 * rho_s;
        if (cm.residuals) {
          const long long orow = row0 + (long long)fam.perm[si] * NR;
#pragma unroll
          for (int a = 0; a < NR; ++a) cm.residuals[orow + a] = r[a];
        }
      }
      if (scale != 1.0) {
#pragma unroll
        for (int a = 0; a < NR; ++a) {
          r[a] *= scale;
#pragma unroll
          for (int c = 0; c < NC; ++c) J[a][c] *= scale;
        }
      }
    }
    if (!want_ne) continue;
    KT(1)
    if constexpr (PanelMajor<F>::on) {
      const unsigned long long vm = __ballot(valid);
      for (int g0 = 0; g0 < LB; g0 += GL) {                  // one panel = the rows of GL lanes
        const unsigned long long pmask = (GL >= 64 ? ~0ull : ((1ull << GL) - 1ull)) << g0;
        unsigned long long rem = vm & pmask;
        if (!rem) continue;                                   // wave-uniform
        if (lane >= g0 && lane < g0 + GL) {
          const int li = lane - g0;
#pragma unroll
          for (int a = 0; a < NR; ++a) {
            double* prow = P + (li * NR + a) * LDP;
#pragma unroll
            for (int c = 0; c < NK; ++c) prow[c] = valid ? J[a][F::jm(c)] : 0.0;
#pragma unroll
            for (int g = 0; g < NG; ++g) prow[NKL + g] = valid ? J[a][F::jm(NK + g)] : 0.0;
            prow[NKL + NG] = valid ? r[a] : 0.0;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        KT(2)
        while (rem) {                                         // windows inside the panel (wave-uniform control flow)
          const int l0 = __ffsll((long long)rem) - 1;
          const int kw = __builtin_amdgcn_readfirstlane(__shfl(key, l0));
          const unsigned long long wm = __ballot(valid && key == kw) & pmask;
          rem &= ~wm;
          const int lhi = 63 - __clzll((long long)wm);
          const int wb = (kw - k_lo) * 6;
          const int r_lo = (l0 - g0) * NR, r_hi = (lhi - g0 + 1) * NR;   // the window's row span in the panel
          const unsigned long long wsh = wm >> g0;
          d4 D[G::NTP];
#pragma unroll
          for (int t = 0; t < G::NTP; ++t) D[t] = d4{0.0, 0.0, 0.0, 0.0};
          const int nks = (r_hi - r_lo + 3) >> 2;
          const bool contig = __popcll(wm) == lhi - l0 + 1;   // sorted families: always (wave-uniform)
          // U k-steps per trip, their LDS reads issued before the first product: one read -> select -> MFMA chain per k-step leaves the LDS
          // latency exposed (measured: 480 cycles per k-step).  A full interval of 8 IMU samples is 6 k-steps = one trip; a padded k-step
          // multiplies zeros.
          auto trip = [&](int ks, auto UC) {
            constexpr int U = decltype(UC)::value;
            double f[U][NT];
#pragma unroll
            for (int q = 0; q < U; ++q) {
              const int rr = r_lo + 4 * (ks + q) + (lane >> 4);
              bool mine = rr < r_hi;
              if (!contig) mine = mine && ((wsh >> (rr / NR)) & 1ull);   // rows of other windows inside the span contribute zeros
              const double* src = P + min(rr, PR - 1) * LDP + (lane & 15);
#pragma unroll
              for (int c = 0; c < NT; ++c) { const double x = src[c * 16]; f[q][c] = mine ? x : 0.0; }
            }
#pragma unroll
            for (int q = 0; q < U; ++q) {
              int t = 0;
#pragma unroll
              for (int ci = 0; ci < NT; ++ci)
#pragma unroll
                for (int cj = ci; cj < NT; ++cj, ++t) D[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[q][ci], f[q][cj], D[t], 0, 0, 0);
            }
          };
          int ks = 0;
          for (; ks + 6 <= nks; ks += 6) trip(ks, std::integral_constant<int, 6>{});
          for (; ks < nks; ks += 2) trip(ks, std::integral_constant<int, 2>{});
          KT(3)
          const int wbB = wb * ACC_BW;
          // branch-free: every lane issues every LDS add — to its accumulator entry, or +0.0 to a dummy slot of its own (no entry at this tile position, or beyond the
          // accumulator window).  A `continue` per slot made every slot a basic block: branch + exec-mask juggling + the wait for the MFMA result, slot after slot
#pragma unroll
          for (int t = 0; t < G::NTP; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int te = rtab[t * 4 + v];
              const bool ok = te >= 0 && wb + (te >> 18) < ACC_LV;
              const int m = (te >> 16) & 3;
              atomicAdd(&sm[ok ? (te & 0xffff) + (m == 2 ? wbB : (m == 1 ? wb : 0)) : dummy + lane], ok ? D[t][v] : 0.0);
            }
          KT(4)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                      // the next panel overwrites these rows
      }
      continue;
    }
    unsigned long long rem = __ballot(valid);
    while (rem) {                                            // one window per iteration (wave-uniform control flow)
      const int l0 = __ffsll((long long)rem) - 1;
      const int kw = __builtin_amdgcn_readfirstlane(__shfl(key, l0));
      const int ww = __builtin_amdgcn_readfirstlane(__shfl(aux.wid, l0));
      const bool inw = valid && aux.wid >= ww && aux.wid < ww + WS && key >= kw && key < kw + WS;
      const unsigned long long wm = __ballot(inw);
      rem &= ~wm;
      const int lhi = 63 - __clzll((long long)wm);
      const int sh = inw ? (key - kw) * KPK : 0;
      const int wb = (kw - k_lo) * 6;
      d4 D[G::NTP];
#pragma unroll
      for (int t = 0; t < G::NTP; ++t) D[t] = d4{0.0, 0.0, 0.0, 0.0};
      for (int g0 = l0; g0 <= lhi; g0 += GL) {
        const int gs = min(g0, 64 - GL);                     // the panel always maps GL existing lanes, so every panel row is rewritten
        if (lane >= gs && lane < gs + GL) {
          const int li = lane - gs;
          const bool mrow = inw && lane >= g0;               // lanes before g0 belong to the previous panel of this window
#pragma unroll
          for (int a = 0; a < NR; ++a) {
            double* prow = P + (li * NR + a) * LDP;
#pragma unroll
            for (int c = 0; c < NK; ++c) prow[sh + c] = mrow ? J[a][F::jm(c)] : 0.0;
#pragma unroll
            for (int z = 0; z < (WS - 1) * KPK; ++z) prow[z < sh ? z : z + NK] = 0.0;
#pragma unroll
            for (int g = 0; g < NG; ++g) prow[NKL + g] = mrow ? J[a][F::jm(NK + g)] : 0.0;
            prow[NKL + NG] = mrow ? r[a] : 0.0;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        KT(2)
        const int cnt = min(GL, lhi + 1 - gs);
        const int nks = (cnt * NR + 3) >> 2;
        for (int ks = 0; ks < nks; ++ks) {
          double f[NT];
#pragma unroll
          for (int c = 0; c < NT; ++c) f[c] = P[ks * 4 * LDP + frag_off + c * 16];
          int t = 0;
#pragma unroll
          for (int ci = 0; ci < NT; ++ci)
#pragma unroll
            for (int cj = ci; cj < NT; ++cj, ++t) D[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[ci], f[cj], D[t], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        KT(3)
      }
      // window accumulators -> workgroup accumulators (LDS atomics; other waves work on overlapping windows).  (Round 5, measured: the table-driven branch-free scatter that
      // took 6 % off the IMU kernel makes this one 8 % slower — at two wavefronts per SIMD the branches hide behind the other wavefront, while the unconditional adds of the
      // structurally zero and lower-triangle entries load the LDS atomic unit.)
      {
      int t = 0;
#pragma unroll
      for (int ci = 0; ci < NT; ++ci)
#pragma unroll
        for (int cj = ci; cj < NT; ++cj, ++t)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const double val = D[t][v];
            if (val == 0.0) continue;
            if (ci == cj && (lane & 15) < (lane >> 4) + 4 * v) continue;   // lower triangle of a diagonal tile
            const int ra = cls(ci * 16 + (lane >> 4) + 4 * v), cb = cls(cj * 16 + (lane & 15));   // recomputed here: keeping them across phase 1 costs registers
            if (ra >= 0) {
              const int la = wb + ra;
              if (la >= ACC_LV) continue;
              if (cb >= 0) { const int d = cb - ra; if (d < ACC_BW && wb + cb < ACC_LV) atomicAdd(&acc_band[la * ACC_BW + d], val); }
              else if (cb > -100) atomicAdd(&acc_bd[(-1 - cb) * ACC_LV + la], val);
              else if (cb == -100) atomicAdd(&acc_gk[la], val);
            } else if (ra > -100) {
              const int ga = -1 - ra;
              if (cb > -100) {
                const int gb = -1 - cb;
                if (!F::SKIP_GG) atomicAdd(&acc_gg[ga * NG + gb], val);
              } else if (cb == -100) {
                if (!F::SKIP_GG) atomicAdd(&acc_gG[ga], val);
              }
            }
          }
      }
      KT(4)
    }
  }
  if (!F::SECONDARY) {
    mycost = wave_sum(mycost);
    if (lane == 0) atomicAdd(&cm.cost[rep], mycost);
  }
  if (!want_ne) return;
  KT(5)
  __syncthreads();
  KT(6)
  // flush the workgroup's accumulators: ONE global atomic per touched entry
  for (int e = tid; e < LVU * ACC_BW; e += 256) {
    const double v = acc_band[e];
    if (v == 0.0) continue;
    const int la = e / ACC_BW, lb = la + e % ACC_BW;
    if (lb >= LVU) continue;
    const int pa = kpos[la], pb = kpos[lb];
    if (pa == LVX_DEAD || pb == LVX_DEAD) continue;
    add_H(cm, pa, pb, v, rep);
  }
  for (int g = 0; g < NG; ++g)
    for (int e = tid; e < LVU; e += 256) {
      const double v = acc_bd[g * ACC_LV + e];
      if (v == 0.0) continue;
      const int pg = gpos[g], pk2 = kpos[e];
      if (pg == LVX_DEAD || pk2 == LVX_DEAD) continue;
      add_H(cm, pg, pk2, v, rep);
    }
  for (int e = tid; e < NG * NG; e += 256) {
    const int ga = e / NG, gb2 = e % NG;
    if (gb2 < ga) continue;
    const double v = acc_gg[e];
    if (v == 0.0 || gpos[ga] == LVX_DEAD || gpos[gb2] == LVX_DEAD) continue;
    add_H(cm, gpos[ga], gpos[gb2], v, rep);
  }
  for (int e = tid; e < LVU; e += 256) { const double v = acc_gk[e]; if (v != 0.0 && kpos[e] != LVX_DEAD) add_g(cm, kpos[e], v, rep); }
  if (tid < NG) { const double v = acc_gG[tid]; if (v != 0.0 && gpos[tid] != LVX_DEAD) add_g(cm, gpos[tid], v, rep); }
#ifdef LVX_KTIME
  KT(7)
#ifndef LVX_KTIME_FAM
#define LVX_KTIME_FAM SurfAcc
#endif
  if (std::is_same<F, LVX_KTIME_FAM>::value && lane == 0 && blockIdx.x % (gridDim.x / 8 + 1) == 5)
    printf("KT wg %d wv %d rows %d CR %d: init %lld eval %lld [seg %lld lookup %lld value %lld chain %lld pull %lld] panel %lld mfma %lld wflush %lld tail %lld sync %lld flush %lld total %lld\n", (int)blockIdx.x, wv, m1 - m0, CR,
           kt_[0], kt_[1], kt_[8], kt_[9], kt_[10], kt_[11], kt_[12], kt_[2], kt_[3], kt_[4], kt_[5], kt_[6], kt_[7], (long long)__builtin_amdgcn_s_memtime() - kts_);
#endif
}

// ---------------------------------------------------------------------------------------------------------
// Fused IMU kernel: gyroscope + accelerometer blocks of a sample in ONE workgroup pass.  Both residuals evaluate the same SO3 spline at the same
// time, touch the same 4 knots and flush into the same band entries (the gyroscope's are a subset of the accelerometer's): one pose evaluation,
// one set of LDS accumulators, one flush, and half the workgroup rounds of the two separate kernels (each of them runs one 4-wavefront workgroup
// per CU).  Panel-major assembly as in k_family_mfma, once per geometry:
//   G: 3 rows x [12 rotation columns | b_g (3) | r]  = 16 columns, one MFMA tile;     A: 3 rows x [24 knot columns | roll pitch b_a (5) | r] = 30, three tile pairs.
// Shared global list: [roll, pitch, b_a (3), b_g (3)] = tangent scalars 6 N + 0 .. 7.
// ---------------------------------------------------------------------------------------------------------
// LDP (panel row stride in doubles) is EVEN: rows start on 16 bytes and are written two columns per ds_write_b128 (odd strides took one ds_write_b64 per column with a
// quarter / half of the lanes active; the fragment reads — four rows x 16 consecutive columns per instruction — hit every bank at most three times either way)
struct ImuG { enum { NR = 3, NCJ = GYRO_NC, NK = 12, NG = 3, KPK = 3, LVO = 3, GL = 32, GOFF = 5, NKL = 12, NCL = 16, NT = 1, NTP = 1, LDP = 18, PR = 96 }; };
struct ImuA { enum { NR = 3, NCJ = ACC_NC, NK = 24, NG = 5, KPK = 6, LVO = 0, GL = 16, GOFF = 0, NKL = 24, NCL = 30, NT = 2, NTP = 3, LDP = 34, PR = 48 }; };
#define IMU_NGA 8
struct ImuAccLds { double* band; double* bd; double* gg; double* gk; double* gG; int lv; };
template <class PG> __device__ __forceinline__ int imu_cls(int lc) {
  return lc < PG::NKL ? 6 * (lc / PG::KPK) + PG::LVO + lc % PG::KPK : (lc < PG::NKL + PG::NG ? -1 - (PG::GOFF + lc - PG::NKL) : (lc == PG::NKL + PG::NG ? -100 : -200));
}
template <class PG> __device__ __forceinline__ void imu_build_rtab(int* rtab, int lane, const double* sm, const ImuAccLds& A) {
  int t = 0;
#pragma unroll
  for (int ci = 0; ci < PG::NT; ++ci)
#pragma unroll
    for (int cj = ci; cj < PG::NT; ++cj, ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = ci * 16 + (lane >> 4) + 4 * v, col = cj * 16 + (lane & 15);
        int ent = -1;
        if (!(ci == cj && col < row)) {
          const int ra = imu_cls<PG>(row), cb = imu_cls<PG>(col);
          if (ra >= 0) {
            if (cb >= 0) { const int d = cb - ra; if (d >= 0 && d < ACC_BW) ent = ((int)(A.band - sm) + ra * ACC_BW + d) | (2 << 16) | (cb << 18); }
            else if (cb > -100) ent = ((int)(A.bd - sm) + (-1 - cb) * A.lv + ra) | (1 << 16) | (ra << 18);
            else if (cb == -100) ent = ((int)(A.gk - sm) + ra) | (1 << 16) | (ra << 18);
          } else if (ra > -100) {
            if (cb > -100 && cb < 0) ent = (int)(A.gg - sm) + (-1 - ra) * IMU_NGA + (-1 - cb);
            else if (cb == -100) ent = (int)(A.gG - sm) + (-1 - ra);
          }
        }
        rtab[t * 4 + v] = ent;
      }
}
// rows of this wavefront's 64 lanes -> panel by panel -> windows (runs of equal knot interval) -> MFMA -> LDS accumulators
template <class PG> __device__ __forceinline__ void imu_assemble(double* sm, double* P, const int* rtab, bool valid, int key, int k_lo, int acc_lv, const double* r, const double (*J)[PG::NCJ], int lane, int dummy) {
  constexpr int NR = PG::NR, NT = PG::NT, LDP = PG::LDP, PR = PG::PR, GL = PG::GL;
  const unsigned long long vm = __ballot(valid);
  for (int g0 = 0; g0 < 64; g0 += GL) {
    const unsigned long long pmask = ((1ull << GL) - 1ull) << g0;
    unsigned long long rem = vm & pmask;
    if (!rem) continue;
    if (lane >= g0 && lane < g0 + GL) {
      const int li = lane - g0;
#pragma unroll
      for (int a = 0; a < NR; ++a) {
        double2* prow2 = (double2*)(P + (li * NR + a) * LDP);
        auto col = [&](int c) { return c < PG::NK + PG::NG ? (valid ? J[a][c] : 0.0) : (c == PG::NK + PG::NG ? (valid ? r[a] : 0.0) : 0.0); };   // [knots | globals | residual | tile padding (the other geometry's data sits there)]
#pragma unroll
        for (int c2 = 0; c2 < 8 * NT; ++c2) prow2[c2] = make_double2(col(2 * c2), col(2 * c2 + 1));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    while (rem) {
      const int l0 = __ffsll((long long)rem) - 1;
      const int kw = __builtin_amdgcn_readfirstlane(__shfl(key, l0));
      const unsigned long long wm = __ballot(valid && key == kw) & pmask;
      rem &= ~wm;
      const int lhi = 63 - __clzll((long long)wm);
      const int wb = (kw - k_lo) * 6;
      const int r_lo = (l0 - g0) * NR, r_hi = (lhi - g0 + 1) * NR;
      const unsigned long long wsh = wm >> g0;
      const bool contig = __popcll(wm) == lhi - l0 + 1;
      d4 D[PG::NTP];
#pragma unroll
      for (int t = 0; t < PG::NTP; ++t) D[t] = d4{0.0, 0.0, 0.0, 0.0};
      const int nks = (r_hi - r_lo + 3) >> 2;
      auto trip = [&](int ks, auto UC) {
        constexpr int U = decltype(UC)::value;
        double f[U][NT];
#pragma unroll
        for (int q = 0; q < U; ++q) {
          const int rr = r_lo + 4 * (ks + q) + (lane >> 4);
          bool mine = rr < r_hi;
          if (!contig) mine = mine && ((wsh >> (rr / NR)) & 1ull);
          const double* src = P + min(rr, PR - 1) * LDP + (lane & 15);
#pragma unroll
          for (int c = 0; c < NT; ++c) { const double x = src[c * 16]; f[q][c] = mine ? x : 0.0; }
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
          int t = 0;
#pragma unroll
          for (int ci = 0; ci < NT; ++ci)
#pragma unroll
            for (int cj = ci; cj < NT; ++cj, ++t) D[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[q][ci], f[q][cj], D[t], 0, 0, 0);
        }
      };
      int ks = 0;
      for (; ks + 6 <= nks; ks += 6) trip(ks, std::integral_constant<int, 6>{});
      for (; ks < nks; ks += 2) trip(ks, std::integral_constant<int, 2>{});
      const int wbB = wb * ACC_BW;
      // BRANCH-FREE scatter: every lane issues every LDS add — to its accumulator entry, or +0.0 to a dummy slot of its own when the tile position has no entry (lower
      // triangle, padding) or lies beyond the window.  With a `continue` per slot every slot was a basic block of its own: exec-mask juggling, a branch and the 76-cycle
      // wait for the MFMA result, one slot after the other (180 cycles per slot on the one wavefront per SIMD this kernel runs at).
#pragma unroll
      for (int t = 0; t < PG::NTP; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int te = rtab[t * 4 + v];
          const bool ok = te >= 0 && wb + (te >> 18) < acc_lv;
          const int m = (te >> 16) & 3;
          const int idx = ok ? (te & 0xffff) + (m == 2 ? wbB : (m == 1 ? wb : 0)) : dummy + lane;
          atomicAdd(&sm[idx], ok ? D[t][v] : 0.0);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}
struct ImuFused { int n; const double* t; const double* gyro; const double* acc; const int* perm; double w_gyro, w_acc; };
// OWNER-COMPUTES schedule of the fused kernel (round 4).  Round 3 launched one workgroup per chunk of 32 knot intervals (256 samples) and flushed its LDS accumulators
// with ~7.1 k FP64 atomics (39 MB of HBM writes for 11 MB of algorithmic traffic), 782 workgroups = 3.05 rounds of the 256 CUs.  Now:
//   * ONE workgroup per CU owns a contiguous range of samples (boundaries on knot intervals, equal row counts: 781 samples = three batches of 256 + one of 13 whose
//     assembly is 2 windows — the work quantum of the 64-lane evaluation no longer costs a fourth, nearly empty round);
//   * it walks its range batch by batch with the SAME LDS accumulators: the window [k_lo, k_lo + IMU_CR + 5) of knots slides with the data, the columns of the knots the
//     next batch can no longer touch are flushed and the rest carried over — so every band column, gradient entry and IMU-calibration border entry inside the range has
//     ONE writer, which STORES it (coalesced 16-byte stores, zeros included: these entries need no clear, k_clear reads the same ownership table);
//   * only the 5 knots either side of a range boundary, knots next to the hub gap and knots no window covers keep the cleared-and-added-atomically protocol;
//   * everything the batches look up — band positions, owners, the control-point-pair table — is loaded ONCE per workgroup for its whole knot range: between the flush
//     stores of one batch and the row loads of the next there is no vector-memory load (a load's s_waitcnt vmcnt(0) waits for every store issued before it).
// The kernel runs FIRST on the band (behind k_clear, before the LiDAR kernels), so the stores cannot overwrite another family's sums; every later family adds.
#define IMU_CR 32                        // knot intervals a batch may span
#define IMU_LV ((IMU_CR + 5) * 6)        // tangent scalars of the accumulator window (37 knots)
struct ImuOwn { const int* wg_c0; const int* own_k; int nch, span; };   // batches of workgroup g: [wg_c0[g], wg_c0[g + 1]); own_k[knot] = the workgroup that stores the knot's six band columns (-1: nobody); span: knots of the widest workgroup range
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's outstanding global stores (vmcnt(0))
#define LVX_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
static size_t imu_fused_lds_bytes(int span) {
  const int pan = std::max((int)ImuG::PR * (int)ImuG::LDP, (int)ImuA::PR * (int)ImuA::LDP);
  return (size_t)(IMU_LV * ACC_BW + IMU_NGA * IMU_LV + IMU_NGA * IMU_NGA + IMU_LV + IMU_NGA + 64 + 4 * pan) * 8 + (size_t)span * sizeof(So3Pre) + (size_t)(7 * span + IMU_NGA) * 4 + 64;
}
// the per-lane scatter targets of both geometries are constants of the kernel: built once per layout (64 lanes x 16 entries) instead of by every
// wavefront of every workgroup (~4 k cycles of integer divisions and branches each)
__global__ void k_imu_rtab(int* out) {
  const int lane = threadIdx.x;
  const double* sm = (const double*)out;    // only differences of the accumulator addresses enter the table
  ImuAccLds A;
  A.band = (double*)sm; A.bd = A.band + IMU_LV * ACC_BW; A.gg = A.bd + IMU_NGA * IMU_LV; A.gk = A.gg + IMU_NGA * IMU_NGA; A.gG = A.gk + IMU_LV; A.lv = IMU_LV;
  int rtg[ImuG::NTP * 4], rta[ImuA::NTP * 4];
  imu_build_rtab<ImuG>(rtg, lane, sm, A);
  imu_build_rtab<ImuA>(rta, lane, sm, A);
  for (int i = 0; i < ImuG::NTP * 4; ++i) out[i * 64 + lane] = rtg[i];
  for (int i = 0; i < ImuA::NTP * 4; ++i) out[(ImuG::NTP * 4 + i) * 64 + lane] = rta[i];
}
__global__ __launch_bounds__(256, 1) void k_imu_own(ImuFused fam, DevCommon cm, const int* __restrict__ chunk_off, ImuOwn ow, long long row0_g, long long row0_a, int det, const int* __restrict__ rtab_g) {
  constexpr int PAN = (ImuG::PR * ImuG::LDP > ImuA::PR * ImuA::LDP) ? ImuG::PR * ImuG::LDP : ImuA::PR * ImuA::LDP;
  constexpr int ACC_LV = IMU_LV, CR = IMU_CR, HB = ACC_BW / 2;
  extern __shared__ __attribute__((aligned(16))) double smo[];
  double* sm = smo;
  ImuAccLds A;
  A.band = sm; A.bd = A.band + ACC_LV * ACC_BW; A.gg = A.bd + IMU_NGA * ACC_LV; A.gk = A.gg + IMU_NGA * IMU_NGA; A.gG = A.gk + ACC_LV; A.lv = ACC_LV;
  double* panels = A.gG + IMU_NGA + 64;              // (64 dummy slots of the branch-free scatter sit between the accumulators and the panels)
  So3Pre* pre_all = (So3Pre*)(panels + 4 * PAN);     // [span] pairs (k_base + e, k_base + e + 1)
  int* kpos_all = (int*)(pre_all + ow.span);         // [6 span] band / border position of every tangent scalar of the range
  int* kown_all = kpos_all + 6 * ow.span;            // [span] owner of the knot's columns
  int* gpos = kown_all + ow.span;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wg = blockIdx.x;
  const int c0 = ow.wg_c0[wg], c1 = ow.wg_c0[wg + 1];
  if (c0 >= c1) return;
  const int nwv = det ? 1 : 4;
  const bool want_ne = (cm.what & LVX_EVAL_NORMAL_EQ) != 0;
  constexpr int NACC = ACC_LV * ACC_BW + IMU_NGA * ACC_LV + IMU_NGA * IMU_NGA + ACC_LV + IMU_NGA;
  for (int e = tid; e < NACC; e += 256) sm[e] = 0.0;   // (the panels need no clearing: every row a k-step reads is rewritten, the padding columns with it)
  int* urng = gpos + IMU_NGA;                        // [2]: first / last accumulator row of a batch end whose columns this workgroup does not store
  if (tid < IMU_NGA) gpos[tid] = cm.ord[6 * cm.N + tid];
  if (tid == 0) { urng[0] = 1 << 30; urng[1] = -1; }
  const int k_base = chunk_off[ow.nch + 1 + c0] - 1;   // one interval of slack below the first batch's first interval (a non-zero IMU time offset moves a sample by at most one)
  for (int e = tid; e < 6 * ow.span; e += 256) { const int kn = k_base + e / 6; kpos_all[e] = (kn >= 0 && kn < cm.N) ? cm.ord[6 * kn + e % 6] : LVX_DEAD; }
  for (int e = tid; e < ow.span; e += 256) {
    const int kn = k_base + e;
    kown_all[e] = (ow.own_k && kn >= 0 && kn < cm.N) ? ow.own_k[kn] : -1;
    if (kn >= 0 && kn + 1 < cm.N) pre_all[e] = cm.pre[kn];
    else { pre_all[e].Om = mk(0, 0, 0); pre_all[e].on = 0.0; pre_all[e].Jri = m3_identity(); pre_all[e].c3 = 1.0 / 12.0; pre_all[e].ok = 1; }
  }
  const SplineRef sp{cm.t0, cm.dt, cm.N, cm.state, cm.state + 3 * (size_t)cm.N};
  const Cal cal = load_cal(cm);
  if (fabs(cal.imu.tau) >= cm.dt && tid == 0) atomicOr(cm.err, LVX_ERR_FALLBACK);   // the ownership rule assumes a sample moves by at most one interval (reference: tau_imu = 0 always)
  double* P = panels + wv * PAN;
  const int rep = wg % cm.nrep;
  int rtg[ImuG::NTP * 4], rta[ImuA::NTP * 4];
#pragma unroll
  for (int i = 0; i < ImuG::NTP * 4; ++i) rtg[i] = rtab_g[i * 64 + lane];
#pragma unroll
  for (int i = 0; i < ImuA::NTP * 4; ++i) rta[i] = rtab_g[(ImuG::NTP * 4 + i) * 64 + lane];
  double mycost = 0.0;
  const int ld = cm.bw + 1;
#ifdef LVX_KTIME_IMU
  long long kt_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long kt0_ = __builtin_amdgcn_s_memtime(); const long long kts_ = kt0_;
#define IKT(i) { const long long n_ = __builtin_amdgcn_s_memtime(); kt_[i] += n_ - kt0_; kt0_ = n_; }
#else
#define IKT(i)
#endif
  for (int c = c0; c < c1; ++c) {
    const int m0 = chunk_off[c], m1 = chunk_off[c + 1];
    const int k_lo = chunk_off[ow.nch + 1 + c] - 1;   // window base knot of this batch
    const int woff = k_lo - k_base;
    const So3Pre* pre_tab = pre_all + woff;
    const int* kpos = kpos_all + 6 * woff;
    if (c == c0) __syncthreads(); else LVX_LDS_BARRIER();   // the tables (global loads: full barrier) / the previous batch's carry-over are in place
    IKT(0)
    for (int base = wv < nwv ? m0 + wv * 64 : m1; base < m1; base += nwv * 64) {
      const int si = base + lane;
      const bool in = si < m1;
      int key = -1;
      bool valid = false;
      So3Eval e;
      KnotRef k;
      if (in) {
        int status = RES_OK;
        if (!knot_lookup(sp.t0, sp.dt, sp.n, fam.t[si], fam.t[si] + cal.imu.tau, &k)) status = RES_RANGE;
        else {
          key = k.i0;
          const So3Pre* pre = (k.i0 >= k_lo && k.i0 + 2 < k_lo + CR + 4) ? pre_tab + (k.i0 - k_lo) : nullptr;
          if (!pre) status = RES_OUTSIDE;
          else {
            quat cq[4]; load_so3_cp(sp, k.i0, cq);
            const int bad = so3_eval_pre<true, true>(cq, pre, k.u, sp.dt, &e);
            if (bad) status = (bad & 1) ? RES_NONUNIT : RES_OUTSIDE;
          }
        }
        valid = status == RES_OK;
        if (valid && (key < k_lo || key - k_lo > CR + 1)) { valid = false; fallback_row(cm, LVX_FAM_GYRO, si); }
        else if (!valid && status == RES_OUTSIDE) fallback_row(cm, LVX_FAM_GYRO, si);   // (the sample: its gyroscope AND accelerometer block go to the exact kernels)
        else if (!valid) atomicOr(cm.err, status);
      }
      IKT(1)
      {   // gyroscope block (gyro_residual, lvx_resid.h)
        double r[3], J[3][GYRO_NC];
        if (valid) {
          const v3 wm = load_v3(fam.gyro + 3 * (size_t)si);
          const v3 pred = e.w_body + cal.imu.bg;
          const double w = fam.w_gyro;
          r[0] = w * (wm.x - pred.x); r[1] = w * (wm.y - pred.y); r[2] = w * (wm.z - pred.z);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
              for (int b = 0; b < 3; ++b) J[a][3 * kk + b] = -w * e.dw[kk].a[3 * a + b];
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) J[a][12 + b] = (a == b) ? -w : 0.0;
          mycost += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
          if (cm.residuals) { const long long orow = row0_g + (long long)fam.perm[si] * 3; cm.residuals[orow] = r[0]; cm.residuals[orow + 1] = r[1]; cm.residuals[orow + 2] = r[2]; }
        }
        if (want_ne) imu_assemble<ImuG>(sm, P, rtg, valid, key, k_lo, ACC_LV, r, J, lane, NACC);
        IKT(2)
      }
      {   // accelerometer block (accel_residual, lvx_resid.h)
        double r[3], J[3][ACC_NC];
        if (valid) {
          R3Basis b; r3_basis(k.u, sp.dt, &b);
          v3 acc = mk(0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = acc + b.Ba[j] * load_v3(sp.r3 + 3 * (k.i0 + j));
          const v3 am = load_v3(fam.acc + 3 * (size_t)si);
          const double w = fam.w_acc;
          const double G = -9.79;   // imu.h:25
          const double cr = cos(cal.imu.roll), sr = sin(cal.imu.roll), cp = cos(cal.imu.pitch), sp_ = sin(cal.imu.pitch);
          const v3 g = mk(-sp_ * cr * G, sr * G, -cr * cp * G);
          const v3 y = acc + g;
          const v3 yb = qrot_inv(e.q, y);
          const v3 pred = yb + cal.imu.ba;
          r[0] = w * (am.x - pred.x); r[1] = w * (am.y - pred.y); r[2] = w * (am.z - pred.z);
          const m3 Rt = transpose(rotmat(e.q));
          const m3 S = skew(yb);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const m3 Mx = S * e.dxi[kk];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
              for (int bb = 0; bb < 3; ++bb) { J[a][6 * kk + bb] = -w * b.Ba[kk] * Rt.a[3 * a + bb]; J[a][6 * kk + 3 + bb] = -w * Mx.a[3 * a + bb]; }
          }
          const v3 dg_dr = Rt * mk(sp_ * sr * G, cr * G, sr * cp * G);
          const v3 dg_dp = Rt * mk(-cp * cr * G, 0.0, cr * sp_ * G);
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            J[a][24] = -w * comp(dg_dr, a); J[a][25] = -w * comp(dg_dp, a);
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) J[a][26 + bb] = (a == bb) ? -w : 0.0;
          }
          mycost += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
          if (cm.residuals) { const long long orow = row0_a + (long long)fam.perm[si] * 3; cm.residuals[orow] = r[0]; cm.residuals[orow + 1] = r[1]; cm.residuals[orow + 2] = r[2]; }
        }
        if (want_ne) imu_assemble<ImuA>(sm, P, rta, valid, key, k_lo, ACC_LV, r, J, lane, NACC);
        IKT(3)
      }
    }
    if (!want_ne) continue;
    // ---- batch end: flush the columns no later batch of this workgroup can touch, carry the others over ----
    int bt = tid; asm volatile("" : "+v"(bt));   // opaque copy: the index arithmetic below is thread-invariant — hoisted out of the batch loop it would live (spilled) across the hot loop
    const bool more = c + 1 < c1;
    const int nfl = more ? min(ACC_LV, max(0, 6 * (chunk_off[ow.nch + 1 + c + 1] - 1 - k_lo))) : ACC_LV;
    const int* kown = kown_all + woff;   // per knot
    const bool un_ = bt < nfl && kpos[bt] != LVX_DEAD && kown[bt / 6] != wg;
    { const unsigned long long ub = __ballot(un_);
      if (ub && lane == 0) { atomicMin(&urng[0], wv * 64 + __ffsll((long long)ub) - 1); atomicMax(&urng[1], wv * 64 + 63 - __clzll((long long)ub)); } }
    const int any_unowned = __syncthreads_or(un_);   // (the barrier: every wavefront's sums are in the LDS accumulators)
    IKT(4)
    if (any_unowned) {
      // a range's first / last batch, the hub gap, hub knots = border rows: columns this workgroup does not store keep the cleared-and-added protocol.  Only the rows
      // [r0, r1) between the first and the last such column are walked (a handful of knots at a range boundary; walking all 222 rows cost 17 k cycles per such batch)
      const int r0 = urng[0], r1 = urng[1] + 1;
      for (int e = r0 * ACC_BW + bt; e < r1 * ACC_BW; e += 256) {
        const int la = e / ACC_BW, lb = la + e % ACC_BW;
        const int pa = kpos[la], o = kown[la / 6];
        if (o == wg || pa == LVX_DEAD || lb >= ACC_LV) continue;
        const double v = A.band[e];
        if (v == 0.0) continue;
        if (o >= 0) { atomicOr(cm.err, LVX_ERR_FALLBACK); continue; }   // another workgroup STORES this column: cannot happen while |tau_imu| < dt
        const int pb = kpos[lb];
        if (pb != LVX_DEAD) add_H(cm, pa, pb, v, rep);
      }
      for (int gi = 0; gi < IMU_NGA; ++gi)
        for (int la = r0 + bt; la < r1; la += 256) {
          const int pg = gpos[gi], pa = kpos[la], o = kown[la / 6];
          if (pg == LVX_DEAD || pa == LVX_DEAD || (o == wg && pg < 0)) continue;
          const double v = A.bd[gi * ACC_LV + la];
          if (v == 0.0) continue;
          if (o >= 0) atomicOr(cm.err, LVX_ERR_FALLBACK); else add_H(cm, pg, pa, v, rep);
        }
      if (bt < nfl && kpos[bt] != LVX_DEAD && kown[bt / 6] != wg) {
        const double v = A.gk[bt];
        if (v != 0.0) { if (kown[bt / 6] >= 0) atomicOr(cm.err, LVX_ERR_FALLBACK); else add_g(cm, kpos[bt], v, rep); }
      }
      LVX_LDS_BARRIER();   // the register pass below zeroes what it reads
      if (tid == 0) { urng[0] = 1 << 30; urng[1] = -1; }
    }
    IKT(6)
    {
      // Every thread takes its share of the accumulators into registers (band: pairs of entries, 16 bytes) together with what it needs to route them, and leaves
      // ZERO behind; ONE barrier; then each value goes where it belongs: a finished column to HBM (stored by its owner), a live one down by nfl rows in LDS.  The
      // phase is a chain of dependent LDS reads and branches on one wavefront per SIMD: table reads hoisted in front of the barrier, routing by predication.
      constexpr int NQB = (ACC_LV * HB + 255) / 256, NQD = (IMU_NGA * ACC_LV + 255) / 256;
      double2* band2 = (double2*)A.band;
      double2 vb[NQB]; int pb_[NQB], ob_[NQB];
#pragma unroll
      for (int q = 0; q < NQB; ++q) {
        const int e2 = bt + 256 * q;
        const bool inr = e2 < ACC_LV * HB;
        const int la = inr ? e2 / HB : 0;
        vb[q] = inr ? band2[e2] : make_double2(0.0, 0.0);
        pb_[q] = kpos[la]; ob_[q] = kown[la / 6];
        if (inr && more) band2[e2] = make_double2(0.0, 0.0);
      }
      double vd[NQD], vg = 0.0; int pd_[NQD], od_[NQD], pg_ = LVX_DEAD, og_ = -1;
#pragma unroll
      for (int q = 0; q < NQD; ++q) {
        const int e = bt + 256 * q;
        const bool inr = e < IMU_NGA * ACC_LV;
        const int la = inr ? e % ACC_LV : 0;
        vd[q] = inr ? A.bd[e] : 0.0;
        pd_[q] = kpos[la]; od_[q] = kown[la / 6];
        if (inr && more) A.bd[e] = 0.0;
      }
      if (bt < ACC_LV) { vg = A.gk[bt]; pg_ = kpos[bt]; og_ = kown[bt / 6]; if (more) A.gk[bt] = 0.0; }
      int gp_[NQD];
#pragma unroll
      for (int q = 0; q < NQD; ++q) { const int e = bt + 256 * q; gp_[q] = gpos[e < IMU_NGA * ACC_LV ? e / ACC_LV : 0]; }
      LVX_LDS_BARRIER();
      IKT(7)
#pragma unroll
      for (int q = 0; q < NQB; ++q) {
        const int e2 = bt + 256 * q;
        const int la = e2 / HB, d = 2 * (e2 % HB);
        const bool inr = e2 < ACC_LV * HB;
        if (inr && la >= nfl) band2[e2 - nfl * HB] = vb[q];
        if (inr && la < nfl && ob_[q] == wg) {
          double* dst = cm.Hb + (size_t)pb_[q] * ld + d;
          if (d + 1 < ld && !(((size_t)pb_[q] * ld + d) & 1)) *(double2*)dst = vb[q];
          else { if (d < ld) dst[0] = vb[q].x; if (d + 1 < ld) dst[1] = vb[q].y; }
        }
      }
      IKT(8)
#pragma unroll
      for (int q = 0; q < NQD; ++q) {
        const int e = bt + 256 * q;
        const int la = e % ACC_LV;
        const bool inr = e < IMU_NGA * ACC_LV;
        if (inr && la >= nfl) A.bd[e - nfl] = vd[q];
        if (inr && la < nfl && od_[q] == wg && gp_[q] < 0 && gp_[q] != LVX_DEAD) cm.Bd[(size_t)(-1 - gp_[q]) * cm.nb + pd_[q]] = vd[q];
      }
      if (bt < ACC_LV) {
        if (bt >= nfl) A.gk[bt - nfl] = vg;
        else if (og_ == wg) cm.gb[pg_] = vg;
      }
      IKT(9)
    }
    IKT(5)
  }
#ifdef LVX_KTIME_IMU
  if (lane == 0 && (wg == 5 || wg == 100)) printf("IKT wg %d wv %d batches %d: sync0 %lld eval %lld G %lld A %lld sync %lld | slow %lld read+bar %lld band %lld bd+gk %lld zero %lld | total %lld\n", wg, wv, c1 - c0, kt_[0], kt_[1], kt_[2], kt_[3], kt_[4], kt_[6], kt_[7], kt_[8], kt_[9], kt_[5],
                                                   (long long)__builtin_amdgcn_s_memtime() - kts_);
#endif
  mycost = wave_sum(mycost);
  if (lane == 0) atomicAdd(&cm.cost[rep], mycost);
  if (!want_ne) return;
  for (int e = tid; e < IMU_NGA * IMU_NGA; e += 256) {
    const int ga = e / IMU_NGA, gb2 = e % IMU_NGA;
    if (gb2 < ga) continue;
    const double v = A.gg[e];
    if (v == 0.0 || gpos[ga] == LVX_DEAD || gpos[gb2] == LVX_DEAD) continue;
    add_H(cm, gpos[ga], gpos[gb2], v, rep);
  }
  if (tid < IMU_NGA) { const double v = A.gG[tid]; if (v != 0.0 && gpos[tid] != LVX_DEAD) add_g(cm, gpos[tid], v, rep); }
}

// fold the pseudo-pose rows of the border back onto the hub control points: x_pseudo = M_hub x_hub  =>
//   Bd[hub] += M^T Bd[pseudo],  C <- (I + E) C (I + E)^T,  g_c[hub] += M^T g_c[pseudo]      (E = M^T placed at [hub rows, pseudo cols])
// store: this set is the first to fold in this pass — outside [hub_lo, hub_hi) the hub rows were NOT cleared (nothing but the fold writes
// there: 36 MB of the per-pass clear at config 4), so it stores its sums (or zero) instead of adding
__device__ __forceinline__ void fold_border_rows_block(const DevCommon& cm, int set, int blk, bool store) {
  const HubShared* hub = ((const HubShared*)cm.hubs) + set;
  if (hub->ok != 1) return;
  __shared__ double M[6][24];
  __shared__ int hrow[24];
  if (threadIdx.x < 144) M[threadIdx.x / 24][threadIdx.x % 24] = hub->M[threadIdx.x / 24][threadIdx.x % 24];
  if (threadIdx.x < 24) { const int o = cm.ord[6 * (hub->A.k.i0 + threadIdx.x / 6) + threadIdx.x % 6]; hrow[threadIdx.x] = (o != LVX_DEAD && o < 0) ? -1 - o : -1; }   // hub control points are border variables
  __syncthreads();
  const int j = blk * blockDim.x + threadIdx.x;
  double P[6];
  bool any = false;
  if (j < cm.nb) {
#pragma unroll
    for (int p = 0; p < 6; ++p) { P[p] = cm.Bd[(size_t)(cm.nbd_solve + 6 * set + p) * cm.nb + j]; any = any || P[p] != 0.0; }
  }
  const bool far = store && j < cm.nb && (j < cm.hub_lo || j >= cm.hub_hi);
  if (any || far) {
#pragma unroll
    for (int c = 0; c < 24; ++c) {
      if (hrow[c] < 0) continue;
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < 6; ++p) s += M[p][c] * P[p];
      double* dst = &cm.Bd[(size_t)hrow[c] * cm.nb + j];
      if (far) *dst = any ? s : 0.0; else *dst += s;   // the two sets may share hub rows: a thread folds them one after the other (k_fold_all)
    }
  }
  __syncthreads();   // M / hrow are reused by the next set
}
__global__ __launch_bounds__(256) void k_fold_border_rows(DevCommon cm, int set, int store) { fold_border_rows_block(cm, set, (int)blockIdx.x, store != 0); }
__device__ __forceinline__ void fold_border_dense_block(const DevCommon& cm) {
  extern __shared__ double Cf[];   // full symmetric [n][n] then g[n]
  const int n = cm.nbd;
  double* g = Cf + n * n;
  // one workgroup, latency bound: issue a batch of 8 independent loads per thread before the first LDS store
  for (int e0 = threadIdx.x; e0 < n * n; e0 += 8 * 256) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int e = e0 + 256 * q; if (e < n * n) { const int a = e / n, b = e % n; v[q] = a >= b ? cm.C[(size_t)a * n + b] : cm.C[(size_t)b * n + a]; } }
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int e = e0 + 256 * q; if (e < n * n) Cf[e] = v[q]; }
  }
  for (int e = threadIdx.x; e < n; e += 256) g[e] = cm.gc[e];
  __syncthreads();
  for (int set = 0; set < 2; ++set) {
    const HubShared* hub = ((const HubShared*)cm.hubs) + set;
    if (hub->ok != 1) continue;   // uniform
    __shared__ double M[6][24];
    __shared__ int hrow[24];
    if (threadIdx.x < 144) M[threadIdx.x / 24][threadIdx.x % 24] = hub->M[threadIdx.x / 24][threadIdx.x % 24];
    if (threadIdx.x < 24) { const int o = cm.ord[6 * (hub->A.k.i0 + threadIdx.x / 6) + threadIdx.x % 6]; hrow[threadIdx.x] = (o != LVX_DEAD && o < 0) ? -1 - o : -1; }
    __syncthreads();
    const int p0 = cm.nbd_solve + 6 * set;
    // rows: C[hub_c][:] += sum_p M[p][c] C[p0+p][:]
    for (int e = threadIdx.x; e < 24 * n; e += 256) {
      const int c = e / n, col = e % n;
      if (hrow[c] < 0) continue;
      double s = 0.0;
      for (int p = 0; p < 6; ++p) s += M[p][c] * Cf[(p0 + p) * n + col];
      Cf[hrow[c] * n + col] += s;   // distinct (c, col) -> distinct destination, pseudo rows are read-only here
    }
    if (threadIdx.x < 24 && hrow[threadIdx.x] >= 0) { double s = 0.0; for (int p = 0; p < 6; ++p) s += M[p][threadIdx.x] * g[p0 + p]; g[hrow[threadIdx.x]] += s; }
    __syncthreads();
    // columns: C[:][hub_c] += sum_p C[:][p0+p] M[p][c]
    for (int e = threadIdx.x; e < 24 * n; e += 256) {
      const int c = e / n, row = e % n;
      if (hrow[c] < 0) continue;
      double s = 0.0;
      for (int p = 0; p < 6; ++p) s += Cf[row * n + p0 + p] * M[p][c];
      Cf[row * n + hrow[c]] += s;
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < n * n; e += 256) { const int a = e / n, b = e % n; if (a >= b) cm.C[(size_t)a * n + b] = Cf[e]; }
  for (int e = threadIdx.x; e < n; e += 256) cm.gc[e] = g[e];
}

__global__ __launch_bounds__(256) void k_fold_border_dense(DevCommon cm) { fold_border_dense_block(cm); }

// fold the replicas of the dense border accumulators into replica 0
// sum of n values with stride `stride` in index order, 16 loads in flight (the replica count is a run-time value since deterministic mode: a plain
// loop became a chain of dependent loads, 47 us instead of 7)
__device__ __forceinline__ double strided_sum(const double* p, size_t stride, int n) {
  double s = 0.0;
  for (int r0 = 0; r0 < n; r0 += 16) {
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = r0 + q < n ? p[(size_t)(r0 + q) * stride] : 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += v[q];
  }
  return s;
}
__device__ __forceinline__ void fold_replicas_block(const DevCommon& cm, int blk) {
  const int n2 = cm.nbd * cm.nbd;
  const int i = blk * blockDim.x + threadIdx.x;
  if (cm.what & LVX_EVAL_NORMAL_EQ) {
    if (i < n2) cm.C[i] = strided_sum(cm.C + i, (size_t)n2, cm.nrep);
    if (i < cm.nbd) cm.gc[i] = strided_sum(cm.gc + i, (size_t)cm.nbd, cm.nrep);
  }
  if (i == 0) cm.cost[0] = strided_sum(cm.cost, 1, cm.nrep);
}
__global__ void k_fold_replicas(DevCommon cm) { fold_replicas_block(cm, (int)blockIdx.x); }
// The whole fold in ONE launch (round 5b): blocks [0, n_rep) sum the replicas of the dense border accumulators, and the LAST of them to finish (a ticket) folds the dense
// block; blocks behind them fold the border rows of set `set0` (then of `set1` if >= 0).  Three launches on two streams before — replicas 10 us -> dense 17 us on the chain,
// border rows 17 us beside them, plus the fork / join packets around the side stream.
__global__ __launch_bounds__(256) void k_fold_all(DevCommon cm, int n_rep, int n_rows, int set0, int store0, int set1, int store1, int* ticket) {
  __shared__ int s_last;
  const int b = blockIdx.x;
  if (b < n_rep) {
    fold_replicas_block(cm, b);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) { const int t = atomicAdd(ticket, 1); s_last = t == n_rep - 1; if (s_last) *ticket = 0; }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    fold_border_dense_block(cm);
    return;
  }
  const int r = b - n_rep;
  fold_border_rows_block(cm, set0, r, store0 != 0);
  if (set1 >= 0) fold_border_rows_block(cm, set1, r, store1 != 0);   // the two sets may share hub rows: the same thread folds them one after the other
}
// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
int fail(lvx_ctx* ctx, int code, const std::string& msg) { if (ctx) ctx->last_error = msg; return code; }

int dev_alloc(lvx_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes == 0) bytes = 8;
  if (b.p && b.bytes >= bytes) return LVX_OK;
  if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
  if (hipMalloc(&b.p, bytes) != hipSuccess) { b.p = nullptr; return fail(ctx, LVX_E_ALLOC, "hipMalloc failed"); }
  b.bytes = bytes;
  return LVX_OK;
}
int upload(lvx_ctx* ctx, DevBuf& b, const void* src, size_t bytes) {
  int rc = dev_alloc(ctx, b, bytes);
  if (rc) return rc;
  if (bytes) LVX_HIP(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return LVX_OK;
}
// layout-time uploads come from temporaries (sorted copies, tables) that die right after the call: a copy from pageable memory may still be
// reading its source when hipMemcpyAsync returns (large copies are pinned in place and DMA'd), so these wait for it
int upload_tmp(lvx_ctx* ctx, DevBuf& b, const void* src, size_t bytes) {
  int rc = upload(ctx, b, src, bytes);
  if (rc) return rc;
  LVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LVX_OK;
}

namespace {

// host loops over a million measurements (keys, gathers, row-ordered plane copies) on a few threads: the layout of a config-4 stage is host time between two solves
template <class F> void par_for(size_t n, F&& fn) {
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t nt = n < 200000 ? 1 : std::min<size_t>(8, hw ? hw : 1);
  if (nt <= 1) { fn((size_t)0, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n + nt - 1) / nt;
  for (size_t t = 1; t < nt; ++t) { const size_t a = std::min(n, t * per), b = std::min(n, a + per); if (a < b) th.emplace_back([&fn, a, b] { fn(a, b); }); }
  fn((size_t)0, std::min(n, per));
  for (auto& x : th) x.join();
}
template <class T> std::vector<T> gather(const std::vector<T>& v, const std::vector<int>& perm, int width) {
  std::vector<T> o(perm.size() * width);
  par_for(perm.size(), [&](size_t a, size_t b) { for (size_t i = a; i < b; ++i) for (int k = 0; k < width; ++k) o[i * width + k] = v[(size_t)perm[i] * width + k]; });
  return o;
}

// Pair table of a family's local columns, ORDERED FOR COALESCED ATOMICS: consecutive lanes of phase 2 take consecutive pairs, so
// the order decides the address pattern (measured on MI355X: coalesced f64 atomics 140-180 G/s, scattered 23 G/s).
// cls[c] = 0: band variable (knot / landmark), 1: border variable.  (band, band): a-major -> runs along the band column;
// (band, border): border-major -> runs along a border row; (border, border) last.
int upload_pairs(lvx_ctx* ctx, DevBuf& b, int NC, const std::vector<int>& border_cols = {}) {
  std::vector<int> cls(NC, 0);
  for (int c : border_cols) if (c >= 0 && c < NC) cls[c] = 1;
  std::vector<uint16_t> tab;
  for (int a = 0; a < NC; ++a) for (int c = a; c < NC; ++c) if (!cls[a] && !cls[c]) tab.push_back((uint16_t)(a | (c << 8)));
  for (int g = 0; g < NC; ++g) if (cls[g]) for (int k = 0; k < NC; ++k) if (!cls[k]) tab.push_back((uint16_t)(std::min(g, k) | (std::max(g, k) << 8)));
  for (int a = 0; a < NC; ++a) for (int c = a; c < NC; ++c) if (cls[a] && cls[c]) tab.push_back((uint16_t)(a | (c << 8)));
  return upload_tmp(ctx, b, tab.data(), tab.size() * sizeof(uint16_t));
}

// host-side knot index of a single-time lookup (tau = 0), -1 if out of range
int host_i0(const lvx_ctx* c, double t) {
  KnotRef k;
  if (!knot_lookup(c->t0, c->dt, c->N, t, t, &k)) return -1;
  return k.i0;
}

}  // namespace

// Knot intervals per workgroup of the MFMA assembly kernels: every workgroup has the same expected work (~ R intervals), the launch runs in
// ceil(workgroups / CUs) rounds, so pick R in [lo, hi] that minimises rounds * R (e.g. 25 k intervals on 256 CUs: R = 20 -> 1252 workgroups =
// 4.9 rounds instead of 6.1 half-empty ones at R = 16; families that run two workgroups per CU count 2 slots per CU).  LVX_CHUNK_R / LVX_CHUNK_R_REP / LVX_CHUNK_R_IMU (env) force a value.
// (Measured against a finer work model — fixed cost + cost per interval + cost per batch of 4 x LB rows: the plain rule picks the faster sizes,
// e.g. IMU R = 33 (3 rounds, 264 rows = a second batch for 8 rows) beats R = 25 (4 rounds of one batch) by 7 %.)
// fixed: a workgroup's setup + accumulator flush in units of one interval's work (LiDAR kernel: 19 k of its cycles against 8.6 k per interval
// => 2.2; with it 13 intervals = 3.8 rounds beat 10 = 4.9 rounds, measured 0.232 vs 0.241 ms)
static int pick_chunk(const lvx_ctx* ctx, int lo, int hi, int forced, int wg_per_cu = 1, double fixed = 0.0) {
  if (forced >= 4 && forced <= 64) return forced;
  int ncu = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
  int best = lo; long long best_cost = -1;
  for (int r = lo; r <= hi; ++r) {
    const long long nwg = (ctx->N + r - 1) / r + 1;
    const long long slots = (long long)ncu * wg_per_cu;
    const long long cost = (long long)(((nwg + slots - 1) / slots) * (r + fixed) * 64.0);
    if (best_cost < 0 || cost < best_cost) { best = r; best_cost = cost; }
  }
  return best;
}
// Families whose kernels run NEXT TO other kernels (second stage of a pass) are not served by whole rounds of their own launch but by full
// batches: a workgroup processes its rows in batches of 4 wavefronts x LB rows, and a chunk of 33 intervals x 8 IMU samples = 264 rows pays a
// whole second batch for 8 rows (measured: R = 32 instead of 33 takes 3 % off the pass although the kernels alone get slower).  Largest R
// whose expected rows fill nb batches to >= 90 %, smallest nb first.
static int pick_chunk_batches(int lo, int hi, int forced, double rows_per_interval, int rows_per_batch) {
  if (forced >= 4 && forced <= 64) return forced;
  if (!(rows_per_interval > 0.0)) return lo;
  int best = lo; double best_eff = 0.0;
  for (int nb = 1; nb <= 4; ++nb) {
    const int r = std::min(hi, (int)std::floor(rows_per_batch * nb / rows_per_interval));
    if (r < lo) continue;
    const double eff = rows_per_interval * r / (rows_per_batch * (double)nb);
    if (eff >= 0.9) return r;
    if (eff > best_eff) { best_eff = eff; best = r; }
  }
  return best;
}
// Chunks of equal ROW count for the dense LiDAR families: a workgroup processes its rows in batches of 256 (4 wavefronts x 64), and chunks of
// R intervals hold 40 R +- a few rows — e.g. 527 at R = 13: a third batch for 15 rows on one wavefront while three wait (15 % of the kernel).
// Here every chunk has exactly `rows` rows (the last one fewer); a chunk may start or end inside an interval (the accumulators are
// added to HBM atomically anyway) and spans at most rmax intervals (LDS accumulator size).
static int upload_chunks_rows(lvx_ctx* ctx, int fam, const std::vector<int>& sorted_keys, int rmax, int rows) {
  const int n = (int)sorted_keys.size();
  std::vector<int> off, k0;
  int i = 0;
  do {
    off.push_back(i);
    const int kf = n > 0 ? std::max(0, sorted_keys[std::min(i, n - 1)]) : 0;
    k0.push_back(kf);
    int e = std::min(n, i + rows);
    while (e > i + 1 && sorted_keys[e - 1] >= kf + rmax) --e;   // span limit (always at least one row)
    i = std::max(e, i + (n > 0 ? 1 : 0));
  } while (i < n);
  off.push_back(n);
  const int nch = (int)k0.size();
  ctx->n_chunk[fam] = nch; ctx->chunk_r[fam] = rmax; ctx->chunk_var[fam] = 1;
  ctx->h_chunk_k0[fam] = k0; ctx->h_chunk_rows[fam].assign(nch, 0); for (int c = 0; c < nch; ++c) ctx->h_chunk_rows[fam][c] = off[c + 1] - off[c];
  std::vector<int> span(nch, 1);   // knot intervals a chunk's rows cover (k_family_mfma clears / flushes that many accumulator rows)
  for (int c = 0; c < nch; ++c) if (off[c + 1] > off[c]) span[c] = std::max(1, sorted_keys[off[c + 1] - 1] - k0[c] + 1);
  off.insert(off.end(), k0.begin(), k0.end());
  off.insert(off.end(), span.begin(), span.end());
  return upload_tmp(ctx, ctx->d_chunk[fam], off.data(), off.size() * 4);
}
// Same for rows whose keys are only grouped, not sorted (the reprojection passes share one row order): a chunk is a run of at most `rows` rows
// whose keys span at most rmax intervals; its first interval is the smallest key of the run.
static int upload_chunks_rows_grouped(lvx_ctx* ctx, int fam, const std::vector<int>& keys, int rmax, int rows) {
  const int n = (int)keys.size();
  std::vector<int> off, k0;
  int i = 0;
  std::vector<int> span;
  do {
    off.push_back(i);
    int lo = 1 << 30, hi = -1, e = i;
    while (e < n && e - i < rows) {
      const int k = keys[e];
      if (k >= 0) { const int nlo = std::min(lo, k), nhi = std::max(hi, k); if (nhi - nlo + 1 > rmax && e > i) break; lo = nlo; hi = nhi; }
      ++e;
    }
    k0.push_back(hi >= 0 ? lo : 0); span.push_back(hi >= 0 ? hi - lo + 1 : 1);
    i = std::max(e, i + (n > 0 ? 1 : 0));
  } while (i < n);
  off.push_back(n);
  // the kernel's accumulator window (and its LDS footprint, hence how many of these latency-bound workgroups a CU holds) follows the widest chunk that EXISTS, not the
  // widest one allowed: a camera frame spans ~4 intervals, rmax is 48 (100 KB of LDS = one workgroup per CU)
  int smax = 4; for (int v : span) smax = std::max(smax, v);
  ctx->n_chunk[fam] = (int)k0.size(); ctx->chunk_r[fam] = std::min(rmax, smax); ctx->chunk_var[fam] = 1;
  ctx->h_chunk_k0[fam] = k0; ctx->h_chunk_rows[fam].assign(k0.size(), 0); for (size_t c = 0; c < k0.size(); ++c) ctx->h_chunk_rows[fam][c] = off[c + 1] - off[c];
  off.insert(off.end(), k0.begin(), k0.end());
  off.insert(off.end(), span.begin(), span.end());
  return upload_tmp(ctx, ctx->d_chunk[fam], off.data(), off.size() * 4);
}
static int upload_chunks(lvx_ctx* ctx, int fam, const std::vector<int>& sorted_keys, int R) {
  const int nch = (ctx->N + R - 1) / R + 1;
  std::vector<int> off(nch + 1);
  for (int c = 0; c <= nch; ++c)
    off[c] = c == 0 ? 0 : (int)(std::lower_bound(sorted_keys.begin(), sorted_keys.end(), c * R) - sorted_keys.begin());
  off[nch] = (int)sorted_keys.size();
  ctx->n_chunk[fam] = nch;
  ctx->chunk_r[fam] = R; ctx->chunk_var[fam] = 0;
  ctx->h_chunk_k0[fam].assign(nch, 0); ctx->h_chunk_rows[fam].assign(nch, 0); for (int c = 0; c < nch; ++c) { ctx->h_chunk_k0[fam][c] = c * R; ctx->h_chunk_rows[fam][c] = off[c + 1] - off[c]; }
  return upload_tmp(ctx, ctx->d_chunk[fam], off.data(), off.size() * 4);
}

// stable permutation that sorts small integer keys (knot intervals, -1 = out of range): a counting sort — std::stable_sort over 1 M indirect keys was a large part of
// the 54 ms a config-4 layout took on the host
static void stable_perm_by_key(const std::vector<int>& key, std::vector<int>& perm) {
  const size_t n = key.size();
  perm.resize(n);
  if (n == 0) return;
  int lo = key[0], hi = key[0];
  for (size_t i = 1; i < n; ++i) { lo = std::min(lo, key[i]); hi = std::max(hi, key[i]); }
  if ((long long)hi - lo > (long long)(4 * n + 1024)) {   // (sparse keys: comparison sort)
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return key[a] < key[b]; });
    return;
  }
  std::vector<int> cnt((size_t)(hi - lo) + 2, 0);
  for (size_t i = 0; i < n; ++i) cnt[(size_t)(key[i] - lo) + 1]++;
  for (size_t k = 1; k < cnt.size(); ++k) cnt[k] += cnt[k - 1];
  for (size_t i = 0; i < n; ++i) perm[(size_t)cnt[(size_t)(key[i] - lo)]++] = (int)i;
}
int ensure_layout(lvx_ctx* ctx) {
  if (!ctx->layout_dirty) return LVX_OK;
  if (!ctx->have_spline) return fail(ctx, LVX_E_STATE, "lvx_set_spline has not been called");
  const int N = ctx->N, L = ctx->L;
  const uint32_t locks = ctx->locks;
  int rc;
  {   // the plane / landmark tables may have been replaced (shrunk) after the families that index them were set
    const long long npl = (long long)ctx->planes.size() / 3;
    for (int i = 0; i < ctx->surf.n; ++i) if (ctx->surf.id0[i] < 0 || ctx->surf.id0[i] >= npl) return fail(ctx, LVX_E_ARG, "surfel plane id out of range for the current plane table");
    for (int i = 0; i < ctx->cs.n; ++i) {
      if (ctx->cs.id1[i] < 0 || ctx->cs.id1[i] >= npl) return fail(ctx, LVX_E_ARG, "cam-surfel plane id out of range for the current plane table");
      if (ctx->cs.id0[i] < 0 || ctx->cs.id0[i] >= L) return fail(ctx, LVX_E_ARG, "cam-surfel landmark id out of range for the current landmark table");
    }
  }
  const SplineRef sp{ctx->t0, ctx->dt, N, nullptr, nullptr};
  // ---- sort every family by knot interval and upload ----
  {
    Family& f = ctx->imu;
    std::vector<int> key(f.n), perm(f.n);
    par_for((size_t)f.n, [&](size_t a, size_t b) { for (size_t i = a; i < b; ++i) key[i] = host_i0(ctx, f.t[i]); });
    stable_perm_by_key(key, perm);
    { std::vector<int> sk(f.n); for (int i = 0; i < f.n; ++i) sk[i] = key[perm[i]];
      if ((rc = upload_chunks(ctx, LVX_FAM_GYRO, sk, pick_chunk_batches(16, 64, ctx->sw.chunk_r_imu, (double)f.n / std::max(1, N - 3), 4 * (int)GyroAcc::LB)))) return rc;   // the gyroscope-only kernel of Solve #0
      // owner-computes schedule (k_imu_own): one workgroup per CU, equal row counts, boundaries on knot intervals; inside a workgroup batches of <= 256 rows spanning <= IMU_CR
      // intervals.  The workgroup keeps the tables of its whole knot range in LDS: more (smaller) ranges when the widest one does not fit (sparse sample streams).
      int ncu = 256;
      { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
      const int cr = IMU_CR;
      std::vector<int> off, k0, wg_c0;
      int G = std::max(1, std::min(ncu, (f.n + 255) / 256)), span = 0;
      for (;; G *= 2) {
        off.clear(); k0.clear(); wg_c0.clear(); span = cr + 5;
        int r = 0;
        for (int g = 0; g < G; ++g) {
          wg_c0.push_back((int)k0.size());
          int r1 = g + 1 == G ? f.n : (int)((long long)f.n * (g + 1) / G);
          while (r1 < f.n && r1 > 0 && sk[r1] == sk[r1 - 1]) ++r1;   // next interval start
          r1 = std::max(r1, r);
          const size_t first = k0.size();
          for (int i = r; i < r1;) {
            const int kf = std::max(0, sk[i]);
            int e = std::min(r1, i + 256);
            while (e > i + 1 && sk[e - 1] >= kf + cr) --e;
            off.push_back(i); k0.push_back(kf);
            i = e;
          }
          if (k0.size() > first) span = std::max(span, k0.back() - k0[first] + cr + 5);
          r = r1;
        }
        wg_c0.push_back((int)k0.size());
        if (imu_fused_lds_bytes(span) <= 160 * 1024 || G >= std::max(1, f.n)) break;
      }
      ctx->imu_wg = G; ctx->imu_nch = (int)k0.size(); ctx->imu_span = span; ctx->imu_h_k0 = k0; ctx->imu_h_wg_c0 = wg_c0;
      off.push_back(f.n);
      off.insert(off.end(), k0.begin(), k0.end());
      if ((rc = upload_tmp(ctx, ctx->d_imu_chunk, off.data(), off.size() * 4))) return rc;
      if ((rc = upload_tmp(ctx, ctx->d_imu_wg, wg_c0.data(), wg_c0.size() * 4))) return rc;
    }
    auto ts = gather(f.t, perm, 1); auto g = gather(f.a3, perm, 3); auto a = gather(f.b3, perm, 3);
    if ((rc = upload_tmp(ctx, f.d_t, ts.data(), ts.size() * 8))) return rc;
    if ((rc = upload_tmp(ctx, f.d_a3, g.data(), g.size() * 8))) return rc;
    if ((rc = upload_tmp(ctx, f.d_b3, a.data(), a.size() * 8))) return rc;
    if ((rc = upload_tmp(ctx, f.d_perm, perm.data(), perm.size() * 4))) return rc;
  }
  {
    Family& f = ctx->surf;
    std::vector<int> key(f.n), perm(f.n);
    par_for((size_t)f.n, [&](size_t a, size_t b) { for (size_t i = a; i < b; ++i) key[i] = host_i0(ctx, f.t[i]); });
    stable_perm_by_key(key, perm);
    { std::vector<int> sk(f.n); for (int i = 0; i < f.n; ++i) sk[i] = key[perm[i]]; if (ctx->sw.chunk_r) rc = upload_chunks(ctx, LVX_FAM_SURFEL, sk, pick_chunk(ctx, 8, 16, ctx->sw.chunk_r, 2, 2.2));
      else rc = upload_chunks_rows(ctx, LVX_FAM_SURFEL, sk, 16, ctx->sw.chunk_rows > 0 ? ctx->sw.chunk_rows : 512);
      if (rc) return rc; }
    auto ts = gather(f.t, perm, 1); auto pt = gather(f.a3, perm, 3); auto pl = gather(f.id0, perm, 1);
    if ((rc = upload_tmp(ctx, f.d_t, ts.data(), ts.size() * 8))) return rc;
    if ((rc = upload_tmp(ctx, f.d_a3, pt.data(), pt.size() * 8))) return rc;
    if ((rc = upload_tmp(ctx, f.d_id0, pl.data(), pl.size() * 4))) return rc;
    if ((rc = upload_tmp(ctx, f.d_perm, perm.data(), perm.size() * 4))) return rc;
    // the fused kernel reads each row's plane from a row-ordered copy (planes are inputs, fixed between layouts): no dependent gather
    std::vector<double> rowpl((size_t)f.n * 3, 0.0);
    const long long npl = (long long)ctx->planes.size() / 3;
    par_for((size_t)f.n, [&](size_t a, size_t b) { for (size_t i = a; i < b; ++i) if (pl[i] >= 0 && pl[i] < npl) for (int c = 0; c < 3; ++c) rowpl[3 * i + c] = ctx->planes[3 * (size_t)pl[i] + c]; });
    if ((rc = upload_tmp(ctx, f.d_b3, rowpl.data(), rowpl.size() * 8))) return rc;
  }
  {
    Family& f = ctx->rep;
    // One device order for every reprojection pass (see RepSideAcc / k_reproj_cross): (observation window, reference window, observation
    // interval, reference interval, landmark), a window being 4 aligned knot intervals.  Rolling shutter: the evaluation time is
    // t0 + v * readout / rows.  The per-segment kernel uses the same arrays.
    const double row_delta = ctx->cam.rows > 0 ? ctx->cam.readout / (double)ctx->cam.rows : 0.0;
    std::vector<int> k1(f.n), k0(f.n), perm(f.n);
    for (int i = 0; i < f.n; ++i) {
      const int l = f.id0[i];
      k1[i] = host_i0(ctx, f.t[i] + f.a3[2 * (size_t)i + 1] * row_delta);
      k0[i] = (l >= 0 && l < L) ? host_i0(ctx, ctx->lm_t0[l] + ctx->lm_uv[2 * (size_t)l + 1] * row_delta) : -1;
      if (k0[i] < 0 || k1[i] < 0) k0[i] = k1[i] = -1;   // out of range: reported by the Jacobian kernel, skipped by the assembly
    }
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) {
      const int wa1 = k1[a] >> 2, wb1 = k1[b] >> 2, wa0 = k0[a] >> 2, wb0 = k0[b] >> 2;
      if (wa1 != wb1) return wa1 < wb1;
      if (wa0 != wb0) return wa0 < wb0;
      if (k1[a] != k1[b]) return k1[a] < k1[b];
      if (k0[a] != k0[b]) return k0[a] < k0[b];
      return f.id0[a] < f.id0[b];
    });
    {
      std::vector<int> s1(f.n), s0(f.n);
      for (int i = 0; i < f.n; ++i) { s1[i] = k1[perm[i]]; s0[i] = k0[perm[i]]; }
      const int rows = ctx->sw.rep_rows > 0 ? ctx->sw.rep_rows : 16 * (int)RepSideAcc<1>::LB;   // 256 rows = four batches of 4 wavefronts x LB = 16 rows (measured: 64 rows 73 + 88 us, 256 rows 52 + 62 us)
      const int rmax = (ctx->sw.chunk_r_rep >= 4 && ctx->sw.chunk_r_rep <= 48) ? ctx->sw.chunk_r_rep : 48;
      if ((rc = upload_chunks_rows_grouped(ctx, LVX_FAM_REPROJ, s1, rmax, rows))) return rc;
      if ((rc = upload_chunks_rows_grouped(ctx, LVX_FAM_PRIOR /* slot reused: the prior has no chunks */, s0, rmax, rows))) return rc;
      // groups of the cross-term kernel: runs of equal (reference window, observation window); rows that are out of range form no group
      std::vector<int> goff, gw0, gw1;
      const int gcap = 32;   // measured round 4 (cross kernel solo / pass): 8: 78 us / 0.606 ms, 16: 57 / 0.595, 24: 57 / 0.584, 32: 59 / 0.582, 64: 52 / 0.604, 128: 82 / 0.633 — longer groups serialise a wavefront, shorter ones add atomics
      for (int i = 0; i < f.n; ++i) {
        if (s1[i] < 0) continue;
        const int a0 = s0[i] >> 2, a1 = s1[i] >> 2;
        // a wavefront walks its group 8 blocks at a time: at most 32 blocks per group, a larger (window, window) pair is shared by several
        if (goff.empty() || gw0.back() != a0 || gw1.back() != a1 || i - goff.back() >= gcap) { goff.push_back(i); gw0.push_back(a0); gw1.push_back(a1); }
      }
      {   // blocks of every landmark (k_reproj_lmrows), positions in this row order
        std::vector<int> ptr((size_t)L + 1, 0), rows((size_t)std::max(f.n, 1));
        for (int i = 0; i < f.n; ++i) { const int l = f.id0[perm[i]]; if (l >= 0 && l < L) ptr[l + 1]++; }
        for (int l = 0; l < L; ++l) ptr[l + 1] += ptr[l];
        std::vector<int> fill(ptr.begin(), ptr.end() - 1);
        for (int i = 0; i < f.n; ++i) { const int l = f.id0[perm[i]]; if (l >= 0 && l < L) rows[fill[l]++] = i; }
        ptr.insert(ptr.end(), rows.begin(), rows.end());
        if ((rc = upload_tmp(ctx, ctx->d_repB[3], ptr.data(), ptr.size() * 4))) return rc;
      }
      ctx->det_cross_col.clear();
      if (ctx->sw.deterministic && !goff.empty()) {
        // groups that can add to the same entry must not share a launch: a group touches the unordered pairs of 4-knot blocks {w0, w0 + 1} x {w1, w1 + 1}
        std::vector<std::vector<uint64_t>> used;   // per colour: sorted block-pair keys
        std::vector<std::vector<int>> members;
        for (size_t g = 0; g < goff.size(); ++g) {
          uint64_t key[4]; int nk = 0;
          for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) { const uint64_t x = (uint64_t)(gw0[g] + a), y = (uint64_t)(gw1[g] + b); key[nk++] = x < y ? (x << 32) | y : (y << 32) | x; }
          size_t col = 0;
          for (; col < used.size(); ++col) { bool hit = false; for (int q = 0; q < 4 && !hit; ++q) hit = std::binary_search(used[col].begin(), used[col].end(), key[q]); if (!hit) break; }
          if (col == used.size()) { used.emplace_back(); members.emplace_back(); }
          for (int q = 0; q < 4; ++q) { auto it = std::lower_bound(used[col].begin(), used[col].end(), key[q]); if (it == used[col].end() || *it != key[q]) used[col].insert(it, key[q]); }
          members[col].push_back((int)g);
        }
        std::vector<int> list; ctx->det_cross_col.push_back(0);
        for (auto& m : members) { list.insert(list.end(), m.begin(), m.end()); ctx->det_cross_col.push_back((int)list.size()); }
        if ((rc = upload_tmp(ctx, ctx->d_det_cross, list.data(), list.size() * 4))) return rc;
      }
      // groups of the single-launch kernel (k_reproj_fused): the same runs of equal (reference window, observation window), up to 64 blocks (a wavefront's lanes),
      // largest first — wavefront w of the launch takes groups w, w + W, ...: the second round holds the smallest.  OPT-IN (switch REP_FUSED = 1): measured at config 4
      // it takes 218 us against the chain's ~120 us critical path (the 512-register evaluation pins the whole kernel at one wavefront per SIMD, where its
      // 6 M scatter atomics retire at ~50 G/s), so the layout never picks it by itself.
      ctx->rep_fused_wg = 0;
      if (f.n > 0 && ctx->sw.rep_fused > 0 && !ctx->sw.deterministic) {
        std::vector<int> gs, gc;
        long long npairs = 0;
        int pa0 = -1, pa1 = -1;
        for (int i = 0; i < f.n; ++i) {
          if (s1[i] < 0) continue;
          const int a0 = s0[i] >> 2, a1 = s1[i] >> 2;
          const bool newpair = gs.empty() || pa0 != a0 || pa1 != a1;
          if (newpair) ++npairs;
          if (newpair || i - gs.back() >= 64 || gs.back() + gc.back() != i) { gs.push_back(i); gc.push_back(0); pa0 = a0; pa1 = a1; }
          gc.back()++;
        }
        (void)npairs;
        if (!gs.empty()) {
          std::vector<int> order(gs.size());
          std::iota(order.begin(), order.end(), 0);
          std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return gc[a] > gc[b]; });
          std::vector<int> tab;
          for (int g : order) tab.push_back(gs[g]);
          for (int g : order) tab.push_back(gc[g]);
          if ((rc = upload_tmp(ctx, ctx->d_repF, tab.data(), tab.size() * 4))) return rc;
          ctx->rep_fused_wg = (int)gs.size();
        }
      }
      ctx->rep_groups = (int)goff.size();
      goff.push_back(f.n);
      goff.insert(goff.end(), gw0.begin(), gw0.end()); goff.insert(goff.end(), gw1.begin(), gw1.end());
      if ((rc = upload_tmp(ctx, ctx->d_repB[2], goff.data(), goff.size() * 4))) return rc;
    }
    auto ts = gather(f.t, perm, 1); auto uv = gather(f.a3, perm, 2); auto lm = gather(f.id0, perm, 1);
    if ((rc = upload_tmp(ctx, f.d_t, ts.data(), ts.size() * 8))) return rc;
    if ((rc = upload_tmp(ctx, f.d_a3, uv.data(), uv.size() * 8))) return rc;
    if ((rc = upload_tmp(ctx, f.d_id0, lm.data(), lm.size() * 4))) return rc;
    if ((rc = upload_tmp(ctx, f.d_perm, perm.data(), perm.size() * 4))) return rc;
    if ((rc = dev_alloc(ctx, ctx->d_repB[0], (size_t)std::max(f.n, 1) * (2 * (REP_NC + 1) + 2) * 8))) return rc;   // materialised Jacobians (+ the time-offset column) + residuals
    if ((rc = dev_alloc(ctx, ctx->d_repB[1], (size_t)std::max(f.n, 1) * 2 * 4))) return rc;                    // knot intervals
    if ((rc = dev_alloc(ctx, ctx->d_repT, (size_t)std::max(f.n, 1) * 57 * 8))) return rc;                      // landmark-row records
  }
  {
    Family& f = ctx->cs;
    std::vector<int> key(f.n), perm(f.n);
    for (int i = 0; i < f.n; ++i) key[i] = (f.id0[i] >= 0 && f.id0[i] < L) ? host_i0(ctx, ctx->lm_t0[f.id0[i]]) : -1;
    stable_perm_by_key(key, perm);
    { std::vector<int> sk(f.n); for (int i = 0; i < f.n; ++i) sk[i] = key[perm[i]]; if (ctx->sw.chunk_r) rc = upload_chunks(ctx, LVX_FAM_CAMSURF, sk, pick_chunk(ctx, 8, 16, ctx->sw.chunk_r, 2, 2.2));
      else rc = upload_chunks_rows(ctx, LVX_FAM_CAMSURF, sk, 16, ctx->sw.chunk_rows > 0 ? ctx->sw.chunk_rows : 512);
      if (rc) return rc; }
    auto lm = gather(f.id0, perm, 1); auto pl = gather(f.id1, perm, 1);
    if ((rc = upload_tmp(ctx, f.d_id0, lm.data(), lm.size() * 4))) return rc;
    if ((rc = upload_tmp(ctx, f.d_id1, pl.data(), pl.size() * 4))) return rc;
    if ((rc = upload_tmp(ctx, f.d_perm, perm.data(), perm.size() * 4))) return rc;
  }
  if ((rc = upload_tmp(ctx, ctx->d_planes, ctx->planes.data(), ctx->planes.size() * 8))) return rc;
  if ((rc = upload_tmp(ctx, ctx->d_lm_uv, ctx->lm_uv.data(), ctx->lm_uv.size() * 8))) return rc;
  if ((rc = upload_tmp(ctx, ctx->d_lm_t0, ctx->lm_t0.data(), ctx->lm_t0.size() * 8))) return rc;
  // ---- hub knots (all surfel / cam-surfel residuals evaluate the trajectory at t_map: arrowhead) ----
  ctx->n_hub = 0; ctx->hub0 = 0;
  if (ctx->surf.n > 0 || ctx->cs.n > 0) {
    const bool any_free = !(locks & LVX_LOCK_LIDAR_TAU) || !(locks & LVX_LOCK_CAM_TAU);
    const double pad = any_free ? ctx->sensor_mto : 0.0;
    const double tl = ctx->t_map - pad, th = ctx->t_map + pad;
    const double tmax = ctx->t0 + (double)(N - 3) * ctx->dt;
    if (tl < ctx->t0 || th >= tmax) return fail(ctx, LVX_E_RANGE, "t_map outside the spline");
    const int i1 = (int)std::floor((tl - ctx->t0) / ctx->dt), i2 = (int)std::floor((th - ctx->t0) / ctx->dt);
    ctx->hub0 = i1;
    ctx->n_hub = std::min(N, i2 + 4 + 1) - i1;   // one spare knot for the merged-segment corner of spline_base.h:196-203
  }
  const int nh = ctx->n_hub, h0 = ctx->hub0;
  ctx->nbd = 6 * nh + 22;
  ctx->nbd_ext = ctx->nbd + 12;   // + pseudo pose rows: surfel hub (6), cam-surfel hub (6)
  // ---- band ordering: non-hub knots in time order; landmarks have their own rows (DevCommon::lmH) ----
  const int nt = 6 * N + 22 + L;
  ctx->ord.assign(nt + 12, LVX_DEAD);
  for (int k = 0; k < 12; ++k) ctx->ord[nt + k] = -1 - (ctx->nbd + k);
  std::vector<int> lm_first(L, N), lm_last(L, -1);
  const bool cam_tau_locked = (locks & LVX_LOCK_CAM_TAU) != 0;
  std::vector<int> rep_kmin(ctx->rep.n, 0), rep_kmax(ctx->rep.n, 0);
  for (int i = 0; i < ctx->rep.n; ++i) {
    const int l = ctx->rep.id0[i];
    if (l < 0 || l >= L) return fail(ctx, LVX_E_ARG, "reprojection landmark id out of range");
    double t1 = std::min(ctx->lm_t0[l], ctx->rep.t[i]), t2 = std::max(ctx->lm_t0[l], ctx->rep.t[i]);
    if (!cam_tau_locked) { t1 -= ctx->sensor_mto; t2 += ctx->sensor_mto; }
    const double spans[2][2] = {{t1 - 1e-3, t1 + ctx->cam.readout + 1e-3}, {t2 - 1e-3, t2 + ctx->cam.readout + 1e-3}};
    Segs sg;
    if (!build_segments(sp, spans, 2, &sg)) return fail(ctx, LVX_E_RANGE, "reprojection time span out of range for trajectory");
    int kmin = sg.i1[0], kmax = sg.i1[sg.nseg - 1] + sg.n[sg.nseg - 1] - 1;
    rep_kmin[i] = kmin; rep_kmax[i] = kmax;
    lm_first[l] = std::min(lm_first[l], kmin); lm_last[l] = std::max(lm_last[l], kmax);
  }
  int pos = 0;
  auto is_hub = [&](int k) { return nh > 0 && k >= h0 && k < h0 + nh; };
  for (int k = 0; k <= N; ++k) {
    if (k < N) {
      if (is_hub(k)) {
        for (int d = 0; d < 6; ++d) if (!tangent_locked(6 * k + d, N, L, locks)) ctx->ord[6 * k + d] = -1 - (6 * (k - h0) + d);
      } else {
        for (int d = 0; d < 6; ++d) if (!tangent_locked(6 * k + d, N, L, locks)) ctx->ord[6 * k + d] = pos++;
      }
    }
  }
  ctx->nb = pos;
  if (!(locks & LVX_LOCK_LANDMARKS)) for (int l = 0; l < L; ++l) ctx->ord[6 * N + 22 + l] = LVX_LM_BASE + l;
  for (int c = 0; c < 22; ++c) if (!tangent_locked(6 * N + c, N, L, locks)) ctx->ord[6 * N + c] = -1 - (6 * nh + c);
  // ---- scalar half-bandwidth ----
  int bw = 0;
  auto span_pos = [&](int ka, int kb, int& lo, int& hi) {
    for (int k = ka; k <= kb && k < N; ++k) for (int d = 0; d < 6; ++d) { const int o = ctx->ord[6 * k + d]; if (o >= 0) { lo = std::min(lo, o); hi = std::max(hi, o); } }
  };
  const int kspan = (!(locks & LVX_LOCK_LIDAR_TAU) || !(locks & LVX_LOCK_CAM_TAU)) ? 4 : 3;   // a free time offset pads the spans: segments of up to 5 control points
  std::vector<int> top((size_t)std::max(ctx->nb, 1), -1);   // furthest band position a span that STARTS at a position reaches (-> h_colhi: the column profile of the band, lvx_nd.h)
  auto mark_span = [&](int lo, int hi) { if (hi >= lo && lo < (int)top.size()) top[lo] = std::max(top[lo], hi); };
  for (int k = 0; k + 3 < N; ++k) { int lo = 1 << 30, hi = -1; span_pos(k, k + kspan, lo, hi); if (hi >= lo) bw = std::max(bw, hi - lo); mark_span(lo, hi); }
  const int bw_near = bw;   // reach of the families that touch 4 neighbouring knots only (IMU, LiDAR, prior)
  std::vector<uint8_t> colfull((size_t)std::max(ctx->nb, 1), 0);   // band columns a reprojection block / a landmark reaches further from (k_clear)
  for (int i = 0; i < ctx->rep.n; ++i) {
    int lo = 1 << 30, hi = -1;
    span_pos(rep_kmin[i], rep_kmax[i], lo, hi);
    if (hi >= lo) { bw = std::max(bw, hi - lo); if (hi - lo > bw_near) std::fill(colfull.begin() + lo, colfull.begin() + hi + 1, (uint8_t)1); }
    mark_span(lo, hi);
  }
  // the landmark elimination couples everything a landmark touches (fill of the reduced band); a landmark's row covers the same positions
  std::vector<int> lm_p0((size_t)std::max(L, 1), 0);
  int lm_wl = 1;
  for (int l = 0; l < L; ++l) if (lm_last[l] >= 0) {
    int lo = 1 << 30, hi = -1; span_pos(lm_first[l], lm_last[l], lo, hi);
    if (hi >= lo) {
      lm_p0[l] = lo; lm_wl = std::max(lm_wl, hi - lo + 1);
      if (!(locks & LVX_LOCK_LANDMARKS)) {   // a free landmark: the in-place elimination writes fill across its WHOLE reach, which may exceed every single block's
        bw = std::max(bw, hi - lo);
        if (hi - lo > bw_near) std::fill(colfull.begin() + lo, colfull.begin() + hi + 1, (uint8_t)1);
        mark_span(lo, hi);
      }
    }
  }
  ctx->lm_wl = lm_wl; ctx->lm_ls = lm_wl + ctx->nbd_ext + 2;
  {   // groups for the landmark elimination: landmarks of one reference frame start at the same band position and reach the same knots, so their rank-1
      // updates are summed per group (<= 32 landmarks whose first positions lie within 24 scalars) before they touch the band (k_lm_schur_grp)
    std::vector<int> order;
    for (int l = 0; l < L; ++l) if (lm_last[l] >= 0) order.push_back(l);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lm_p0[a] < lm_p0[b]; });
    std::vector<int> goff;
    int spread = 0;
    for (size_t i = 0; i < order.size(); ++i) {
      if (goff.empty() || (int)i - goff.back() >= 32 || lm_p0[order[i]] - lm_p0[order[goff.back()]] > 24) goff.push_back((int)i);
      spread = std::max(spread, lm_p0[order[i]] - lm_p0[order[goff.back()]]);
    }
    ctx->lm_ngrp = (int)goff.size(); ctx->lm_gspread = spread;
    goff.push_back((int)order.size());
    goff.insert(goff.end(), order.begin(), order.end());
    if ((rc = upload_tmp(ctx, ctx->d_lm_grp, goff.data(), goff.size() * 4))) return rc;
  }
  if ((rc = upload_tmp(ctx, ctx->d_lm_p0, lm_p0.data(), lm_p0.size() * 4))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_lmH, (size_t)std::max(L, 1) * ctx->lm_ls * 8 + 16))) return rc;
  ctx->bw = std::min(std::max(bw, 0), std::max(ctx->nb - 1, 0));
  ctx->clear_npre = std::min(bw_near, ctx->bw) + 1;
  {   // band positions that couple to a hub knot directly: rows of the 4-knot families within 5 knots of the hub, reprojection blocks within their knot span
    int reach = 5;
    for (int i = 0; i < ctx->rep.n; ++i) reach = std::max(reach, rep_kmax[i] - rep_kmin[i] + 2);
    int lo = 1 << 30, hi = -1;
    if (nh > 0) span_pos(std::max(0, h0 - reach), std::min(N - 1, h0 + nh + reach), lo, hi);
    ctx->hub_near_lo = hi >= lo ? lo : 0; ctx->hub_near_hi = hi >= lo ? hi + 1 : 0;
  }
  if ((rc = upload_tmp(ctx, ctx->d_colfull, colfull.data(), colfull.size()))) return rc;
  {   // column profile for the solver's elimination plan: h_colhi[j] = last band position column j couples to (j itself if none)
    ctx->h_colhi.assign((size_t)std::max(ctx->nb, 0), 0);
    int run = -1;
    for (int j = 0; j < ctx->nb; ++j) { run = std::max(run, top[j]); ctx->h_colhi[j] = std::max(run, j); }
    ctx->h_colfull.assign(colfull.begin(), colfull.begin() + std::max(ctx->nb, 0));
    ctx->bw_near = std::min(bw_near, ctx->bw);
    ++ctx->layout_epoch;
  }
  {   // band columns k_imu_own STORES (ImuOwn::own): knots whose every contribution comes from ONE workgroup's samples and whose 24-entry column maps onto
      // position-contiguous neighbours.  A sample sorted into interval i evaluates in i - 1 .. i + 1 (|tau_imu| < dt) and touches 4 knots, so a range whose first
      // interval is B shares the knots [B - 1, B + 3] with its left neighbour: those, knots next to the hub gap or the spline's end and knots no batch window covers stay
      // -1 = cleared by k_clear and added atomically.
    std::vector<int> own((size_t)std::max(ctx->nb, 1), -1), own_k((size_t)std::max(N, 1), -1);   // by band position (k_clear) and by knot (k_imu_own)
    const int G = ctx->imu_wg, cr = IMU_CR;
    const std::vector<int>& k0 = ctx->imu_h_k0; const std::vector<int>& wc = ctx->imu_h_wg_c0;
    ctx->imu_owned_cols = 0;
    if (ctx->nb > 0 && !(locks & LVX_LOCK_R3) && ctx->imu.n > 0 && ctx->dt > ctx->imu_mto) {
      int prev = -1;
      for (int g = 0; g < G; ++g) {
        if (wc[g] >= wc[g + 1]) continue;
        int nxt = -1; for (int h = g + 1; h < G && nxt < 0; ++h) if (wc[h] < wc[h + 1]) nxt = h;
        const int lo = prev < 0 ? 0 : k0[wc[g]] + 4, hi = nxt < 0 ? N - 1 : k0[wc[nxt]] - 2;
        for (int c = wc[g]; c < wc[g + 1]; ++c)
          for (int k = std::max(lo, k0[c] - 1); k <= std::min(hi, k0[c] - 1 + cr + 4); ++k) {
            if (k < 0 || k + 4 >= N) continue;
            const int p0 = ctx->ord[6 * k];
            if (p0 < 0 || p0 >= LVX_LM_BASE) continue;
            bool contig = true;
            for (int e = 0; e < 30 && contig; ++e) contig = ctx->ord[6 * k + e] == p0 + e;
            if (!contig) continue;
            for (int d = 0; d < 6; ++d) { if (own[p0 + d] < 0) ctx->imu_owned_cols++; own[p0 + d] = g; }
            own_k[k] = g;
          }
        prev = g;
      }
    }
    ctx->imu_own_k_off = own.size();
    own.insert(own.end(), own_k.begin(), own_k.end());
    if ((rc = upload_tmp(ctx, ctx->d_imu_own, own.data(), own.size() * 4))) return rc;
  }
  ctx->bd_row_live.assign((size_t)ctx->nbd_ext, 0);
  for (int b = 0; b < 6 * nh; ++b) ctx->bd_row_live[b] = 1;
  for (int c = 0; c < 22; ++c) if (ctx->ord[6 * N + c] != LVX_DEAD) ctx->bd_row_live[6 * nh + c] = 1;
  for (int p = 0; p < 6; ++p) { ctx->bd_row_live[ctx->nbd + p] = ctx->surf.n > 0; ctx->bd_row_live[ctx->nbd + 6 + p] = ctx->cs.n > 0; }
  // ---- deterministic mode: colours of chunks with pairwise disjoint knot ranges; one replica of the dense accumulators per workgroup ----
  ctx->nrep = LVX_NREP;
  for (auto& v : ctx->det_col) v.clear();
  if (ctx->sw.deterministic) {
    int need = std::max(LVX_NREP, (ctx->rep.n + 63) / 64);
    for (size_t q = 0; q + 1 < ctx->det_cross_col.size(); ++q) need = std::max(need, ctx->det_cross_col[q + 1] - ctx->det_cross_col[q]);
    for (int slot : {LVX_FAM_GYRO, LVX_FAM_SURFEL, LVX_FAM_CAMSURF, LVX_FAM_REPROJ, LVX_FAM_PRIOR}) {
      const int nch = ctx->n_chunk[slot];
      if (nch <= 0 || (int)ctx->h_chunk_k0[slot].size() != nch) continue;
      const int span = ctx->chunk_r[slot] + 5;
      std::vector<std::vector<uint8_t>> occ; std::vector<std::vector<int>> members;
      for (int c = 0; c < nch; ++c) {
        if (ctx->h_chunk_rows[slot][c] <= 0) continue;
        const int lo = std::max(0, ctx->h_chunk_k0[slot][c] - 1), hi = std::min(N, ctx->h_chunk_k0[slot][c] - 1 + span);
        size_t col = 0;
        for (; col < occ.size(); ++col) { bool hit = false; for (int k = lo; k < hi && !hit; ++k) hit = occ[col][k] != 0; if (!hit) break; }
        if (col == occ.size()) { occ.emplace_back((size_t)N, (uint8_t)0); members.emplace_back(); }
        for (int k = lo; k < hi; ++k) occ[col][k] = 1;
        members[col].push_back(c);
      }
      std::vector<int> list; ctx->det_col[slot].push_back(0);
      for (auto& m : members) { list.insert(list.end(), m.begin(), m.end()); ctx->det_col[slot].push_back((int)list.size()); }
      if (list.empty()) { ctx->det_col[slot].clear(); continue; }
      if ((rc = upload_tmp(ctx, ctx->d_det_list[slot], list.data(), list.size() * 4))) return rc;
      need = std::max(need, nch);
    }
    if (need > (1 << 16)) return fail(ctx, LVX_E_STATE, "deterministic mode: more than 65536 workgroups per launch (one replica of the dense accumulators each)");
    ctx->nrep = need;
  }
  // ---- buffers ----
  if ((rc = upload_tmp(ctx, ctx->d_ord, ctx->ord.data(), ctx->ord.size() * 4))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_Hb, (size_t)std::max(ctx->nb, 1) * (ctx->bw + 1) * 8 + 16))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_gb, (size_t)std::max(ctx->nb, 1) * 8 + 16))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_Bd, (size_t)ctx->nbd_ext * std::max(ctx->nb, 1) * 8 + 16))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_C, (size_t)ctx->nrep * ctx->nbd_ext * ctx->nbd_ext * 8 + 16))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_gc, (size_t)ctx->nrep * ctx->nbd_ext * 8 + 16))) return rc;   // + 16 each: k_clear works in 16-byte words
  // the per-pass clear is structural (k_clear): what it never touches must be zero from the start
  LVX_HIP(ctx, hipMemsetAsync(ctx->d_Hb.p, 0, ctx->d_Hb.bytes, ctx->stream));
  LVX_HIP(ctx, hipMemsetAsync(ctx->d_Bd.p, 0, ctx->d_Bd.bytes, ctx->stream));
  LVX_HIP(ctx, hipMemsetAsync(ctx->d_lmH.p, 0, ctx->d_lmH.bytes, ctx->stream));
  if ((rc = dev_alloc(ctx, ctx->d_hubs, 2 * sizeof(HubShared)))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_pre, (size_t)std::max(N, 1) * sizeof(So3Pre)))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_cost, (size_t)ctx->nrep * 8))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_err, 64))) return rc;   // [error bits, -, -, - | fallback rows per family (6) | -]
  if ((rc = dev_alloc(ctx, ctx->d_fb, (size_t)LVX_NUM_FAM * LVX_FB_CAP * 4))) return rc;
  if ((rc = dev_alloc(ctx, ctx->d_state, (size_t)lvx_state_size(ctx) * 8))) return rc;
  auto range = [](int a, int b) { std::vector<int> v; for (int i = a; i < b; ++i) v.push_back(i); return v; };
  auto cat = [](std::vector<int> a, const std::vector<int>& b) { a.insert(a.end(), b.begin(), b.end()); return a; };
  if ((rc = upload_pairs(ctx, ctx->d_pairs[0], GYRO_NC, range(12, 15)))) return rc;
  if ((rc = upload_pairs(ctx, ctx->d_pairs[1], ACC_NC, range(24, 29)))) return rc;
  if ((rc = upload_pairs(ctx, ctx->d_pairs[2], PRI_NC))) return rc;
  // a free sensor time offset adds one (border) column to the per-segment kernels of the families that evaluate at t + tau
  const int tL = (locks & LVX_LOCK_LIDAR_TAU) ? 0 : 1, tC = (locks & LVX_LOCK_CAM_TAU) ? 0 : 1;
  if ((rc = upload_pairs(ctx, ctx->d_pairs[3], SURF_NC + tL, cat(range(0, 24), range(48, 54 + tL))))) return rc;
  if ((rc = upload_pairs(ctx, ctx->d_pairs[4], REP_NC + tC, cat(range(48, 54), range(55, 55 + tC))))) return rc;
  if ((rc = upload_pairs(ctx, ctx->d_pairs[5], CS_NC + tC, cat(range(0, 24), range(48, 60 + tC))))) return rc;
  ctx->force_legacy = false; ctx->fb_on = false; ctx->fb_mask = 0; ctx->fallback_rows = 0;
  { static const int zero[2] = {0, 0}; if ((rc = upload_tmp(ctx, ctx->d_zero, zero, 8))) return rc; }   // [0]: identity permutation of the single prior block, [1]: ticket of k_fold_all (returns to zero by itself)
  if ((rc = dev_alloc(ctx, ctx->d_imu_rtab, (size_t)64 * (ImuG::NTP * 4 + ImuA::NTP * 4) * 4))) return rc;
  hipLaunchKernelGGL(k_imu_rtab, dim3(1), dim3(64), 0, ctx->stream, (int*)ctx->d_imu_rtab.p);
  ctx->cfg_version++;   // captured evaluation graphs of the previous layout are stale
  // ---- residual row offsets ----
  const int64_t cnt[LVX_NUM_FAM] = {ctx->imu.n, (locks & LVX_LOCK_R3) ? 0 : ctx->imu.n, ctx->has_prior ? 1 : 0, ctx->surf.n, ctx->rep.n, ctx->cs.n};
  const int nrs[LVX_NUM_FAM] = {3, 3, 1, 1, 2, 1};
  ctx->n_blocks = 0; ctx->fam_row0[0] = 0;
  for (int f = 0; f < LVX_NUM_FAM; ++f) { ctx->fam_row0[f + 1] = ctx->fam_row0[f] + cnt[f] * nrs[f]; ctx->n_blocks += cnt[f]; }
  ctx->n_residuals = ctx->fam_row0[LVX_NUM_FAM];
  ctx->layout_dirty = false;
  return LVX_OK;
}

DevCommon make_common(lvx_ctx* ctx, const double* state_d, uint32_t what) {
  DevCommon cm{};
  cm.state = state_d; cm.N = ctx->N; cm.L = ctx->L; cm.t0 = ctx->t0; cm.dt = ctx->dt; cm.locks = ctx->locks; cm.what = what;
  cm.imu_mto = ctx->imu_mto; cm.sensor_mto = ctx->sensor_mto; cm.cam = ctx->cam;
  cm.ord = (const int*)ctx->d_ord.p; cm.nb = ctx->nb; cm.bw = ctx->bw; cm.nbd = ctx->nbd_ext; cm.nbd_solve = ctx->nbd; cm.hubs = ctx->d_hubs.p; cm.pre = (const So3Pre*)ctx->d_pre.p;
  cm.Hb = (double*)ctx->d_Hb.p; cm.gb = (double*)ctx->d_gb.p; cm.Bd = (double*)ctx->d_Bd.p; cm.C = (double*)ctx->d_C.p; cm.gc = (double*)ctx->d_gc.p;
  cm.cost = (double*)ctx->d_cost.p; cm.err = (int*)ctx->d_err.p;
  cm.fb_list = ctx->fb_on ? (int*)ctx->d_fb.p : nullptr; cm.fb_cap = LVX_FB_CAP | (ctx->fb_mask << 16);   // (capacity | families with a list << 16)
  cm.lmH = (double*)ctx->d_lmH.p; cm.lm_p0 = (const int*)ctx->d_lm_p0.p; cm.lm_wl = ctx->lm_wl; cm.lm_ls = ctx->lm_ls;
  cm.hub_lo = 0; cm.hub_hi = ctx->nb; cm.nrep = ctx->nrep;
  cm.residuals = nullptr; cm.jcols = nullptr; cm.jvals = nullptr;
  return cm;
}

static size_t next_event(lvx_ctx* c) {
  // timing only: no system-scope fence (cache writeback + invalidation) at every record
  if (c->ev_used == c->ev_pool.size()) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return (size_t)-1; c->ev_pool.push_back(e); }
  return c->ev_used++;
}
ProfScope::ProfScope(lvx_ctx* ctx, int k, hipStream_t stream) : c(ctx), kernel(k), on(ctx->profiling && (ctx->profile_only < 0 || ctx->profile_only == k)), st(stream ? stream : ctx->stream) {
  if (!on) return;
  e0 = next_event(c);
  if (e0 == (size_t)-1) { on = false; return; }
  (void)hipEventRecord(c->ev_pool[e0], st);
}
ProfScope::~ProfScope() {
  if (!on) return;
  const size_t e1 = next_event(c);
  if (e1 == (size_t)-1) return;
  (void)hipEventRecord(c->ev_pool[e1], st);
  c->ev_recs.push_back(lvx_ctx::EvRec{kernel, e0, e1});
}

// clears up to 16 device buffers (sizes rounded up to 16 bytes: every buffer is allocated with that slack by dev_alloc's callers) in one launch
struct ClearList { uint4* p[16]; size_t words[16]; int n; };
// The band is cleared STRUCTURALLY: every column's first `npre` entries (what the IMU / LiDAR families can touch: 4 neighbouring knots and
// the landmarks ordered between them), whole columns only where `colfull` says a reprojection block or a landmark reaches further
// (ensure_layout).  Everything else was zeroed once at layout time and is never written.  Config 4 with ORB-like tracks: 60 MB instead of 242 MB.
struct BandClear { double* Hb; const uint8_t* colfull; int nb, ld, npre, nblk; double* Bd; int hub_rows, hub_lo, hub_hi, hub_blk;   // hub rows of Bd: only [hub_lo, hub_hi) (the fold stores the rest)
                   const int* own; double* gb; double* Bd_imu; unsigned imu_rows; };   // own[j] >= 0: k_imu_own STORES the first ACC_BW entries of band column j, gb[j] and the IMU-calibration border rows (Bd_imu: 8 rows, live ones in imu_rows) at j
// First launch of a pass.  Its first npre blocks do what depends on the state only (control-point-pair table, hub poses: k_state_prepass's
// work — a separate kernel on a side stream costs a ~30 us cross-stream join before the LiDAR kernels); the rest clear the accumulators.
__global__ __launch_bounds__(256) void k_clear(ClearList cl, BandClear bc, int npre, DevCommon cm, So3Pre* tab, int nblk_tab, double t_map, int want_surf, int want_cs, HubShared* hubs) {
  if ((int)blockIdx.x < npre) { state_prepass_block(cm, tab, nblk_tab, t_map, want_surf, want_cs, hubs, blockIdx.x); return; }
  if ((int)blockIdx.x < npre + bc.nblk) {   // 16 lanes per band column, 16 columns per trip; 16-byte stores when every column starts 16-byte aligned (ld even)
    const int l16 = threadIdx.x & 15;
    for (int j = ((int)blockIdx.x - npre) * 16 + (threadIdx.x >> 4); j < bc.nb; j += bc.nblk * 16) {
      const int len = bc.colfull[j] ? bc.ld : bc.npre;
      const bool owned = bc.own && bc.own[j] >= 0;
      const int beg = owned ? min(ACC_BW, len) : 0;   // (even)
      if (bc.Hb) {
        double* col = bc.Hb + (size_t)j * bc.ld;
        if (!(bc.ld & 1)) {
          uint4* c4 = (uint4*)col;
          for (int e = (beg >> 1) + l16; e < (len >> 1); e += 16) c4[e] = make_uint4(0u, 0u, 0u, 0u);
          if ((len & 1) && l16 == 0 && len > beg) col[len - 1] = 0.0;
        } else {
          for (int e = beg + l16; e < len; e += 16) col[e] = 0.0;
        }
      }
      if (bc.own && !owned) {
        if (l16 == 8) bc.gb[j] = 0.0;
        else if (l16 < 8 && ((bc.imu_rows >> l16) & 1u)) bc.Bd_imu[(size_t)l16 * bc.nb + j] = 0.0;
      }
    }
    return;
  }
  if ((int)blockIdx.x < npre + bc.nblk + bc.hub_blk) {   // near range of the hub rows: one row per block
    const int row = (int)blockIdx.x - npre - bc.nblk;
    double* dst = bc.Bd + (size_t)row * bc.nb;
    for (int e = bc.hub_lo + threadIdx.x; e < bc.hub_hi; e += blockDim.x) dst[e] = 0.0;
    return;
  }
  const int first = npre + bc.nblk + bc.hub_blk;
  const size_t stride = (size_t)(gridDim.x - first) * blockDim.x, t0 = (size_t)(blockIdx.x - first) * blockDim.x + threadIdx.x;
  for (int b = 0; b < cl.n; ++b) {   // 4 stores per trip: the prepass code leaves this kernel 2 wavefronts per SIMD, the stores keep HBM busy anyway
    const size_t nw = cl.words[b];
    size_t i = t0;
    for (; i + 3 * stride < nw; i += 4 * stride) {
      cl.p[b][i] = make_uint4(0u, 0u, 0u, 0u); cl.p[b][i + stride] = make_uint4(0u, 0u, 0u, 0u);
      cl.p[b][i + 2 * stride] = make_uint4(0u, 0u, 0u, 0u); cl.p[b][i + 3 * stride] = make_uint4(0u, 0u, 0u, 0u);
    }
    for (; i < nw; i += stride) cl.p[b][i] = make_uint4(0u, 0u, 0u, 0u);
  }
}

static bool fast_fb(const lvx_ctx* ctx, uint32_t what) { return ctx->fb_on && !ctx->force_legacy && !(what & LVX_EVAL_JACOBIAN) && !ctx->sw.force_legacy; }
int run_evaluate(lvx_ctx* ctx, const double* state_d, uint32_t what, double* cost, bool want_res_buffer) {
  int rc = ensure_layout(ctx);
  if (rc) return rc;
  hipStream_t st = ctx->stream;
  DevCommon cm = make_common(ctx, state_d, what);
  if (want_res_buffer) {
    if ((rc = dev_alloc(ctx, ctx->d_res, (size_t)std::max<int64_t>(ctx->n_residuals, 1) * 8))) return rc;
    cm.residuals = (double*)ctx->d_res.p;
  }
  if (what & LVX_EVAL_JACOBIAN) {
    const size_t nrow = (size_t)std::max<int64_t>(ctx->n_residuals, 1);
    if ((rc = dev_alloc(ctx, ctx->d_jcols, nrow * LVX_JAC_WIDTH * 4))) return rc;
    if ((rc = dev_alloc(ctx, ctx->d_jvals, nrow * LVX_JAC_WIDTH * 8))) return rc;
    cm.jcols = (int32_t*)ctx->d_jcols.p; cm.jvals = (double*)ctx->d_jvals.p;
    LVX_HIP(ctx, hipMemsetAsync(cm.jcols, 0xff, nrow * LVX_JAC_WIDTH * 4, st));
    LVX_HIP(ctx, hipMemsetAsync(cm.jvals, 0, nrow * LVX_JAC_WIDTH * 8, st));
  }
  // Everything from the clears to the fold kernels is one static launch sequence for a given (state buffer, request, configuration):
  // it is captured once into a HIP graph and replayed — a pass is ~35 API calls (memsets, cross-stream events, ~12 launches), which bounds small
  // problems at ~370 us per evaluation when issued call by call.  LVX_NO_GRAPH=1 issues the calls directly; profiling and the debug
  // Jacobian always do.
  auto enqueue = [&]() -> int {
    int rc = LVX_OK;
    const bool fast = !ctx->force_legacy && !(what & LVX_EVAL_JACOBIAN) && !ctx->sw.force_legacy;
    const bool fb = fast_fb(ctx, what);   // row-level exact fallback: the per-segment kernel follows every fused kernel over its fallback list
    // free time offsets need d pose / d t at both evaluations: one more global column in the fused LiDAR / camera-surfel kernels (SurfAccT<true>, CamSurfAccT<true>);
    // the reprojection path a time-offset column in its materialised rows; FORCE_LEGACY: the per-segment TAU kernels for everything
    const bool tauL = !(ctx->locks & LVX_LOCK_LIDAR_TAU), tauC = !(ctx->locks & LVX_LOCK_CAM_TAU);
    const bool fast_surf = fast && ctx->surf.n > 0, fast_cs = fast && ctx->cs.n > 0;
    // the control-point-pair table and the shared t_map poses (one thread, ~25 us) depend on the state only: the first blocks of the clear kernel
    const int nblk_tab = fast ? (ctx->N + 255) / 256 : 0, npre = fast ? nblk_tab + ((fast_surf || fast_cs) ? 1 : 0) : 0;
    const lvx::Switches& sw = ctx->sw;
    const bool det = sw.deterministic != 0;   // fixed order of every addition: one stream, coloured launches, one wavefront per workgroup
    const bool imu_fused_on = fast && ctx->imu.n > 0 && !(ctx->locks & LVX_LOCK_R3) && ctx->imu_nch > 0 && imu_fused_lds_bytes(ctx->imu_span) <= 160 * 1024;
    const bool imu_own = imu_fused_on && (what & LVX_EVAL_NORMAL_EQ) && ctx->nb > 0 && ctx->imu_owned_cols > 0 && !sw.clear_all;   // k_imu_own stores its band columns: they are not cleared
    {   // one launch clears every accumulator of the pass (cost, error flags, and for the normal equations band, gradient, border rows, dense border)
      ClearList cl{};
      BandClear bc{};
      auto add = [&](void* p, size_t bytes) { if (cl.n < 16) { cl.p[cl.n] = (uint4*)p; cl.words[cl.n] = (bytes + 15) / 16; cl.n++; } };
      add(cm.cost, (size_t)ctx->nrep * 8); add(cm.err, 64);
      if (what & LVX_EVAL_NORMAL_EQ) {
        const size_t nb1 = (size_t)std::max(ctx->nb, 1);
        if (ctx->sw.clear_all || ctx->nb == 0) add(cm.Hb, nb1 * (ctx->bw + 1) * 8);
        else { bc.Hb = cm.Hb; bc.colfull = (const uint8_t*)ctx->d_colfull.p; bc.nb = ctx->nb; bc.ld = ctx->bw + 1; bc.npre = ctx->clear_npre; bc.nblk = std::min((ctx->nb + 15) / 16, 2048); }
        if (imu_own) {   // gradient and the IMU-calibration border rows: cleared column by column where nobody stores (the band loop of k_clear)
          bc.own = (const int*)ctx->d_imu_own.p; bc.gb = cm.gb; bc.Bd_imu = cm.Bd + (size_t)6 * ctx->n_hub * nb1; bc.imu_rows = 0;
          for (int c = 0; c < 8; ++c) if (ctx->bd_row_live[6 * ctx->n_hub + c]) bc.imu_rows |= 1u << c;
          if (!bc.nblk) { bc.colfull = (const uint8_t*)ctx->d_colfull.p; bc.nb = ctx->nb; bc.ld = ctx->bw + 1; bc.npre = ctx->clear_npre; bc.nblk = std::min((ctx->nb + 15) / 16, 2048); }
        } else add(cm.gb, nb1 * 8);
        // hub rows of Bd: with the fused LiDAR kernels only the fold fills them beyond the near range — it stores there, the clear skips them
        const bool fb_lidar = fb && (ctx->fb_mask & ((1 << LVX_FAM_SURFEL) | (1 << LVX_FAM_CAMSURF)));   // listed LiDAR rows add to the hub rows directly, anywhere
        const bool hub_partial = !fb_lidar && !ctx->sw.clear_all && !(nb1 & 1) && ctx->nb > 0 && ctx->n_hub > 0 && (fast_surf || fast_cs) && (ctx->surf.n == 0 || fast_surf) && (ctx->cs.n == 0 || fast_cs);
        cm.hub_lo = hub_partial ? ctx->hub_near_lo : 0; cm.hub_hi = hub_partial ? ctx->hub_near_hi : ctx->nb;
        if (hub_partial) { bc.Bd = cm.Bd; bc.hub_rows = 6 * ctx->n_hub; bc.hub_lo = cm.hub_lo; bc.hub_hi = cm.hub_hi; bc.hub_blk = cm.hub_hi > cm.hub_lo ? 6 * ctx->n_hub : 0; if (!bc.nb) bc.nb = ctx->nb; }
        // border rows: only the rows some residual can reach (a locked calibration scalar and an unused pseudo-pose set keep their zeros); 16-byte words: whole rows when nb is even
        if (ctx->sw.clear_all || (nb1 & 1)) add(cm.Bd, (size_t)ctx->nbd_ext * nb1 * 8);
        else for (int b0 = 0; b0 < ctx->nbd_ext;) {
          const int first_live = hub_partial ? 6 * ctx->n_hub : 0;   // hub rows: HubClear
          const int imu0 = imu_own ? 6 * ctx->n_hub : -1;   // the 8 IMU-calibration rows: BandClear
          if (b0 < first_live || !ctx->bd_row_live[b0] || (imu0 >= 0 && b0 >= imu0 && b0 < imu0 + 8)) { ++b0; continue; }
          int b1 = b0; while (b1 < ctx->nbd_ext && ctx->bd_row_live[b1] && !(imu0 >= 0 && b1 >= imu0 && b1 < imu0 + 8)) ++b1;
          add(cm.Bd + (size_t)b0 * nb1, (size_t)(b1 - b0) * nb1 * 8);
          b0 = b1;
        }
        add(cm.C, (size_t)ctx->nrep * ctx->nbd_ext * ctx->nbd_ext * 8); add(cm.gc, (size_t)ctx->nrep * ctx->nbd_ext * 8);
        const bool rep_fast = fast && ctx->rep_groups > 0 && ctx->rep_fused_wg == 0;   // k_reproj_lmrows stores whole rows: nothing to clear (k_reproj_fused adds to them)
        if (ctx->L > 0 && ctx->rep.n > 0 && !(ctx->locks & LVX_LOCK_LANDMARKS) && !rep_fast) add(cm.lmH, (size_t)ctx->L * ctx->lm_ls * 8);
      }
      size_t total = 0; for (int i = 0; i < cl.n; ++i) total += cl.words[i];
      const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 8);   // 8 per CU (round 5b, 37 MB to clear: 512 blocks 0.540 ms per pass, 1 024 0.542, 2 048 0.5396, 8 192 — one 16-byte store per thread — 0.5478)
      ProfScope ps(ctx, LVX_KERNEL_CLEAR, st);
      hipLaunchKernelGGL(k_clear, dim3(blocks + (unsigned)npre + (unsigned)bc.nblk + (unsigned)bc.hub_blk), dim3(256), 0, st, cl, bc, npre, cm, (So3Pre*)ctx->d_pre.p, nblk_tab, ctx->t_map, fast_surf ? 1 : 0, fast_cs ? 1 : 0,
                         (HubShared*)ctx->d_hubs.p);
    }
    auto grid = [](int n) { return dim3((unsigned)((n + 63) / 64)); };
    // exact per-segment kernel over the rows a fused kernel put on family f's fallback list (row-level fallback: enabled after the first pass that needed it)
    const dim3 fb_grid((unsigned)(LVX_FB_CAP / 64));
    const int* fb_rows_base = (const int*)ctx->d_fb.p;
    auto fb_rows = [&](int f) { return fb_rows_base + (size_t)f * LVX_FB_CAP; };
    auto fb_cnt = [&](int f) { return (const int*)cm.err + 4 + f; };
    auto fb_fam = [&](int f) { return fb && ((ctx->fb_mask >> f) & 1); };
    // The listed rows are evaluated by the (slow: 40-100 us for a single block) per-segment kernels on a SIDE stream, right behind the fused kernel that filled the list and
    // beside the rest of the pass; they join in front of the fold and add to the same accumulators as everything else.  What the pass STORES instead of adding to is
    // arranged accordingly: k_imu_own's columns are stored before the IMU lists run, the fold's store of the far hub rows is switched off when a LiDAR family has a list
    // (hub_partial below), the landmark rows are stored by k_reproj_lmrows on the chain before the reprojection list runs there.
    hipStream_t s_fb = (sw.serial || det) ? st : ctx->fam_stream[1];
    bool fb_used = false;
    int fb_ev = 0;
    auto fb_fork = [&]() -> int { if (s_fb != st) { LVX_HIP(ctx, hipEventRecord(ctx->ev_fb[fb_ev], st)); LVX_HIP(ctx, hipStreamWaitEvent(s_fb, ctx->ev_fb[fb_ev], 0)); fb_ev ^= 1; fb_used = true; } return LVX_OK; };
    if (imu_fused_on) {   // gyroscope + accelerometer blocks in one owner-computes kernel, FIRST on the band: it stores what it owns (k_imu_own)
      const ImuFused f{ctx->imu.n, (const double*)ctx->imu.d_t.p, (const double*)ctx->imu.d_a3.p, (const double*)ctx->imu.d_b3.p, (const int*)ctx->imu.d_perm.p, ctx->imu.weight, ctx->imu.huber /*w_acc*/};
      const size_t lds_ = imu_fused_lds_bytes(ctx->imu_span);
      LVX_HIP(ctx, hipFuncSetAttribute((const void*)k_imu_own, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_));
      const ImuOwn ow{(const int*)ctx->d_imu_wg.p, imu_own ? (const int*)ctx->d_imu_own.p + ctx->imu_own_k_off : nullptr, ctx->imu_nch, ctx->imu_span};
      ProfScope ps(ctx, LVX_FAM_GYRO, st);
      hipLaunchKernelGGL(k_imu_own, dim3(ctx->imu_wg), dim3(256), lds_, st, f, cm, (const int*)ctx->d_imu_chunk.p, ow, (long long)ctx->fam_row0[0], (long long)ctx->fam_row0[1], det ? 1 : 0, (const int*)ctx->d_imu_rtab.p);
      if (fb_fam(LVX_FAM_GYRO)) {   // listed samples: both blocks by the exact kernels (they ADD to what k_imu_own stored)
        if ((rc = fb_fork())) return rc;
        ProfScope psf(ctx, LVX_KERNEL_FIXUP, s_fb);
        GyroFam gf{ctx->imu.n, (const double*)ctx->imu.d_t.p, (const double*)ctx->imu.d_a3.p, (const int*)ctx->imu.d_perm.p, ctx->imu.weight, 0.0};
        hipLaunchKernelGGL((k_family<GyroFam, 1>), fb_grid, dim3(64), 0, s_fb, gf, cm, (const uint16_t*)ctx->d_pairs[0].p, (long long)ctx->fam_row0[0], fb_rows(LVX_FAM_GYRO), fb_cnt(LVX_FAM_GYRO));
        AccelFam af{ctx->imu.n, (const double*)ctx->imu.d_t.p, (const double*)ctx->imu.d_b3.p, (const int*)ctx->imu.d_perm.p, ctx->imu.huber /*w_acc*/, 0.0};
        hipLaunchKernelGGL((k_family<AccelFam, LVX_PW>), fb_grid, dim3(64 * LVX_PW), 0, s_fb, af, cm, (const uint16_t*)ctx->d_pairs[1].p, (long long)ctx->fam_row0[1], fb_rows(LVX_FAM_GYRO), fb_cnt(LVX_FAM_GYRO));
      }
    }
    // Schedule.  ONE chain on the caller's stream — clear -> fused IMU kernel (stores its band columns) -> prior -> LiDAR kernels -> reprojection Jacobian -> observation
    // pass -> cross terms -> landmark rows -> fold — and ONE side stream that takes the reprojection reference pass (it only reads the materialised rows and adds
    // atomically) behind the Jacobian kernel and joins in front of the fold.  A hand-over between streams costs 10-30 us on this stack, graph or no graph, so nothing
    // else forks (everything concurrent, a stream per family, the staged two-chain schedule of rounds 1-3 were measured and dropped).  SERIAL and
    // DETERMINISTIC keep everything on the chain.
    const bool side_on = !sw.serial && !det;
    hipStream_t s_side = side_on ? ctx->fam_stream[0] : st;
    bool side_used = false;
  #define LVX_T2(...) __VA_ARGS__
  #define LVX_LAUNCH_MFMA1(FT, OCCV, fam_obj, chunk_slot, stream, row0v)                                                                        \
    do {                                                                                                                                     \
      const size_t lds_ = mfma_lds_bytes<FT>(ctx->chunk_r[chunk_slot]);                                                                      \
      LVX_HIP(ctx, hipFuncSetAttribute((const void*)k_family_mfma<FT, OCCV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_));       \
      if (det && ctx->det_col[chunk_slot].size() > 1) {   /* colour after colour: no two workgroups of a launch touch the same knots */            \
        const std::vector<int>& dc_ = ctx->det_col[chunk_slot];                                                                              \
        for (size_t q_ = 0; q_ + 1 < dc_.size(); ++q_)                                                                                       \
          hipLaunchKernelGGL((k_family_mfma<FT, OCCV>), dim3(dc_[q_ + 1] - dc_[q_]), dim3(256), lds_, stream, fam_obj, cm, (const int*)ctx->d_chunk[chunk_slot].p, (long long)(row0v), \
                             ctx->chunk_r[chunk_slot], ctx->chunk_var[chunk_slot] ? ctx->n_chunk[chunk_slot] : 0, (const int*)ctx->d_det_list[chunk_slot].p + dc_[q_]); \
      } else                                                                                                                                 \
      hipLaunchKernelGGL((k_family_mfma<FT, OCCV>), dim3(ctx->n_chunk[chunk_slot]), dim3(256), lds_, stream, fam_obj, cm, (const int*)ctx->d_chunk[chunk_slot].p, (long long)(row0v), \
                         ctx->chunk_r[chunk_slot], ctx->chunk_var[chunk_slot] ? ctx->n_chunk[chunk_slot] : 0, (const int*)nullptr);          \
    } while (0)
  #define LVX_LAUNCH_MFMA(FT, fam_obj, chunk_slot, stream, row0v)                                                                               \
    do { if ((int)FT::OCC == 1) LVX_LAUNCH_MFMA1(FT, 1, fam_obj, chunk_slot, stream, row0v); else LVX_LAUNCH_MFMA1(FT, 2, fam_obj, chunk_slot, stream, row0v); } while (0)
    // ---- IMU blocks when the fused kernel does not apply: Solve #0 (no R3 spline: gyroscope blocks only) on the MFMA path, everything else per segment ----
    if (ctx->imu.n > 0 && !imu_fused_on) {
      if (fast) {
        GyroAcc g{ctx->imu.n, (const double*)ctx->imu.d_t.p, (const double*)ctx->imu.d_a3.p, (const int*)ctx->imu.d_perm.p, ctx->imu.weight, 0.0};
        ProfScope ps(ctx, LVX_FAM_GYRO, st); LVX_LAUNCH_MFMA(GyroAcc, g, LVX_FAM_GYRO, st, ctx->fam_row0[0]);
        if (fb_fam(LVX_FAM_GYRO)) { if ((rc = fb_fork())) return rc;
          ProfScope psf(ctx, LVX_KERNEL_FIXUP, s_fb);
          GyroFam gf{ctx->imu.n, (const double*)ctx->imu.d_t.p, (const double*)ctx->imu.d_a3.p, (const int*)ctx->imu.d_perm.p, ctx->imu.weight, 0.0};
          hipLaunchKernelGGL((k_family<GyroFam, 1>), fb_grid, dim3(64), 0, s_fb, gf, cm, (const uint16_t*)ctx->d_pairs[0].p, (long long)ctx->fam_row0[0], fb_rows(LVX_FAM_GYRO), fb_cnt(LVX_FAM_GYRO)); }
      } else {
        GyroFam g{ctx->imu.n, (const double*)ctx->imu.d_t.p, (const double*)ctx->imu.d_a3.p, (const int*)ctx->imu.d_perm.p, ctx->imu.weight, 0.0};
        ProfScope ps(ctx, LVX_FAM_GYRO, st);
        hipLaunchKernelGGL((k_family<GyroFam, 1>), grid(g.n), dim3(64), 0, st, g, cm, (const uint16_t*)ctx->d_pairs[0].p, (long long)ctx->fam_row0[0]);
      }
      if (!(ctx->locks & LVX_LOCK_R3)) {
        AccelFam a{ctx->imu.n, (const double*)ctx->imu.d_t.p, (const double*)ctx->imu.d_b3.p, (const int*)ctx->imu.d_perm.p, ctx->imu.huber /*w_acc*/, 0.0};
        ProfScope ps(ctx, LVX_FAM_ACCEL, st);
        hipLaunchKernelGGL((k_family<AccelFam, LVX_PW>), grid(a.n), dim3(64 * LVX_PW), 0, st, a, cm, (const uint16_t*)ctx->d_pairs[1].p, (long long)ctx->fam_row0[1]);
      }
    }
    if (ctx->has_prior) {
      DevBuf& pb = ctx->d_zero;   // identity permutation for the single prior block (zeroed by ensure_layout)
      PriorFam p{1, ctx->prior_t, mkq(ctx->prior_q[0], ctx->prior_q[1], ctx->prior_q[2], ctx->prior_q[3]), (const int*)pb.p, ctx->prior_w, 0.0};
      ProfScope ps(ctx, LVX_FAM_PRIOR, st);
      hipLaunchKernelGGL((k_family<PriorFam, 1>), dim3(1), dim3(64), 0, st, p, cm, (const uint16_t*)ctx->d_pairs[2].p, (long long)ctx->fam_row0[2]);
    }
    // ---- LiDAR surfel blocks ----
    if (ctx->surf.n > 0) {
      ProfScope ps(ctx, LVX_FAM_SURFEL, st);
      if (fast_surf && tauL) {
        SurfAccT<true> f{ctx->surf.n, (const double*)ctx->surf.d_t.p, (const double*)ctx->surf.d_a3.p, (const double*)ctx->surf.d_b3.p, (const int*)ctx->surf.d_perm.p, ctx->t_map, ctx->surf.weight, ctx->surf.huber};
        LVX_LAUNCH_MFMA(SurfAccT<true>, f, LVX_FAM_SURFEL, st, ctx->fam_row0[3]);
        if (fb_fam(LVX_FAM_SURFEL)) { if ((rc = fb_fork())) return rc; ProfScope psf(ctx, LVX_KERNEL_FIXUP, s_fb); SurfFamT<true> ff{ctx->surf.n, (const double*)ctx->surf.d_t.p, (const double*)ctx->surf.d_a3.p, (const int*)ctx->surf.d_id0.p, (const int*)ctx->surf.d_perm.p, (const double*)ctx->d_planes.p, ctx->t_map, ctx->surf.weight, ctx->surf.huber};
          hipLaunchKernelGGL((k_family<SurfFamT<true>, LVX_PW>), fb_grid, dim3(64 * LVX_PW), 0, s_fb, ff, cm, (const uint16_t*)ctx->d_pairs[3].p, (long long)ctx->fam_row0[3], fb_rows(LVX_FAM_SURFEL), fb_cnt(LVX_FAM_SURFEL)); }
      } else if (fast_surf) {
        SurfAcc f{ctx->surf.n, (const double*)ctx->surf.d_t.p, (const double*)ctx->surf.d_a3.p, (const double*)ctx->surf.d_b3.p, (const int*)ctx->surf.d_perm.p, ctx->t_map, ctx->surf.weight, ctx->surf.huber};
        LVX_LAUNCH_MFMA(SurfAcc, f, LVX_FAM_SURFEL, st, ctx->fam_row0[3]);
        if (fb_fam(LVX_FAM_SURFEL)) { if ((rc = fb_fork())) return rc; ProfScope psf(ctx, LVX_KERNEL_FIXUP, s_fb); SurfFam ff{ctx->surf.n, (const double*)ctx->surf.d_t.p, (const double*)ctx->surf.d_a3.p, (const int*)ctx->surf.d_id0.p, (const int*)ctx->surf.d_perm.p, (const double*)ctx->d_planes.p, ctx->t_map, ctx->surf.weight, ctx->surf.huber};
          hipLaunchKernelGGL((k_family<SurfFam, LVX_PW>), fb_grid, dim3(64 * LVX_PW), 0, s_fb, ff, cm, (const uint16_t*)ctx->d_pairs[3].p, (long long)ctx->fam_row0[3], fb_rows(LVX_FAM_SURFEL), fb_cnt(LVX_FAM_SURFEL)); }
      } else if (tauL) {
        SurfFamT<true> f{ctx->surf.n, (const double*)ctx->surf.d_t.p, (const double*)ctx->surf.d_a3.p, (const int*)ctx->surf.d_id0.p, (const int*)ctx->surf.d_perm.p, (const double*)ctx->d_planes.p, ctx->t_map, ctx->surf.weight, ctx->surf.huber};
        hipLaunchKernelGGL((k_family<SurfFamT<true>, 1>), grid(f.n), dim3(64), 0, st, f, cm, (const uint16_t*)ctx->d_pairs[3].p, (long long)ctx->fam_row0[3]);
      } else {
        SurfFam f{ctx->surf.n, (const double*)ctx->surf.d_t.p, (const double*)ctx->surf.d_a3.p, (const int*)ctx->surf.d_id0.p, (const int*)ctx->surf.d_perm.p, (const double*)ctx->d_planes.p, ctx->t_map, ctx->surf.weight, ctx->surf.huber};
        hipLaunchKernelGGL((k_family<SurfFam, 1>), grid(f.n), dim3(64), 0, st, f, cm, (const uint16_t*)ctx->d_pairs[3].p, (long long)ctx->fam_row0[3]);
      }
    }
    // ---- camera-landmark-to-surfel blocks ----
    if (ctx->cs.n > 0) {
      ProfScope ps(ctx, LVX_FAM_CAMSURF, st);
      if (fast_cs && tauC) {
        CamSurfAccT<true> f{ctx->cs.n, (const int*)ctx->cs.d_id0.p, (const int*)ctx->cs.d_id1.p, (const int*)ctx->cs.d_perm.p, (const double*)ctx->d_planes.p, (const double*)ctx->d_lm_uv.p, (const double*)ctx->d_lm_t0.p, ctx->t_map, ctx->cs.weight, ctx->cs.huber};
        LVX_LAUNCH_MFMA(CamSurfAccT<true>, f, LVX_FAM_CAMSURF, st, ctx->fam_row0[5]);
        if (fb_fam(LVX_FAM_CAMSURF)) { if ((rc = fb_fork())) return rc; ProfScope psf(ctx, LVX_KERNEL_FIXUP, s_fb); CamSurfFamT<true> ff{ctx->cs.n, (const int*)ctx->cs.d_id0.p, (const int*)ctx->cs.d_id1.p, (const int*)ctx->cs.d_perm.p, (const double*)ctx->d_planes.p, (const double*)ctx->d_lm_uv.p, (const double*)ctx->d_lm_t0.p, ctx->t_map, ctx->cs.weight, ctx->cs.huber};
          hipLaunchKernelGGL((k_family<CamSurfFamT<true>, LVX_PW>), fb_grid, dim3(64 * LVX_PW), 0, s_fb, ff, cm, (const uint16_t*)ctx->d_pairs[5].p, (long long)ctx->fam_row0[5], fb_rows(LVX_FAM_CAMSURF), fb_cnt(LVX_FAM_CAMSURF)); }
      } else if (fast_cs) {
        CamSurfAcc f{ctx->cs.n, (const int*)ctx->cs.d_id0.p, (const int*)ctx->cs.d_id1.p, (const int*)ctx->cs.d_perm.p, (const double*)ctx->d_planes.p, (const double*)ctx->d_lm_uv.p, (const double*)ctx->d_lm_t0.p, ctx->t_map, ctx->cs.weight, ctx->cs.huber};
        LVX_LAUNCH_MFMA(CamSurfAcc, f, LVX_FAM_CAMSURF, st, ctx->fam_row0[5]);
        if (fb_fam(LVX_FAM_CAMSURF)) { if ((rc = fb_fork())) return rc; ProfScope psf(ctx, LVX_KERNEL_FIXUP, s_fb); CamSurfFam ff{ctx->cs.n, (const int*)ctx->cs.d_id0.p, (const int*)ctx->cs.d_id1.p, (const int*)ctx->cs.d_perm.p, (const double*)ctx->d_planes.p, (const double*)ctx->d_lm_uv.p, (const double*)ctx->d_lm_t0.p, ctx->t_map, ctx->cs.weight, ctx->cs.huber};
          hipLaunchKernelGGL((k_family<CamSurfFam, LVX_PW>), fb_grid, dim3(64 * LVX_PW), 0, s_fb, ff, cm, (const uint16_t*)ctx->d_pairs[5].p, (long long)ctx->fam_row0[5], fb_rows(LVX_FAM_CAMSURF), fb_cnt(LVX_FAM_CAMSURF)); }
      } else if (tauC) {
        CamSurfFamT<true> f{ctx->cs.n, (const int*)ctx->cs.d_id0.p, (const int*)ctx->cs.d_id1.p, (const int*)ctx->cs.d_perm.p, (const double*)ctx->d_planes.p, (const double*)ctx->d_lm_uv.p, (const double*)ctx->d_lm_t0.p, ctx->t_map, ctx->cs.weight, ctx->cs.huber};
        hipLaunchKernelGGL((k_family<CamSurfFamT<true>, 1>), grid(f.n), dim3(64), 0, st, f, cm, (const uint16_t*)ctx->d_pairs[5].p, (long long)ctx->fam_row0[5]);
      } else {
        CamSurfFam f{ctx->cs.n, (const int*)ctx->cs.d_id0.p, (const int*)ctx->cs.d_id1.p, (const int*)ctx->cs.d_perm.p, (const double*)ctx->d_planes.p, (const double*)ctx->d_lm_uv.p, (const double*)ctx->d_lm_t0.p, ctx->t_map, ctx->cs.weight, ctx->cs.huber};
        hipLaunchKernelGGL((k_family<CamSurfFam, 1>), grid(f.n), dim3(64), 0, st, f, cm, (const uint16_t*)ctx->d_pairs[5].p, (long long)ctx->fam_row0[5]);
      }
    }
    // ---- reprojection blocks ----
    if (ctx->rep.n > 0) {
      ReprojFam r{ctx->rep.n, (const int*)ctx->rep.d_id0.p, (const double*)ctx->rep.d_a3.p, (const double*)ctx->rep.d_t.p, (const int*)ctx->rep.d_perm.p,
                  (const double*)ctx->d_lm_uv.p, (const double*)ctx->d_lm_t0.p, ctx->rep.weight, ctx->rep.huber};
      // the fused path (Jacobian rows materialised once, three MFMA assembly passes, landmark rows stored) for a locked AND for a free camera time offset
      auto rep_fused = [&](auto TAUC) -> int {
        constexpr bool T = decltype(TAUC)::value;
        constexpr int RJ = REP_NC + (T ? 1 : 0);
        double* Jb = (double*)ctx->d_repB[0].p; double* rb = Jb + (size_t)2 * RJ * r.n; int* kb = (int*)ctx->d_repB[1].p;
        { ProfScope ps(ctx, LVX_KERNEL_REP_JAC, st);
          const ReprojFamT<T> rf{r.n, r.lm, r.uv, r.t0o, r.perm, r.lm_uv, r.lm_t0, r.weight, r.huber};
          double* trec = (ctx->L > 0 && !(ctx->locks & LVX_LOCK_LANDMARKS) && ctx->rep_groups > 0 && (what & LVX_EVAL_NORMAL_EQ)) ? (double*)ctx->d_repT.p : (double*)nullptr;
          hipLaunchKernelGGL(k_reproj_jac<T>, grid(r.n), dim3(64), trec ? (size_t)64 * ((56 + (T ? 1 : 0)) | 1) * 8 : 0, st, rf, cm, Jb, rb, kb, (long long)ctx->fam_row0[4], trec); }
        if (!(what & LVX_EVAL_NORMAL_EQ)) return LVX_OK;
        const RepJac jac{Jb, rb, kb, r.n};
        if (s_side != st) { LVX_HIP(ctx, hipEventRecord(ctx->ev_jac, st)); LVX_HIP(ctx, hipStreamWaitEvent(s_side, ctx->ev_jac, 0)); side_used = true; }
        { RepSideAcc<0, T> rr{r.n, jac, 0.0};
          ProfScope ps(ctx, LVX_KERNEL_REP_REF, s_side); LVX_LAUNCH_MFMA1(LVX_T2(RepSideAcc<0, T>), 1, rr, LVX_FAM_PRIOR, s_side, ctx->fam_row0[4]); }
        { RepSideAcc<1, T> ro{r.n, jac, 0.0};
          ProfScope ps(ctx, LVX_KERNEL_REP_OBS, st); LVX_LAUNCH_MFMA1(LVX_T2(RepSideAcc<1, T>), 1, ro, LVX_FAM_REPROJ, st, ctx->fam_row0[4]); }
        if (ctx->rep_groups > 0) {
          const int* gt = (const int*)ctx->d_repB[2].p;
          RepCross rx{jac, r.lm, gt, gt + ctx->rep_groups + 1, ctx->rep_groups, (double*)ctx->d_repT.p, nullptr};
          { ProfScope ps(ctx, LVX_KERNEL_REP_CROSS, st);
            if (det && ctx->det_cross_col.size() > 1) {
              for (size_t q = 0; q + 1 < ctx->det_cross_col.size(); ++q) {
                rx.det_list = (const int*)ctx->d_det_cross.p + ctx->det_cross_col[q];
                hipLaunchKernelGGL(k_reproj_cross<T>, dim3((unsigned)(ctx->det_cross_col[q + 1] - ctx->det_cross_col[q])), dim3(64 * RX_NW), 0, st, rx, cm);
              }
            } else
            hipLaunchKernelGGL(k_reproj_cross<T>, dim3((unsigned)std::min((ctx->rep_groups + RX_NW - 1) / RX_NW, 256 * 8)), dim3(64 * RX_NW), 0, st, rx, cm); }
          if (ctx->L > 0 && !(ctx->locks & LVX_LOCK_LANDMARKS)) {
            const int* lp = (const int*)ctx->d_repB[3].p;
            const RepLmRows lq{(const double*)ctx->d_repT.p, kb, r.n, lp, lp + ctx->L + 1, ctx->L};
            const size_t lds = (size_t)4 * ctx->lm_ls * 8;
            LVX_HIP(ctx, hipFuncSetAttribute((const void*)k_reproj_lmrows<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ProfScope ps(ctx, LVX_KERNEL_REP_LMROWS, st);
            hipLaunchKernelGGL(k_reproj_lmrows<T>, dim3((unsigned)((ctx->L + 3) / 4)), dim3(256), lds, st, lq, cm);
          }
        }
        return LVX_OK;
      };
      // ONE launch (round 6): evaluation, assembly and landmark rows per landmark-owning workgroup; nothing materialised
      auto rep_one = [&](auto TAUC) -> int {
        constexpr bool T = decltype(TAUC)::value;
        const ReprojFamT<T> rf{r.n, r.lm, r.uv, r.t0o, r.perm, r.lm_uv, r.lm_t0, r.weight, r.huber};
        const int* tab = (const int*)ctx->d_repF.p;
        const int ng = ctx->rep_fused_wg;
        const RepFused q{tab, tab + ng, ng};
        int ncu = 256;
        { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
        ProfScope ps(ctx, LVX_KERNEL_REP_FUSED, st);
        hipLaunchKernelGGL(k_reproj_fused<T>, dim3((unsigned)std::min((ng + 3) / 4, ncu)), dim3(256), 0, st, rf, q, cm, (long long)ctx->fam_row0[4]);
        return LVX_OK;
      };
      auto rep_fixup = [&]() {   // listed blocks by the exact kernel, behind the landmark rows' stores
        if (!fb_fam(LVX_FAM_REPROJ)) return;
        ProfScope psf(ctx, LVX_KERNEL_FIXUP, st);
        if (tauC) { ReprojFamT<true> rt{r.n, r.lm, r.uv, r.t0o, r.perm, r.lm_uv, r.lm_t0, r.weight, r.huber};
          hipLaunchKernelGGL((k_family<ReprojFamT<true>, LVX_PW>), fb_grid, dim3(64 * LVX_PW), 0, st, rt, cm, (const uint16_t*)ctx->d_pairs[4].p, (long long)ctx->fam_row0[4], fb_rows(LVX_FAM_REPROJ), fb_cnt(LVX_FAM_REPROJ)); }
        else hipLaunchKernelGGL((k_family<ReprojFam, LVX_PW>), fb_grid, dim3(64 * LVX_PW), 0, st, r, cm, (const uint16_t*)ctx->d_pairs[4].p, (long long)ctx->fam_row0[4], fb_rows(LVX_FAM_REPROJ), fb_cnt(LVX_FAM_REPROJ));
      };
      if (fast && ctx->rep_fused_wg > 0) {
        const int rcf = tauC ? rep_one(std::true_type{}) : rep_one(std::false_type{});
        if (rcf) return rcf;
        rep_fixup();
      } else if (fast) {
        const int rcf = tauC ? rep_fused(std::true_type{}) : rep_fused(std::false_type{});
        if (rcf) return rcf;
        rep_fixup();
      } else if (tauC) {
        ProfScope ps(ctx, LVX_FAM_REPROJ, st);
        ReprojFamT<true> rt{r.n, r.lm, r.uv, r.t0o, r.perm, r.lm_uv, r.lm_t0, r.weight, r.huber};
        hipLaunchKernelGGL((k_family<ReprojFamT<true>, LVX_PW>), grid(r.n), dim3(64 * LVX_PW), 0, st, rt, cm, (const uint16_t*)ctx->d_pairs[4].p, (long long)ctx->fam_row0[4]);
      } else {
        ProfScope ps(ctx, LVX_FAM_REPROJ, st);
        hipLaunchKernelGGL((k_family<ReprojFam, LVX_PW>), grid(r.n), dim3(64 * LVX_PW), 0, st, r, cm, (const uint16_t*)ctx->d_pairs[4].p, (long long)ctx->fam_row0[4]);
      }
    }
    if (side_used) { LVX_HIP(ctx, hipEventRecord(ctx->ev_join[0], s_side)); LVX_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_join[0], 0)); }
    if (fb_used) { LVX_HIP(ctx, hipEventRecord(ctx->ev_join[2], s_fb)); LVX_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_join[2], 0)); }
    { ProfScope ps(ctx, LVX_KERNEL_FOLD);
      const bool fold_fast = (what & LVX_EVAL_NORMAL_EQ) && (fast_surf || fast_cs);
      // replica sums -> dense block (the last replica block to finish) and the border-row fold (Bd, streaming; disjoint buffers) in one launch
      const unsigned n_rep_blk = (unsigned)((ctx->nbd_ext * ctx->nbd_ext + 255) / 256);
      if (fold_fast && ctx->nb > 0 && !sw.serial) {
        const unsigned n_rows_blk = (unsigned)((ctx->nb + 255) / 256);
        const int set0 = fast_surf ? 0 : 1, set1 = (fast_surf && fast_cs) ? 1 : -1;
        const size_t lds = ((size_t)ctx->nbd_ext * ctx->nbd_ext + ctx->nbd_ext) * 8;
        LVX_HIP(ctx, hipFuncSetAttribute((const void*)k_fold_all, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_fold_all, dim3(n_rep_blk + n_rows_blk), dim3(256), lds, st, cm, (int)n_rep_blk, (int)n_rows_blk, set0, 1, set1, 0, (int*)ctx->d_zero.p + 1);
      } else {
        hipStream_t s_fold = (fold_fast && side_on) ? ctx->fam_stream[0] : st;
        if (s_fold != st) { LVX_HIP(ctx, hipEventRecord(ctx->ev_fork, st)); LVX_HIP(ctx, hipStreamWaitEvent(s_fold, ctx->ev_fork, 0)); }
        hipLaunchKernelGGL(k_fold_replicas, dim3(n_rep_blk), dim3(256), 0, st, cm);
        if (fold_fast) {
          for (int set = 0; set < 2; ++set) if (ctx->nb > 0 && ((set == 0 && fast_surf) || (set == 1 && fast_cs)))
            hipLaunchKernelGGL(k_fold_border_rows, dim3((unsigned)((ctx->nb + 255) / 256)), dim3(256), 0, s_fold, cm, set, (set == 0 || !fast_surf) ? 1 : 0);
          const size_t lds = ((size_t)ctx->nbd_ext * ctx->nbd_ext + ctx->nbd_ext) * 8;
          LVX_HIP(ctx, hipFuncSetAttribute((const void*)k_fold_border_dense, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          hipLaunchKernelGGL(k_fold_border_dense, dim3(1), dim3(256), lds, st, cm);
          if (s_fold != st) { LVX_HIP(ctx, hipEventRecord(ctx->ev_join[1], s_fold)); LVX_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_join[1], 0)); }
        }
      } }
    LVX_HIP(ctx, hipGetLastError());
    return rc;
  };
  // graph replay pays for small problems (a pass is ~35 API calls: launch-bound at ~0.37 ms issued call by call, 0.26 ms replayed); at config-4 size the kernels are long
  // enough for the host to stay ahead and the replayed graph is the SLOWER one (0.61 against 0.58 ms: its cross-stream edges become barrier packets between every node)
  const bool use_graph = !ctx->sw.no_graph && !ctx->profiling && !(what & LVX_EVAL_JACOBIAN) && ctx->n_blocks <= 400000;
  if (!use_graph) { if ((rc = enqueue())) return rc; }
  else {
    const int flags = (want_res_buffer ? 1 : 0) | (ctx->force_legacy ? 2 : 0) | (ctx->fb_on ? 4 : 0) | (ctx->fb_mask << 3);   // a switch change bumps cfg_version
    hipGraphExec_t exec = nullptr;
    for (const auto& e : ctx->graphs) if (e.state == state_d && e.what == what && e.flags == flags && e.cfg == ctx->cfg_version) { exec = (hipGraphExec_t)e.exec; break; }
    if (!exec) {
      if (ctx->graphs.size() >= 16) { for (auto& e : ctx->graphs) (void)hipGraphExecDestroy((hipGraphExec_t)e.exec); ctx->graphs.clear(); }
      hipGraph_t graph = nullptr;
      LVX_HIP(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
      rc = enqueue();
      const hipError_t ce = hipStreamEndCapture(st, &graph);
      if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
      if (ce != hipSuccess || !graph) return fail(ctx, LVX_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
      const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ie != hipSuccess) return fail(ctx, LVX_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
      ctx->graphs.push_back({state_d, what, flags, ctx->cfg_version, (void*)exec});
    }
    LVX_HIP(ctx, hipGraphLaunch(exec, st));
  }
  ctx->last_what = what;
  ctx->last_state_d = state_d; ctx->last_want_res = want_res_buffer;
  ctx->err_unchecked = cost == nullptr;   // the device error word of this pass has not been looked at yet (check_last_eval)
  if (cost) {
    if (!ctx->pin) { LVX_HIP(ctx, hipHostMalloc((void**)&ctx->pin, 128 * 8, hipHostMallocDefault)); std::memset(ctx->pin, 0, 128 * 8); }
    int err[16] = {0};
    LVX_HIP(ctx, hipMemcpyAsync(ctx->pin, cm.cost, 8, hipMemcpyDeviceToHost, st));
    LVX_HIP(ctx, hipMemcpyAsync(ctx->pin + 1, cm.err, 4 * (4 + LVX_NUM_FAM), hipMemcpyDeviceToHost, st));
    if (ctx->before_eval_sync) ctx->before_eval_sync();   // (a pass repeated below calls it again: it then reads the repeated pass's results)
    LVX_HIP(ctx, hipStreamSynchronize(st));
    const double c = ctx->pin[0];
    std::memcpy(err, ctx->pin + 1, 4 * (4 + LVX_NUM_FAM));
    *cost = c;
    ctx->fallback_rows = 0; for (int f = 0; f < LVX_NUM_FAM; ++f) ctx->fallback_rows += err[4 + f];
    if ((err[0] & LVX_ERR_FALLBACK) && !ctx->force_legacy) {
      // rows the fused kernels cannot take exactly.  First the ROW-LEVEL fallback: the same pass with fallback lists — the fused kernels skip those rows, the exact
      // per-segment kernel evaluates just them (a few small launches more per pass from now on).  Only when the lists overflow, or for a corner that is not a row's
      // (|tau_imu| >= dt), everything goes to the per-segment kernels.
      int need = 0; for (int f = 0; f < LVX_NUM_FAM; ++f) if (err[4 + f] > 0) need |= 1 << f;
      if (!ctx->sw.force_legacy && need && (need & ~ctx->fb_mask)) { ctx->fb_on = true; ctx->fb_mask |= need; }   // a family without a list yet: give it one
      else ctx->force_legacy = true;                                                                              // a list overflowed, or the corner is not a row's
      return run_evaluate(ctx, state_d, what, cost, want_res_buffer);
    }
    if (err[0] & RES_RANGE) return fail(ctx, LVX_E_RANGE, "time span out of range for trajectory");
    if (err[0] & RES_NONUNIT) return fail(ctx, LVX_E_NONUNIT_QUAT, "logq: only implemented for unit quaternions");
    if (err[0] & 4) return fail(ctx, LVX_E_STATE, "normal-equation entry outside the computed bandwidth");
  }
  return LVX_OK;
}

// An evaluation queued without a cost pointer returns before its device error word exists.  Everything that consumes its normal equations
// on the host side (solve step, dense export) calls this first: a pass that met the merged-hub-segment corner is repeated with the exact
// per-segment kernels, a range / unit-quaternion error is returned instead of a step computed from incomplete sums.
int check_last_eval(lvx_ctx* c) {
  if (!c->err_unchecked || !c->d_err.p) return LVX_OK;
  int err = 0;
  LVX_HIP(c, hipMemcpyAsync(&err, c->d_err.p, 4, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  c->err_unchecked = false;
  if ((err & LVX_ERR_FALLBACK) && !c->force_legacy) {
    int errw[4 + LVX_NUM_FAM] = {0}, need = 0;
    LVX_HIP(c, hipMemcpy(errw, c->d_err.p, sizeof(errw), hipMemcpyDeviceToHost));
    for (int f = 0; f < LVX_NUM_FAM; ++f) if (errw[4 + f] > 0) need |= 1 << f;
    if (!c->sw.force_legacy && need && (need & ~c->fb_mask)) { c->fb_on = true; c->fb_mask |= need; } else c->force_legacy = true;
    double cost = 0;
    return run_evaluate(c, c->last_state_d, c->last_what, &cost, c->last_want_res);
  }
  if (err & RES_RANGE) return fail(c, LVX_E_RANGE, "time span out of range for trajectory");
  if (err & RES_NONUNIT) return fail(c, LVX_E_NONUNIT_QUAT, "logq: only implemented for unit quaternions");
  if (err & 4) return fail(c, LVX_E_STATE, "normal-equation entry outside the computed bandwidth");
  return LVX_OK;
}

}  // namespace lvx
