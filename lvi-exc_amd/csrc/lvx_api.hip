// lvx_api.hip — the evaluation half of the C ABI (include/lvx.h): context lifetime, the batched lvx_set_* inputs, switches, evaluation entry points, exports of the
// structured normal equations (dense form, gradient, border block, checksums), state transfer.  The kernels and the pass behind them are in lvx_eval.hip
// (ensure_layout, run_evaluate), the LM step in lvx_solver.hip / lvx_bcr.hip, the upstream kernels in lvx_upstream.hip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <cstdlib>

#include "lvx_ctx.h"

namespace lvx {

#define LVX_SYNC_EVENT_FLAGS(c) hipEventDisableTiming   // events that order kernels of this context's streams
const SwitchName* switch_table(int* count) {
  static const SwitchName tab[] = {
    {"FORCE_LEGACY", &Switches::force_legacy, false}, {"SERIAL", &Switches::serial, false}, {"NO_GRAPH", &Switches::no_graph, false}, {"DETERMINISTIC", &Switches::deterministic, true},
    {"CLEAR_ALL", &Switches::clear_all, false}, {"SOLVER_SEQ", &Switches::solver_seq, false}, {"SOLVER_TIMING", &Switches::solver_timing, false},
    {"CHUNK_R", &Switches::chunk_r, true}, {"CHUNK_R_IMU", &Switches::chunk_r_imu, true}, {"CHUNK_R_REP", &Switches::chunk_r_rep, true}, {"CHUNK_ROWS", &Switches::chunk_rows, true}, {"REP_ROWS", &Switches::rep_rows, true}, {"REP_FUSED", &Switches::rep_fused, true}, {"SOLVER_ND", &Switches::solver_nd, false},
    {"DA_SYNC", &Switches::da_sync, false},   // lvx_data_association: always the synchronous chain (four host stops), never the speculative one
  };
  *count = (int)(sizeof(tab) / sizeof(tab[0]));
  return tab;
}
static void read_env_switches(lvx_ctx* c) {
  int n; const SwitchName* t = switch_table(&n);
  for (int i = 0; i < n; ++i) {
    const std::string name = std::string("LVX_") + t[i].name;
    if (const char* e = std::getenv(name.c_str())) c->sw.*(t[i].field) = (*e == 0) ? 1 : atoi(e);
  }
}

// order-independent checksum of a buffer's bit patterns: sum over i of bits[i] * (2 i + 1) mod 2^64 (integer atomics)
__global__ void k_checksum(const unsigned long long* p, size_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long b = p[i];
    if (b == 0x8000000000000000ull) b = 0;   // -0.0 == +0.0
    acc += b * (2ull * i + 1ull);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
__global__ void k_export_tail(const double* cost, const int* err, double* out) { out[0] = cost[0]; out[1] = (double)err[0]; }

}  // namespace lvx

using namespace lvx;

extern "C" {

const char* lvx_version(void) { return "lvx 0.1 (gfx950)"; }
const char* lvx_last_error(const lvx_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int lvx_create(lvx_ctx** out, int device, uint32_t /*flags*/) {
  if (!out) return LVX_E_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LVX_E_NODEVICE;
  if (device < 0 || device >= ndev) return LVX_E_ARG;
  if (hipSetDevice(device) != hipSuccess) return LVX_E_HIP;
  lvx_ctx* c = new (std::nothrow) lvx_ctx();
  if (!c) return LVX_E_ALLOC;
  c->device = device;
  read_env_switches(c);
  if (hipStreamCreate(&c->own_stream) != hipSuccess) { delete c; return LVX_E_HIP; }
  c->stream = c->own_stream;
  for (int k = 0; k < 4; ++k) { if (hipStreamCreateWithFlags(&c->fam_stream[k], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_join[k], LVX_SYNC_EVENT_FLAGS(c)) != hipSuccess) { delete c; return LVX_E_HIP; } }
  if (hipEventCreateWithFlags(&c->ev_fork, LVX_SYNC_EVENT_FLAGS(c)) != hipSuccess) { delete c; return LVX_E_HIP; }
  if (hipEventCreateWithFlags(&c->ev_jac, LVX_SYNC_EVENT_FLAGS(c)) != hipSuccess) { delete c; return LVX_E_HIP; }
  for (int k = 0; k < 2; ++k) if (hipEventCreateWithFlags(&c->ev_fb[k], LVX_SYNC_EVENT_FLAGS(c)) != hipSuccess) { delete c; return LVX_E_HIP; }
  *out = c;
  return LVX_OK;
}

static void free_family(Family& f) { for (DevBuf* b : {&f.d_t, &f.d_a3, &f.d_b3, &f.d_id0, &f.d_id1, &f.d_perm}) if (b->p) (void)hipFree(b->p); }

void lvx_destroy(lvx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  free_family(c->imu); free_family(c->surf); free_family(c->rep); free_family(c->cs);
  for (DevBuf* b : {&c->d_planes, &c->d_lm_uv, &c->d_lm_t0, &c->d_ord, &c->d_Hb, &c->d_gb, &c->d_Bd, &c->d_C, &c->d_gc, &c->d_cost, &c->d_err, &c->d_state,
                    &c->d_res, &c->d_jcols, &c->d_jvals, &c->d_L, &c->d_Y, &c->d_S, &c->d_delta, &c->d_diag, &c->d_scal, &c->d_state_try, &c->d_zero})
    if (b->p) (void)hipFree(b->p);
  for (auto& b : c->d_pairs) if (b.p) (void)hipFree(b.p);
  for (auto& b : c->d_up) if (b.p) (void)hipFree(b.p);
  for (auto& b : c->d_assoc) if (b.p) (void)hipFree(b.p);
  for (auto& b : c->d_da) if (b.p) (void)hipFree(b.p);
  for (DevBuf* b : {&c->d_da_key, &c->d_da_aux}) if (b->p) (void)hipFree(b->p);
  for (auto& b : c->d_chunk) if (b.p) (void)hipFree(b.p);
  for (auto& b : c->d_repB) if (b.p) (void)hipFree(b.p);
  if (c->d_hubs.p) (void)hipFree(c->d_hubs.p);
  if (c->d_pre.p) (void)hipFree(c->d_pre.p);
  if (c->d_colfull.p) (void)hipFree(c->d_colfull.p);
  for (auto& b : c->d_det_list) if (b.p) (void)hipFree(b.p);
  for (DevBuf* b : {&c->d_imu_rtab, &c->d_chk, &c->d_det_cross, &c->d_lm_grp, &c->d_lmH, &c->d_lm_p0, &c->d_Hr, &c->d_Br, &c->d_red, &c->d_repT, &c->d_repF, &c->d_fb}) if (b->p) (void)hipFree(b->p);
  if (c->vox.graph) (void)hipGraphExecDestroy((hipGraphExec_t)c->vox.graph);
  if (c->vox.h_info) (void)hipHostFree(c->vox.h_info);
  if (c->da_pinned) (void)hipHostFree(c->da_pinned);
  if (c->pin) (void)hipHostFree(c->pin);
  for (DevBuf* b : {&c->vox.misc, &c->vox.keys, &c->vox.vals, &c->vox.runs, &c->vox.cells, &c->vox.tmp, &c->vox.leaf_i, &c->vox.leaf_d, &c->vox.leaf_f}) if (b->p) (void)hipFree(b->p);
  (void)lvx_rccl_finalize(c);
  if (c->d_comm.p) (void)hipFree(c->d_comm.p);
  bcr_destroy(c);
  for (DevBuf* b : {&c->d_bcrD, &c->d_bcrG, &c->d_bcrInfo, &c->d_Y2, &c->d_gram, &c->d_bcrLinv}) if (b->p) (void)hipFree(b->p);
  for (auto& e : c->graphs) (void)hipGraphExecDestroy((hipGraphExec_t)e.exec);
  for (auto e : c->ev_pool) (void)hipEventDestroy(e);
  for (int k = 0; k < 4; ++k) { if (c->fam_stream[k]) (void)hipStreamDestroy(c->fam_stream[k]); if (c->ev_join[k]) (void)hipEventDestroy(c->ev_join[k]); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_jac) (void)hipEventDestroy(c->ev_jac);
  for (int k = 0; k < 2; ++k) if (c->ev_fb[k]) (void)hipEventDestroy(c->ev_fb[k]);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int lvx_set_spline(lvx_ctx* c, double t0, double dt, int n_knots) { if (c) c->cfg_version++;
  if (!c || !(dt > 0) || n_knots < 4) return c ? fail(c, LVX_E_ARG, "spline needs dt > 0 and >= 4 control points (spline_base.h:58-62)") : LVX_E_ARG;
  c->t0 = t0; c->dt = dt; c->N = n_knots; c->have_spline = true; c->layout_dirty = true; return LVX_OK;
}
int lvx_set_camera(lvx_ctx* c, const lvx_pinhole* p) { if (c) c->cfg_version++;
  if (!c || !p) return LVX_E_ARG;
  CamIntr& k = c->cam;
  k.fx = p->fx; k.fy = p->fy; k.cx = p->cx; k.cy = p->cy; k.k1 = p->k1; k.k2 = p->k2; k.p1 = p->p1; k.p2 = p->p2; k.k3 = p->k3; k.readout = p->readout;
  k.rows = p->rows; k.cols = p->cols;
  k.inv_K11 = 1.0 / k.fx; k.inv_K13 = -k.cx / k.fx; k.inv_K22 = 1.0 / k.fy; k.inv_K23 = -k.cy / k.fy;   // pinhole_camera.h:59-75
  k.do_distortion = (std::fabs(k.k1) > 1e-5 || std::fabs(k.k2) > 1e-5 || std::fabs(k.p1) > 1e-5 || std::fabs(k.p1) > 1e-5) ? 1 : 0;  // sic (:78)
  c->layout_dirty = true; return LVX_OK;
}
int lvx_set_imu(lvx_ctx* c, int n, const double* t, const double* gyro3, const double* acc3, double w_gyro, double w_acc) { if (c) c->cfg_version++;
  if (!c || n < 0 || (n > 0 && (!t || !gyro3 || !acc3))) return LVX_E_ARG;
  Family& f = c->imu; f.n = n; f.t.assign(t, t + n); f.a3.assign(gyro3, gyro3 + 3 * (size_t)n); f.b3.assign(acc3, acc3 + 3 * (size_t)n);
  f.weight = w_gyro; f.huber = w_acc;   // (huber slot reused for the accelerometer weight; IMU blocks have no loss function)
  c->layout_dirty = true; return LVX_OK;
}
int lvx_set_orientation_prior(lvx_ctx* c, int enable, double t, const double* q_wxyz, double weight) { if (c) c->cfg_version++;
  if (!c) return LVX_E_ARG;
  c->has_prior = enable != 0; c->prior_t = t; if (q_wxyz) std::memcpy(c->prior_q, q_wxyz, 32); c->prior_w = weight; c->layout_dirty = true; return LVX_OK;
}
int lvx_set_planes(lvx_ctx* c, int n, const double* pi3) { if (c) c->cfg_version++;
  if (!c || n < 0 || (n > 0 && !pi3)) return LVX_E_ARG;
  c->planes.assign(pi3, pi3 + 3 * (size_t)n); c->layout_dirty = true; return LVX_OK;
}
int lvx_set_surfel(lvx_ctx* c, int n, const double* pt3, const double* t, const int32_t* plane_id, double t_map, double huber, double weight) { if (c) c->cfg_version++;
  if (!c || n < 0 || (n > 0 && (!pt3 || !t || !plane_id))) return LVX_E_ARG;
  for (int i = 0; i < n; ++i) if (plane_id[i] < 0 || (size_t)plane_id[i] * 3 >= c->planes.size()) return fail(c, LVX_E_ARG, "plane id out of range (call lvx_set_planes first)");
  Family& f = c->surf; f.n = n; f.t.assign(t, t + n); f.a3.assign(pt3, pt3 + 3 * (size_t)n); f.id0.assign(plane_id, plane_id + n);
  f.huber = huber; f.weight = weight; c->t_map = t_map; c->layout_dirty = true; return LVX_OK;
}
int lvx_set_landmarks(lvx_ctx* c, int n, const double* uv_ref2, const double* t0_ref) { if (c) c->cfg_version++;
  if (!c || n < 0 || (n > 0 && (!uv_ref2 || !t0_ref))) return LVX_E_ARG;
  c->L = n; c->lm_uv.assign(uv_ref2, uv_ref2 + 2 * (size_t)n); c->lm_t0.assign(t0_ref, t0_ref + n); c->layout_dirty = true; return LVX_OK;
}
int lvx_set_reproj(lvx_ctx* c, int n, const int32_t* lm, const double* uv_obs2, const double* t0_obs, double huber, double weight) { if (c) c->cfg_version++;
  if (!c || n < 0 || (n > 0 && (!lm || !uv_obs2 || !t0_obs))) return LVX_E_ARG;
  Family& f = c->rep; f.n = n; f.t.assign(t0_obs, t0_obs + n); f.a3.assign(uv_obs2, uv_obs2 + 2 * (size_t)n); f.id0.assign(lm, lm + n);
  f.huber = huber; f.weight = weight; c->layout_dirty = true; return LVX_OK;
}
int lvx_set_camsurf(lvx_ctx* c, int n, const int32_t* lm, const int32_t* plane_id, double t_map, double huber, double weight) { if (c) c->cfg_version++;
  if (!c || n < 0 || (n > 0 && (!lm || !plane_id))) return LVX_E_ARG;
  for (int i = 0; i < n; ++i) {
    if (plane_id[i] < 0 || (size_t)plane_id[i] * 3 >= c->planes.size()) return fail(c, LVX_E_ARG, "plane id out of range (call lvx_set_planes first)");
    if (lm[i] < 0 || lm[i] >= c->L) return fail(c, LVX_E_ARG, "landmark id out of range (call lvx_set_landmarks first)");
  }
  Family& f = c->cs; f.n = n; f.id0.assign(lm, lm + n); f.id1.assign(plane_id, plane_id + n);
  f.huber = huber; f.weight = weight; c->t_map = t_map; c->layout_dirty = true; return LVX_OK;
}
int lvx_set_switch(lvx_ctx* c, const char* name, int value) {
  if (!c || !name) return LVX_E_ARG;
  if (!strncmp(name, "LVX_", 4)) name += 4;
  int n; const SwitchName* t = switch_table(&n);
  for (int i = 0; i < n; ++i) if (!strcmp(t[i].name, name)) {
    c->sw.*(t[i].field) = value; c->cfg_version++;   // captured graphs are stale
    if (t[i].relayout) c->layout_dirty = true;
    return LVX_OK;
  }
  return fail(c, LVX_E_ARG, std::string("unknown switch ") + name);
}
int lvx_set_locks(lvx_ctx* c, uint32_t mask) { if (c) c->cfg_version++; if (!c) return LVX_E_ARG; if (c->locks != mask) { c->locks = mask; c->layout_dirty = true; } return LVX_OK; }
int lvx_set_time_offset_bounds(lvx_ctx* c, double imu_max, double sensor_max) { if (c) c->cfg_version++; if (!c) return LVX_E_ARG; c->imu_mto = imu_max; c->sensor_mto = sensor_max; c->layout_dirty = true; return LVX_OK; }

int lvx_state_size(const lvx_ctx* c) { return c ? 7 * c->N + 32 + c->L : 0; }
int lvx_tangent_size(const lvx_ctx* c) { return c ? 6 * c->N + 22 + c->L : 0; }

int lvx_get_layout(lvx_ctx* c, lvx_layout* o) {
  if (!c || !o) return LVX_E_ARG;
  int rc = ensure_layout(c); if (rc) return rc;
  o->n_knots = c->N; o->n_landmarks = c->L; o->n_tangent = lvx_tangent_size(c); o->n_band = c->nb; o->bandwidth = c->bw; o->n_border = c->nbd; o->border_ld = c->nbd_ext;
  o->n_hub_knots = c->n_hub; o->hub_knot0 = c->hub0; o->n_blocks = c->n_blocks; o->n_residuals = c->n_residuals;
  o->exact_fallback = c->force_legacy ? 1 : 0; o->solver_fallbacks = c->solver_fallbacks; o->fallback_rows = c->fallback_rows;
  nd_counts(c, &o->solver_separators, &o->solver_leaves);
  return LVX_OK;
}

int lvx_get_family_rows(lvx_ctx* c, int64_t row0[LVX_NUM_FAM + 1]) {
  if (!c || !row0) return LVX_E_ARG;
  int rc = ensure_layout(c); if (rc) return rc;
  for (int f = 0; f <= LVX_NUM_FAM; ++f) row0[f] = c->fam_row0[f];
  return LVX_OK;
}

int lvx_evaluate_d(lvx_ctx* c, const double* state_d, uint32_t what, double* cost) {
  if (!c) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  if (!state_d) { int rc = ensure_layout(c); if (rc) return rc; state_d = (const double*)c->d_state.p; }
  return run_evaluate(c, state_d, what, cost, false);
}

int lvx_evaluate(lvx_ctx* c, const double* state, uint32_t what, double* cost, double* residuals) {
  if (!c || !state) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  int rc = ensure_layout(c); if (rc) return rc;
  LVX_HIP(c, hipMemcpyAsync(c->d_state.p, state, (size_t)lvx_state_size(c) * 8, hipMemcpyHostToDevice, c->stream));
  double cst = 0;
  const bool want_res = residuals != nullptr || (what & LVX_EVAL_RESIDUALS);
  rc = run_evaluate(c, (const double*)c->d_state.p, what, &cst, want_res);
  if (cost) *cost = cst;
  if (rc) return rc;
  if (residuals && c->n_residuals > 0) {
    LVX_HIP(c, hipMemcpyAsync(residuals, c->d_res.p, (size_t)c->n_residuals * 8, hipMemcpyDeviceToHost, c->stream));
    LVX_HIP(c, hipStreamSynchronize(c->stream));
  }
  return LVX_OK;
}

int lvx_set_stream(lvx_ctx* c, void* s) {
  if (!c) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return LVX_OK;
}
int lvx_export_border_d(lvx_ctx* c, double* out_d) {
  if (!c || !out_d) return LVX_E_ARG;
  if (!(c->last_what & LVX_EVAL_NORMAL_EQ)) return fail(c, LVX_E_STATE, "last evaluation did not request LVX_EVAL_NORMAL_EQ");
  LVX_HIP(c, hipSetDevice(c->device));
  const size_t n2 = (size_t)c->nbd_ext * c->nbd_ext;
  LVX_HIP(c, hipMemcpyAsync(out_d, c->d_C.p, n2 * 8, hipMemcpyDeviceToDevice, c->stream));
  LVX_HIP(c, hipMemcpyAsync(out_d + n2, c->d_gc.p, (size_t)c->nbd_ext * 8, hipMemcpyDeviceToDevice, c->stream));
  // cost and the pass's device error word (as a double: 0 = clean) ride along, so that after the all-reduce EVERY rank sees whether any
  // rank's sums are incomplete — without a host synchronisation here
  hipLaunchKernelGGL(k_export_tail, dim3(1), dim3(1), 0, c->stream, (const double*)c->d_cost.p, (const int*)c->d_err.p, out_d + n2 + c->nbd_ext);
  LVX_HIP(c, hipGetLastError());
  return LVX_OK;
}
int lvx_normal_eq_checksum(lvx_ctx* c, uint64_t out[6]) {
  if (!c || !out) return LVX_E_ARG;
  if (!(c->last_what & LVX_EVAL_NORMAL_EQ)) return fail(c, LVX_E_STATE, "last evaluation did not request LVX_EVAL_NORMAL_EQ");
  LVX_HIP(c, hipSetDevice(c->device));
  int rc = check_last_eval(c); if (rc) return rc;
  if ((rc = dev_alloc(c, c->d_chk, 64))) return rc;
  LVX_HIP(c, hipMemsetAsync(c->d_chk.p, 0, 48, c->stream));
  const size_t nb1 = (size_t)std::max(c->nb, 1);
  const bool lm = c->L > 0 && c->rep.n > 0 && !(c->locks & LVX_LOCK_LANDMARKS);
  const void* bufs[6] = {c->d_Hb.p, c->d_gb.p, c->d_Bd.p, c->d_C.p, c->d_gc.p, lm ? c->d_lmH.p : nullptr};
  const size_t cnt[6] = {nb1 * (c->bw + 1), nb1, (size_t)c->nbd * nb1, (size_t)c->nbd_ext * c->nbd_ext, (size_t)c->nbd_ext, lm ? (size_t)c->L * c->lm_ls : 0};
  for (int i = 0; i < 6; ++i) if (bufs[i] && cnt[i])
    hipLaunchKernelGGL(k_checksum, dim3((unsigned)std::min<size_t>((cnt[i] + 255) / 256, 4096)), dim3(256), 0, c->stream, (const unsigned long long*)bufs[i], cnt[i], (unsigned long long*)c->d_chk.p + i);
  LVX_HIP(c, hipMemcpyAsync(out, c->d_chk.p, 48, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}
int lvx_set_state(lvx_ctx* c, const double* state) {
  if (!c || !state) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  int rc = ensure_layout(c); if (rc) return rc;
  LVX_HIP(c, hipMemcpyAsync(c->d_state.p, state, (size_t)lvx_state_size(c) * 8, hipMemcpyHostToDevice, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}
int lvx_get_state(lvx_ctx* c, double* out) {
  if (!c || !out) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  int rc = ensure_layout(c); if (rc) return rc;
  LVX_HIP(c, hipMemcpyAsync(out, c->d_state.p, (size_t)lvx_state_size(c) * 8, hipMemcpyDeviceToHost, c->stream));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  return LVX_OK;
}
int lvx_synchronize(lvx_ctx* c) {
  if (!c) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  if (!c->d_err.p) return LVX_OK;
  int errw[4 + LVX_NUM_FAM] = {0};
  LVX_HIP(c, hipMemcpy(errw, c->d_err.p, sizeof(errw), hipMemcpyDeviceToHost));
  const int err = errw[0];
  c->err_unchecked = false;
  c->fallback_rows = 0; for (int f = 0; f < LVX_NUM_FAM; ++f) c->fallback_rows += errw[4 + f];
  if (err & LVX_ERR_FALLBACK) {
    { int need = 0; for (int f = 0; f < LVX_NUM_FAM; ++f) if (errw[4 + f] > 0) need |= 1 << f;
      if (!c->sw.force_legacy && need && (need & ~c->fb_mask)) { c->fb_on = true; c->fb_mask |= need; } else c->force_legacy = true; }   // row-level fallback lists first, the per-segment kernels for everything after that
    return fail(c, LVX_E_STATE, "fused assembly kernels met rows only the per-segment kernels evaluate exactly: evaluate again (the exact fallback is now selected)");
  }
  if (err & RES_RANGE) return fail(c, LVX_E_RANGE, "time span out of range for trajectory");
  if (err & RES_NONUNIT) return fail(c, LVX_E_NONUNIT_QUAT, "logq: only implemented for unit quaternions");
  if (err & 4) return fail(c, LVX_E_STATE, "normal-equation entry outside the computed bandwidth");
  return LVX_OK;
}
int lvx_set_profiling(lvx_ctx* c, int enable) { if (!c) return LVX_E_ARG; c->profiling = enable != 0; c->profile_only = enable >= 2 ? enable - 2 : -1; return LVX_OK; }
int lvx_get_kernel_ms(lvx_ctx* c, double* ms_sum, int64_t* launches) {
  if (!c || !ms_sum || !launches) return LVX_E_ARG;
  LVX_HIP(c, hipSetDevice(c->device));
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < LVX_NUM_KERNELS; ++k) { ms_sum[k] = 0.0; launches[k] = 0; }
  for (const auto& r : c->ev_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_pool[r.e0], c->ev_pool[r.e1]) == hipSuccess && r.kernel >= 0 && r.kernel < LVX_NUM_KERNELS) { ms_sum[r.kernel] += ms; launches[r.kernel] += 1; }
  }
  c->ev_recs.clear(); c->ev_used = 0;
  return LVX_OK;
}

int lvx_get_jacobian(lvx_ctx* c, int32_t* cols, double* vals) {
  if (!c || !cols || !vals) return LVX_E_ARG;
  if (!(c->last_what & LVX_EVAL_JACOBIAN)) return fail(c, LVX_E_STATE, "last evaluation did not request LVX_EVAL_JACOBIAN");
  LVX_HIP(c, hipSetDevice(c->device));
  const size_t n = (size_t)c->n_residuals * LVX_JAC_WIDTH;
  LVX_HIP(c, hipMemcpy(cols, c->d_jcols.p, n * 4, hipMemcpyDeviceToHost));
  LVX_HIP(c, hipMemcpy(vals, c->d_jvals.p, n * 8, hipMemcpyDeviceToHost));
  return LVX_OK;
}

int lvx_get_normal_eq_dense(lvx_ctx* c, double* H, double* g) {
  if (!c || !H || !g) return LVX_E_ARG;
  if (!(c->last_what & LVX_EVAL_NORMAL_EQ)) return fail(c, LVX_E_STATE, "last evaluation did not request LVX_EVAL_NORMAL_EQ");
  const int nt = lvx_tangent_size(c);
  if (nt > 20000) return fail(c, LVX_E_ARG, "dense expansion is a parity/debug path for small problems");
  LVX_HIP(c, hipSetDevice(c->device));
  { const int rce = check_last_eval(c); if (rce) return rce; }
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  const int nb = c->nb, bw = c->bw, nbd = c->nbd, ldc = c->nbd_ext;
  std::vector<double> Hb((size_t)std::max(nb, 1) * (bw + 1)), gb(std::max(nb, 1)), Bd((size_t)nbd * std::max(nb, 1)), C((size_t)ldc * ldc), gc(nbd);
  LVX_HIP(c, hipMemcpy(Hb.data(), c->d_Hb.p, Hb.size() * 8, hipMemcpyDeviceToHost));
  LVX_HIP(c, hipMemcpy(gb.data(), c->d_gb.p, gb.size() * 8, hipMemcpyDeviceToHost));
  LVX_HIP(c, hipMemcpy(Bd.data(), c->d_Bd.p, Bd.size() * 8, hipMemcpyDeviceToHost));
  LVX_HIP(c, hipMemcpy(C.data(), c->d_C.p, C.size() * 8, hipMemcpyDeviceToHost));
  LVX_HIP(c, hipMemcpy(gc.data(), c->d_gc.p, gc.size() * 8, hipMemcpyDeviceToHost));
  std::memset(H, 0, sizeof(double) * (size_t)nt * nt);
  std::memset(g, 0, sizeof(double) * nt);
  std::vector<int> band_var(std::max(nb, 1), -1), bord_var(nbd, -1);
  for (int v = 0; v < nt; ++v) { const int o = c->ord[v]; if (o == LVX_DEAD || o >= LVX_LM_BASE) continue; if (o >= 0) band_var[o] = v; else if (-1 - o < nbd) bord_var[-1 - o] = v; }
  for (int j = 0; j < nb; ++j) {
    g[band_var[j]] = gb[j];
    for (int d = 0; d <= bw && j + d < nb; ++d) { const double v = Hb[(size_t)j * (bw + 1) + d]; const int a = band_var[j + d], b = band_var[j]; H[(size_t)a * nt + b] = v; H[(size_t)b * nt + a] = v; }
  }
  for (int b = 0; b < nbd; ++b) {
    if (bord_var[b] < 0) continue;
    g[bord_var[b]] = gc[b];
    for (int j = 0; j < nb; ++j) { const double v = Bd[(size_t)b * nb + j]; H[(size_t)bord_var[b] * nt + band_var[j]] = v; H[(size_t)band_var[j] * nt + bord_var[b]] = v; }
    for (int b2 = 0; b2 <= b; ++b2) { if (bord_var[b2] < 0) continue; const double v = C[(size_t)b * ldc + b2]; H[(size_t)bord_var[b] * nt + bord_var[b2]] = v; H[(size_t)bord_var[b2] * nt + bord_var[b]] = v; }
  }
  if (c->L > 0 && c->rep.n > 0 && !(c->locks & LVX_LOCK_LANDMARKS)) {   // landmark rows
    const int wl = c->lm_wl, ls = c->lm_ls;
    std::vector<double> R((size_t)c->L * ls); std::vector<int> p0(c->L);
    LVX_HIP(c, hipMemcpy(R.data(), c->d_lmH.p, R.size() * 8, hipMemcpyDeviceToHost));
    LVX_HIP(c, hipMemcpy(p0.data(), c->d_lm_p0.p, p0.size() * 4, hipMemcpyDeviceToHost));
    for (int l = 0; l < c->L; ++l) {
      const double* row = &R[(size_t)l * ls];
      const int vl = 6 * c->N + 22 + l;
      H[(size_t)vl * nt + vl] = row[wl + ldc]; g[vl] = row[wl + ldc + 1];
      for (int k = 0; k < wl; ++k) if (row[k] != 0.0 && p0[l] + k < nb) { const int vk = band_var[p0[l] + k]; H[(size_t)vl * nt + vk] = row[k]; H[(size_t)vk * nt + vl] = row[k]; }
      for (int b = 0; b < nbd; ++b) if (row[wl + b] != 0.0 && bord_var[b] >= 0) { H[(size_t)vl * nt + bord_var[b]] = row[wl + b]; H[(size_t)bord_var[b] * nt + vl] = row[wl + b]; }
    }
  }
  return LVX_OK;
}

// g = J^T r and diag(J^T J) of the last LVX_EVAL_NORMAL_EQ evaluation in the tangent layout, any problem size (constant scalars: 0)
int lvx_get_gradient(lvx_ctx* c, double* g, double* diag) {
  if (!c || (!g && !diag)) return LVX_E_ARG;
  if (!(c->last_what & LVX_EVAL_NORMAL_EQ)) return fail(c, LVX_E_STATE, "last evaluation did not request LVX_EVAL_NORMAL_EQ");
  LVX_HIP(c, hipSetDevice(c->device));
  { const int rce = check_last_eval(c); if (rce) return rce; }
  LVX_HIP(c, hipStreamSynchronize(c->stream));
  const int nt = lvx_tangent_size(c), nb = c->nb, bw = c->bw, nbd = c->nbd, ldc = c->nbd_ext;
  std::vector<double> gb(std::max(nb, 1)), hd(std::max(nb, 1)), C((size_t)ldc * ldc), gc(std::max(nbd, 1));
  if (nb > 0) {
    LVX_HIP(c, hipMemcpy(gb.data(), c->d_gb.p, (size_t)nb * 8, hipMemcpyDeviceToHost));
    LVX_HIP(c, hipMemcpy2D(hd.data(), 8, c->d_Hb.p, (size_t)(bw + 1) * 8, 8, (size_t)nb, hipMemcpyDeviceToHost));   // Hb[j][0]: the diagonal of the lower band
  }
  if (nbd > 0) {
    LVX_HIP(c, hipMemcpy(C.data(), c->d_C.p, C.size() * 8, hipMemcpyDeviceToHost));
    LVX_HIP(c, hipMemcpy(gc.data(), c->d_gc.p, (size_t)nbd * 8, hipMemcpyDeviceToHost));
  }
  const bool lm = c->L > 0 && c->rep.n > 0 && !(c->locks & LVX_LOCK_LANDMARKS);
  std::vector<double> R;
  if (lm) { R.resize((size_t)c->L * c->lm_ls); LVX_HIP(c, hipMemcpy(R.data(), c->d_lmH.p, R.size() * 8, hipMemcpyDeviceToHost)); }
  for (int v = 0; v < nt; ++v) {
    const int o = c->ord[v];
    double gv = 0.0, dv = 0.0;
    if (o == LVX_DEAD) { }
    else if (o >= LVX_LM_BASE) { if (lm) { const double* row = &R[(size_t)(o - LVX_LM_BASE) * c->lm_ls]; dv = row[c->lm_wl + ldc]; gv = row[c->lm_wl + ldc + 1]; } }
    else if (o >= 0) { gv = gb[o]; dv = hd[o]; }
    else if (-1 - o < nbd) { gv = gc[-1 - o]; dv = C[(size_t)(-1 - o) * ldc + (-1 - o)]; }
    if (g) g[v] = gv;
    if (diag) diag[v] = dv;
  }
  return LVX_OK;
}

int lvx_plus(lvx_ctx* c, const double* s, const double* d, double* o) {
  if (!c || !s || !d || !o) return LVX_E_ARG;
  const int N = c->N;
  std::memcpy(o, s, sizeof(double) * (size_t)lvx_state_size(c));
  auto qplus = [](const double* x, const double* dl, double* out) {   // ceres::EigenQuaternionParameterization::Plus (restated)
    const double nd = std::sqrt(dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2]);
    if (nd > 0.0) {
      const double sd = std::sin(nd) / nd;
      const quat r = qmul(mkq(std::cos(nd), sd * dl[0], sd * dl[1], sd * dl[2]), load_q(x));
      out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
    } else { for (int k = 0; k < 4; ++k) out[k] = x[k]; }
  };
  for (int k = 0; k < N; ++k) {
    for (int j = 0; j < 3; ++j) o[3 * k + j] = s[3 * k + j] + d[6 * k + j];
    qplus(s + 3 * N + 4 * k, d + 6 * k + 3, o + 3 * N + 4 * k);
  }
  const double* si = s + 7 * N; double* oi = o + 7 * N; const double* di = d + 6 * N;
  oi[8] = si[8] + di[0]; oi[9] = si[9] + di[1];
  for (int j = 0; j < 3; ++j) { oi[10 + j] = si[10 + j] + di[2 + j]; oi[13 + j] = si[13 + j] + di[5 + j]; }
  qplus(si + 16, di + 8, oi + 16); for (int j = 0; j < 3; ++j) oi[20 + j] = si[20 + j] + di[11 + j]; oi[23] = si[23] + di[14];
  qplus(si + 24, di + 15, oi + 24); for (int j = 0; j < 3; ++j) oi[28 + j] = si[28 + j] + di[18 + j]; oi[31] = si[31] + di[21];
  for (int l = 0; l < c->L; ++l) oi[32 + l] = si[32 + l] + di[22 + l];
  // box constraints, applied by projection as ceres::ParameterBlock::Plus does (see k_plus)
  if (!(c->locks & LVX_LOCK_LIDAR_TAU)) oi[23] = std::min(std::max(oi[23], -c->sensor_mto), c->sensor_mto);
  if (!(c->locks & LVX_LOCK_CAM_TAU)) oi[31] = std::min(std::max(oi[31], -c->sensor_mto), c->sensor_mto);
  if (!(c->locks & LVX_LOCK_LANDMARKS)) for (int l = 0; l < c->L; ++l) oi[32 + l] = std::max(oi[32 + l], 0.0);
  return LVX_OK;
}

}  // extern "C"
