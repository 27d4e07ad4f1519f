// lvx_ctx.h — context shared by the translation units of liblvx.so (evaluator, solver, upstream kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <array>
#include <string>
#include <functional>
#include <vector>

#include "../../include/lvx.h"
#include "lvx_resid.h"

#define LVX_NREP 64                 // replicas of the dense border accumulators (spreads same-address atomics)
#define LVX_DEAD (-2147483647 - 1)  // ord[] value of a constant (locked) tangent scalar
#define LVX_LM_BASE 0x40000000       // ord[] value of landmark l (inverse depth): LVX_LM_BASE + l — landmarks live in their own rows (DevCommon::lmH), not in the band
#define LVX_ERR_FALLBACK 16         // device error bit: fast assembly kernel met a corner it does not handle; re-run with the legacy kernels

namespace lvx {

// device view handed to every kernel
struct DevCommon {
  const double* state;
  int N, L;
  double t0, dt;
  uint32_t locks;
  uint32_t what;
  double imu_mto, sensor_mto;
  CamIntr cam;
  // layout
  const int* ord;
  int nb, bw, nbd;   // nbd: assembly border size (solve border + pseudo rows) = leading dimension of C
  int nbd_solve;
  int nrep;          // replicas of C / gc / cost in use: LVX_NREP, or one per workgroup in deterministic mode
  int hub_lo, hub_hi;   // band positions [hub_lo, hub_hi) of the hub rows of Bd are cleared per pass and accumulated; outside, the fold STORES (see k_clear's HubClear)
  const void* hubs;  // HubShared[2]: surfel (tau_L) and cam-surfel (tau_C) poses at t_map
  const So3Pre* pre; // [N]: u-independent SO3 quantities of the control-point pairs (k, k+1), rebuilt from the state at the start of every pass (k_state_prepass)
  double* Hb;    // [nb][bw+1] lower band, column-major by column
  double* gb;    // [nb]
  double* Bd;    // [nbd][nb]
  double* C;     // [LVX_NREP][nbd*nbd] (lower triangle used)
  double* gc;    // [LVX_NREP][nbd]
  // landmark rows: the inverse depth of landmark l couples to the band positions [lm_p0[l], lm_p0[l] + lm_wl) (the knots its views touch), to
  // border variables (camera extrinsics, hub knots) and to itself; no two landmarks share a residual, so the landmark block of J^T J is
  // diagonal and the solver eliminates it first (what Ceres' SPARSE_SCHUR does with its e-blocks).
  // lmH[l * lm_ls + ...] = [ band couplings (lm_wl) | border couplings (nbd) | H_ll | g_l ]
  double* lmH; const int* lm_p0; int lm_wl, lm_ls;
  double* cost;  // [LVX_NREP]
  int* err;      // bit0 range, bit1 non-unit quaternion, bit2 band overflow; err[4 + f]: rows of family f on the fallback list
  int* fb_list; int fb_cap;   // [LVX_NUM_FAM][fb_cap] rows a fused kernel handed to the exact per-segment kernel (null: a row it cannot take sets LVX_ERR_FALLBACK)
  // outputs (may be null)
  double* residuals;
  int32_t* jcols;
  double* jvals;
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

// Experiment / debug switches: read ONCE from the environment (LVX_<NAME>) when the context is created and changed afterwards only
// through lvx_set_switch — the evaluation path never calls getenv.
struct Switches {
  int force_legacy = 0, serial = 0, no_graph = 0, deterministic = 0, clear_all = 0, solver_seq = 0, solver_timing = 0, chunk_r = 0, chunk_r_imu = 0, chunk_r_rep = 0, chunk_rows = 0, rep_rows = 0, da_sync = 0, rep_fused = 0, solver_nd = 0;   // rep_fused: 1 = the single-launch reprojection kernel (k_reproj_fused; measured slower, opt-in), otherwise the five-launch chain
};
struct SwitchName { const char* name; int Switches::*field; bool relayout; };
const SwitchName* switch_table(int* count);

struct Family {
  int n = 0;
  // host copies (original order)
  std::vector<double> t, a3, b3;        // imu: t, gyro, acc | surfel: t, pt | reproj: t0_obs, uv_obs(2)
  std::vector<int32_t> id0, id1;        // surfel: plane | reproj: landmark | camsurf: landmark, plane
  double huber = 0, weight = 1;
  // device (sorted by key)
  DevBuf d_t, d_a3, d_b3, d_id0, d_id1, d_perm;
};

}  // namespace lvx

struct lvx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  hipStream_t fam_stream[4] = {nullptr, nullptr, nullptr, nullptr};   // concurrent family kernels
  hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr}, ev_jac = nullptr;   // ev_jac: reprojection Jacobians materialised
  hipEvent_t ev_fb[2] = {nullptr, nullptr};   // row-level fallback: a fused kernel's list is complete (the exact kernels over it run on fam_stream[1], beside the rest of the pass)
  std::string last_error;
  // problem
  bool have_spline = false;
  double t0 = 0, dt = 1;
  int N = 0, L = 0;
  uint32_t locks = LVX_LOCK_LIDAR_TAU | LVX_LOCK_CAM_TAU;
  double imu_mto = 0.01, sensor_mto = 0.001;
  lvx::CamIntr cam{};
  lvx::Family imu, surf, rep, cs;
  bool has_prior = false;
  double prior_t = 0, prior_q[4] = {1, 0, 0, 0}, prior_w = 1;
  double t_map = 0;
  std::vector<double> planes, lm_uv, lm_t0;
  lvx::DevBuf d_planes, d_lm_uv, d_lm_t0;
  // layout
  bool layout_dirty = true;
  std::vector<int> ord;
  int nb = 0, bw = 0, nbd = 0, nbd_ext = 0, n_hub = 0, hub0 = 0;   // nbd: solve border (hub knots + 22 calib); nbd_ext = nbd + 12 pseudo rows
  lvx::Switches sw;
  int rep_groups = 0;          // (reference window, observation window) groups of the reprojection cross-term kernel (d_repB[2])
  int rep_fused_wg = 0; lvx::DevBuf d_repF;   // fused reprojection kernel (k_reproj_fused): its groups (0: the five-launch chain runs) and their table [start | count], largest first
  bool fb_on = false; int fb_mask = 0, fallback_rows = 0; lvx::DevBuf d_fb;   // row-level exact fallback (run_evaluate): lists enabled after the first pass that needed them; rows on them in the last checked pass
  // Small device -> host reads of the evaluation / LM loop go through PINNED words: hipMemcpyAsync into pageable memory (a stack variable) is staged and returns only when
  // the copy is done — every such "asynchronous" read was a host stop of its own (three per LM iteration between the pass and the solve).  Slots (doubles): 0 cost,
  // 1 .. 8 the pass's error words, 16 .. 21 k_plus' sums, 24 gradient max norm, 25 .. 38 shared gradient, 40 .. 47 the step's sums, 48 .. 49 pivot codes,
  // 50 .. 63 shared diagonal, 70 .. 71 shared time offsets.
  double* pin = nullptr;
  std::function<void()> before_eval_sync;   // run_evaluate calls it right before it waits for the pass's cost: what the caller wants on the host after the SAME host stop (the LM loop: step norms, the candidate's diagonal and gradient norm)
  bool force_legacy = false;   // set when the fast assembly kernels hit a case only the per-segment kernels handle exactly
  lvx::DevBuf d_pre;   // So3Pre[N]
  lvx::DevBuf d_repT;   // [rep.n][56] landmark-row records of the reprojection blocks (k_reproj_cross -> k_reproj_lmrows)
  lvx::DevBuf d_lm_grp; int lm_ngrp = 0, lm_gspread = 0;   // landmark groups of the elimination kernel (k_lm_schur_grp): [ngrp + 1 offsets | landmark ids sorted by first band position]
  lvx::DevBuf d_lmH, d_lm_p0, d_Hr, d_Br, d_red; int lm_wl = 0, lm_ls = 0; const double* p_Hs = nullptr;   // landmark rows (DevCommon::lmH); solver: band / border rows / [g_b | C | g_c] after the landmark elimination
  int hub_near_lo = 0, hub_near_hi = 0;   // band positions a residual can couple to a hub knot DIRECTLY (not through the pseudo pose): IMU / LiDAR rows within 4 knots, reprojection blocks within their span
  std::vector<int> h_colhi; std::vector<uint8_t> h_colfull; int bw_near = 0, layout_epoch = 0;   // band column profile (ensure_layout) for the elimination plan of lvx_nd.h
  lvx::DevBuf d_colfull; int clear_npre = 0; std::vector<uint8_t> bd_row_live;   // structural clear of the band / border rows (k_clear)
  lvx::DevBuf d_hubs, d_chunk[LVX_NUM_FAM], d_repB[4];   // reprojection MFMA path: [0] materialised Jacobians + residuals, [1] knot intervals, [2] landmark and [3] observation-order index of the rows in (reference interval, landmark) order
  // deterministic mode (LVX_DETERMINISTIC=1): chunks of every family grouped into colours of pairwise disjoint knot ranges (det_list: chunk ids, colour by colour;
  // det_col: offsets), launched colour after colour on one stream with one wavefront per workgroup — no two additions to one address can race
  int nrep = LVX_NREP;
  std::vector<int> h_chunk_k0[LVX_NUM_FAM], h_chunk_rows[LVX_NUM_FAM], det_col[LVX_NUM_FAM], det_cross_col;
  lvx::DevBuf d_det_list[LVX_NUM_FAM], d_det_cross, d_chk, d_imu_rtab;
  // owner-computes IMU kernel (k_imu_own): batches [offsets | first interval], first batch of every workgroup, owner of every band column (-1: cleared + atomics)
  lvx::DevBuf d_imu_chunk, d_imu_wg, d_imu_own; int imu_wg = 0, imu_nch = 0, imu_span = 0, imu_owned_cols = 0; size_t imu_own_k_off = 0; std::vector<int> imu_h_k0, imu_h_wg_c0;
  int chunk_var[LVX_NUM_FAM] = {0};   // != 0: chunks of equal ROW count (first interval of chunk c at d_chunk[n_chunk + 1 + c]) instead of equal interval count
  int n_chunk[LVX_NUM_FAM] = {0}, chunk_r[LVX_NUM_FAM] = {0};   // workgroups and knot intervals per workgroup of the MFMA assembly kernels (pick_chunk)
  lvx::DevBuf d_ord, d_Hb, d_gb, d_Bd, d_C, d_gc, d_cost, d_err, d_state, d_res, d_jcols, d_jvals, d_pairs[LVX_NUM_FAM];
  // solver workspace (lvx_solver.hip)
  lvx::DevBuf d_L, d_Y, d_S, d_delta, d_diag, d_scal, d_state_try, d_zero;
  // block cyclic reduction (lvx_bcr.hip): diagonal blocks, per-level coupling blocks, pivot info; rocBLAS handle
  int solver_fallbacks = 0;   // lvx_layout::solver_fallbacks
  lvx::DevBuf d_bcrD, d_bcrG, d_bcrInfo, d_Y2, d_gram, d_bcrLinv; bool bcr_linv = false;   // d_bcrLinv: inverses of the factors' 16 x 16 diagonal triangles (k_potrf_batched -> k_trsm_reg)
  void* blas = nullptr;
  int bcr_b = 0, bcr_nblk = 0, bcr_nreal = 0;
  void* nd = nullptr;   // lvx::NdPlan (lvx_nd.h): leaves + separators elimination of a band that is narrow except for isolated wide runs
  int64_t n_blocks = 0, n_residuals = 0;
  int64_t fam_row0[LVX_NUM_FAM + 1] = {0};
  uint32_t last_what = 0;
  const double* last_state_d = nullptr; bool last_want_res = false, err_unchecked = false;   // see check_last_eval
  // upstream kernels (lvx_upstream.hip)
  lvx::DevBuf d_up[8];
  size_t assoc_rings = 0; int assoc_wpr = 0, assoc_list_total = 0;
  int coresident_compact = -1, coresident_emit = -1;   // workgroups of k_surfel_compact_mb / k_assoc_emit_fused the device holds at once (occupancy API x CUs, with a margin): both spin on words published by every other workgroup of their launch
  lvx::DevBuf d_pub; unsigned emit_epoch = 0, compact_epoch = 0;   // publication words (epoch | count) of the single-launch compactions: [0, 2048) k_assoc_emit_fused, [2048, 4096) k_surfel_compact_mb; zero once   // shape the association work buffer (d_assoc[3]) was cleared for
  const double* assoc_map_planes = nullptr; int assoc_map_P = 0; bool assoc_map_ready = false;   // lvx_surfel_map_prepare_d: the association grid of this plane table is built
  lvx::DevBuf d_assoc[4];   // surfel association: grid geometry + cell counts / offsets, cell lists, emission counters, hit bitmasks + counts (private: cleared once per shape)
  // device-resident DataAssociation (lvx_set_scans / lvx_data_association): [0] raw scans, [1] state, [2] map time, [3] map pose, [4] scans in the map frame = map cloud,
  // [5] plane table, [6] flags, [7] SurfelPoint arrays
  lvx::DevBuf d_da_key, d_da_aux;   // lvx_data_association_poses (first map): the key-scan map cloud; per-scan [time | q | p | valid | present | pose16 | key list]
  lvx::DevBuf d_da[8]; int da_S = 0, da_H = 0, da_W = 0, da_points = 0; std::vector<lvx_surfel_plane> da_planes;
  // lvx_data_association without host stops (round 5): capacities learned from the previous call (leaves, planes, association-grid list entries; 0 = none yet, the call
  // runs the synchronous path), the pinned mirror the device writes its counts into, the planes still to fetch from d_da[5] when somebody asks for them
  int da_cap_nl = 0, da_cap_P = 0, da_cap_list = 0, da_planes_pending = 0; int* da_pinned = nullptr; long long da_spec_runs = 0, da_spec_misses = 0;
  // the last lvx_scan_register / lvx_scan_register_batch (its results stay in d_up[0]): sweeps, rings, points, per-sweep input offsets and kept points, byte offsets of
  // {cloud, lflat_r, scan_start, cnt} inside the scratch buffer
  int sr_S = 0, sr_rings = 0; long long sr_N = 0; std::vector<int32_t> sr_off; std::vector<int> sr_m; std::array<size_t, 4> sr_batch_off{}; std::array<size_t, 8> sr_lay{}; std::vector<int32_t> sr_counts;   // sr_lay: {cloud, curv, label, sort, pick, lists, scan_start, scan_end}
  struct Voxels {
    float leaf = 0; int min_pts = 0, n_points = 0, n_leaves = 0;
    const void* d_pts = nullptr;   // the cloud of the last build (device; caller- or context-owned), read again by lvx_surfel_extract
    int grid[13] = {0};   // VxGrid: min_b, max_b, div_b, mul, inv(float bits)
    lvx::DevBuf misc, keys, vals, runs, cells, tmp, leaf_i, leaf_d, leaf_f;
    // sync-free build (lvx_upstream.hip: voxel_build_device / vox_info): capacities, the pinned host mirror of the device-computed VxInfo, the captured launch chain
    double eig_mult = 0; int cap = 0, sort_bits = 32; long long cells_cap = 0; size_t tmp_bytes[3] = {0, 0, 0}; void* h_info = nullptr; bool pending = false;
    std::array<uint64_t, 16> graph_key{};   // what the chain was captured with: cloud buffer, size, parameters, work buffers, stream
    void* graph = nullptr; bool graph_failed = false;
  } vox;
  // captured evaluation passes (lvx_eval.hip: run_evaluate), keyed by state buffer / request / flags / configuration version
  struct GraphEntry { const double* state; uint32_t what; int flags; uint64_t cfg; void* exec; };
  std::vector<GraphEntry> graphs;
  uint64_t cfg_version = 0;   // bumped by every lvx_set_* and every layout rebuild
  std::vector<double> lm_cost, lm_radius;
  std::vector<int> lm_accept;
  // sequence-per-GPU joint solve: host all-reduce hook, shared-extrinsics bookkeeping (lvx_solver.hip)
  lvx_allreduce_fn ar_fn = nullptr;
  void* ar_user = nullptr;
  void* rccl_comm = nullptr; int comm_rank = 0, comm_world = 1;   // lvx_rccl_init: the reductions run as ncclAllReduce on the context's stream
  lvx::DevBuf d_comm;                                            // device staging buffer of the reductions
  int64_t n_collectives = 0;                                     // reductions issued (either transport): lvx_collective_count
  int ns = 0;                 // free shared scalars = the last ns border variables
  int last_ns = 0;            // ns of the last solve (lvx_joint_shared_count)
  int sh_slot[LVX_N_SHARED] = {0};   // canonical slot (0..13: lidar theta p tau, cam theta p tau) of each of them
  double sh_lmd[LVX_N_SHARED] = {0}; // LM diagonal of the shared scalars (from the JOINT diagonal), added once after the reduction
  // profiling: (start, stop) event pairs per launch, read lazily by lvx_get_kernel_ms
  bool profiling = false;
  int profile_only = -1;   // >= 0: only this kernel's launches are timed
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  struct EvRec { int kernel; size_t e0, e1; };
  std::vector<EvRec> ev_recs;
};

namespace lvx {
int fail(lvx_ctx* ctx, int code, const std::string& msg);
int dev_alloc(lvx_ctx* ctx, DevBuf& b, size_t bytes);
int upload(lvx_ctx* ctx, DevBuf& b, const void* src, size_t bytes);
int upload_tmp(lvx_ctx* ctx, DevBuf& b, const void* src, size_t bytes);
int ensure_layout(lvx_ctx* ctx);
int check_last_eval(lvx_ctx* ctx);
int run_evaluate(lvx_ctx* ctx, const double* state_d, uint32_t what, double* cost, bool want_res_buffer);   // lvx_eval.hip: one evaluation pass
DevCommon make_common(lvx_ctx* ctx, const double* state_d, uint32_t what);
int bcr_plan(lvx_ctx* c);
int bcr_factor(lvx_ctx* c, const double* scale, const double* lmd, double inv_radius, int* info_out_d, double* Z = nullptr, int ldz = 0, int nrhs = 0);
int bcr_forward(lvx_ctx* c, double* Zin, double* Zy, int ldz, int nrhs);
int bcr_backward(lvx_ctx* c, double* Zy, double* Zx, int ldz, int nrhs);
int bcr_gram(lvx_ctx* c, const double* Z, int ldz, int n, double* M, int row_major_nz = 0);
void bcr_destroy(lvx_ctx* c);
// leaves + separators elimination (lvx_nd.h): nd_plan decides from the column profile whether it applies (nd_active afterwards) and sizes its buffers
int nd_plan(lvx_ctx* c, int nrhs);
bool nd_active(const lvx_ctx* c);
int nd_ldz(const lvx_ctx* c);
int nd_nz(const lvx_ctx* c);   // right-hand sides of the leaves + separators elimination are ROW-major [nd_ldz][nd_nz]
void nd_counts(const lvx_ctx* c, int* separators, int* leaves);
int nd_dense_start(lvx_ctx* c, const double* scale, const double* lmd, double inv_radius, const double* Bs, const double* gbs, double* Z, int ldz);   // the dense leaves, on their own stream
int nd_factor(lvx_ctx* c, const double* scale, const double* lmd, double inv_radius, int* info_out_d, const double* Bs, const double* gbs, double* Z, int ldz, int nrhs);   // Z is OUTPUT: L^-1 of the right-hand sides formed from the border rows Bs and g_b
int nd_backward(lvx_ctx* c, double* zb);
// profiling scope: records a (start, stop) HIP event pair on ctx->stream around a launch when profiling is on
struct ProfScope {
  lvx_ctx* c; int kernel; size_t e0 = 0; bool on; hipStream_t st;
  ProfScope(lvx_ctx* ctx, int k, hipStream_t stream = nullptr);
  ~ProfScope();
};
}  // namespace lvx

#define LVX_HIP(ctx, expr)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return lvx::fail(ctx, LVX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
