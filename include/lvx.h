/* lvx.h — C ABI of the MI355X-native evaluator for LVI-ExC's continuous-time calibration solve.
 *
 * Drop-in boundary.  The reference has no FFI layer; its seam is the Ceres cost-functor surface that every
 * Kontiki measurement builds in AddToEstimator and that ceres::Solve drives per residual block
 * (reference: src/lvi_exc/thirdparty/Kontiki/include/kontiki/trajectory_estimator.h:38-74).  A per-block FFI
 * would serialise the GPU, so this ABI is BATCHED: the host hands over whole measurement arrays once
 * (lvx_set_*: what TrajectoryManagerLVI::add*Measurement loops build, src/lvi_exc/src/core/trajectory_manager_lvi.cpp:464-606)
 * and then asks for one evaluation of all residual blocks + Jacobians + J^T J / J^T r per iterate
 * (lvx_evaluate: what ceres::Problem::Evaluate does through DynamicAutoDiffCostFunction::Evaluate).
 * INTEGRATION.md shows the reference-side binding.
 *
 * Conventions: every pointer is a HOST pointer unless the name ends in _d; the caller owns host buffers,
 * the context owns device buffers; a context is bound to one GPU and is not thread-safe; no exceptions
 * cross the ABI — the reference's std::range_error / std::runtime_error become LVX_E_RANGE / LVX_E_NONUNIT_QUAT.
 * All arithmetic is FP64.  Quaternions are stored (x, y, z, w) like Eigen::Map<Quaternion>
 * (kontiki/trajectories/spline_base.h:118-127, kontiki/sensors/sensors.h:36-43).
 *
 * STATE vector (flat doubles), n = n_knots, L = n_landmarks:
 *   r3_cp[n][3] | so3_cp[n][4] | imu{q4,p3,tau,roll,pitch,b_a3,b_g3} (16) | lidar{q4,p3,tau} (8) | cam{q4,p3,tau} (8) | rho[L]
 * TANGENT vector (what delta / gradient / normal equations are expressed in):
 *   knot k: 6k..6k+2 position, 6k+3..6k+5 rotation (ceres::EigenQuaternionParameterization delta)
 *   | 6n + {0 roll, 1 pitch, 2..4 b_a, 5..7 b_g, 8..10 lidar theta, 11..13 lidar p, 14 lidar tau,
 *           15..17 cam theta, 18..20 cam p, 21 cam tau} | 6n + 22 + l : rho_l
 */
#ifndef LVX_H
#define LVX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lvx_ctx lvx_ctx;

/* error codes (0 = success) */
#define LVX_OK 0
#define LVX_E_RANGE (-1)         /* std::range_error: time outside the spline / unordered spans (spline_base.h:207-221, trajectory_estimator.h:102-127) */
#define LVX_E_NONUNIT_QUAT (-2)  /* std::runtime_error in logq (kontiki/math/quaternion_math.h:19-23) */
#define LVX_E_ALLOC (-3)
#define LVX_E_HIP (-4)
#define LVX_E_COMM (-5)          /* the host-supplied all-reduce callback failed (lvx_lm_solve_shared) */
#define LVX_E_RCCL LVX_E_COMM
#define LVX_E_ARG (-6)
#define LVX_E_STATE (-7)         /* call order (e.g. evaluate before set_spline) */
#define LVX_E_NODEVICE (-8)      /* no HIP device: the product path never falls back to the CPU */
#define LVX_E_NOTPD (-9)         /* damped normal equations not positive definite */

/* lock mask: mirrors Lock / IsLocked flags (kontiki/sensors/sensors.h:113-135, kontiki/trajectories/trajectory.h:149-155,
 * kontiki/sensors/constant_bias_imu.h:68-81, kontiki/sfm/landmark.h Lock).  IMU q_rel/p_rel/tau are always constant, as in the reference. */
#define LVX_LOCK_TRAJ (1u << 0)
#define LVX_LOCK_R3 (1u << 1)        /* R3 spline absent: TrajectoryEstimator<UniformSO3SplineTrajectory> of Solve #0 */
#define LVX_LOCK_LIDAR_Q (1u << 2)
#define LVX_LOCK_LIDAR_P (1u << 3)
#define LVX_LOCK_LIDAR_TAU (1u << 4)
#define LVX_LOCK_CAM_Q (1u << 5)
#define LVX_LOCK_CAM_P (1u << 6)
#define LVX_LOCK_CAM_TAU (1u << 7)
#define LVX_LOCK_ACC_BIAS (1u << 8)
#define LVX_LOCK_GYRO_BIAS (1u << 9)
#define LVX_LOCK_LANDMARKS (1u << 10)

/* lvx_evaluate `what` bits */
#define LVX_EVAL_COST (1u << 0)
#define LVX_EVAL_RESIDUALS (1u << 1)
#define LVX_EVAL_NORMAL_EQ (1u << 2)
#define LVX_EVAL_JACOBIAN (1u << 3)   /* debug: per-row (cols, vals) in tangent coordinates, see lvx_get_jacobian */

/* residual families, in the order residual rows are laid out */
#define LVX_FAM_GYRO 0
#define LVX_FAM_ACCEL 1
#define LVX_FAM_PRIOR 2
#define LVX_FAM_SURFEL 3
#define LVX_FAM_REPROJ 4
#define LVX_FAM_CAMSURF 5
#define LVX_NUM_FAM 6

#define LVX_JAC_WIDTH 64   /* columns per row of the debug Jacobian */

/* PinholeMeta (kontiki/sensors/pinhole_camera.h:20-41) + CameraMeta (kontiki/sensors/camera.h:25-29) */
typedef struct lvx_pinhole {
  int32_t rows, cols;
  double readout;
  double fx, fy, cx, cy;
  double k1, k2, p1, p2, k3;
} lvx_pinhole;

/* layout of the structured normal equations kept on the device */
typedef struct lvx_layout {
  int32_t n_knots, n_landmarks, n_tangent;
  int32_t n_band;      /* variables in the banded part (non-hub knots interleaved with landmarks) */
  int32_t bandwidth;   /* scalar half-bandwidth of the banded part */
  int32_t n_border;    /* dense border: hub knots (6 each) then the 22 calibration scalars */
  int32_t border_ld;   /* leading dimension of the border block as exported by lvx_export_border_d (n_border + pseudo-pose rows) */
  int32_t n_hub_knots, hub_knot0;
  int64_t n_blocks;    /* residual blocks per evaluation */
  int64_t n_residuals; /* residual rows per evaluation */
  int32_t exact_fallback; /* 1 once an evaluation needed the per-segment kernels for EVERYTHING (more rows than the fallback lists hold, |tau_imu| >= dt): they are used from then on */
  int32_t solver_fallbacks; /* LM steps so far whose cyclic-reduction factorisation lost positive definiteness and were redone by the sequential band Cholesky */
  int32_t fallback_rows;    /* blocks of the last evaluation whose cost was read that the fused kernels handed, row by row, to the exact per-segment kernel (a control-point
                             * pair beyond 0.8 rad, the merged map-time segment corner): the pass stays on the fused kernels for everything else */
  int32_t solver_separators; /* the last solver plan: separators of the leaves + separators elimination of the band (a band that is narrow except for isolated wide runs:
                              * csrc/lvx_nd.h); 0: the uniform block chain (block cyclic reduction with b = bandwidth) runs */
  int32_t solver_leaves;     /* ... and its leaves (narrow stretches and wide runs) */
} lvx_layout;

/* lifetime ------------------------------------------------------------------------------------------------*/
int lvx_create(lvx_ctx** out, int device, uint32_t flags);
void lvx_destroy(lvx_ctx* ctx);
const char* lvx_version(void);
const char* lvx_last_error(const lvx_ctx* ctx);

/* problem description ---------------------------------------------------------------------------------------*/
/* SplineSegmentMeta of both splines (kontiki/trajectories/spline_base.h:31-39; split_trajectory.h:76-82: same dt, t0) */
int lvx_set_spline(lvx_ctx* ctx, double t0, double dt, int n_knots);
int lvx_set_camera(lvx_ctx* ctx, const lvx_pinhole* cam);
/* IO::IMUData stream (src/lvi_exc/include/utils/dataset_reader.h:45-50): one gyro block and one accel block per sample
 * (trajectory_manager_lvi.cpp:464-503); weights = CalibParamManager::global_opt_{gyro,acce}_weight */
int lvx_set_imu(lvx_ctx* ctx, int n, const double* t, const double* gyro3, const double* acc3, double w_gyro, double w_acc);
/* OrientationMeasurement of initialSO3TrajWithGyro (trajectory_manager_lvi.cpp:52-58); q as (w, x, y, z); enable = 0 removes it */
int lvx_set_orientation_prior(lvx_ctx* ctx, int enable, double t, const double* q_wxyz, double weight);
/* closest-point plane parameters Pi[3] of SurfelAssociation::get_surfel_planes() (trajectory_manager_lvi.cpp:567-569) */
int lvx_set_planes(lvx_ctx* ctx, int n_planes, const double* plane_pi3);
/* SurfelPoint list (src/lvi_exc/include/core/surfel_association.h:41-46) -> LiDARSurfelPoint blocks (trajectory_manager_lvi.cpp:571-581) */
int lvx_set_surfel(lvx_ctx* ctx, int n, const double* pt3, const double* t, const int32_t* plane_id, double t_map, double huber, double weight);
/* sfm::Landmark table: reference observation uv and its view's t0 (kontiki/sfm/landmark.h:50-53, observation.h:35-37, view.h:34-35);
 * inverse depths live in the state vector */
int lvx_set_landmarks(lvx_ctx* ctx, int n_landmarks, const double* uv_ref2, const double* t0_ref);
/* StaticRsCameraMeasurement blocks (trajectory_manager_lvi.cpp:505-531): huber = w_cam, weight = 1 reproduces the reference's argument swap */
int lvx_set_reproj(lvx_ctx* ctx, int n, const int32_t* landmark_id, const double* uv_obs2, const double* t0_obs, double huber, double weight);
/* CameraSurfelLandmark blocks (trajectory_manager_lvi.cpp:584-606) */
int lvx_set_camsurf(lvx_ctx* ctx, int n, const int32_t* landmark_id, const int32_t* plane_id, double t_map, double huber, double weight);
int lvx_set_locks(lvx_ctx* ctx, uint32_t lock_mask);
/* experiment / debug switches.  They are read once from the environment (LVX_<NAME>) by lvx_create; this call changes one on a
 * live context (name with or without the LVX_ prefix, e.g. "FORCE_LEGACY", 1).  Unknown name: LVX_E_ARG. */
int lvx_set_switch(lvx_ctx* ctx, const char* name, int value);
/* max |time offset| bounds used to widen spans when a time offset is free (sensors.h:161-162; trajectory_manager_lvi.h:118-119) */
int lvx_set_time_offset_bounds(lvx_ctx* ctx, double imu_max, double sensor_max);

int lvx_state_size(const lvx_ctx* ctx);
int lvx_tangent_size(const lvx_ctx* ctx);
int lvx_get_layout(lvx_ctx* ctx, lvx_layout* out);
/* first residual row of every family (LVX_FAM_* order) and, at [LVX_NUM_FAM], the total: block i of family f owns rows row0[f] + i * rows_per_block */
int lvx_get_family_rows(lvx_ctx* ctx, int64_t row0[LVX_NUM_FAM + 1]);

/* evaluation -------------------------------------------------------------------------------------------------*/
/* One pass over every residual block at `state`: cost = sum 1/2 rho(|r|^2) (HuberLoss where the reference attaches one),
 * raw weighted residuals (family-major, input order inside a family), and — for LVX_EVAL_NORMAL_EQ — the robustified
 * J^T J and J^T r in tangent coordinates, left on the device in the structured layout.  residuals may be NULL. */
int lvx_evaluate(lvx_ctx* ctx, const double* state, uint32_t what, double* cost, double* residuals);
/* same, state already resident on the device; nothing is copied back except *cost (if not NULL) */
int lvx_evaluate_d(lvx_ctx* ctx, const double* state_d, uint32_t what, double* cost);
/* debug / parity: expand the structured normal equations into dense host arrays (n_tangent^2 and n_tangent); small problems only */
int lvx_get_normal_eq_dense(lvx_ctx* ctx, double* H, double* g);
/* g = J^T r and diag(J^T J) (robustified, tangent layout, 0 for constant scalars) of the last LVX_EVAL_NORMAL_EQ evaluation — what
 * ceres::Problem::Evaluate returns as `gradient`; any problem size; either pointer may be NULL */
int lvx_get_gradient(lvx_ctx* ctx, double* g, double* diag);
/* debug / parity: rows of the last LVX_EVAL_JACOBIAN evaluation: cols[n_residuals][LVX_JAC_WIDTH] (-1 = unused), vals likewise (pre-loss) */
int lvx_get_jacobian(lvx_ctx* ctx, int32_t* cols, double* vals);
/* run on a caller-owned HIP stream (e.g. torch's current stream) instead of the context's own; NULL restores the own stream */
int lvx_set_stream(lvx_ctx* ctx, void* hip_stream);
/* multi-GPU (one calibration sequence per GPU): copy the dense border block of the last normal equations —
 * C[border_ld^2] (lower triangle), g_c[border_ld], cost, error word — into a caller-owned DEVICE buffer of border_ld^2 + border_ld + 2 doubles,
 * queued on the context's stream, ready for one RCCL all-reduce.  The last double is the pass's device error word (0 = every residual block
 * was evaluated): after a sum over the ranks a non-zero value tells EVERY rank that some rank's sums are incomplete, with no host
 * synchronisation in between (lvx_synchronize returns the matching error code on the rank that produced it). */
int lvx_export_border_d(lvx_ctx* ctx, double* out_d);
/* debug / repeatability: order-independent 64-bit checksums of the bit patterns of the structured normal equations of the last LVX_EVAL_NORMAL_EQ
 * evaluation — band, band gradient, border rows, dense border block, border gradient, landmark rows.  With LVX_DETERMINISTIC=1 (lvx_set_switch
 * "DETERMINISTIC") every addition of a pass happens in a fixed order — one stream, launches of workgroups with pairwise disjoint knot ranges, one
 * wavefront per workgroup, one replica of the dense accumulators per workgroup — and two evaluations of the same state give identical checksums
 * (fused locked-offset kernels; the per-segment kernels of a free time offset are not covered).  Several times slower than the default. */
int lvx_normal_eq_checksum(lvx_ctx* ctx, uint64_t out[6]);
/* keep `state` resident in the context's device buffer; lvx_evaluate_d(ctx, NULL, ...) then evaluates it without any host traffic */
int lvx_set_state(lvx_ctx* ctx, const double* state);
int lvx_get_state(lvx_ctx* ctx, double* state_out);
/* block until every launch queued on the context's stream has finished; returns the first device-side error of the last evaluation */
int lvx_synchronize(lvx_ctx* ctx);
/* per-kernel timing with HIP events recorded on the context's own stream around every launch (no host sync until read).
 * lvx_get_kernel_ms: sum of durations [ms] and launch counts since the last call, indexed by LVX_FAM_* (+ LVX_KERNEL_* below). */
#define LVX_KERNEL_FOLD 6
#define LVX_KERNEL_SOLVE 7
#define LVX_KERNEL_UPSTREAM 8
#define LVX_KERNEL_CLEAR 9          /* k_clear: structural clear of the accumulators + the state-only prepass */
#define LVX_KERNEL_REP_JAC 10       /* the five kernels of the fused reprojection path (LVX_FAM_REPROJ then times only the per-segment kernel) */
#define LVX_KERNEL_REP_OBS 11
#define LVX_KERNEL_REP_REF 12
#define LVX_KERNEL_REP_CROSS 13
#define LVX_KERNEL_REP_LMROWS 14
#define LVX_KERNEL_REP_FUSED 15     /* round 6: the single-launch reprojection kernel (k_reproj_fused); the five above then stay at zero */
#define LVX_KERNEL_FIXUP 16         /* the exact per-segment kernel over the fused kernels' fallback lists (lvx_layout::fallback_rows), all families */
#define LVX_NUM_KERNELS 17
/* enable: 0 off | 1 every launch | 2 + k: only the launches of kernel k (e.g. 2 + LVX_FAM_SURFEL: the dominant kernel — two event
 * records per pass instead of ~20, which cost ~5 % of a config-4 pass).  While profiling is on, passes are issued launch by launch (no graph replay). */
int lvx_set_profiling(lvx_ctx* ctx, int enable);
int lvx_get_kernel_ms(lvx_ctx* ctx, double* ms_sum, int64_t* launches);
/* Levenberg-Marquardt --------------------------------------------------------------------------------------------
 * Replaces ceres::Solve as configured by TrajectoryEstimator::Solve (kontiki/trajectory_estimator.h:38-68: TRUST_REGION,
 * LEVENBERG_MARQUARDT, SPARSE_SCHUR, max_num_iterations per stage; everything else Ceres defaults).  The trust-region loop,
 * LM damping and Jacobi scaling are restated from Ceres' public semantics (out-of-tree, parity unpinned). */
typedef struct lvx_lm_options {
  int32_t max_iterations;          /* options.max_num_iterations (30 / 50 / 80 / 200 per stage: trajectory_manager_lvi.cpp:60,86,125,182,244,340) */
  double initial_radius;           /* 1e4  */
  double max_radius;               /* 1e16 */
  double min_radius;               /* 1e-32 */
  double min_relative_decrease;    /* 1e-3 */
  double min_lm_diagonal;          /* 1e-6 */
  double max_lm_diagonal;          /* 1e32 */
  double function_tolerance;       /* 1e-6 */
  double gradient_tolerance;       /* 1e-10 */
  double parameter_tolerance;      /* 1e-8 */
  int32_t jacobi_scaling;          /* 1 */
  int32_t verbose;
} lvx_lm_options;
#define LVX_LM_NO_CONVERGENCE 0
#define LVX_LM_FUNCTION_TOLERANCE 1
#define LVX_LM_PARAMETER_TOLERANCE 2
#define LVX_LM_GRADIENT_TOLERANCE 3
#define LVX_LM_MAX_ITERATIONS 4
#define LVX_LM_FAILURE 5
#define LVX_LM_MIN_RADIUS 6       /* Ceres: CONVERGENCE, "minimum trust region radius reached" */
typedef struct lvx_lm_summary {
  int32_t iterations, successful_steps, termination;
  double initial_cost, final_cost, final_radius;
} lvx_lm_summary;
int lvx_lm_default_options(lvx_lm_options* opt);
/* minimise from `state` (in/out, host); summary may be NULL */
int lvx_lm_solve(lvx_ctx* ctx, double* state, const lvx_lm_options* opt, lvx_lm_summary* summary);
/* per-iteration trace of the last lvx_lm_solve: cost after the iteration, trust-region radius, accepted (1) / rejected (0) / invalid (-1); returns count */
int lvx_lm_get_history(lvx_ctx* ctx, int max_n, double* cost, double* radius, int32_t* accepted);
/* one damped solve on the normal equations of the last LVX_EVAL_NORMAL_EQ evaluation (if that evaluation was queued without a cost pointer its
 * device error word is read first: the exact kernels re-run a pass that needs them, a range / unit-quaternion error is returned):
 * (S H S + clamp(diag(S H S), 1e-6, 1e32) / radius) y = -S g, delta = S y (S = Jacobi scaling or identity); delta[n_tangent] on the host */
int lvx_solve_step(lvx_ctx* ctx, double radius, int jacobi_scaling, double* delta, double* model_cost_change);

/* x (+) delta with ceres::EigenQuaternionParameterization::Plus on quaternion blocks */
int lvx_plus(lvx_ctx* ctx, const double* state, const double* delta, double* state_out);

/* upstream point-cloud kernels ---------------------------------------------------------------------------------------*/
/* RsPointXYZIRT as the reference's scanRegistration receives it (src/aloam/src/scanRegistration.cpp:57-66), 32 bytes */
typedef struct lvx_rs_point {
  float x, y, z, pad;
  uint8_t intensity, pad2;
  uint16_t ring;
  uint32_t pad3;
  double timestamp;
} lvx_rs_point;
/* caller-owned host buffers with capacity n_in (scan_start/scan_end: n_rings); any pointer may be NULL.  Mirrors the globals
 * cloudCurvature / cloudSortInd / cloudNeighborPicked / cloudLabel (scanRegistration.cpp:82-85) and the four published clouds
 * as index lists into `cloud` (less_flat BEFORE the 0.2 m VoxelGrid of :440-444, which lvx_scan_less_flat_downsample applies) */
typedef struct lvx_scanreg_out {
  int32_t n;            /* points kept after the range / NaN filter = laserCloud size */
  float* cloud;         /* [n][4] x, y, z, intensity = ring + (t - t_first) */
  float* curvature;     /* [n] */
  int32_t* label;       /* [n] 2 sharp, 1 less sharp, 0 none, -1 flat */
  int32_t* sort_ind;    /* [n] */
  int32_t* picked;      /* [n] */
  int32_t* scan_start;  /* [n_rings] */
  int32_t* scan_end;    /* [n_rings] */
  int32_t* sharp; int32_t* less_sharp; int32_t* flat; int32_t* less_flat;
  int32_t counts[4];
} lvx_scanreg_out;
int lvx_scan_register(lvx_ctx* ctx, int n, const lvx_rs_point* pts, int n_rings, float min_range, lvx_scanreg_out* out);
/* The same for n_sweeps sweeps in ONE call (laserCloudHandler is invoked once per sweep, scanRegistration.cpp:134; sweeps are independent): pts holds the sweeps
 * back to back, sweep s = pts[sweep_offsets[s] .. sweep_offsets[s + 1]) (sweep_offsets[0] = 0), outs[s] its caller-owned output buffers (capacity = its input count).
 * One upload, one launch per stage over all sweeps' rings, one download — a single sweep keeps 16 of 256 CUs busy. */
int lvx_scan_register_batch(lvx_ctx* ctx, int n_sweeps, const int32_t* sweep_offsets, const lvx_rs_point* pts, int n_rings, float min_range, lvx_scanreg_out* outs);
/* device-resident variant: the points are already on the device (pts_d, sweeps back to back as above); the results STAY on the device inside the context — only the
 * kept-point count and the four list lengths per sweep come back (n_kept[n_sweeps], counts4[n_sweeps][4]; either may be NULL) — and lvx_scan_register_get downloads
 * one sweep's arrays on demand (the buffers of `out` that are not NULL). */
int lvx_scan_register_batch_d(lvx_ctx* ctx, int n_sweeps, const int32_t* sweep_offsets, const lvx_rs_point* pts_d, int n_rings, float min_range, int32_t* n_kept, int32_t* counts4);
int lvx_scan_register_get(lvx_ctx* ctx, int sweep, lvx_scanreg_out* out);
/* The published less-flat cloud: pcl::VoxelGrid (leaf_size 0.2 m) over every ring's less-flat points, rings concatenated (scanRegistration.cpp:425-447),
 * for the sweep of the last lvx_scan_register of this context.  out_xyzi4 [max_out][4], ring_counts [n_rings] (may be NULL), *n_out = total
 * (also when larger than max_out).  Points of a voxel are averaged in their input order (pcl's std::sort leaves the order of equal keys open). */
int lvx_scan_less_flat_downsample(lvx_ctx* ctx, float leaf_size, int max_out, float* out_xyzi4, int32_t* ring_counts, int32_t* n_out);
/* the same for sweep `sweep` of the context's last lvx_scan_register_batch (sweep 0 = the call above) */
int lvx_scan_less_flat_downsample_sweep(lvx_ctx* ctx, int sweep, float leaf_size, int max_out, float* out_xyzi4, int32_t* ring_counts, int32_t* n_out);

/* pclomp::VoxelGridCovariance::applyFilter (src/ndt_omp/include/pclomp/voxel_grid_covariance_omp_impl.hpp:49-374): the grid stays on the
 * device inside the context; leaves are numbered in ascending voxel-key order (std::map iteration order of the reference) */
typedef struct lvx_voxel_info { int32_t n_leaves, n_points; int32_t min_b[3], max_b[3], div_b[3], divb_mul[3]; } lvx_voxel_info;
int lvx_voxel_build(lvx_ctx* ctx, int n, const float* xyzi4, float leaf_size, int min_points_per_voxel, double min_covar_eigvalue_mult, lvx_voxel_info* info);
int lvx_voxel_build_d(lvx_ctx* ctx, int n, const float* xyzi4_d, float leaf_size, int min_points_per_voxel, double min_covar_eigvalue_mult);
/* lvx_voxel_build_d is ASYNCHRONOUS: the whole build is one launch chain on the context's stream without a host hop (extents, cell table, leaf count stay on the
 * device; the cloud must stay alive and unchanged until a consumer has run).  This call waits for it and returns what VoxelGridCovariance's getters report after
 * applyFilter (getMinBoxCoordinates / getMaxBoxCoordinates / getNrDivisions / getDivisionMultiplier, leaves_.size(): pcl/filters/voxel_grid.h, voxel_grid_covariance_omp.h:298-305). */
int lvx_voxel_get_info(lvx_ctx* ctx, lvx_voxel_info* info);
/* Leaf fields (voxel_grid_covariance_omp.h:92-190): nr_points (-1 = rejected), mean_, cov_, icov_, evecs_ (columns), evals_, centroid, pointList_
 * as offsets[n_leaves + 1] into point_ids (input indices, input order inside a leaf); any pointer may be NULL */
int lvx_voxel_get(lvx_ctx* ctx, int32_t* leaf_key, int32_t* leaf_n, double* mean3, double* cov9, double* icov9, double* evecs9, double* evals3, float* centroid3,
                  int32_t* offsets, int32_t* point_ids);
/* getNeighborhoodAtPoint7 (:423-438): leaf index per displacement {0,+x,-x,+y,-y,+z,-z} or -1 */
int lvx_voxel_lookup7(lvx_ctx* ctx, int nq, const float* xyzi4, int32_t* leaf_ids7);
int lvx_voxel_lookup7_d(lvx_ctx* ctx, int nq, const float* xyzi4_d, int32_t* leaf_ids7_d);
/* getNeighborhoodAtPoint1 (:440-446): the leaf of the query's own cell or -1 */
int lvx_voxel_lookup1(lvx_ctx* ctx, int nq, const float* xyzi4, int32_t* leaf_ids1);
int lvx_voxel_lookup1_d(lvx_ctx* ctx, int nq, const float* xyzi4_d, int32_t* leaf_ids1_d);
/* getNeighborhoodAtPoint(relative_coordinates, reference_point, neighbors) (:378-408), the general form behind the 1- / 7- / 26-cell variants (:411-446):
 * leaf_ids[q][r] = leaf at displacement rel3[r] = (di, dj, dk) from the query's cell, or -1 (outside the grid, empty, fewer than min_points_per_voxel points).
 * The reference returns the hits in this order without the misses.  DIRECT26 = pcl::getAllNeighborCellIndices() as rel3. */
int lvx_voxel_lookup_rel(lvx_ctx* ctx, int nq, const float* xyzi4, int n_rel, const int32_t* rel3, int32_t* leaf_ids);
/* SurfelAssociation::getAssociation flag pass (src/lvi_exc/src/core/surfel_association.cpp:111-138): plane_of_point[H*W] = surfel id or -1.
 * Conflicts resolve as the reference's SERIAL plane loop (highest plane id wins); W <= 4096 */
int lvx_surfel_assoc(lvx_ctx* ctx, int H, int W, const float* scan_map_xyzi4, int n_planes, const double* plane_p4, const double* box_min3, const double* box_max3,
                     double radius, int sel_per_ring, int32_t* plane_of_point);
/* device-resident variant: planes10_d = p4[P][4] | box_min[P][3] | box_max[P][3] */
int lvx_surfel_assoc_d(lvx_ctx* ctx, int H, int W, const float* scan_d, int n_planes, const double* planes10_d, double radius, int sel_per_ring, int32_t* plane_of_point_d);

/* n_scans organised scans [n_scans][H][W] against one surfel table in one call (scans are independent: this is also the unit that shards over GPUs) */
int lvx_surfel_assoc_batch_d(lvx_ctx* ctx, int n_scans, int H, int W, const float* scans_d, int n_planes, const double* planes10_d, double radius, int sel_per_ring,
                             int32_t* plane_of_point_d);

/* SurfelAssociation::setSurfelMap happens once per DataAssociation, getAssociation once per scan (lvi_initialize_surfel_orb.cpp:1184-1199): build the association
 * grid of a surfel table once — later lvx_surfel_assoc_batch_d / lvx_surfel_assoc_d calls with THE SAME planes10_d pointer and n_planes reuse it (the caller
 * promises the table's contents are unchanged) until the next prepare or a call with another table. */
int lvx_surfel_map_prepare_d(lvx_ctx* ctx, int n_planes, const double* planes10_d);
/* End of the prepared table's lifetime: call it BEFORE the table's memory is freed, reused or rewritten.  The grid is keyed by (address, n_planes) only — a
 * caching allocator hands the same address to the next table of the same size, and a stale grid would silently miss hits — so a caller that prepares owns
 * the table until it releases (or prepares another one).  Calls without a preceding prepare build the grid per call and are always safe. */
int lvx_surfel_map_release(lvx_ctx* ctx);

/* sequence-per-GPU joint solve (BASELINE config 5) ---------------------------------------------------------*/
/* Every rank owns one calibration sequence (trajectory, gravity, biases, landmarks are private); the rig extrinsics — lidar theta(3) p(3)
 * tau, camera theta(3) p(3) tau = 14 tangent scalars — are shared.  One LM iteration of the JOINT problem: every rank eliminates its private
 * variables on its GPU, ONE all-reduce of the 14 x 14 reduced system (+ rhs + flags, 211 doubles), every rank solves it and back-substitutes;
 * plus tiny reductions of the cost, the model cost change and the step / gradient norms so that all ranks take the same accept / reject and
 * termination decisions.  The library does not link a communication library: the host supplies the reduction over HOST buffers (RCCL via
 * torch.distributed, MPI, ...); messages are <= 211 doubles and latency-bound.  All ranks must use the same lock mask and options. */
#define LVX_N_SHARED 14
#define LVX_REDUCE_SUM 0
#define LVX_REDUCE_MAX 1
typedef int (*lvx_allreduce_fn)(void* user, double* buf, int n, int op);   /* in-place over all ranks; returns 0 on success */
/* lvx_solve_step / lvx_lm_solve of the joint problem; fn == NULL degenerates to the single-sequence calls */
int lvx_solve_step_shared(lvx_ctx* ctx, double radius, int jacobi_scaling, lvx_allreduce_fn fn, void* user, double* delta, double* model_cost_change);
int lvx_lm_solve_shared(lvx_ctx* ctx, double* state, const lvx_lm_options* opt, lvx_allreduce_fn fn, void* user, lvx_lm_summary* summary);

/* RCCL transport: with a communicator installed the reductions of lvx_solve_step_shared / lvx_lm_solve_shared (fn = NULL) run as ncclAllReduce on the
 * context's stream — the per-step [S | rhs | votes] block is packed, reduced and solved on the device with no host round trip.  librccl is loaded with
 * dlopen (no link-time dependency; a process that already loaded RCCL, e.g. through torch, shares it).  Rank 0 creates the 128-byte id, the host
 * distributes it (any side channel), every rank calls lvx_rccl_init; lvx_rccl_finalize (or lvx_destroy) frees the communicator.
 * Collectives: one after the first evaluation, then exactly two per LM iteration — the reduced 14 x 14 system, and the decision block (candidate cost, model terms,
 * norms, the candidate's joint diagonal / gradient); the two cannot merge, the candidate depends on the reduced system's solution.  One more (a vote) when the
 * iteration cap ends the loop right behind a rejected step: the re-evaluation of x that follows a rejection has met no collective yet.
 * A rank that fails locally votes in the next collective and EVERY rank returns there (the failing one with its code, the others LVX_E_COMM). */
int lvx_rccl_unique_id(lvx_ctx* ctx, void* id128);
int lvx_rccl_init(lvx_ctx* ctx, const void* id128, int rank, int world);
int lvx_rccl_finalize(lvx_ctx* ctx);
/* in-place all-reduce (LVX_REDUCE_SUM / LVX_REDUCE_MAX) of a caller's device buffer of n doubles over the installed communicator, queued on the context's stream:
 * the transport of the per-step border-block reduction of a sequence-per-GPU evaluation (lvx_export_border_d -> this) */
int lvx_rccl_allreduce_d(lvx_ctx* ctx, double* buf_d, int n, int op);
/* number of shared scalars the last joint solve of this context kept out of its local elimination (LVX_N_SHARED when a transport was active, 0 for a
 * single-sequence solve): lets a caller / test assert that the joint path was taken */
int lvx_joint_shared_count(lvx_ctx* ctx);
/* reductions issued by this context since the last reset (either transport) */
int64_t lvx_collective_count(lvx_ctx* ctx, int reset);

/* NDT registration derivatives ------------------------------------------------------------------------*/
/* pclomp::NormalDistributionsTransform::computeDerivatives with DIRECT7 search (src/ndt_omp/include/pclomp/ndt_omp_impl.hpp:180-285, 289-430, 484-536)
 * against the voxel grid of the last lvx_voxel_build of this context (resolution = its leaf size): score, 6-gradient and 6 x 6 Hessian (row-major)
 * of the NDT objective at the transform vector p6 = (tx, ty, tz, rx, ry, rz); input = source cloud, trans = the same cloud already transformed by p6
 * (pcl::transformPointCloud, as the caller does before every call).  Per-point arithmetic in float like the reference, accumulation in double. */
int lvx_ndt_derivatives(lvx_ctx* ctx, int n, const float* input_xyzi4, const float* trans_xyzi4, const double* p6, double outlier_ratio, int compute_hessian,
                        double* score, double* gradient6, double* hessian36);

/* NDT registration (the loop around the derivatives) ---------------------------------------------------------------------------*/
/* pcl::Registration::align -> pclomp::NormalDistributionsTransform::computeTransformation (src/ndt_omp/include/pclomp/ndt_omp_impl.hpp:81-171): Newton steps
 * (JacobiSVD solve of the 6 x 6 system :127-129) with the More-Thuente step length (computeStepLengthMT :772-931, updateIntervalMT :648-685, trialValueSelectionMT
 * :689-768), derivatives by computeDerivatives (:180-285) and, after a line search that moved, computeHessian (:540-645), all against the voxel grid of the last
 * lvx_voxel_build of this context (= setInputTarget at that resolution, ndt_omp.h:117-122,275-282).  The Newton / line-search bookkeeping (a handful of scalars per
 * iteration) runs on the host as in the reference; every evaluation over the cloud is one kernel.  Defaults = the reference's constructor (:46-76).
 * search: 1 = DIRECT1, 7 = DIRECT7, 26 = DIRECT26 (ndt_omp.h:59-64; KDTREE is not offered: the calibration uses DIRECT7, lidar_odometry.cpp:32-43). */
typedef struct lvx_ndt_options { double step_size, outlier_ratio, transformation_epsilon; int32_t max_iterations, search; } lvx_ndt_options;
typedef struct lvx_ndt_result {
  float final_transformation[16];   /* row-major 4 x 4: getFinalTransformation() */
  double p6[6];                     /* (tx, ty, tz, rx, ry, rz) the loop ended at */
  double score, trans_probability;  /* the last evaluated score and score / n (getTransformationProbability, :170) */
  int32_t iterations, converged, n_evaluations, reserved;   /* nr_iterations_, converged_, derivative evaluations over the cloud */
} lvx_ndt_result;
int lvx_ndt_default_options(lvx_ndt_options* opt);
/* src = the input cloud (setInputSource); guess16 = row-major initial guess or NULL (identity, as align(output) passes); aligned_xyzi4 (may be NULL) receives the
 * source under the final transformation = the `output` cloud of align(). */
int lvx_ndt_align(lvx_ctx* ctx, int n, const float* src_xyzi4, const float* guess16, const lvx_ndt_options* opt, lvx_ndt_result* result, float* aligned_xyzi4);
/* pcl::Registration::getFitnessScore(max_range) (what src/ndt_omp/apps/align.cpp:30 prints): mean over the source points of the squared distance from the point
 * under `transform16` to its nearest target point (float L2, exact search), counting distances <= max_range; DBL_MAX when nothing is counted. */
int lvx_ndt_fitness(lvx_ctx* ctx, int n_src, const float* src_xyzi4, const float* transform16, int n_tgt, const float* tgt_xyzi4, double max_range, double* fitness);

/* surfel map extraction ------------------------------------------------------------------------------*/
/* SurfelAssociation::setSurfelMap + checkPlaneType (src/lvi_exc/src/core/surfel_association.cpp:50-86, 246-266) over the leaves of the last
 * lvx_voxel_build of this context (the cloud passed to it must still be alive): leaves with >= min_leaf_points points and planarity >= p_lambda,
 * plane fit, Pi = -d n, AABB of the leaf's points; output in voxel-key (std::map) order.  The reference fits with pcl RANSAC (random); this is
 * the deterministic variant documented in DESIGN.md: leaf PCA plane -> inliers within dist_threshold -> PCA refit -> reselect. */
typedef struct lvx_surfel_plane { double p4[4]; double Pi[3]; double box_min[3]; double box_max[3]; int32_t leaf, n_points, n_inliers, plane_type; } lvx_surfel_plane;
int lvx_surfel_extract(lvx_ctx* ctx, double p_lambda, double dist_threshold, int min_leaf_points, int min_inliers, int max_planes, lvx_surfel_plane* planes, int32_t* n_planes);

/* scan de-skew ------------------------------------------------------------------------------------*/
/* licalib PointXYZIT (src/lvi_exc/include/utils/pcl_utils.h:39-44), 32 bytes */
typedef struct lvx_point_xyzit { float x, y, z, pad; float intensity; float pad2; double timestamp; } lvx_point_xyzit;
/* The chronological SurfelPoint emission of getAssociation (src/lvi_exc/src/core/surfel_association.cpp:141-158) for n_scans associated scans (flags from
 * lvx_surfel_assoc_batch_d), all device-resident: per scan column-major (w outer, h inner), points with a flag and a non-zero raw timestamp; scans concatenated.
 * SurfelPoint fields as arrays: raw point (LiDAR frame), point in the map frame, timestamp, plane id.  *n_out = total (also when larger than max_out: nothing is
 * written then; size the outputs and call again); per_scan_counts[n_scans] may be NULL. */
int lvx_surfel_emit_d(lvx_ctx* ctx, int n_scans, int H, int W, const int32_t* flags_d, const float* scans_map_d, const lvx_point_xyzit* scans_raw_d, int max_out,
                      double* pt3_d, double* pt_map3_d, double* t_d, int32_t* plane_d, int32_t* n_out, int32_t* per_scan_counts);
/* getAssociation for n_scans scans with host buffers in and out (what DataAssociation's loop over the scans does, lvi_initialize_surfel_orb.cpp:1192-1199): flags
 * plane_of_point[n_scans * H * W] (may be NULL) and the concatenated SurfelPoint list; *n_out = its length (nothing is written when it exceeds max_out) */
int lvx_surfel_assoc_emit(lvx_ctx* ctx, int n_scans, int H, int W, const float* scans_map_xyzi4, const lvx_point_xyzit* scans_raw, int n_planes, const double* plane_p4,
                          const double* box_min3, const double* box_max3, double radius, int sel_per_ring, int32_t* plane_of_point, int max_out, double* pt3, double* pt_map3, double* t,
                          int32_t* plane, int32_t* n_out);
/* LIinitializer::DataAssociation, refinement branch (src/lvi_exc/test/lvi_initialize_surfel_orb.cpp:1180-1201), device-resident end to end:
 *   ScanUndistortion::undistortScanInMap (every point of every scan de-skewed into the LiDAR frame at the map time; map cloud = the scans concatenated),
 *   LiDAROdometry::ndtInit(resolution) + setInputTarget(map cloud) (voxel covariance grid), SurfelAssociation::setSurfelMap, getAssociation per scan.
 * lvx_set_scans hands over the dataset's organised raw scans once (LioDataset::get_scan_data: [n_scans][H][W], per-point timestamps, NaN x = no return);
 * lvx_data_association runs one association round at `state` and leaves the surfel map and the chronological SurfelPoint list in the context.
 * The FIRST round on a set of scans stops at the host four times (leaf count, plane count, list length of the association grid, SurfelPoint counts); every later round
 * launches the whole chain over capacities learned from the round before — the kernels read the true counts from device memory — and stops ONCE, at the end; a count that
 * outgrew its capacity (or a cell table that was too small) discards that attempt and the round runs the four-stop chain (lvx_data_association_stats counts both).
 * The plane records stay on the device until lvx_get_surfel_map asks for them.
 * averageTimeDownSmaple (every step-th point, surfel_association.cpp:240-244) is the caller's: it picks from the arrays of lvx_get_surfel_points. */
typedef struct lvx_assoc_options {
  float ndt_resolution;              /* lvi.yaml:26, 0.5 */
  int32_t min_points_per_voxel;      /* voxel_grid_covariance_omp.h:207, 6 */
  double min_covar_eigvalue_mult;    /* :208, 0.01 */
  double plane_lambda;               /* 0.7 (lvi_initialize_surfel_orb.cpp:1183) */
  double fit_threshold;              /* RANSAC distance threshold 0.05 (surfel_association.cpp:279) */
  int32_t min_leaf_points, min_inliers;   /* 10 (:61), 20 (:283) */
  double radius;                     /* associated_radius_ 0.05 */
  int32_t selected_per_ring, reserved;    /* getAssociation(..., 2) */
} lvx_assoc_options;
int lvx_assoc_default_options(lvx_assoc_options* opt);
int lvx_set_scans(lvx_ctx* ctx, int n_scans, int H, int W, const lvx_point_xyzit* raw);
int lvx_data_association(lvx_ctx* ctx, const double* state, double map_time, const lvx_assoc_options* opt, int32_t* n_planes, int32_t* n_points);
/* The FIRST DataAssociation of a calibration — the InitializationDone branch (lvi_initialize_surfel_orb.cpp:1175-1178), where the map comes from per-scan odometry poses
 * (LOAM's, ReadPoseGT :458-516) and only the SO3 spline of Solve #0 exists:
 *   Mapping() (:1262-1300) = ScanUndistortion::undistortScan() (scan_undistortion.h:40-57: rotation-only de-skew of every scan into its own start frame, scans whose stamp
 *     lies outside the spline are dropped) + LiDAROdometry::feedScan(scan_t, scan, pose, update_map = true, using_loam = true) per scan (lidar_odometry.cpp:45-74);
 *     updateKeyScan / checkKeyScan (:89-128): the first scan, or one > key_dist [m] from the last key scan or turned by > key_angle_deg in yaw / pitch / roll, is moved with
 *     its pose and appended to the key-scan map = the NDT target (reference: 0.2 m, 5 deg);
 *   undistortScanInMap(odom_data_map) (scan_undistortion.h:95-116): every de-skewed scan moved with its pose = the scans in the map frame;
 *   setSurfelMap(lidar_odom->getNDTPtr(), map_time) on the voxel grid of the KEY-SCAN map with opt->plane_lambda (the constructor's 0.6, :127,240); getAssociation per scan.
 * scan_t[n_scans]: the scans' header stamps; pose16[n_scans][16]: row-major 4 x 4 scan -> map (LiDAROdometry::OdomData::pose); has_pose (may be NULL = all): 0 = no
 * odometry pose for that stamp (loam_poses_map_.find fails: the scan is skipped).  key_scan (may be NULL): out, 1 where the scan joined the key-scan map.
 * Results as lvx_data_association: lvx_get_surfel_map / lvx_get_surfel_points / lvx_get_scans_in_map (absent scans: NaN). */
int lvx_data_association_poses(lvx_ctx* ctx, const double* state, const double* scan_t, const double* pose16, const int32_t* has_pose, double key_dist, double key_angle_deg,
                               const lvx_assoc_options* opt, int32_t* n_planes, int32_t* n_points, int32_t* key_scan);
int lvx_get_surfel_map(lvx_ctx* ctx, int max_planes, lvx_surfel_plane* planes);
/* rounds that took the one-stop chain since the context was created, and how many of those had to be repeated on the four-stop chain */
int lvx_data_association_stats(lvx_ctx* ctx, int64_t* one_stop_rounds, int64_t* repeated_rounds);
int lvx_get_surfel_points(lvx_ctx* ctx, int max_points, double* pt3, double* pt_map3, double* t, int32_t* plane);
/* parity / debug: the de-skewed scans of the last lvx_data_association, [n_scans][H][W][4] float */
int lvx_get_scans_in_map(lvx_ctx* ctx, float* xyzi4);

/* SurfelAssociation::associateVisualPointsWithPlanes (surfel_association.cpp:161-214) over the landmark table of lvx_set_landmarks and the inverse depths in `state`:
 * plane_of_landmark[l] = index of the surfel whose AABB strictly contains the landmark (map frame) within 2 * radius of its plane — the highest such index, as the
 * reference's loop leaves it — or -1 (also for rho < 0.05 and for reference views outside the spline).  q_LtoC (x, y, z, w), t_LinC: LiDAR pose in the camera frame. */
int lvx_landmark_assoc(lvx_ctx* ctx, const double* state, const double* q_LtoC_xyzw, const double* t_LinC3, double map_time, int n_planes, const double* plane_p4,
                       const double* box_min3, const double* box_max3, double radius, int32_t* plane_of_landmark);
/* TrajectoryManagerLVI::evaluateLidarPose for a batch of times (trajectory_manager_lvi.cpp:398-408): q_LtoG (x,y,z,w), p_LinG, valid = 0 outside the spline */
int lvx_evaluate_lidar_pose(lvx_ctx* ctx, const double* state, int n, const double* t, double* q_xyzw4, double* p3, int32_t* valid);
/* ScanUndistortion::undistort (src/lvi_exc/include/core/scan_undistortion.h:132-180): every point is moved with the spline pose at ITS timestamp into the
 * target frame (q_G_to_target, p_target_in_G); out = float xyzi per point */
int lvx_undistort_scan(lvx_ctx* ctx, const double* state, int n, const lvx_point_xyzit* raw, const double* q_G_to_target_xyzw, const double* p_target_in_G3,
                       int correct_position, float* out_xyzi4);

#ifdef __cplusplus
}
#endif
#endif /* LVX_H */
