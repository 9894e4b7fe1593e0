/*
 * svsdf_c.h -- C ABI of the MI355X-native SVSDF cost/gradient evaluator.
 *
 * Drop-in boundary for ONE hot path of ZJU-FAST-Lab/Implicit-SVSDF-Planner: the per-query-point
 * swept-volume-SDF safety penalty and its gradient evaluated inside the back-end optimizer's
 * cost callback.  Every entry point cites the reference interface it replaces; paths are
 * relative to the reference root, with
 *   BEO = src/planner_algorithm/include/planner_algorithm/back_end_optimizer.hpp
 *   SWM = src/swept_volume/include/swept_volume/sw_manager.hpp
 *   SHP = src/utils/include/utils/Shape.hpp
 *   MNC = src/utils/include/utils/minco.hpp
 *
 * Plain pointers and sizes only; no Eigen / torch / HIP types.  All matrices are COLUMN-major
 * exactly as the reference's Eigen objects store them.  One thread at a time per context (the
 * reference's LMBM driver is single-threaded and non-reentrant: src/utils/src/lmbm.cpp:4-6).
 *
 * The library needs a gfx950 device: every compute entry point returns a non-zero error code
 * (and svsdf_lmbm_evaluate returns +inf with g zeroed) when no HIP device is usable -- there is
 * no CPU fallback.
 */
#ifndef SVSDF_C_H
#define SVSDF_C_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Shape ids: registry order of SWM:187-235 (the reference keys the registry by the stem of
 * conf.inputdata, SWM:352-355), then the fallback Polygon (SWM:363-372). */
enum svsdf_shape_id {
  SVSDF_SHAPE_sdUnevenCapsule = 0,
  SVSDF_SHAPE_sdCutDisk = 1,
  SVSDF_SHAPE_sdTrapezoid = 2,
  SVSDF_SHAPE_sdRhombus = 3,
  SVSDF_SHAPE_star = 4,
  SVSDF_SHAPE_sdTunnel = 5,
  SVSDF_SHAPE_sdHorseshoe = 6,
  SVSDF_SHAPE_sdHeart = 7,
  SVSDF_SHAPE_sdOrientedVesica = 8,
  SVSDF_SHAPE_sdRoundedCross = 9,
  SVSDF_SHAPE_sdRoundedX = 10,
  SVSDF_SHAPE_bigX = 11,
  SVSDF_SHAPE_sdMoon = 12,
  SVSDF_SHAPE_sdPie = 13,
  SVSDF_SHAPE_sdPie2 = 14,
  SVSDF_SHAPE_sdArc = 15,
  SVSDF_SHAPE_Polygon = 16,
  SVSDF_SHAPE_COUNT = 17
};

#define SVSDF_MAX_PIECES 128       /* MINCO pieces per trajectory handled on the device (64 until round 5).  The reference has no
                                      cap (minco.hpp:433-513); what bounds a trajectory here beyond this is the LDS pose table: one pose
                                      per 0.15 s of duration (SWM:567), ~ 36 B each, in one block's share of the CU's 160 KB --
                                      SVSDF_ERR_INVALID "trajectory too long for the LDS pose table" otherwise */
#define SVSDF_MAX_POLY_VERTS 8190  /* Polygon outline vertices + one per loop of a multi-loop outline (the z = 0 outlines of
                                      the reference meshes have 77 ... 754) */
#define SVSDF_MAX_DEVICES 8        /* GPUs one context can drive (one xGMI node) */

/* Error codes (0 = ok).  HIP runtime errors are returned as SVSDF_ERR_HIP_BASE + hipError_t. */
enum svsdf_status {
  SVSDF_OK = 0,
  SVSDF_ERR_INVALID = 1,       /* bad argument (null ctx, N out of range, n != 4N-3, ...) */
  SVSDF_ERR_NO_DEVICE = 2,     /* no usable gfx950 device / HIP runtime unavailable */
  SVSDF_ERR_NO_POINTS = 3,     /* evaluate called before svsdf_set_points */
  SVSDF_ERR_NONFINITE = 4,     /* a non-finite value was produced on the device */
  SVSDF_ERR_RCCL = 5,          /* RCCL unavailable / failed (SVSDF_COMBINE_RCCL) */
  SVSDF_ERR_HIP_BASE = 1000
};

/* What the callee reads from TrajOptimizer / Config / SweptVolumeManager (BEO:50-95,
 * src/utils/include/utils/config.hpp:13-224). */
typedef struct svsdf_config {
  int shape_id;              /* enum svsdf_shape_id; see svsdf_shape_id_from_inputdata() */
  double poly_params[3];     /* Config::poly_params: shape offset x, y [m], yaw [deg]  SHP:281-294 */
  double safety_hor;         /* TrajOptimizer::safety_hor                              BEO:83 */
  double weight_p;           /* TrajOptimizer::weight_p                                BEO:77 */
  double rho;                /* TrajOptimizer::rho (time weight)                       BEO:48 */
  double head_state[9];      /* initState  3x3 col-major: col0 pos, col1 vel, col2 acc BEO:49 */
  double tail_state[9];      /* finalState 3x3 col-major                               BEO:50 */
  int device;                /* HIP device ordinal; -1 = current device */
  int polygon_nverts;        /* Polygon only: outline vertices (0 -> the 12 x 0.2 fallback  */
  const double *polygon_xy;  /*   rectangle of SWM:363-369); xy interleaved, copied.  For a mesh
                                (conf.inputdata = an .obj the shape registry does not know) pass
                                the outline svsdf_mesh_outline_obj() returns */
  int rank, world_size;      /* point sharding: this context keeps points k with
                                (sorted index k) % world_size == rank.  1 process per GPU. */
  int flags;                 /* SVSDF_FLAG_* */
  /* In-process multi-GPU (the reference is ONE process: the LMBM shim keeps its callback and instance in
   * file-static globals, src/utils/src/lmbm.cpp:4-6, and plan_manager.cpp:199 calls one optimizer): with
   * n_devices >= 2 the context drives devices[0..n_devices) from one host thread per device; device k keeps
   * stripe (rank * n_devices + k) of (world_size * n_devices) of the Morton order, every entry point fans the
   * evaluation out, sums the n_devices (19N+1)-double partials and synchronises before returning.
   * n_devices <= 1: the single device `device` (above).  A device may be listed more than once (several
   * stripes on one GPU; host combine only). */
  int n_devices;
  int devices[SVSDF_MAX_DEVICES];
  int combine;               /* SVSDF_COMBINE_*: how the per-device partials are summed */
  /* Polygon of several closed loops (a mesh section with a hole, two solids; svsdf_mesh_section): polygon_xy holds the
   * loops one after the other, polygon_loop_sizes[k] (>= 3) vertices each, summing to polygon_nverts.  The reference's
   * Polygon::getonlySDF (Shape.hpp:1448-1476) is the minimum of dis2Seg over all edges, negated on an odd count of
   * isCrossRayOnXDir over all edges: it does not care how the edges are chained, so the union of the loops' edges is
   * evaluated exactly like its single chain.  polygon_nloops <= 1 (or a NULL pointer): one loop, as the reference. */
  int polygon_nloops;
  const int *polygon_loop_sizes;
} svsdf_config;

#define SVSDF_COMBINE_AUTO 0 /* = HOST (measured: DESIGN.md "Multi-GPU") */
#define SVSDF_COMBINE_HOST 1 /* every device writes its partial to pinned host memory, fixed-order host sum */
#define SVSDF_COMBINE_RCCL 2 /* ncclAllReduce(ncclDouble, ncclSum) over an in-process communicator
                                (ncclCommInitAll), then one read-back; needs distinct devices */

#define SVSDF_FLAG_DEFAULT 0
#define SVSDF_FLAG_KEEP_INPUT_ORDER 1 /* do not Morton-sort the cloud at upload (debug) */
#define SVSDF_FLAG_HOST_ONLY 2        /* no device: only the host-side entry points work
                                         (svsdf_lmbm_prepare / svsdf_lmbm_finish, MINCO helpers);
                                         every device entry point fails with SVSDF_ERR_NO_DEVICE */

/* Piece-local time.  The reference subtracts the piece durations one after the other from t
 * (Trajectory::locatePieceIdx, trajectory.hpp:498-516).  Default (neither flag): the library does the same whenever
 * that can differ from t - (T_0+...+T_{i-1}) in a single subtraction, i.e. unless every duration is a coarse dyadic
 * number (multiple of 2^-20, like the 2.5 s of the BASELINE configs), in which case both forms are exact and equal and
 * the cheaper one runs.  svsdf_stats.piece_time_exact tells which form the last evaluation used. */
#define SVSDF_FLAG_EXACT_PIECE_TIME 4  /* always the reference's chain (O(piece index) per SDF evaluation) */
#define SVSDF_FLAG_FAST_PIECE_TIME 8   /* always the single subtraction: <= i ulp(t) away from the reference for generic
                                         durations, which flat stretches of SDF(t) amplify (differential fuzzing: gradient
                                         up to 5e-5 relative off, vs 1e-5 with the chain); round 1's behaviour */

typedef struct svsdf_ctx svsdf_ctx;

/* ---- lifetime ---------------------------------------------------------------------------- */
/* Replaces TrajOptimizer::setParam + SweptVolumeManager::init/initShape (SWM:339-374) for the
 * state this path reads.  Returns NULL on failure (see svsdf_last_error_string). */
svsdf_ctx *svsdf_create(const svsdf_config *cfg);
void svsdf_destroy(svsdf_ctx *ctx);
void svsdf_config_default(svsdf_config *cfg);
/* shape registry lookup as SWM:350-356 does it: stem of "shapes/star.obj" -> "star" -> id;
 * unknown stems give SVSDF_SHAPE_Polygon (the reference's fallback). */
/* Boundary states only (first lines of TrajOptimizer::optimize_traj_lmbm, back_end_optimizer.cpp:13-19):
 * keeps the context, the resident cloud, the launch plan and SweptVolumeManager's persistent
 * traj_duration (SWM:376-385) across optimisations. */
int svsdf_set_conditions(svsdf_ctx *ctx, const double head_state[9], const double tail_state[9]);
int svsdf_shape_id_from_inputdata(const char *inputdata);
const char *svsdf_shape_name(int shape_id);
const char *svsdf_last_error_string(const svsdf_ctx *ctx); /* ctx may be NULL */

/* ---- query points -------------------------------------------------------------------------- */
/* Replaces the fill of TrajOptimizer::parallel_points (src/plan_manager/src/plan_manager.cpp:
 * 168-175): P points, AoS xyz doubles (Eigen::Vector3d layout); z is ignored (BEO:790-791).
 * Copies; the cloud stays resident in HBM for all following evaluations. */
int svsdf_set_points(svsdf_ctx *ctx, const double *xyz_aos, size_t P);
/* Same, but xyz_aos is a DEVICE pointer on ctx's device (devices[0] of a multi-device context): inputs already
 * resident in HBM.  Both entry points plan the cloud on the device (bounding box -> Morton keys -> radix sort ->
 * gather of this context's stripe); svsdf_shard_plan below is the same plan computed on the host. */
int svsdf_set_points_device(svsdf_ctx *ctx, const double *d_xyz_aos, size_t P);
size_t svsdf_num_points(const svsdf_ctx *ctx);        /* points owned by this rank's shard */
/* Pure host: the original indices rank `rank` of `world_size` owns (Morton order, striped), in
 * the order the device stores them.  idx_out may be NULL to query the count. */
int svsdf_shard_plan(const double *xyz_aos, size_t P, int rank, int world_size, int flags,
                     long long *idx_out, size_t *count_out);

/* ---- the inner operator ---------------------------------------------------------------------- */
/* Replaces TrajOptimizer::addSaftyPenaOnSweptVolumeParallelTrueSDF (BEO:774-869):
 *   coeffs  : (6N) x 3 col-major, row 6i+k = coefficient of s^k of piece i (minco.getCoeffs())
 *   T       : N piece durations
 *   cost, gradT[N], gradC[(6N) x 3 col-major] are ACCUMULATED INTO (+=) like the reference.
 * Also performs SweptVolumeManager::updateTraj (SWM:376-385). Synchronises before returning. */
int svsdf_eval_penalty(svsdf_ctx *ctx, int N, const double *coeffs, const double *T,
                       double *cost, double *gradT, double *gradC);

/* Multi-process form (one process per GPU): leaves this rank's partial
 *   [ cost, gradC (18N, col-major), gradT (N) ]  = 19N + 1 doubles
 * in a device buffer owned by the context so the caller can all-reduce it in place (RCCL via
 * torch.distributed / ncclAllReduce), then hand it back to svsdf_accumulate_partial. */
int svsdf_eval_penalty_partial(svsdf_ctx *ctx, int N, const double *coeffs, const double *T,
                               double **d_partial, size_t *partial_len);
int svsdf_accumulate_partial(svsdf_ctx *ctx, int N, const double *partial_host,
                             double *cost, double *gradT, double *gradC);
/* Pure host: out[e] = partials[0][e] + partials[1][e] + ... in index order (the fixed-order sum the
 * multi-device context applies to its per-device partials; `partials` = G rows of `len` doubles). */
int svsdf_sum_partials(const double *partials, int G, size_t len, double *out);

/* ---- the full optimizer callback --------------------------------------------------------------- */
/* Same signature as lmbm_evaluate_t (src/utils/include/utils/lmbm.h:206-209); replaces
 * TrajOptimizer::costFunctionLmbmParallel (BEO:344-408) including MINCO forward/adjoint
 * (MNC:433-654) and the tau/xi maps (BEO:174-314).  x = [tau_0..tau_{N-1}, q_0(x,y,yaw), ...,
 * q_{N-2}], n = 4N - 3.  Returns the total cost and OVERWRITES g[0..n).  On a device error
 * returns +infinity with g zeroed (the reference has no error channel here). */
double svsdf_lmbm_evaluate(void *ctx, const double *x, double *g, const int n);
/* cost_pos, cost_other, cost_total of the last svsdf_lmbm_evaluate (BEO:396-398). */
int svsdf_last_costs(const svsdf_ctx *ctx, double costs3[3]);
/* Full callback split around the collective for the one-process-per-GPU form:
 * begin -> all-reduce the device partial -> finish. */
int svsdf_lmbm_begin(svsdf_ctx *ctx, const double *x, int n, double **d_partial, size_t *partial_len);
/* Host half of svsdf_lmbm_begin only: tau -> T, MINCO forward, energy partials (BEO:351-366);
 * writes the coefficients ((6N) x 3 col-major) and durations the device stage would receive. */
int svsdf_lmbm_prepare(svsdf_ctx *ctx, const double *x, int n, double *coeffs_out, double *T_out);
double svsdf_lmbm_finish(svsdf_ctx *ctx, const double *partial_host, double *g, int n);

/* ---- optimizer driver (host) ------------------------------------------------------- */
/* A limited-memory BFGS driver with the Lewis-Overton weak-Wolfe line search so that the callback can be
 * run end to end without the reference's Fortran LMBM (TrajOptimizer::optimize_traj_lmbm,
 * src/planner_algorithm/src/back_end_optimizer.cpp:3-95 -> lmbm::lmbm_optimize, src/utils/include/utils/lmbm.h:
 * 214-220).  Field names, defaults and status codes follow lbfgs_parameter_t / the LBFGS* enum of
 * src/utils/include/utils/lbfgs.hpp:33-160 (the solver the reference's mid end uses).  It is NOT the bundle
 * method: iterates differ from LMBM's, the objective and gradient it is fed are the same. */
typedef struct svsdf_lbfgs_params {
  int mem_size;           /* 8 */
  double g_epsilon;       /* 1e-5: stop when |g|_inf / max(1, |x|_inf) <= g_epsilon */
  int past;               /* 3 */
  double delta;           /* 1e-6: stop when the relative decrease over `past` iterations < delta */
  int max_iterations;     /* 0 = unlimited */
  int max_linesearch;     /* 64 */
  double min_step;        /* 1e-20 */
  double max_step;        /* 1e+20 */
  double f_dec_coeff;     /* 1e-4 */
  double s_curv_coeff;    /* 0.9 */
  double cautious_factor; /* 1e-6 */
  double machine_prec;    /* 1e-16 */
} svsdf_lbfgs_params;
enum svsdf_lbfgs_status {
  SVSDF_LBFGS_CONVERGENCE = 0, SVSDF_LBFGS_STOP = 1, SVSDF_LBFGS_CANCELED = 2,
  SVSDF_LBFGSERR_UNKNOWNERROR = -1024, SVSDF_LBFGSERR_INVALID_N, SVSDF_LBFGSERR_INVALID_MEMSIZE,
  SVSDF_LBFGSERR_INVALID_GEPSILON, SVSDF_LBFGSERR_INVALID_TESTPERIOD, SVSDF_LBFGSERR_INVALID_DELTA,
  SVSDF_LBFGSERR_INVALID_MINSTEP, SVSDF_LBFGSERR_INVALID_MAXSTEP, SVSDF_LBFGSERR_INVALID_FDECCOEFF,
  SVSDF_LBFGSERR_INVALID_SCURVCOEFF, SVSDF_LBFGSERR_INVALID_MACHINEPREC, SVSDF_LBFGSERR_INVALID_MAXLINESEARCH,
  SVSDF_LBFGSERR_INVALID_FUNCVAL, SVSDF_LBFGSERR_MINIMUMSTEP, SVSDF_LBFGSERR_MAXIMUMSTEP,
  SVSDF_LBFGSERR_MAXIMUMLINESEARCH, SVSDF_LBFGSERR_MAXIMUMITERATION, SVSDF_LBFGSERR_WIDTHTOOSMALL,
  SVSDF_LBFGSERR_INVALIDPARAMETERS, SVSDF_LBFGSERR_INCREASEGRADIENT
};
/* Same shape as lmbm_evaluate_t (lmbm.h:206-209): returns f(x), overwrites g[0..n). */
typedef double (*svsdf_evaluate_t)(void *instance, const double *x, double *g, const int n);
/* Called after every accepted iteration k (ls = evaluations of its line search); non-zero return cancels
 * (cf. lmbm_progress_t lmbm.h:211-213 and earlyExitLMBM, back_end_optimizer.hpp:1068-1085). */
typedef int (*svsdf_progress_t)(void *user, const double *x, const double *g, double fx, double step, int n,
                                int k, int ls);
void svsdf_lbfgs_params_default(svsdf_lbfgs_params *params);
/* Generic driver: minimise eval over R^n starting from x (in/out).  Returns a status of enum svsdf_lbfgs_status. */
int svsdf_lbfgs_minimize(int n, double *x, svsdf_evaluate_t eval, void *instance, svsdf_progress_t progress,
                         void *progress_user, const svsdf_lbfgs_params *params, double *final_cost,
                         int *iterations, int *evaluations);
/* optimize_traj_lmbm analogue: drives svsdf_lmbm_evaluate (points must be set) from x = [tau, q] in/out;
 * afterwards svsdf_lmbm_prepare(ctx, x, ...) gives the MINCO coefficients of the result (svsdf_last_costs
 * reflects the last callback evaluation, which is the returned point unless the final line search failed).  params may be NULL (defaults).  Returns a status of enum svsdf_lbfgs_status; values >= 0 mean a usable result. */
int svsdf_optimize_traj(svsdf_ctx *ctx, double *x, int n, const svsdf_lbfgs_params *params,
                        svsdf_progress_t progress, void *progress_user, double *final_cost, int *iterations,
                        int *evaluations);

/* ---- front-end consumers of the same shape SDFs (device) ------------------------- */
/* Replaces SweptVolumeManager::checkSubSWCollision (src/swept_volume/include/swept_volume/sw_manager.hpp:
 * 1171-1211), batched: the reference calls it once per A* edge from AstarPathSearcher::AstarGetSucc
 * (src/planner_algorithm/include/planner_algorithm/front_end_Astar.hpp:192-241) with the obstacle points of
 * getPointsInAABB2D around the child cell.  Edge e: father_states[3e..] / child_states[3e..] = (x, y, yaw),
 * obstacle points pts_xy[2*pts_offset[e] .. 2*pts_offset[e+1]) (pts_offset has n_edges+1 entries,
 * pts_offset[0] == 0).  free_out[e] = 1 where the reference returns true (no interpolated pose kt = 0,
 * 0.02, ..., <= 1 has sdf < 0 at any point), else 0.  Needs no trajectory and no svsdf_set_points. */
int svsdf_check_sub_sw_collision(svsdf_ctx *ctx, size_t n_edges, const double *father_states,
                                 const double *child_states, const size_t *pts_offset, const double *pts_xy,
                                 unsigned char *free_out);
/* Replaces BasicShape::initShape (src/utils/include/utils/Shape.hpp:386-430) + byteShapeKernel::
 * generateByteKernel (:194-216): per yaw index k (yaw_k from the reference's accumulated loop
 * `for (yaw = -PI; yaw < PI; yaw += 2*PI/kernel_count)`), cell (a, b) of the kernel_size^2 grid is occupied iff
 * getonlySDF((resu*a - side*resu, resu*b - side*resu), R(yaw_k)) <= safemargin
 * (safemargin = max(front_end_safeh, occupancy_resolution/2) at Shape.hpp:399).
 * map_out: kernel_count * kernel_size^2 bools [k][a][b]; bytes_out (may be NULL): kernel_count * kernel_size *
 * ((kernel_size+7)/8) bytes, bit (0x80 >> b%8) of byte [k][a][b/8]; yaw_out (may be NULL): kernel_count yaws;
 * loop_count (may be NULL): iterations the reference loop makes (kernel_count, or kernel_count+1 when
 * rounding lets the last yaw stay below PI -- the reference then writes past its arrays; only kernel_count
 * kernels are produced here).  Polygon -> SVSDF_ERR_INVALID (no such overload in the reference). */
int svsdf_shape_kernels(svsdf_ctx *ctx, int kernel_size, int kernel_count, double kernel_resolution,
                        double safemargin, unsigned char *map_out, unsigned char *bytes_out, double *yaw_out,
                        int *loop_count);

/* ---- query-point producer (host) ------------------------------------------------ */
/* Replaces, for the data this path consumes, PCSmapManager::rcvGlobalMapHandler
 * (src/map_manager/src/PCSmap_manager.cpp:88-210: cloud -> bounds -> occupancy grid with
 * `sta_threshold`) and the waypoint loop around getPointsInAABBOutOfLastOne
 * (src/plan_manager/src/plan_manager.cpp:156-175, PCSmap_manager.h:184-219).  xyz are float32
 * like pcl::PointXYZ.  svsdf_map_gather returns the occupied-voxel centres (deduplicated, ordered
 * by the reference's unified voxel id) ready for svsdf_set_points; out_xyz may be NULL to query
 * the count. */
typedef struct svsdf_map svsdf_map;
svsdf_map *svsdf_map_create(const float *xyz, size_t n, double resolution, int sta_threshold);
void svsdf_map_destroy(svsdf_map *map);
int svsdf_map_info(const svsdf_map *map, int dims[3], double bmin[3], double bmax[3], size_t *occupied);
int svsdf_map_gather(const svsdf_map *map, const double *centres_xyz, size_t ncentres, const double halfbd[3],
                     double *out_xyz, size_t capacity, size_t *count);
/* ASCII PCD v0.7, FIELDS x y z (src/plan_manager/pcds/map_*.pcd); xyz may be NULL to query n. */
int svsdf_pcd_read_ascii(const char *path, float *xyz, size_t capacity, size_t *n);

/* ---- mesh shapes (host; BASELINE config 5: "arbitrary .obj mesh, no analytic shape SDF") ------------------------ */
/* The reference loads conf.inputdata with igl::read_triangle_mesh (src/utils/include/utils/Shape.hpp:281-313) and,
 * when the file's stem is not in its shape registry, plans with the generic Polygon shape over an outline
 * (sw_manager.hpp:350-372; the outline is hard-coded there).  Every query of the planar planner has z = 0
 * (BEO:790-791), so the part of a mesh it can see is the cross-section z = z0 = 0: these two entry points return that
 * outline (the longest closed loop; crossing points of the straddling triangles, chained through shared mesh edges) as
 * the xy-interleaved vertex list svsdf_config::polygon_xy takes.  xy_out may be NULL to query *count; loops (may be
 * NULL) receives the number of closed loops the section has (1 for one solid).
 * V: nv x 3 doubles, F: nf x 3 zero-based vertex indices.  The .obj reader takes `v` / `f` records (1-based or negative
 * indices, a/b/c forms, polygons fanned into triangles). */
int svsdf_mesh_outline(const double *V, size_t nv, const int *F, size_t nf, double z0, double *xy_out,
                       size_t capacity_verts, size_t *count, int *loops);
int svsdf_mesh_outline_obj(const char *obj_path, double z0, double *xy_out, size_t capacity_verts, size_t *count,
                           int *loops);
/* The whole section: every closed loop of z = z0, largest enclosed area first, as svsdf_config::polygon_xy /
 * polygon_loop_sizes / polygon_nloops take them (the planner then sees holes and separate solids, not just the largest
 * loop).  xy_out and loop_sizes may both be NULL to query the counts. */
int svsdf_mesh_section(const double *V, size_t nv, const int *F, size_t nf, double z0, double *xy_out, size_t capacity_verts,
                       size_t *n_verts, int *loop_sizes, size_t capacity_loops, size_t *n_loops);
int svsdf_mesh_section_obj(const char *obj_path, double z0, double *xy_out, size_t capacity_verts, size_t *n_verts,
                           int *loop_sizes, size_t capacity_loops, size_t *n_loops);

/* ---- swept-volume outline -------------------------------------------------------------------------------
 * What the reference's (unused) swept-volume surface extraction is for -- SweptVolumeManager::calculateSwept
 * (sw_manager.hpp:321-336) -> sw_calculate::calculation / getmesh (sw_calculate.cpp:4-305): a sparse-voxel continuation
 * with one serial gradient descent per voxel corner, then igl::marching_cubes; shown by vis->visMesh -- for this
 * planar planner: the boundary of the swept volume's z = 0 section, i.e. the zero set of the swept-volume implicit
 * function min over t of the shape SDF (getSDFofSweptVolume<false,true>, SWM:844-866: the hot path's argmin solve; the
 * reference's calculateSwept marches the same function, its scalarFunc SWM:1426-1446; the sign is the sign of the value
 * the optimizer is penalised with, SWM:921), as closed polylines with the inside on their left (outer boundaries
 * counter-clockwise, holes clockwise).  Batched on the GPU: a hierarchical narrow band
 * (coarse nodes first, then only cells whose four corners all lie within 1.5 cell diagonals of zero or change sign; cells across a found crossing are added until every chain closes) and marching
 * squares over the finest band cells of size `cell`; runs in a private context, the caller's resident cloud and launch
 * plan are untouched.  margin >= 0 widens the searched box (path bounding box + 2 shape bound radii + margin).
 * xy_out (interleaved, all loops one after the other) / loop_sizes (vertices per loop) may both be NULL to query the
 * counts.  stats (may be NULL): nodes evaluated vs what a dense grid of that cell size holds; open_chains > 0 means the
 * zero set left the searched box. */
typedef struct svsdf_outline_stats {
  unsigned long long nodes_evaluated, dense_nodes, cells_marched, batches;
  int open_chains;
} svsdf_outline_stats;
int svsdf_swept_outline(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, double cell, double margin,
                        double *xy_out, size_t capacity_verts, size_t *n_verts, int *loop_sizes, size_t capacity_loops,
                        size_t *n_loops, svsdf_outline_stats *stats);

/* The mesh vis->visMesh("sweptmesh2D", ...) is given (SWM:331): the surface of the outline's extrusion over z in
 * [z0, z1] -- per loop of n vertices 2n mesh vertices (bottom ring, top ring) and 2n wall triangles with their normals
 * away from the inside; with caps != 0 also the bottom and top faces (every counter-clockwise loop triangulated with
 * the holes it contains, on the rings' own vertices): a closed, consistently oriented triangle surface like the
 * reference's marching-cubes output.  V_out: 3 doubles per vertex, F_out: 3 zero-based indices per triangle; both may
 * be NULL to query the counts.  Host only (no GPU). */
int svsdf_outline_extrude(const double *xy, const int *loop_sizes, size_t n_loops, double z0, double z1, int caps,
                          double *V_out, size_t capacity_verts, size_t *n_verts, int *F_out, size_t capacity_tris,
                          size_t *n_tris);

/* ---- host-side MINCO helpers (MNC:397-655) ------------------------------------------------------- */
/* waypoints inPs: 3 x (N-1) col-major; out coeffs (6N) x 3 col-major. */
int svsdf_minco_coeffs(const double head_state[9], const double tail_state[9], int N,
                       const double *inPs, const double *T, double *coeffs);
void svsdf_forward_T(const double *tau, double *T, int N);   /* BEO:213-226 */
void svsdf_backward_T(const double *T, double *tau, int N);  /* BEO:228-241 */

/* ---- diagnostics ------------------------------------------------------------------------------------ */
/* Per-point results of getTrueSDFofSweptVolume<true> (SWM:916-1018) for the last trajectory
 * given to this context, in the ORIGINAL input order of this rank's shard: sdf[P], tstar[P],
 * grad_xy[2P] (any may be NULL).  Runs the device pipeline for (N, coeffs, T) first. */
int svsdf_query_points(svsdf_ctx *ctx, int N, const double *coeffs, const double *T,
                       double *sdf, double *tstar, double *grad_xy);
/* Work counters of the last evaluation. */
typedef struct svsdf_stats {
  unsigned long long points;          /* main queries in this shard */
  unsigned long long interior_points; /* points that entered the GSIP loop */
  unsigned long long solves;          /* argmin solves executed (main + selected GSIP samples) */
  unsigned long long gsip_samples;    /* GSIP circle samples emitted (the reference solves all of them) */
  unsigned long long sdf_evals;       /* SDF-at-time evaluations executed by the argmin kernel k_solve (layer-1 table
                                         evaluations + full evaluations: polynomial, sincos, transform, shape) */
  unsigned long long scan_evals;      /* of which layer-1 table evaluations (transform + shape only) */
  double device_ms;                   /* HIP-event time of the whole device pipeline (profiling on) */
  double solve_ms;                    /* HIP-event time during which >= 1 k_solve launch was executing (profiling on) */
  unsigned int solve_launches;        /* k_solve launches of the last evaluation */
  unsigned int gsip_iterations;       /* GSIP iterations that had work (rounds + supplementary) */
  unsigned long long culled_points;   /* main queries proven inactive (sdf > safety_hor) without a solve */
  int gsip_bound_mode;                /* 0 = cheap chunk bound, 1 = table scan of every GSIP sample, 2 = lazy: table scan of
                                         the samples within the selection band of the cheap bound only, 3 = anchor scans */
  int bound_mode_decided;             /* 1 once the mode is fixed for this point set (after <= 1 evaluation) */
  double bound_ratio;                 /* GSIP solves / samples of the deciding evaluation (rule: > 0.5 -> full) */
  int n_devices;                      /* devices that took part (1 unless svsdf_config::n_devices > 1) */
  int combine;                        /* SVSDF_COMBINE_* used by the last evaluation */
  double combine_ms;                  /* host wall time from "all devices done" to "summed partial on the host" */
  double setup_ms;                    /* host wall time of the last svsdf_set_points (sort + upload) */
  int piece_time_exact;               /* 0: single-subtraction piece-local time (exactly equivalent for this trajectory's
                                         durations, or forced by SVSDF_FLAG_FAST_PIECE_TIME); 1, 2: the reference's chain */
  double solve_ms_sum;                /* plain sum of the k_solve launch durations (solve_ms merges the intervals of
                                         launches that ran concurrently on different streams) (profiling on) */
  unsigned long long round_scan_evals; /* layer-1 table evaluations executed by k_round (seed scans of the GSIP samples
                                          in the scanning bound modes; transform + shape only); NOT part of sdf_evals */
  double round_ms;                    /* HIP-event time during which >= 1 k_round launch was executing (profiling on) */
  double round_ms_sum;                /* plain sum of the k_round launch durations (profiling on) */
  int batches;                        /* point batches (concurrent streams) the last evaluation ran as */
  unsigned long long speculative_evals; /* of sdf_evals: halving-ladder candidates evaluated BEHIND the accepted one (the
                                          lane group evaluates G candidates per step; the reference's sequential loop
                                          stops at the accepted one) */
  int plan_settled;                   /* 1 once bound mode, batch count and launch widths are fixed for this point set:
                                         the first evaluations after svsdf_set_points decide them (<= 6 evaluations, all
                                         with identical results); steady-state timing starts here */
  int tail_iter;                      /* GSIP iteration from which the fused tail kernel (k_tail: every remaining iteration of a
                                         batch in one launch, a half-wave owns a point until it is finished) took over; -1: none */
  unsigned int tail_launches;         /* k_tail launches of the last evaluation (one per batch) */
  unsigned long long tail_points;     /* GSIP points still active when the tail took over */
  double tail_ms;                     /* HIP-event time during which >= 1 k_tail launch was executing (profiling on) */
  double tail_ms_sum;                 /* plain sum of the k_tail launch durations (profiling on) */
  double shader_clock_mhz;            /* shader clock the evaluation ran at, measured by the kernel itself: cycles of the
                                         shader-clock counter (s_memtime) per cycle of the constant-rate counter (s_memrealtime,
                                         hipDeviceAttributeWallClockRate) over the life of the first wave of the main solve; a
                                         multi-device context reports its slowest device; 0 when nothing ran */
  double fanout_ms;                   /* multi-device contexts: host wall time of the evaluation that is neither a device's
                                         pipeline nor the combine -- waking the per-device threads (post -> the last thread
                                         starts) + joining them (the last thread done -> the caller runs again) */
} svsdf_stats;
int svsdf_last_stats(const svsdf_ctx *ctx, svsdf_stats *out);

/* Launch plan of the resident point set.  Every field only moves TIME: each setting returns the same bits (cost, gradient
 * and per-point results).  By default everything follows deterministic rules: the first evaluation
 * after svsdf_set_points runs with the cheap GSIP bound and decides the bound mode from its counters, the batch count
 * follows from mode and shard size, the second evaluation records the launch widths; `settled` is 1 from then on.
 * svsdf_set_plan pins fields (the AUTO value leaves a field to its rule) for this and every later point set of the
 * context; svsdf_get_plan reports what is in force.  Multi-device contexts apply / report per device (get: device 0). */
#define SVSDF_PLAN_AUTO (-1)
typedef struct svsdf_plan {
  int bound_mode;       /* GSIP upper-bound mode: 0 cheap chunk bound, 1 table scan of every sample, 2 lazy, 3 anchor scans (every
                           third sample, the others only if their Lipschitz bound reaches the selection band); SVSDF_PLAN_AUTO */
  int batches;          /* concurrent point batches 1..8; SVSDF_PLAN_AUTO: by rule; -2: measured (three HIP-event timings
                           per candidate count, best median) */
  int lanes_per_query;  /* lanes of the main solve's lane groups: 1, 2, 4, 8, 16, 32; SVSDF_PLAN_AUTO: by shard size */
  int tail_iter;        /* GSIP iteration from which the fused tail kernel runs: >= 0; -2: never; SVSDF_PLAN_AUTO: by rule */
  int settled;          /* get only: 1 once nothing is left to decide for the resident point set */
} svsdf_plan;
int svsdf_get_plan(const svsdf_ctx *ctx, svsdf_plan *out);
int svsdf_set_plan(svsdf_ctx *ctx, const svsdf_plan *plan);
/* In-process multi-device contexts (svsdf_config::n_devices >= 2, or 1 with SVSDF_COMBINE_RCCL): switch how the devices'
 * partials are summed -- SVSDF_COMBINE_HOST (fixed-order sum of the pinned partials) or SVSDF_COMBINE_RCCL (one
 * ncclAllReduce per device thread; the communicator is created on first use, distinct devices required).  Both give the
 * sum of the same G partials; the host form is bit-reproducible, RCCL's order is the library's.  Replaces, for this one
 * sum, what the reference does in its OpenMP critical section (back_end_optimizer.hpp:855-863). */
int svsdf_set_combine(svsdf_ctx *ctx, int combine);
/* Devices driven by the context, combine mode in force, and the rank count of the RCCL communicator as reported by the
 * communicator itself (ncclCommCount; 0 when none exists).  Any out pointer may be NULL. */
int svsdf_group_info(const svsdf_ctx *ctx, int *n_devices, int *combine, int *rccl_ranks);
/* One stripe of a multi-device context (k < n_devices; a single-device context has the one stripe 0): the device it lives
 * on, its point count, the counters / times of ITS part of the last evaluation and the launch plan it follows.  Any out
 * pointer may be NULL.  What an 8-GPU run is judged by -- stripe balance (device_ms max / mean), the plan every stripe
 * picked -- can so be read per stripe; with several stripes on one GPU (svsdf_config::devices repeating an ordinal)
 * svsdf_set_group_serial(ctx, 1) makes every stripe's device_ms its own. */
int svsdf_group_stripe(const svsdf_ctx *ctx, int k, int *device, size_t *points, svsdf_stats *stats, svsdf_plan *plan);
/* Diagnostic: serial != 0 runs the stripes' evaluations one after the other instead of concurrently (same results). */
int svsdf_set_group_serial(svsdf_ctx *ctx, int serial);
/* Shape bound used by the exact scan pruning and the exact cull: out2[0] = R with sdf_shape(q) >= |q| - R
 * (analytic circumradius of the shape + |offset|, Shape.hpp:281-294 / :531-1476), out2[1] = the largest
 * |q| - sdf_shape(q) found on a polar grid out to 60 m at context creation (self-check: <= out2[0]). */
int svsdf_shape_bound(const svsdf_ctx *ctx, double out2[2]);
/* The same two numbers plus out3[2] = the 1-Lipschitz self-check of the shape SDF taken on the same grid at context
 * creation: 0 when |sdf(q') - sdf(q)| <= |q' - q| held for every sampled pair (true of every exact distance function,
 * Shape.hpp:531-1476), otherwise the largest excess found -- the context then runs without the value-based second cull
 * and without the anchor GSIP bound mode, the two devices that rest on that property (same results, more work). */
int svsdf_shape_selfcheck(const svsdf_ctx *ctx, double out3[3]);
/* Per-launch HIP-event timing of the dominant (argmin solve) kernel on the library's own
 * streams; off by default (also env SVSDF_PROFILE=1).  Fills device_ms / solve_ms / solve_ms_sum.
 * enable = 2: additionally runs the point batches one after the other while profiling (a single batch), so that a
 * launch's duration is its own cost rather than stretched by the other batches' kernels it normally overlaps with;
 * enable = 0 restores the split.
 * enable = 3: device_ms only -- the span between the first and the last event of the evaluation, which are recorded
 * anyway; no per-launch events (they cost ~ 5 % of a 500 k-point evaluation), solve_ms / round_ms stay 0. */
int svsdf_set_profiling(svsdf_ctx *ctx, int enable);
/* Self-check: number of n equispaced arguments in [lo, hi] for which the kernels' inlined sincos
 * differs by even one bit from the ROCm device library's sincos (must be 0); -1 on error. */
long long svsdf_debug_sincos_mismatches(svsdf_ctx *ctx, double lo, double hi, int n);
/* Diagnostic / test: getSDFAtTimeStamp<false> (sw_manager.hpp:741-750) of n (point, time) pairs on the device, through
 * the code the solve kernels inline.  points_xy: n x 2, t: n; out8: n x 8 = sdf, pose x, y, cos(yaw), sin(yaw), body-frame
 * x, y of the point, piece-time mode (0 cumulative, 1 / 2 the reference's chain).  The unit of work of the whole path:
 * tests compare it bit for bit with the oracle in device-arithmetic mode for every shape. */
int svsdf_debug_sdf_at(svsdf_ctx *ctx, int N, const double *coeffs_colmajor, const double *T, size_t n,
                       const double *points_xy, const double *t, double *out8);
/* Original indices (into the array given to svsdf_set_points) of this rank's shard, in the
 * order svsdf_query_points reports them. */
int svsdf_shard_indices(const svsdf_ctx *ctx, long long *idx_out);

#ifdef __cplusplus
}
#endif
#endif
