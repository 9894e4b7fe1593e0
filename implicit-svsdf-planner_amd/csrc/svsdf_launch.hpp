// svsdf_launch.hpp -- host-side launchers of the shape-templated kernels.
//
// The kernels are specialised per shape id (17 shapes x lane-group widths x GSIP bound modes: ~250 kernels).  They are
// compiled in SVSDF_NSLICES translation units (svsdf_shape_slice.hip with -DSVSDF_SLICE=k holds the shapes with
// id % SVSDF_NSLICES == k) so that the build runs in parallel; svsdf_pipeline.hip only sees these plain functions.
#pragma once
#include <hip/hip_runtime.h>

#include "svsdf_kernels.hpp"
#include "svsdf_frontend.hpp"

namespace svsdf {

constexpr int kNSlices = 4;

struct SolveLaunch {   // arguments of k_solve<SHAPE, G, 1>
  const TrajDev *traj; const double *tk; const Pose *pose; const Chunk *chunks; ShapeParams sp; QuerySet qs;
  double *out_sdf, *out_t; int prune; BatchCtl *ctl; int work_idx; double cull_thresh;
  const double *rot; double slack_max;   // second exact cull (main points): per-chunk yaw allowance W_c h, max_c of the linear one
};
struct RoundLaunch {   // arguments of k_round<SHAPE, LP, MODE>
  const TrajDev *traj; const Pose *pose; const Chunk *chunks; ShapeParams sp; const double *px, *py; GsipState gs;
  size_t stride; int it; double delta, band_delta; double *res_sdf, *res_t, *res_gx, *res_gy; BatchCtl *ctl;
  int clist_on;   // 1: scans / cheap bounds walk the per-point candidate-chunk list (0: all chunks; same results)
};
struct TailLaunch {    // arguments of k_tail<SHAPE, MODE, WAVES>
  const TrajDev *traj; const double *tk; const Pose *pose; const Chunk *chunks; ShapeParams sp; const double *px, *py;
  GsipState gs; size_t stride; int it0, prev_mode; double delta, band_delta; int all_after, ppw;
  double *res_sdf, *res_t, *res_gx, *res_gy; BatchCtl *ctl; int clist_on, prune;
  int latency;   // not a kernel argument: 1 = the instantiation that keeps its registers (kTailLatencyWaves per SIMD, no scratch)
};
struct ClassifyLaunch {   // arguments of k_classify<SHAPE>
  const TrajDev *traj; ShapeParams sp; const double *px, *py, *sdf, *t; double *res_sdf, *res_t, *res_gx, *res_gy;
  GsipState gs; BatchCtl *ctl; int *n_int; int icap;
};

// each returns false when the shape id is not compiled into the library (development builds)
bool launch_k_solve(int shape, int G, unsigned grid, unsigned block, size_t lds, hipStream_t st, const SolveLaunch &a);
bool launch_k_round(int shape, int lp, int mode, unsigned grid, size_t lds, hipStream_t st, const RoundLaunch &a);
bool launch_k_classify(int shape, unsigned grid, size_t lds, hipStream_t st, const ClassifyLaunch &a);
bool launch_k_tail(int shape, int mode, unsigned grid, size_t lds, hipStream_t st, const TailLaunch &a);
bool launch_k_rbound(int shape, unsigned grid, hipStream_t st, ShapeParams sp, double rmax, int nrad, int nang, double *out);
bool launch_k_subsw(int shape, dim3 grid, hipStream_t st, ShapeParams sp, const double *father, const double *child,
                    const unsigned long long *offs, const double *pts, const double *kt, int nkt, int *flag);
bool launch_k_shape_kernels(int shape, unsigned grid, hipStream_t st, ShapeParams sp, int ks, int count, double resu,
                            int size_side, double safemargin, const double *yaw, unsigned char *map);
bool launch_k_debug_sdf_at(int shape, unsigned grid, size_t lds, hipStream_t st, const TrajDev *traj, ShapeParams sp,
                           const double *pxy, const double *t, int n, double *out);

// per-slice entry points (defined by svsdf_shape_slice.hip, one set per slice)
#define SVSDF_DECLARE_SLICE(K)                                                                                              \
  bool launch_k_solve_s##K(int shape, int G, unsigned grid, unsigned block, size_t lds, hipStream_t st, const SolveLaunch &a); \
  bool launch_k_round_s##K(int shape, int lp, int mode, unsigned grid, size_t lds, hipStream_t st, const RoundLaunch &a);      \
  bool launch_k_classify_s##K(int shape, unsigned grid, size_t lds, hipStream_t st, const ClassifyLaunch &a);                  \
  bool launch_k_tail_s##K(int shape, int mode, unsigned grid, size_t lds, hipStream_t st, const TailLaunch &a);                \
  bool launch_k_rbound_s##K(int shape, unsigned grid, hipStream_t st, ShapeParams sp, double rmax, int nrad, int nang, double *out); \
  bool launch_k_subsw_s##K(int shape, dim3 grid, hipStream_t st, ShapeParams sp, const double *father, const double *child,    \
                           const unsigned long long *offs, const double *pts, const double *kt, int nkt, int *flag);         \
  bool launch_k_shape_kernels_s##K(int shape, unsigned grid, hipStream_t st, ShapeParams sp, int ks, int count, double resu,   \
                                   int size_side, double safemargin, const double *yaw, unsigned char *map);                   \
  bool launch_k_debug_sdf_at_s##K(int shape, unsigned grid, size_t lds, hipStream_t st, const TrajDev *traj, ShapeParams sp,    \
                                  const double *pxy, const double *t, int n, double *out);
SVSDF_DECLARE_SLICE(0)
SVSDF_DECLARE_SLICE(1)
SVSDF_DECLARE_SLICE(2)
SVSDF_DECLARE_SLICE(3)
#undef SVSDF_DECLARE_SLICE

}  // namespace svsdf
