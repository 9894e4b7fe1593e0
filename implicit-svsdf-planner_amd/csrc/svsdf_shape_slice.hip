// svsdf_shape_slice.hip -- one slice of the shape-templated kernels (compile with -DSVSDF_SLICE=k, k = 0 .. 3).
//
// Slice k instantiates k_solve / k_round / k_classify / k_rbound / k_subsw / k_shape_kernels for the shapes with
// id % 4 == k and exports the launchers svsdf_pipeline.hip dispatches to (svsdf_launch.hpp).  Splitting the ~250 kernel
// instantiations over four translation units lets the build run in parallel (one TU took 140 s).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "svsdf_launch.hpp"

#ifndef SVSDF_SLICE
#define SVSDF_SLICE 0   // a bare `hipcc -c` of this file compiles slice 0; build.py passes -DSVSDF_SLICE=0..3
#endif
#define SVSDF_CAT_(a, b) a##b
#define SVSDF_CAT(a, b) SVSDF_CAT_(a, b)
#define SLICE_FN(name) SVSDF_CAT(name, SVSDF_CAT(_s, SVSDF_SLICE))

namespace svsdf {
namespace {

// shape ids of this slice: SVSDF_SLICE + 4 j; ids >= kShapeCount do not exist
template <int S>
constexpr bool shape_enabled() {
#ifdef SVSDF_FAST_BUILD   // development builds: star / sdHorseshoe / sdHeart / Polygon only
  return S == 4 || S == 6 || S == 7 || S == 16;
#else
  return S >= 0 && S < kShapeCount;
#endif
}

// k_solve / k_round also exist for the template value kPolygonLds (Polygon with its edges in LDS, svsdf_shapes.hpp)
template <int S>
constexpr bool kernel_shape_enabled() { return shape_enabled<S>() || S == kPolygonLds; }

template <int S>
bool solve_s(int G, unsigned grid, unsigned block, size_t lds, hipStream_t st, const SolveLaunch &a) {
  if constexpr (!kernel_shape_enabled<S>()) {
    return false;
  } else {
#define SOLVE(GG)                                                                                                   \
  hipLaunchKernelGGL((k_solve<S, GG, 1>), dim3(grid), dim3(block), lds, st, a.traj, a.tk, a.pose, a.chunks, a.sp,    \
                     a.qs, a.out_sdf, a.out_t, a.prune, a.ctl, a.work_idx, a.cull_thresh, a.rot, a.slack_max)
    // G lanes per query (G candidates / samples per step); the U = 2 interleaving (two evaluations per lane) was
    // measured and dropped, only U = 1 is instantiated
    switch (G) {
      case 1: SOLVE(1); break;
      case 2: SOLVE(2); break;
      case 8: SOLVE(8); break;
      case 16: SOLVE(16); break;
      case 32: SOLVE(32); break;
      default: SOLVE(4); break;
    }
#undef SOLVE
    return true;
  }
}

template <int S>
bool round_s(int lp, int mode, unsigned grid, size_t lds, hipStream_t st, const RoundLaunch &a) {
  if constexpr (!kernel_shape_enabled<S>()) {
    return false;
  } else {
#define ROUND(LP, MODE)                                                                                             \
  hipLaunchKernelGGL((k_round<S, LP, MODE>), dim3(grid), dim3(kRoundBlock), lds, st, a.traj, a.pose, a.chunks, a.sp, \
                     a.px, a.py, a.gs, a.stride, a.it, a.delta, a.band_delta, a.res_sdf, a.res_t, a.res_gx,          \
                     a.res_gy, a.ctl, a.clist_on)
    if (lp == 8) {
      if (mode == 3) ROUND(8, 3); else if (mode == 2) ROUND(8, 2); else if (mode == 1) ROUND(8, 1); else ROUND(8, 0);
    } else {
      if (mode == 3) ROUND(32, 3); else if (mode == 2) ROUND(32, 2); else if (mode == 1) ROUND(32, 1); else ROUND(32, 0);
    }
#undef ROUND
    return true;
  }
}

template <int S>
bool tail_s(int mode, unsigned grid, size_t lds, hipStream_t st, const TailLaunch &a) {
  if constexpr (!kernel_shape_enabled<S>()) {
    return false;
  } else {
#define TAIL_W(MODE, WAVES)                                                                                                 \
  hipLaunchKernelGGL((k_tail<S, MODE, WAVES>), dim3(grid), dim3(kTailBlock), lds, st, a.traj, a.tk, a.pose, a.chunks, a.sp,  \
                     a.px, a.py, a.gs, a.stride, a.it0, a.prev_mode, a.delta, a.band_delta, a.all_after, a.ppw, a.res_sdf,   \
                     a.res_t, a.res_gx, a.res_gy, a.ctl, a.clist_on, a.prune)
#define TAIL(MODE) do { if (a.latency) TAIL_W(MODE, kTailLatencyWaves); else TAIL_W(MODE, SVSDF_TAIL_WAVES); } while (0)
    if (mode == 3) TAIL(3); else if (mode == 2) TAIL(2); else if (mode == 1) TAIL(1); else TAIL(0);
#undef TAIL
#undef TAIL_W
    return true;
  }
}

template <int S>
bool classify_s(unsigned grid, size_t lds, hipStream_t st, const ClassifyLaunch &a) {
  if constexpr (!shape_enabled<S>()) {
    return false;
  } else {
    hipLaunchKernelGGL((k_classify<S>), dim3(grid), dim3(kBlock), lds, st, a.traj, a.sp, a.px, a.py, a.sdf, a.t,
                       a.res_sdf, a.res_t, a.res_gx, a.res_gy, a.gs, a.ctl, a.n_int, a.icap);
    return true;
  }
}

template <int S>
bool rbound_s(unsigned grid, hipStream_t st, ShapeParams sp, double rmax, int nrad, int nang, double *out) {
  if constexpr (!shape_enabled<S>()) {
    return false;
  } else {
    hipLaunchKernelGGL((k_rbound<S>), dim3(grid), dim3(kBlock), 0, st, sp, rmax, nrad, nang, out);
    return true;
  }
}

template <int S>
bool subsw_s(dim3 grid, hipStream_t st, ShapeParams sp, const double *father, const double *child,
             const unsigned long long *offs, const double *pts, const double *kt, int nkt, int *flag) {
  if constexpr (!shape_enabled<S>()) {
    return false;
  } else {
    hipLaunchKernelGGL((k_subsw<S>), grid, dim3(kSubswBlock), 0, st, sp, father, child, offs, pts, kt, nkt, flag);
    return true;
  }
}

template <int S>
bool shape_kernels_s(unsigned grid, hipStream_t st, ShapeParams sp, int ks, int count, double resu, int size_side,
                     double safemargin, const double *yaw, unsigned char *map) {
  if constexpr (!shape_enabled<S>() || is_polygon<S>()) {   // Polygon has no (pos_rel, R_obj) overload (SHP:1477)
    return false;
  } else {
    hipLaunchKernelGGL((k_shape_kernels<S>), dim3(grid), dim3(kBlock), 0, st, sp, ks, count, resu, size_side, safemargin, yaw, map);
    return true;
  }
}

template <int S>
bool debug_sdf_at_s(unsigned grid, size_t lds, hipStream_t st, const TrajDev *traj, ShapeParams sp, const double *pxy,
                    const double *t, int n, double *out) {
  if constexpr (!shape_enabled<S>()) {
    return false;
  } else {
    hipLaunchKernelGGL((k_debug_sdf_at<S>), dim3(grid), dim3(64), lds, st, traj, sp, pxy, t, n, out);
    return true;
  }
}

}  // namespace

#define SLICE_SWITCH(CALL)                                        \
  switch (shape) {                                                \
    case SVSDF_SLICE: return CALL(SVSDF_SLICE);                   \
    case SVSDF_SLICE + 4: return CALL(SVSDF_SLICE + 4);           \
    case SVSDF_SLICE + 8: return CALL(SVSDF_SLICE + 8);           \
    case SVSDF_SLICE + 12: return CALL(SVSDF_SLICE + 12);         \
    case SVSDF_SLICE + 16: return CALL(SVSDF_SLICE + 16);         \
    default: return false;                                        \
  }

bool SLICE_FN(launch_k_solve)(int shape, int G, unsigned grid, unsigned block, size_t lds, hipStream_t st, const SolveLaunch &a) {
#define CALL(S) solve_s<S>(G, grid, block, lds, st, a)
  SLICE_SWITCH(CALL)
#undef CALL
}
bool SLICE_FN(launch_k_round)(int shape, int lp, int mode, unsigned grid, size_t lds, hipStream_t st, const RoundLaunch &a) {
#define CALL(S) round_s<S>(lp, mode, grid, lds, st, a)
  SLICE_SWITCH(CALL)
#undef CALL
}
bool SLICE_FN(launch_k_tail)(int shape, int mode, unsigned grid, size_t lds, hipStream_t st, const TailLaunch &a) {
#define CALL(S) tail_s<S>(mode, grid, lds, st, a)
  SLICE_SWITCH(CALL)
#undef CALL
}
bool SLICE_FN(launch_k_classify)(int shape, unsigned grid, size_t lds, hipStream_t st, const ClassifyLaunch &a) {
#define CALL(S) classify_s<S>(grid, lds, st, a)
  SLICE_SWITCH(CALL)
#undef CALL
}
bool SLICE_FN(launch_k_rbound)(int shape, unsigned grid, hipStream_t st, ShapeParams sp, double rmax, int nrad, int nang, double *out) {
#define CALL(S) rbound_s<S>(grid, st, sp, rmax, nrad, nang, out)
  SLICE_SWITCH(CALL)
#undef CALL
}
bool SLICE_FN(launch_k_subsw)(int shape, dim3 grid, hipStream_t st, ShapeParams sp, const double *father, const double *child,
                              const unsigned long long *offs, const double *pts, const double *kt, int nkt, int *flag) {
#define CALL(S) subsw_s<S>(grid, st, sp, father, child, offs, pts, kt, nkt, flag)
  SLICE_SWITCH(CALL)
#undef CALL
}
bool SLICE_FN(launch_k_debug_sdf_at)(int shape, unsigned grid, size_t lds, hipStream_t st, const TrajDev *traj, ShapeParams sp,
                                     const double *pxy, const double *t, int n, double *out) {
#define CALL(S) debug_sdf_at_s<S>(grid, lds, st, traj, sp, pxy, t, n, out)
  SLICE_SWITCH(CALL)
#undef CALL
}
bool SLICE_FN(launch_k_shape_kernels)(int shape, unsigned grid, hipStream_t st, ShapeParams sp, int ks, int count, double resu,
                                      int size_side, double safemargin, const double *yaw, unsigned char *map) {
#define CALL(S) shape_kernels_s<S>(grid, st, sp, ks, count, resu, size_side, safemargin, yaw, map)
  SLICE_SWITCH(CALL)
#undef CALL
}

}  // namespace svsdf
