// svsdf_frontend.hpp -- SURVEY.md §8 row f3: the front end's consumers of the same shape SDFs.
//
//   k_subsw<SHAPE>         SweptVolumeManager::checkSubSWCollision (SWM:1171-1211), batched over A* edges
//   k_shape_kernels<SHAPE> BasicShape::initShape (SHP:386-430): occupancy of the kernel cells per yaw
//
// Both are maps over independent (edge, obstacle point, interpolation step) / (yaw, cell) items with a
// boolean reduction; the reference's early exits only shorten its loops, they never change the result, so
// the order of evaluation is free.  FP64, strict arithmetic (no contraction), same operation order as the
// reference's Eigen expressions.
#pragma once
#include "svsdf_kernels.hpp"

namespace svsdf {

constexpr int kSubswPoints = 64;   // obstacle points per block (one per lane)
constexpr int kSubswBlock = 256;   // 4 waves share the interpolation steps of those points
constexpr int kMaxKt = 64;         // interpolation steps per edge (the reference loop yields 50: kt = 0 ... 0.98)

// One block = one (edge, 64-point chunk).  LDS holds the edge's interpolated poses
// linear_state(kt) = kt*child + (1-kt)*father (SWM:1191) with sin/cos of its yaw; lane = point, wave w takes
// the steps w, w+4, ...  hit_flag[e] starts at 0 and is set by any (point, step) with sdf < 0
// (SWM:1201-1204: `min_sdf < 0` can only become true through the current temp_sdf).
template <int SHAPE>
__global__ void __launch_bounds__(kSubswBlock)
k_subsw(ShapeParams sp, const double *__restrict__ father, const double *__restrict__ child,
        const unsigned long long *__restrict__ offs, const double *__restrict__ pts_xy,
        const double *__restrict__ kt_tab, int nkt, int *__restrict__ hit_flag) {
  __shared__ double s_x[kMaxKt], s_y[kMaxKt], s_c[kMaxKt], s_s[kMaxKt];
  const unsigned e = blockIdx.x;
  const unsigned long long p0 = offs[e], p1 = offs[e + 1];
  const unsigned long long first = p0 + (unsigned long long)blockIdx.y * kSubswPoints;
  if (first >= p1) return;
  if (threadIdx.x < (unsigned)nkt) {
    const double kt = kt_tab[threadIdx.x];
    const double omk = 1 - kt;
    const double lx = kt * child[3 * e + 0] + omk * father[3 * e + 0];
    const double ly = kt * child[3 * e + 1] + omk * father[3 * e + 1];
    const double yaw = kt * child[3 * e + 2] + omk * father[3 * e + 2];
    double sn, cs;
    sincos_exact(yaw, &sn, &cs);
    s_x[threadIdx.x] = lx; s_y[threadIdx.x] = ly; s_c[threadIdx.x] = cs; s_s[threadIdx.x] = sn;
  }
  __syncthreads();
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long pi = first + lane;
  bool hit = false;
  if (pi < p1) {
    const double px = pts_xy[2 * pi], py = pts_xy[2 * pi + 1];
    for (int k = (int)wave; k < nkt; k += kSubswBlock / 64) {
      const double dx = px - s_x[k], dy = py - s_y[k];
      const double c = s_c[k], s = s_s[k];
      const double rx = c * dx + s * dy;       // posEva2Rel: Rt^T (p - x)  SWM:521-526
      const double ry = (-s) * dx + c * dy;
      if (shape_sdf<SHAPE>(sp, rx, ry) < 0) { hit = true; break; }
    }
  }
  if (__any(hit)) {
    if (lane == 0) hit_flag[e] = 1;
  }
}

// One thread per (yaw index, a, b) cell: x = resu*a - size_side*resu, y likewise (SHP:413-414),
// occupied iff getonlySDF(pos, R(yaw)) <= safemargin (SHP:418-423).
template <int SHAPE>
__global__ void __launch_bounds__(kBlock)
k_shape_kernels(ShapeParams sp, int ks, int count, double resu, int size_side, double safemargin,
                const double *__restrict__ yaw_tab, unsigned char *__restrict__ map) {
  const int cells = ks * ks;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)cells * count) return;
  const int ind = (int)(gid / cells);
  const int ab = (int)(gid - (long long)ind * cells);
  const int a = ab / ks, b = ab - a * ks;
  const double x = resu * a - size_side * resu;
  const double y = resu * b - size_side * resu;
  double sn, cs;
  sincos_exact(yaw_tab[ind], &sn, &cs);
  if constexpr (SHAPE == kPolygon) {
    map[gid] = 0;
  } else {
    map[gid] = (shape_sdf_rot<SHAPE>(sp, x, y, cs, sn) <= safemargin) ? 1 : 0;
  }
}

// Diagnostic / test kernel (svsdf_debug_sdf_at): getSDFAtTimeStamp<false> (SWM:741-750) for arbitrary (point, time) pairs
// through the very pose_at / sdf_from_pose the solve kernels inline, with the intermediates: out[8 i ..] = sdf, x, y, cos,
// sin, body-frame x, body-frame y, piece-local-time path taken (0 cumulative, 1 / 2 chain).  One wave per block: the
// faithful piece-time chain works on whole waves.
template <int SHAPE>
__global__ void __launch_bounds__(64)
k_debug_sdf_at(const TrajDev *__restrict__ trg, ShapeParams sp, const double *__restrict__ pxy, const double *__restrict__ t_,
               int n, double *__restrict__ out) {
  extern __shared__ double dbg_lds[];
  const TrajL tr = stage_traj(trg, dbg_lds);
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  PieceCache pc = piece_cache_init();
  const Pose p = pose_at(tr, t_[i], pc);
  const double px = pxy[2 * i], py = pxy[2 * i + 1];
  const double dx = px - p.x, dy = py - p.y;
  const double rx = p.cs * dx + p.sn * dy;
  const double ry = (-p.sn) * dx + p.cs * dy;
  double *o = out + 8 * (size_t)i;
  o[0] = sdf_from_pose<SHAPE>(sp, p, px, py);
  o[1] = p.x; o[2] = p.y; o[3] = p.cs; o[4] = p.sn; o[5] = rx; o[6] = ry; o[7] = (double)tr.exact;
}

}  // namespace svsdf
