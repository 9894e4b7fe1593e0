// svsdf_kernels.hpp -- gfx950 device kernels of the SVSDF cost/gradient pipeline.
//
// Behavioural spec (what, not how): reference
//   SWM = src/swept_volume/include/swept_volume/sw_manager.hpp
//   BEO = src/planner_algorithm/include/planner_algorithm/back_end_optimizer.hpp
//   TRJ = src/utils/include/utils/trajectory.hpp
// Pipeline per evaluation (all FP64, one stream):
//   k_prep      trajectory -> per-piece monomials, cumulative start times, layer-1 pose table
//   k_solve     one argmin-over-t solve per query (main points and GSIP circle samples)
//   k_classify  exterior: FD gradient; interior: GSIP state + first circle samples
//   k_gsip      per GSIP round: max over samples, radius update, termination / next samples
//   k_assemble  per-point cost/gradient contribution + block-level segmented reduction
//   k_final     fixed-order sum of block partials, suffix sum for the duration gradient
#pragma once
#include <hip/hip_runtime.h>

#include "svsdf_shapes.hpp"

namespace svsdf {

constexpr int kMaxPieces = 64;
constexpr int kMaxSlots = 24;  // GSIP samples per round: 2, 6, 18, 21, 21, ... (SWM:60-71,105-110)
constexpr int kBlock = 256;

// Trajectory as the device sees it (global memory; staged into LDS by each block).
struct TrajDev {
  int N;
  int K;        // layer-1 samples (t = 0; t <= dur; t += 0.15, SWM:567)
  double dur;   // SweptVolumeManager::traj_duration (SWM:376-385)
  double T[kMaxPieces];
  double S[kMaxPieces + 1];     // S[i] = T[0] + ... + T[i-1] (sequential sum)
  double c[kMaxPieces * 18];    // c[(i*6 + k)*3 + d]: coefficient of s^k, dim d (x, y, yaw)
};

struct Pose { double x, y, cs, sn; };

// LDS view of the trajectory
struct TrajL {
  const double *T, *S, *c;
  int N;
  double dur;
};

constexpr int kTrajLdsDoubles = kMaxPieces * 18 + kMaxPieces + (kMaxPieces + 1);

__device__ __forceinline__ TrajL stage_traj(const TrajDev *__restrict__ g, double *lds) {
  const int N = g->N;
  double *T = lds, *S = lds + kMaxPieces, *c = lds + 2 * kMaxPieces + 1;
  for (int i = threadIdx.x; i < N; i += blockDim.x) T[i] = g->T[i];
  for (int i = threadIdx.x; i <= N; i += blockDim.x) S[i] = g->S[i];
  for (int i = threadIdx.x; i < 18 * N; i += blockDim.x) c[i] = g->c[i];
  __syncthreads();
  TrajL tr;
  tr.T = T; tr.S = S; tr.c = c; tr.N = N; tr.dur = g->dur;
  return tr;
}

// Trajectory::locatePieceIdx (TRJ:498-516) on cumulative start times: piece = first i with
// t <= S[i+1] (clamped to N-1), local time = t - S[i].  `i` is a per-lane cache of the last piece.
__device__ __forceinline__ int locate_piece(const TrajL &tr, double t, int i) {
  while (i < tr.N - 1 && t > tr.S[i + 1]) ++i;
  while (i > 0 && !(t > tr.S[i])) --i;
  return i;
}

// Piece::getPos (TRJ:104-114): pos = sum_k tn_k * c_k with tn built by repeated multiplication.
__device__ __forceinline__ void piece_pos(const double *__restrict__ c, double s, double &x,
                                          double &y, double &yaw) {
  const double s2 = s * s, s3 = s2 * s, s4 = s3 * s, s5 = s4 * s;
  x = c[0]; y = c[1]; yaw = c[2];
  x += s * c[3];   y += s * c[4];   yaw += s * c[5];
  x += s2 * c[6];  y += s2 * c[7];  yaw += s2 * c[8];
  x += s3 * c[9];  y += s3 * c[10]; yaw += s3 * c[11];
  x += s4 * c[12]; y += s4 * c[13]; yaw += s4 * c[14];
  x += s5 * c[15]; y += s5 * c[16]; yaw += s5 * c[17];
}

// Piece::getVel (TRJ:116-128): vel += (n * tn) * c_k
__device__ __forceinline__ void piece_vel(const double *__restrict__ c, double s, double &vx,
                                          double &vy, double &w) {
  double tn = 1.0;
  vx = 0.0; vy = 0.0; w = 0.0;
#pragma unroll
  for (int k = 1; k <= 5; ++k) {
    const double f = (double)k * tn;
    vx += f * c[k * 3 + 0];
    vy += f * c[k * 3 + 1];
    w += f * c[k * 3 + 2];
    tn *= s;
  }
}

__device__ __forceinline__ Pose pose_at(const TrajL &tr, double t, int &piece) {
  piece = locate_piece(tr, t, piece);
  const double s = t - tr.S[piece];
  double x, y, yaw;
  piece_pos(tr.c + piece * 18, s, x, y, yaw);
  Pose p;
  p.x = x; p.y = y;
  sincos(yaw, &p.sn, &p.cs);  // Rt = AngleAxisd(yaw, Z)  (SWM:465-474)
  return p;
}

// posEva2Rel (SWM:521-526): Rt^T (p - xt), then the shape SDF.
template <int SHAPE>
__device__ __forceinline__ double sdf_from_pose(const ShapeParams &sp, const Pose &p, double px,
                                                double py) {
  const double dx = px - p.x, dy = py - p.y;
  const double rx = p.cs * dx + p.sn * dy;
  const double ry = (-p.sn) * dx + p.cs * dy;
  return shape_sdf<SHAPE>(sp, rx, ry);
}

// getSDFAtTimeStamp<false> (SWM:741-750)
template <int SHAPE>
__device__ __forceinline__ double sdf_at(const TrajL &tr, const ShapeParams &sp, double px,
                                         double py, double t, int &piece) {
  const Pose p = pose_at(tr, t, piece);
  return sdf_from_pose<SHAPE>(sp, p, px, py);
}

// ---------------------------------------------------------------------------------------------
// k_prep: one block.  coeffs is the reference's (6N) x 3 column-major matrix.
// ---------------------------------------------------------------------------------------------
__global__ void k_prep(const double *__restrict__ coeffs, const double *__restrict__ T, int N,
                       double dur, int K, const double *__restrict__ tk, TrajDev *__restrict__ tr,
                       Pose *__restrict__ pose) {
  __shared__ double lds[kTrajLdsDoubles];
  if (threadIdx.x == 0) {
    tr->N = N; tr->K = K; tr->dur = dur;
    double s = 0.0;
    for (int i = 0; i < N; ++i) { tr->T[i] = T[i]; tr->S[i] = s; s += T[i]; }
    tr->S[N] = s;
  }
  for (int e = threadIdx.x; e < 18 * N; e += blockDim.x) {
    const int i = e / 18, k = (e % 18) / 3, d = e % 3;
    tr->c[e] = coeffs[(size_t)d * 6 * N + 6 * i + k];
  }
  __threadfence_block();
  __syncthreads();
  const TrajL tl = stage_traj(tr, lds);
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    int piece = 0;
    pose[k] = pose_at(tl, tk[k], piece);
  }
}

// ---------------------------------------------------------------------------------------------
// k_solve: getSDFofSweptVolume<false, true> without the (separately computed) FD gradient
// (SWM:844-866) = choiceTInit (SWM:538-581) + gradientDescent (SWM:1249-1325).
// Query q -> array slot (q / n_inner) * stride + (q % n_inner); NaN qx marks an unused slot.
// ---------------------------------------------------------------------------------------------
template <int SHAPE>
__device__ __forceinline__ void solve_one(const TrajL &tr, const ShapeParams &sp,
                                          const double *__restrict__ tk,
                                          const Pose *__restrict__ pose, int K, double px, double py,
                                          double &sdf_star, double &t_star, unsigned &n_scan,
                                          unsigned &n_eval) {
  // ---- choiceTInit layer 1: shared pose table (poses at the scan times do not depend on p)
  double min_dis = 1e9;
  int kbest = 0;
  for (int k = 0; k < K; ++k) {
    const Pose p = pose[k];
    const double dis = sdf_from_pose<SHAPE>(sp, p, px, py);
    if (dis < min_dis) { kbest = k; min_dis = dis; }
  }
  n_scan += K;
  double time_seed = tk[kbest];
  // ---- layers 2-4
  int piece = 0;
  double dt = 0.15;
  dt *= 0.1;
  for (int layer = 2; layer <= 4; ++layer) {
    double t = dmax(0.0, time_seed - 10 * dt);
    const double loop_terminal = dmin(tr.dur, time_seed + 10 * dt);
    for (; t <= loop_terminal; t += dt) {
      const double dis = sdf_at<SHAPE>(tr, sp, px, py, t, piece);
      ++n_eval;
      if (dis < min_dis) { time_seed = t; min_dis = dis; }
    }
    dt *= 0.1;
  }
  // ---- gradientDescent on [ts - 3.4, ts + 3.4] ∩ [0, dur]
  const double t_min = dmax(0.0, time_seed - 3.4);
  const double t_max = dmin(time_seed + 3.4, tr.dur);
  double x = time_seed, fx = 0.0, prev_x = 10000000.0;
  int iter = 0;
  bool stop = false;
  while (iter < 1000 && !stop && fabs(x - prev_x) > 1e-16) {
    if (iter == 0) { fx = sdf_at<SHAPE>(tr, sp, px, py, x, piece); ++n_eval; }
    // getSDF_DOTAtTimeStamp (SWM:799-806): x is unchanged inside the halving ladder, so the
    // reference's per-trial re-evaluation returns this same number -- evaluate it once.
    const double t1 = dmax(0.0, x - 0.000001);
    const double t2 = dmin(tr.dur, x + 0.000001);
    const double sdf1 = sdf_at<SHAPE>(tr, sp, px, py, t1, piece);
    const double sdf2 = sdf_at<SHAPE>(tr, sp, px, py, t2, piece);
    n_eval += 2;
    const double g = (sdf2 - sdf1) * 500000;
    const double sgn = (double)((int)(g > 0) - (int)(g < 0));
    double tau = 0.01;
    prev_x = x;
    for (int div = 1; div < 30; ++div) {
      iter = iter + 1;
      const double change = -tau * sgn;
      double xc = x + change;
      xc = dmax(dmin(xc, t_max), t_min);
      const double fc = sdf_at<SHAPE>(tr, sp, px, py, xc, piece);
      ++n_eval;
      if ((fc - fx) < 0) { x = xc; fx = fc; break; }
      tau = 0.5 * tau;
      if (div == 29) stop = true;
    }
  }
  sdf_star = fx;
  t_star = x;
}

template <int SHAPE>
__global__ void __launch_bounds__(kBlock)
k_solve(const TrajDev *__restrict__ trg, const double *__restrict__ tk,
        const Pose *__restrict__ pose, ShapeParams sp, const double *__restrict__ qx,
        const double *__restrict__ qy, int n_inner, int n_outer, size_t stride,
        double *__restrict__ out_sdf, double *__restrict__ out_t,
        unsigned long long *__restrict__ stats) {
  __shared__ double lds[kTrajLdsDoubles];
  const TrajL tr = stage_traj(trg, lds);
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (long long)n_inner * n_outer) return;
  const size_t slot = (size_t)(q / n_inner) * stride + (size_t)(q % n_inner);
  const double px = qx[slot], py = qy[slot];
  if (px != px) return;  // unused slot
  double sdf_star, t_star;
  unsigned n_scan = 0, n_eval = 0;
  solve_one<SHAPE>(tr, sp, tk, pose, trg->K, px, py, sdf_star, t_star, n_scan, n_eval);
  out_sdf[slot] = sdf_star;
  out_t[slot] = t_star;
  atomicAdd(&stats[0], 1ull);
  atomicAdd(&stats[1], (unsigned long long)n_eval + n_scan);
  atomicAdd(&stats[2], (unsigned long long)n_scan);
}

// ---------------------------------------------------------------------------------------------
// GSIP state (getTrueSDFofSweptVolume SWM:916-1018), one entry per interior point.
// ---------------------------------------------------------------------------------------------
struct GsipState {
  int *pt;          // index of the (sorted) main point
  double *r;        // current circle radius
  double *theta0;   // first sample angle of the current round
  double *theta_res;
  int *iter;        // 1..9
  int *nsamp;       // samples emitted for the current round
  int *done;
  // sub-query slots [j * stride + a]
  double *sqx, *sqy, *sqth, *sq_sdf, *sq_t;
};

// SampleSet2D::getElements + getElementPos (SWM:36-39, 60-71) for the single ring rk = 1.0.
__device__ __forceinline__ int emit_samples(const GsipState &gs, int a, size_t stride, double cx,
                                            double cy, double r, double theta0, double theta_res) {
  int n = 0;
  for (double theta = theta0; theta < theta0 + 2 * kPI; theta += theta_res) {
    if (n < kMaxSlots) {
      const size_t s = (size_t)n * stride + a;
      gs.sqx[s] = cx + 1.0 * r * cos(theta);
      gs.sqy[s] = cy + 1.0 * r * sin(theta);
      gs.sqth[s] = theta;
    }
    ++n;
  }
  n = n < kMaxSlots ? n : kMaxSlots;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  for (int j = n; j < kMaxSlots; ++j) gs.sqx[(size_t)j * stride + a] = nan;
  return n;
}

// Per main point after the first solve: exterior -> FD gradient (getGradPrelAtTimeStamp,
// SWM:779-788) and done; interior -> GSIP init (SWM:926-963).
template <int SHAPE>
__global__ void __launch_bounds__(kBlock)
k_classify(const TrajDev *__restrict__ trg, ShapeParams sp, const double *__restrict__ px_,
           const double *__restrict__ py_, int P, const double *__restrict__ sdf_,
           const double *__restrict__ t_, double *__restrict__ res_sdf,
           double *__restrict__ res_t, double *__restrict__ res_gx, double *__restrict__ res_gy,
           GsipState gs, size_t stride, int *__restrict__ n_interior) {
  __shared__ double lds[kTrajLdsDoubles];
  const TrajL tr = stage_traj(trg, lds);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const double px = px_[i], py = py_[i];
  const double sdf = sdf_[i], ts = t_[i];
  if (sdf > 0) {  // outside case (SWM:921-924)
    int piece = 0;
    const Pose p = pose_at(tr, ts, piece);
    const double dx = px - p.x, dy = py - p.y;
    const double rx = p.cs * dx + p.sn * dy;
    const double ry = (-p.sn) * dx + p.cs * dy;
    double gx, gy;
    shape_grad<SHAPE>(sp, rx, ry, gx, gy);
    res_sdf[i] = sdf; res_t[i] = ts; res_gx[i] = gx; res_gy[i] = gy;
    return;
  }
  // interior: velocity at t* with the low-speed rescans (SWM:929-954)
  int piece = locate_piece(tr, ts, 0);
  double vx, vy, w;
  piece_vel(tr.c + piece * 18, ts - tr.S[piece], vx, vy, w);
  if (sqrt(vx * vx + vy * vy + w * w) < 0.01) {
    if (ts < 0.1) {
      for (double t_scan = ts; t_scan <= tr.dur; t_scan += 0.1) {
        piece = locate_piece(tr, t_scan, piece);
        piece_vel(tr.c + piece * 18, t_scan - tr.S[piece], vx, vy, w);
        if (sqrt(vx * vx + vy * vy + w * w) >= 0.01) break;
      }
    } else if (ts > tr.dur - 0.1) {
      for (double t_scan = ts; t_scan >= 0; t_scan -= 0.1) {
        piece = locate_piece(tr, t_scan, piece);
        piece_vel(tr.c + piece * 18, t_scan - tr.S[piece], vx, vy, w);
        if (sqrt(vx * vx + vy * vy + w * w) >= 0.01) break;
      }
    }
  }
  const int a = atomicAdd(n_interior, 1);
  // SampleSet2D::initSet (SWM:73-103)
  double theta0 = atan2(vx, -vy);
  if (theta0 < 0) theta0 += 2 * kPI;
  const double theta_res = kPI + 0.1;
  const double r0 = 10;
  gs.pt[a] = i;
  gs.r[a] = r0;
  gs.theta0[a] = theta0;
  gs.theta_res[a] = theta_res;
  gs.iter[a] = 1;
  gs.done[a] = 0;
  gs.nsamp[a] = emit_samples(gs, a, stride, px, py, r0, theta0, theta_res);
  res_t[i] = ts;  // real_t_star fallback
}

// One GSIP round per interior point (SWM:965-1009) and the final assembly (SWM:1011-1017).
__global__ void __launch_bounds__(kBlock)
k_gsip(const double *__restrict__ px_, const double *__restrict__ py_, GsipState gs, size_t stride,
       int n_int, double *__restrict__ res_sdf, double *__restrict__ res_t,
       double *__restrict__ res_gx, double *__restrict__ res_gy, int *__restrict__ n_active) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_int) return;
  if (gs.done[a]) return;
  const int i = gs.pt[a];
  const double cx = px_[i], cy = py_[i];
  double max_g = -100000;
  double real_t = res_t[i], star_th = 0.0;
  const int n = gs.nsamp[a];
  for (int j = 0; j < n; ++j) {
    const size_t s = (size_t)j * stride + a;
    const double cur_g = gs.sq_sdf[s];
    if (cur_g > max_g) { max_g = cur_g; real_t = gs.sq_t[s]; star_th = gs.sqth[s]; }
  }
  const double r_star = gs.r[a] - max_g;
  const int iter = gs.iter[a];
  if (iter > 8 || fabs(max_g) < 0.1) {
    const double corx = cx + 1.0 * r_star * cos(star_th);
    const double cory = cy + 1.0 * r_star * sin(star_th);
    double gx = corx - cx, gy = cory - cy;
    const double z = gx * gx + gy * gy;
    if (z > 0.0) { const double nn = sqrt(z); gx = gx / nn; gy = gy / nn; }
    res_sdf[i] = -r_star; res_t[i] = real_t; res_gx[i] = gx; res_gy[i] = gy;
    gs.done[a] = 1;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (int j = 0; j < kMaxSlots; ++j) gs.sqx[(size_t)j * stride + a] = nan;
    return;
  }
  // expandSet(2, theta*) (SWM:105-110)
  double theta_res = gs.theta_res[a] / (2 + 1);
  theta_res = dmax(0.3, theta_res);
  gs.r[a] = r_star;
  gs.theta_res[a] = theta_res;
  gs.theta0[a] = star_th;
  gs.iter[a] = iter + 1;
  res_t[i] = real_t;
  gs.nsamp[a] = emit_samples(gs, a, stride, cx, cy, r_star, star_th, theta_res);
  atomicAdd(n_active, 1);
}

// ---------------------------------------------------------------------------------------------
// k_assemble: loop body of BEO:786-865 for one point given (sdf, t*, grad_prel), then a
// block-level segmented reduction keyed by piece.  partial layout (per block and final):
//   [0] cost, [1 .. 18N] gradC column-major ((6N) x 3), [18N+1 .. 19N] per-piece sum of gdT.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool smoothed_l1(double x, double mu, double &f, double &df) {  // BEO:316-340
  if (x < 0.0) return false;
  else if (x > mu) { f = x - 0.5 * mu; df = 1.0; return true; }
  else {
    const double xdmu = x / mu;
    const double sqrxdmu = xdmu * xdmu;
    const double mumxd2 = mu - 0.5 * x;
    f = mumxd2 * sqrxdmu * xdmu;
    df = sqrxdmu * ((-0.5) * xdmu + 3.0 * mumxd2 / mu);
    return true;
  }
}

__global__ void __launch_bounds__(kBlock)
k_assemble(const TrajDev *__restrict__ trg, const double *__restrict__ px_,
           const double *__restrict__ py_, int P, const double *__restrict__ res_sdf,
           const double *__restrict__ res_t, const double *__restrict__ res_gx,
           const double *__restrict__ res_gy, double safety_hor, double weight_p,
           double *__restrict__ block_partials, int *__restrict__ nonfinite) {
  __shared__ double lds[kTrajLdsDoubles];
  __shared__ double acc[19 * kMaxPieces + 1];
  const TrajL tr = stage_traj(trg, lds);
  const int N = tr.N;
  const int plen = 19 * N + 1;
  for (int e = threadIdx.x; e < plen; e += blockDim.x) acc[e] = 0.0;
  __syncthreads();
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < P) {
    const double px = px_[idx], py = py_[idx];
    const double sdf_value = res_sdf[idx];
    const double time_star = res_t[idx];
    double gr0 = res_gx[idx], gr1 = res_gy[idx];
    if (!(sdf_value == sdf_value) || !(time_star == time_star) || !(gr0 == gr0) || !(gr1 == gr1))
      atomicAdd(nonfinite, 1);
    double sdf_cost = -1.0, sdf_out_grad = 0.0;
    smoothed_l1(safety_hor - sdf_value, 0.01, sdf_cost, sdf_out_grad);
    if (sdf_cost > 0) {
      const int i = locate_piece(tr, time_star, 0);
      const double *c = tr.c + i * 18;
      const double s1 = time_star - tr.S[i];
      const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
      const double beta0[6] = {1.0, s1, s2, s3, s4, s5};
      const double beta1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
      double pos[3], vel[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        double p = 0.0, v = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) { p += c[k * 3 + d] * beta0[k]; v += c[k * 3 + d] * beta1[k]; }
        pos[d] = p; vel[d] = v;
      }
      const double yaw = pos[2];
      double sy, cy;
      sincos(yaw, &sy, &cy);
      if (sdf_value < 0) {  // BEO:832
        const double gx = cy * gr0 + sy * gr1;
        const double gy = (-sy) * gr0 + cy * gr1;
        gr0 = gx; gr1 = gy;
      }
      // grad_cost_p_sw (BEO:1031-1066) with St = I
      const double mrx = (-cy) * gr0 + (sy) * gr1;
      const double mry = (-sy) * gr0 + (-cy) * gr1;
      const double sgx = -sdf_out_grad * mrx, sgy = -sdf_out_grad * mry;
      const double dx = px - pos[0], dy = py - pos[1];
      const double v0 = (-sy) * dx + (cy) * dy;
      const double v1 = (-cy) * dx + (-sy) * dy;
      const double grad_yaw = (-sdf_out_grad * gr0) * v0 + (-sdf_out_grad * gr1) * v1;
      const double gPx = weight_p * sgx, gPy = weight_p * sgy, gYaw = weight_p * grad_yaw;
      const double pena = weight_p * sdf_cost;
      const double gdT = -((gPx * vel[0] + gPy * vel[1]) + gYaw * vel[2]);
      atomicAdd(&acc[0], pena);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        atomicAdd(&acc[1 + 0 * 6 * N + 6 * i + k], beta0[k] * gPx);
        atomicAdd(&acc[1 + 1 * 6 * N + 6 * i + k], beta0[k] * gPy);
        atomicAdd(&acc[1 + 2 * 6 * N + 6 * i + k], beta0[k] * gYaw);
      }
      atomicAdd(&acc[1 + 18 * N + i], gdT);
    }
  }
  __syncthreads();
  double *out = block_partials + (size_t)blockIdx.x * plen;
  for (int e = threadIdx.x; e < plen; e += blockDim.x) out[e] = acc[e];
}

// Fixed-order sum over block partials; gradT[j] = sum_{i > j} hist[i] (BEO:859-862).
__global__ void k_final(const double *__restrict__ block_partials, int nblocks, int N,
                        double *__restrict__ partial /* 19N+1: cost, gradC, gradT */) {
  __shared__ double hist[kMaxPieces];
  const int plen = 19 * N + 1;
  for (int e = threadIdx.x; e < plen; e += blockDim.x) {
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += block_partials[(size_t)b * plen + e];
    if (e <= 18 * N) partial[e] = s;
    else hist[e - (18 * N + 1)] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double suf = 0.0;
    for (int j = N - 1; j >= 0; --j) { partial[1 + 18 * N + j] = suf; suf += hist[j]; }
  }
}

}  // namespace svsdf
