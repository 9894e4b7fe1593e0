// svsdf_kernels.hpp -- gfx950 device kernels of the SVSDF cost/gradient pipeline.
//
// Behavioural spec (what, not how): reference
//   SWM = src/swept_volume/include/swept_volume/sw_manager.hpp
//   BEO = src/planner_algorithm/include/planner_algorithm/back_end_optimizer.hpp
//   TRJ = src/utils/include/utils/trajectory.hpp
// Pipeline per evaluation (all FP64; see DESIGN.md "Kernels"):
//   k_prep      trajectory -> per-piece monomials, cumulative start times, layer-1 pose table,
//               chunk bounding circles for the exact scan pruning
//   k_seed      choiceTInit layer 1 (SWM:538-581) over the shared pose table
//   k_refine    choiceTInit layers 2-4 + gradientDescent (SWM:1249-1325)
//   k_classify  exterior: FD gradient; interior: GSIP state + first circle samples
//   k_gsip      per GSIP round: max over samples, radius update, termination / next samples
//   k_assemble  per-point cost/gradient contribution (BEO:786-865) + block-level reduction
//   k_final     fixed-order sum of block partials, suffix sum for the duration gradient
// The query points are split into batches that run the whole chain on their own HIP stream
// without any host round trip: every count a later kernel needs (interior points, still-active
// GSIP points) lives in device memory (BatchCtl) and the solve kernels are persistent
// (wave-granular dynamic work fetch), so the batches' phases overlap and fill each other's tails.
#pragma once
#include <hip/hip_runtime.h>

#include "svsdf_shapes.hpp"

namespace svsdf {

constexpr int kMaxPieces = 64;
constexpr int kMaxSlots = 24;   // GSIP samples per round: 2, 6, 18, 21, 21, ... (SWM:60-71,105-110)
constexpr int kMaxRounds = 9;   // SWM:995 (iter > 8)
constexpr int kBlock = 256;
constexpr int kMaxBatches = 8;
// GSIP iterations: a round normally takes one iteration, plus one supplementary iteration when
// the upper-bound selection (k_select / k_gsip) has to solve more samples of the same round.
constexpr int kMaxIter = 24;
constexpr int kWorkCounters = 2 * (kMaxIter + 1);  // one per (seed|refine) launch of a batch
constexpr double kUnsolved = -1e300;               // sq_sdf marker: sample not (yet) solved

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

// Layer-1 pruning: bounding circle of the robot-origin positions of kChunk consecutive scan
// samples, inflated by the shape's bound radius R (sdf_shape(q) >= |q| - R for every q), so
// that  sdf(sample) >= |p - c| - rb  for every sample of the chunk.
constexpr int kChunk = 8;
struct Chunk { double cx, cy, rb, pad; };

// Per-batch control block (device memory; the counters are cleared by k_prep at the start of
// every evaluation, start/count are written by the host once per point upload).
struct BatchCtl {
  int start, count;               // points [start, start + count) of the sorted shard
  int n_active[kMaxIter + 2];     // [i] = GSIP points entering iteration i; [0] = interior points
  int n_solve[kMaxIter + 2];      // [i] = sample solves requested for iteration i
  int n_seed[kMaxIter + 2];       // [i] = new samples to seed in iteration i
  unsigned work[kWorkCounters];   // dynamic work-fetch cursors, one per solve launch
  int nonfinite;
  int pad;
  unsigned long long stat_solves, stat_evals, stat_scan;
};

// LDS view of the trajectory
struct TrajL {
  const double *T, *S, *c;
  int N;
  double dur;
};

__host__ __device__ __forceinline__ int traj_lds_doubles(int N) { return 20 * N + 1; }

// lds must hold 20N+1 doubles: T[N] | S[N+1] | c[18N]
__device__ __forceinline__ TrajL stage_traj(const TrajDev *__restrict__ g, double *lds) {
  const int N = g->N;
  double *T = lds, *S = lds + N, *c = lds + 2 * N + 1;
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
// k_prep: one block.  in = [coeffs (6N x 3 column-major) | T (N) | tk (K)] as uploaded.
// Also clears the per-batch control blocks for this evaluation.
// ---------------------------------------------------------------------------------------------
__global__ void k_prep(const double *__restrict__ in, int N, double dur, int K,
                       TrajDev *__restrict__ tr, Pose *__restrict__ pose,
                       Chunk *__restrict__ chunks, double r_bound, BatchCtl *__restrict__ ctl,
                       int nbatch) {
  extern __shared__ double prep_lds[];
  const double *coeffs = in, *T = in + 18 * N, *tk = in + 19 * N;
  for (int b = threadIdx.x; b < nbatch; b += blockDim.x) {
    BatchCtl &c = ctl[b];
    for (int r = 0; r < kMaxIter + 2; ++r) { c.n_active[r] = 0; c.n_solve[r] = 0; c.n_seed[r] = 0; }
    for (int r = 0; r < kWorkCounters; ++r) c.work[r] = 0u;
    c.nonfinite = 0;
    c.stat_solves = 0ull; c.stat_evals = 0ull; c.stat_scan = 0ull;
  }
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
  const TrajL tl = stage_traj(tr, prep_lds);
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    int piece = 0;
    pose[k] = pose_at(tl, tk[k], piece);
  }
  __threadfence_block();
  __syncthreads();
  const int nch = (K + kChunk - 1) / kChunk;
  for (int c = threadIdx.x; c < nch; c += blockDim.x) {
    const int k0 = c * kChunk, k1 = (k0 + kChunk < K) ? k0 + kChunk : K;
    double xmin = pose[k0].x, xmax = xmin, ymin = pose[k0].y, ymax = ymin;
    for (int k = k0 + 1; k < k1; ++k) {
      xmin = fmin(xmin, pose[k].x); xmax = fmax(xmax, pose[k].x);
      ymin = fmin(ymin, pose[k].y); ymax = fmax(ymax, pose[k].y);
    }
    const double cx = 0.5 * (xmin + xmax), cy = 0.5 * (ymin + ymax);
    double r = 0.0;
    for (int k = k0; k < k1; ++k) r = fmax(r, norm2(pose[k].x - cx, pose[k].y - cy));
    Chunk ch;
    ch.cx = cx; ch.cy = cy; ch.rb = r * (1.0 + 1e-12) + r_bound + 1e-9; ch.pad = 0.0;
    chunks[c] = ch;
  }
}

// Shape bound radius: max over a polar grid of |q| - sdf_shape(q) (body frame, including the
// shape's own offset/rotation).  For an exact SDF this is the circumradius about the origin.
template <int SHAPE>
__global__ void k_rbound(ShapeParams sp, double rmax, int nrad, int nang, double *__restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  double v = -1e300;
  if (idx < nrad * nang) {
    const int ir = idx / nang, ia = idx % nang;
    const double r = rmax * (double)(ir + 1) / (double)nrad;
    const double a = 2.0 * kPI * (double)ia / (double)nang;
    const double x = r * cos(a), y = r * sin(a);
    v = r - shape_sdf<SHAPE>(sp, x, y);
  }
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
  if ((threadIdx.x & 63) == 0) {
    // atomic max on a non-negative double via its integer ordering
    if (v > 0.0) atomicMax((unsigned long long *)out, (unsigned long long)__double_as_longlong(v));
  }
}

// ---------------------------------------------------------------------------------------------
// Query sets.  A solve launch works on either the main points of a batch or the GSIP circle
// samples of the batch's still-active interior points:
//   query q in [0, n * n_outer):  e = q % n, j = q / n, a = list ? list[e] : e,
//   slot = j * stride + base + a.
// n comes from device memory (count_ptr) when the host cannot know it.
// ---------------------------------------------------------------------------------------------
struct QuerySet {
  const double *qx, *qy;
  size_t stride;
  const int *count_ptr;   // device count (may be null -> count_fixed)
  int count_fixed;
  const int *list;        // compacted active list (may be null)
  int base;
  int n_outer;
  const int *slots;       // explicit slot list (n = *count_ptr entries, overrides the rest) or null
  const int *skip;        // per-entry skip flags indexed by base + a (may be null)
};

// GSIP circle samples are materialised by the seed kernel itself (one lane group per sample):
// sample j of interior point ia sits at centre + r (cos th_j, sin th_j), th_j = th0 + j*th_res
// by repeated addition (SampleSet2D::getElements / getElementPos, SWM:36-39, 60-71).
struct SampleGen {
  const double *cx, *cy;      // main points
  const int *pt;              // interior -> main point index
  const double *r, *theta0, *theta_res;
  double *sqx, *sqy, *sqth, *sq_sdf;
  size_t stride;
};

__device__ __forceinline__ long long qs_total(const QuerySet &qs, int &n) {
  n = qs.count_ptr ? *qs.count_ptr : qs.count_fixed;
  return qs.slots ? (long long)n : (long long)n * qs.n_outer;
}
// returns false when the query is to be skipped
__device__ __forceinline__ bool qs_slot(const QuerySet &qs, int n, long long q, size_t &slot) {
  if (qs.slots) { slot = (size_t)qs.slots[q]; return true; }
  const int e = (int)(q % n), j = (int)(q / n);
  const int a = qs.list ? qs.list[e] : e;
  slot = (size_t)j * qs.stride + (size_t)qs.base + (size_t)a;
  return !(qs.skip && qs.skip[qs.base + a]);
}

// G lanes cooperate on one query (64/G queries per wave).
template <int G>
struct Grp {
  static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "group size");
  __device__ static __forceinline__ int li() { return (int)(threadIdx.x & (G - 1)); }
  __device__ static __forceinline__ double bcast(double v, int src) {
    if constexpr (G == 1) return v; else return __shfl(v, src, G);
  }
  // lexicographic min of (d, k) over the group; all lanes get the result
  __device__ static __forceinline__ void min_dk(double &d, int &k) {
    if constexpr (G > 1) {
#pragma unroll
      for (int m = G / 2; m >= 1; m >>= 1) {
        const double od = __shfl_xor(d, m, G);
        const int ok = __shfl_xor(k, m, G);
        if (od < d || (od == d && ok < k)) { d = od; k = ok; }
      }
    }
  }
  // bit i set <=> lane i of this group has pred
  __device__ static __forceinline__ unsigned ballot(bool pred) {
    if constexpr (G == 1) return pred ? 1u : 0u;
    else {
      const unsigned long long m = __ballot(pred);
      const int base = (int)(threadIdx.x & 63) & ~(G - 1);
      return (unsigned)((m >> base) & ((1ull << G) - 1ull));
    }
  }
};

// wave-granular dynamic work fetch: every wave takes the next 64/G queries; returns the
// wave's first query in `wave_base` (wave-uniform) and this lane's query as the result
template <int G, int M = 1>
__device__ __forceinline__ long long fetch_work(unsigned *cursor, long long &wave_base) {
  unsigned base = 0;
  if ((threadIdx.x & 63) == 0) base = atomicAdd(cursor, (unsigned)(M * 64 / G));
  base = __builtin_amdgcn_readfirstlane(base);
  wave_base = (long long)base;
  return (long long)base + (long long)((threadIdx.x & 63) / G);
}

// ---------------------------------------------------------------------------------------------
// k_seed: choiceTInit layer 1 (SWM:549-576, first pass of the while loop) over the shared pose
// table.  Poses at the scan times do not depend on the query point, so they are computed once
// (k_prep) and every query only does the rigid transform + shape SDF per sample.  Chunks of 8
// samples whose lower bound exceeds the running minimum are skipped: exact, because a skipped
// sample can never be (or tie with) the minimum the reference's strict `<` scan keeps.
// ---------------------------------------------------------------------------------------------
template <int SHAPE, int G>
__global__ void __launch_bounds__(kBlock)
k_seed(const TrajDev *__restrict__ trg, const double *__restrict__ tk, const Pose *__restrict__ pose_g,
       const Chunk *__restrict__ chunks_g, ShapeParams sp, QuerySet qs, SampleGen sg,
       double *__restrict__ seed_t, double *__restrict__ seed_min, int prune,
       BatchCtl *__restrict__ ctl, int work_idx) {
  extern __shared__ double seed_lds[];
  int n;
  const long long total = qs_total(qs, n);
  if (total <= 0 || (long long)blockIdx.x * (blockDim.x / G) >= total) return;
  const int K = trg->K;
  const int nch = (K + kChunk - 1) / kChunk;
  Pose *pose = reinterpret_cast<Pose *>(seed_lds);
  Chunk *chunks = reinterpret_cast<Chunk *>(seed_lds + 4 * (size_t)K);
  {
    const double *src = reinterpret_cast<const double *>(pose_g);
    for (int i = threadIdx.x; i < 4 * K; i += blockDim.x) seed_lds[i] = src[i];
    const double *srcc = reinterpret_cast<const double *>(chunks_g);
    for (int i = threadIdx.x; i < 4 * nch; i += blockDim.x) seed_lds[4 * (size_t)K + i] = srcc[i];
  }
  __syncthreads();
  const int li = Grp<G>::li();
  unsigned n_scan = 0;
  long long gq_first = 0;
  // the scan of one query is short: take up to 8 wave-loads per atomic, fewer when the launch
  // has little work per resident wave (keeps every wave busy instead of a few waves for long)
  const long long per_wave = total / ((long long)gridDim.x * (blockDim.x / 64) * (64 / G));
  const int nfetch = (int)(per_wave < 1 ? 1 : (per_wave > 8 ? 8 : per_wave));
  for (int guard = 0; guard < (1 << 24); ++guard) {
    long long wave_base = 0;
    {
      unsigned base = 0;
      if ((threadIdx.x & 63) == 0) base = atomicAdd(&ctl->work[work_idx], (unsigned)(nfetch * 64 / G));
      base = __builtin_amdgcn_readfirstlane(base);
      wave_base = (long long)base;
      gq_first = wave_base + (long long)((threadIdx.x & 63) / G);
    }
    if (wave_base >= total) break;
    for (int sub = 0; sub < nfetch; ++sub) {
    const long long gq = gq_first + (long long)sub * (64 / G);
    double px = 0.0, py = 0.0;
    size_t slot = 0;
    bool live = gq < total;
    if (live) live = qs_slot(qs, n, gq, slot);
    if (live) {
      if (sg.sqx) {  // materialise the circle sample (slot = j * stride + ia)
        const int j = (int)(slot / sg.stride);
        const size_t ia = slot - (size_t)j * sg.stride;
        double theta = sg.theta0[ia];
        const double theta_res = sg.theta_res[ia];
        for (int q = 0; q < j; ++q) theta += theta_res;
        const double r = sg.r[ia];
        const int i = sg.pt[ia];
        px = sg.cx[i] + 1.0 * r * cos(theta);
        py = sg.cy[i] + 1.0 * r * sin(theta);
        if (li == 0) { sg.sqx[slot] = px; sg.sqy[slot] = py; sg.sqth[slot] = theta; sg.sq_sdf[slot] = kUnsolved; }
      } else {
        px = qs.qx[slot]; py = qs.qy[slot];
        live = (px == px);  // NaN marks an unused slot (whole group)
      }
    }
    if (live) {
    double best_d = 1e9;   // min_dis initial value (SWM:545)
    int best_k = 0x7fffffff;

    auto eval_chunk = [&](int c) {
      double d_loc = 1e300;
      int k_loc = 0x7fffffff;
      constexpr int GS = (G < kChunk) ? G : kChunk;
#pragma unroll
      for (int m = 0; m < kChunk / GS; ++m) {
        const int k = c * kChunk + li + GS * m;
        if (li < kChunk && k < K) {
          const Pose p = pose[k];
          const double d = sdf_from_pose<SHAPE>(sp, p, px, py);
          ++n_scan;
          if (d < d_loc) { d_loc = d; k_loc = k; }  // k increases with m: earliest kept on ties
        }
      }
      Grp<G>::min_dk(d_loc, k_loc);
      if (d_loc < best_d || (d_loc == best_d && k_loc < best_k)) { best_d = d_loc; best_k = k_loc; }
    };

    if (!prune) {
      for (int c = 0; c < nch; ++c) eval_chunk(c);
    } else {
      // 1. the chunk with the smallest lower bound gives the first upper bound
      double lb_loc = 1e300;
      int c_loc = 0;
      for (int c = li; c < nch; c += G) {
        const Chunk ch = chunks[c];
        const double lb = norm2(px - ch.cx, py - ch.cy) - ch.rb;
        if (lb < lb_loc) { lb_loc = lb; c_loc = c; }
      }
      Grp<G>::min_dk(lb_loc, c_loc);
      const int c0 = c_loc;
      eval_chunk(c0);
      // 2. every other chunk whose lower bound does not exceed the running minimum
      int c = 0;
      while (c < nch) {
        const int cc = c + li;
        bool need = false;
        if (cc < nch && cc != c0) {
          const Chunk ch = chunks[cc];
          const double lb = norm2(px - ch.cx, py - ch.cy) - ch.rb;
          need = !(lb > best_d);
        }
        const unsigned m = Grp<G>::ballot(need);
        if (m == 0u) { c += G; continue; }
        const int first = __ffs(m) - 1;
        eval_chunk(c + first);
        c = c + first + 1;
      }
    }
    if (li == 0) {
      seed_t[slot] = tk[best_k];
      seed_min[slot] = best_d;
    }
    }  // live
    }  // sub
  }
  unsigned long long tot = n_scan;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) tot += __shfl_xor(tot, m, 64);
  if ((threadIdx.x & 63) == 0 && tot) { atomicAdd(&ctl->stat_scan, tot); atomicAdd(&ctl->stat_evals, tot); }
}

// ---------------------------------------------------------------------------------------------
// k_refine: choiceTInit layers 2-4 (SWM:557-577) + gradientDescent (SWM:1249-1325) from the
// layer-1 seed.  The 21 samples of a layer and the <= 29 candidates of a halving ladder are
// evaluated G at a time: a ladder's candidates do not depend on each other, so the first
// accepted one is exactly the one the sequential reference loop accepts (bit-identical result,
// shorter dependent chain).  getSDF_DOT (SWM:799-806) is evaluated once per descent pass: x does
// not change inside the ladder, so the reference's per-trial re-evaluation returns the same number.
// ---------------------------------------------------------------------------------------------
template <int SHAPE, int G>
__global__ void __launch_bounds__(kBlock)
k_refine(const TrajDev *__restrict__ trg, ShapeParams sp, QuerySet qs,
         const double *__restrict__ seed_t, const double *__restrict__ seed_min,
         double *__restrict__ out_sdf, double *__restrict__ out_t, BatchCtl *__restrict__ ctl,
         int work_idx) {
  extern __shared__ double refine_lds[];
  int n;
  const long long total = qs_total(qs, n);
  if (total <= 0 || (long long)blockIdx.x * (blockDim.x / G) >= total) return;
  const TrajL tr = stage_traj(trg, refine_lds);
  const int li = Grp<G>::li();
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  unsigned n_eval = 0, n_solved = 0;
  for (int guard = 0; guard < (1 << 26); ++guard) {
    long long wave_base;
    const long long gq = fetch_work<G>(&ctl->work[work_idx], wave_base);
    if (wave_base >= total) break;
    double px = 0.0, py = 0.0;
    size_t slot = 0;
    bool live = gq < total;
    if (live) live = qs_slot(qs, n, gq, slot);
    if (live) {
      px = qs.qx[slot]; py = qs.qy[slot];
      live = (px == px);
    }
    if (live) {
    int piece = 0;
    double time_seed = seed_t[slot];
    double min_dis = seed_min[slot];

    // ---- choiceTInit layers 2-4
    double dt = 0.15;
    dt *= 0.1;
    for (int layer = 2; layer <= 4; ++layer) {
      const double t0 = dmax(0.0, time_seed - 10 * dt);
      const double loop_terminal = dmin(tr.dur, time_seed + 10 * dt);
      double t = t0;  // lane li starts at the li-th accumulated sample
      for (int i = 0; i < li; ++i) t += dt;
      int kbase = 0;
      while (Grp<G>::bcast(t, 0) <= loop_terminal) {
        const bool valid = t <= loop_terminal;
        double d = inf;
        if (valid) { d = sdf_at<SHAPE>(tr, sp, px, py, t, piece); ++n_eval; }
        int k = kbase + li;
        Grp<G>::min_dk(d, k);
        const double tb = Grp<G>::bcast(t, k - kbase);
        if (d < min_dis) { time_seed = tb; min_dis = d; }
#pragma unroll
        for (int i = 0; i < G; ++i) t += dt;
        kbase += G;
      }
      dt *= 0.1;
    }

    // ---- gradientDescent
    const double t_min = dmax(0.0, time_seed - 3.4);
    const double t_max = dmin(time_seed + 3.4, tr.dur);
    double x = time_seed, fx = 0.0, prev_x = 10000000.0;
    int iter = 0;
    bool stop = false;
    while (iter < 1000 && !stop && fabs(x - prev_x) > 1e-16) {
      // tasks: 0 -> sdf(t1), 1 -> sdf(t2), 2 -> sdf(x) (first pass only)
      const int ntask = (iter == 0) ? 3 : 2;
      const double t1 = dmax(0.0, x - 0.000001);
      const double t2 = dmin(tr.dur, x + 0.000001);
      double sdf1 = 0.0, sdf2 = 0.0, f0 = 0.0;
#pragma unroll
      for (int base = 0; base < 3; base += G) {
        if (base < ntask) {
          const int task = base + li;
          double d = 0.0;
          if (task < ntask) {
            const double tt = (task == 0) ? t1 : (task == 1) ? t2 : x;
            d = sdf_at<SHAPE>(tr, sp, px, py, tt, piece);
            ++n_eval;
          }
          if (0 >= base && 0 < base + G) sdf1 = Grp<G>::bcast(d, 0 - base);
          if (1 >= base && 1 < base + G) sdf2 = Grp<G>::bcast(d, 1 - base);
          if (2 >= base && 2 < base + G) f0 = Grp<G>::bcast(d, 2 - base);
        }
      }
      if (iter == 0) fx = f0;
      const double g = (sdf2 - sdf1) * 500000;
      const double sgn = (double)((int)(g > 0) - (int)(g < 0));
      prev_x = x;
      bool accepted = false;
      for (int j0 = 1; j0 <= 29 && !accepted; j0 += G) {
        const int j = j0 + li;  // div
        const bool valid = j <= 29;
        const double tau = ldexp(0.01, 1 - j);  // alpha halved (div - 1) times: exact
        const double change = -tau * sgn;
        double xc = x + change;
        xc = dmax(dmin(xc, t_max), t_min);
        double fc = inf;
        if (valid) { fc = sdf_at<SHAPE>(tr, sp, px, py, xc, piece); ++n_eval; }
        const unsigned m = Grp<G>::ballot(valid && ((fc - fx) < 0));
        if (m != 0u) {
          const int first = __ffs(m) - 1;
          x = Grp<G>::bcast(xc, first);
          fx = Grp<G>::bcast(fc, first);
          iter += first + 1;
          accepted = true;
        } else {
          const int left = 29 - j0 + 1;
          iter += (left < G) ? left : G;
        }
      }
      if (!accepted) stop = true;
    }
    if (li == 0) {
      out_sdf[slot] = fx;
      out_t[slot] = x;
      ++n_solved;
    }
    }  // live
  }
  unsigned long long te = n_eval, ts = n_solved;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { te += __shfl_xor(te, m, 64); ts += __shfl_xor(ts, m, 64); }
  if ((threadIdx.x & 63) == 0 && te) { atomicAdd(&ctl->stat_evals, te); atomicAdd(&ctl->stat_solves, ts); }
}

// ---------------------------------------------------------------------------------------------
// GSIP state (getTrueSDFofSweptVolume SWM:916-1018), one entry per interior point; every array
// is indexed by (batch start + interior index within the batch).
// ---------------------------------------------------------------------------------------------
struct GsipState {
  int *pt;          // index of the (sorted) main point
  double *r;        // current circle radius
  double *theta0;   // first sample angle of the current round
  double *theta_res;
  int *iter;        // 1..9
  int *nsamp;       // samples emitted for the current round
  int *supp;        // 1: the point waits for supplementary sample solves of its current round
  int *list[2];     // ping-pong compacted lists of still-active interior indices
  int *solve[2];    // ping-pong lists of sample slots to solve (capacity kMaxSlots per point)
  int *seedl[2];    // ping-pong lists of freshly emitted sample slots to seed
  // sub-query slots [j * stride + batch start + a]
  double *sqx, *sqy, *sqth, *sq_sdf, *sq_t;
};

// SampleSet2D::getElements + getElementPos (SWM:36-39, 60-71) for the single ring rk = 1.0.
// Number of samples of a round (the theta loop of SampleSet2D::getElements, SWM:60-71) and their
// slots pushed to the seed list; the seed kernel materialises the positions.
__device__ __forceinline__ int emit_samples(size_t ia, size_t stride, double theta0, double theta_res,
                                            int *__restrict__ seed_list, int *__restrict__ seed_count) {
  int n = 0;
  for (double theta = theta0; theta < theta0 + 2 * kPI; theta += theta_res) ++n;
  n = n < kMaxSlots ? n : kMaxSlots;
  int pos = atomicAdd(seed_count, n);
  for (int j = 0; j < n; ++j) seed_list[pos++] = (int)((size_t)j * stride + ia);
  return n;
}

// Per main point after the first solve: exterior -> FD gradient (getGradPrelAtTimeStamp,
// SWM:779-788) and done; interior -> GSIP init (SWM:926-963).
template <int SHAPE>
__global__ void __launch_bounds__(kBlock)
k_classify(const TrajDev *__restrict__ trg, ShapeParams sp, const double *__restrict__ px_,
           const double *__restrict__ py_, const double *__restrict__ sdf_,
           const double *__restrict__ t_, double *__restrict__ res_sdf,
           double *__restrict__ res_t, double *__restrict__ res_gx, double *__restrict__ res_gy,
           GsipState gs, size_t stride, BatchCtl *__restrict__ ctl) {
  extern __shared__ double classify_lds[];
  const TrajL tr = stage_traj(trg, classify_lds);
  const int start = ctl->start, count = ctl->count;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < count; e += gridDim.x * blockDim.x) {
    const int i = start + e;
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
      continue;
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
    const int a = atomicAdd(&ctl->n_active[0], 1);
    const size_t ia = (size_t)start + a;
    // SampleSet2D::initSet (SWM:73-103)
    double theta0 = atan2(vx, -vy);
    if (theta0 < 0) theta0 += 2 * kPI;
    const double theta_res = kPI + 0.1;
    const double r0 = 10;
    gs.pt[ia] = i;
    gs.r[ia] = r0;
    gs.theta0[ia] = theta0;
    gs.theta_res[ia] = theta_res;
    gs.iter[ia] = 1;
    gs.supp[ia] = 0;
    gs.list[0][ia] = a;
    gs.nsamp[ia] = emit_samples(ia, stride, theta0, theta_res, gs.seedl[0] + (size_t)start * kMaxSlots,
                                &ctl->n_seed[0]);
    res_t[i] = ts;  // real_t_star fallback
  }
}

// ---------------------------------------------------------------------------------------------
// Upper-bound selection of the GSIP samples (exact).  A round only uses the arg-max sample:
// max_g, its t* and its angle (SWM:975-990).  Every sample's solved value is bounded above by
// its layer-1 seed value U_j: layers 2-4 only lower min_dis, the descent starts at
// f(time_seed) == min_dis and only accepts strict decreases.  So after solving a subset whose
// best value is g*, any sample with U_j < g* can neither be nor tie with the maximum and need not
// be solved.  k_select requests the samples within `delta` of the best bound; k_gsip requests the
// rest of {U_j >= g*} (one supplementary iteration, rarely non-empty) before it closes the round.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_select(GsipState gs, const double *__restrict__ seed_min, size_t stride, int it, double delta,
         BatchCtl *__restrict__ ctl) {
  const int n_act = ctl->n_active[it];
  const int start = ctl->start;
  const int *cur = gs.list[it & 1] + start;
  int *out = gs.solve[it & 1] + (size_t)start * kMaxSlots;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_act; e += gridDim.x * blockDim.x) {
    const int a = cur[e];
    const size_t ia = (size_t)start + a;
    if (gs.supp[ia]) continue;  // its supplementary solves were requested by k_gsip
    const int n = gs.nsamp[ia];
    double umax = -1e300;
    for (int j = 0; j < n; ++j) umax = fmax(umax, seed_min[(size_t)j * stride + ia]);
    const double thr = umax - delta;
    int cnt = 0;
    for (int j = 0; j < n; ++j) cnt += (seed_min[(size_t)j * stride + ia] >= thr) ? 1 : 0;
    int pos = atomicAdd(&ctl->n_solve[it], cnt);
    for (int j = 0; j < n; ++j) {
      const size_t sl = (size_t)j * stride + ia;
      if (seed_min[sl] >= thr) out[pos++] = (int)sl;
    }
  }
}

// One GSIP iteration per still-active interior point: close the round (SWM:965-1009, final
// assembly SWM:1011-1017) or request supplementary solves.  Points that continue are appended to
// the next iteration's compacted list.
__global__ void __launch_bounds__(kBlock)
k_gsip(const double *__restrict__ px_, const double *__restrict__ py_, GsipState gs,
       const double *__restrict__ seed_min, size_t stride, int it, double *__restrict__ res_sdf,
       double *__restrict__ res_t, double *__restrict__ res_gx, double *__restrict__ res_gy,
       BatchCtl *__restrict__ ctl) {
  const int n_act = ctl->n_active[it];
  const int start = ctl->start;
  const int *cur = gs.list[it & 1] + start;
  int *nxt = gs.list[(it + 1) & 1] + start;
  int *solve_nxt = gs.solve[(it + 1) & 1] + (size_t)start * kMaxSlots;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_act; e += gridDim.x * blockDim.x) {
    const int a = cur[e];
    const size_t ia = (size_t)start + a;
    const int i = gs.pt[ia];
    const double cx = px_[i], cy = py_[i];
    double max_g = -100000;
    double real_t = res_t[i], star_th = 0.0;
    const int n = gs.nsamp[ia];
    for (int j = 0; j < n; ++j) {  // solved samples only; unsolved ones carry kUnsolved
      const size_t s = (size_t)j * stride + ia;
      const double cur_g = gs.sq_sdf[s];
      if (cur_g > max_g) { max_g = cur_g; real_t = gs.sq_t[s]; star_th = gs.sqth[s]; }
    }
    // unsolved samples that could still reach max_g
    int more = 0;
    for (int j = 0; j < n; ++j) {
      const size_t s = (size_t)j * stride + ia;
      more += (gs.sq_sdf[s] == kUnsolved && seed_min[s] >= max_g) ? 1 : 0;
    }
    if (more > 0) {
      int pos = atomicAdd(&ctl->n_solve[it + 1], more);
      for (int j = 0; j < n; ++j) {
        const size_t s = (size_t)j * stride + ia;
        if (gs.sq_sdf[s] == kUnsolved && seed_min[s] >= max_g) solve_nxt[pos++] = (int)s;
      }
      gs.supp[ia] = 1;
      nxt[atomicAdd(&ctl->n_active[it + 1], 1)] = a;
      continue;
    }
    gs.supp[ia] = 0;
    const double r_star = gs.r[ia] - max_g;
    const int iter = gs.iter[ia];
    if (iter > 8 || fabs(max_g) < 0.1) {
      const double corx = cx + 1.0 * r_star * cos(star_th);
      const double cory = cy + 1.0 * r_star * sin(star_th);
      double gx = corx - cx, gy = cory - cy;
      const double z = gx * gx + gy * gy;
      if (z > 0.0) { const double nn = sqrt(z); gx = gx / nn; gy = gy / nn; }
      res_sdf[i] = -r_star; res_t[i] = real_t; res_gx[i] = gx; res_gy[i] = gy;
      continue;
    }
    // expandSet(2, theta*) (SWM:105-110)
    double theta_res = gs.theta_res[ia] / (2 + 1);
    theta_res = dmax(0.3, theta_res);
    gs.r[ia] = r_star;
    gs.theta_res[ia] = theta_res;
    gs.theta0[ia] = star_th;
    gs.iter[ia] = iter + 1;
    res_t[i] = real_t;
    gs.nsamp[ia] = emit_samples(ia, stride, star_th, theta_res, gs.seedl[(it + 1) & 1] + (size_t)start * kMaxSlots,
                                &ctl->n_seed[it + 1]);
    nxt[atomicAdd(&ctl->n_active[it + 1], 1)] = a;
  }
}

// ---------------------------------------------------------------------------------------------
// k_assemble: loop body of BEO:786-865 for one point given (sdf, t*, grad_prel), then a
// block-level segmented reduction keyed by piece.  Block partials are stored entry-major
// ([entry][block]) so that k_final reads them coalesced.  Entries:
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
  extern __shared__ double asm_lds[];
  const TrajL tr = stage_traj(trg, asm_lds);
  const int N = tr.N;
  const int plen = 19 * N + 1;
  double *acc = asm_lds + traj_lds_doubles(N);
  for (int e = threadIdx.x; e < plen; e += blockDim.x) acc[e] = 0.0;
  __syncthreads();
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < P; idx += gridDim.x * blockDim.x) {
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
  for (int e = threadIdx.x; e < plen; e += blockDim.x)
    block_partials[(size_t)e * gridDim.x + blockIdx.x] = acc[e];
}

// One wave per entry: fixed-order sum over the block partials.
__global__ void __launch_bounds__(64)
k_final(const double *__restrict__ block_partials, int nblocks, double *__restrict__ sums) {
  const int e = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) s += block_partials[(size_t)e * nblocks + b];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if (threadIdx.x == 0) sums[e] = s;
}

// partial = [cost, gradC (18N), gradT (N)] with gradT[j] = sum_{i > j} hist[i] (BEO:859-862);
// also gathers the per-batch counters.
__global__ void k_finish(const double *__restrict__ sums, int N, double *__restrict__ partial,
                         const BatchCtl *__restrict__ ctl, int nbatch, int it_end,
                         unsigned long long *__restrict__ stats_out) {
  for (int k = threadIdx.x; k <= 18 * N; k += blockDim.x) partial[k] = sums[k];
  if (threadIdx.x == 0) {
    double suf = 0.0;
    for (int j = N - 1; j >= 0; --j) { partial[1 + 18 * N + j] = suf; suf += sums[1 + 18 * N + j]; }
    unsigned long long so = 0, ev = 0, sc = 0, in = 0, nf = 0, rem = 0, seeded = 0, iters = 0;
    for (int b = 0; b < nbatch; ++b) {
      so += ctl[b].stat_solves; ev += ctl[b].stat_evals; sc += ctl[b].stat_scan;
      in += (unsigned long long)ctl[b].n_active[0]; nf += (unsigned long long)ctl[b].nonfinite;
      rem += (unsigned long long)ctl[b].n_active[it_end];   // > 0: more iterations are needed
      for (int i = 0; i <= it_end; ++i) {
        seeded += (unsigned long long)ctl[b].n_seed[i];
        if (ctl[b].n_active[i] > 0 && (unsigned long long)(i + 1) > iters) iters = (unsigned long long)(i + 1);
      }
    }
    stats_out[0] = so; stats_out[1] = ev; stats_out[2] = sc; stats_out[3] = in; stats_out[4] = nf;
    stats_out[5] = rem; stats_out[6] = seeded; stats_out[7] = iters;
  }
}

}  // namespace svsdf
