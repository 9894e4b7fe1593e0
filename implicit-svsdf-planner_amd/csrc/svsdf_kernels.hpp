// svsdf_kernels.hpp -- gfx950 device kernels of the SVSDF cost/gradient pipeline.
//
// Behavioural spec (what, not how): reference
//   SWM = src/swept_volume/include/swept_volume/sw_manager.hpp
//   BEO = src/planner_algorithm/include/planner_algorithm/back_end_optimizer.hpp
//   TRJ = src/utils/include/utils/trajectory.hpp
// Pipeline per evaluation (all FP64; see DESIGN.md "Kernels"):
//   k_prep      trajectory -> per-piece monomials, cumulative start times, layer-1 pose table,
//               chunk bounding circles for the exact scan pruning
//   k_solve     one argmin-over-t solve per query: choiceTInit (SWM:538-581, layer 1 over the
//               shared pose table with exact chunk pruning, layers 2-4) + gradientDescent
//               (SWM:1249-1325); G lanes cooperate on one query
//   k_classify  exterior: FD gradient; interior: GSIP state
//   k_round     32 lanes per interior point: close the previous GSIP round (max over the solved
//               samples, radius update, termination) and open the next one (circle samples, cheap
//               upper bounds, selection of the samples worth solving)
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

#ifndef SVSDF_SOLVE_WAVES
#define SVSDF_SOLVE_WAVES 1
#endif
#ifndef SVSDF_ELASTIC
#define SVSDF_ELASTIC 1   // the descent's halving ladders share the wave's lanes (descend_from_seed)
#endif
#ifndef SVSDF_FUSED_PASS
#define SVSDF_FUSED_PASS 1   // a wave's LAST open descent: derivative + both signs of the ladder in one step (descend_from_seed)
#endif
constexpr int kMaxPieces = 128;   // = SVSDF_MAX_PIECES (64 until round 5); sizes TrajDev and the result rows, nothing on the hot path
constexpr int kMaxSlots = 24;   // GSIP samples per round: 2, 6, 18, 21, 21, ... (SWM:60-71,105-110)
constexpr int kMaxRounds = 9;   // SWM:995 (iter > 8)
constexpr int kBlock = 256;
constexpr int kMaxBatches = 8;
constexpr int kMaxCand = 48;    // candidate chunks k_round keeps per (interior point, round); more: all chunks are walked
// GSIP iterations: a round normally takes one iteration, plus one supplementary iteration when
// the upper-bound selection (k_round) has to solve more samples of the same round.
constexpr int kMaxIter = 24;
constexpr int kWorkCounters = kMaxIter + 2;        // one per k_solve launch of a batch
constexpr double kUnsolved = -1e300;               // sq_sdf marker: sample not (yet) solved

// Trajectory as the device sees it (global memory; staged into LDS by each block).
struct TrajDev {
  int N;
  int K;        // layer-1 samples (t = 0; t <= dur; t += 0.15, SWM:567)
  int exact;    // piece-local time by the reference's successive subtractions: 0 no (cumulative form), 1 yes, 2 yes and
                // without the comparison-free prefix (some piece shorter than 1e-6 s)
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
struct Chunk { double cx, cy, rb, slack; };  // slack: continuous-path allowance V_c * h for the exact cull (host)
// Anchor of a chunk (round 6; k_prep writes the table behind the nch chunk records, same buffer): the pose `ka` of the chunk
// whose largest distance to the chunk's other poses is smallest.  A shape SDF is 1-Lipschitz, and for a pose k and the anchor
// a of one chunk the body-frame images of a world point q differ by  R_k^T (x_a - x_k) + (R_k - R_a)^T (q - x_a),  so
//   sdf_k(q) >= sdf_a(q) - (max_k |x_k - x_a| + max_k |R_k - R_a| |q - x_a|)        (|R_k - R_a| = 2 |sin((yaw_k - yaw_a) / 2)|)
// adx = the first maximum (with its rounding allowance), rot2 = the square of the second (likewise).
struct ChunkAnchor { double adx, rot2; int ka, pad_; double pad2_; };   // 32 bytes like Chunk
#ifndef SVSDF_ANCHOR_MIN
#define SVSDF_ANCHOR_MIN 3   // circle survivors from which a seed scan evaluates their anchors first (below: the chunks themselves)
#endif

constexpr int kStatSlots = 32;
#ifdef SVSDF_SITE_STATS
struct StatSlot { unsigned long long solves, evals, scan, culled, round_scan, spec, pad[26]; };   // diagnostic build: two lines
#else
struct StatSlot { unsigned long long solves, evals, scan, culled, round_scan, spec, pad[10]; };   // one 128-byte line
#endif
// (solves / evals / scan / culled / spec: k_solve -- spec = ladder candidates evaluated behind the accepted one;
// round_scan: table evaluations of k_round's seed scans)
__device__ __forceinline__ StatSlot *stat_slot(StatSlot *slots) {
  return slots + (((blockIdx.x * blockDim.x + threadIdx.x) >> 6) & (kStatSlots - 1));
}
// Per-batch control block (device memory; the counters are cleared by k_prep at the start of
// every evaluation, start/count are written by the host once per point upload).
struct BatchCtl {
  int start, count;               // points [start, start + count) of the sorted shard
  int n_active[kMaxIter + 2];     // [i] = GSIP points entering iteration i; [0] = interior points
  int n_solve[kMaxIter + 2];      // [i] = sample solves requested for iteration i
  int n_seed[kMaxIter + 2];       // [i] = circle samples emitted in iteration i (statistics)
  unsigned work[kWorkCounters];   // dynamic work-fetch cursors, one per solve launch
  unsigned rwork[kMaxIter + 2];   // the same for the k_round launches (block iterations beyond the grid's first generation)
  int nonfinite;
  int n_int;                      // ctl[0] only: interior points of the whole shard so far = next compact interior index
  // work counters (statistics), added to by every wave at the end of a launch: one address takes ~80 M atomics / s
  // (tools/experiments/coh_latency.hip), a full grid of waves ending together would queue up behind four words of one
  // cache line -- the waves spread over kStatSlots lines, k_finish adds them up
  StatSlot stat[kStatSlots];
  // clock probe (round 5): shader-clock cycles (s_memtime) and constant-rate cycles (s_memrealtime) block 0's first wave of
  // this batch's MAIN solve spent between its table staging and its last query -- both counters read by the same wave, so
  // their ratio is the shader clock the evaluation actually ran at (svsdf_stats.shader_clock_mhz), whatever rocm-smi says
  unsigned long long clk[2];
};

// Piece durations for wave-uniform indices: the same T[] of the TrajDev in global memory, read through the constant
// address space, i.e. with scalar loads (s_load_dwordx8: four durations per instruction into SGPRs, no VALU or LDS
// issue slot).  Only kernels launched AFTER the k_prep that wrote them may use it (the scalar cache is not coherent
// with stores of the running kernel).
typedef const __attribute__((address_space(4))) double *UniformDoubles;

// LDS view of the trajectory
struct TrajL {
  const double *T, *S, *c;
  UniformDoubles Tu;   // T[] for wave-uniform indices (scalar loads); see chain_local_time
  int N;
  int exact;
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
  tr.T = T; tr.S = S; tr.c = c; tr.N = N; tr.dur = g->dur; tr.exact = g->exact;
  tr.Tu = (UniformDoubles)(g->T);
  return tr;
}

// kPolygonLds kernels: copy the outline's edges (6 doubles each) to the START of the block's dynamic LDS (shape_sdf reads
// them from there, svsdf_shapes.hpp); the kernel's other LDS tables follow at poly_lds_doubles<SHAPE>(nverts).  Call
// before the block's first __syncthreads.
template <int SHAPE>
__device__ __forceinline__ void stage_poly_edges(const ShapeParams &sp, double *lds) {
  if constexpr (SHAPE == kPolygonLds) {
    const double *src = reinterpret_cast<const double *>(sp.edges);
    for (int i = threadIdx.x; i < kPolyEdgeDoubles * sp.nverts; i += blockDim.x) lds[i] = src[i];
  }
}

// Trajectory::locatePieceIdx (TRJ:498-516) on cumulative start times: piece = first i with
// t <= S[i+1] (clamped to N-1), local time = t - S[i].  `i` is a per-lane cache of the last piece.
__device__ __forceinline__ int locate_piece(const TrajL &tr, double t, int i) {
  while (i < tr.N - 1 && t > tr.S[i + 1]) ++i;
  while (i > 0 && !(t > tr.S[i])) --i;
  return i;
}

// Opt-in faithful form (SVSDF_FLAG_EXACT_PIECE_TIME): Trajectory::locatePieceIdx exactly as written (TRJ:498-516) --
// the durations are subtracted one after the other from t (i roundings instead of one) and the piece index follows
// from comparing the running remainder with T[i].  O(N) per call: +19 % (C2) ... +68 % (C3) evaluation time, which is
// why the default locates on the cumulative start times (identical whenever the partial sums are exact, e.g. the
// equal 2.5 s pieces of every BASELINE config; within i ulp(t) otherwise).
__device__ __forceinline__ int locate_local_exact(const TrajL &tr, double t, double &s) {
  int i = 0;
  double dur = 0.0;
  for (; i < tr.N && t > (dur = tr.T[i]); ++i) t -= dur;
  if (i == tr.N) { --i; t += tr.T[i]; }
  s = t;
  return i;
}
// piece index and local time of global time t (either form)
__device__ __forceinline__ int locate_local(const TrajL &tr, double t, int hint, double &s) {
  if (tr.exact) return locate_local_exact(tr, t, s);
  const int i = locate_piece(tr, t, hint);
  s = t - tr.S[i];
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

// sincos for |a| < 2^30: the exact operation sequence of the ROCm device library's
// __ocml_sincos_f64 (trigredsmall + sincosred2 + quadrant selection, ROCm 7.2 ocml.bc) written
// out so that it inlines without the large-argument (Payne-Hanek) branch and the inf/nan
// handling; bit-identical results (checked on device by svsdf_debug_sincos_mismatches).
// Larger arguments take the library routine.
__device__ __forceinline__ void sincos_exact(double a, double *sn, double *cs) {
  const double ax = fabs(a);
  if (!(ax < 0x1p30)) { sincos(a, sn, cs); return; }
  // __ocmlpriv_trigredsmall_f64
  const double r = rint(ax * 0x1.45f306dc9c883p-1);
  const double t4 = __builtin_fma(r, -0x1.921fb54442d18p+0, ax);
  const double t5 = __builtin_fma(r, -0x1.1a62633145c00p-54, t4);
  const double t6 = r * 0x1.1a62633145c00p-54;
  const double t8 = __builtin_fma(r, 0x1.1a62633145c00p-54, -t6);
  const double t9 = t4 - t6;
  const double t10 = t4 - t9;
  const double t11 = t10 - t6;
  const double t12 = t9 - t5;
  const double t13 = t12 + t11;
  const double t14 = t13 - t8;
  const double t15 = __builtin_fma(r, -0x1.b839a252049c0p-104, t14);
  const double x = t5 + t15;            // reduced argument, high part
  const double y = t15 - (x - t5);      // low part
  const int q = (int)r & 3;
  // __ocmlpriv_sincosred2_f64(x, y)
  const double s = x * x;
  const double h = s * 0.5;
  const double u5 = 1.0 - h;
  const double u7 = (1.0 - u5) - h;
  const double s2 = s * s;
  double pc = __builtin_fma(s, -0x1.907db46cc5e42p-37, 0x1.1eeb69037ab78p-29);
  pc = __builtin_fma(s, pc, -0x1.27e4fa17f65f6p-22);
  pc = __builtin_fma(s, pc, 0x1.a01a019f4ec90p-16);
  pc = __builtin_fma(s, pc, -0x1.6c16c16c16967p-10);
  pc = __builtin_fma(s, pc, 0x1.5555555555555p-5);
  const double ny = -y;
  const double c15 = __builtin_fma(x, ny, u7);
  const double c16 = __builtin_fma(s2, pc, c15);
  const double cv = u5 + c16;
  double ps = __builtin_fma(s, 0x1.5e0b2f9a43bb8p-33, -0x1.ae600b42fdfa7p-26);
  ps = __builtin_fma(s, ps, 0x1.71de3796cde01p-19);
  ps = __builtin_fma(s, ps, -0x1.a01a019e83e5cp-13);
  ps = __builtin_fma(s, ps, 0x1.1111111110bb3p-7);
  const double xs = x * (-s);
  const double s25 = __builtin_fma(xs, ps, y * 0.5);
  const double s26 = __builtin_fma(s, s25, ny);
  const double s27 = __builtin_fma(xs, -0x1.5555555555555p-3, s26);
  const double sv = x - s27;
  // quadrant selection of __ocml_sincos_f64
  const int flip = (q > 1) ? (int)0x80000000 : 0;
  const bool even = (q & 1) == 0;
  const double so = even ? sv : cv;
  const double co = even ? cv : -sv;
  const int sgn_in = __double2hiint(a) & (int)0x80000000;
  *sn = __hiloint2double(__double2hiint(so) ^ sgn_in ^ flip, __double2loint(so));
  *cs = __hiloint2double(__double2hiint(co) ^ flip, __double2loint(co));
}

#ifdef SVSDF_API_TU   // shape-independent kernel: compiled once, by svsdf_pipeline.hip
// mismatch counter for sincos_exact vs the library sincos (diagnostics / test only)
__global__ void k_sincos_check(double lo, double hi, int n, unsigned long long *mism) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = lo + (hi - lo) * ((double)i / (double)(n - 1));
  double s0, c0, s1, c1;
  sincos(a, &s0, &c0);
  sincos_exact(a, &s1, &c1);
  if (__double_as_longlong(s0) != __double_as_longlong(s1) || __double_as_longlong(c0) != __double_as_longlong(c1))
    atomicAdd(mism, 1ull);
}

#endif  // SVSDF_API_TU

// per-lane cache of the current piece and its [S_lo, S_hi] interval (avoids the LDS walk when
// consecutive evaluations stay in one piece, which is the rule in the scan layers and the descent)
struct PieceCache {
  int piece;
  double lo, hi;
};
__device__ __forceinline__ PieceCache piece_cache_init() { PieceCache pc; pc.piece = 0; pc.lo = 1.0; pc.hi = 0.0; return pc; }

// The reference's chain of subtractions (TRJ:498-516): r_0 = t, r_{j+1} = r_j - T_j, without the comparisons for
// j < nb = piece - 1 (see pose_at), with them from there on.
//  * Uniform phase.  Every lane walks the SAME index j, so the durations are wave-uniform operands: a window of 8 is
//    fetched in one go (LDS reads at a uniform address; SVSDF_CHAIN == 2: scalar loads into SGPRs) and a step is one
//    v_add_f64 for the whole wave, in blocks of four while every active lane still has four comparison-free steps left,
//    then singly until the first lane reaches its nb.  Neighbouring queries sit in the same or adjacent pieces: almost
//    the whole chain.
//  * Per-lane rest: the lanes that are further along finish their comparison-free steps (LDS reads, divergent).
//  * Comparisons: normally exactly two -- r > T[piece - 1] holds, r > T[piece] does not -- on two durations the caller
//    fetched per lane BEFORE the chain (their LDS latency runs under the uniform phase, like that of the piece's
//    coefficients); anything else (rounding moved t across a piece boundary) continues the reference's loop as written.
// Same operations on the same operands in the same order as the per-lane loop of rounds 1-3: identical bits.
// Round 4 measured what this buys and what it cannot: k_solve is VALU/scalar-issue bound and a chain of nb dependent
// additions is nb instructions no form can drop (profiles/r04_chain_*: + 7 % at 32 pieces for the additions alone);
// the per-lane loop of rounds 1-3 cost + 22 / + 28 % (C3 / NS), this form + 20 / + 20 %; straight-line predicated blocks
// with scalar-loaded windows of 16 + 17 / + 18 % but 3 % on the cumulative-time path.
#ifndef SVSDF_CHAIN
#define SVSDF_CHAIN 3   // where the window comes from: 2 scalar loads (TrajL::Tu), 3 LDS (uniform address)
#endif
struct ChainWin { double w[8]; };
__device__ __forceinline__ ChainWin chain_window(const TrajL &tr, int jb) {
  ChainWin c;
#if SVSDF_CHAIN == 2
  const UniformDoubles Tu = tr.Tu + jb;      // (may read a few doubles past T[N - 1]: still inside TrajDev)
#pragma unroll
  for (int k = 0; k < 8; ++k) c.w[k] = Tu[k];
#else
  const double *Tl = tr.T + jb;              // (may read a few doubles past T[N - 1]: S[] follows in LDS)
#pragma unroll
  for (int k = 0; k < 8; ++k) c.w[k] = Tl[k];
#endif
  return c;
}
__device__ __forceinline__ double chain_local_time(const TrajL &tr, double t, int piece, double tprev, double tcur, ChainWin w, int &idx) {
  const int N = tr.N;
  const int nb = (tr.exact == 2) ? 0 : piece - 1;
  double r = t;
  int j = 0;
  for (int jb = 0;; jb += 8) {   // wave-uniform
    if (jb) w = chain_window(tr, jb);
    if (__all(nb >= jb + 4)) {
      r = (((r - w.w[0]) - w.w[1]) - w.w[2]) - w.w[3];
      j = jb + 4;
      if (__all(nb >= jb + 8)) {
        r = (((r - w.w[4]) - w.w[5]) - w.w[6]) - w.w[7];
        j = jb + 8;
        continue;
      }
      if (__all(nb > j)) { r -= w.w[4]; ++j; if (__all(nb > j)) { r -= w.w[5]; ++j; if (__all(nb > j)) { r -= w.w[6]; ++j; } } }
    } else {
      if (__all(nb > j)) { r -= w.w[0]; ++j; if (__all(nb > j)) { r -= w.w[1]; ++j; if (__all(nb > j)) { r -= w.w[2]; ++j; } } }
    }
    break;
  }
  for (int jj = j; jj < nb; ++jj) r -= tr.T[jj];   // per lane: the lanes further along than the wave's first
  int jj = piece;
  if (tr.exact == 2) {
    jj = 0;   // (some piece shorter than 1e-6 s: every step with its comparison)
  } else {
    if (piece > 0) {
      if (!(r > tprev)) { idx = piece - 1; return r; }
      r -= tprev;
    }
    if (!(r > tcur)) { idx = piece; return r; }
    r -= tcur;
    ++jj;
  }
  double dur_ = 0.0;
  for (; jj < N && r > (dur_ = tr.T[jj]); ++jj) r -= dur_;
  if (jj == N) { --jj; r += tr.T[jj]; }   // past the end: the last piece, TRJ:510-514
  idx = jj;
  return r;
}

__device__ __forceinline__ Pose pose_at(const TrajL &tr, double t, PieceCache &pc) {
  if (tr.exact) {   // wave-uniform
    // Faithful piece-local time at the evaluation site.  The cached cumulative locate gives a candidate index ih
    // (t in (S[ih], S[ih+1]]); the reference's chain r_0 = t, r_{j+1} = r_j - T_j runs without its comparisons for
    // j < ih - 1 -- there r_j - T_j ~ t - S_{j+1} >= T_{ih-1} > 0 by a margin of >= min T (host: >= 1e-6, else
    // exact == 2 and the full loop runs) against rounding of <= 64 ulp(t) -- and with them from there on, so index
    // and local time are the reference's to the last bit (TRJ:498-516).
    const ChainWin w0 = chain_window(tr, 0);   // (issued first: in flight during the cache check)
    if (!(t > pc.lo && t <= pc.hi)) {
      pc.piece = locate_piece(tr, t, pc.piece);
      pc.lo = (pc.piece == 0) ? -1e300 : tr.S[pc.piece];
      pc.hi = (pc.piece == tr.N - 1) ? 1e300 : tr.S[pc.piece + 1];
    }
    // everything the chain's end needs is requested before the chain starts: the two durations its comparisons use and
    // the coefficients of the piece it normally ends in (it ends elsewhere only when rounding moved t across a boundary)
    const int piece = pc.piece;
    const double tprev = tr.T[(piece > 0) ? piece - 1 : 0], tcur = tr.T[piece];
    const double *cp = tr.c + piece * 18;
    double cc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) cc[k] = cp[k];
    int jj;
    const double r = chain_local_time(tr, t, piece, tprev, tcur, w0, jj);
    if (jj != piece) {
      const double *cq = tr.c + jj * 18;
#pragma unroll
      for (int k = 0; k < 18; ++k) cc[k] = cq[k];
    }
    double x_, y_, yaw_;
    piece_pos(cc, r, x_, y_, yaw_);
    Pose p_;
    p_.x = x_; p_.y = y_;
    sincos_exact(yaw_, &p_.sn, &p_.cs);
    return p_;
  }
  // same piece as locate_piece: t in (S[i], S[i+1]] (t <= S[1] for i = 0, t > S[N-1] for i = N-1)
  if (!(t > pc.lo && t <= pc.hi)) {
    pc.piece = locate_piece(tr, t, pc.piece);
    pc.lo = (pc.piece == 0) ? -1e300 : tr.S[pc.piece];
    pc.hi = (pc.piece == tr.N - 1) ? 1e300 : tr.S[pc.piece + 1];
  }
  const int piece = pc.piece;
  const double s = t - ((piece == 0) ? 0.0 : pc.lo);
  double x, y, yaw;
  piece_pos(tr.c + piece * 18, s, x, y, yaw);
  Pose p;
  p.x = x; p.y = y;
  sincos_exact(yaw, &p.sn, &p.cs);
  return p;
}

__device__ __forceinline__ Pose pose_at(const TrajL &tr, double t, int &piece) {
  double s;
  piece = locate_local(tr, t, piece, s);
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
template <int SHAPE>
__device__ __forceinline__ double sdf_at(const TrajL &tr, const ShapeParams &sp, double px,
                                         double py, double t, PieceCache &pc) {
  const Pose p = pose_at(tr, t, pc);
  return sdf_from_pose<SHAPE>(sp, p, px, py);
}

#ifdef SVSDF_API_TU   // shape-independent kernel: compiled once, by svsdf_pipeline.hip
// ---------------------------------------------------------------------------------------------
// k_prep: one block.  in = [coeffs (6N x 3 column-major) | T (N) | tk (K)] as uploaded.
// Also clears the per-batch control blocks for this evaluation.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_prep(const double *__restrict__ in, int N, double dur, int K, int exact,
                       TrajDev *__restrict__ tr, Pose *__restrict__ pose,
                       Chunk *__restrict__ chunks, double r_bound, BatchCtl *__restrict__ ctl,
                       int nbatch, int *__restrict__ nonfinite) {
  extern __shared__ double prep_lds[];
  if (threadIdx.x == 0 && nonfinite) *nonfinite = 0;   // (was a memset node of its own in front of every evaluation)
  const double *coeffs = in, *T = in + 18 * N, *tk = in + 19 * N, *slack = in + 19 * N + K;
  for (int b = threadIdx.x; b < nbatch; b += blockDim.x) {
    BatchCtl &c = ctl[b];
    for (int r = 0; r < kMaxIter + 2; ++r) { c.n_active[r] = 0; c.n_solve[r] = 0; c.n_seed[r] = 0; }
    for (int r = 0; r < kWorkCounters; ++r) c.work[r] = 0u;
    for (int r = 0; r < kMaxIter + 2; ++r) c.rwork[r] = 0u;
    c.nonfinite = 0;
    c.n_int = 0;
    c.clk[0] = 0ull; c.clk[1] = 0ull;
  }
  for (int i = threadIdx.x; i < nbatch * kStatSlots; i += blockDim.x) {
    StatSlot &ss = ctl[i / kStatSlots].stat[i % kStatSlots];
    ss.solves = 0ull; ss.evals = 0ull; ss.scan = 0ull; ss.culled = 0ull; ss.round_scan = 0ull; ss.spec = 0ull;
#ifdef SVSDF_SITE_STATS
    for (int j = 0; j < 26; ++j) ss.pad[j] = 0ull;
#endif
  }
  if (threadIdx.x == 0) {
    tr->N = N; tr->K = K; tr->dur = dur; tr->exact = exact;
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
    ch.cx = cx; ch.cy = cy; ch.rb = r * (1.0 + 1e-12) + r_bound + 1e-9; ch.slack = slack[c];
    chunks[c] = ch;
    // the chunk's anchor (ChunkAnchor): the pose with the smallest largest distance to the others, first one wins ties.
    // The chunk's poses go to registers first (a short chunk repeats its last pose: no distance changes) and everything is
    // unrolled, squared distances, one root: the first form -- 64 correctly rounded roots on poses re-read from global
    // memory -- doubled this one-block kernel's time (10 -> 20 us of a 350 us reference-scale callback).
    double ax_[kChunk], ay_[kChunk], ac_[kChunk], as_[kChunk];
#pragma unroll
    for (int q = 0; q < kChunk; ++q) {
      const Pose pq = pose[(k0 + q < k1) ? k0 + q : k1 - 1];
      ax_[q] = pq.x; ay_[q] = pq.y; ac_[q] = pq.cs; as_[q] = pq.sn;
    }
    int qa = 0;
    double adx2 = 1e300;
#pragma unroll
    for (int a = 0; a < kChunk; ++a) {
      double m2 = 0.0;
#pragma unroll
      for (int q = 0; q < kChunk; ++q) {
        const double ux = ax_[q] - ax_[a], uy = ay_[q] - ay_[a];
        m2 = fmax(m2, ux * ux + uy * uy);
      }
      if (m2 < adx2) { adx2 = m2; qa = a; }
    }
    const double adx = sqrt(adx2);
    double rot2 = 0.0, acs = ac_[0], asn = as_[0];
#pragma unroll
    for (int q = 1; q < kChunk; ++q)
      if (qa == q) { acs = ac_[q]; asn = as_[q]; }
#pragma unroll
    for (int q = 0; q < kChunk; ++q) {
      const double dc = ac_[q] - acs, ds = as_[q] - asn;
      rot2 = fmax(rot2, dc * dc + ds * ds);
    }
    const int ka = (k0 + qa < k1) ? k0 + qa : k1 - 1;
    ChunkAnchor an;
    an.adx = adx * (1.0 + 1e-9) + 2e-9;    // (+ the absolute allowance of the test: values are compared as computed)
    an.rot2 = rot2 * (1.0 + 1e-8) + 1e-18;
    an.ka = ka; an.pad_ = 0; an.pad2_ = 0.0;
    reinterpret_cast<ChunkAnchor *>(chunks + nch)[c] = an;
  }
}

#endif  // SVSDF_API_TU

// Shape bound radius: max over a polar grid of |q| - sdf_shape(q) (body frame, including the
// shape's own offset/rotation).  For an exact SDF this is the circumradius about the origin.
// out[1] (round 5): 1-Lipschitz self-check of the shape SDF -- the second exact cull (scan_layer1) and the
// anchor bound mode (k_round MODE 3) rest on |f(a) - f(b)| <= |a - b|, which holds for every exact distance function but is
// asserted nowhere else: the largest excess |f(q') - f(q)| - |q' - q| (1 + 1e-9) - 1e-12 over the grid samples q and four
// neighbours q' each (1 mm steps in x and y: the scale of the cull's allowances; the radial and angular grid neighbours:
// ~ 0.1 m).  A positive value switches both devices off for the context (svsdf_create).
template <int SHAPE>
__global__ void k_rbound(ShapeParams sp, double rmax, int nrad, int nang, double *__restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  double v = -1e300, lip = -1e300;
  if (idx < nrad * nang) {
    const int ir = idx / nang, ia = idx % nang;
    const double r = rmax * (double)(ir + 1) / (double)nrad;
    const double a = 2.0 * kPI * (double)ia / (double)nang;
    const double x = r * cos(a), y = r * sin(a);
    const double f0 = shape_sdf<SHAPE>(sp, x, y);
    v = r - f0;
    auto excess = [&](double xx, double yy) {
      const double d = norm2(xx - x, yy - y);
      return fabs(shape_sdf<SHAPE>(sp, xx, yy) - f0) - d * (1.0 + 1e-9) - 1e-12;
    };
    const double r2 = rmax * (double)(ir + 2) / (double)nrad, a2 = 2.0 * kPI * (double)(ia + 1) / (double)nang;
    lip = fmax(fmax(excess(x + 1e-3, y), excess(x, y + 1e-3)), fmax(excess(r2 * cos(a), r2 * sin(a)), excess(r * cos(a2), r * sin(a2))));
  }
  for (int m = 32; m >= 1; m >>= 1) { v = fmax(v, __shfl_xor(v, m, 64)); lip = fmax(lip, __shfl_xor(lip, m, 64)); }
  if ((threadIdx.x & 63) == 0) {
    // atomic max on a non-negative double via its integer ordering
    if (v > 0.0) atomicMax((unsigned long long *)out, (unsigned long long)__double_as_longlong(v));
    if (lip > 0.0) atomicMax((unsigned long long *)(out + 1), (unsigned long long)__double_as_longlong(lip));
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
  const int *seed_k;      // per slot: layer-1 seed already known (GSIP samples scanned by k_round), or null
  const double *seed_d;
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
  return true;
}

// -DSVSDF_SITE_STATS (diagnostic builds, tools/site_stats.py): per evaluation site of k_solve -- table scan 0, layers 2-4 1,
// FD derivative 2, ladder 3 -- the number of wave-level executions (StatSlot::pad[site]) and of lanes that evaluate there
// (pad[4 + site]): lanes / (64 x executions) is the site's lane occupancy.
#ifdef SVSDF_SITE_STATS
#define SVSDF_SITE(cnt, site, evaluates)                                                     \
  do {                                                                                       \
    const unsigned long long ex__ = __ballot(1);                                             \
    if ((int)(threadIdx.x & 63) == __ffsll((long long)ex__) - 1) ++(cnt)[site];              \
    if (evaluates) ++(cnt)[4 + (site)];                                                      \
  } while (0)
#define SVSDF_SITE_CLOCK() ((unsigned long long)__builtin_readcyclecounter())
// wave-level cycles of a phase (8 scan, 9 layers 2-4, 10 descent), counted once per wave
#define SVSDF_SITE_CYCLES(cnt, slot, t0) do { if ((threadIdx.x & 63) == 0) (cnt)[slot] += SVSDF_SITE_CLOCK() - (t0); } while (0)
// k_round phases (pad[12 ..]: close, candidate list, samples + cheap bound, seed scans, selection, flush, whole wave, staging)
#define SVSDF_PHASE(cnt, slot, t0) do { if ((threadIdx.x & 63) == 0) { const unsigned long long n__ = SVSDF_SITE_CLOCK(); (cnt)[slot] += n__ - (t0); (t0) = n__; } } while (0)
#else
#define SVSDF_SITE(cnt, site, evaluates) do { } while (0)
#define SVSDF_SITE_CLOCK() 0ull
#define SVSDF_SITE_CYCLES(cnt, slot, t0) do { (void)(t0); } while (0)
#define SVSDF_PHASE(cnt, slot, t0) do { (void)(t0); } while (0)
#endif

// G lanes cooperate on one query (64/G queries per wave).
template <int G>
struct Grp {
  static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32, "group size");
  __device__ static __forceinline__ int li() { return (int)(threadIdx.x & (G - 1)); }
  __device__ static __forceinline__ double bcast(double v, int src) {
    if constexpr (G == 1) return v; else return __shfl(v, src, G);
  }
  // Exchange partner `step` of the butterfly over the group (step 0 .. log2(G) - 1).  Within 16 lanes the
  // partners come through DPP (a VALU operand modifier, no LDS round trip): xor 1 and xor 2 as quad
  // permutations, then the mirror of the 8-lane half (lane i <-> 7 - i) and of the 16-lane row (i <-> 15 - i),
  // which pair lanes of different quads / halves just like xor 4 / xor 8 would; the 32-lane step is a shuffle.
  template <int STEP>
  __device__ static __forceinline__ int xchg(int v) {
    if constexpr (STEP == 0) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if constexpr (STEP == 1) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if constexpr (STEP == 2) return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    else if constexpr (STEP == 3) return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false);  // row_mirror
    else return __shfl_xor(v, 16, 32);
  }
  template <int STEP>
  __device__ static __forceinline__ double xchg(double v) {
    const int lo = xchg<STEP>(__double2loint(v)), hi = xchg<STEP>(__double2hiint(v));
    return __hiloint2double(hi, lo);
  }
  // min over the group; all lanes get the result
  __device__ static __forceinline__ int min_i(int v) {
    if constexpr (G >= 2) v = min(v, xchg<0>(v));
    if constexpr (G >= 4) v = min(v, xchg<1>(v));
    if constexpr (G >= 8) v = min(v, xchg<2>(v));
    if constexpr (G >= 16) v = min(v, xchg<3>(v));
    if constexpr (G >= 32) v = min(v, xchg<4>(v));
    return v;
  }
  template <int STEP>
  __device__ static __forceinline__ void min_dk_step(double &d, int &k) {
    const double od = xchg<STEP>(d);
    const int ok = xchg<STEP>(k);
    if (od < d || (od == d && ok < k)) { d = od; k = ok; }
  }
  // lexicographic min of (d, k) over the group; all lanes get the result
  __device__ static __forceinline__ void min_dk(double &d, int &k) {
    if constexpr (G >= 2) min_dk_step<0>(d, k);
    if constexpr (G >= 4) min_dk_step<1>(d, k);
    if constexpr (G >= 8) min_dk_step<2>(d, k);
    if constexpr (G >= 16) min_dk_step<3>(d, k);
    if constexpr (G >= 32) min_dk_step<4>(d, k);
  }
  // min of a double over the group; all lanes get the result
  __device__ static __forceinline__ double min_d(double v) {
    if constexpr (G >= 2) v = dmin(v, xchg<0>(v));
    if constexpr (G >= 4) v = dmin(v, xchg<1>(v));
    if constexpr (G >= 8) v = dmin(v, xchg<2>(v));
    if constexpr (G >= 16) v = dmin(v, xchg<3>(v));
    if constexpr (G >= 32) v = dmin(v, xchg<4>(v));
    return v;
  }
  // lexicographic min of (d, k) over the group like min_dk, in two plain butterflies (round 6): the minimum VALUE first
  // (3 instructions per step), then the smallest index among the lanes that hold it (2 per step) -- 5 instead of the 11 a
  // step of the pair butterfly costs (three exchanges, three compares, two mask operations, three selects).  Same pair: the
  // lexicographic minimum IS the smallest index among the carriers of the smallest value.  (A NaN never carries it.)
  __device__ static __forceinline__ void min_dk2(double &d, int &k) {
    const double m = min_d(d);
    k = min_i((d == m) ? k : 0x7fffffff);
    d = m;
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
__device__ __forceinline__ long long fetch_work(unsigned *cursor, long long &wave_base, int per_wave = M * 64 / G) {
  unsigned base = 0;
  if ((threadIdx.x & 63) == 0) base = atomicAdd(cursor, (unsigned)per_wave);
  base = __builtin_amdgcn_readfirstlane(base);
  wave_base = (long long)base;
  return (long long)base + (long long)((threadIdx.x & 63) / G);
}

// ---------------------------------------------------------------------------------------------
// choiceTInit layer 1 (SWM:549-576, first pass) for one query by G cooperating lanes, on the pose table in LDS:
// the chunk with the smallest lower bound first, then every chunk whose bound does not exceed the running
// minimum (exact: a skipped sample can never be, or tie with, the strict-`<` minimum).  Returns the seed
// (best_d = min_dis, best_k = index of the earliest minimal sample); `culled` only with a finite cull_thresh.
// ---------------------------------------------------------------------------------------------
// LITE (GSIP samples: no cull bound needed): square-root-free chunk tests -- first chunk = nearest centre, then
// every chunk with |q - c|^2 <= (min + rb)^2 (1 + 1e-12), a superset of the exact test, so the seed is the same.
// LITE scans may be restricted to a list of candidate chunks (clist[0 .. ncl), ascending; ncl < 0: all chunks): the list
// k_round builds per (interior point, round) holds every chunk that can matter for ANY sample of the round's circle.
template <int SHAPE, int G, bool LITE = false>
__device__ __forceinline__ void scan_layer1(const ShapeParams &sp, const Pose *pose, const Chunk *chunks, int K,
                                            int nch, double px, double py, int prune, double cull_thresh,
                                            double &best_d, int &best_k, bool &culled, unsigned &n_scan,
                                            const unsigned short *clist = nullptr, int ncl = -1,
                                            unsigned long long *sc = nullptr, const double *__restrict__ rot = nullptr,
                                            double slack_max = 0.0, const ChunkAnchor *__restrict__ anch = nullptr) {
  const int li = Grp<G>::li();
  best_d = 1e9;   // min_dis initial value (SWM:545)
  best_k = 0x7fffffff;
  culled = false;
  // Every lane keeps the lexicographic minimum (value, index) of the poses IT evaluated; after a chunk only the VALUE is
  // reduced over the group (the pruning tests need nothing else: 3 instructions per butterfly step instead of 9), the
  // (value, index) pairs meet once, at the end of the scan (finish_scan).  Same minimum, same earliest index.
  double d_lane = 1e300;
  int k_lane = 0x7fffffff;
  double vbound = 1e300;   // min over the evaluated chunks of (table minimum of the chunk - allowance), main points only
  auto eval_chunk = [&](int c) {
    double d_loc = 1e300;
    constexpr int GS = (G < kChunk) ? G : kChunk;
#pragma unroll
    for (int m = 0; m < kChunk / GS; ++m) {
      const int k = c * kChunk + li + GS * m;
      if (sc) SVSDF_SITE(sc, 0, li < kChunk && k < K);
      if (li < kChunk && k < K) {
        const Pose p = pose[k];
        const double d = sdf_from_pose<SHAPE>(sp, p, px, py);
        ++n_scan;
        if (d < d_loc) d_loc = d;
        if (d < d_lane || (d == d_lane && k < k_lane)) { d_lane = d; k_lane = k; }
      }
    }
    if constexpr (G >= 2) d_loc = dmin(d_loc, Grp<G>::template xchg<0>(d_loc));
    if constexpr (G >= 4) d_loc = dmin(d_loc, Grp<G>::template xchg<1>(d_loc));
    if constexpr (G >= 8) d_loc = dmin(d_loc, Grp<G>::template xchg<2>(d_loc));
    if constexpr (G >= 16) d_loc = dmin(d_loc, Grp<G>::template xchg<3>(d_loc));
    if constexpr (G >= 32) d_loc = dmin(d_loc, Grp<G>::template xchg<4>(d_loc));
    if (d_loc < best_d) best_d = d_loc;
    if constexpr (!LITE) {
      if (rot) {   // second exact cull (round 4), see below: the chunk's table minimum less its continuous-path allowance
        const Chunk ch = chunks[c];
        vbound = dmin(vbound, d_loc - (ch.slack + rot[c] * (norm2(px - ch.cx, py - ch.cy) + ch.rb)));
      }
    }
  };
  auto finish_scan = [&]() {
    if (best_d < 1e9) {
      // (group-uniform) best_d = the smallest of all evaluated values = min over the lanes of d_lane: the earliest index is
      // the smallest k_lane among the lanes that hold that value -- one integer butterfly instead of the pair butterfly
      best_k = Grp<G>::min_i((d_lane == best_d) ? k_lane : 0x7fffffff);
    } else {
      Grp<G>::min_dk(d_lane, k_lane);
      // (lexicographic minimum with the initial (1e9, none), like the sequential update it replaces)
      if (d_lane < 1e9 || (d_lane == 1e9 && k_lane != 0x7fffffff)) { best_d = d_lane; best_k = k_lane; }
    }
  };
  if constexpr (LITE) {
    const int nl = (ncl < 0) ? nch : ncl;
    auto chunk_at = [&](int j) -> int { return (ncl < 0) ? j : (int)clist[j]; };
    double d2_loc = 1e300;
    int c_loc = 0;
    for (int j = li; j < nl; j += G) {
      const int c = chunk_at(j);
      const Chunk ch = chunks[c];
      const double ex = px - ch.cx, ey = py - ch.cy;
      const double d2 = ex * ex + ey * ey;
      if (d2 < d2_loc) { d2_loc = d2; c_loc = c; }
    }
    Grp<G>::min_dk2(d2_loc, c_loc);
    const int c0 = c_loc;
    eval_chunk(c0);
    int j = 0;
    if constexpr (G == 8) {
      // Anchored walk (round 6; `anch`: the chunks' anchor table, null = off).  The circle test |q - c| - rb knows a chunk only
      // by its bounding circle plus the shape's circumradius: for a large or thin shape (sdHeart: rb ~ 5 m) a scan evaluated 8
      // chunks where 2 - 3 hold a pose that matters.  Here the survivors of the circle test (against the first chunk's minimum)
      // FIRST get their anchor pose evaluated, eight chunks per step -- each a real table value, so it lowers the running
      // minimum at once -- and a chunk is then evaluated in full only if the anchored bound (ChunkAnchor) also reaches the
      // running minimum.  CPU study (tools/experiments/anchor_chunk_bound.py): 8.1 -> 4.4 steps per scan at C4, 5.0 -> 3.9 at
      // C3; with fewer than SVSDF_ANCHOR_MIN survivors the anchors cost more steps than they save (star: 2.4 -> 3.0) and the
      // plain walk runs.  A chunk is skipped only when a lower bound of ALL its poses lies strictly above a value already
      // found: the (value, earliest index) minimum is the same.
      if (anch != nullptr && nl <= 64) {
        unsigned long long M = 0ull;   // bit j <-> list position j survives the circle test
        for (int jb = 0; jb < nl; jb += 8) {
          const int jj = jb + li;
          bool need = false;
          if (jj < nl) {
            const int cc = chunk_at(jj);
            if (cc != c0) {
              const Chunk ch = chunks[cc];
              const double ex = px - ch.cx, ey = py - ch.cy;
              const double t = best_d + ch.rb;
              need = (t >= 0.0) && (ex * ex + ey * ey <= t * t * (1.0 + 1e-12));
            }
          }
          M |= (unsigned long long)Grp<G>::ballot(need) << jb;
        }
        const int ns = __popcll(M);
        if (ns >= SVSDF_ANCHOR_MIN) {
          // lane li takes the survivor of rank li (ascending list position); survivors beyond the eighth -- rare -- are left
          // to the plain walk below, which resumes behind the eighth one (and tests them against the minimum found by then)
          unsigned long long m = M;
          for (int q = 0; q < li; ++q) m &= m - 1ull;
          j = nl;
          if (ns > 8) {
            for (int q = 0; q < 8; ++q) M &= M - 1ull;
            j = __ffsll((long long)M) - 1;
          }
          bool nf = m != 0ull;
          const int cc = chunk_at(nf ? (__ffsll((long long)m) - 1) : 0);
          double lba = 1e300, da = 1e300;   // lba: lower bound of every pose of chunk cc from its anchor
          if (sc) SVSDF_SITE(sc, 0, nf);
          if (nf) {
            const ChunkAnchor an = anch[cc];
            const int ka = an.ka;
            const Pose p = pose[ka];
            const double ax = px - p.x, ay = py - p.y;
            // (the allowance before the evaluation: one number is live across it instead of the anchor's record and position)
            const double allow = an.adx + sqrt(an.rot2 * (ax * ax + ay * ay)) * (1.0 + 1e-12);
            da = sdf_from_pose<SHAPE>(sp, p, px, py);
            ++n_scan;
            if (da < d_lane || (da == d_lane && ka < k_lane)) { d_lane = da; k_lane = ka; }
            lba = da - allow;
          }
          da = dmin(da, Grp<G>::template xchg<0>(da));
          da = dmin(da, Grp<G>::template xchg<1>(da));
          da = dmin(da, Grp<G>::template xchg<2>(da));
          if (da < best_d) best_d = da;
          for (;;) {
            if (nf) {
              const Chunk ch = chunks[cc];
              const double ex = px - ch.cx, ey = py - ch.cy;
              const double t = best_d + ch.rb;
              nf = !(lba > best_d) && (t >= 0.0) && (ex * ex + ey * ey <= t * t * (1.0 + 1e-12));
            }
            const unsigned mm = Grp<G>::ballot(nf);
            if (mm == 0u) break;
            const int first = __ffs(mm) - 1;
            eval_chunk(__shfl(cc, first, G));
            if (li == first) nf = false;
          }
        }
      }
    }
    while (j < nl) {
      const int jj = j + li;
      bool need = false;
      int cc = 0;
      if (jj < nl) {
        cc = chunk_at(jj);
        if (cc != c0) {
          const Chunk ch = chunks[cc];
          const double ex = px - ch.cx, ey = py - ch.cy;
          const double t = best_d + ch.rb;
          need = (t >= 0.0) && (ex * ex + ey * ey <= t * t * (1.0 + 1e-12));
        }
      }
      const unsigned m = Grp<G>::ballot(need);
      if (m == 0u) { j += G; continue; }
      const int first = __ffs(m) - 1;
      eval_chunk(__shfl(cc, first, G));
      j = j + first + 1;
    }
    finish_scan();
  } else if (!prune) {
    for (int c = 0; c < nch; ++c) eval_chunk(c);
    finish_scan();
  } else {
    // 1. the chunk with the smallest lower bound gives the first upper bound
    double lb_loc = 1e300, lbc_loc = 1e300;
    int c_loc = 0;
    for (int c = li; c < nch; c += G) {
      const Chunk ch = chunks[c];
      const double lb = norm2(px - ch.cx, py - ch.cy) - ch.rb;
      if (lb < lb_loc) { lb_loc = lb; c_loc = c; }
      lbc_loc = dmin(lbc_loc, lb - ch.slack);
    }
    Grp<G>::min_dk2(lb_loc, c_loc);
    if constexpr (G >= 2) lbc_loc = dmin(lbc_loc, Grp<G>::template xchg<0>(lbc_loc));
    if constexpr (G >= 4) lbc_loc = dmin(lbc_loc, Grp<G>::template xchg<1>(lbc_loc));
    if constexpr (G >= 8) lbc_loc = dmin(lbc_loc, Grp<G>::template xchg<2>(lbc_loc));
    if constexpr (G >= 16) lbc_loc = dmin(lbc_loc, Grp<G>::template xchg<3>(lbc_loc));
    if constexpr (G >= 32) lbc_loc = dmin(lbc_loc, Grp<G>::template xchg<4>(lbc_loc));
    // exact cull (main points only; cull_thresh = +inf otherwise): every pose of the continuous path keeps
    // sdf >= lbc_loc > safety_hor (upload_traj), so smoothedL1 is inactive (BEO:316-340, x < 0) whatever
    // local minimum the reference's search would return: the point contributes exactly zero
    culled = lbc_loc > cull_thresh;
    if (culled) { best_d = lbc_loc; best_k = 0; }
    const int c0 = c_loc;
    if (!culled) eval_chunk(c0);
    // 2. every other chunk whose lower bound does not exceed the running minimum
    int c = culled ? nch : 0;
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
    if (!culled) finish_scan();
    // Second exact cull (round 4; main points; needs a 1-Lipschitz shape SDF: true of every exact distance function, i.e.
    // all 17 shapes; k_rbound samples the property per context as a sanity check -- a shape that fails it runs without the cull).
    // The first cull knows a chunk only by its bounding circle (|p - c| - rb: loose by up to the shape's circumradius --
    // 15 % of C3's points are inactive yet survive it).  After the scan the table VALUES are known.  For a time t of an
    // EVALUATED chunk's interval, within h of a table time t_k, the body-frame point q(t) = R(t)^T (p - x(t)) obeys
    //   q(t) - q(t_k) = R(t)^T (x_k - x(t)) + (R(t) - R(t_k))^T (p - x_k),
    // so |q(t) - q(t_k)| <= h V_c + h W_c |p - x_k|  (V_c, W_c: Bernstein bounds of planar speed and yaw rate on the
    // interval, host; |R(t) - R(t_k)| <= |yaw(t) - yaw(t_k)|), and |p - x_k| <= |p - c| + r_c <= |p - c| + rb_c (x_k lies in
    // the chunk's circle).  By the Lipschitz property  sdf(t) >= sdf(t_k) - h (V_c + W_c (|p - c| + rb_c)).  A PRUNED chunk
    // has every table value above the running minimum and, by the circle argument,  sdf(t) >= lb_c - slack_c > best_d -
    // max_c slack_c.  If the smaller of the two bounds exceeds safety_hor, every pose of the continuous path keeps the
    // point inactive whatever local minimum the reference's search returns: it contributes exactly zero and the 215
    // evaluations of layers 2-4 and the descent are skipped.
    if (!culled && rot) {
      const double bound = dmin(vbound, best_d - slack_max) - 1e-9;
      if (bound > cull_thresh) { culled = true; best_d = bound; best_k = 0; }
    }
  }
}

// Layers 2-4 of choiceTInit (SWM:557-577) + gradientDescent (SWM:1249-1325) for ONE query by G cooperating lanes,
// from the layer-1 seed (time tk[best_k], value best_d): the argmin time x and its value fx (all lanes of the group get
// them).
//
// Called by ALL lanes of the wave (`on`: this group has a query): layers 2-4 run per group; the halving ladder of the
// descent is shared out over the whole wave (SVSDF_ELASTIC, below; wave_lds: this wave's ladder_lds_bytes(G) of LDS).
// (80-byte rows: 20 dwords, so the rows of 16 groups start in 16 different LDS banks)
struct alignas(16) LadderState { double px, py, seed, x, fx, prev_x, lo, hi; int sgn, piece, pad_[2]; };   // one group's descent
__host__ __device__ constexpr size_t ladder_lds_bytes(int G) { return (size_t)(64 / G) * sizeof(LadderState) + 64; }
template <int SHAPE, int G, int U>
__device__ __forceinline__ void descend_from_seed(const TrajL &tr, const double *__restrict__ tk, const ShapeParams &sp,
                                                  double px, double py, bool on, int best_k, double best_d,
                                                  double &x_out, double &fx_out, unsigned &n_eval, unsigned &n_spec,
                                                  unsigned long long (&sc)[12], void *wave_lds) {
  const int li = Grp<G>::li();
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
    PieceCache piece = piece_cache_init();
    double time_seed = 0.0;
    double min_dis = best_d;
  const unsigned long long t_lay0 = SVSDF_SITE_CLOCK();
  if (on) {
    time_seed = tk[best_k];

    // ---- choiceTInit layers 2-4: W = G*U samples per step, lane li takes li, li+G, ...
    constexpr int W = G * U;
    double dt = 0.15;
    dt *= 0.1;
    for (int layer = 2; layer <= 4; ++layer) {
      const double t0 = dmax(0.0, time_seed - 10 * dt);
      const double loop_terminal = dmin(tr.dur, time_seed + 10 * dt);
      double t[U];
      t[0] = t0;
      for (int i = 0; i < li; ++i) t[0] += dt;   // the li-th accumulated sample
#pragma unroll
      for (int u = 1; u < U; ++u) {
        t[u] = t[u - 1];
#pragma unroll
        for (int i = 0; i < G; ++i) t[u] += dt;
      }
      int kbase = 0;
      while (Grp<G>::bcast(t[0], 0) <= loop_terminal) {
        double d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          d[u] = inf;
          SVSDF_SITE(sc, 1, t[u] <= loop_terminal);
          if (t[u] <= loop_terminal) { d[u] = sdf_at<SHAPE>(tr, sp, px, py, t[u], piece); ++n_eval; }
        }
        double db = d[0], tb = t[0];
        int k = kbase + li;
#pragma unroll
        for (int u = 1; u < U; ++u)
          if (d[u] < db) { db = d[u]; tb = t[u]; k = kbase + li + G * u; }  // strict: earliest kept
        const int kmine = k;
        Grp<G>::min_dk(db, k);
        // broadcast the time of the winning sample from the lane that holds it
        const int src = (k - kbase) % G;
        const double tw = Grp<G>::bcast((k == kmine) ? tb : 0.0, src);
        if (db < min_dis) { time_seed = tw; min_dis = db; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int i = 0; i < W; ++i) t[u] += dt;
        }
        kbase += W;
      }
      dt *= 0.1;
    }
  }  // on
  SVSDF_SITE_CYCLES(sc, 9, t_lay0);
  const unsigned long long t_desc0 = SVSDF_SITE_CLOCK();

#if SVSDF_ELASTIC
    // ---- gradientDescent, the ladder shared out over the wave.  A pass = the FD derivative at x (per group, 2-3
    // evaluations) + the halving ladder: candidates x - 0.01 * 2^(1-j) * sgn, j = 1 .. 29, the FIRST with f < fx is taken.
    // The candidates of a ladder do not depend on each other, so any number of them may be evaluated at once, by any
    // lane: the 64 lanes of the wave are dealt out evenly to the groups whose ladder is still open (16 groups: 4 lanes
    // each as before; 5 groups: 12 each; one group: its 29 candidates in one step).  Groups of a wave need different
    // numbers of passes and of ladder steps (site counters, tools/site_stats.py: with fixed 4-lane ladders 46 % of the
    // lanes evaluate at this site, with shared ones 85 %); a group that is done hands its lanes to the others instead of
    // idling.  Same accepted candidate, bit for bit.
    // The groups' descent state lives in the wave's LDS rows (one wave reads and writes them in program order: no
    // barrier): a serving lane reads the state of the group it works for and, if its candidate is the first accepted
    // one, writes the new (x, fx) back; the owner picks them up at the top of its next pass.  All open ladders of a
    // wave started their pass together and advance by the same width, so the next candidate index j0 is wave-uniform.
    {
      // plain accesses between wavefront-scope fences (compiler ordering; one wave's LDS operations execute in order)
      LadderState *gs = reinterpret_cast<LadderState *>(wave_lds);
      unsigned char *srcmap = reinterpret_cast<unsigned char *>(gs + 64 / G);
      const int lane = (int)(threadIdx.x & 63);
      const int grp = lane / G;
      LadderState *mine_gs = gs + grp;
      // (the descent's clamp [t_min, t_max] follows from time_seed)
      if (on && li == 0) {
        mine_gs->px = px; mine_gs->py = py; mine_gs->seed = time_seed; mine_gs->x = time_seed; mine_gs->fx = 0.0;
        mine_gs->prev_x = 10000000.0;
        mine_gs->piece = piece.piece; mine_gs->lo = 1.0; mine_gs->hi = 0.0;   // (empty interval: located at first use)
      }
      int iter = 0;
      bool run = on;
      for (bool first_pass = true;; first_pass = false) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (run) run = iter < 1000 && fabs(mine_gs->x - mine_gs->prev_x) > 1e-16;
        const unsigned long long runm = __ballot(run && li == 0);
        if (runm == 0ull) break;
#if SVSDF_FUSED_PASS
        if (__popcll(runm) == 1) {
          // ONE descent left in the wave (round 5; the rule at the reference's scale, where a wave holds one or two queries:
          // a callback there is ~ 230 dependent evaluation steps of ~ 2 us each, and a pass was two of them -- the FD
          // derivative, then the ladder).  The ladder's candidates depend on the derivative only through its SIGN, so both
          // signs are evaluated with it in one step: lanes 0 .. 2 the derivative's tasks, lanes 3 .. 31 the candidates x - tau_j
          // (j = 1 .. 29), lanes 35 .. 63 the candidates x + tau_j; afterwards the sign picks its half, whose first accepted
          // candidate is the one the sequential loop accepts.  Same operations on the same operands: same bits; the other
          // half's 29 evaluations are counted as speculative.  A wave with several open descents keeps the two-step pass
          // (its lanes are busy with the other ladders).
          const int srcg = (__ffsll((long long)runm) - 1) / G;
          LadderState *S = gs + srcg;
          const double x = S->x, qx = S->px, qy = S->py, sseed = S->seed;
          PieceCache pcs;
          pcs.piece = S->piece; pcs.lo = S->lo; pcs.hi = S->hi;
          const int ntask = first_pass ? 3 : 2;
          const bool is_task = lane < ntask;
          const bool is_cand = (lane >= 3 && lane < 32) || lane >= 35;
          const int j = (lane < 32) ? lane - 2 : lane - 34;             // div (candidates)
          const double ssgn = (lane < 32) ? 1.0 : -1.0;
          const double stmin = dmax(0.0, sseed - 3.4), stmax = dmin(sseed + 3.4, tr.dur);
          const double tau = ldexp(0.01, 1 - j);
          const double change = -tau * ssgn;
          double xc = x + change;
          xc = dmax(dmin(xc, stmax), stmin);
          const double t1 = dmax(0.0, x - 0.000001);
          const double t2 = dmin(tr.dur, x + 0.000001);
          const double tt = is_task ? ((lane == 0) ? t1 : (lane == 1) ? t2 : x) : xc;
          double d = inf;
          SVSDF_SITE(sc, 3, is_task || is_cand);
          if (is_task || is_cand) { d = sdf_at<SHAPE>(tr, sp, qx, qy, tt, pcs); ++n_eval; }
          const double sdf1 = __shfl(d, 0, 64), sdf2 = __shfl(d, 1, 64), f0 = __shfl(d, 2, 64);
          const double g = (sdf2 - sdf1) * 500000;
          const int sgn = (int)(g > 0) - (int)(g < 0);
          const double sfx = first_pass ? f0 : S->fx;
          const unsigned long long accm = __ballot(is_cand && (d - sfx) < 0);
          const unsigned bits = (sgn > 0) ? (unsigned)((accm >> 3) & 0x1fffffffull) : (sgn < 0) ? (unsigned)((accm >> 35) & 0x1fffffffull) : 0u;
          const int jacc = bits ? __ffs(bits) : 0;                       // accepted div (1 .. 29), 0: none
          // lane 0 evaluated at t1 = x - 1e-6: its piece interval is the next pass's starting point (as in the two-step pass)
          const int p0 = __shfl(pcs.piece, 0, 64);
          const double lo0 = __shfl(pcs.lo, 0, 64), hi0 = __shfl(pcs.hi, 0, 64);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) {
            S->sgn = sgn; S->prev_x = x; S->piece = p0; S->lo = lo0; S->hi = hi0;
            if (first_pass && !jacc) S->fx = f0;
          }
          if (is_cand && jacc && j == jacc && ((sgn > 0) == (lane < 32))) { S->x = xc; S->fx = d; }
          if (is_cand) n_spec += (jacc && ((sgn > 0) == (lane < 32)) && sgn != 0) ? (j > jacc ? 1u : 0u) : 1u;
          if (run) {   // the owning group's bookkeeping (SWM:1295-1321): trials of this ladder, stop when none was accepted
            iter += jacc ? jacc : 29;
            if (!jacc) run = false;
          }
          continue;
        }
#endif
        if (run) {
          // tasks: 0 -> sdf(t1), 1 -> sdf(t2), 2 -> sdf(x) (first pass only)
          const double x = mine_gs->x, qx = mine_gs->px, qy = mine_gs->py;
          PieceCache dpc;
          dpc.piece = mine_gs->piece; dpc.lo = mine_gs->lo; dpc.hi = mine_gs->hi;
          const int ntask = first_pass ? 3 : 2;
          const double t1 = dmax(0.0, x - 0.000001);
          const double t2 = dmin(tr.dur, x + 0.000001);
          double sdf1 = 0.0, sdf2 = 0.0, f0 = 0.0;
#pragma unroll
          for (int base = 0; base < 3; base += G) {
            if (base < ntask) {
              const int task = base + li;
              double d = 0.0;
              SVSDF_SITE(sc, 2, task < ntask);
              if (task < ntask) {
                const double tt = (task == 0) ? t1 : (task == 1) ? t2 : x;
                d = sdf_at<SHAPE>(tr, sp, qx, qy, tt, dpc);
                ++n_eval;
              }
              if (0 >= base && 0 < base + G) sdf1 = Grp<G>::bcast(d, 0 - base);
              if (1 >= base && 1 < base + G) sdf2 = Grp<G>::bcast(d, 1 - base);
              if (2 >= base && 2 < base + G) f0 = Grp<G>::bcast(d, 2 - base);
            }
          }
          const double g = (sdf2 - sdf1) * 500000;
          if (li == 0) {   // (lane 0 evaluated at t1 = x - 1e-6: its piece interval is the ladder's starting point)
            mine_gs->sgn = (int)(g > 0) - (int)(g < 0);
            mine_gs->prev_x = x;
            mine_gs->piece = dpc.piece; mine_gs->lo = dpc.lo; mine_gs->hi = dpc.hi;
            if (first_pass) mine_gs->fx = f0;
          }
        }
        bool lad = run;   // this group's ladder is open
        int j0 = 1;       // next candidate of every open ladder (wave-uniform)
        for (;;) {
          const unsigned long long am = __ballot(lad && li == 0);
          if (am == 0ull) break;
          const int n_act = __popcll(am);
#ifdef SVSDF_SITE_STATS
          if (lane == 0 && n_act == 64 / G) ++sc[11];   // ladder steps with every group's ladder open
#endif
          // lanes per open ladder (wave-uniform), this group's number among the open ones, and for every lane the
          // ladder it works on (a-th open one) and its place in that ladder's step
          int wd, rank, a, off, src;
          bool serve;
          if (n_act == 64 / G) {   // every ladder open: each group works on its own
            wd = G; rank = grp; a = grp; off = li; src = grp; serve = true;
          } else {
            // floor(64 / n) and floor(lane / wd) through the reciprocal: the quotients stay >= 0.5 / 64 away from the
            // next integer, far more than the reciprocal's error (all n, wd, lane checked: tests/test_elastic_widths.py)
            wd = min(32, (int)(64.5f * __builtin_amdgcn_rcpf((float)n_act)));
            rank = __popcll(am & ((1ull << (lane & ~(G - 1))) - 1ull));
            if (lad && li == 0) srcmap[rank] = (unsigned char)grp;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            a = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)wd));
            off = lane - a * wd;
            serve = a < n_act;
            src = (int)srcmap[serve ? a : 0];
          }
          LadderState *S = gs + src;   // the served group's state
          const double spx = S->px, spy = S->py, sseed = S->seed, sx = S->x, sfx = S->fx;
          const double ssgn = (double)S->sgn;
          PieceCache pcs;
          pcs.piece = S->piece; pcs.lo = S->lo; pcs.hi = S->hi;
          const double stmin = dmax(0.0, sseed - 3.4), stmax = dmin(sseed + 3.4, tr.dur);
          const int j = j0 + off;                    // div
          const double tau = ldexp(0.01, 1 - j);     // alpha halved (div - 1) times: exact
          const double change = -tau * ssgn;
          double xc = sx + change;
          xc = dmax(dmin(xc, stmax), stmin);
          double fc = inf;
          const bool mine = serve && j <= 29;
          SVSDF_SITE(sc, 3, mine);
          if (mine) { fc = sdf_at<SHAPE>(tr, sp, spx, spy, xc, pcs); ++n_eval; }
          const unsigned long long accm = __ballot(mine && (fc - sfx) < 0);
          const unsigned long long wmask = (1ull << wd) - 1ull;
          if (mine) {
            const unsigned b = (unsigned)((accm >> (a * wd)) & wmask);
            if (b != 0u) {
              const int firsts = __ffs(b) - 1;
              if (off == firsts) { S->x = xc; S->fx = fc; }   // the first accepted candidate of that ladder
              if (off > firsts) ++n_spec;   // behind the accepted one: never looked at by the sequential loop
            }
          }
          if (lad) {
            const unsigned bits = (unsigned)((accm >> (rank * wd)) & wmask);
            if (bits != 0u) {
              iter += __ffs(bits);
              lad = false;
            } else {
              const int left = 29 - j0 + 1;
              iter += (left < wd) ? left : wd;
              if (j0 + wd > 29) { lad = false; run = false; }   // no candidate accepted: the descent stops (SWM:1318-1321)
            }
          }
          j0 += wd;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      x_out = mine_gs->x;
      fx_out = mine_gs->fx;
    }
#else
  if (on) {
    // ---- gradientDescent
    constexpr int W = G * U;
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
          SVSDF_SITE(sc, 2, task < ntask);
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
      for (int j0 = 1; j0 <= 29 && !accepted; j0 += W) {
        double xc[U], fc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = j0 + li + G * u;  // div
          const double tau = ldexp(0.01, 1 - j);  // alpha halved (div - 1) times: exact
          const double change = -tau * sgn;
          xc[u] = x + change;
          xc[u] = dmax(dmin(xc[u], t_max), t_min);
          fc[u] = inf;
          SVSDF_SITE(sc, 3, j <= 29);
          if (j <= 29) { fc[u] = sdf_at<SHAPE>(tr, sp, px, py, xc[u], piece); ++n_eval; }
        }
        // first accepted div over the W candidates of this step
        int jacc = 0x7fffffff;
        double xa = 0.0, fa = 0.0;
#pragma unroll
        for (int u = U - 1; u >= 0; --u)
          if ((fc[u] - fx) < 0) { jacc = li + G * u; xa = xc[u]; fa = fc[u]; }
        const int jbest = Grp<G>::min_i(jacc);
        if (jbest != 0x7fffffff) {
          // candidates of this block behind the accepted one: evaluated G at a time, never looked at by the sequential loop
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (j0 + li + G * u <= 29 && li + G * u > jbest) ++n_spec;
          const int src = jbest % G;
          x = Grp<G>::bcast((jacc == jbest) ? xa : 0.0, src);
          fx = Grp<G>::bcast((jacc == jbest) ? fa : 0.0, src);
          iter += jbest + 1;
          accepted = true;
        } else {
          const int left = 29 - j0 + 1;
          iter += (left < W) ? left : W;
        }
      }
      if (!accepted) stop = true;
    }
    x_out = x;
    fx_out = fx;
  }  // on
#endif
  SVSDF_SITE_CYCLES(sc, 10, t_desc0);
}

// ---------------------------------------------------------------------------------------------
// k_solve: getSDFofSweptVolume<false,true> (SWM:844-866) without its FD gradient, for the main
// points of a batch or for the selected GSIP circle samples.
//  * choiceTInit layer 1 (SWM:549-576, first pass): poses at the scan times do not depend on the
//    query point, so they come from the table k_prep built (LDS) and a query only does the rigid
//    transform + shape SDF per sample.  Chunks of 8 samples whose lower bound exceeds the running
//    minimum are skipped: exact, a skipped sample can never be (or tie with) the minimum the
//    reference's strict `<` scan keeps.
//  * layers 2-4 (SWM:557-577) and gradientDescent (SWM:1249-1325): the 21 samples of a layer and
//    the <= 29 candidates of a halving ladder are evaluated G at a time.  A ladder's candidates do
//    not depend on each other, so the first accepted one is exactly the one the sequential loop
//    accepts (bit-identical result, shorter dependent chain).  getSDF_DOT (SWM:799-806) is
//    evaluated once per descent pass: x does not change inside the ladder, so the reference's
//    per-trial re-evaluation returns the same number.
// LDS: [Polygon edges 6 nverts (kPolygonLds only) | pose table 4K | chunks 4*nch | trajectory 20N+1] doubles, then
// ladder_lds_bytes(G) per wave of the block (descent state of its 64 / G groups).
// ---------------------------------------------------------------------------------------------
// (Polygon: 172 VGPRs would mean 2 waves per SIMD for a kernel that waits on its candidate-record loads; asking for 3
// blocks of 4 waves per CU caps it at 168 with two spilled registers: C5 43.3 -> 38.7 ms)
template <int SHAPE, int G, int U>
__global__ void __launch_bounds__(kBlock, is_polygon<SHAPE>() ? 3 : SVSDF_SOLVE_WAVES)
k_solve(const TrajDev *__restrict__ trg, const double *__restrict__ tk, const Pose *__restrict__ pose_g,
        const Chunk *__restrict__ chunks_g, ShapeParams sp, QuerySet qs, double *__restrict__ out_sdf,
        double *__restrict__ out_t, int prune /* bit 0: exact chunk pruning; bit 1: one query per wave */, BatchCtl *__restrict__ ctl, int work_idx, double cull_thresh,
        const double *__restrict__ rot, double slack_max) {
  extern __shared__ double solve_lds[];
  int n;
  const long long total = qs_total(qs, n);
  if (total <= 0 || (long long)blockIdx.x * (((prune & 2) != 0) ? (blockDim.x >> 6) : (blockDim.x / G)) >= total) return;
  const int K = trg->K;
  const int nch = (K + kChunk - 1) / kChunk;
  stage_poly_edges<SHAPE>(sp, solve_lds);
  double *tab_lds = solve_lds + poly_lds_doubles<SHAPE>(sp.nverts);
  Pose *pose = reinterpret_cast<Pose *>(tab_lds);
  Chunk *chunks = reinterpret_cast<Chunk *>(tab_lds + 4 * (size_t)K);
  {
    const double *src = reinterpret_cast<const double *>(pose_g);
    for (int i = threadIdx.x; i < 4 * K; i += blockDim.x) tab_lds[i] = src[i];
    const double *srcc = reinterpret_cast<const double *>(chunks_g);
    for (int i = threadIdx.x; i < 4 * nch; i += blockDim.x) tab_lds[4 * (size_t)K + i] = srcc[i];
  }
  const TrajL tr = stage_traj(trg, tab_lds + 4 * (size_t)K + 4 * (size_t)nch);  // ends with __syncthreads
  // per-wave descent state behind the trajectory (16-byte aligned: the tables before it are whole doubles, rounded up)
  const size_t tables = poly_lds_doubles<SHAPE>(sp.nverts) + 4 * (size_t)K + 4 * (size_t)nch + (size_t)traj_lds_doubles(tr.N);
  char *wave_lds = reinterpret_cast<char *>(solve_lds + ((tables + 1) & ~(size_t)1)) + (threadIdx.x >> 6) * ladder_lds_bytes(G);
  const int li = Grp<G>::li();
  unsigned n_eval = 0, n_scan = 0, n_solved = 0, n_culled = 0, n_spec = 0;
  unsigned long long sc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // SVSDF_SITE_STATS builds only
#ifndef SVSDF_CLOCK_PROBE
#define SVSDF_CLOCK_PROBE 1
#endif
  const bool clk_probe = SVSDF_CLOCK_PROBE && work_idx == 0 && blockIdx.x == 0 && threadIdx.x < 64;   // (wave-uniform; BatchCtl::clk)
  const long long clk_c0 = clk_probe ? clock64() : 0ll, clk_r0 = clk_probe ? wall_clock64() : 0ll;
  // Work distribution: a wave's FIRST 64 / G queries are its own (wave index: no atomic), the following ones come from
  // the launch's cursor.  (All waves of a launch start together: with a fetch first, their 3000 atomics on one address
  // take ~ 12 ns each, one after the other -- the last wave would start ~ 37 us late, in every launch of the chain.)
  // `prune` bit 1 (round 6; launches of a few hundred queries, i.e. the main solve at the reference's own scale): ONE query per
  // wave -- the other lane groups stay empty -- so that every descent is "the wave's last open one" from its first pass on and
  // takes the fused pass (derivative + both signs of the ladder in one step: descend_from_seed) instead of two dependent
  // steps per pass.  The chip has a wave slot for every query there; the launch is a chain of dependent evaluations.
  const bool solo = (prune & 2) != 0;
  prune &= 1;
  const long long per_wave = solo ? 1 : 64 / G;
  const long long n_static = (long long)gridDim.x * (blockDim.x >> 6) * per_wave;
  for (int guard = 0; guard < (1 << 26); ++guard) {
    long long wave_base, gq;
    if (guard == 0) {
      wave_base = (long long)((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * per_wave;
      gq = wave_base + (long long)((threadIdx.x & 63) / G);
    } else {
      gq = fetch_work<G>(&ctl->work[work_idx], wave_base, (int)per_wave) + n_static;
      wave_base += n_static;
    }
    if (wave_base >= total) break;
    double px = 0.0, py = 0.0;
    size_t slot = 0;
    bool live = gq < total && (long long)((threadIdx.x & 63) / G) < per_wave;
    if (live) live = qs_slot(qs, n, gq, slot);
    if (live) {
      px = qs.qx[slot]; py = qs.qy[slot];
      live = (px == px);  // NaN marks an unused slot (whole group)
    }
    // ---- choiceTInit layer 1 over the pose table (or the seed k_round already found for a GSIP sample)
    double best_d = 1e9;
    int best_k = 0x7fffffff;
    bool culled = false;
    const unsigned long long t_scan0 = SVSDF_SITE_CLOCK();
    if (live) {
    if (qs.seed_k) {
      best_k = qs.seed_k[slot];
      best_d = qs.seed_d[slot];
    }
    if (!qs.seed_k || best_k < 0) {   // no seed for this query (main points, cheap-bound samples, unscanned lazy samples)
      scan_layer1<SHAPE, G>(sp, pose, chunks, K, nch, px, py, prune, cull_thresh, best_d, best_k, culled, n_scan, nullptr, -1,
                            sc, rot, slack_max);
    }
    if (culled && li == 0) { out_sdf[slot] = best_d; out_t[slot] = 0.0; ++n_culled; }
    }  // live
    const bool on = live && !culled;
    SVSDF_SITE_CYCLES(sc, 8, t_scan0);
    double x = 0.0, fx = 0.0;
    descend_from_seed<SHAPE, G, U>(tr, tk, sp, px, py, on, best_k, best_d, x, fx, n_eval, n_spec, sc, wave_lds);   // whole wave
    if (on && li == 0) {
      out_sdf[slot] = fx;
      out_t[slot] = x;
      ++n_solved;
    }
  }
  if (clk_probe && threadIdx.x == 0) {
    ctl->clk[0] = (unsigned long long)(clock64() - clk_c0);
    ctl->clk[1] = (unsigned long long)(wall_clock64() - clk_r0);
  }
  unsigned long long te = (unsigned long long)n_eval + n_scan, ts = n_solved, tc = n_scan, tu = n_culled, tp = n_spec;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    te += __shfl_xor(te, m, 64); ts += __shfl_xor(ts, m, 64); tc += __shfl_xor(tc, m, 64); tu += __shfl_xor(tu, m, 64);
    tp += __shfl_xor(tp, m, 64);
  }
  if ((threadIdx.x & 63) == 0 && (te || tu)) {
    StatSlot *ss = stat_slot(ctl->stat);
    atomicAdd(&ss->evals, te); atomicAdd(&ss->solves, ts); atomicAdd(&ss->scan, tc);
    if (tu) atomicAdd(&ss->culled, tu);
    if (tp) atomicAdd(&ss->spec, tp);
  }
#ifdef SVSDF_SITE_STATS
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    unsigned long long v = sc[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&stat_slot(ctl->stat)->pad[i], v);
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// GSIP state (getTrueSDFofSweptVolume SWM:916-1018), one entry per interior point.  Round 4: the per-interior-point
// arrays and the sample arrays are indexed by a COMPACT interior index ia over the whole shard (k_classify hands them
// out from one counter, BatchCtl::n_int of batch 0) and sized by the interior capacity `icap`, not by the point count;
// a batch's active lists hold ia.  An evaluation that finds more interior points than icap drops the surplus, reports
// the count, and the host grows the arrays and repeats it (svsdf_pipeline.hip, evaluate_points).
// ---------------------------------------------------------------------------------------------
#ifndef SVSDF_LAZY_REPS
#define SVSDF_LAZY_REPS 2   // scan passes of the lazy bound mode: the band, then its extension (a third changes nothing)
#endif
enum : int { kPhaseEval = 0, kPhaseSupp = 1, kPhaseNew = 2 };
#ifndef SVSDF_LEAN_HANDOFF
#define SVSDF_LEAN_HANDOFF 0   // 1: only requested GSIP samples get a record (position, angle); 0: every emitted sample.  Measured
                               // round 4 (profiles/r04_handoff_ab.txt): the lean form writes a third of the bytes and is 1.5 % (C3)
                               // to 5 % (NS) SLOWER -- its stores wait for the selection at the end of a point's chain instead of
                               // running under the scans, and HBM is at 2 % of its peak either way.  Kept as a build option.
#endif
// Sample slot of (interior point ia, sample j): slot-major [j * stride + ia] (stride = points in the shard).  A point-major
// layout [ia * 24 + j] -- the samples of a point in one or two cache lines for k_round's lanes -- was measured in round 3
// (tools/traffic_ab.sh, -DSVSDF_POINT_MAJOR): same evaluation time, but MORE HBM traffic (k_solve's reads of the selected
// samples 69 -> 183 MB per C3 evaluation: a solve list walks neighbouring points at similar j), so slot-major stays.
#ifdef SVSDF_POINT_MAJOR
__host__ __device__ __forceinline__ size_t sample_slot(size_t /*stride*/, size_t ia, int j) { return ia * (size_t)kMaxSlots + (size_t)j; }
#else
__host__ __device__ __forceinline__ size_t sample_slot(size_t stride, size_t ia, int j) { return (size_t)j * stride + ia; }
#endif

struct GsipState {
  int *pt;          // index of the (sorted) main point
  double *r;        // current circle radius
  double *theta0;   // first sample angle of the current round
  double *theta_res;
  int *iter;        // 1..9
  int *nsamp;       // samples emitted for the current round
  int *phase;       // kPhaseNew: a round has to be opened; kPhaseEval / kPhaseSupp: samples are out
  unsigned *req;    // samples of the current round requested so far (bit j <-> sample j): the only ones that carry a record
  int *list[2];     // ping-pong compacted lists of still-active interior indices
  int *solve;       // sample slots to solve in the current iteration (capacity kMaxSlots per point)
  // sample slots: sample_slot(stride, batch start + a, j) = [j * stride + batch start + a].  Every emitted sample has its
  // upper bound sq_ub and (scanning modes) seed index sq_k; only REQUESTED samples (gs.req) get a record -- position,
  // angle -- and, from the solve, value and time: 12 B per emitted sample + 40 B per requested one instead of 52 B per
  // emitted one (round 4, SVSDF_LEAN_HANDOFF)
  double *sqx, *sqy, *sqth, *sq_ub, *sq_sdf, *sq_t;
  int *sq_k;        // layer-1 seed index of the sample (scanning modes: sq_ub is then the seed value); -1: none, the solve scans
};

// Per main point after the first solve: exterior -> FD gradient (getGradPrelAtTimeStamp,
// SWM:779-788) and done; interior -> GSIP init (SWM:926-963), first round opened by k_round.
template <int SHAPE>
__global__ void __launch_bounds__(kBlock)
k_classify(const TrajDev *__restrict__ trg, ShapeParams sp, const double *__restrict__ px_,
           const double *__restrict__ py_, const double *__restrict__ sdf_,
           const double *__restrict__ t_, double *__restrict__ res_sdf,
           double *__restrict__ res_t, double *__restrict__ res_gx, double *__restrict__ res_gy,
           GsipState gs, BatchCtl *__restrict__ ctl, int *__restrict__ n_int, int icap) {
  extern __shared__ double classify_lds[];
  const TrajL tr = stage_traj(trg, classify_lds);
  const int start = ctl->start, count = ctl->count;
  const int lane = (int)(threadIdx.x & 63);
  // (wave-uniform trip count: the interior lanes of a wave take their indices together, below)
  for (int e0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) & ~63u); e0 < count; e0 += gridDim.x * blockDim.x) {
    const int e = e0 + lane;
    const bool in_range = e < count;
    const int i = start + (in_range ? e : 0);
    const double px = px_[i], py = py_[i];
    const double sdf = sdf_[i], ts = t_[i];
    const bool inter = in_range && !(sdf > 0);
    double vx = 0.0, vy = 0.0, w = 0.0;
    if (in_range && sdf > 0) {  // outside case (SWM:921-924)
      int piece = 0;
      const Pose p = pose_at(tr, ts, piece);
      const double dx = px - p.x, dy = py - p.y;
      const double rx = p.cs * dx + p.sn * dy;
      const double ry = (-p.sn) * dx + p.cs * dy;
      double gx, gy;
      shape_grad<SHAPE>(sp, rx, ry, gx, gy);
      res_sdf[i] = sdf; res_t[i] = ts; res_gx[i] = gx; res_gy[i] = gy;
    } else if (inter) {
      // interior: velocity at t* with the low-speed rescans (SWM:929-954)
      double sl;
      int piece = locate_local(tr, ts, 0, sl);
      piece_vel(tr.c + piece * 18, sl, vx, vy, w);
      if (sqrt(vx * vx + vy * vy + w * w) < 0.01) {
        if (ts < 0.1) {
          for (double t_scan = ts; t_scan <= tr.dur; t_scan += 0.1) {
            piece = locate_local(tr, t_scan, piece, sl);
            piece_vel(tr.c + piece * 18, sl, vx, vy, w);
            if (sqrt(vx * vx + vy * vy + w * w) >= 0.01) break;
          }
        } else if (ts > tr.dur - 0.1) {
          for (double t_scan = ts; t_scan >= 0; t_scan -= 0.1) {
            piece = locate_local(tr, t_scan, piece, sl);
            piece_vel(tr.c + piece * 18, sl, vx, vy, w);
            if (sqrt(vx * vx + vy * vy + w * w) >= 0.01) break;
          }
        }
      }
    }
    // compact interior indices, one block of consecutive ones per wave (one atomic for the wave instead of one per
    // point; neighbouring points keep neighbouring entries in the interior-sized arrays)
    const unsigned long long mi = __ballot(inter);
    if (mi == 0ull) continue;   // wave-uniform
    const int leader = __ffsll((long long)mi) - 1;
    int base_i = 0;
    if (lane == leader) base_i = atomicAdd(n_int, __popcll(mi));
    base_i = __shfl(base_i, leader, 64);
    const int ia_ = base_i + __popcll(mi & ((1ull << lane) - 1ull));
    const bool kept = inter && ia_ < icap;
    if (inter && !kept) {   // no room: dropped (reads as inactive); the host sees n_int > icap, grows the arrays and repeats
      res_sdf[i] = 1e300; res_t[i] = ts; res_gx[i] = 0.0; res_gy[i] = 0.0;
    }
    const unsigned long long mk = __ballot(kept);
    if (mk == 0ull) continue;
    const int leader2 = __ffsll((long long)mk) - 1;
    int base_a = 0;
    if (lane == leader2) base_a = atomicAdd(&ctl->n_active[0], __popcll(mk));
    base_a = __shfl(base_a, leader2, 64);
    if (kept) {
      const int a = base_a + __popcll(mk & ((1ull << lane) - 1ull));
      const size_t ia = (size_t)ia_;
      // SampleSet2D::initSet (SWM:73-103)
      double theta0 = atan2(vx, -vy);
      if (theta0 < 0) theta0 += 2 * kPI;
      gs.pt[ia] = i;
      gs.r[ia] = 10;               // r0 (SWM:927)
      gs.theta0[ia] = theta0;
      gs.theta_res[ia] = kPI + 0.1;
      gs.iter[ia] = 1;
      gs.nsamp[ia] = 0;
      gs.phase[ia] = kPhaseNew;
      gs.list[0][start + a] = ia_;
      res_t[i] = ts;  // real_t_star fallback
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_round: 32 lanes per still-active interior point, lane j <-> circle sample j (<= 22 per round).
//
// Closing a round (SWM:965-1009; final assembly SWM:1011-1017): max_g, its t* and angle come
// from the arg-max sample only.  Upper-bound selection (exact): every sample's solved value is
// bounded above by ANY sdf value along its own scan -- layers 2-4 only lower min_dis, the descent
// starts at f(time_seed) == min_dis and only accepts strict decreases -- so with ub_j = the
// smallest of the 8 table-pose values of the chunk nearest to the sample, a sample with
// ub_j < g* (g* = best solved value so far) can neither be nor tie with the round's maximum and
// is never solved.  Opening a round requests the samples within `delta` of the best bound;
// closing it first requests whatever of {ub_j >= g*} is still unsolved (supplementary iteration,
// rarely non-empty).  The reference solves every sample; only the arg-max one is used.
//
// Opening a round: SampleSet2D::getElements / getElementPos (SWM:36-39, 60-71; one ring rk = 1)
// with theta_j = theta0 + j * theta_res by repeated addition; expandSet(2, theta*) (SWM:105-110).
// ---------------------------------------------------------------------------------------------
// One GSIP step of ONE interior point (index a of its batch) by its LP lanes: close the round whose samples were solved
// (or request supplementary solves), write the result when the point is finished, or open the next round (samples,
// bounds, selection).  Outputs: which of this lane's samples are to be solved (list_me / mlist: per pass, the mask over
// the point's lanes), whether the point stays active (push_next), samples emitted (n_emit), finished.  Called by k_round
// (one launch per GSIP iteration).
template <int NP>
struct RoundOut {
  bool list_me[NP];
  unsigned mlist[NP];
  int n_emit;
  bool push_next, finished;
};
template <int SHAPE, int LP, int MODE, bool EAGER = false>
__device__ __forceinline__ void round_point(const ShapeParams &sp, const Pose *pose, const Chunk *chunks, int K, int nch,
                                            const double *__restrict__ px_, const double *__restrict__ py_,
                                            const GsipState &gs, size_t stride, int start, int a, double delta,
                                            double band_delta, double *__restrict__ res_sdf, double *__restrict__ res_t,
                                            double *__restrict__ res_gx, double *__restrict__ res_gy, unsigned &n_scan,
                                            RoundOut<(kMaxSlots + LP - 1) / LP> &out, unsigned short *clist, int clist_on,
                                            unsigned long long (&rc)[16], int duo = -1, const ChunkAnchor *__restrict__ anch_ = nullptr) {
  // (the lazy mode scans few samples, and picks itself where the circle bounds are already good -- star: the anchored walk is
  // compiled into the full and anchor modes only; in the lazy kernels its registers cost a wave per SIMD for nothing)
#ifdef SVSDF_NO_SCAN_ANCHORS
  const ChunkAnchor *__restrict__ anch = nullptr; (void)anch_;
#else
#ifdef SVSDF_ANCHORS_NOT_LP8_M3
  const ChunkAnchor *__restrict__ anch = (MODE == 1 || (MODE == 3 && LP == 32)) ? anch_ : nullptr;
#else
  // (and not for the Polygon: its SDF is several times an analytic shape's code, and a fourth inlined copy of it -- the anchor
  // site -- cost C5 8 % with the walk switched off and 12 % with it on: 17.7 -> 19.1 / 19.9 ms)
  const ChunkAnchor *__restrict__ anch = ((MODE == 1 || MODE == 3) && !is_polygon<SHAPE>()) ? anch_ : nullptr;
#endif
#endif
  // duo (k_tail with ONE point per wave, round 6): -1 off; 0 / 1: BOTH half-waves of the wave run this function for the SAME
  // point (identical state, identical decisions, identical stores) and share its seed scans -- half h takes the scan passes
  // 2 q + h -- then exchange the results across the halves.  The second half-wave of such a wave had nothing to do; a
  // reference-scale callback is a chain of dependent steps and a round's 18 - 21 scans were 6 passes of it, now 3.
  constexpr bool FULL = MODE == 1;
  unsigned long long tph = SVSDF_SITE_CLOCK();
  constexpr int NP = (kMaxSlots + LP - 1) / LP;  // sample passes per point
  const int l = (int)(threadIdx.x & (LP - 1));
  const unsigned lt_mask = (1u << l) - 1u;
  auto ballot_g = [&](bool p) -> unsigned {   // bit i <=> lane i of this point's lane group
    const unsigned long long m = __ballot(p);
    const int base = (int)(threadIdx.x & 63) & ~(LP - 1);
    return (unsigned)((m >> base) & ((LP == 32) ? 0xffffffffull : 0xffull));
  };
  bool (&list_me)[NP] = out.list_me;
  unsigned (&mlist)[NP] = out.mlist;
  bool &push_next = out.push_next;
  int &n_emit = out.n_emit;
  int i = 0;
  size_t ia = 0;
  bool open = false;
  double cx = 0.0, cy = 0.0, r = 0.0, theta0 = 0.0, theta_res = 0.0;
    ia = (size_t)a;   // (a: compact interior index of the point; `start` is the batch's offset in the point arrays)
    // Two memory round trips instead of five (round 5): the point's whole state in one go, then -- before anything is
    // decided -- the point's coordinates and THIS LANE'S samples of the round that may have to be closed (solved value,
    // time, angle, bound).  The old order (state -> phase -> sample count -> solved values -> arg-max -> its time and
    // angle -> the bounds of the unsolved ones) was a chain of dependent loads, each a trip to L2 / HBM: a fifth of
    // k_round's wave cycles in the lazy mode, ~ 40 us of a reference-scale callback.  Same values, same decisions.
    i = gs.pt[ia];
    r = gs.r[ia]; theta0 = gs.theta0[ia]; theta_res = gs.theta_res[ia];
    const int phase_ = gs.phase[ia];
    const int n_pre = gs.nsamp[ia];
    const unsigned req_pre = gs.req[ia];
    const int iter_pre = gs.iter[ia];
    cx = px_[i]; cy = py_[i];
    const double rest_pre = res_t[i];
    // (the sample prefetch only with 32 lanes per point -- one sample per lane: four registers; with 8 lanes and three
    // samples per lane it costs 25 VGPRs and a wave per SIMD in exactly the launches that hold every interior point)
    constexpr bool PRE = !SVSDF_LEAN_HANDOFF && LP == 32;
    double g_pre[NP], t_pre[NP], th_pre[NP], ub_pre[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int j = l + LP * ps;
      const size_t s_ = sample_slot(stride, ia, (j < kMaxSlots) ? j : 0);
      // EAGER (k_tail: the state sits in the wave's LDS or the callback is a latency chain): all 24 slots at once, before the
      // state has arrived.  k_round (round 6): only the samples that exist and only when a round is to be closed -- one more
      // dependent trip than the eager form, but the eager form read 4 x 24 sample entries for every point of every launch,
      // also the points whose first round is still to be opened: k_round FETCH 170 -> 398 MB per C3 evaluation
      const bool ld = PRE && (EAGER ? j < kMaxSlots : (phase_ != kPhaseNew && j < n_pre));
      g_pre[ps] = ld ? gs.sq_sdf[s_] : kUnsolved;
      ub_pre[ps] = ld ? gs.sq_ub[s_] : 0.0;
      // time and angle are only ever used of ONE sample, the arg-max: k_round fetches those two numbers once it is known (a
      // dependent trip, two values) instead of two more arrays for every sample of every point (round 6; HBM traffic)
      t_pre[ps] = (EAGER && ld) ? gs.sq_t[s_] : 0.0;
      th_pre[ps] = (EAGER && ld) ? gs.sqth[s_] : 0.0;
    }
    open = phase_ == kPhaseNew;
    if (!open) {
      // ---- close the round: max over the solved samples, first index wins ties (strict >)
      const int n = n_pre;
      const unsigned req = req_pre;   // samples requested so far: the solved ones
      double g_mine[NP];
      double g = kUnsolved;
      int idx = 0x7fffffff;
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int j = l + LP * ps;
        if constexpr (PRE) g_mine[ps] = (j < n) ? g_pre[ps] : kUnsolved;
        else g_mine[ps] = (j < n && (!SVSDF_LEAN_HANDOFF || ((req >> j) & 1u))) ? gs.sq_sdf[sample_slot(stride, ia, j)] : kUnsolved;
        if (g_mine[ps] > g || (g_mine[ps] == g && j < idx)) { g = g_mine[ps]; idx = j; }
      }
      {  // lexicographic (max g, min index) over the LP lanes: butterfly through DPP (Grp<LP>::xchg)
        auto step = [&](double og, int oi) { if (og > g || (og == g && oi < idx)) { g = og; idx = oi; } };
        step(Grp<LP>::template xchg<0>(g), Grp<LP>::template xchg<0>(idx));
        step(Grp<LP>::template xchg<1>(g), Grp<LP>::template xchg<1>(idx));
        step(Grp<LP>::template xchg<2>(g), Grp<LP>::template xchg<2>(idx));
        if constexpr (LP == 32) {
          step(Grp<LP>::template xchg<3>(g), Grp<LP>::template xchg<3>(idx));
          step(Grp<LP>::template xchg<4>(g), Grp<LP>::template xchg<4>(idx));
        }
      }
      double max_g = -100000, real_t = rest_pre, star_th = 0.0;
      if (g > max_g) {
        max_g = g;
        if constexpr (!PRE || !EAGER) {
          const size_t sb = sample_slot(stride, ia, idx);
          real_t = gs.sq_t[sb]; star_th = gs.sqth[sb];
        } else {
          // the arg-max sample's time and angle: from the lane that holds it (pass idx / LP of lane idx % LP)
          double tsel = t_pre[0], thsel = th_pre[0];
#pragma unroll
          for (int ps = 1; ps < NP; ++ps)
            if (idx / LP == ps) { tsel = t_pre[ps]; thsel = th_pre[ps]; }
          real_t = __shfl(tsel, idx % LP, LP);
          star_th = __shfl(thsel, idx % LP, LP);
        }
      }
      // unsolved samples that could still reach max_g -> supplementary solves
      bool any = false;
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int j = l + LP * ps;
        if constexpr (PRE) list_me[ps] = (j < n) && (g_mine[ps] == kUnsolved) && ub_pre[ps] >= max_g;
        else list_me[ps] = (j < n) && (SVSDF_LEAN_HANDOFF ? !((req >> j) & 1u) : (g_mine[ps] == kUnsolved)) && gs.sq_ub[sample_slot(stride, ia, j)] >= max_g;
        mlist[ps] = ballot_g(list_me[ps]);
        any = any || (mlist[ps] != 0u);
      }
      if (any && SVSDF_LEAN_HANDOFF) {
        // (rare) the newly requested samples get their record now: position and angle recomputed exactly as the opening
        // of the round computed them (same operations on the same operands); no scan seed -- the solve scans itself
        double theta = theta0;
        for (int q = 0; q < l; ++q) theta += theta_res;
        unsigned nreq = req;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
          if (list_me[ps]) {
            const size_t s = sample_slot(stride, ia, l + LP * ps);
            gs.sqx[s] = cx + 1.0 * r * cos(theta);
            gs.sqy[s] = cy + 1.0 * r * sin(theta);
            gs.sqth[s] = theta;
          }
          nreq |= mlist[ps] << (LP * ps);
          if (ps + 1 < NP) {
#pragma unroll
            for (int q = 0; q < LP; ++q) theta += theta_res;
          }
        }
        if (l == 0) gs.req[ia] = nreq;
      }
      if (any) {
        push_next = true;
        if (l == 0) gs.phase[ia] = (int)kPhaseSupp;
      } else {
        const double r_star = r - max_g;
        const int iter = iter_pre;
        if (iter > 8 || fabs(max_g) < 0.1) {
          if (l == 0) {
            const double corx = cx + 1.0 * r_star * cos(star_th);
            const double cory = cy + 1.0 * r_star * sin(star_th);
            double gx = corx - cx, gy = cory - cy;
            const double z = gx * gx + gy * gy;
            if (z > 0.0) { const double nn = sqrt(z); gx = gx / nn; gy = gy / nn; }
            res_sdf[i] = -r_star; res_t[i] = real_t; res_gx[i] = gx; res_gy[i] = gy;
          }
          out.finished = true;
        } else {
          // expandSet(2, theta*)
          theta_res = theta_res / (2 + 1);
          theta_res = dmax(0.3, theta_res);
          r = r_star;
          theta0 = star_th;
          if (l == 0) {
            gs.r[ia] = r; gs.theta_res[ia] = theta_res; gs.theta0[ia] = theta0;
            gs.iter[ia] = iter + 1;
            res_t[i] = real_t;
          }
          open = true;
        }
      }
    }
    SVSDF_PHASE(rc, 0, tph);
    if (open) {
      // ---- candidate chunks of this round: every sample y of the circle |y - p| = r has, for every table pose k of a
      // chunk c,  |y - c| - rb_c <= sdf_k(y) <= |y - c| + rb_c  (rb_c = chunk radius + shape bound R: the shape has a
      // point within R of the body origin), and |p - c| - |r| <= |y - c| <= |p - c| + |r|.  So every sample's table minimum is
      // <= U = min_c (|p - c| + rb_c) + r, and a chunk with |p - c| - r - rb_c > U holds no table pose that can be, or tie
      // with, ANY sample's minimum -- nor can it be the chunk with the centre nearest to a sample (|p - c| - r > min |p - c'|
      // + r follows).  The scans and the cheap bound below walk this list (ascending, in LDS) instead of all chunks:
      // same seeds, same bounds; most of a long path's chunks drop out once the circle is small (rounds >= 3).
      int ncl = -1;
      // (round 6) a circle of radius >= 8 m -- the first round's r0 = 10 (SWM:927), i.e. EVERY point of the chain's first
      // launch -- reaches most of a path: the list then holds (nearly) all chunks or overflows (ncl = -1) and the scans walk
      // everything anyway, but building it cost two passes over all chunks with a correctly rounded root each, a third of that
      // launch's non-scan instructions.  Any list is exact (ncl = -1: all chunks), so this only moves time.
      if (fabs(r) < 8.0) {
        double u = 1e300;
        for (int c = l; c < nch; c += LP) {
          const Chunk ch = chunks[c];
          u = dmin(u, norm2(cx - ch.cx, cy - ch.cy) + ch.rb);
        }
        u = dmin(u, Grp<LP>::template xchg<0>(u));
        u = dmin(u, Grp<LP>::template xchg<1>(u));
        u = dmin(u, Grp<LP>::template xchg<2>(u));
        if constexpr (LP == 32) {
          u = dmin(u, Grp<LP>::template xchg<3>(u));
          u = dmin(u, Grp<LP>::template xchg<4>(u));
        }
        // (the reference's radius update r <- r - max_g can turn r NEGATIVE when a sample's local argmin search ended in a
        // far basin, max_g > r: the circle then has radius |r|)
        const double ra = fabs(r);
        const double thr = (u + ra) * (1.0 + 1e-12) + 1e-9;
        int cnt = 0;
        for (int cb = 0; cb < nch; cb += LP) {
          const int c = cb + l;
          bool keep = false;
          if (c < nch) {
            const Chunk ch = chunks[c];
            keep = (norm2(cx - ch.cx, cy - ch.cy) - ra) - ch.rb <= thr;
          }
          const unsigned mk = ballot_g(keep);
          const int pos = cnt + __popc(mk & lt_mask);
          if (keep && pos < kMaxCand) clist[pos] = (unsigned short)c;
          cnt += __popc(mk);
        }
        if (cnt <= kMaxCand && nch <= 65535 && clist_on) ncl = cnt;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the point's lanes (one wave) read each other's entries
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      SVSDF_PHASE(rc, 1, tph);
      // ---- open a round: lane l takes samples l, l + LP, ...
      double theta = theta0;
      for (int q = 0; q < l; ++q) theta += theta_res;
      double ub[NP], umax = -1e300;
      double sqx_l[NP], sqy_l[NP], th_l[NP];
      int kk[NP];
      bool valid[NP];
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int j = l + LP * ps;
        valid[ps] = (theta < theta0 + 2 * kPI) && (j < kMaxSlots);
        n_emit += __popc(ballot_g(valid[ps]));  // theta increases with j: the valid samples are a prefix
        ub[ps] = -1e300;
        kk[ps] = 0;
        sqx_l[ps] = 0.0; sqy_l[ps] = 0.0; th_l[ps] = 0.0;
        if (valid[ps]) {
          const size_t s = sample_slot(stride, ia, j);
          // (sincos_exact -- one argument reduction for the pair -- was measured here in round 6: same bits, fewer instructions,
          // but 18 more live registers in the three unrolled sample passes of the 8-lane kernels: 128 -> 146 VGPRs, a wave per
          // SIMD less; the two library calls stay)
          const double qx = cx + 1.0 * r * cos(theta);
          const double qy = cy + 1.0 * r * sin(theta);
          sqx_l[ps] = qx; sqy_l[ps] = qy;
          if constexpr (MODE == 0 || MODE == 2) {
            // cheap upper bound: best table pose of the chunk with the nearest centre (any chunk is valid)
            double d2min = 1e300;
            int c0 = 0;
            const int ncl_c = (clist_on & 2) ? ncl : -1;
            const int nl_c = (ncl_c < 0) ? nch : ncl_c;
            for (int jc = 0; jc < nl_c; ++jc) {
              const int c = (ncl_c < 0) ? jc : (int)clist[jc];
              const Chunk ch = chunks[c];
              const double ex = qx - ch.cx, ey = qy - ch.cy;
              const double d2 = ex * ex + ey * ey;
              if (d2 < d2min) { d2min = d2; c0 = c; }
            }
            double u = 1e300;
            const int k1 = (c0 * kChunk + kChunk < K) ? c0 * kChunk + kChunk : K;
            for (int k = c0 * kChunk; k < k1; ++k) u = dmin(u, sdf_from_pose<SHAPE>(sp, pose[k], qx, qy));
            ub[ps] = u;
            gs.sq_ub[s] = u;
          }
          th_l[ps] = theta;
#if !SVSDF_LEAN_HANDOFF
          gs.sqx[s] = qx; gs.sqy[s] = qy; gs.sqth[s] = theta; gs.sq_sdf[s] = kUnsolved;
#endif
        }
        if (ps + 1 < NP) {
#pragma unroll
          for (int q = 0; q < LP; ++q) theta += theta_res;
        }
      }
      SVSDF_PHASE(rc, 2, tph);
      if constexpr (MODE == 3) {
        // Anchor scans (round 4; MODE 3 = the full-scan mode with this first stage).  The table minimum ub(q) = min_k sdf_k(q) is 1-Lipschitz in q (a minimum of exact
        // distance functions), so a scanned sample a bounds its neighbours:  g_j <= ub(q_j) <= ub(q_a) + |q_j - q_a|.
        // Every third sample of a round with more than 6 is scanned first (the anchors); a sample whose bound from its
        // neighbour anchors stays more than the selection band below the best anchor cannot be requested now and is not
        // scanned at all -- it keeps the Lipschitz bound (a valid upper bound: closing the round still requests it if
        // the solved values fall that low; its solve then scans itself, sq_k = -1).  The others are scanned in a second
        // pass, and so on until no unscanned sample reaches the band.  On a circle of radius r the bound is loose by
        // 0.3 r .. 0.6 r: the samples on the far side of the circle from the round's maximum drop out (C3: 5.5 M ->
        // about half the scans).  Same requests for the samples that matter, same results (any selection is exact).
        constexpr int SG = LP / 8;
        const int sg = l >> 3;
        const int pos3 = (l & 7) % 3;                       // position between anchors inside the 8-lane run of a pass
        bool scanned[NP], pend[NP];
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) { scanned[ps] = false; kk[ps] = -1; pend[ps] = valid[ps] && (n_emit <= 6 || pos3 == 0); }
        double u2 = -1e300;   // best scanned bound so far
        for (int rep = 0; rep < 8; ++rep) {
          unsigned mp[NP];
          int nb = 0;
#pragma unroll
          for (int ps = 0; ps < NP; ++ps) {
            mp[ps] = ballot_g(pend[ps]);
            nb += __popc(mp[ps]);
          }
          // rank of this lane's sample of pass ps among the pending ones (recomputed where it is used: three registers less
          // across the scans in the kernels with three sample passes per lane)
          auto myrank = [&](int ps) -> int {
            int rk_ = __popc(mp[ps] & lt_mask);
#pragma unroll
            for (int q = 0; q < NP; ++q) if (q < ps) rk_ += __popc(mp[q]);
            return rk_;
          };
          if (nb == 0) break;
          for (int pq = 0; (duo >= 0 ? 2 * pq : pq) * SG < nb; ++pq) {
            const int p = duo >= 0 ? 2 * pq + duo : pq;
            const int r = p * SG + sg;          // rank (among the pending samples) this 8-lane sub-group scans
            int rr = r, sps = 0, sl = 0;
            bool found = false;
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
              const int c = __popc(mp[ps]);
              if (!found && r < nb && rr < c) {
                unsigned m = mp[ps];
                for (int q = 0; q < rr; ++q) m &= m - 1u;
                sl = __ffs(m) - 1;
                sps = ps;
                found = true;
              } else if (!found) {
                rr -= c;
              }
            }
            double sxs = sqx_l[0], sys = sqy_l[0];
#pragma unroll
            for (int ps = 1; ps < NP; ++ps)
              if (sps == ps) { sxs = sqx_l[ps]; sys = sqy_l[ps]; }
            const double qx = __shfl(sxs, sl, LP), qy = __shfl(sys, sl, LP);
            double bd = -1e300;
            int bk = 0;
            if (found) {
              bool cu;
              scan_layer1<SHAPE, 8, true>(sp, pose, chunks, K, nch, qx, qy, 1, __longlong_as_double(0x7ff0000000000000ll), bd, bk, cu, n_scan, clist, (clist_on & 1) ? ncl : -1, rc + 8, nullptr, 0.0, anch);
            }
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
              const bool mine = pend[ps] && (myrank(ps) / SG == p);
              const double rb = __shfl(bd, (myrank(ps) % SG) * 8, LP);
              const int rk = __shfl(bk, (myrank(ps) % SG) * 8, LP);
              if (mine) { ub[ps] = rb; kk[ps] = rk; scanned[ps] = true; }
            }
          }
          if (duo >= 0) {   // results of the passes the other half-wave ran (same lane there owns the same sample)
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
              const double ou = __shfl_xor(ub[ps], 32, 64);
              const int ok = __shfl_xor(kk[ps], 32, 64);
              if (pend[ps] && ((myrank(ps) / SG) & 1) != duo) { ub[ps] = ou; kk[ps] = ok; scanned[ps] = true; }
            }
          }
          if (rep == 0 && n_emit > 6) {
            // bounds of the samples between the anchors: left anchor (always there), right anchor (same 8-lane run)
            const int la = l - pos3, ra = la + 3;
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
              const double ul = __shfl(ub[ps], la, LP), xl = __shfl(sqx_l[ps], la, LP), yl = __shfl(sqy_l[ps], la, LP);
              const int rsrc = ((ra & 7) > (l & 7) && ra < LP) ? ra : la;   // (no right anchor in this run: the left one again)
              const double ur = __shfl(ub[ps], rsrc, LP), xr = __shfl(sqx_l[ps], rsrc, LP), yr = __shfl(sqy_l[ps], rsrc, LP);
              const bool rv = __shfl((int)(valid[ps] && scanned[ps]), rsrc, LP) != 0;
              if (valid[ps] && !scanned[ps]) {
                double b = ul + norm2(sqx_l[ps] - xl, sqy_l[ps] - yl);
                if (rv) b = dmin(b, ur + norm2(sqx_l[ps] - xr, sqy_l[ps] - yr));
                ub[ps] = b + 1e-9;
              }
            }
          }
#pragma unroll
          for (int ps = 0; ps < NP; ++ps) u2 = scanned[ps] ? fmax(u2, ub[ps]) : u2;
          u2 = fmax(u2, Grp<LP>::template xchg<0>(u2));
          u2 = fmax(u2, Grp<LP>::template xchg<1>(u2));
          u2 = fmax(u2, Grp<LP>::template xchg<2>(u2));
          if constexpr (LP == 32) {
            u2 = fmax(u2, Grp<LP>::template xchg<3>(u2));
            u2 = fmax(u2, Grp<LP>::template xchg<4>(u2));
          }
#pragma unroll
          for (int ps = 0; ps < NP; ++ps) pend[ps] = valid[ps] && !scanned[ps] && ub[ps] >= u2 - delta;
        }
#pragma unroll
        for (int ps = 0; ps < NP; ++ps)
          if (valid[ps]) {
            const size_t s = sample_slot(stride, ia, l + LP * ps);
            gs.sq_ub[s] = ub[ps];
            gs.sq_k[s] = kk[ps];
            if (!scanned[ps]) ub[ps] = -1e300;   // only scanned samples take part in the selection below
          }
      } else if constexpr (MODE == 1) {
        // Tightest bound layer 1 can give: the sample's own seed (the full pruned scan its solve would start
        // with), found here by 8 cooperating lanes per sample -- LP / 8 samples at a time -- and handed to
        // k_solve, which then skips its scan.  For shapes / trajectories where the nearest chunk is a poor
        // guess (sdHorseshoe: 5.3 -> 2 solves per point) this is the difference between solving most samples
        // and solving the one that matters.
        constexpr int SG = LP / 8;
        const int sg = l >> 3;
        for (int pq = 0; (duo >= 0 ? 2 * pq : pq) * SG < n_emit; ++pq) {
          const int p = duo >= 0 ? 2 * pq + duo : pq;
          const int sidx = p * SG + sg;             // sample this 8-lane sub-group scans in this pass
          const int slot = sidx / LP;               // uniform over the point's lanes (SG == 1 when NP > 1)
          double sxs = sqx_l[0], sys = sqy_l[0];
#pragma unroll
          for (int ps = 1; ps < NP; ++ps)
            if (slot == ps) { sxs = sqx_l[ps]; sys = sqy_l[ps]; }
          const double qx = __shfl(sxs, sidx % LP, LP), qy = __shfl(sys, sidx % LP, LP);
          double bd = -1e300;
          int bk = 0;
          if (sidx < n_emit) {
            bool cu;
            scan_layer1<SHAPE, 8, true>(sp, pose, chunks, K, nch, qx, qy, 1, __longlong_as_double(0x7ff0000000000000ll), bd, bk, cu, n_scan, clist, (clist_on & 1) ? ncl : -1, rc + 8, nullptr, 0.0, anch);
          }
          // hand the result to the lane that owns the sample: sub-group (j % SG) scanned sample j in pass j / SG
          const double rb = __shfl(bd, (l % SG) * 8, LP);
          const int rk = __shfl(bk, (l % SG) * 8, LP);
#pragma unroll
          for (int ps = 0; ps < NP; ++ps)
            if (valid[ps] && (l + LP * ps) / SG == p) { ub[ps] = rb; kk[ps] = rk; }
        }
        if (duo >= 0) {   // the other half-wave scanned the passes of the other parity: lane l there owns the same sample
#pragma unroll
          for (int ps = 0; ps < NP; ++ps) {
            const double ou = __shfl_xor(ub[ps], 32, 64);
            const int ok = __shfl_xor(kk[ps], 32, 64);
            if (valid[ps] && (((l + LP * ps) / SG) & 1) != duo) { ub[ps] = ou; kk[ps] = ok; }
          }
        }
#pragma unroll
        for (int ps = 0; ps < NP; ++ps)
          if (valid[ps]) {
            const size_t s = sample_slot(stride, ia, l + LP * ps);
            gs.sq_ub[s] = ub[ps];
            gs.sq_k[s] = kk[ps];   // (every emitted sample keeps its seed: a supplementary solve starts from it)
          }
      }
      if constexpr (MODE == 2) {
        // Lazy scans: the cheap bounds single out the samples the cheap mode would solve (within band_delta of the
        // best one); those get their own pruned table scan (8 lanes per sample, LP / 8 at a time) and only the best
        // of THEM by the scanned bound are requested; the seeds go to k_solve (sq_k >= 0), unscanned samples keep
        // their cheap bound (sq_k = -1: a supplementary solve scans itself).  Any selection is exact: closing the
        // round requests whatever unsolved sample's bound still reaches the best solved value.
        constexpr int SG = LP / 8;
        const int sg = l >> 3;
        double uc = -1e300;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) uc = fmax(uc, ub[ps]);
        uc = fmax(uc, Grp<LP>::template xchg<0>(uc));
        uc = fmax(uc, Grp<LP>::template xchg<1>(uc));
        uc = fmax(uc, Grp<LP>::template xchg<2>(uc));
        if constexpr (LP == 32) {
          uc = fmax(uc, Grp<LP>::template xchg<3>(uc));
          uc = fmax(uc, Grp<LP>::template xchg<4>(uc));
        }
        bool inband[NP], scanned[NP];
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) { inband[ps] = valid[ps] && ub[ps] >= uc - band_delta; scanned[ps] = false; kk[ps] = -1; }
        double u2 = -1e300;   // best scanned bound so far
        for (int rep = 0; rep < SVSDF_LAZY_REPS; ++rep) {
          unsigned mp[NP];
          int nb = 0;
#pragma unroll
          for (int ps = 0; ps < NP; ++ps) {
            mp[ps] = ballot_g(inband[ps] && !scanned[ps]);
            nb += __popc(mp[ps]);
          }
          auto myrank = [&](int ps) -> int {   // (see the anchor mode above)
            int rk_ = __popc(mp[ps] & lt_mask);
#pragma unroll
            for (int q = 0; q < NP; ++q) if (q < ps) rk_ += __popc(mp[q]);
            return rk_;
          };
          if (nb == 0) break;
          for (int pq = 0; (duo >= 0 ? 2 * pq : pq) * SG < nb; ++pq) {
            const int p = duo >= 0 ? 2 * pq + duo : pq;
            const int r = p * SG + sg;          // rank (among the pending samples) this 8-lane sub-group scans
            int rr = r, sps = 0, sl = 0;
            bool found = false;
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
              const int c = __popc(mp[ps]);
              if (!found && r < nb && rr < c) {
                unsigned m = mp[ps];
                for (int q = 0; q < rr; ++q) m &= m - 1u;
                sl = __ffs(m) - 1;
                sps = ps;
                found = true;
              } else if (!found) {
                rr -= c;
              }
            }
            double sxs = sqx_l[0], sys = sqy_l[0];
#pragma unroll
            for (int ps = 1; ps < NP; ++ps)
              if (sps == ps) { sxs = sqx_l[ps]; sys = sqy_l[ps]; }
            const double qx = __shfl(sxs, sl, LP), qy = __shfl(sys, sl, LP);
            double bd = -1e300;
            int bk = 0;
            if (found) {
              bool cu;
              scan_layer1<SHAPE, 8, true>(sp, pose, chunks, K, nch, qx, qy, 1, __longlong_as_double(0x7ff0000000000000ll), bd, bk, cu, n_scan, clist, (clist_on & 1) ? ncl : -1, rc + 8, nullptr, 0.0, anch);
            }
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
              const bool mine = inband[ps] && !scanned[ps] && (myrank(ps) / SG == p);
              const double rb = __shfl(bd, (myrank(ps) % SG) * 8, LP);
              const int rk = __shfl(bk, (myrank(ps) % SG) * 8, LP);
              if (mine) { ub[ps] = rb; kk[ps] = rk; scanned[ps] = true; }
            }
          }
          if (duo >= 0) {   // results of the passes the other half-wave ran (same lane there owns the same sample)
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
              const double ou = __shfl_xor(ub[ps], 32, 64);
              const int ok = __shfl_xor(kk[ps], 32, 64);
              if (inband[ps] && !scanned[ps] && ((myrank(ps) / SG) & 1) != duo) { ub[ps] = ou; kk[ps] = ok; scanned[ps] = true; }
            }
          }
          // extend the band to unscanned samples whose cheap bound still reaches the best scanned one
#pragma unroll
          for (int ps = 0; ps < NP; ++ps) u2 = scanned[ps] ? fmax(u2, ub[ps]) : u2;
          u2 = fmax(u2, Grp<LP>::template xchg<0>(u2));
          u2 = fmax(u2, Grp<LP>::template xchg<1>(u2));
          u2 = fmax(u2, Grp<LP>::template xchg<2>(u2));
          if constexpr (LP == 32) {
            u2 = fmax(u2, Grp<LP>::template xchg<3>(u2));
            u2 = fmax(u2, Grp<LP>::template xchg<4>(u2));
          }
#pragma unroll
          for (int ps = 0; ps < NP; ++ps) inband[ps] = inband[ps] || (valid[ps] && !scanned[ps] && ub[ps] >= u2 - delta);
        }
#pragma unroll
        for (int ps = 0; ps < NP; ++ps)
          if (valid[ps]) {
            const size_t s = sample_slot(stride, ia, l + LP * ps);
            gs.sq_ub[s] = ub[ps];
            gs.sq_k[s] = kk[ps];
            if (!scanned[ps]) ub[ps] = -1e300;   // only scanned samples take part in the selection below
          }
      }
      SVSDF_PHASE(rc, 3, tph);
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) umax = fmax(umax, ub[ps]);
      umax = fmax(umax, Grp<LP>::template xchg<0>(umax));
      umax = fmax(umax, Grp<LP>::template xchg<1>(umax));
      umax = fmax(umax, Grp<LP>::template xchg<2>(umax));
      if constexpr (LP == 32) {
        umax = fmax(umax, Grp<LP>::template xchg<3>(umax));
        umax = fmax(umax, Grp<LP>::template xchg<4>(umax));
      }
      unsigned nreq = 0u;
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        list_me[ps] = valid[ps] && ub[ps] >= umax - delta;
        mlist[ps] = ballot_g(list_me[ps]);
        nreq |= mlist[ps] << (LP * ps);
        if (SVSDF_LEAN_HANDOFF && list_me[ps]) {   // the record of a requested sample: what its solve and the closing of the round read
          const size_t s = sample_slot(stride, ia, l + LP * ps);
          gs.sqx[s] = sqx_l[ps]; gs.sqy[s] = sqy_l[ps]; gs.sqth[s] = th_l[ps];
        }
      }
      push_next = true;
      if (l == 0) { gs.nsamp[ia] = n_emit; gs.phase[ia] = (int)kPhaseEval; gs.req[ia] = nreq; }
      SVSDF_PHASE(rc, 4, tph);
    }
  }

// LP lanes per point (8 or 32): the early rounds have 2 and 6 samples, the later ones 18-21; a
// point with more samples than lanes is handled in ceil(n / LP) passes.
// Block size of k_round.  Every block iteration ends in barriers (block-aggregated list atomics), and a block waits for its
// slowest point (a round with 21 sample scans next to one that just closes).  With 1024 threads a block IS the CU's whole
// complement of waves (116 VGPRs -> 4 waves per SIMD), so the CU idles at every barrier: counters showed the waves
// waiting 55 ... 67 % of their cycles.  256 threads leave four independent blocks per CU: C3 8.6 -> 7.7 ms, NS 11.0 ->
// 10.1 ms (round 3; 128 / 64 threads lose again to 4 - 8 x the list atomics).  Round 2 had measured smaller blocks at
// -3.6 % only -- before the concurrent batches, whose tails the free blocks now fill.
#ifndef SVSDF_ROUND_BLOCK
#define SVSDF_ROUND_BLOCK 256
#endif
constexpr int kRoundBlock = SVSDF_ROUND_BLOCK;
// MODE: 0 cheap bound (nearest chunk), 1 full (every new sample scanned), 2 lazy (cheap bound for all, the sample's own
// table scan only for those within `band_delta` of the best cheap bound -- the ones the cheap mode would solve), 3 anchor
// (every third sample scanned, the others only if their Lipschitz bound from the anchors reaches the selection band)
#ifndef SVSDF_ROUND_M3_WAVES
#define SVSDF_ROUND_M3_WAVES 1   // (round 6) 4: hold k_round<analytic shape, 8 lanes, anchor mode> at 128 VGPRs = 4 waves per SIMD -- with the anchored
                                 // walk it needs 134 (3 waves).  The allocator then spills ONE VGPR (8 - 20 B of scratch) and C4 gains 2.1 - 2.5 %
                                 // (500 k: 3.43 -> 3.36 ms, 4 M: 19.41 -> 18.93), but tests/test_gpu_plan.py::test_fused_tail_is_invisible aborts
                                 // with a device fault, like SVSDF_ROUND_WAVES=5 did: these kernels keep ~ 200 spilled SGPRs in VGPR lanes, and a
                                 // VGPR spilled to scratch under a partial EXEC mask is the suspect.  No variant of k_round with scratch is shipped.
#endif
#ifndef SVSDF_ROUND_WAVES
#define SVSDF_ROUND_WAVES 1   // waves per SIMD the register allocation of k_round aims at (1: whatever its registers allow = 4).  Round 6
                              // measured 5 (96 VGPRs, 68 - 190 B of scratch per lane): + 1 ... 4 % -- and k_round<star, 8, anchor> then faulted
                              // (memory aperture violation in the second evaluation of a 8 k-point cloud, tests/test_gpu_plan.py; gone with 4
                              // waves, i.e. without scratch, and not understood): no spilling variant of this kernel is shipped
#endif
template <int SHAPE, int LP, int MODE>
__global__ void __launch_bounds__(kRoundBlock, (LP == 8 && MODE == 3 && SVSDF_ROUND_M3_WAVES > 1 && !is_polygon<SHAPE>()) ? SVSDF_ROUND_M3_WAVES : SVSDF_ROUND_WAVES)
k_round(const TrajDev *__restrict__ trg, const Pose *__restrict__ pose_g,
        const Chunk *__restrict__ chunks_g, ShapeParams sp, const double *__restrict__ px_,
        const double *__restrict__ py_, GsipState gs, size_t stride, int it, double delta, double band_delta,
        double *__restrict__ res_sdf, double *__restrict__ res_t, double *__restrict__ res_gx,
        double *__restrict__ res_gy, BatchCtl *__restrict__ ctl, int clist_on) {
  static_assert(LP == 8 || LP == 32, "lanes per point");
  constexpr int NP = (kMaxSlots + LP - 1) / LP;  // sample passes per point
  extern __shared__ double round_lds[];
  // List positions are handed out per FLUSH, not per block iteration: a point slot buffers what its last LP points requested
  // (kRoundBlock entries per block = one per thread), then one block scan + one set of atomics places them all.  The block
  // meets at 4 barriers per LP iterations instead of 3 per iteration -- a block used to wait, every iteration, for its
  // slowest point (21 sample scans next to a point that only closes its round) and for one lane's serial prefix loop.
  constexpr int PPB = kRoundBlock / LP;            // point slots per block
  constexpr int kEnt = kRoundBlock;                // buffered entries: LP iterations x PPB slots
  constexpr int kWaves = (kRoundBlock + 63) / 64;
  __shared__ unsigned s_mask[kEnt][NP];            // per entry: requested-sample masks (over the point's lanes)
  __shared__ int s_a[kEnt];                        // per entry: interior index | push_next << 30; -1: no point
  __shared__ unsigned short s_emit[kEnt];          // per entry: samples emitted
  __shared__ int s_off[2][kEnt];                   // per entry: offset in the solve / next list within this flush
  __shared__ int s_wsum[3][kWaves];
  __shared__ int s_base[2];
  __shared__ unsigned short s_clist[PPB * kMaxCand];   // per point slot: candidate chunks of its round
  __shared__ int s_next[2];                        // the block's next iteration (dynamic fetch), double-buffered by iteration parity
  const int n_act = ctl->n_active[it];
  const int ppb = blockDim.x / LP;  // points per block (== PPB)
  if (n_act <= 0 || (int)blockIdx.x * ppb >= n_act) return;
  unsigned long long rc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // SVSDF_SITE_STATS builds only (8 .. 15: the seed scans' evaluation site)
  const unsigned long long t_wave0 = SVSDF_SITE_CLOCK();
  unsigned long long tfl = t_wave0;
  const int K = trg->K;
  const int nch = (K + kChunk - 1) / kChunk;
  stage_poly_edges<SHAPE>(sp, round_lds);
  double *tab_lds = round_lds + poly_lds_doubles<SHAPE>(sp.nverts);
  Pose *pose = reinterpret_cast<Pose *>(tab_lds);
  Chunk *chunks = reinterpret_cast<Chunk *>(tab_lds + 4 * (size_t)K);
  {
    const double *src = reinterpret_cast<const double *>(pose_g);
    for (int i = threadIdx.x; i < 4 * K; i += blockDim.x) tab_lds[i] = src[i];
    const double *srcc = reinterpret_cast<const double *>(chunks_g);
    for (int i = threadIdx.x; i < 4 * nch; i += blockDim.x) tab_lds[4 * (size_t)K + i] = srcc[i];
  }
  __syncthreads();
  SVSDF_PHASE(rc, 7, tfl);
  const int start = ctl->start;
  const int *cur = gs.list[it & 1] + start;
  int *nxt = gs.list[(it + 1) & 1] + start;
  int *solve = gs.solve + (size_t)start * kMaxSlots;
  const int l = (int)(threadIdx.x & (LP - 1));
  const int hw = (int)(threadIdx.x / LP);
  const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
  unsigned n_scan = 0;   // table evaluations of the cooperative seed scans (FULL)
  const unsigned lt_mask = (1u << l) - 1u;
  int nbuf = 0;          // block iterations buffered since the last flush (block-uniform)
  // Block iterations (ppb points each): the first is the block's own (block index: no atomic), the following ones come from
  // the launch's cursor (round 6).  The grid-stride loop this replaces gave every block the same NUMBER of iterations, but an
  // iteration costs anything between a few state loads (its points only close a round and finish) and 18 - 21 seed scans per
  // point (they open one), and Morton-sorted neighbours behave alike: counters showed the launches of the first GSIP
  // iterations at 60 - 70 % of their wave slots on average -- blocks that drew light iterations had left, the launch waited
  // for those that drew heavy ones (SQ_WAVE_CYCLES / duration: 2 450 of 4 096 waves resident in iteration 2 of C3).
  // block-uniform trip count; lane groups without a point still take part in the flushes
  int par = 0;
  for (int e0 = (int)blockIdx.x * ppb; e0 < n_act;) {
    const int e = e0 + hw;
    const bool active = e < n_act;
    int a = 0;
    RoundOut<NP> ro;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) { ro.list_me[ps] = false; ro.mlist[ps] = 0u; }
    ro.n_emit = 0; ro.push_next = false; ro.finished = false;
    if (active) {
      a = cur[e];
      round_point<SHAPE, LP, MODE>(sp, pose, chunks, K, nch, px_, py_, gs, stride, start, a, delta, band_delta, res_sdf,
                                   res_t, res_gx, res_gy, n_scan, ro, s_clist + (size_t)hw * kMaxCand, clist_on, rc, -1,
                                   (clist_on & 16) ? reinterpret_cast<const ChunkAnchor *>(chunks_g + nch) : nullptr);
    }
    if (l == 0) {
      const int ent = nbuf * PPB + hw;
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) s_mask[ent][ps] = ro.mlist[ps];
      s_a[ent] = active ? (a | (ro.push_next ? (1 << 30) : 0)) : -1;
      s_emit[ent] = (unsigned short)ro.n_emit;
    }
    ++nbuf;
    if (threadIdx.x == 0) s_next[par] = ((int)gridDim.x + (int)atomicAdd(&ctl->rwork[it], 1u)) * ppb;
    __syncthreads();
    e0 = s_next[par];   // (the other parity is written one iteration later: no wave can still be reading it)
    par ^= 1;
    const bool last = e0 >= n_act;
    if (nbuf < LP && !last) continue;
    // ---- flush: one block scan, one set of list atomics, then every point slot writes the entries of its points
    tfl = SVSDF_SITE_CLOCK();
    __syncthreads();
    const int nent = nbuf * PPB;
    int c0 = 0, c1 = 0, c2 = 0;
    if ((int)threadIdx.x < nent) {
      const int av = s_a[threadIdx.x];
      if (av != -1) {
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) c0 += __popc(s_mask[threadIdx.x][ps]);
        c1 = (av >> 30) & 1;
        c2 = (int)s_emit[threadIdx.x];
      }
    }
    int x0 = c0, x1 = c1, x2 = c2;   // inclusive scans over the wave (entry order = thread order); x2: plain sum
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const int y0 = __shfl_up(x0, m, 64), y1 = __shfl_up(x1, m, 64);
      if (lane >= m) { x0 += y0; x1 += y1; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x2 += __shfl_xor(x2, m, 64);
    if (lane == 63) { s_wsum[0][wv] = x0; s_wsum[1][wv] = x1; s_wsum[2][wv] = x2; }
    __syncthreads();
    int pre0 = 0, pre1 = 0, tot0 = 0, tot1 = 0, tot2 = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      if (w < wv) { pre0 += s_wsum[0][w]; pre1 += s_wsum[1][w]; }
      tot0 += s_wsum[0][w]; tot1 += s_wsum[1][w]; tot2 += s_wsum[2][w];
    }
    if (threadIdx.x == 0) {
      s_base[0] = tot0 ? atomicAdd(&ctl->n_solve[it], tot0) : 0;
      s_base[1] = tot1 ? atomicAdd(&ctl->n_active[it + 1], tot1) : 0;
      if (tot2) atomicAdd(&ctl->n_seed[it], tot2);
    }
    if ((int)threadIdx.x < nent) { s_off[0][threadIdx.x] = pre0 + x0 - c0; s_off[1][threadIdx.x] = pre1 + x1 - c1; }
    __syncthreads();
    for (int k = 0; k < nbuf; ++k) {
      const int ent = k * PPB + hw;
      const int av = s_a[ent];
      if (av == -1) continue;
      const int a2 = av & 0x3fffffff;
      const size_t ia2 = (size_t)a2;
      int pos = s_base[0] + s_off[0][ent];
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const unsigned m = s_mask[ent][ps];
        if ((m >> l) & 1u) solve[pos + __popc(m & lt_mask)] = (int)sample_slot(stride, ia2, l + LP * ps);
        pos += __popc(m);
      }
      if (((av >> 30) & 1) && l == 0) nxt[s_base[1] + s_off[1][ent]] = a2;
    }
    __syncthreads();   // the buffers are rewritten by the next iterations
    SVSDF_PHASE(rc, 5, tfl);
    nbuf = 0;
  }
#ifdef SVSDF_SITE_STATS
  if ((threadIdx.x & 63) == 0) {
    rc[6] = SVSDF_SITE_CLOCK() - t_wave0;
    StatSlot *ss = stat_slot(ctl->stat);
    for (int i = 0; i < 8; ++i) if (rc[i]) atomicAdd(&ss->pad[12 + i], rc[i]);
  }
  {   // the seed scans' evaluation site: wave-level executions (pad[20]) and evaluating lanes (pad[21])
    unsigned long long ex = rc[8], ln = rc[12];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { ex += __shfl_xor(ex, m, 64); ln += __shfl_xor(ln, m, 64); }
    if ((threadIdx.x & 63) == 0 && ex) { StatSlot *ss = stat_slot(ctl->stat); atomicAdd(&ss->pad[20], ex); atomicAdd(&ss->pad[21], ln); }
  }
#endif
  if constexpr (MODE != 0) {
    unsigned long long tc = n_scan;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) tc += __shfl_xor(tc, m, 64);
    if ((threadIdx.x & 63) == 0 && tc) {
      StatSlot *ss = stat_slot(ctl->stat);
      atomicAdd(&ss->round_scan, tc);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_tail: ALL remaining GSIP iterations of a batch in one launch (from iteration `it0` on, i.e. the launches
// R_it0 S_it0 R_it0+1 ... of the chain).  A GSIP point is independent of every other point, so once the active set is small
// the launch chain -- two kernels per iteration, each ending in a tail where the chip drains, ~10 iterations -- only
// costs latency.  Here a half-wave (32 lanes, lane j <-> circle sample j, round_point<.., 32, ..>) OWNS a point until it
// is finished:  close the round -> open the next -> select -> solve the selected samples IN THE WAVE -> close ... .  No
// hand-off between workgroups, no list atomics, no block barrier after the table staging; a half-wave that has finished
// its point takes the next one from the active list.  The solves of the wave's two points run together: the wave's
// <= 48 selected samples are dealt out to lane groups of 32 (up to 2 queries), 8 (up to 8) or 2 lanes, each group runs the very
// scan_layer1 / descend_from_seed of k_solve (the descent's ladders share the wave as there).  Same per-sample and
// per-point arithmetic as the launch chain, so identical bits whatever it0 is (the host picks it from the previous
// evaluation's active counts).
//  * `prev_mode`: bound mode (k_round MODE) of the launch chain that opened the rounds this kernel finds open: decides
//    whether those samples carry a scan seed (sq_k); rounds opened here follow MODE.
//  * selection band: `delta` for the first `all_after` steps of a point in this kernel, everything afterwards.
// LDS: [Polygon edges | pose table 4K | chunks 4 nch | trajectory 20N+1] doubles, then kTailWaveLds bytes per wave.
// ---------------------------------------------------------------------------------------------
constexpr int kTailBlock = 256;
constexpr int kTailFetch = 2;     // points a wave takes from the active list per atomic (8 made the last waves serialise four pairs each: + 1 ms beyond 6144 points)
// the wave's own copy of the GSIP state and samples of its two points (it0 == 0, see k_tail): 6 + 6 * 48 doubles, 10 + 48 ints
constexpr size_t kTailLocalBytes = ((6 + 6 * 2 * kMaxSlots) * sizeof(double) + (10 + 2 * kMaxSlots) * sizeof(int) + 15) & ~(size_t)15;
constexpr size_t kTailWaveLds = ((ladder_lds_bytes(2) + 15) & ~(size_t)15) + 2 * kMaxCand * sizeof(unsigned short) + 64 * sizeof(unsigned) + kTailLocalBytes;
template <int SHAPE, int G>
__device__ __forceinline__ void tail_solve_pass(const TrajL &tr, const double *__restrict__ tk, const ShapeParams &sp,
                                                const Pose *pose, const Chunk *chunks, int K, int nch, const GsipState &gs,
                                                const unsigned *qlist, int base, int nq, int prune, void *wave_lds,
                                                unsigned &n_eval, unsigned &n_scan, unsigned &n_solved, unsigned &n_spec,
                                                unsigned long long (&sc)[12]) {
  const int lane = (int)(threadIdx.x & 63);
  const int li = Grp<G>::li();
  const int q = base + lane / G;
  const bool live = q < nq;
  const unsigned long long t_scan0 = SVSDF_SITE_CLOCK();
  size_t slot = 0;
  double px = 0.0, py = 0.0;
  double best_d = 1e9;
  int best_k = 0x7fffffff;
  if (live) {
    const unsigned e = qlist[q];
    slot = (size_t)(e & 0x7fffffffu);
    px = gs.sqx[slot]; py = gs.sqy[slot];
    const bool seeded = (e >> 31) != 0u;
    if (seeded) { best_k = gs.sq_k[slot]; best_d = gs.sq_ub[slot]; }
    if (!seeded || best_k < 0) {
      bool culled;
      scan_layer1<SHAPE, G>(sp, pose, chunks, K, nch, px, py, prune, __longlong_as_double(0x7ff0000000000000ll), best_d, best_k,
                            culled, n_scan, nullptr, -1, nullptr);
    }
  }
  SVSDF_SITE_CYCLES(sc, 8, t_scan0);
  double x = 0.0, fx = 0.0;
  descend_from_seed<SHAPE, G, 1>(tr, tk, sp, px, py, live, best_k, best_d, x, fx, n_eval, n_spec, sc, wave_lds);   // whole wave
  if (live && li == 0) {
    gs.sq_sdf[slot] = fx;
    gs.sq_t[slot] = x;
    ++n_solved;
  }
}

#ifndef SVSDF_TAIL_WAVES
#define SVSDF_TAIL_WAVES 3   // waves per SIMD the register allocation aims at (168 VGPRs)
#endif
// WAVES: waves per SIMD the register allocation aims at.  3 (168 VGPRs) is what a cloud of thousands of points wants -- and
// costs ~ 100 spilled VGPRs (300 - 400 B of scratch per lane).  A launch with a wave slot for every point at TWO waves per
// SIMD (<= 2048 points on 256 CUs: the reference's own scale) is a chain of dependent steps of single waves, where a scratch
// round trip is pure latency: kTailLatencyWaves = 2 lets the kernel keep its ~ 240 VGPRs, no scratch (end of round 6:
// reference-scale callbacks - 4 ... - 5 %, 3 k-point clouds - 1.5 %; 10 k points + 2 % -- hence two instantiations,
// chosen per launch: launch_tail).
constexpr int kTailLatencyWaves = 2;
template <int SHAPE, int MODE, int WAVES = SVSDF_TAIL_WAVES>
__global__ void __launch_bounds__(kTailBlock, WAVES)
k_tail(const TrajDev *__restrict__ trg, const double *__restrict__ tk, const Pose *__restrict__ pose_g,
       const Chunk *__restrict__ chunks_g, ShapeParams sp, const double *__restrict__ px_, const double *__restrict__ py_,
       GsipState gs, size_t stride, int it0, int prev_mode, double delta, double band_delta, int all_after, int ppw,
       double *__restrict__ res_sdf, double *__restrict__ res_t, double *__restrict__ res_gx, double *__restrict__ res_gy,
       BatchCtl *__restrict__ ctl, int clist_on, int prune) {
  extern __shared__ double tail_lds[];
  constexpr int LP = 32;
  const int n_act = ctl->n_active[it0];
  const int wave_g = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (n_act <= 0 || (int)blockIdx.x * (kTailBlock / 64) * ppw >= n_act) return;   // block-uniform
  const int K = trg->K;
  const int nch = (K + kChunk - 1) / kChunk;
  stage_poly_edges<SHAPE>(sp, tail_lds);
  double *tab_lds = tail_lds + poly_lds_doubles<SHAPE>(sp.nverts);
  Pose *pose = reinterpret_cast<Pose *>(tab_lds);
  Chunk *chunks = reinterpret_cast<Chunk *>(tab_lds + 4 * (size_t)K);
  {
    const double *src = reinterpret_cast<const double *>(pose_g);
    for (int i = threadIdx.x; i < 4 * K; i += blockDim.x) tab_lds[i] = src[i];
    const double *srcc = reinterpret_cast<const double *>(chunks_g);
    for (int i = threadIdx.x; i < 4 * nch; i += blockDim.x) tab_lds[4 * (size_t)K + i] = srcc[i];
  }
  const TrajL tr = stage_traj(trg, tab_lds + 4 * (size_t)K + 4 * (size_t)nch);  // ends with __syncthreads (the only block barrier)
  const size_t tables = poly_lds_doubles<SHAPE>(sp.nverts) + 4 * (size_t)K + 4 * (size_t)nch + (size_t)traj_lds_doubles(tr.N);
  const bool local = it0 == 0 && (clist_on & 4) != 0;   // (bit 2 of clist_on: the host's switch, SVSDF_TAIL_LOCAL=0 turns it off; see below)
  // per-wave stride: the wave-local GSIP block (the last part of a wave's region) only exists when it is used (launch_tail
  // sizes the launch's LDS the same way)
  const size_t wave_stride = local ? kTailWaveLds : kTailWaveLds - kTailLocalBytes;
  char *wave_lds = reinterpret_cast<char *>(tail_lds + ((tables + 1) & ~(size_t)1)) + (threadIdx.x >> 6) * wave_stride;
  unsigned short *clist_w = reinterpret_cast<unsigned short *>(wave_lds + ((ladder_lds_bytes(2) + 15) & ~(size_t)15));
  unsigned *qlist = reinterpret_cast<unsigned *>(clist_w + 2 * kMaxCand);
  // Round 5: when the WHOLE GSIP loop runs here (it0 == 0: every point arrives fresh from k_classify, no round open), the
  // state and the samples of the wave's two points live in the wave's own LDS instead of the interior-sized arrays in
  // global memory: round_point and the solve passes are handed a GsipState whose pointers address that LDS block (generic
  // pointers; slot = sample * 2 + half), nothing else changes.  A GSIP step was four dependent trips to L2 (state,
  // samples, the solve's inputs, its results) of ~ 1.5 us each around ~ 20 us of arithmetic, ten steps per point.
  GsipState gl = gs;
  if (local) {
    double *ld = reinterpret_cast<double *>(qlist + 64);
    gl.r = ld; gl.theta0 = ld + 2; gl.theta_res = ld + 4;
    gl.sqx = ld + 6; gl.sqy = gl.sqx + 2 * kMaxSlots; gl.sqth = gl.sqy + 2 * kMaxSlots; gl.sq_ub = gl.sqth + 2 * kMaxSlots;
    gl.sq_sdf = gl.sq_ub + 2 * kMaxSlots; gl.sq_t = gl.sq_sdf + 2 * kMaxSlots;
    int *li_ = reinterpret_cast<int *>(gl.sq_t + 2 * kMaxSlots);
    gl.pt = li_; gl.iter = li_ + 2; gl.nsamp = li_ + 4; gl.phase = li_ + 6; gl.req = reinterpret_cast<unsigned *>(li_ + 8);
    gl.sq_k = li_ + 10;
  }
  const GsipState ga = local ? gl : gs;
  const size_t stride_a = local ? (size_t)2 : stride;
  const int start = ctl->start;
  const int *cur = gs.list[it0 & 1] + start;
  const int lane = (int)(threadIdx.x & 63);
  const int h = lane >> 5, l = lane & (LP - 1);
  // One point per wave (ppw == 1: the launch is a latency chain and has a wave for every point): both half-waves OWN that
  // point -- same state, same decisions, same stores -- and share its seed scans (round_point's `duo`); only half 0's
  // requests go to the wave's solve list.  (The scanning bound modes; the cheap mode has no scans to share.)
  const bool duo = ppw == 1 && MODE != 0 && (clist_on & 8) == 0;
  const int hs = duo ? 0 : h;   // the half whose slot of the wave-local GSIP state this lane uses
  const unsigned lt_mask = (1u << l) - 1u;
  unsigned n_eval = 0, n_scan = 0, n_solved = 0, n_spec = 0, n_rscan = 0;
  unsigned long long rc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // SVSDF_SITE_STATS builds only (8 .. 15: the seed scans' evaluation site)
  unsigned long long sc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_wave0 = SVSDF_SITE_CLOCK();
  int n_emit_tot = 0;
  // work: the wave's first two points are its own (wave index: no atomic), further ones come kTailFetch at a time
  // ppw = points per wave: 2 (one per half-wave), or 1 when the launch holds few points -- a pure latency chain then, and
  // a wave with one point's handful of samples solves them with wider lane groups
  const int n_static = (int)(gridDim.x * (blockDim.x >> 6)) * ppw;
  int q_next = wave_g * ppw, q_end = min(wave_g * ppw + ppw, n_act);   // wave-uniform
  bool exhausted = false;
  int a = -1;          // interior index of this half's point (-1: none)
  int steps = 0;       // GSIP steps this half's point has taken in this kernel
  bool own = false;    // the open round of this half's point was opened here (MODE) and not by the launch chain (prev_mode)
  for (int guard = 0; guard < (1 << 24); ++guard) {
    // ---- refill: a half without a point takes the next one
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int ah = __builtin_amdgcn_readlane(a, hh * 32);
      if (ah >= 0 || hh >= ppw) continue;
      if (q_next >= q_end && !exhausted) {
        unsigned b = 0;
        if (lane == 0) b = atomicAdd(&ctl->work[min(it0 + 1, kWorkCounters - 1)], (unsigned)kTailFetch);
        b = __builtin_amdgcn_readfirstlane(b);
        const long long bb = (long long)b + n_static;
        if (bb >= n_act) exhausted = true;
        else { q_next = (int)bb; q_end = (int)min((long long)n_act, bb + kTailFetch); }
      }
      if (q_next < q_end) {
        const int e = q_next++;
        if (h == hh || duo) {
          a = cur[e]; steps = 0; own = false;
          if (local && l == 0 && h == hh) {   // the point's state as k_classify left it, into this half's slot of the wave's LDS copy
            const size_t ia_ = (size_t)a;
            gl.pt[hs] = gs.pt[ia_]; gl.r[hs] = gs.r[ia_]; gl.theta0[hs] = gs.theta0[ia_]; gl.theta_res[hs] = gs.theta_res[ia_];
            gl.iter[hs] = gs.iter[ia_]; gl.nsamp[hs] = gs.nsamp[ia_]; gl.phase[hs] = gs.phase[ia_]; gl.req[hs] = 0u;
          }
        }
      }
    }
    if (__builtin_amdgcn_readlane(a, 0) < 0 && __builtin_amdgcn_readlane(a, 32) < 0) break;
    if (local) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // ---- one GSIP step per half: close / finish / open + select (k_round's round_point, 32 lanes per point)
    RoundOut<1> ro;
    ro.list_me[0] = false; ro.mlist[0] = 0u; ro.n_emit = 0; ro.push_next = false; ro.finished = false;
    if (a >= 0) {
      const double dl = (steps >= all_after) ? 1e300 : delta, bd = (steps >= all_after) ? 1e300 : band_delta;
      round_point<SHAPE, LP, MODE, true>(sp, pose, chunks, K, nch, px_, py_, ga, stride_a, start, local ? hs : a, dl, bd, res_sdf, res_t, res_gx,
                                   res_gy, n_rscan, ro, clist_w + (size_t)h * kMaxCand, clist_on, rc, duo ? h : -1,
                                   (clist_on & 16) ? reinterpret_cast<const ChunkAnchor *>(chunks_g + nch) : nullptr);
      ++steps;
      if (ro.n_emit > 0) own = true;
      if (l == 0 && (!duo || h == 0)) n_emit_tot += ro.n_emit;
    }
    const size_t ia = local ? (size_t)hs : (size_t)(a >= 0 ? a : 0);
    if (ro.finished) a = -1;
    // ---- the wave's solve list: the selected samples of both halves (duo: of the one point, from half 0)
    const unsigned m_mine = (a >= 0 && (!duo || h == 0)) ? ro.mlist[0] : 0u;
    const int n0 = __popc((unsigned)__builtin_amdgcn_readlane((int)m_mine, 0));
    const int n1 = __popc((unsigned)__builtin_amdgcn_readlane((int)m_mine, 32));
    const int nq = n0 + n1;
    if ((m_mine >> l) & 1u) {
      const bool seeded = own ? (MODE != 0) : (prev_mode != 0);
      qlist[(h ? n0 : 0) + __popc(m_mine & lt_mask)] = (unsigned)sample_slot(stride_a, ia, l) | (seeded ? 0x80000000u : 0u);
    }
    // the samples and the point state just written are read by other lanes of this wave, the solved values below by the
    // next step's close
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (nq > 0) {
      if (nq <= 2) {
        tail_solve_pass<SHAPE, 32>(tr, tk, sp, pose, chunks, K, nch, ga, qlist, 0, nq, prune, wave_lds, n_eval, n_scan, n_solved, n_spec, sc);
      } else if (nq <= 8) {
        tail_solve_pass<SHAPE, 8>(tr, tk, sp, pose, chunks, K, nch, ga, qlist, 0, nq, prune, wave_lds, n_eval, n_scan, n_solved, n_spec, sc);
      } else {
        for (int base = 0; base < nq; base += 32)
          tail_solve_pass<SHAPE, 2>(tr, tk, sp, pose, chunks, K, nch, ga, qlist, base, nq, prune, wave_lds, n_eval, n_scan, n_solved, n_spec, sc);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
  unsigned long long te = (unsigned long long)n_eval + n_scan, ts = n_solved, tc = n_scan, tp = n_spec, tr_ = n_rscan;
  int em = n_emit_tot;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    te += __shfl_xor(te, m, 64); ts += __shfl_xor(ts, m, 64); tc += __shfl_xor(tc, m, 64); tp += __shfl_xor(tp, m, 64);
    tr_ += __shfl_xor(tr_, m, 64); em += __shfl_xor(em, m, 64);
  }
  if (lane == 0 && (te || tr_ || em)) {
    StatSlot *ss = stat_slot(ctl->stat);
    if (te) atomicAdd(&ss->evals, te);
    if (ts) atomicAdd(&ss->solves, ts);
    if (tc) atomicAdd(&ss->scan, tc);
    if (tp) atomicAdd(&ss->spec, tp);
    if (tr_) atomicAdd(&ss->round_scan, tr_);
    if (em) atomicAdd(&ctl->n_seed[it0], em);
    if (ts) atomicAdd(&ctl->n_solve[it0], (int)ts);
  }
#ifdef SVSDF_SITE_STATS
  if (lane == 0) {   // wave cycles per phase: round_point's (pad[12 ..], like k_round) and the solve passes' (pad[8 .. 10], like k_solve)
    rc[6] = SVSDF_SITE_CLOCK() - t_wave0;
    StatSlot *ss = stat_slot(ctl->stat);
    for (int i = 0; i < 8; ++i) if (rc[i]) atomicAdd(&ss->pad[12 + i], rc[i]);
    for (int i = 8; i < 11; ++i) if (sc[i]) atomicAdd(&ss->pad[i], sc[i]);
  }
  {
    unsigned long long ex = rc[8], ln = rc[12];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { ex += __shfl_xor(ex, m, 64); ln += __shfl_xor(ln, m, 64); }
    if (lane == 0 && ex) { StatSlot *ss = stat_slot(ctl->stat); atomicAdd(&ss->pad[20], ex); atomicAdd(&ss->pad[21], ln); }
    // the launch's SLOWEST wave (round 6: the dependent depth of a reference-scale callback): evaluation-site executions of
    // one wave -- its seed scans' and its solve passes' -- and the cycles it lived; the maximum over the waves (pad[22], pad[23])
    unsigned long long steps = ex;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned long long v = sc[i];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      steps += v;
    }
    if (lane == 0) { StatSlot *ss = stat_slot(ctl->stat); atomicMax(&ss->pad[22], steps); atomicMax(&ss->pad[23], rc[6]); }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {   // site executions (0 .. 3) and evaluating lanes (4 .. 7), like k_solve
    unsigned long long v = sc[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0 && v) atomicAdd(&stat_slot(ctl->stat)->pad[i], v);
  }
#endif
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

#ifdef SVSDF_API_TU   // shape-independent kernels: compiled once, by svsdf_pipeline.hip
// The 20 terms one query point adds to the sums (BEO:797-863, grad_cost_p_sw BEO:1031-1066): v[0] cost, v[1..18] gradC of
// its piece i (x, y, yaw blocks of 6), v[19] the piece's gradT history term.  Returns false (v all zero) for an inactive point.
__device__ __forceinline__ bool assemble_terms(const TrajL &tr, const double *__restrict__ px_, const double *__restrict__ py_,
                                               int idx, const double *__restrict__ res_sdf, const double *__restrict__ res_t,
                                               const double *__restrict__ res_gx, const double *__restrict__ res_gy,
                                               double safety_hor, double weight_p, double (&v)[20], int &i, int &bad) {
  const double px = px_[idx], py = py_[idx];
  const double sdf_value = res_sdf[idx];
  const double time_star = res_t[idx];
  double gr0 = res_gx[idx], gr1 = res_gy[idx];
  if (!(sdf_value == sdf_value) || !(time_star == time_star) || !(gr0 == gr0) || !(gr1 == gr1)) ++bad;
  double sdf_cost = -1.0, sdf_out_grad = 0.0;
  smoothed_l1(safety_hor - sdf_value, 0.01, sdf_cost, sdf_out_grad);
  if (!(sdf_cost > 0)) return false;
  double s1;
  i = locate_local(tr, time_star, 0, s1);
  const double *c = tr.c + i * 18;
  const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
  const double beta0[6] = {1.0, s1, s2, s3, s4, s5};
  const double beta1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
  double pos[3], vel[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    double p = 0.0, vv = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { p += c[k * 3 + d] * beta0[k]; vv += c[k * 3 + d] * beta1[k]; }
    pos[d] = p; vel[d] = vv;
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
  v[0] = weight_p * sdf_cost;
#pragma unroll
  for (int k = 0; k < 6; ++k) { v[1 + k] = beta0[k] * gPx; v[7 + k] = beta0[k] * gPy; v[13 + k] = beta0[k] * gYaw; }
  v[19] = -((gPx * vel[0] + gPy * vel[1]) + gYaw * vel[2]);
  return true;
}

// doubles of LDS behind the trajectory that the one-block form of the assembly uses: the sums [plen], the terms of the
// block's points term-major [20][kBlock], and per wave the lane masks of the active points and of every piece [N + 1] (64-bit)
__host__ __device__ __forceinline__ size_t assemble_small_doubles(int N) {
  return (size_t)(19 * N + 1) + 20 * (size_t)kBlock + (size_t)(kBlock / 64) * (size_t)(N + 1);
}

__device__ __forceinline__ void assemble_body(const TrajDev *__restrict__ trg, const double *__restrict__ px_,
           const double *__restrict__ py_, int P, const double *__restrict__ res_sdf,
           const double *__restrict__ res_t, const double *__restrict__ res_gx,
           const double *__restrict__ res_gy, double safety_hor, double weight_p,
           double *__restrict__ block_partials, int *__restrict__ nonfinite, double *asm_lds, bool keep_in_lds = false) {
  // Deterministic (bit-reproducible run to run): no floating-point atomics.  Every wave owns a private
  // accumulator row in LDS; per grid-stride step the wave walks the distinct piece ids among its active lanes
  // (lowest lane first), sums each of the 20 per-point terms of that piece with a fixed xor butterfly (every
  // lane ends with the same bits) and lane 0 adds the totals to the wave's row.  The point -> (block, wave, lane,
  // step) assignment depends on P and the grid only; rows are then summed in wave order, block partials in block
  // order (k_final).  Morton-sorted neighbours share their piece, so a wave sees 1-3 distinct ids per step.
  const TrajL tr = stage_traj(trg, asm_lds);
  const int N = tr.N;
  const int plen = 19 * N + 1;
  constexpr int kWaves = kBlock / 64;
  double *acc_all = asm_lds + traj_lds_doubles(N);
  const int lane = (int)(threadIdx.x & 63);
  if (keep_in_lds) {
    // One block holds the whole cloud (P <= kBlock: the reference's own scale, 101 .. 139 points on its demo maps; round 6).
    // A wave of such a cloud holds a dozen DIFFERENT pieces and the walk below costs 20 six-step butterflies for each of them,
    // one wave per SIMD, nothing to hide the exchange latency behind: 25 of the 29 us this launch took in a 340 - 400 us
    // callback.  Here every active point leaves its 20 terms in LDS (term-major: lane-contiguous, no bank conflicts) and the
    // waves leave one lane mask per piece; then ONE THREAD PER ENTRY of the result adds the terms of its piece's points in
    // ascending point index -- the order of the reference's serial loop (BEO:797-863) and of the oracle.  A handful of
    // dependent additions per entry instead of 120 exchange steps per piece.  (Fixed order: bit-reproducible like the
    // butterflies; the rounding of the sums differs from theirs in the last bits, as any two orders do.)
    double *vals = acc_all + plen;
    unsigned long long *masks = reinterpret_cast<unsigned long long *>(vals + 20 * (size_t)kBlock);   // [kWaves][N + 1], entry N: active
    const int idx = (int)threadIdx.x, wv = (int)(threadIdx.x >> 6);
    int i = -1, bad = 0;
    double v[20];
#pragma unroll
    for (int q = 0; q < 20; ++q) v[q] = 0.0;
    const bool act = idx < P && assemble_terms(tr, px_, py_, idx, res_sdf, res_t, res_gx, res_gy, safety_hor, weight_p, v, i, bad);
    if (act) {
#pragma unroll
      for (int q = 0; q < 20; ++q) vals[(size_t)q * kBlock + threadIdx.x] = v[q];
    }
    for (int p0 = 0; p0 <= N; ++p0) {
      const unsigned long long mk = __ballot(act && (p0 == N || i == p0));
      if (lane == 0) masks[(size_t)wv * (N + 1) + p0] = mk;
    }
    if (bad) atomicAdd(nonfinite, bad);
    __syncthreads();
    for (int e = threadIdx.x; e < plen; e += blockDim.x) {
      // entry e <-> (piece, term): e = 0 the cost (all pieces), 1 + blk 6 N + 6 i + r gradC, 1 + 18 N + i the gradT history
      int pc = N, q = 0;
      if (e > 18 * N) { pc = e - 1 - 18 * N; q = 19; }
      else if (e > 0) { const int blk = (e - 1) / (6 * N), rem = (e - 1) % (6 * N); pc = rem / 6; q = 1 + blk * 6 + rem % 6; }
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        unsigned long long m = masks[(size_t)w * (N + 1) + pc];
        while (m) {
          const int ln = __ffsll((long long)m) - 1;
          s += vals[(size_t)q * kBlock + w * 64 + ln];
          m &= m - 1ull;
        }
      }
      block_partials[(size_t)e] = s;   // (gridDim.x == 1)
      acc_all[e] = 0.0 + s;            // k_final's sum of this entry over one block partial, for k_reduce
    }
    return;
  }
  for (int e = threadIdx.x; e < kWaves * plen; e += blockDim.x) acc_all[e] = 0.0;
  __syncthreads();
  double *acc = acc_all + (size_t)(threadIdx.x >> 6) * plen;
  int bad = 0;
  for (int base = blockIdx.x * blockDim.x; base < P; base += gridDim.x * blockDim.x) {
    const int idx = base + (int)threadIdx.x;
    bool act = false;
    int i = -1;
    double v[20];
#pragma unroll
    for (int q = 0; q < 20; ++q) v[q] = 0.0;
    if (idx < P) act = assemble_terms(tr, px_, py_, idx, res_sdf, res_t, res_gx, res_gy, safety_hor, weight_p, v, i, bad);
    unsigned long long m = __ballot(act);
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      const int i0 = __shfl(i, src, 64);
      const bool sel = act && (i == i0);
      // the 20 butterflies advance together, one exchange step at a time (round 5): twenty independent dependent chains
      // of cross-lane exchanges instead of one after the other -- a small cloud's wave holds a dozen distinct pieces and
      // spent 46 us here for 101 points.  Same partners, same order of additions per entry: same bits.
      double s[20];
#pragma unroll
      for (int q = 0; q < 20; ++q) s[q] = sel ? v[q] : 0.0;
#pragma unroll
      for (int x = 1; x <= 32; x <<= 1) {
#pragma unroll
        for (int q = 0; q < 20; ++q) s[q] += __shfl_xor(s[q], x, 64);
      }
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 20; ++q) {
          const int e = (q == 0) ? 0 : (q == 19) ? (1 + 18 * N + i0) : (1 + ((q - 1) / 6) * 6 * N + 6 * i0 + (q - 1) % 6);
          acc[e] += s[q];
        }
      }
      m &= ~__ballot(sel);
    }
  }
  if (bad) atomicAdd(nonfinite, bad);
  __syncthreads();
  for (int e = threadIdx.x; e < plen; e += blockDim.x) {
    double s = acc_all[e];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) s += acc_all[(size_t)w * plen + e];
    block_partials[(size_t)e * gridDim.x + blockIdx.x] = s;
  }
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
__device__ __forceinline__ void finish_body(const double *__restrict__ sums, int N, double *__restrict__ partial,
                                            const BatchCtl *__restrict__ ctl, int nbatch, int it_end,
                                            const int *__restrict__ nonfinite, unsigned long long *__restrict__ stats_out) {
  for (int k = threadIdx.x; k <= 18 * N; k += blockDim.x) partial[k] = sums[k];
  if (threadIdx.x == 0) {
    double suf = 0.0;
    for (int j = N - 1; j >= 0; --j) { partial[1 + 18 * N + j] = suf; suf += sums[1 + 18 * N + j]; }
  }
  if (threadIdx.x < 64) {
    // the counters, gathered by the first wave (round 5: one thread used to walk nbatch x 32 slots x 6 words and three
    // per-iteration arrays on its own -- 13 us of a 490 us reference-scale callback): lane k takes stat slot k (k < 32) and
    // GSIP iteration k (k < kMaxIter); integer sums, so the order is free
    const int lane = (int)threadIdx.x;
    unsigned long long so = 0, ev = 0, sc = 0, cu = 0, rs = 0, sp_ = 0, seeded = 0, iters = 0, ns = 0, na = 0;
    for (int b = 0; b < nbatch; ++b) {
      if (lane < kStatSlots) {
        const StatSlot &ss = ctl[b].stat[lane];
        so += ss.solves; ev += ss.evals; sc += ss.scan; cu += ss.culled; rs += ss.round_scan; sp_ += ss.spec;
      }
      if (lane < kMaxIter) {
        ns += (unsigned long long)ctl[b].n_solve[lane];
        na += (unsigned long long)ctl[b].n_active[lane];
      }
      if (lane <= it_end && lane < kMaxIter + 2) {
        seeded += (unsigned long long)ctl[b].n_seed[lane];
        if (ctl[b].n_solve[lane] > 0 && (unsigned long long)(lane + 1) > iters) iters = (unsigned long long)(lane + 1);
      }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      so += __shfl_xor(so, m, 64); ev += __shfl_xor(ev, m, 64); sc += __shfl_xor(sc, m, 64); cu += __shfl_xor(cu, m, 64);
      rs += __shfl_xor(rs, m, 64); sp_ += __shfl_xor(sp_, m, 64); seeded += __shfl_xor(seeded, m, 64);
      const unsigned long long oi = __shfl_xor(iters, m, 64);
      iters = (oi > iters) ? oi : iters;
    }
    if (lane < kMaxIter) {
      stats_out[9 + lane] = ns;                  // solves per GSIP iteration: the host sizes the next evaluation's lane groups
      stats_out[11 + kMaxIter + lane] = na;      // active GSIP points per iteration: the host places the fused tail (k_tail) by them
    }
    if (lane == 0) {
      unsigned long long in = 0, nf = (unsigned long long)__hip_atomic_load(nonfinite, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), rem = 0;
      for (int b = 0; b < nbatch; ++b) {
        in += (unsigned long long)ctl[b].n_active[0]; nf += (unsigned long long)ctl[b].nonfinite;
        rem += (unsigned long long)ctl[b].n_solve[it_end];    // > 0: solves requested but not run yet
      }
      stats_out[0] = so; stats_out[1] = ev; stats_out[2] = sc; stats_out[3] = in; stats_out[4] = nf;
      stats_out[5] = rem; stats_out[6] = seeded; stats_out[7] = iters; stats_out[8] = cu;
      stats_out[9 + kMaxIter] = rs;
      stats_out[10 + kMaxIter] = sp_;
      stats_out[11 + 2 * kMaxIter] = (unsigned long long)ctl[0].n_int;   // interior points found (may exceed the capacity: repeat)
      stats_out[12 + 2 * kMaxIter] = ctl[0].clk[0];                       // clock probe of batch 0's main solve (BatchCtl::clk)
      stats_out[13 + 2 * kMaxIter] = ctl[0].clk[1];
    }
  }
}

// What the host reads of a result row [sums (19 kMaxPieces + 1) | counters]: the 19 N + 1 sums in use and the counters.
__device__ __forceinline__ void copy_result_to_host(double *__restrict__ host_out, const double *__restrict__ out, int N,
                                                    int out_partial, int out_doubles) {
  const int plen = 19 * N + 1;
  for (int k = threadIdx.x; k < plen; k += blockDim.x) host_out[k] = out[k];
  for (int k = out_partial + threadIdx.x; k < out_doubles; k += blockDim.x) host_out[k] = out[k];
}

// (host_out / out_doubles: the pinned host buffer the result -- partial + counters, laid out like `partial` -- is also
// written to, or null)
__global__ void k_finish(const double *__restrict__ sums, int N, double *__restrict__ partial,
                         const BatchCtl *__restrict__ ctl, int nbatch, int it_end,
                         const int *__restrict__ nonfinite, unsigned long long *__restrict__ stats_out,
                         double *__restrict__ host_out, int out_doubles) {
  finish_body(sums, N, partial, ctl, nbatch, it_end, nonfinite, stats_out);
  if (host_out) {
    __threadfence();
    __syncthreads();
    copy_result_to_host(host_out, partial, N, (int)(reinterpret_cast<const double *>(stats_out) - partial), out_doubles);
  }
}

// ---------------------------------------------------------------------------------------------
// k_reduce (round 5): assembly + block reduction (assemble_body), then -- in the block that finishes LAST -- the fixed-order
// sum of the block partials (what k_final did: entry e by one wave, lane b takes blocks b, b + 64, ..., xor butterfly) and
// k_finish's suffix sum / counter gathering, and the result written both to the device buffer (collectives read it there) and
// straight into the pinned host buffer: one launch and no copy command where the evaluation's tail was three launches and a
// device-to-host copy (64 us + 10 us of the 490 us a reference-scale callback took).  Same additions in the same order as the
// three kernels: same bits.  `ticket` counts finished blocks; the last block resets it.
// fuse == 0 (grids of more than 4 blocks): assembly only -- one block summing 19N+1 entries over hundreds of block partials
// is a serial tail (57 us at C3's 512 blocks against k_final's 5 us with one wave PER entry): k_final and k_finish follow
// as launches of their own there, where two launches are nothing against the evaluation.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_reduce(const TrajDev *__restrict__ trg, const double *__restrict__ px_, const double *__restrict__ py_, int P,
         const double *__restrict__ res_sdf, const double *__restrict__ res_t, const double *__restrict__ res_gx,
         const double *__restrict__ res_gy, double safety_hor, double weight_p, double *__restrict__ block_partials,
         int *__restrict__ nonfinite, double *__restrict__ out, const BatchCtl *__restrict__ ctl, int nbatch, int it_end,
         int out_partial, int out_doubles, unsigned *__restrict__ ticket, double *__restrict__ host_out, int fuse) {
  extern __shared__ double asm_lds[];
  __shared__ unsigned s_last;
  const bool one_block = gridDim.x == 1;   // up to 256 points (the reference's demo maps give 101 .. 139): nothing to wait for
  assemble_body(trg, px_, py_, P, res_sdf, res_t, res_gx, res_gy, safety_hor, weight_p, block_partials, nonfinite, asm_lds, fuse && one_block);
  if (!fuse) return;
  const int N = trg->N;
  const int plen = 19 * N + 1;
  const int nblocks = (int)gridDim.x;
  double *sums = asm_lds + traj_lds_doubles(N);   // (the accumulator rows are free again: plen <= 4 plen doubles)
  if (!one_block) {
  __threadfence();   // this block's partials (and its non-finite count) are visible device-wide before its ticket is
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(ticket, 1u) == gridDim.x - 1u) ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // (other blocks wrote the partials: read at device scope, past this CU's vector cache)
  auto part = [&](int e, int b) { return __hip_atomic_load(&block_partials[(size_t)e * nblocks + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  {
    // Up to 1024 points (the reference's own scale): one THREAD per entry.  k_final's wave leaves in lane 0, for lane
    // values p_b = 0.0 + partial[e][b] (b < nblocks <= 4, zero beyond), the tree ((p0 + p2) + (p1 + p3)) -- the xor steps
    // 32 .. 4 only add zeros to lanes 0 .. 3 -- so the same bits come from four independent loads per thread instead of a
    // dependent load + butterfly per entry, one entry after the other (that loop cost 60 us of a 480 us callback).
    for (int e = threadIdx.x; e < plen; e += blockDim.x) {
      double p[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) p[b] = (b < nblocks) ? 0.0 + part(e, b) : 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) { p[0] += 0.0; p[1] += 0.0; p[2] += 0.0; p[3] += 0.0; }   // xor 32, 16, 8, 4: the partners hold zeros
      sums[e] = (p[0] + p[2]) + (p[1] + p[3]);                                             // xor 2, then xor 1
    }
  }
  }  // !one_block (one block: assemble_body left sums[e] = 0.0 + its partial in place)
  __syncthreads();
  finish_body(sums, N, out, ctl, nbatch, it_end, nonfinite, reinterpret_cast<unsigned long long *>(out + out_partial));
  __threadfence();
  __syncthreads();
  if (host_out) copy_result_to_host(host_out, out, N, out_partial, out_doubles);
  if (threadIdx.x == 0 && !one_block) *ticket = 0u;
}

// ---------------------------------------------------------------------------------------------
// Query-point upload on the device (svsdf_set_points / svsdf_set_points_device): bounding box, Morton keys with the
// input index in the low 32 bits, radix sort of the Morton half (host: hipcub), stripe gather into the SoA arrays.
// The key formula is the host planner's (svsdf_shard_plan), operation for operation, so both give the same order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned part1by1_dev(unsigned x) {
  x &= 0x0000ffffu;
  x = (x ^ (x << 8)) & 0x00ff00ffu;
  x = (x ^ (x << 4)) & 0x0f0f0f0fu;
  x = (x ^ (x << 2)) & 0x33333333u;
  x = (x ^ (x << 1)) & 0x55555555u;
  return x;
}

// per-block partial bounding boxes [xmin, xmax, ymin, ymax] + count of non-finite coordinates
__global__ void __launch_bounds__(kBlock)
k_points_bbox(const double *__restrict__ xyz, size_t P, double *__restrict__ part, int *__restrict__ nonfinite) {
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  double xmin = inf, xmax = -inf, ymin = inf, ymax = -inf;
  int bad = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (size_t)gridDim.x * blockDim.x) {
    const double x = xyz[3 * i], y = xyz[3 * i + 1];
    if (!(fabs(x) < inf) || !(fabs(y) < inf)) ++bad;   // NaN or inf
    if (x < xmin) xmin = x;
    if (x > xmax) xmax = x;
    if (y < ymin) ymin = y;
    if (y > ymax) ymax = y;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    xmin = fmin(xmin, __shfl_xor(xmin, m, 64)); xmax = fmax(xmax, __shfl_xor(xmax, m, 64));
    ymin = fmin(ymin, __shfl_xor(ymin, m, 64)); ymax = fmax(ymax, __shfl_xor(ymax, m, 64));
    bad += __shfl_xor(bad, m, 64);
  }
  __shared__ double s[kBlock / 64][4];
  __shared__ int sb[kBlock / 64];
  if ((threadIdx.x & 63) == 0) {
    s[threadIdx.x >> 6][0] = xmin; s[threadIdx.x >> 6][1] = xmax; s[threadIdx.x >> 6][2] = ymin; s[threadIdx.x >> 6][3] = ymax;
    sb[threadIdx.x >> 6] = bad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) {
      s[0][0] = fmin(s[0][0], s[w][0]); s[0][1] = fmax(s[0][1], s[w][1]);
      s[0][2] = fmin(s[0][2], s[w][2]); s[0][3] = fmax(s[0][3], s[w][3]);
      sb[0] += sb[w];
    }
    for (int q = 0; q < 4; ++q) part[4 * blockIdx.x + q] = s[0][q];
    if (sb[0]) atomicAdd(nonfinite, sb[0]);
  }
}

__global__ void __launch_bounds__(kBlock)
k_points_keys(const double *__restrict__ xyz, size_t P, double xmin, double ymin, double ext, int keep_order,
              unsigned long long *__restrict__ keys) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long code = 0ull;
    if (!keep_order) {
      const double fx = (xyz[3 * i] - xmin) / ext, fy = (xyz[3 * i + 1] - ymin) / ext;
      const unsigned qx = (unsigned)fmin(65535.0, fmax(0.0, fx * 65535.0));
      const unsigned qy = (unsigned)fmin(65535.0, fmax(0.0, fy * 65535.0));
      code = (unsigned long long)(part1by1_dev(qx) | (part1by1_dev(qy) << 1));
    }
    keys[i] = (code << 32) | (unsigned long long)(i & 0xffffffffull);
  }
}

// stripe rk of ws of the sorted order -> SoA coordinates + original indices
__global__ void __launch_bounds__(kBlock)
k_points_gather(const double *__restrict__ xyz, const unsigned long long *__restrict__ keys, size_t P, int rk, int ws,
                size_t Ps, double *__restrict__ px, double *__restrict__ py, long long *__restrict__ idx_out) {
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < Ps; j += (size_t)gridDim.x * blockDim.x) {
    const size_t k = (size_t)rk + j * (size_t)ws;
    const size_t i = (size_t)(keys[k] & 0xffffffffull);
    px[j] = xyz[3 * i];
    py[j] = xyz[3 * i + 1];
    idx_out[j] = (long long)i;
  }
}

#endif  // SVSDF_API_TU

}  // namespace svsdf
