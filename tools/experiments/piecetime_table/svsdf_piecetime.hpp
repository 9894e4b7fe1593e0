// svsdf_piecetime.hpp -- the reference's piece-local time, bit for bit, without walking its chain of subtractions.
//
// Trajectory::locatePieceIdx (trajectory.hpp:498-516) turns a global time t into (piece i, local time s) by
//     r_0 = t;  while (r_j > T_j) r_{j+1} = fl(r_j - T_j);        i = first j with !(r_j > T_j),  s = r_i
// i.e. up to N - 1 ROUNDED subtractions: for generic durations (what an optimiser produces) s differs from
// t - (T_0 + ... + T_{i-1}) by up to i ulp(t), and the flat stretches of SDF(t) amplify that (DESIGN.md "Oracle").  Every
// SDF evaluation of the solve kernels needs (i, s); walking the chain costs +20 ... +30 % of an evaluation's time.
//
// What this file exploits: for all t of a suitable interval the whole chain is the SAME short sequence of subtractions.
//   * Each r_j(t) is a non-decreasing function of t (fl is monotone), so "the chain reaches step j" and "r_j >= 2^e" flip
//     at single thresholds of t, found by bisection over the doubles.  Between two consecutive thresholds the piece index
//     i and the binade of every intermediate r_j are constant.
//   * A step r_{j+1} = fl(r_j - T_j) whose result lands in the binade of T_j or below is exact.  Otherwise its rounding
//     error depends on (r_j - T_j) mod ulp(r_{j+1}); r_j is a multiple of ulp(r_j) >= ulp(r_{j+1}), so that remainder is
//     -T_j mod ulp(r_{j+1}): the same for every t of the interval -- a constant error -- EXCEPT for a tie (remainder =
//     half an ulp) between two r's of one binade, where round-to-even follows the last bit of r_j.  Such a step is kept as
//     a real rounded subtraction; at most one exists per binade level.
//   * Runs of constant-error steps collapse into ONE exact subtraction of a constant C = r_a(t0) - r_b(t0) (t0 = the
//     interval's first double).  C needs up to ~60 bits, so it is applied as two exact subtractions (hi, lo).
// So per interval: s = ((((t - K0) - K1) - K2) - K3) - K4 with K = (hi, lo) [tie-free] or (hi, lo, T_tie, hi', lo') [one
// tie], every operation an ordinary IEEE subtraction whose result equals the reference's.  Intervals that would need more
// (two tie steps, or a t beyond the last piece) are flagged and take the chain.  A trajectory of N pieces has about
// N (1 + number of binades between min T and the duration) intervals (192 for 32 pieces of ~2.5 s); a uniform grid over
// [0, duration] finds the interval.  The table is rebuilt per trajectory by k_prep (one thread per threshold).
//
// tests/cpp/piecetime_host.cpp runs these very functions on the CPU against the plain chain (millions of random t on
// random duration sets, including ties by construction).
#pragma once
#include <hip/hip_runtime.h>

namespace svsdf {

constexpr int kPtMaxSeg = 640;      // intervals per trajectory the table can hold (else: the chain runs)
constexpr int kPtOps = 5;
constexpr int kPtCells = 1024;      // uniform grid over [0, dur] -> first interval of the cell
constexpr int kPtMaxPow = 12;       // binade boundaries 2^e tracked per step

struct PtSeg {
  double K[kPtOps];   // s = ((((t - K0) - K1) - K2) - K3) - K4
  int piece;          // i
  int chain;          // 1: this interval takes the reference's chain (two tie steps / past the last piece)
};

// the reference's chain (TRJ:498-516)
__host__ __device__ inline int pt_chain(const double *T, int N, double t, double &s) {
  int i = 0;
  double dur = 0.0;
  for (; i < N && t > (dur = T[i]); ++i) t -= dur;
  if (i == N) { --i; t += T[i]; }
  s = t;
  return i;
}

// does the chain started at t reach step j (i.e. r_k > T_k for all k < j), and is r_j >= bound?
__host__ __device__ inline bool pt_reach_ge(const double *T, int N, double t, int j, double bound) {
  double r = t;
  for (int k = 0; k < j; ++k) {
    if (!(r > T[k])) return false;
    r -= T[k];
  }
  return r >= bound;
}
// ... and r_j > T_j (the chain goes on to step j + 1)
__host__ __device__ inline bool pt_goes_on(const double *T, int N, double t, int j) {
  double r = t;
  for (int k = 0; k < j; ++k) {
    if (!(r > T[k])) return false;
    r -= T[k];
  }
  return r > T[j];
}

__host__ __device__ inline double pt_from_bits(unsigned long long b) {
  union { unsigned long long u; double d; } x;
  x.u = b;
  return x.d;
}
__host__ __device__ inline unsigned long long pt_bits(double d) {
  union { unsigned long long u; double d; } x;
  x.d = d;
  return x.u;
}

// Threshold number q of a trajectory (q in [0, pt_num_thresholds)): the smallest double t in [0, tmax] at which a
// monotone predicate turns true, or +inf when it is still false at tmax.
//   q <  N                : the chain goes on past step q  (piece boundary)
//   q >= N, (j, m) = ((q - N) / kPtMaxPow, (q - N) % kPtMaxPow):  r_j >= 2^(e0 + m), e0 = binade of the smallest duration
__host__ __device__ inline int pt_num_thresholds(int N) { return N + N * kPtMaxPow; }
__host__ __device__ inline double pt_threshold(const double *T, int N, int e0, double tmax, int q) {
  const double inf = pt_from_bits(0x7ff0000000000000ull);
  int j, kind;
  double bound = 0.0;
  if (q < N) { j = q; kind = 0; }
  else {
    j = (q - N) / kPtMaxPow;
    kind = 1;
    bound = ldexp(1.0, e0 + 1 + (q - N) % kPtMaxPow);
    if (bound > tmax) return inf;
  }
  auto pred = [&](double t) { return kind == 0 ? pt_goes_on(T, N, t, j) : pt_reach_ge(T, N, t, j, bound); };
  if (!pred(tmax)) return inf;
  if (pred(0.0)) return 0.0;
  unsigned long long lo = 0ull, hi = pt_bits(tmax);   // non-negative doubles order like their bit patterns
  while (hi - lo > 1ull) {                            // pred(lo) false, pred(hi) true
    const unsigned long long mid = lo + (hi - lo) / 2ull;
    if (pred(pt_from_bits(mid))) hi = mid; else lo = mid;
  }
  return pt_from_bits(hi);
}

// The interval that starts at t0 (its first double): piece and the constants of its subtraction sequence.
__host__ __device__ inline PtSeg pt_segment(const double *T, int N, double t0) {
  PtSeg sg;
  for (int k = 0; k < kPtOps; ++k) sg.K[k] = 0.0;
  sg.piece = 0;
  sg.chain = 0;
  // walk the chain at t0, looking for tie steps whose outcome follows the last bit of r_j
  double r = t0;
  int i = 0, nties = 0;
  double a_tie = 0.0, b_tie = 0.0, T_tie = 0.0;   // r before / after the tie step
  for (; i < N && r > T[i]; ++i) {
    const double a = r, b = T[i];
    const double d = a - b;                 // a > b > 0
    const double z = d - a;                 // Fast2Sum (|a| >= |b|): exact
    const double err = (-b) - z;            // (a - b) = d + err exactly
    if (err != 0.0) {
      const int ea = (int)((pt_bits(a) >> 52) & 0x7ffull), ed = (int)((pt_bits(d) >> 52) & 0x7ffull);
      const double half_ulp = ldexp(1.0, ed - 1023 - 53);
      if (ea == ed && (err == half_ulp || err == -half_ulp)) {
        ++nties;
        a_tie = a; b_tie = d; T_tie = b;
      }
    }
    r = d;
  }
  if (i == N || nties > 1) {   // past the last piece (TRJ:509-513 adds the last duration back), or two tie steps
    sg.chain = 1;
    double s;
    sg.piece = pt_chain(T, N, t0, s);
    return sg;
  }
  sg.piece = i;
  const double s0 = r;
  if (nties == 0) {
    // s(t) = t - C on the whole interval, C = t0 - s0 (up to ~60 significant bits): hi + lo, both subtractions exact
    const double hi = t0 - s0;
    sg.K[0] = hi;
    sg.K[1] = (t0 - hi) - s0;
  } else {
    const double hi = t0 - a_tie;          // r_tie(t) = t - C1
    sg.K[0] = hi;
    sg.K[1] = (t0 - hi) - a_tie;
    sg.K[2] = T_tie;                       // the tie step itself: a rounded subtraction
    const double hi2 = b_tie - s0;         // s(t) = r_{tie+1}(t) - C2
    sg.K[3] = hi2;
    sg.K[4] = (b_tie - hi2) - s0;
  }
  return sg;
}

// local time from an interval's constants
__host__ __device__ __forceinline__ double pt_apply(const PtSeg &sg, double t) {
  return ((((t - sg.K[0]) - sg.K[1]) - sg.K[2]) - sg.K[3]) - sg.K[4];
}

}  // namespace svsdf
