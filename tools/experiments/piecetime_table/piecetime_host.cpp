// piecetime_host.cpp -- test harness: the product's piece-local-time table (csrc/svsdf_piecetime.hpp, the functions
// k_prep and the solve kernels use) built and applied on the HOST against the reference's plain chain.
//   hipcc -x hip --cuda-host-only -O2 -ffp-contract=off -shared -fPIC piecetime_host.cpp -o libpiecetime_host.so
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

#include "svsdf_piecetime.hpp"

using namespace svsdf;

extern "C" {

// Builds the table for durations T[N] over [0, tmax] and checks n query times.  Returns the number of queries whose
// (piece, local time) from the table differs from the chain's (bit compare); out[0] = intervals, out[1] = intervals that
// take the chain, out[2] = queries that fell into such intervals, out[3] = table usable (0: binade range too wide or too
// many intervals).
long long pt_check(const double *T, int N, double tmax, const double *tq, size_t n, long long *out, double *first_bad) {
  double tmin = T[0];
  for (int i = 1; i < N; ++i) tmin = std::min(tmin, T[i]);
  const int e0 = (int)std::floor(std::log2(tmin));
  out[3] = (std::ldexp(1.0, e0 + kPtMaxPow) >= tmax) ? 1 : 0;
  std::vector<double> th;
  th.push_back(0.0);
  for (int q = 0; q < pt_num_thresholds(N); ++q) {
    const double v = pt_threshold(T, N, e0, tmax, q);
    if (std::isfinite(v)) th.push_back(v);
  }
  std::sort(th.begin(), th.end());
  th.erase(std::unique(th.begin(), th.end()), th.end());
  std::vector<PtSeg> seg(th.size());
  long long nchain = 0;
  for (size_t k = 0; k < th.size(); ++k) { seg[k] = pt_segment(T, N, th[k]); nchain += seg[k].chain; }
  out[0] = (long long)th.size();
  out[1] = nchain;
  if ((long long)th.size() > kPtMaxSeg) out[3] = 0;
  long long bad = 0, inchain = 0;
  for (size_t q = 0; q < n; ++q) {
    const double t = tq[q];
    const size_t k = (size_t)(std::upper_bound(th.begin(), th.end(), t) - th.begin()) - 1;
    double s_ref;
    const int i_ref = pt_chain(T, N, t, s_ref);
    if (seg[k].chain) { ++inchain; continue; }
    const double s = pt_apply(seg[k], t);
    if (seg[k].piece != i_ref || pt_bits(s) != pt_bits(s_ref)) {
      if (bad == 0 && first_bad) { first_bad[0] = t; first_bad[1] = s; first_bad[2] = s_ref; first_bad[3] = (double)seg[k].piece; first_bad[4] = (double)i_ref; first_bad[5] = th[k]; }
      ++bad;
    }
  }
  out[2] = inchain;
  return bad;
}

}  // extern "C"
