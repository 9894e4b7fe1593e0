// svsdf_lbfgs.hpp -- host-side limited-memory BFGS driver.
//
// Lets the accelerated callback be driven end to end without the reference's Fortran LMBM
// (src/utils/include/utils/lmbm.h) or its header-only L-BFGS (src/utils/include/utils/lbfgs.hpp:290-790).
// Written from the published algorithms, not from those sources:
//   * two-loop recursion, Nocedal & Wright, "Numerical Optimization", Alg. 7.4;
//   * weak-Wolfe bracketing line search for nonsmooth objectives, Lewis & Overton, "Nonsmooth optimization
//     via quasi-Newton methods", Math. Program. 141 (2013), Alg. 2.6 (the swept-volume penalty is piecewise
//     smooth: t* jumps between local minima, so the strong-Wolfe search of classical L-BFGS stalls);
//   * cautious update, Li & Fukushima, SIAM J. Optim. 11 (2001): skip the pair unless y's > eps*|g|*s's.
// Parameter names, defaults and return codes follow lbfgs.hpp's lbfgs_parameter_t (:33-160) so a
// maintainer can carry a tuned parameter block over unchanged.  O(n*m) dense vector algebra on n <= a few
// hundred variables: stays on the host.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/svsdf_c.h"

namespace svsdf_host {

inline double dot(const std::vector<double> &a, const std::vector<double> &b) {
  double s = 0.0;
  for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
  return s;
}
inline double inf_norm(const std::vector<double> &a) {
  double m = 0.0;
  for (double v : a) m = std::max(m, std::fabs(v));
  return m;
}

struct LbfgsResult {
  int status = 0, iterations = 0, evaluations = 0;
  double fx = 0.0;
};

inline int lbfgs_check_params(int n, const svsdf_lbfgs_params &p) {
  if (n <= 0) return SVSDF_LBFGSERR_INVALID_N;
  if (p.mem_size <= 0) return SVSDF_LBFGSERR_INVALID_MEMSIZE;
  if (p.g_epsilon < 0.0) return SVSDF_LBFGSERR_INVALID_GEPSILON;
  if (p.past < 0) return SVSDF_LBFGSERR_INVALID_TESTPERIOD;
  if (p.delta < 0.0) return SVSDF_LBFGSERR_INVALID_DELTA;
  if (p.min_step < 0.0) return SVSDF_LBFGSERR_INVALID_MINSTEP;
  if (p.max_step < p.min_step) return SVSDF_LBFGSERR_INVALID_MAXSTEP;
  if (!(p.f_dec_coeff > 0.0 && p.f_dec_coeff < 1.0)) return SVSDF_LBFGSERR_INVALID_FDECCOEFF;
  if (!(p.s_curv_coeff > p.f_dec_coeff && p.s_curv_coeff < 1.0)) return SVSDF_LBFGSERR_INVALID_SCURVCOEFF;
  if (!(p.machine_prec > 0.0)) return SVSDF_LBFGSERR_INVALID_MACHINEPREC;
  if (p.max_linesearch <= 0) return SVSDF_LBFGSERR_INVALID_MAXLINESEARCH;
  return 0;
}

// Minimise f over R^n.  eval(instance, x, g, n) returns f(x) and overwrites g (the LMBM callback type,
// lmbm.h:206-209).  progress (may be null) returning non-zero cancels.
inline LbfgsResult lbfgs_minimize(int n, double *x_io, svsdf_evaluate_t eval, void *instance,
                                  svsdf_progress_t progress, void *progress_user, const svsdf_lbfgs_params &p) {
  LbfgsResult res;
  res.status = lbfgs_check_params(n, p);
  if (res.status) return res;
  const int m = p.mem_size;
  std::vector<double> x(x_io, x_io + n), g(n), xp(n), gp(n), d(n), q(n);
  std::vector<std::vector<double>> S(m, std::vector<double>(n)), Y(m, std::vector<double>(n));
  std::vector<double> rho(m), alpha(m), past_f(std::max(p.past, 1));
  int stored = 0, head = 0;

  double fx = eval(instance, x.data(), g.data(), n);
  res.evaluations = 1;
  res.fx = fx;
  if (!std::isfinite(fx)) { res.status = SVSDF_LBFGSERR_INVALID_FUNCVAL; return res; }
  if (p.past > 0) past_f[0] = fx;
  if (inf_norm(g) / std::max(1.0, inf_norm(x)) <= p.g_epsilon) { res.status = SVSDF_LBFGS_CONVERGENCE; return res; }
  for (int i = 0; i < n; ++i) d[i] = -g[i];
  double step = 1.0 / std::sqrt(dot(d, d));

  for (int k = 1;; ++k) {
    xp = x;
    gp = g;
    const double f0 = fx, dg0 = dot(gp, d);
    if (!(dg0 < 0.0)) { res.status = SVSDF_LBFGSERR_INCREASEGRADIENT; break; }
    // ---- Lewis-Overton weak-Wolfe search on phi(t) = f(xp + t d) ----
    if (step < p.min_step) step = p.min_step;
    if (step > p.max_step) step = p.max_step;
    double lo = 0.0, hi = p.max_step;
    bool bracketed = false;
    int ls = 0, ls_status = 0;
    for (;;) {
      for (int i = 0; i < n; ++i) x[i] = xp[i] + step * d[i];
      fx = eval(instance, x.data(), g.data(), n);
      ++res.evaluations;
      ++ls;
      if (!std::isfinite(fx)) { ls_status = SVSDF_LBFGSERR_INVALID_FUNCVAL; break; }
      if (fx > f0 + step * p.f_dec_coeff * dg0) {          // sufficient decrease fails: shrink from above
        hi = step;
        bracketed = true;
      } else if (dot(g, d) < p.s_curv_coeff * dg0) {        // still descending steeply: grow from below
        lo = step;
      } else {
        break;                                              // weak Wolfe point
      }
      if (ls >= p.max_linesearch) { ls_status = SVSDF_LBFGSERR_MAXIMUMLINESEARCH; break; }
      if (bracketed && (hi - lo) < p.machine_prec * hi) { ls_status = SVSDF_LBFGSERR_WIDTHTOOSMALL; break; }
      step = bracketed ? 0.5 * (lo + hi) : 2.0 * step;
      if (step < p.min_step) { ls_status = SVSDF_LBFGSERR_MINIMUMSTEP; break; }
      if (step > p.max_step) { ls_status = SVSDF_LBFGSERR_MAXIMUMSTEP; break; }
    }
    if (ls_status) {  // restore the best known point (the start of this iteration)
      x = xp;
      g = gp;
      fx = f0;
      res.status = ls_status;
      break;
    }
    res.iterations = k;
    if (progress && progress(progress_user, x.data(), g.data(), fx, step, n, k, ls)) {
      res.status = SVSDF_LBFGS_CANCELED;
      break;
    }
    if (inf_norm(g) / std::max(1.0, inf_norm(x)) <= p.g_epsilon) { res.status = SVSDF_LBFGS_CONVERGENCE; break; }
    if (p.past > 0) {  // relative decrease over the last `past` iterations
      if (k >= p.past) {
        const double rate = std::fabs(past_f[k % p.past] - fx) / std::max(1.0, std::fabs(fx));
        if (rate < p.delta) { res.status = SVSDF_LBFGS_STOP; break; }
      }
      past_f[k % p.past] = fx;
    }
    if (p.max_iterations != 0 && k >= p.max_iterations) { res.status = SVSDF_LBFGSERR_MAXIMUMITERATION; break; }

    // ---- curvature pair, cautious update ----
    std::vector<double> &s = S[head], &y = Y[head];
    for (int i = 0; i < n; ++i) { s[i] = x[i] - xp[i]; y[i] = g[i] - gp[i]; }
    const double ys = dot(y, s), yy = dot(y, y), ss = dot(s, s);
    if (ys > p.cautious_factor * std::sqrt(dot(gp, gp)) * ss && yy > 0.0) {
      rho[head] = 1.0 / ys;
      head = (head + 1) % m;
      stored = std::min(stored + 1, m);
    }
    // ---- two-loop recursion: d = -H g ----
    for (int i = 0; i < n; ++i) d[i] = -g[i];
    if (stored > 0) {
      int j = head;
      for (int c = 0; c < stored; ++c) {
        j = (j + m - 1) % m;
        alpha[j] = rho[j] * dot(S[j], d);
        for (int i = 0; i < n; ++i) d[i] -= alpha[j] * Y[j][i];
      }
      const int last = (head + m - 1) % m;
      const double gamma = 1.0 / (rho[last] * dot(Y[last], Y[last]));   // s'y / y'y scaling of H0
      for (int i = 0; i < n; ++i) d[i] *= gamma;
      for (int c = 0; c < stored; ++c) {
        const double beta = rho[j] * dot(Y[j], d);
        for (int i = 0; i < n; ++i) d[i] += (alpha[j] - beta) * S[j][i];
        j = (j + 1) % m;
      }
    }
    step = 1.0;
  }
  std::copy(x.begin(), x.end(), x_io);
  res.fx = fx;
  return res;
}

}  // namespace svsdf_host
