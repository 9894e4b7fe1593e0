// svsdf_minco.hpp -- host-side minimum-jerk (s = 3) non-uniform-time MINCO spline:
// waypoints + durations -> quintic coefficients, jerk energy and its partials, and the
// adjoint propagation of d(cost)/d(coeffs, T) to d(cost)/d(waypoints, T).
//
// Behavioural spec: reference src/utils/include/utils/minco.hpp (MNC) MINCO_S3NU (:397-655)
// on top of BandedSystem (:43-198), and the tau<->T / xi<->P maps of
// src/planner_algorithm/include/planner_algorithm/back_end_optimizer.hpp (BEO) :174-314.
// This is the O(N) host work on both sides of the GPU hot loop; it is
// product code (used by svsdf_lmbm_evaluate), written independently of oracle/.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace svsdf_host {

// Square band matrix without pivoting, row-band storage: a(i, j) for |i - j| <= bw.
class BandLU {
 public:
  void reset(int n, int bw) {
    n_ = n; bw_ = bw;
    d_.assign((size_t)n * (2 * bw + 1), 0.0);
  }
  double &at(int i, int j) { return d_[(size_t)i * (2 * bw_ + 1) + (j - i + bw_)]; }
  double at(int i, int j) const { return d_[(size_t)i * (2 * bw_ + 1) + (j - i + bw_)]; }
  int n() const { return n_; }

  // In-place Doolittle LU (unit lower), no pivoting -- MNC:96-128 semantics.
  void factorize() {
    for (int k = 0; k + 1 < n_; ++k) {
      const int ilast = std::min(k + bw_, n_ - 1);
      const int jlast = std::min(k + bw_, n_ - 1);
      const double piv = at(k, k);
      for (int i = k + 1; i <= ilast; ++i) {
        double &l = at(i, k);
        if (l == 0.0) continue;
        l /= piv;
        for (int j = k + 1; j <= jlast; ++j) {
          const double u = at(k, j);
          if (u != 0.0) at(i, j) -= l * u;
        }
      }
    }
  }
  // Solve A X = B for `m` right-hand sides stored row-major in b (n x m).  MNC:133-163
  void solve(double *b, int m) const {
    for (int j = 0; j < n_; ++j) {
      const int ilast = std::min(j + bw_, n_ - 1);
      for (int i = j + 1; i <= ilast; ++i) {
        const double l = at(i, j);
        if (l == 0.0) continue;
        for (int c = 0; c < m; ++c) b[i * m + c] -= l * b[j * m + c];
      }
    }
    for (int j = n_ - 1; j >= 0; --j) {
      const double piv = at(j, j);
      for (int c = 0; c < m; ++c) b[j * m + c] /= piv;
      const int ifirst = std::max(0, j - bw_);
      for (int i = ifirst; i < j; ++i) {
        const double u = at(i, j);
        if (u == 0.0) continue;
        for (int c = 0; c < m; ++c) b[i * m + c] -= u * b[j * m + c];
      }
    }
  }
  // Solve A^T X = B.  MNC:168-197
  void solve_transposed(double *b, int m) const {
    for (int j = 0; j < n_; ++j) {
      const double piv = at(j, j);
      for (int c = 0; c < m; ++c) b[j * m + c] /= piv;
      const int ilast = std::min(j + bw_, n_ - 1);
      for (int i = j + 1; i <= ilast; ++i) {
        const double u = at(j, i);
        if (u == 0.0) continue;
        for (int c = 0; c < m; ++c) b[i * m + c] -= u * b[j * m + c];
      }
    }
    for (int j = n_ - 1; j >= 0; --j) {
      const int ifirst = std::max(0, j - bw_);
      for (int i = ifirst; i < j; ++i) {
        const double l = at(j, i);
        if (l == 0.0) continue;
        for (int c = 0; c < m; ++c) b[i * m + c] -= l * b[j * m + c];
      }
    }
  }

 private:
  int n_ = 0, bw_ = 0;
  std::vector<double> d_;
};

// k!/(k-d)! : coefficient of s^(k-d) in the d-th derivative of s^k
inline double falling(int k, int d) {
  double f = 1.0;
  for (int q = 0; q < d; ++q) f *= (double)(k - q);
  return f;
}

class MincoS3 {
 public:
  // head/tail: 3x3 column-major (col 0 pos, col 1 vel, col 2 acc).  MNC:418-433
  void set_conditions(const double head[9], const double tail[9], int pieces) {
    N_ = pieces;
    std::copy(head, head + 9, head_);
    std::copy(tail, tail + 9, tail_);
  }
  int pieces() const { return N_; }

  // waypoints: 3 x (N-1) column-major; T: N durations.  MNC:435-513
  void set_parameters(const double *waypoints, const double *T) {
    const int N = N_, n = 6 * N;
    pw_.assign((size_t)N * 6, 1.0);  // pw_[i*6 + p] = T_i^p with the reference's product tree
    for (int i = 0; i < N; ++i) {
      double *p = &pw_[(size_t)i * 6];
      p[1] = T[i];
      p[2] = p[1] * p[1];
      p[3] = p[2] * p[1];
      p[4] = p[2] * p[2];
      p[5] = p[4] * p[1];
    }
    A_.reset(n, 6);
    b_.assign((size_t)n * 3, 0.0);
    // head boundary: derivative d of piece 0 at s = 0 equals head state d
    for (int d = 0; d < 3; ++d) {
      A_.at(d, d) = falling(d, d);
      for (int c = 0; c < 3; ++c) b_[d * 3 + c] = head_[d * 3 + c];
    }
    for (int i = 0; i + 1 < N; ++i) {
      const double *p = &pw_[(size_t)i * 6];
      const int c0 = 6 * i, n0 = 6 * (i + 1);
      // continuity of jerk (d=3) and snap (d=4) across the knot
      for (int d = 3; d <= 4; ++d) {
        const int row = 6 * i + d;
        for (int k = d; k < 6; ++k) A_.at(row, c0 + k) = falling(k, d) * p[k - d];
        A_.at(row, n0 + d) = -falling(d, d);
      }
      // end of piece i passes through waypoint i
      for (int k = 0; k < 6; ++k) A_.at(6 * i + 5, c0 + k) = p[k];
      for (int c = 0; c < 3; ++c) b_[(6 * i + 5) * 3 + c] = waypoints[i * 3 + c];
      // continuity of position, velocity, acceleration
      for (int d = 0; d <= 2; ++d) {
        const int row = 6 * i + 6 + d;
        for (int k = d; k < 6; ++k) A_.at(row, c0 + k) = falling(k, d) * p[k - d];
        A_.at(row, n0 + d) = -falling(d, d);
      }
    }
    // tail boundary: derivative d of the last piece at s = T equals tail state d
    const double *p = &pw_[(size_t)(N - 1) * 6];
    for (int d = 0; d < 3; ++d) {
      const int row = 6 * N - 3 + d;
      for (int k = d; k < 6; ++k) A_.at(row, 6 * (N - 1) + k) = falling(k, d) * p[k - d];
      for (int c = 0; c < 3; ++c) b_[row * 3 + c] = tail_[d * 3 + c];
    }
    A_.factorize();
    A_.solve(b_.data(), 3);
  }

  // coefficient rows (6N x 3 row-major): row 6i+k = coefficient of s^k of piece i
  const std::vector<double> &coeffs() const { return b_; }
  void coeffs_colmajor(double *out) const {
    const int n = 6 * N_;
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < 3; ++c) out[(size_t)c * n + r] = b_[r * 3 + c];
  }

  static double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

  // jerk energy, MNC:530-543
  double energy() const {
    double e = 0.0;
    for (int i = 0; i < N_; ++i) {
      const double *p = &pw_[(size_t)i * 6];
      const double *c3 = &b_[(6 * i + 3) * 3], *c4 = &b_[(6 * i + 4) * 3], *c5 = &b_[(6 * i + 5) * 3];
      e += 36.0 * dot3(c3, c3) * p[1] + 144.0 * dot3(c4, c3) * p[2] + 192.0 * dot3(c4, c4) * p[3] +
           240.0 * dot3(c5, c3) * p[3] + 720.0 * dot3(c5, c4) * p[4] + 720.0 * dot3(c5, c5) * p[5];
    }
    return e;
  }
  // dE/dc (6N x 3 row-major), MNC:550-567
  void energy_grad_coeffs(double *g) const {
    for (int i = 0; i < N_; ++i) {
      const double *p = &pw_[(size_t)i * 6];
      for (int c = 0; c < 3; ++c) {
        const double c3 = b_[(6 * i + 3) * 3 + c], c4 = b_[(6 * i + 4) * 3 + c], c5 = b_[(6 * i + 5) * 3 + c];
        g[(6 * i + 0) * 3 + c] = 0.0;
        g[(6 * i + 1) * 3 + c] = 0.0;
        g[(6 * i + 2) * 3 + c] = 0.0;
        g[(6 * i + 3) * 3 + c] = 72.0 * c3 * p[1] + 144.0 * c4 * p[2] + 240.0 * c5 * p[3];
        g[(6 * i + 4) * 3 + c] = 144.0 * c3 * p[2] + 384.0 * c4 * p[3] + 720.0 * c5 * p[4];
        g[(6 * i + 5) * 3 + c] = 240.0 * c3 * p[3] + 720.0 * c4 * p[4] + 1440.0 * c5 * p[5];
      }
    }
  }
  // dE/dT, MNC:569-582
  void energy_grad_times(double *g) const {
    for (int i = 0; i < N_; ++i) {
      const double *p = &pw_[(size_t)i * 6];
      const double *c3 = &b_[(6 * i + 3) * 3], *c4 = &b_[(6 * i + 4) * 3], *c5 = &b_[(6 * i + 5) * 3];
      g[i] = 36.0 * dot3(c3, c3) + 288.0 * dot3(c4, c3) * p[1] + 576.0 * dot3(c4, c4) * p[2] +
             720.0 * dot3(c5, c3) * p[2] + 2880.0 * dot3(c5, c4) * p[3] + 3600.0 * dot3(c5, c5) * p[4];
    }
  }

  // Adjoint: (dCost/dc [6N x 3 row-major, destroyed], dCost/dT) -> (dCost/dq [3 x (N-1) col-major],
  // dCost/dT total).  MNC:584-654
  void propagate(double *adj, const double *partial_T, double *grad_q, double *grad_T) const {
    const int N = N_;
    A_.solve_transposed(adj, 3);
    for (int i = 0; i + 1 < N; ++i)
      for (int c = 0; c < 3; ++c) grad_q[i * 3 + c] = adj[(6 * i + 5) * 3 + c];
    for (int i = 0; i < N; ++i) {
      const double *p = &pw_[(size_t)i * 6];
      const double *cf = &b_[(size_t)6 * i * 3];
      // negative d-th derivative of piece i at its end time, d = 1..5
      double nd[6][3];
      for (int d = 1; d <= 5; ++d)
        for (int c = 0; c < 3; ++c) {
          double v = 0.0;
          for (int k = d; k < 6; ++k) {
            const double term = (k == d) ? falling(k, d) * cf[k * 3 + c] : falling(k, d) * p[k - d] * cf[k * 3 + c];
            v = (k == d) ? term : v + term;
          }
          nd[d][c] = -v;
        }
      double s = 0.0;
      if (i + 1 < N) {
        // rows 6i+3 .. 6i+8 of A depend on T_i: jerk, snap, waypoint, pos, vel, acc conditions,
        // whose T-derivatives are snap, crackle, vel, vel, acc, jerk of the piece end.
        const int rows[6] = {4, 5, 1, 1, 2, 3};
        for (int c = 0; c < 3; ++c)
          for (int r = 0; r < 6; ++r) s += nd[rows[r]][c] * adj[(6 * i + 3 + r) * 3 + c];
      } else {
        const int rows[3] = {1, 2, 3};
        for (int c = 0; c < 3; ++c)
          for (int r = 0; r < 3; ++r) s += nd[rows[r]][c] * adj[(6 * N - 3 + r) * 3 + c];
      }
      grad_T[i] = s + partial_T[i];
    }
  }

 private:
  int N_ = 0;
  double head_[9] = {0}, tail_[9] = {0};
  BandLU A_;
  std::vector<double> b_;
  std::vector<double> pw_;
};

// tau -> T (BEO:213-226) and its inverse (BEO:228-241), dT/dtau chain (BEO:268-289)
inline double tau_to_T(double tau) {
  return tau > 0.0 ? ((0.5 * tau + 1.0) * tau + 1.0) : 1.0 / ((0.5 * tau - 1.0) * tau + 1.0);
}
inline double T_to_tau(double T) {
  return T > 1.0 ? (std::sqrt(2.0 * T - 1.0) - 1.0) : (1.0 - std::sqrt(2.0 / T - 1.0));
}
inline double grad_T_to_tau(double tau, double gT) {
  if (tau > 0) return gT * (tau + 1.0);
  const double den = (0.5 * tau - 1.0) * tau + 1.0;
  return gT * (1.0 - tau) / (den * den);
}

}  // namespace svsdf_host
