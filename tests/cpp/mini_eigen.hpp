// mini_eigen.hpp -- TEST-ONLY stand-in for the two Eigen types the lbfgs_evaluate_t adapter touches
// (this image ships no Eigen).  Only what include/svsdf_traj_optimizer.hpp uses: data(), size(), column-major
// storage for MatrixX3d.  A real build includes <Eigen/Core> instead (SVSDF_EIGEN_HEADER unset).
#pragma once
#include <cstddef>
#include <vector>

namespace Eigen {
class VectorXd {
 public:
  VectorXd() = default;
  explicit VectorXd(std::ptrdiff_t n) : v_((std::size_t)n, 0.0) {}
  double *data() { return v_.data(); }
  const double *data() const { return v_.data(); }
  std::ptrdiff_t size() const { return (std::ptrdiff_t)v_.size(); }
  double &operator()(std::ptrdiff_t i) { return v_[(std::size_t)i]; }
  double operator()(std::ptrdiff_t i) const { return v_[(std::size_t)i]; }

 private:
  std::vector<double> v_;
};
class MatrixX3d {   // rows x 3, column-major like Eigen's default
 public:
  MatrixX3d() = default;
  explicit MatrixX3d(std::ptrdiff_t rows) : rows_(rows), v_((std::size_t)rows * 3, 0.0) {}
  double *data() { return v_.data(); }
  const double *data() const { return v_.data(); }
  std::ptrdiff_t rows() const { return rows_; }
  double &operator()(std::ptrdiff_t r, std::ptrdiff_t c) { return v_[(std::size_t)(c * rows_ + r)]; }
  double operator()(std::ptrdiff_t r, std::ptrdiff_t c) const { return v_[(std::size_t)(c * rows_ + r)]; }

 private:
  std::ptrdiff_t rows_ = 0;
  std::vector<double> v_;
};
}  // namespace Eigen
