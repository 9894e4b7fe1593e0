// Drives the Eigen-typed entry points of the C++ mirror the way the reference's L-BFGS does
// (src/utils/include/utils/lbfgs.hpp:213-216: `double (*lbfgs_evaluate_t)(void *instance, const Eigen::VectorXd &x,
// Eigen::VectorXd &g, double &p_cost)`, called at lbfgs.hpp:342,566,761,770) and the Eigen overload of
// addSaftyPenaOnSweptVolumeParallelTrueSDF (back_end_optimizer.hpp:774-779).
// Input (stdin) as traj_optimizer_driver.cpp plus 18N coefficient doubles ((6N) x 3 column-major).
// Output: f p_cost cost_pos, g[0..n), then the penalty's cost, gradT[0..N), gradC[0..18N).
#include <cstdio>
#include <vector>

#include "svsdf_traj_optimizer.hpp"

namespace lbfgs {
typedef double (*lbfgs_evaluate_t)(void *instance, const Eigen::VectorXd &x, Eigen::VectorXd &g, double &p_cost);
}

int main(int argc, char **argv) {
  if (argc > 1) {  // --compile-only: proves the adapter instantiates and has the reference's callback type
    lbfgs::lbfgs_evaluate_t f = &svsdf::TrajOptimizerHip::costFunctionLbfgs;
    std::printf("%d\n", f != nullptr);
    return 0;
  }
  char name[256];
  double sh, wp, rho;
  int N, P;
  if (std::scanf("%255s %lf %lf %lf %d %d", name, &sh, &wp, &rho, &N, &P) != 6) return 2;
  double hs[9], ts[9];
  for (double &v : hs) if (std::scanf("%lf", &v) != 1) return 2;
  for (double &v : ts) if (std::scanf("%lf", &v) != 1) return 2;
  const int n = 4 * N - 3;
  Eigen::VectorXd x(n), g(n);
  std::vector<double> pts(3 * (size_t)P);
  for (int i = 0; i < n; ++i) if (std::scanf("%lf", &x(i)) != 1) return 2;
  for (double &v : pts) if (std::scanf("%lf", &v) != 1) return 2;
  Eigen::MatrixX3d coeffs(6 * N), gradC(6 * N);
  Eigen::VectorXd T(N), gradT(N);
  for (int i = 0; i < 18 * N; ++i) if (std::scanf("%lf", coeffs.data() + i) != 1) return 2;
  for (int i = 0; i < N; ++i) if (std::scanf("%lf", &T(i)) != 1) return 2;
  svsdf::TrajOptimizerHip opt;
  opt.inputdata = name; opt.safety_hor = sh; opt.weight_p = wp; opt.rho = rho; opt.device = 0;
  opt.setConditions(hs, ts, N);
  opt.setPoints(pts.data(), (size_t)P);
  lbfgs::lbfgs_evaluate_t eval = &svsdf::TrajOptimizerHip::costFunctionLbfgs;
  double p_cost = -1.0;
  const double f = eval(&opt, x, g, p_cost);
  std::printf("%.17g %.17g %.17g\n", f, p_cost, opt.cost_pos);
  for (int i = 0; i < n; ++i) std::printf("%.17g\n", g(i));
  double cost = 0.25;                       // accumulated into (+=), like the reference
  for (int i = 0; i < N; ++i) gradT(i) = 1.0;
  svsdf::TrajOptimizerHip::addSaftyPenaOnSweptVolumeParallelTrueSDF(&opt, T, coeffs, cost, gradT, gradC);
  std::printf("%.17g\n", cost);
  for (int i = 0; i < N; ++i) std::printf("%.17g\n", gradT(i));
  for (int i = 0; i < 18 * N; ++i) std::printf("%.17g\n", gradC.data()[i]);
  return 0;
}
