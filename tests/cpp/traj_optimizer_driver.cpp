// Drives the C++ mirror (include/svsdf_traj_optimizer.hpp) exactly like the reference's
// optimize_traj_lmbm drives TrajOptimizer: a raw lmbm_evaluate_t function pointer + void* instance.
// Input (stdin): shape_inputdata safety_hor weight_p rho N P, then 9+9 state doubles, n x-doubles,
// 3P point doubles.  Output: cost, cost_pos, cost_other, cost_total, then g[0..n); with --optimize also
// ret, iterations, final cost, cost_total and the optimised x[0..n).  `--devices 0,1,...` makes the ONE TrajOptimizerHip
// of this process drive several GPUs (svsdf_config::n_devices; a device may repeat: several stripes on one GPU).
// `--reconfig`: after the first evaluation the public member safety_hor is edited (x 1.5) like the reference's members
// can be, and the callback is evaluated again (the mirror rebuilds its context): cost and cost_pos of the second call
// follow.  An inputdata whose stem the shape registry does not know is read as an .obj mesh (z = 0 outline -> Polygon).
// `--swept`: calculateSwept for the trajectory x (cell 0.1): rc, mesh vertices, triangles, loops, outline vertices and the
// sums of their coordinates follow.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "svsdf_traj_optimizer.hpp"

typedef double (*lmbm_evaluate_t)(void *instance, const double *x, double *g, const int n);  // lmbm.h:206-209

int main(int argc, char **argv) {
  if (argc > 1 && std::string(argv[1]) == "--host-only") {
    // no GPU needed: registry lookup + MINCO helper through the same library
    std::printf("%d %d\n", svsdf_shape_id_from_inputdata("shapes/sdHeart.obj"), svsdf_shape_id_from_inputdata("x/unknown.obj"));
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
  std::vector<double> x(n), g(n), pts(3 * (size_t)P);
  for (double &v : x) if (std::scanf("%lf", &v) != 1) return 2;
  for (double &v : pts) if (std::scanf("%lf", &v) != 1) return 2;
  std::vector<int> devices;
  for (int a = 1; a + 1 < argc; ++a)
    if (std::string(argv[a]) == "--devices") {
      const std::string list = argv[a + 1];
      size_t pos = 0;
      while (pos < list.size()) {
        size_t e = list.find(',', pos);
        if (e == std::string::npos) e = list.size();
        devices.push_back(std::atoi(list.substr(pos, e - pos).c_str()));
        pos = e + 1;
      }
    }
  svsdf::TrajOptimizerHip opt;
  opt.inputdata = name; opt.safety_hor = sh; opt.weight_p = wp; opt.rho = rho; opt.device = 0;
  opt.devices = devices;
  opt.setConditions(hs, ts, N);
  opt.setPoints(pts.data(), (size_t)P);
  lmbm_evaluate_t eval = &svsdf::TrajOptimizerHip::costFunctionLmbmParallel;
  const double f = eval(&opt, x.data(), g.data(), n);
  std::printf("%.17g %.17g %.17g %.17g\n", f, opt.cost_pos, opt.cost_other, opt.cost_total);
  for (int i = 0; i < n; ++i) std::printf("%.17g\n", g[i]);
  if (argc > 1 && std::string(argv[1]) == "--reconfig") {
    opt.safety_hor = 1.5 * sh;
    const double f2 = eval(&opt, x.data(), g.data(), n);
    std::printf("%.17g %.17g\n", f2, opt.cost_pos);
  }
  if (argc > 1 && std::string(argv[1]) == "--swept") {
    // SweptVolumeManager::calculateSwept(U_, G_) call shape (SWM:321-336): mesh of the swept volume for the trajectory x
    std::vector<double> T(N), coeffs(18 * (size_t)N), U, xy;
    std::vector<int> G, sizes;
    svsdf_forward_T(x.data(), T.data(), N);
    if (svsdf_minco_coeffs(hs, ts, N, x.data() + N, T.data(), coeffs.data()) != SVSDF_OK) return 3;
    const int rc = opt.calculateSwept(T.data(), coeffs.data(), N, U, G, 0.1, -0.5, 0.5, &xy, &sizes);
    double sx = 0.0, sy = 0.0;
    for (size_t k = 0; k < xy.size() / 2; ++k) { sx += xy[2 * k]; sy += xy[2 * k + 1]; }
    std::printf("%d %zu %zu %zu %zu %.17g %.17g\n", rc, U.size() / 3, G.size() / 3, sizes.size(), xy.size() / 2, sx, sy);
  }
  if (argc > 1 && std::string(argv[1]) == "--optimize") {
    // optimize_traj_lmbm(initS, finalS, opt_x, N, traj) call shape of plan_manager.cpp:176
    svsdf_lbfgs_params prm;
    svsdf_lbfgs_params_default(&prm);
    prm.max_iterations = 15;
    double final_cost = 0.0;
    svsdf::TrajOptimizerHip opt2;
    opt2.inputdata = name; opt2.safety_hor = sh; opt2.weight_p = wp; opt2.rho = rho; opt2.device = 0;
    opt2.setPoints(pts.data(), (size_t)P);
    const int ret = opt2.optimize_traj_lmbm(hs, ts, x.data(), N, &final_cost, &prm);
    std::printf("%d %d %.17g %.17g\n", ret, opt2.iter, final_cost, opt2.cost_total);
    for (int i = 0; i < n; ++i) std::printf("%.17g\n", x[i]);
  }
  return 0;
}
