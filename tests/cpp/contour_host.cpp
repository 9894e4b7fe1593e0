// contour_host.cpp -- host harness for csrc/svsdf_contour.hpp (tests/test_contour.py): zero contours of analytic fields.
// usage: contour_host <field> <h> <levels>     field: disc | two | ring | saddle | steep
// prints: nodes_evaluated dense_nodes open_chains nloops, then per loop "size signed_area", then a checksum of all vertices
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../implicit-svsdf-planner_amd/csrc/svsdf_contour.hpp"

static double field(const std::string &name, double x, double y) {
  auto disc = [](double px, double py, double cx, double cy, double r) { return std::hypot(px - cx, py - cy) - r; };
  if (name == "disc") return disc(x, y, 0.3, -0.2, 1.7);
  if (name == "two") return std::fmin(disc(x, y, -2.0, 0.0, 1.0), disc(x, y, 2.1, 0.4, 0.8));
  if (name == "ring") return std::fabs(std::hypot(x - 0.1, y + 0.05) - 2.0) - 0.5;   // annulus: outer boundary + a hole
  if (name == "saddle") return std::fmin(disc(x, y, -0.51, -0.51, 0.7), disc(x, y, 0.51, 0.51, 0.7));   // two discs touching diagonally
  // not 1-Lipschitz (slope 4, and a jump across x = 0.3 that moves the zero set): the band misses cells, the
  // continuation step has to complete the chains
  if (name == "steep") return (x < 0.3) ? 4.0 * disc(x, y, 0.0, 0.0, 1.7) : 4.0 * disc(x, y, 0.0, 0.0, 1.22) + 0.9;
  return 1.0;
}

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const std::string name = argv[1];
  svsdf_host::ContourGrid g;
  g.h = std::atof(argv[2]);
  g.levels = std::atoi(argv[3]);
  g.x0 = -4.03; g.y0 = -3.51;
  g.nx = (long long)std::ceil(8.06 / g.h); g.ny = (long long)std::ceil(7.02 / g.h);
  const svsdf_host::FieldEval f = [&](const std::vector<double> &xy, std::vector<double> &val) -> int {
    val.resize(xy.size() / 2);
    for (size_t k = 0; k < val.size(); ++k) val[k] = field(name, xy[2 * k], xy[2 * k + 1]);
    return 0;
  };
  std::vector<double> xy;
  std::vector<int> loops;
  svsdf_host::ContourStats st;
  const int rc = svsdf_host::swept_contour(g, f, 1.0, xy, loops, &st);
  if (rc) { std::printf("error %d\n", rc); return 1; }
  std::printf("%llu %llu %d %zu\n", st.nodes_evaluated, st.dense_nodes, st.open_chains, loops.size());
  size_t off = 0;
  double maxres = 0.0;
  for (int n : loops) {
    std::printf("%d %.12f\n", n, svsdf_host::polyline_area(xy.data() + 2 * off, n));
    off += (size_t)n;
  }
  double sx = 0.0, sy = 0.0;
  for (size_t k = 0; k < xy.size() / 2; ++k) {
    sx += xy[2 * k]; sy += xy[2 * k + 1];
    maxres = std::fmax(maxres, std::fabs(field(name, xy[2 * k], xy[2 * k + 1])));
  }
  std::printf("%.12f %.12f %.6e\n", sx, sy, maxres);
  return 0;
}
