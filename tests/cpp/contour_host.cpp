// contour_host.cpp -- host harness for csrc/svsdf_contour.hpp (tests/test_contour.py): zero contours of analytic fields.
// usage: contour_host <field> <h> <levels>     field: disc | two | ring | saddle | steep
// prints: nodes_evaluated dense_nodes open_chains nloops, then per loop "size signed_area", a checksum of all vertices,
// then the closed-surface check of the extrusion: bad edges (0) and the enclosed volume
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
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
  // closed surface of the extrusion (walls + caps): every directed edge once, its reverse once; volume = area x height
  std::vector<double> V;
  std::vector<int> F;
  svsdf_host::extrude_outline(xy.data(), loops.data(), loops.size(), -0.5, 0.5, true, V, F);
  std::map<std::pair<int, int>, int> edges;
  double vol = 0.0;
  for (size_t t = 0; t + 2 < F.size(); t += 3) {
    const int id[3] = {F[t], F[t + 1], F[t + 2]};
    for (int k = 0; k < 3; ++k) edges[{id[k], id[(k + 1) % 3]}]++;
    const double *a = &V[3 * (size_t)id[0]], *b = &V[3 * (size_t)id[1]], *c = &V[3 * (size_t)id[2]];
    vol += a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
  }
  int bad = 0;
  for (const auto &e : edges) {
    if (e.second != 1) ++bad;
    if (!edges.count({e.first.second, e.first.first})) ++bad;
  }
  std::printf("%d %.12f\n", bad, vol / 6.0);
  return 0;
}
