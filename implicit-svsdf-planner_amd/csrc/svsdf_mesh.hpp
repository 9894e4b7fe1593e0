// svsdf_mesh.hpp -- host: triangle mesh (.obj) -> outline of its z = z0 cross-section.
//
// BASELINE config 5: "arbitrary .obj mesh ... (no analytic shape SDF)".  The reference loads conf.inputdata with
// igl::read_triangle_mesh (Shape.hpp:281-313) and, for a stem its shape registry does not know, falls back to the generic
// `Polygon` shape over an outline (sw_manager.hpp:350-372).  Its planner is planar: every query has z = 0 (BEO:790-791),
// so what the optimizer can see of a mesh is its cross-section with the plane z = 0 -- for the extruded slabs under
// src/plan_manager/shapes/ (|z| <= 0.5) the outline of the robot.  This file produces that outline as the vertex loop
// svsdf_config::polygon_xy takes: every triangle that straddles the plane contributes the segment between its two
// crossed edges; crossing points are keyed by the (undirected) mesh edge they lie on, so neighbouring triangles share
// them exactly and the segments chain into closed loops without any tolerance.  A vertex exactly on the plane counts
// as above it.  mesh_section returns every closed loop, largest |area| first (round 5: a section with a hole or of several
// solids is planned with all of them -- Polygon::getonlySDF is a minimum / a crossing count over edges, it does not care
// how they are chained, svsdf_polygon.hpp); mesh_outline the first of them (a closed, orientable mesh of one solid gives
// exactly one loop).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace svsdf_host {

// Wavefront .obj: `v x y z` and `f a b c ...` (1-based, negative = relative, `a/b/c` forms; polygons are fanned).
inline bool read_obj(const char *path, std::vector<double> &V, std::vector<int> &F) {
  V.clear(); F.clear();
  FILE *f = std::fopen(path, "r");
  if (!f) return false;
  char line[4096];
  while (std::fgets(line, sizeof line, f)) {
    const char *p = line;
    while (*p == ' ' || *p == '\t') ++p;
    if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
      double x, y, z;
      if (std::sscanf(p + 1, "%lf %lf %lf", &x, &y, &z) == 3) { V.push_back(x); V.push_back(y); V.push_back(z); }
    } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
      std::vector<int> idx;
      const char *q = p + 1;
      for (;;) {
        while (*q == ' ' || *q == '\t') ++q;
        if (*q == '\0' || *q == '\n' || *q == '\r' || *q == '#') break;
        char *end = nullptr;
        const long v = std::strtol(q, &end, 10);
        if (end == q) break;
        const long nv = (long)(V.size() / 3);
        idx.push_back((int)(v > 0 ? v - 1 : nv + v));
        q = end;
        while (*q != '\0' && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') ++q;   // skip /vt/vn
      }
      for (size_t k = 1; k + 1 < idx.size(); ++k) { F.push_back(idx[0]); F.push_back(idx[k]); F.push_back(idx[k + 1]); }
    }
  }
  std::fclose(f);
  return !V.empty() && !F.empty();
}

// Cross-section z = z0 of the mesh (V: nv x 3, F: nf x 3 vertex indices): all its closed loops, ordered by decreasing
// |enclosed area|; xy_out: interleaved vertices, loop after loop, each in chaining order; sizes_out: vertices per loop.
// Returns false on a malformed mesh (index out of range) or when no triangle straddles the plane.
inline bool mesh_section(const double *V, size_t nv, const int *F, size_t nf, double z0, std::vector<double> &xy_out,
                         std::vector<int> &sizes_out) {
  xy_out.clear();
  sizes_out.clear();
  typedef std::pair<int, int> Key;
  std::map<Key, int> id;             // mesh edge -> crossing point
  std::vector<double> px, py;
  std::vector<int> nbr;              // two neighbours per crossing point (-1 = none yet)
  auto above = [&](int v) { return V[3 * (size_t)v + 2] >= z0; };
  auto crossing = [&](int a, int b) -> int {
    const Key k(a < b ? a : b, a < b ? b : a);
    auto it = id.find(k);
    if (it != id.end()) return it->second;
    const double *lo = V + 3 * (size_t)k.first, *hi = V + 3 * (size_t)k.second;
    const double t = (z0 - lo[2]) / (hi[2] - lo[2]);
    px.push_back(lo[0] + t * (hi[0] - lo[0]));
    py.push_back(lo[1] + t * (hi[1] - lo[1]));
    nbr.push_back(-1); nbr.push_back(-1);
    const int n = (int)px.size() - 1;
    id[k] = n;
    return n;
  };
  for (size_t f = 0; f < nf; ++f) {
    const int v[3] = {F[3 * f], F[3 * f + 1], F[3 * f + 2]};
    for (int k = 0; k < 3; ++k)
      if (v[k] < 0 || (size_t)v[k] >= nv) return false;
    const bool u[3] = {above(v[0]), above(v[1]), above(v[2])};
    if (u[0] == u[1] && u[1] == u[2]) continue;
    int c[2], nc = 0;
    for (int k = 0; k < 3; ++k)
      if (u[k] != u[(k + 1) % 3]) c[nc++] = crossing(v[k], v[(k + 1) % 3]);
    // exactly two edges of a straddling triangle are crossed; link the two crossing points
    for (int s = 0; s < 2; ++s) {
      int *slot = &nbr[2 * (size_t)c[s]];
      if (slot[0] < 0) slot[0] = c[1 - s];
      else if (slot[1] < 0) slot[1] = c[1 - s];
      // a third segment at one point: non-manifold edge; the extra link is dropped
    }
  }
  const int np = (int)px.size();
  if (np < 3) return false;
  std::vector<char> seen(np, 0);
  std::vector<std::vector<int>> loops;
  std::vector<double> areas;
  for (int s = 0; s < np; ++s) {
    if (seen[s]) continue;
    std::vector<int> loop;
    int prev = -1, cur = s;
    bool closed = false;
    while (cur >= 0 && !seen[cur]) {
      seen[cur] = 1;
      loop.push_back(cur);
      const int a = nbr[2 * (size_t)cur], b = nbr[2 * (size_t)cur + 1];
      const int nxt = (a != prev) ? a : b;
      prev = cur;
      cur = nxt;
      if (cur == s) { closed = true; break; }
    }
    if (closed && loop.size() >= 3) {
      double a2 = 0.0;   // twice the signed area (shoelace)
      for (size_t q = 0; q < loop.size(); ++q) {
        const int u = loop[q], w = loop[(q + 1) % loop.size()];
        a2 += px[u] * py[w] - px[w] * py[u];
      }
      loops.push_back(std::move(loop));
      areas.push_back(std::fabs(a2));
    }
  }
  if (loops.empty()) return false;
  // (a finely tessellated small part must not come before a coarse large one: by area, not by vertex count; ties keep the
  // order in which the loops were found, which is the order of the mesh's faces)
  std::vector<size_t> order(loops.size());
  for (size_t k = 0; k < order.size(); ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return areas[a] > areas[b]; });
  for (size_t k : order) {
    for (int i : loops[k]) { xy_out.push_back(px[i]); xy_out.push_back(py[i]); }
    sizes_out.push_back((int)loops[k].size());
  }
  return true;
}

// The loop enclosing the largest area (xy_out) and the number of closed loops of the section (loops_out, may be null).
inline bool mesh_outline(const double *V, size_t nv, const int *F, size_t nf, double z0, std::vector<double> &xy_out,
                         int *loops_out) {
  std::vector<int> sizes;
  if (loops_out) *loops_out = 0;
  if (!mesh_section(V, nv, F, nf, z0, xy_out, sizes)) { xy_out.clear(); return false; }
  if (loops_out) *loops_out = (int)sizes.size();
  xy_out.resize(2 * (size_t)sizes[0]);
  return true;
}

}  // namespace svsdf_host
