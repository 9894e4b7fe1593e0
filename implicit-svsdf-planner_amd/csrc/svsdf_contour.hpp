// svsdf_contour.hpp -- host: boundary of the swept volume = zero set of the swept-volume SDF, as closed polylines.
//
// SURVEY §8 f4, second half.  The reference's swept-volume surface extraction (sw_calculate.cpp:4-305, driven by
// SweptVolumeManager::calculateSwept SWM:321-336; dead code in the release, kept for visual validation) grows a sparse
// voxel set from seeds along the trajectory -- one serial gradient descent over t per voxel corner, time seeds handed
// from neighbour to neighbour -- and runs igl::marching_cubes over it.  The planner is planar (every query has z = 0,
// BEO:790-791) and its robots are extruded slabs, so the swept volume is the extrusion of its z = 0 section and the
// surface is that section's outline.  On a GPU that evaluates a million swept SDF values in a few milliseconds a serial
// continuation is the wrong shape; this file takes the field as a batch evaluator and does
//   * a hierarchical narrow band: all nodes of a coarse grid, then only the children of cells that can hold a piece of
//     the zero set -- a field with Lipschitz constant 1 (a minimum over t of signed distance functions is one) that
//     vanishes somewhere in a cell of diagonal d has |f| <= d at all four corners (`slack` x d is used; cells whose
//     corners change sign are kept whatever their values) -- down to the requested cell size: O(boundary length / h)
//     evaluations instead of O(area / h^2).  The computed field is the reference's LOCAL argmin next to its seed and
//     jumps where the seed changes basin, so the band can miss a cell: the marching step below completes it;
//   * marching squares over the finest band cells.  Crossing points are keyed by the grid edge they lie on and computed
//     from that edge's two node values only, so neighbouring cells share them exactly and the segments chain into
//     closed loops without a tolerance (same device as svsdf_mesh.hpp).  Segments run with the inside (f < 0) on
//     their left: outer boundaries come out counter-clockwise, holes clockwise.  Saddle cells (two opposite corners
//     inside) are resolved by the sign of the mean of the four corners.
// The field evaluator the library passes is the hot path's own argmin solve (getSDFofSweptVolume<false,true>, SWM:844-866;
// the reference's calculateSwept marches the same function through its scalarFunc, SWM:1426-1446): what is drawn is
// the zero set of exactly the quantity whose sign decides interior / exterior for the optimizer's penalty (SWM:921).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <unordered_map>
#include <utility>
#include <vector>

namespace svsdf_host {

struct ContourGrid {
  double x0 = 0.0, y0 = 0.0;   // node (0, 0)
  double h = 0.05;             // finest cell size
  long long nx = 0, ny = 0;    // finest cells per direction (rounded up to a multiple of 2^levels by swept_contour)
  int levels = 4;              // coarsest cells are 2^levels finest cells wide
};
struct ContourStats {
  unsigned long long nodes_evaluated = 0, cells_marched = 0, batches = 0;
  unsigned long long dense_nodes = 0;   // what a dense grid of the same cell size would have evaluated
  int open_chains = 0;                  // > 0: the zero set left the grid (or the band lost a cell): result incomplete
};
// values of the field at the points xy (interleaved); returns 0 on success
using FieldEval = std::function<int(const std::vector<double> &xy, std::vector<double> &val)>;

namespace contour_detail {
inline uint64_t node_key(long long i, long long j) { return ((uint64_t)(uint32_t)i << 32) | (uint64_t)(uint32_t)j; }
struct Cell { long long i, j; };   // lower-left node, finest units
}  // namespace contour_detail

// Zero contour of f over the grid.  xy_out: interleaved vertices of all loops one after the other; loop_sizes: vertices
// per loop (closed: the last vertex connects to the first).  Returns 0, or the evaluator's error code.
inline int swept_contour(ContourGrid g, const FieldEval &f, double slack, std::vector<double> &xy_out,
                         std::vector<int> &loop_sizes, ContourStats *stats) {
  using namespace contour_detail;
  xy_out.clear();
  loop_sizes.clear();
  ContourStats st;
  if (!(g.h > 0.0) || g.nx < 1 || g.ny < 1 || g.levels < 0 || g.levels > 12) return -1;
  const long long step0 = 1ll << g.levels;
  g.nx = (g.nx + step0 - 1) / step0 * step0;
  g.ny = (g.ny + step0 - 1) / step0 * step0;
  if (g.nx >= (1ll << 31) || g.ny >= (1ll << 31)) return -1;
  st.dense_nodes = (unsigned long long)(g.nx + 1) * (unsigned long long)(g.ny + 1);
  std::unordered_map<uint64_t, double> val;
  auto ensure = [&](const std::vector<Cell> &cells, long long step) -> int {
    std::vector<double> pts, out;
    std::vector<uint64_t> keys;
    for (const Cell &c : cells)
      for (int k = 0; k < 4; ++k) {
        const long long i = c.i + ((k == 1 || k == 2) ? step : 0), j = c.j + ((k >= 2) ? step : 0);
        const uint64_t key = node_key(i, j);
        if (val.emplace(key, std::nan("")).second) {
          keys.push_back(key);
          pts.push_back(g.x0 + (double)i * g.h);
          pts.push_back(g.y0 + (double)j * g.h);
        }
      }
    if (keys.empty()) return 0;
    const int rc = f(pts, out);
    if (rc) return rc;
    if (out.size() != keys.size()) return -2;
    for (size_t k = 0; k < keys.size(); ++k) val[keys[k]] = out[k];
    st.nodes_evaluated += keys.size();
    st.batches++;
    return 0;
  };
  std::vector<Cell> cells;
  for (long long j = 0; j < g.ny; j += step0)
    for (long long i = 0; i < g.nx; i += step0) cells.push_back(Cell{i, j});
  for (int lev = g.levels; lev >= 0; --lev) {
    const long long step = 1ll << lev;
    const int rc = ensure(cells, step);
    if (rc) return rc;
    if (lev == 0) break;
    const double band = slack * std::sqrt(2.0) * (double)step * g.h;
    std::vector<Cell> next;
    for (const Cell &c : cells) {
      const double v[4] = {val[node_key(c.i, c.j)], val[node_key(c.i + step, c.j)], val[node_key(c.i + step, c.j + step)],
                           val[node_key(c.i, c.j + step)]};
      bool all_in_band = true, neg = false, pos = false;
      for (double x : v) {
        all_in_band = all_in_band && std::fabs(x) <= band;
        neg = neg || x < 0.0;
        pos = pos || !(x < 0.0);
      }
      if (all_in_band || (neg && pos)) {
        const long long hstep = step / 2;
        next.push_back(Cell{c.i, c.j});
        next.push_back(Cell{c.i + hstep, c.j});
        next.push_back(Cell{c.i, c.j + hstep});
        next.push_back(Cell{c.i + hstep, c.j + hstep});
      }
    }
    cells.swap(next);
  }
  // ---- marching squares over the finest band cells
  // vertex id: grid edge = (lower / left node, direction 0 = +x, 1 = +y)
  std::unordered_map<uint64_t, int> vid[2];
  std::vector<double> vx, vy;
  std::vector<int> nxt, has_in;
  auto vertex = [&](long long i, long long j, int dir) -> int {
    const uint64_t key = node_key(i, j);
    auto it = vid[dir].find(key);
    if (it != vid[dir].end()) return it->second;
    const double va = val[key], vb = val[node_key(i + (dir == 0), j + (dir == 1))];
    const double t = va / (va - vb);   // one node < 0, the other >= 0: the denominator is not 0
    const int id = (int)vx.size();
    vx.push_back(g.x0 + ((double)i + (dir == 0 ? t : 0.0)) * g.h);
    vy.push_back(g.y0 + ((double)j + (dir == 1 ? t : 0.0)) * g.h);
    nxt.push_back(-1);
    has_in.push_back(0);
    vid[dir].emplace(key, id);
    return id;
  };
  // segments per case as (from edge, to edge), inside on the left; edges: 0 bottom, 1 right, 2 top, 3 left
  static const signed char seg[16][4] = {
      {-1, -1, -1, -1}, {0, 3, -1, -1}, {1, 0, -1, -1}, {1, 3, -1, -1}, {2, 1, -1, -1}, {0, 3, 2, 1} /* saddle */,
      {2, 0, -1, -1},   {2, 3, -1, -1}, {3, 2, -1, -1}, {0, 2, -1, -1}, {1, 0, 3, 2} /* saddle */, {1, 2, -1, -1},
      {3, 1, -1, -1},   {0, 1, -1, -1}, {3, 0, -1, -1}, {-1, -1, -1, -1}};
  // The band can miss a finest cell next to a crossing where the field jumps (see above): every crossing vertex lies on a
  // grid edge shared by two cells and both must be marched for the chain to continue, so the cells on the other side of
  // freshly found crossings are added (nodes evaluated, marched) until none is missing -- the reference's continuation
  // idea (sw_calculate.cpp: grow the voxel set along the surface), applied only where the band was not enough.
  std::unordered_map<uint64_t, char> marched;
  std::vector<Cell> todo = cells;
  while (!todo.empty()) {
    {
      std::vector<Cell> fresh;
      for (const Cell &c : todo)
        if (c.i >= 0 && c.j >= 0 && c.i < g.nx && c.j < g.ny && marched.emplace(node_key(c.i, c.j), 1).second) fresh.push_back(c);
      todo.swap(fresh);
    }
    if (todo.empty()) break;
    const int rcm = ensure(todo, 1);
    if (rcm) return rcm;
    std::vector<Cell> more;
    for (const Cell &c : todo) {
      const double v0 = val[node_key(c.i, c.j)], v1 = val[node_key(c.i + 1, c.j)], v2 = val[node_key(c.i + 1, c.j + 1)],
                   v3 = val[node_key(c.i, c.j + 1)];
      const int code = (v0 < 0.0 ? 1 : 0) | (v1 < 0.0 ? 2 : 0) | (v2 < 0.0 ? 4 : 0) | (v3 < 0.0 ? 8 : 0);
      st.cells_marched++;
      if (code == 0 || code == 15) continue;
      signed char s4[4] = {seg[code][0], seg[code][1], seg[code][2], seg[code][3]};
      if (code == 5 || code == 10) {
        const bool centre_in = (((v0 + v1) + v2) + v3) * 0.25 < 0.0;
        if (centre_in) {   // the two inside corners are connected: cut off the two outside corners instead
          if (code == 5) { s4[0] = 0; s4[1] = 1; s4[2] = 2; s4[3] = 3; }
          else { s4[0] = 3; s4[1] = 0; s4[2] = 1; s4[3] = 2; }
        }
      }
      auto edge_vertex = [&](int e) -> int {
        switch (e) {
          case 0: more.push_back(Cell{c.i, c.j - 1}); return vertex(c.i, c.j, 0);
          case 1: more.push_back(Cell{c.i + 1, c.j}); return vertex(c.i + 1, c.j, 1);
          case 2: more.push_back(Cell{c.i, c.j + 1}); return vertex(c.i, c.j + 1, 0);
          default: more.push_back(Cell{c.i - 1, c.j}); return vertex(c.i, c.j, 1);
        }
      };
      for (int k = 0; k < 4 && s4[k] >= 0; k += 2) {
        const int a = edge_vertex(s4[k]), b = edge_vertex(s4[k + 1]);
        nxt[a] = b;
        has_in[b] = 1;
      }
    }
    todo.swap(more);
  }
  // ---- chain: open chains first (from vertices nobody points at), then the closed loops
  const int nv = (int)vx.size();
  std::vector<char> seen(nv, 0);
  for (int s = 0; s < nv; ++s)
    if (!has_in[s]) st.open_chains++;
  for (int s = 0; s < nv; ++s) {
    if (seen[s] || !has_in[s]) continue;
    // walk; a chain that runs into a vertex without successor is open and dropped
    std::vector<int> loop;
    int v = s;
    bool closed = false;
    while (v >= 0 && !seen[v]) {
      seen[v] = 1;
      loop.push_back(v);
      v = nxt[v];
      if (v == s) { closed = true; break; }
    }
    if (!closed || loop.size() < 3) continue;
    for (int id : loop) { xy_out.push_back(vx[id]); xy_out.push_back(vy[id]); }
    loop_sizes.push_back((int)loop.size());
  }
  if (stats) *stats = st;
  return 0;
}

// signed area of a closed polyline (> 0: counter-clockwise)
inline double polyline_area(const double *xy, int n) {
  double a = 0.0;
  for (int k = 0; k < n; ++k) {
    const int m = (k + 1 == n) ? 0 : k + 1;
    a += xy[2 * k] * xy[2 * m + 1] - xy[2 * m] * xy[2 * k + 1];
  }
  return 0.5 * a;
}

namespace contour_detail {
inline double cross2(const double *o, const double *a, const double *b) {
  return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0]);
}
inline bool point_in_loop(const double *xy, int n, double px, double py) {   // even-odd
  bool in = false;
  for (int k = 0, j = n - 1; k < n; j = k++) {
    const double xi = xy[2 * k], yi = xy[2 * k + 1], xj = xy[2 * j], yj = xy[2 * j + 1];
    if (((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi) + xi)) in = !in;
  }
  return in;
}
inline bool in_triangle(const double *a, const double *b, const double *c, const double *p) {   // closed triangle, ccw
  return cross2(a, b, p) >= 0.0 && cross2(b, c, p) >= 0.0 && cross2(c, a, p) >= 0.0;
}

// Triangulation of one counter-clockwise loop with its clockwise holes (vertex ids into P, 2 doubles per vertex): the
// holes are joined to the outer loop by bridge edges (hole by hole from the right: the hole's right-most vertex sees
// the outer vertex chosen as in Eberly, "Triangulation by ear clipping"), then ears are clipped.  Output triangles are
// counter-clockwise.  Every triangle uses input vertices only, so caps and walls share their edges.
inline void triangulate(const double *P, std::vector<int> outer, std::vector<std::vector<int>> holes, std::vector<int> &tri) {
  auto pt = [&](int id) { return P + 2 * (size_t)id; };
  std::vector<std::pair<double, size_t>> order;
  for (size_t h = 0; h < holes.size(); ++h) {
    double mx = -1e300;
    for (int id : holes[h]) mx = std::max(mx, pt(id)[0]);
    order.push_back({-mx, h});
  }
  std::sort(order.begin(), order.end());
  for (const auto &oh : order) {
    std::vector<int> &H = holes[oh.second];
    size_t mi = 0;
    for (size_t k = 1; k < H.size(); ++k)
      if (pt(H[k])[0] > pt(H[mi])[0]) mi = k;
    const double *M = pt(H[mi]);
    // nearest crossing of the ray M + (t, 0), t > 0, with the outer polygon
    double best_x = 1e300;
    long long be = -1;
    const size_t n = outer.size();
    for (size_t k = 0; k < n; ++k) {
      const double *a = pt(outer[k]), *b = pt(outer[(k + 1) % n]);
      if ((a[1] > M[1]) == (b[1] > M[1])) continue;
      const double x = a[0] + (M[1] - a[1]) * (b[0] - a[0]) / (b[1] - a[1]);
      if (x >= M[0] && x < best_x) { best_x = x; be = (long long)k; }
    }
    if (be < 0) continue;   // not inside this loop (numerically): leave the hole out
    size_t pi = (pt(outer[be])[0] > pt(outer[(be + 1) % n])[0]) ? (size_t)be : (size_t)((be + 1) % n);
    const double I[2] = {best_x, M[1]};
    // a reflex outer vertex inside triangle (M, I, P) hides P: take the one closest in angle to the ray
    {
      const double *Pp = pt(outer[pi]);
      const double *ta = M, *tb = (Pp[1] < M[1]) ? Pp : I, *tc = (Pp[1] < M[1]) ? I : Pp;   // counter-clockwise
      double best_t = 1e300, best_d = 1e300;
      size_t cand = pi;
      for (size_t k = 0; k < n; ++k) {
        const double *v = pt(outer[k]);
        if (k == pi || !(v[0] > M[0])) continue;
        if (!in_triangle(ta, tb, tc, v)) continue;
        const double *pv = pt(outer[(k + n - 1) % n]), *nv = pt(outer[(k + 1) % n]);
        if (cross2(pv, v, nv) > 0.0) continue;   // convex vertices cannot hide anything
        const double dx = v[0] - M[0], dy = std::fabs(v[1] - M[1]);
        const double t = dy / dx, d = dx * dx + dy * dy;
        if (t < best_t || (t == best_t && d < best_d)) { best_t = t; best_d = d; cand = k; }
      }
      pi = cand;
    }
    std::vector<int> merged;
    merged.reserve(outer.size() + H.size() + 2);
    for (size_t k = 0; k <= pi; ++k) merged.push_back(outer[k]);
    for (size_t k = 0; k <= H.size(); ++k) merged.push_back(H[(mi + k) % H.size()]);
    for (size_t k = pi; k < outer.size(); ++k) merged.push_back(outer[k]);
    outer.swap(merged);
  }
  // ---- ear clipping
  const int n = (int)outer.size();
  if (n < 3) return;
  std::vector<int> prv(n), nxt(n);
  for (int k = 0; k < n; ++k) { prv[k] = (k + n - 1) % n; nxt[k] = (k + 1) % n; }
  auto is_ear = [&](int k) -> bool {
    const double *a = pt(outer[prv[k]]), *b = pt(outer[k]), *c = pt(outer[nxt[k]]);
    if (!(cross2(a, b, c) > 0.0)) return false;
    for (int q = nxt[nxt[k]]; q != prv[k]; q = nxt[q]) {
      const double *v = pt(outer[q]);
      if ((v[0] == a[0] && v[1] == a[1]) || (v[0] == b[0] && v[1] == b[1]) || (v[0] == c[0] && v[1] == c[1])) continue;
      const double *pv = pt(outer[prv[q]]), *nv = pt(outer[nxt[q]]);
      if (cross2(pv, v, nv) > 0.0) continue;                 // only reflex (or flat) vertices can lie inside an ear
      if (in_triangle(a, b, c, v)) return false;
    }
    return true;
  };
  int left = n, cur = 0, since = 0;
  while (left > 3) {
    bool clip = is_ear(cur);
    if (!clip && since > left) {
      // no ear in a whole round (collinear runs, coincident bridge vertices): clip the flattest non-reflex corner
      int bestk = cur;
      double bestc = 1e300;
      for (int q = cur, c = 0; c < left; q = nxt[q], ++c) {
        const double cr = cross2(pt(outer[prv[q]]), pt(outer[q]), pt(outer[nxt[q]]));
        if (cr >= 0.0 && cr < bestc) { bestc = cr; bestk = q; }
      }
      cur = bestk;
      clip = true;
    }
    if (clip) {
      tri.push_back(outer[prv[cur]]); tri.push_back(outer[cur]); tri.push_back(outer[nxt[cur]]);
      nxt[prv[cur]] = nxt[cur];
      prv[nxt[cur]] = prv[cur];
      cur = nxt[cur];
      --left;
      since = 0;
    } else {
      cur = nxt[cur];
      ++since;
    }
  }
  tri.push_back(outer[prv[cur]]); tri.push_back(outer[cur]); tri.push_back(outer[nxt[cur]]);
}
}  // namespace contour_detail

// Surface of the extrusion of closed polylines over z in [z0, z1] (what the reference's marching-cubes mesh of an extruded
// slab's sweep shows, SWM:321-336 / vis->visMesh): per loop of n vertices 2n mesh vertices (bottom ring, then top ring),
// 2n wall triangles with their normals away from the inside (the loops have the inside on their left) and, with `caps`,
// the bottom and top faces: every counter-clockwise loop triangulated together with the clockwise loops (holes) it
// contains, on the rings' own vertices -- a closed, consistently oriented surface.  V: 3 doubles per vertex, F: 3
// zero-based indices per triangle.
inline void extrude_outline(const double *xy, const int *loop_sizes, size_t n_loops, double z0, double z1, bool caps,
                            std::vector<double> &V, std::vector<int> &F) {
  V.clear();
  F.clear();
  std::vector<size_t> loop_off(n_loops + 1, 0);
  std::vector<int> base(n_loops, 0);
  for (size_t l = 0; l < n_loops; ++l) loop_off[l + 1] = loop_off[l] + (size_t)loop_sizes[l];
  for (size_t l = 0; l < n_loops; ++l) {
    const int n = loop_sizes[l];
    const size_t off = loop_off[l];
    base[l] = (int)(V.size() / 3);
    for (int ring = 0; ring < 2; ++ring)
      for (int k = 0; k < n; ++k) {
        V.push_back(xy[2 * (off + k)]); V.push_back(xy[2 * (off + k) + 1]); V.push_back(ring ? z1 : z0);
      }
    for (int k = 0; k < n; ++k) {
      const int a0 = base[l] + k, b0 = base[l] + (k + 1) % n, a1 = a0 + n, b1 = b0 + n;
      F.push_back(a0); F.push_back(b0); F.push_back(b1);
      F.push_back(a0); F.push_back(b1); F.push_back(a1);
    }
  }
  if (!caps) return;
  // holes go to the smallest counter-clockwise loop that contains them
  std::vector<double> area(n_loops);
  for (size_t l = 0; l < n_loops; ++l) area[l] = polyline_area(xy + 2 * loop_off[l], loop_sizes[l]);
  std::vector<std::vector<size_t>> holes_of(n_loops);
  for (size_t h = 0; h < n_loops; ++h) {
    if (area[h] >= 0.0) continue;
    long long best = -1;
    for (size_t o = 0; o < n_loops; ++o) {
      if (area[o] <= 0.0 || area[o] < -area[h]) continue;
      if (!contour_detail::point_in_loop(xy + 2 * loop_off[o], loop_sizes[o], xy[2 * loop_off[h]], xy[2 * loop_off[h] + 1])) continue;
      if (best < 0 || area[o] < area[(size_t)best]) best = (long long)o;
    }
    if (best >= 0) holes_of[(size_t)best].push_back(h);
  }
  for (size_t o = 0; o < n_loops; ++o) {
    if (area[o] <= 0.0) continue;
    std::vector<int> outer(loop_sizes[o]);
    for (int k = 0; k < loop_sizes[o]; ++k) outer[k] = (int)(loop_off[o] + (size_t)k);
    std::vector<std::vector<int>> holes;
    for (size_t h : holes_of[o]) {
      std::vector<int> H(loop_sizes[h]);
      for (int k = 0; k < loop_sizes[h]; ++k) H[k] = (int)(loop_off[h] + (size_t)k);
      holes.push_back(H);
    }
    std::vector<int> tri;   // ids into xy (outline vertex numbering)
    contour_detail::triangulate(xy, outer, holes, tri);
    auto mesh_id = [&](int id, bool top) -> int {
      size_t l = 0;
      while (l + 1 < n_loops && (size_t)id >= loop_off[l + 1]) ++l;
      return base[l] + (int)((size_t)id - loop_off[l]) + (top ? loop_sizes[l] : 0);
    };
    for (size_t t = 0; t + 2 < tri.size(); t += 3) {
      // top face: counter-clockwise seen from above (normal +z); bottom face: reversed (normal -z)
      F.push_back(mesh_id(tri[t], true)); F.push_back(mesh_id(tri[t + 1], true)); F.push_back(mesh_id(tri[t + 2], true));
      F.push_back(mesh_id(tri[t], false)); F.push_back(mesh_id(tri[t + 2], false)); F.push_back(mesh_id(tri[t + 1], false));
    }
  }
}

}  // namespace svsdf_host
