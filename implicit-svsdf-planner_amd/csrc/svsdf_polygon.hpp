// svsdf_polygon.hpp -- the generic robot shape: a closed 2-D outline (reference: class Polygon, Shape.hpp:1352-1531).
//
// BASELINE config 5 ("arbitrary .obj mesh, no analytic shape SDF") reaches this path through the z = 0 outline of the
// mesh (csrc/svsdf_mesh.hpp): 77 ... 754 vertices for the reference's shapes/*.obj.  Polygon::getonlySDF
// (SHP:1448-1476) is a loop over ALL edges per evaluation -- distance to the segment (dis2Seg, SHP:1385-1401) and an
// atan2-based crossing test against the +x ray (isCrossRayOnXDir, SHP:1370-1383).  This file evaluates the same value,
// bit for bit, from candidate lists built once per outline on the host:
//
//   distance  two uniform grids in the body frame (a fine one around the outline, a coarse one out to ~3 shape
//             sizes).  A cell lists, in ascending edge order, every edge that can be the nearest one for SOME point of
//             the (slightly enlarged) cell:  lb_e <= U,  lb_e = exact distance cell <-> edge,  U = min over edges of
//             the largest corner-to-edge distance (distance to a segment is convex, so that is its maximum over the
//             cell).  An edge that is not listed is farther than the nearest listed one by a margin (1e-9) far above
//             the rounding of the per-edge arithmetic, so the minimum over the listed edges is the minimum over all
//             edges: the same double the reference's loop ends with.  Points outside both grids take the full loop.
//   parity    horizontal slabs over the outline's y-range; a slab lists the edges whose y-range (enlarged by `tol`)
//             meets it.  An edge with both end points on one side of the query's ray line by more than the rounding
//             of two atan2 calls can never satisfy |theta_s - theta_e| >= PI (both angles strictly inside (0, PI) or
//             (PI, 2 PI)), so only the listed edges are put through the reference's test; a query above / below the
//             outline or to the right of it (x > xmax + tol: every angle in (PI/2, 3 PI/2)) counts no crossing.
//
// The per-edge arithmetic is the reference's, operation for operation (true division, no contraction; v = end - start
// and v.squaredNorm() are the same IEEE operations whether done here per evaluation or once on the host).
// __host__ __device__: tests/cpp/poly_host.cpp runs the very same functions on the CPU against the oracle's plain loop.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <vector>

namespace svsdf {

struct PolyEdge { double sx, sy, vx, vy, vv; };   // start, v = end - start, v.squaredNorm(); the end is the next edge's start

// A candidate list in one 32-byte record of 16-bit words: [0] = count; count <= 15: [1 .. count] = the edges, ascending;
// count > 15: [1] | [2] << 16 = offset of the list in `over`.  One 32-byte load gives a query its whole list in registers
// (the evaluation walks it with shifts: no dependent load per edge).
struct alignas(32) PolyRec { unsigned w[8]; };
constexpr int kPolyInline = 15;

struct PolyLevel {
  double x0, y0, inv_h;   // cell (ix, iy) = floor((x - x0) * inv_h), floor((y - y0) * inv_h)
  int nx, ny;
  unsigned base;          // first record of this level in `cells` (nx * ny records)
  int pad;
};

struct PolyAccel {
  int n;                         // edges = vertices
  int nslab;
  const PolyEdge *edges;
  const PolyRec *cells;          // candidate edges per grid cell (both levels)
  const PolyRec *slabs;          // candidate edges per parity slab
  const unsigned short *over;    // lists longer than kPolyInline
  PolyLevel lv[2];               // 0 fine, 1 coarse
  double ymin, ymax, xmax, tol, slab_inv_h;
};

constexpr int kPolyMaxVerts = 4096;   // = SVSDF_MAX_POLY_VERTS (16-bit edge indices; host build time)

// isCrossRayOnXDir (SHP:1370-1383): theta = atan2 wrapped to [0, 2pi), crossing iff |theta_s - theta_e| >= pi.
// atan2(y, x) lies in (0, pi) for y > 0 and wraps into (pi, 2pi) for y < 0, so with both y != 0 the test is:
// opposite signs of y and the vector with y > 0 leading the other by more than pi counter-clockwise, i.e.
// sign(cross(s2, e2)) -- decided without atan2 unless the angle difference is within ~1e-9 rad of 0 or pi
// (sin^2 <= 1e-18), where the rounding of the two atan2 values (~1e-16) could matter and the reference
// formula itself is evaluated.
__host__ __device__ __forceinline__ bool poly_cross_ray(double s2x, double s2y, double e2x, double e2y) {
  const double crs = s2x * e2y - s2y * e2x;
  const double n2 = (s2x * s2x + s2y * s2y) * (e2x * e2x + e2y * e2y);
  if (s2y != 0.0 && e2y != 0.0 && crs * crs > 1e-18 * n2) {
    const bool sneg = s2y < 0.0, eneg = e2y < 0.0;
    return (sneg != eneg) && ((crs < 0.0) == eneg);
  }
  const double PI_ = 3.14159265358979323846;   // SHP:31
  double theta_s = atan2(s2y, s2x);
  double theta_e = atan2(e2y, e2x);
  theta_s = (theta_s < 0.0) ? (theta_s + 2 * PI_) : theta_s;
  theta_e = (theta_e < 0.0) ? (theta_e + 2 * PI_) : theta_e;
  return !(fabs(theta_s - theta_e) < PI_);
}

// dis2Seg (SHP:1385-1401) up to the closest point c; returns |p - c|^2 (the reference takes its root)
__host__ __device__ __forceinline__ double poly_edge_d2(const PolyEdge &e, double x, double y, double &cx, double &cy) {
  const double wx = x - e.sx, wy = y - e.sy;
  double t = (wx * e.vx + wy * e.vy) / e.vv;
  t = (t < 0.0) ? 0.0 : ((t > 1.0) ? 1.0 : t);   // if (t < 0) t = 0; else if (t > 1) t = 1;  (a NaN stays a NaN)
  cx = e.sx + t * e.vx;
  cy = e.sy + t * e.vy;
  const double dx = x - cx, dy = y - cy;
  return dx * dx + dy * dy;
}

// the cell of (x, y) in level lv, or -1 when outside
__host__ __device__ __forceinline__ int poly_cell(const PolyLevel &lv, double x, double y) {
  const double fx = (x - lv.x0) * lv.inv_h, fy = (y - lv.y0) * lv.inv_h;
  if (!(fx >= 0.0 && fy >= 0.0 && fx < (double)lv.nx && fy < (double)lv.ny)) return -1;   // also rejects NaN
  return (int)fy * lv.nx + (int)fx;
}

// Walk the candidate list of a record in ascending order: f(edge index, edge, start of the next edge).  The list sits in
// registers (shifted out 16 bits at a time) or, when longer than kPolyInline, in `over`; the edge of step k + 1 is
// loaded while step k computes (the loop is a dependent chain of loads otherwise).  NEXT: also the following edge's
// start = this edge's end (crossing test).
template <bool NEXT, typename F>
__host__ __device__ __forceinline__ void poly_for_each(const PolyRec &rec, const unsigned short *over, const PolyEdge *edges,
                                                       int n, F &&f) {
  unsigned w0 = rec.w[0], w1 = rec.w[1], w2 = rec.w[2], w3 = rec.w[3], w4 = rec.w[4], w5 = rec.w[5], w6 = rec.w[6], w7 = rec.w[7];
  const unsigned cnt = w0 & 0xffffu;
  if (cnt == 0u) return;
  const bool inl = cnt <= (unsigned)kPolyInline;
  const unsigned off = (w0 >> 16) | (w1 << 16);
  auto funnel = [](unsigned hi, unsigned lo) -> unsigned { return (lo >> 16) | (hi << 16); };   // v_alignbit_b32
  auto next_index = [&](unsigned k) -> unsigned {
    if (inl) {   // next 16-bit word of the 256-bit record
      w0 = funnel(w1, w0); w1 = funnel(w2, w1); w2 = funnel(w3, w2); w3 = funnel(w4, w3);
      w4 = funnel(w5, w4); w5 = funnel(w6, w5); w6 = funnel(w7, w6); w7 >>= 16;
      return w0 & 0xffffu;
    }
    return over[off + k];
  };
  unsigned idx = next_index(0);
  PolyEdge cur = edges[idx];
  double nsx = 0.0, nsy = 0.0;
  if constexpr (NEXT) { const PolyEdge &nx = edges[(idx + 1u == (unsigned)n) ? 0u : idx + 1u]; nsx = nx.sx; nsy = nx.sy; }
  for (unsigned k = 0; k < cnt; ++k) {
    const PolyEdge e = cur;
    const double ex = nsx, ey = nsy;
    const int i = (int)idx;
    if (k + 1u < cnt) {
      idx = next_index(k + 1u);
      cur = edges[idx];
      if constexpr (NEXT) { const PolyEdge &nx = edges[(idx + 1u == (unsigned)n) ? 0u : idx + 1u]; nsx = nx.sx; nsy = nx.sy; }
    }
    f(i, e, ex, ey);
  }
}

// Polygon::getonlySDF (SHP:1448-1476).  CLOSEST: also the closest point the reference's loop ends with (first edge
// among equal rooted distances; needed by the analytic gradient SHP:1505-1531) -- then the per-edge roots are taken
// like the reference does; value only: min_i sqrt(d2_i) == sqrt(min_i d2_i) exactly (sqrt is correctly rounded and
// monotone), one root per evaluation.  `edges` = pa.edges or a copy of it (LDS).
template <bool CLOSEST>
__host__ __device__ inline double poly_sdf(const PolyAccel &pa, const PolyEdge *edges, double x, double y, double *cminx,
                                           double *cminy) {
  double best = CLOSEST ? 1e9 : 1e300, mx = 0.0, my = 0.0;
  auto visit = [&](int, const PolyEdge &e, double, double) {
    double cx, cy;
    const double d2 = poly_edge_d2(e, x, y, cx, cy);
    if constexpr (CLOSEST) {
      const double dis = sqrt(d2);
      if (dis < best) { best = dis; mx = cx; my = cy; }
    } else {
      best = (d2 < best) ? d2 : best;
    }
  };
  int cell = poly_cell(pa.lv[0], x, y);
  unsigned base = pa.lv[0].base;
  if (cell < 0) { cell = poly_cell(pa.lv[1], x, y); base = pa.lv[1].base; }
  // the slab of the query's ray, when any edge can cross it
  const bool ray = y >= pa.ymin - pa.tol && y <= pa.ymax + pa.tol && x <= pa.xmax + pa.tol;
  const double fs = (y - pa.ymin) * pa.slab_inv_h;
  const int slab = !(fs >= 0.0) ? 0 : (fs >= (double)pa.nslab) ? pa.nslab - 1 : (int)fs;
  // both candidate records are fetched before either list is walked (two independent loads in flight instead of the
  // second one waiting behind the distance loop)
  PolyRec rec_d, rec_p;
  for (int k = 0; k < 8; ++k) { rec_d.w[k] = 0u; rec_p.w[k] = 0u; }
  if (cell >= 0) rec_d = pa.cells[base + (unsigned)cell];
  if (ray) rec_p = pa.slabs[slab];
  if (cell >= 0) {
    poly_for_each<false>(rec_d, pa.over, edges, pa.n, visit);
  } else {
    for (int i = 0; i < pa.n; ++i) visit(i, edges[i], 0.0, 0.0);
  }
  int rs = 0;
  poly_for_each<true>(rec_p, pa.over, edges, pa.n, [&](int, const PolyEdge &e, double ex, double ey) {
    if (poly_cross_ray(e.sx - x, e.sy - y, ex - x, ey - y)) rs++;   // end of edge i = start of the next edge
  });
  double dis_min;
  if constexpr (CLOSEST) {
    dis_min = best;
    *cminx = mx; *cminy = my;
  } else {
    const double r = sqrt(best);
    dis_min = (r < 1e9) ? r : 1e9;   // the reference's running minimum starts at 1e9
  }
  return (rs % 2 == 0) ? dis_min : -dis_min;
}

// ------------------------------------------------------------------------------------------------------------------
// Host: candidate lists of an outline (xy interleaved, n vertices; edge i = vertex i -> vertex (i + 1) % n).
// ------------------------------------------------------------------------------------------------------------------
struct PolyAccelHost {
  std::vector<PolyEdge> edges;
  std::vector<PolyRec> cells, slabs;
  std::vector<unsigned short> over;
  PolyAccel hdr{};   // pointers unset
  // statistics
  size_t cand_total = 0, cand_max = 0, slab_max = 0;
};

namespace poly_detail {
struct Seg { double sx, sy, ex, ey, vx, vy, vv; };
inline double seg_point_dist(const Seg &e, double x, double y) {
  const double wx = x - e.sx, wy = y - e.sy;
  double t = (e.vv > 0.0) ? (wx * e.vx + wy * e.vy) / e.vv : 0.0;
  t = std::min(1.0, std::max(0.0, t));
  const double dx = x - (e.sx + t * e.vx), dy = y - (e.sy + t * e.vy);
  return std::sqrt(dx * dx + dy * dy);
}
inline double rect_point_dist(double x0, double y0, double x1, double y1, double x, double y) {
  const double dx = std::max(std::max(x0 - x, 0.0), x - x1), dy = std::max(std::max(y0 - y, 0.0), y - y1);
  return std::hypot(dx, dy);
}
// does the segment meet the closed rectangle?  (Liang-Barsky clip of the parameter range)
inline bool seg_meets_rect(const Seg &e, double x0, double y0, double x1, double y1) {
  double t0 = 0.0, t1 = 1.0;
  const double p[4] = {-e.vx, e.vx, -e.vy, e.vy};
  const double q[4] = {e.sx - x0, x1 - e.sx, e.sy - y0, y1 - e.sy};
  for (int k = 0; k < 4; ++k) {
    if (p[k] == 0.0) { if (q[k] < 0.0) return false; continue; }
    const double r = q[k] / p[k];
    if (p[k] < 0.0) { if (r > t1) return false; t0 = std::max(t0, r); }
    else { if (r < t0) return false; t1 = std::min(t1, r); }
  }
  return t0 <= t1;
}
// exact distance between a closed rectangle and a segment (two convex sets: zero when they meet, else attained at a
// vertex of one of them)
inline double rect_seg_dist(const Seg &e, double x0, double y0, double x1, double y1) {
  if (seg_meets_rect(e, x0, y0, x1, y1)) return 0.0;
  double d = std::min(rect_point_dist(x0, y0, x1, y1, e.sx, e.sy), rect_point_dist(x0, y0, x1, y1, e.ex, e.ey));
  d = std::min(d, seg_point_dist(e, x0, y0));
  d = std::min(d, seg_point_dist(e, x1, y0));
  d = std::min(d, seg_point_dist(e, x0, y1));
  d = std::min(d, seg_point_dist(e, x1, y1));
  return d;
}
// a candidate list (ascending edge indices) as a record; long lists go to `over`
inline PolyRec pack_list(const std::vector<unsigned short> &list, std::vector<unsigned short> &over) {
  unsigned short w[16] = {0};
  w[0] = (unsigned short)list.size();
  if (list.size() <= (size_t)kPolyInline) {
    for (size_t k = 0; k < list.size(); ++k) w[1 + k] = list[k];
  } else {
    const unsigned off = (unsigned)over.size();
    w[1] = (unsigned short)(off & 0xffffu);
    w[2] = (unsigned short)(off >> 16);
    over.insert(over.end(), list.begin(), list.end());
  }
  PolyRec r;
  for (int k = 0; k < 8; ++k) r.w[k] = (unsigned)w[2 * k] | ((unsigned)w[2 * k + 1] << 16);
  return r;
}
}  // namespace poly_detail

// returns false when the outline cannot be handled (n out of range, non-finite vertex)
inline bool build_poly_accel(const double *xy, int n, PolyAccelHost &out, int ng_fine = 128, int ng_coarse = 256,
                             int nslab = 256) {
  using namespace poly_detail;
  if (n < 3 || n > kPolyMaxVerts) return false;
  out = PolyAccelHost{};
  out.edges.resize(n);
  std::vector<Seg> seg(n);
  double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
  for (int i = 0; i < n; ++i) {
    const int j = (i + 1 == n) ? 0 : i + 1;
    Seg &e = seg[i];
    e.sx = xy[2 * i]; e.sy = xy[2 * i + 1]; e.ex = xy[2 * j]; e.ey = xy[2 * j + 1];
    if (!std::isfinite(e.sx) || !std::isfinite(e.sy)) return false;
    e.vx = e.ex - e.sx; e.vy = e.ey - e.sy;       // Eigen::Vector2d v = end - start
    e.vv = e.vx * e.vx + e.vy * e.vy;             // v.squaredNorm()
    out.edges[i] = PolyEdge{e.sx, e.sy, e.vx, e.vy, e.vv};
    xmin = std::min(xmin, e.sx); xmax = std::max(xmax, e.sx);
    ymin = std::min(ymin, e.sy); ymax = std::max(ymax, e.sy);
  }
  const double L = std::max(std::max(xmax - xmin, ymax - ymin), 1e-6);
  const double scale = std::max(L, std::max(std::max(std::fabs(xmin), std::fabs(xmax)), std::max(std::fabs(ymin), std::fabs(ymax))));
  PolyAccel &h = out.hdr;
  h.n = n;
  // ---- distance grids
  const double margins[2] = {0.25 * L, 3.0 * L};
  const int ngs[2] = {ng_fine, ng_coarse};
  const double grow = 1e-7 * scale;   // every cell is enlarged by this on all sides: a query whose cell index is decided
                                      // by the last bit of (x - x0) * inv_h is still covered by the neighbour's list
  std::vector<double> ub(n), row[2];
  std::vector<unsigned short> list;
  for (int l = 0; l < 2; ++l) {
    PolyLevel &lv = h.lv[l];
    const double m = margins[l];
    const double ext = L + 2.0 * m;
    const int ng = std::max(1, ngs[l]);
    const double hcell = ext / ng;
    lv.x0 = 0.5 * (xmin + xmax) - 0.5 * ext;
    lv.y0 = 0.5 * (ymin + ymax) - 0.5 * ext;
    lv.inv_h = 1.0 / hcell;
    lv.nx = ng; lv.ny = ng;
    lv.base = (unsigned)out.cells.size();
    lv.pad = 0;
    // distances grid node -> edge, one node row at a time (a node is a corner of up to four cells)
    auto fill_row = [&](std::vector<double> &r, int iy) {
      r.resize((size_t)(ng + 1) * n);
      const double y = lv.y0 + iy * hcell;
      for (int ix = 0; ix <= ng; ++ix) {
        const double x = lv.x0 + ix * hcell;
        for (int i = 0; i < n; ++i) r[(size_t)ix * n + i] = seg_point_dist(seg[i], x, y);
      }
    };
    fill_row(row[0], 0);
    const double diam = 1.4143 * (hcell + 2.0 * grow);
    for (int iy = 0; iy < ng; ++iy) {
      fill_row(row[(iy + 1) & 1], iy + 1);
      const std::vector<double> &r0 = row[iy & 1], &r1 = row[(iy + 1) & 1];
      for (int ix = 0; ix < ng; ++ix) {
        const double cx0 = lv.x0 + ix * hcell - grow, cx1 = lv.x0 + (ix + 1) * hcell + grow;
        const double cy0 = lv.y0 + iy * hcell - grow, cy1 = lv.y0 + (iy + 1) * hcell + grow;
        // U >= the nearest-edge distance of every point of the enlarged cell: the distance to a segment is convex
        // (maximum over the cell at a corner) and 1-Lipschitz (the enlarged corners are within 1.4143 grow of the nodes)
        double U = 1e300;
        for (int i = 0; i < n; ++i) {
          const double dmax = std::max(std::max(r0[(size_t)ix * n + i], r0[(size_t)(ix + 1) * n + i]),
                                       std::max(r1[(size_t)ix * n + i], r1[(size_t)(ix + 1) * n + i])) + 1.4143 * grow;
          // (a zero-length edge -- repeated vertex -- never wins the reference's `dis < dis_min`: 0/0 gives NaN)
          ub[i] = (seg[i].vv > 0.0) ? dmax : 1e300;
          U = std::min(U, ub[i]);
        }
        const double thr = U * (1.0 + 1e-9) + 1e-9 * scale;
        list.clear();
        for (int i = 0; i < n; ++i) {
          if (ub[i] - diam > thr) continue;   // cheap reject (1-Lipschitz): every point of the cell is farther than thr
          if (rect_seg_dist(seg[i], cx0, cy0, cx1, cy1) <= thr) list.push_back((unsigned short)i);
        }
        out.cand_total += list.size();
        out.cand_max = std::max(out.cand_max, list.size());
        out.cells.push_back(pack_list(list, out.over));
      }
    }
  }
  // ---- parity slabs
  h.nslab = std::max(1, nslab);
  h.ymin = ymin; h.ymax = ymax; h.xmax = xmax;
  h.tol = 1e-7 * scale;
  const double hs = std::max(ymax - ymin, 1e-300) / h.nslab;
  h.slab_inv_h = 1.0 / hs;
  for (int s = 0; s < h.nslab; ++s) {
    // queries mapped to slab s have y in [ymin + s hs, ymin + (s + 1) hs] up to rounding of the index (the first and
    // last slab also take the queries within tol outside the y-range); listed: edges within 2 tol of that interval
    const double a = ymin + s * hs - ((s == 0) ? 3.0 : 2.0) * h.tol, b = ymin + (s + 1) * hs + ((s + 1 == h.nslab) ? 3.0 : 2.0) * h.tol;
    list.clear();
    for (int i = 0; i < n; ++i) {
      const Seg &e = seg[i];
      if (std::max(e.sy, e.ey) >= a && std::min(e.sy, e.ey) <= b) list.push_back((unsigned short)i);
    }
    out.slab_max = std::max(out.slab_max, list.size());
    out.slabs.push_back(pack_list(list, out.over));
  }
  return true;
}
}  // namespace svsdf
