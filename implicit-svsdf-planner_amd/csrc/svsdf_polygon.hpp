// svsdf_polygon.hpp -- the generic robot shape: a closed 2-D outline (reference: class Polygon, Shape.hpp:1352-1531).
//
// BASELINE config 5 ("arbitrary .obj mesh, no analytic shape SDF") reaches this path through the z = 0 outline of the
// mesh (csrc/svsdf_mesh.hpp): 77 ... 754 vertices for the reference's shapes/*.obj.  Polygon::getonlySDF
// (SHP:1448-1476) is a loop over ALL edges per evaluation -- distance to the segment (dis2Seg, SHP:1385-1401) and an
// atan2-based crossing test against the +x ray (isCrossRayOnXDir, SHP:1370-1383).  This file evaluates the same value,
// bit for bit, from candidate lists built once per outline on the host:
//
//   distance  three uniform grids in the body frame: a fine one around the outline, a coarse one out to ~3 shape sizes,
//             a far one out to 40 (beyond that: all edges).  A cell lists, in ascending edge order, every edge that can
//             be the nearest one for SOME point of the (slightly enlarged) cell.  The lists come down a pyramid of grids
//             (16, 32, .. cells per side; one more halving below the fine and the coarse level, united per cell): a cell
//             filters its parent's list with one sample, its centre c (half diagonal rho): with e* the nearest edge at c,
//             an edge e stays iff  d(c, e) - d(c, e*) <= lip rho,  lip = min(2, diam(e u e*) / (d(c, e*) - rho)) bounding
//             |grad (d_e - d_e*)| = the difference of the two unit directions: far from a rounded corner its short edges
//             are all nearly equidistant but their bisectors fan out.  An edge that is not listed is farther than the
//             nearest listed one by a margin (1e-9) far above the rounding of the per-edge arithmetic, so the minimum
//             over the listed edges is the minimum over all edges: the same double the reference's loop ends with.
//   parity    a cell that stays clear of the outline knows the parity of the crossing count of all its queries (the
//             reference's test only depends on rounding within ~1e-15 of an edge); a cell the outline passes through
//             carries its own crossing candidates in the upper half of its record (plus the parity of the edges that
//             cross every ray of the cell); what does not fit, and a query outside the grids, uses horizontal slabs x
//             buckets of x over the outline's bounding box: a slab lists the edges whose y-range (enlarged by `tol`)
//             meets it and that are not entirely to the left of the bucket.  An edge with both end points on one side
//             of the query's ray line by more than the rounding of two atan2 calls can never satisfy
//             |theta_s - theta_e| >= PI, so only the listed edges are put through the reference's test; a query above /
//             below the outline or to the right of it counts no crossing.
//
// The per-edge arithmetic is the reference's, operation for operation, except that the quotient of dis2Seg is obtained
// from the edge's reciprocal by two residual steps that provably end in the division's own rounding (poly_quot); no
// contraction; v = end - start and v.squaredNorm() are the same IEEE operations whether done here per evaluation or once
// on the host.  A wave walks its lanes' lists in step (poly_for_each); a lane the fast path cannot serve (a list longer
// than its record, an operand outside poly_quot's range, a crossing angle within 1e-9 rad of 0 or pi, a query outside
// all grids) re-evaluates on its own with the division and the reference's atan2 formula.
// __host__ __device__: tests/cpp/poly_host.cpp runs the very same functions on the CPU against the oracle's plain loop.
//
// Outlines of several closed loops (round 5; a section with a hole, two solids -- BASELINE config 5 says "arbitrary .obj
// mesh").  Polygon::getonlySDF is a minimum over edges and a crossing count over edges: it does not care how the edges are
// chained.  The edge array holds the loops one after the other, each CLOSED BY A COPY OF ITS FIRST VERTEX whose own edge
// (it would run on to the next loop) is dead: a zero-length edge, v = 0, vv = 0.  Every real edge i still ends at entry
// i + 1, a zero-length edge can neither win the distance (the reference's 0 / 0 = NaN never passes `dis < dis_min`) nor
// count as a crossing (theta_s == theta_e), and it is kept out of every candidate list -- so the device code is the
// single-loop code, unchanged.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

namespace svsdf {

// rv = 1 / vv, correctly rounded (host): the quotient of dis2Seg is refined from it (poly_quot)
struct alignas(16) PolyEdge { double sx, sy, vx, vy, vv, rv; };   // start, v = end - start, v.squaredNorm(); the end is the next edge's start
constexpr int kPolyEdgeDoubles = 6;

// A candidate list in one 32-byte record of 16-bit words h[0 .. 15]: h[15] = count | parity state << 14; count <= 15:
// h[0 .. count - 1] = the edges, ascending; count > 15: h[0] | h[1] << 16 = offset of the list in `over`.  One 32-byte
// load gives a query its whole list in registers (no dependent load per edge).  Parity state (records of grid cells
// only): 0 = take the crossing list of the ray's slab and bucket, 1 = every query of the cell is outside (even
// crossings), 2 = inside, 3 = the cell's own crossing list is in the same record: h[15] = distance count (4 bits) |
// crossing count << 4 (4 bits) | parity of the edges that cross every ray of the cell << 8 | 3 << 14, distance edges in
// h[0 .. 7], crossing edges in h[8 .. 14].
struct alignas(32) PolyRec { unsigned w[8]; };
constexpr int kPolyInline = 15;
constexpr unsigned kPolyCountMask = 0x1fffu;   // counts up to kPolyMaxVerts = 4096

struct PolyLevel {
  double x0, y0, inv_h;   // cell (ix, iy) = [x0 + ix h, x0 + (ix + 1) h] x [y0 + iy h, ..], h = 1 / inv_h (host layout; poly_locate)
  int nx, ny;
  unsigned base;          // first record of this level in `cells` (nx * ny records)
  int pad;
};

struct PolyAccel {
  int n;                         // edges = vertices
  int nslab;
  const PolyEdge *edges;
  const PolyRec *cells;          // candidate edges per grid cell (three levels) + what the cell knows about the crossing parity
  const PolyRec *slabs;          // crossing-parity candidates per (slab of y, bucket of x): nslab * nxb records
  const unsigned short *over;    // lists longer than kPolyInline
  PolyLevel lv[3];               // 0 fine (around the outline), 1 coarse (to 3 shape sizes), 2 far (to 40 shape sizes)
  double ymin, ymax, xmax, tol, slab_inv_h;
  double xmin, xb_inv_h;         // x buckets of the parity lists: [xmin, xmax] in nxb buckets (left of xmin: bucket 0)
  int nxb;
  int div_ok;                    // every edge's vv in [1e-100, 1e100]: poly_quot may refine instead of dividing
  // the three levels as poly_locate reads them: common centre, half extent (shrunk by 1e-9), 1 / cell size, first record
  double cx, cy, lr[3], linv[3];
  unsigned lbase[3];
  int pad2;
};

#ifndef SVSDF_POLY_REFINE
#define SVSDF_POLY_REFINE 2   // fine / coarse level: cells per side of the sub-grid whose lists a cell unites (1: none, 2, 4)
#endif
constexpr int kPolyGrid = 256;        // cells per side of every grid level (poly_locate)
constexpr int kPolyMaxVerts = 8190;   // = SVSDF_MAX_POLY_VERTS: entries of the edge array incl. the loops' closing copies (list counts are 13 bits)

// true when the condition holds in any lane of the wave (the host build has one lane)
#if defined(__HIP_DEVICE_COMPILE__)
#define SVSDF_WAVE_ANY(c) (__any((int)(c)) != 0)
#else
#define SVSDF_WAVE_ANY(c) (c)
#endif
// a wave-uniform value, as a scalar register (readfirstlane of a value the compiler already knows to be uniform folds away)
__host__ __device__ __forceinline__ unsigned poly_uniform(unsigned v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
#else
  return v;
#endif
}
__host__ __device__ __forceinline__ double poly_uniform(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
#else
  return v;
#endif
}

// isCrossRayOnXDir (SHP:1370-1383): theta = atan2 wrapped to [0, 2pi), crossing iff |theta_s - theta_e| >= pi.
// atan2(y, x) lies in (0, pi) for y > 0 and wraps into (pi, 2pi) for y < 0, so with both y != 0 the test is:
// opposite signs of y and the vector with y > 0 leading the other by more than pi counter-clockwise, i.e.
// sign(cross(s2, e2)) -- decided without atan2 unless the angle difference is within ~1e-9 rad of 0 or pi
// (sin^2 <= 1e-18), where the rounding of the two atan2 values (~1e-16) could matter and the reference
// formula itself is evaluated.  poly_cross_fast: the decision when it is certain (returns false: not certain).
__host__ __device__ __forceinline__ bool poly_cross_fast(double s2x, double s2y, double e2x, double e2y, bool &cross) {
  const double crs = s2x * e2y - s2y * e2x;
  const double n2 = (s2x * s2x + s2y * s2y) * (e2x * e2x + e2y * e2y);
  const bool sneg = s2y < 0.0, eneg = e2y < 0.0;
  cross = (sneg != eneg) & ((crs < 0.0) == eneg);
  return (s2y != 0.0) & (e2y != 0.0) & (crs * crs > 1e-18 * n2);
}
__host__ __device__ inline bool poly_cross_ray(double s2x, double s2y, double e2x, double e2y) {
  bool cross;
  if (poly_cross_fast(s2x, s2y, e2x, e2y, cross)) return cross;
  const double PI_ = 3.14159265358979323846;   // SHP:31
  double theta_s = atan2(s2y, s2x);
  double theta_e = atan2(e2y, e2x);
  theta_s = (theta_s < 0.0) ? (theta_s + 2 * PI_) : theta_s;
  theta_e = (theta_e < 0.0) ? (theta_e + 2 * PI_) : theta_e;
  return !(fabs(theta_s - theta_e) < PI_);
}

// a / b as the division rounds it, from rb = RN(1 / b): q0 = RN(a rb) is within 2 ulp, one residual step
// (r = a - q b exactly, q += r rb) leaves it within 1 ulp, and from there the same step gives RN(a / b) (Markstein's
// theorem: correctly rounded reciprocal + faithful quotient).  Five full-rate operations instead of the division's
// thirteen (one of them the quarter-rate v_rcp_f64).  Valid while nothing can over- or underflow: b in [1e-100, 1e100]
// (PolyAccel::div_ok, checked once on the host) and |a| in [1e-150, 1e150] -- `inrange`; a query that ever leaves that
// range (a zero, a NaN) is re-evaluated with the division itself (poly_sdf's second path).
// tests/test_polygon_accel.py compares it with the division on adversarial operands.
__host__ __device__ __forceinline__ double poly_quot(double a, double b, double rb, bool &inrange) {
  const double fa = fabs(a);
  inrange = (fa >= 1e-150) & (fa <= 1e150);
  double q = a * rb;
  double r = __builtin_fma(-q, b, a);
  q = __builtin_fma(r, rb, q);
  r = __builtin_fma(-q, b, a);
  return __builtin_fma(r, rb, q);
}

// dis2Seg (SHP:1385-1401) up to the closest point c; returns |p - c|^2 (the reference takes its root).
// FAST: the quotient from the edge's reciprocal (poly_quot), `inrange` says whether it may be used.
template <bool FAST>
__host__ __device__ __forceinline__ double poly_edge_d2(const PolyEdge &e, double x, double y, double &cx, double &cy, bool &inrange) {
  const double wx = x - e.sx, wy = y - e.sy;
  double t;
  if constexpr (FAST) t = poly_quot(wx * e.vx + wy * e.vy, e.vv, e.rv, inrange);
  else { t = (wx * e.vx + wy * e.vy) / e.vv; inrange = true; }
  t = (t < 0.0) ? 0.0 : ((t > 1.0) ? 1.0 : t);   // if (t < 0) t = 0; else if (t > 1) t = 1;  (a NaN stays a NaN)
  cx = e.sx + t * e.vx;
  cy = e.sy + t * e.vy;
  const double dx = x - cx, dy = y - cy;
  return dx * dx + dy * dy;
}

// the innermost grid level that holds (x, y): its cell and the level's first record; -1 outside all three.  The levels
// share their centre, so the max-norm distance to it picks the level and ONE index computation follows (any level that
// contains the query would do: each level's lists are complete on their own).  The host lays the cells out as
// x0 + ix h; the two ways of computing an index differ by rounding only, which the cells' enlargement (`grow`) covers.
__host__ __device__ __forceinline__ int poly_locate(const PolyAccel &pa, double x, double y, unsigned &base) {
  const double dx = x - pa.cx, dy = y - pa.cy;
  const double ax = fabs(dx), ay = fabs(dy);
  const double m = (ax > ay) ? ax : ay;
  const bool l0 = m < pa.lr[0], l1 = m < pa.lr[1];
  // (the header's values are pinned in scalar registers first: left to itself the compiler selects an ADDRESS per lane
  // and loads the value from the header in memory, dependent loads in front of the record's)
  const double i0 = poly_uniform(pa.linv[0]), i1 = poly_uniform(pa.linv[1]), i2 = poly_uniform(pa.linv[2]);
  const unsigned b0 = poly_uniform(pa.lbase[0]), b1 = poly_uniform(pa.lbase[1]), b2 = poly_uniform(pa.lbase[2]);
  const double inv = l0 ? i0 : l1 ? i1 : i2;
  base = l0 ? b0 : l1 ? b1 : b2;
  const double hn = 0.5 * kPolyGrid;
  const double fx = dx * inv + hn, fy = dy * inv + hn, n = hn + hn;
  if (!((fx >= 0.0) & (fy >= 0.0) & (fx < n) & (fy < n))) return -1;   // also rejects NaN and anything beyond the far level
  return (int)fy * kPolyGrid + (int)fx;
}

__host__ __device__ __forceinline__ unsigned poly_count(const PolyRec &rec) { return (rec.w[7] >> 16) & kPolyCountMask; }

// edge k of a record's list (a lane's own walk; the wave's walk below never shifts through the record like this)
__host__ __device__ __forceinline__ unsigned poly_list_index(const PolyRec &rec, const unsigned short *over, unsigned cnt, unsigned k) {
  if (cnt > (unsigned)kPolyInline) return over[rec.w[0] + k];
  const unsigned j = k >> 1;
  const unsigned wk = (j == 0u) ? rec.w[0] : (j == 1u) ? rec.w[1] : (j == 2u) ? rec.w[2] : (j == 3u) ? rec.w[3]
                    : (j == 4u) ? rec.w[4] : (j == 5u) ? rec.w[5] : (j == 6u) ? rec.w[6] : rec.w[7];
  return (k & 1u) ? (wk >> 16) : (wk & 0xffffu);
}

// The WAVE walks the (inline) candidate lists of its lanes in step, ascending: f(valid, edge, start of the next edge).
// The trip count is the longest list among the lanes (a scalar branch, no exec-mask bookkeeping per edge), two edges per
// step out of one word of the record, and a lane whose list has ended (or is empty: cnt = 0) repeats its first edge
// with valid = false.  No lane-dependent branch anywhere in the body.
// NEXT: also the following edge's start = this edge's end (crossing test).
template <bool NEXT, typename F>
__host__ __device__ __forceinline__ void poly_for_each(const PolyRec &rec, unsigned cnt, const PolyEdge *edges, int n, F &&f) {
  if (!SVSDF_WAVE_ANY(cnt != 0u)) return;
  auto next_start = [&](unsigned idx, double &nsx, double &nsy) {
    const PolyEdge &nx = edges[(idx + 1u == (unsigned)n) ? 0u : idx + 1u];
    nsx = nx.sx; nsy = nx.sy;
  };
  const unsigned first = (cnt != 0u) ? (rec.w[0] & 0xffffu) : 0u;
  auto pair = [&](unsigned wj, unsigned k) {   // edges k, k + 1 of the list = the two halves of word k / 2
    const bool v0 = k < cnt, v1 = k + 1u < cnt;
    const unsigned i0 = v0 ? (wj & 0xffffu) : first, i1 = v1 ? (wj >> 16) : first;
    const PolyEdge e0 = edges[i0], e1 = edges[i1];
    double e0x = 0.0, e0y = 0.0, e1x = 0.0, e1y = 0.0;
    if constexpr (NEXT) { next_start(i0, e0x, e0y); next_start(i1, e1x, e1y); }
    f(v0, e0, e0x, e0y);
    f(v1, e1, e1x, e1y);
  };
  // the record as four 64-bit words, moved down one word every four edges (three 64-bit moves)
  unsigned long long q0 = rec.w[0] | ((unsigned long long)rec.w[1] << 32), q1 = rec.w[2] | ((unsigned long long)rec.w[3] << 32),
                     q2 = rec.w[4] | ((unsigned long long)rec.w[5] << 32), q3 = rec.w[6] | ((unsigned long long)rec.w[7] << 32);
#pragma nounroll   // (peeled and unrolled it is 16 copies of the body at each of a kernel's dozen evaluation sites)
  for (unsigned k = 0;; k += 4u) {
    pair((unsigned)q0, k);
    if (!SVSDF_WAVE_ANY(k + 2u < cnt)) break;
    pair((unsigned)(q0 >> 32), k + 2u);
    if (!SVSDF_WAVE_ANY(k + 4u < cnt)) break;
    q0 = q1; q1 = q2; q2 = q3;
  }
}

// Polygon::getonlySDF (SHP:1448-1476).  CLOSEST: also the closest point the reference's loop ends with (first edge
// among equal rooted distances; needed by the analytic gradient SHP:1505-1531) -- then the per-edge roots are taken
// like the reference does; value only: min_i sqrt(d2_i) == sqrt(min_i d2_i) exactly (sqrt is correctly rounded and
// monotone), one root per evaluation.  `edges` = pa.edges or a copy of it (LDS).
//
// Two paths with the same result.  The wave's path: both lists walked in step (poly_for_each), quotient from the edge's
// reciprocal, crossing decided by the sign of a cross product, parity of a cell that is clear of the outline read from
// its record.  The lane's own path, taken by the lanes the first one cannot serve -- a query outside both grids (all
// edges), a list longer than kPolyInline, a projection outside poly_quot's range, a crossing angle within 1e-9 rad of 0
// or pi: the lists (or all edges) walked by the lane with the division and the reference's atan2 formula.
template <bool CLOSEST>
__host__ __device__ inline double poly_sdf(const PolyAccel &pa, const PolyEdge *edges, double x, double y, double *cminx,
                                           double *cminy) {
  unsigned base;
  const int cell = poly_locate(pa, x, y, base);
  // one 32-byte record per query: the cell's distance candidates and what it knows about the crossing parity
  // (1 even / 2 odd for the whole cell; 3: the cell's own short list in words 4 .. 7 of the record, with the parity of the
  // edges that cross every ray of the cell; 0: the list of the ray's slab and bucket -- a dependent second fetch, rare)
  const PolyRec rec_d = pa.cells[base + (unsigned)((cell >= 0) ? cell : 0)];
  const unsigned h15 = (cell >= 0) ? (rec_d.w[7] >> 16) : 0u;
  const unsigned pstate = h15 >> 14;
  const bool packed = pstate == 3u;
  PolyRec rec_p;
  rec_p.w[0] = rec_d.w[4]; rec_p.w[1] = rec_d.w[5]; rec_p.w[2] = rec_d.w[6]; rec_p.w[3] = rec_d.w[7] & 0xffffu;
  rec_p.w[4] = 0u; rec_p.w[5] = 0u; rec_p.w[6] = 0u; rec_p.w[7] = 0u;
  unsigned cnt_p = packed ? ((h15 >> 4) & 0xfu) : 0u;
  if (SVSDF_WAVE_ANY(pstate == 0u)) {
    const bool ray = (y >= pa.ymin - pa.tol) & (y <= pa.ymax + pa.tol) & (x <= pa.xmax + pa.tol);
    if (ray & (pstate == 0u)) {
      const double fs = (y - pa.ymin) * pa.slab_inv_h;
      const int slab = !(fs >= 0.0) ? 0 : (fs >= (double)pa.nslab) ? pa.nslab - 1 : (int)fs;
      const double fx = (x - pa.xmin) * pa.xb_inv_h;
      const int xb = !(fx >= 0.0) ? 0 : (fx >= (double)pa.nxb) ? pa.nxb - 1 : (int)fx;
      rec_p = pa.slabs[slab * pa.nxb + xb];
      cnt_p = poly_count(rec_p);
    }
  }
  const unsigned cnt_d = packed ? (h15 & 0xfu) : (h15 & kPolyCountMask);
  const int rs0 = (int)((pstate == 2u) | (packed & (((h15 >> 8) & 1u) != 0u)));
  bool own = (cell < 0) | (cnt_d > (unsigned)kPolyInline) | (cnt_p > (unsigned)kPolyInline) | (pa.div_ok == 0);

  double best = CLOSEST ? 1e9 : 1e300, mx = 0.0, my = 0.0;
  poly_for_each<false>(rec_d, own ? 0u : cnt_d, edges, pa.n, [&](bool valid, const PolyEdge &e, double, double) {
    double cx, cy;
    bool inrange;
    const double d2 = poly_edge_d2<true>(e, x, y, cx, cy, inrange);
    own = own | (valid & !inrange);
    if constexpr (CLOSEST) {
      const double dis = sqrt(d2);
      const bool up = valid & (dis < best);
      best = up ? dis : best; mx = up ? cx : mx; my = up ? cy : my;
    } else {
      best = (valid & (d2 < best)) ? d2 : best;
    }
  });
  int rs = rs0;
  poly_for_each<true>(rec_p, own ? 0u : cnt_p, edges, pa.n, [&](bool valid, const PolyEdge &e, double ex, double ey) {
    bool cross;   // end of edge i = start of the next edge
    const bool sure = poly_cross_fast(e.sx - x, e.sy - y, ex - x, ey - y, cross);
    rs += (valid & cross) ? 1 : 0;
    own = own | (valid & !sure);
  });
  if (own) {
    best = CLOSEST ? 1e9 : 1e300; mx = 0.0; my = 0.0;
    const unsigned nd = (cell >= 0) ? cnt_d : (unsigned)pa.n;
    for (unsigned k = 0; k < nd; ++k) {
      const PolyEdge &e = edges[(cell >= 0) ? poly_list_index(rec_d, pa.over, cnt_d, k) : k];
      double cx, cy;
      bool inrange;
      const double d2 = poly_edge_d2<false>(e, x, y, cx, cy, inrange);
      if constexpr (CLOSEST) {
        const double dis = sqrt(d2);
        if (dis < best) { best = dis; mx = cx; my = cy; }
      } else {
        best = (d2 < best) ? d2 : best;
      }
    }
    rs = rs0;
    for (unsigned k = 0; k < cnt_p; ++k) {
      const unsigned idx = poly_list_index(rec_p, pa.over, cnt_p, k);
      const PolyEdge &e = edges[idx], &nx = edges[(idx + 1u == (unsigned)pa.n) ? 0u : idx + 1u];
      if (poly_cross_ray(e.sx - x, e.sy - y, nx.sx - x, nx.sy - y)) rs++;
    }
  }
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
  size_t cand_total = 0, cand_max = 0, slab_max = 0, par_total = 0, par_max = 0, cells_known = 0, cells_packed = 0, far_total = 0, far_max = 0;
};

namespace poly_detail {
struct Seg { double sx, sy, ex, ey, vx, vy, vv, rvv; };   // rvv = 1 / vv (0 for a zero-length edge)
// (list building only: every use carries 1e-9 margins, so the product with the reciprocal stands in for the division)
inline double seg_point_dist(const Seg &e, double x, double y) {
  const double wx = x - e.sx, wy = y - e.sy;
  double t = (wx * e.vx + wy * e.vy) * e.rvv;
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
inline PolyRec pack_list(const std::vector<unsigned short> &list, std::vector<unsigned short> &over, unsigned pstate = 0) {
  unsigned short w[16] = {0};
  w[15] = (unsigned short)(list.size() | (pstate << 14));
  if (list.size() <= (size_t)kPolyInline) {
    for (size_t k = 0; k < list.size(); ++k) w[k] = list[k];
  } else {
    const unsigned off = (unsigned)over.size();
    w[0] = (unsigned short)(off & 0xffffu);
    w[1] = (unsigned short)(off >> 16);
    over.insert(over.end(), list.begin(), list.end());
  }
  PolyRec r;
  for (int k = 0; k < 8; ++k) r.w[k] = (unsigned)w[2 * k] | ((unsigned)w[2 * k + 1] << 16);
  return r;
}
}  // namespace poly_detail

// returns false when the outline cannot be handled (n out of range, non-finite vertex, loop sizes that do not add up)
// loop_sizes / nloops: the vertex list is nloops closed loops one after the other (null / < 2: one loop, the reference's
// chain).  ng_fine / ng_coarse are rounded up to a power of two in [16, 256].
inline bool build_poly_accel(const double *xy_in, int n_in, PolyAccelHost &out, int ng_fine = 128, int ng_coarse = 256,
                             int nslab = 256, int refine = SVSDF_POLY_REFINE, int nxb = 32, const int *loop_sizes = nullptr,
                             int nloops = 0) {
  using namespace poly_detail;
  if (n_in < 3) return false;
  // entries of the edge array: the vertices, plus -- for an outline of several loops -- a copy of each loop's first vertex
  // behind its last one, flagged dead (see the header comment)
  std::vector<double> xy_e;
  std::vector<char> dead;
  const double *xy = xy_in;
  int n = n_in;
  if (loop_sizes && nloops >= 2) {
    long long tot = 0;
    for (int k = 0; k < nloops; ++k) { if (loop_sizes[k] < 3) return false; tot += loop_sizes[k]; }
    if (tot != n_in) return false;
    int b = 0;
    for (int k = 0; k < nloops; ++k) {
      for (int i = 0; i < loop_sizes[k]; ++i) { xy_e.push_back(xy_in[2 * (b + i)]); xy_e.push_back(xy_in[2 * (b + i) + 1]); dead.push_back(0); }
      xy_e.push_back(xy_in[2 * b]); xy_e.push_back(xy_in[2 * b + 1]); dead.push_back(1);
      b += loop_sizes[k];
    }
    xy = xy_e.data();
    n = (int)dead.size();
  } else {
    dead.assign((size_t)n, 0);
  }
  if (n > kPolyMaxVerts) return false;
  out = PolyAccelHost{};
  out.edges.resize(n);
  std::vector<Seg> seg(n);
  double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
  int div_ok = 1;
  for (int i = 0; i < n; ++i) {
    const int j = (i + 1 == n) ? 0 : i + 1;
    Seg &e = seg[i];
    e.sx = xy[2 * i]; e.sy = xy[2 * i + 1]; e.ex = xy[2 * j]; e.ey = xy[2 * j + 1];
    if (dead[i]) { e.ex = e.sx; e.ey = e.sy; }    // the closing copy of a loop's first vertex: no edge leaves it
    if (!std::isfinite(e.sx) || !std::isfinite(e.sy)) return false;
    e.vx = e.ex - e.sx; e.vy = e.ey - e.sy;       // Eigen::Vector2d v = end - start
    e.vv = e.vx * e.vx + e.vy * e.vy;             // v.squaredNorm()
    e.rvv = (e.vv > 0.0) ? 1.0 / e.vv : 0.0;
    out.edges[i] = PolyEdge{e.sx, e.sy, e.vx, e.vy, e.vv, (e.vv > 0.0) ? 1.0 / e.vv : 0.0};
    if (e.vv > 0.0 && !(e.vv >= 1e-100 && e.vv <= 1e100)) div_ok = 0;
    xmin = std::min(xmin, e.sx); xmax = std::max(xmax, e.sx);
    ymin = std::min(ymin, e.sy); ymax = std::max(ymax, e.sy);
  }
  const double L = std::max(std::max(xmax - xmin, ymax - ymin), 1e-6);
  const double scale = std::max(L, std::max(std::max(std::fabs(xmin), std::fabs(xmax)), std::max(std::fabs(ymin), std::fabs(ymax))));
  PolyAccel &h = out.hdr;
  h.n = n;
  h.div_ok = div_ok;
  // ---- distance grids
  const double margins[2] = {0.25 * L, 3.0 * L};
  const int ngs[2] = {ng_fine, ng_coarse};
  const double grow = 1e-7 * scale;   // every cell is enlarged by this on all sides: a query whose cell index is decided
                                      // by the last bit of (x - x0) * inv_h is still covered by the neighbour's list
  std::vector<unsigned short> list, plist;
  std::vector<char> clear_of_outline;   // per cell: no edge within 1e-6 scale of it
  const int nthreads = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
  // Candidate lists of a square grid of G x G cells (origin (gx0, gy0), cell size hc; G = 16 * 2^k), by descent through a
  // pyramid of grids (16, 32, .. G cells per side, then `sub` further halvings whose cells are united into the leaf they
  // tile).  A cell filters its parent's list with one sample, its centre c (rho = half diagonal of the cell enlarged by
  // `grow`): with e* the nearest listed edge at c, g(p) = d(p, e) - d(p, e*) has |grad g| = |u_e - u_e*| (unit vectors
  // from the closest points a, b to p) <= 2 |a - b| / (d(p, e) + d(p, e*)) <= diam(e u e*) / (d(c, e*) - rho) =: lip (and
  // <= 2 always: both distances are 1-Lipschitz), so g(c) > lip rho means e* is nearer than e in the whole cell: e is the
  // nearest edge nowhere in it.  Every filter is valid for its whole cell, which contains the final cell; an edge that
  // is dropped is farther than a listed one by the 1e-9 margins, far above the rounding of the per-edge arithmetic.  Far
  // from a rounded corner all of its short edges are nearly equidistant (a bound on the distances alone keeps them all),
  // but their bisectors fan out and lip is small there.
  // Round 5: all three levels are built this way.  Rounds 3-4 built the fine and the coarse level cell by cell from the
  // distances of the cell's corner nodes to EVERY edge (first pass) and refined each list with 2 x 2 samples: 0.45 / 0.67 s
  // of svsdf_create for outlines of 614 / 754 vertices, almost all of it in first-pass lists of ~ 100 edges per cell that
  // the refinement then cut to ~ 5; the descent never holds more than a parent's (short) list.
  // (lists are kept flat per level -- start offsets + one array of edge indices: a vector per cell made the descent an
  // exercise in malloc)
  struct FlatLists {
    std::vector<unsigned> start;        // cells + 1 offsets into `idx`
    std::vector<unsigned short> idx;
    const unsigned short *list(size_t c) const { return idx.data() + start[c]; }
    size_t size(size_t c) const { return start[c + 1] - start[c]; }
  };
  auto pyramid = [&](double gx0, double gy0, double hc_leaf, int G, int sub, FlatLists &leaf) {
    FlatLists cur, nxt;
    cur.start = {0u, 0u};
    for (int i = 0; i < n; ++i) if (seg[i].vv > 0.0) cur.idx.push_back((unsigned short)i);
    cur.start[1] = (unsigned)cur.idx.size();
    auto sq = [](double x, double y) { return x * x + y * y; };
    const int Gs = G << sub;
    const double ext_ = hc_leaf * G;
    for (int g = 1; g < Gs;) {
      const int g2 = (g == 1) ? std::min(16, Gs) : 2 * g, ratio = g2 / g;
      const double hc = ext_ / g2;
      // (a cell below the leaf level stands for its part of the ENLARGED leaf: enlarged by grow as well)
      const double rho = 0.5 * std::sqrt(2.0) * (hc + 2.0 * grow) * (1.0 + 1e-9);
      // rows are independent (a cell reads its parent's list and writes its own): large levels are split over a few host
      // threads, every cell computed exactly as in the serial loop, the threads' pieces concatenated in row order
      const int nthr = std::max(1, std::min(nthreads, g2 / 2));
      std::vector<std::vector<unsigned short>> tdata(nthr);
      std::vector<std::vector<unsigned>> tcount(nthr);
      auto rows = [&](int t, int y0, int y1) {
        std::vector<double> dsub;
        std::vector<unsigned short> &data = tdata[t];
        std::vector<unsigned> &cnt = tcount[t];
        cnt.reserve((size_t)(y1 - y0) * g2);
        for (int iy = y0; iy < y1; ++iy)
          for (int ix = 0; ix < g2; ++ix) {
            const size_t pc = (size_t)(iy / ratio) * g + ix / ratio;
            const unsigned short *par = cur.list(pc);
            const size_t m = cur.size(pc);
            const size_t before = data.size();
            // (below the leaf level a list of one or two edges is not worth another sample: the children keep it)
            if (g2 > G && m <= 2) { data.insert(data.end(), par, par + m); }
            else if (m > 0) {
              const double px = gx0 + (ix + 0.5) * hc, py = gy0 + (iy + 0.5) * hc;
              dsub.resize(m);
              double dmin = 1e300;
              size_t kmin = 0;
              for (size_t k = 0; k < m; ++k) {
                dsub[k] = seg_point_dist(seg[par[k]], px, py);
                if (dsub[k] < dmin) { dmin = dsub[k]; kmin = k; }
              }
              const Seg &b = seg[par[kmin]];
              const double dlow = dmin * (1.0 - 1e-9) - rho - 1e-9 * scale;
              for (size_t k = 0; k < m; ++k) {
                // (the two ends of the test need no lip: 0 <= lip <= 2)
                const double gap = dsub[k] - dmin, slackk = 1e-9 * scale + 1e-9 * dsub[k];
                if (gap <= slackk) { data.push_back(par[k]); continue; }
                if (gap > 2.0 * rho * (1.0 + 1e-9) + slackk) continue;
                double lip = 2.0;
                if (dlow > 0.0) {
                  const Seg &e = seg[par[k]];
                  const double diam = std::sqrt(std::max(std::max(sq(e.sx - b.sx, e.sy - b.sy), sq(e.sx - b.ex, e.sy - b.ey)),
                                                         std::max(sq(e.ex - b.sx, e.ey - b.sy), sq(e.ex - b.ex, e.ey - b.ey))));
                  lip = std::min(2.0, diam * (1.0 + 1e-9) / dlow);
                }
                if (gap <= lip * rho * (1.0 + 1e-9) + slackk) data.push_back(par[k]);
              }
            }
            cnt.push_back((unsigned)(data.size() - before));
          }
      };
      if (nthr == 1) rows(0, 0, g2);
      else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; ++t) pool.emplace_back(rows, t, (int)((long long)g2 * t / nthr), (int)((long long)g2 * (t + 1) / nthr));
        for (std::thread &t : pool) t.join();
      }
      nxt.start.assign(1, 0u);
      nxt.start.reserve((size_t)g2 * g2 + 1);
      nxt.idx.clear();
      for (int t = 0; t < nthr; ++t) {
        for (unsigned c : tcount[t]) nxt.start.push_back(nxt.start.back() + c);
        nxt.idx.insert(nxt.idx.end(), tdata[t].begin(), tdata[t].end());
      }
      std::swap(cur, nxt);
      g = g2;
    }
    if (sub == 0) { std::swap(leaf, cur); return; }
    // a leaf's list = the union of the lists of the (1 << sub)^2 cells that tile it (ascending, no duplicates)
    const int r = 1 << sub;
    leaf.start.assign(1, 0u);
    leaf.start.reserve((size_t)G * G + 1);
    leaf.idx.clear();
    std::vector<unsigned short> acc, tmp;
    for (int iy = 0; iy < G; ++iy)
      for (int ix = 0; ix < G; ++ix) {
        acc.clear();
        for (int jy = 0; jy < r; ++jy)
          for (int jx = 0; jx < r; ++jx) {
            const size_t c = (size_t)(iy * r + jy) * Gs + (ix * r + jx);
            const unsigned short *l0 = cur.list(c);
            const size_t m = cur.size(c);
            tmp.resize(acc.size() + m);
            tmp.resize((size_t)(std::set_union(acc.begin(), acc.end(), l0, l0 + m, tmp.begin()) - tmp.begin()));
            acc.swap(tmp);
          }
        leaf.idx.insert(leaf.idx.end(), acc.begin(), acc.end());
        leaf.start.push_back((unsigned)leaf.idx.size());
      }
  };
  // ---- fine (around the outline) and coarse (to ~3 shape sizes) levels
  FlatLists lists;
  for (int l = 0; l < 2; ++l) {
    PolyLevel &lv = h.lv[l];
    const double m = margins[l];
    const double ext = L + 2.0 * m;
    // every level is 256 x 256 records around the common centre (poly_locate: one index formula); the fine level
    // only fills the central ngs[0] x ngs[0] block of it, the cells its extent covers -- the rest is never looked up
    // (a power of two >= 16: the pyramid halves it down; even: the filled block must sit symmetrically about the common
    // centre, poly_locate picks the level by the symmetric radius lr[l])
    const int ng = kPolyGrid;
    int nfill = 16;
    while (nfill < ngs[l] && nfill < kPolyGrid) nfill *= 2;
    const int lo = (ng - nfill) / 2, hi = lo + nfill;
    const double hcell = ext / nfill;
    lv.x0 = 0.5 * (xmin + xmax) - 0.5 * ng * hcell;
    lv.y0 = 0.5 * (ymin + ymax) - 0.5 * ng * hcell;
    lv.inv_h = 1.0 / hcell;
    lv.nx = ng; lv.ny = ng;
    lv.base = (unsigned)out.cells.size();
    lv.pad = 0;
    pyramid(lv.x0 + lo * hcell, lv.y0 + lo * hcell, hcell, nfill, (refine >= 4) ? 2 : (refine >= 2) ? 1 : 0, lists);
    for (int iy = 0; iy < ng; ++iy)
      for (int ix = 0; ix < ng; ++ix) {
        if (iy < lo || iy >= hi || ix < lo || ix >= hi) {
          out.cells.push_back(PolyRec{{0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}});
          clear_of_outline.push_back(2);   // 2: not a cell of this level
          continue;
        }
        const size_t lc = (size_t)(iy - lo) * nfill + (ix - lo);
        list.assign(lists.list(lc), lists.list(lc) + lists.size(lc));
        const std::vector<unsigned short> &lst = list;
        // distance of the enlarged cell to the outline = to the nearest of its listed edges (the nearest edge of any of
        // its points is listed)
        const double cx0 = lv.x0 + ix * hcell - grow, cx1 = lv.x0 + (ix + 1) * hcell + grow;
        const double cy0 = lv.y0 + iy * hcell - grow, cy1 = lv.y0 + (iy + 1) * hcell + grow;
        double dcell = 1e300;
        for (unsigned short i : lst) dcell = std::min(dcell, rect_seg_dist(seg[i], cx0, cy0, cx1, cy1));
        clear_of_outline.push_back(dcell > 1e-6 * scale ? 1 : 0);
        out.cand_total += lst.size();
        out.cand_max = std::max(out.cand_max, lst.size());
        out.cells.push_back(pack_list(lst, out.over));
      }
  }
  // ---- far level: 256 x 256 cells out to 40 shape sizes.  (A query this far out used to walk all edges -- and with it
  // the 63 other lanes of its wave: a tenth of the wave-level evaluations of config 5, 40 % of its vector instructions.)
  {
    PolyLevel &lv = h.lv[2];
    const double ext = L + 2.0 * 40.0 * L;
    const int ngf = 256;
    lv.x0 = 0.5 * (xmin + xmax) - 0.5 * ext;
    lv.y0 = 0.5 * (ymin + ymax) - 0.5 * ext;
    lv.inv_h = 1.0 / (ext / ngf);
    lv.nx = ngf; lv.ny = ngf;
    lv.base = (unsigned)out.cells.size();
    lv.pad = 0;
    pyramid(lv.x0, lv.y0, ext / ngf, ngf, 0, lists);
    const double M = 1e-6 * scale + grow;
    for (int iy = 0; iy < ngf; ++iy)
      for (int ix = 0; ix < ngf; ++ix) {
        const size_t lc = (size_t)iy * ngf + ix;
        list.assign(lists.list(lc), lists.list(lc) + lists.size(lc));
        const std::vector<unsigned short> &l2 = list;
        const double hc = ext / ngf;
        const double cx0 = lv.x0 + ix * hc, cx1 = cx0 + hc, cy0 = lv.y0 + iy * hc, cy1 = cy0 + hc;
        // outside the outline's bounding box (by a margin): no ray of the cell meets it, or every edge is to its left
        const bool outside = cx0 > xmax + M || cx1 < xmin - M || cy0 > ymax + M || cy1 < ymin - M;
        out.far_total += l2.size();
        out.far_max = std::max(out.far_max, l2.size());
        out.cells.push_back(pack_list(l2, out.over, outside ? 1u : 0u));
        clear_of_outline.push_back(0);   // (state set here)
      }
  }
  // ---- parity slabs
  h.nslab = std::max(1, nslab);
  h.ymin = ymin; h.ymax = ymax; h.xmax = xmax;
  h.tol = 1e-7 * scale;
  const double hs = std::max(ymax - ymin, 1e-300) / h.nslab;
  h.slab_inv_h = 1.0 / hs;
  h.nxb = std::max(1, nxb);
  h.xmin = xmin;
  const double hx = std::max(xmax - xmin, 1e-300) / h.nxb;
  h.xb_inv_h = 1.0 / hx;
  for (int s = 0; s < h.nslab; ++s) {
    // queries mapped to slab s have y in [ymin + s hs, ymin + (s + 1) hs] up to rounding of the index (the first and
    // last slab also take the queries within tol outside the y-range); listed: edges within 2 tol of that interval
    const double a = ymin + s * hs - ((s == 0) ? 3.0 : 2.0) * h.tol, b = ymin + (s + 1) * hs + ((s + 1 == h.nslab) ? 3.0 : 2.0) * h.tol;
    list.clear();
    for (int i = 0; i < n; ++i) {
      const Seg &e = seg[i];
      // (a zero-length edge -- a repeated vertex, a loop's closing copy -- has theta_s == theta_e: it never counts as a crossing,
      // and listing it would only send its queries down the lane's own path, the cross product being exactly zero)
      if (e.vv > 0.0 && std::max(e.sy, e.ey) >= a && std::min(e.sy, e.ey) <= b) list.push_back((unsigned short)i);
    }
    out.slab_max = std::max(out.slab_max, list.size());
    // ... split by x: bucket b takes the queries with x >= xmin + b hx (bucket 0 also those left of xmin; up to the
    // rounding of the index), whose ray to +x cannot meet an edge that lies entirely to the left of xmin + b hx by more
    // than 2 tol -- both angles of such an edge are in (PI/2, 3 PI/2) with a margin of ~1e-8 rad, |theta_s - theta_e| < PI
    for (int xb = 0; xb < h.nxb; ++xb) {
      const double x0 = (xb == 0) ? -1e300 : xmin + xb * hx - 3.0 * h.tol;
      plist.clear();
      for (unsigned short i : list)
        if (std::max(seg[i].sx, seg[i].ex) >= x0) plist.push_back(i);
      out.par_total += plist.size();
      out.par_max = std::max(out.par_max, plist.size());
      out.slabs.push_back(pack_list(plist, out.over));
    }
  }
  // ---- parity state of the grid cells.  A cell that stays clear of the outline by 1e-6 scale lies on one side of it,
  // and for its queries the reference's crossing count has that parity: an edge's test |theta_s - theta_e| >= PI can only
  // go either way under rounding when the query sees the edge under an angle within ~1e-15 of PI, i.e. lies within
  // ~1e-15 scale of the segment; everywhere else the test is the half-open rule (end point on the ray line = above),
  // applied with the same end-point values by both edges of a vertex.  The state is the count at the cell's centre.
  for (int l = 0; l < 2; ++l) {
    const PolyLevel &lv = h.lv[l];
    const double hcell = 1.0 / lv.inv_h;
    std::vector<int> rowedges;   // edges whose y-range holds the row's centre line: the only ones its centre rays can cross
    for (int iy = 0; iy < lv.ny; ++iy) {
      {
        const double y = lv.y0 + (iy + 0.5) * hcell;
        rowedges.clear();
        if (y >= ymin && y <= ymax)
          for (int i = 0; i < n; ++i)
            if (seg[i].vv > 0.0 && std::max(seg[i].sy, seg[i].ey) >= y && std::min(seg[i].sy, seg[i].ey) <= y) rowedges.push_back(i);
      }
      for (int ix = 0; ix < lv.nx; ++ix) {
        const size_t c = (size_t)lv.base + (size_t)iy * lv.nx + ix;
        if (clear_of_outline[c] == 2) continue;
        if (!clear_of_outline[c]) {
          // The outline passes (or may pass) through the cell: its own crossing list, when it fits the upper half of
          // the record next to a distance list of <= 8 edges.  An edge entirely outside the cell's band of y (by 2 tol)
          // or entirely to its left crosses no ray of the cell; one entirely to its right (by 1e-6 scale) whose end
          // points lie on opposite sides of the band (by 1e-6 scale) crosses every ray of it -- an angle of at least
          // ~1e-7 rad away from pi: those only flip the cell's base parity; the rest (an end point in the band, or
          // overlapping the cell in x) are the list.
          PolyRec &r = out.cells[c];
          const unsigned cd = (r.w[7] >> 16) & kPolyCountMask;
          if (cd > 8u) continue;
          const double cx0 = lv.x0 + ix * hcell - grow, cx1 = lv.x0 + (ix + 1) * hcell + grow;
          const double cy0 = lv.y0 + iy * hcell - grow - 2.0 * h.tol, cy1 = lv.y0 + (iy + 1) * hcell + grow + 2.0 * h.tol;
          const double M = 1e-6 * scale;
          unsigned basebit = 0;
          plist.clear();
          for (int i = 0; i < n; ++i) {
            const Seg &e = seg[i];
            if (!(e.vv > 0.0)) continue;   // zero-length: never a crossing
            const double ylo = std::min(e.sy, e.ey), yhi = std::max(e.sy, e.ey);
            if (yhi < cy0 || ylo > cy1) continue;
            if (std::max(e.sx, e.ex) < cx0 - 2.0 * h.tol) continue;
            if (std::min(e.sx, e.ex) > cx1 + M && ylo < cy0 - M && yhi > cy1 + M) { basebit ^= 1u; continue; }
            plist.push_back((unsigned short)i);
          }
          if (plist.size() > 7) continue;
          for (size_t k = 0; k < plist.size(); ++k) {
            const unsigned hw = 8u + (unsigned)k;   // halfword of the record
            r.w[hw >> 1] = (hw & 1u) ? ((r.w[hw >> 1] & 0x0000ffffu) | ((unsigned)plist[k] << 16))
                                     : ((r.w[hw >> 1] & 0xffff0000u) | (unsigned)plist[k]);
          }
          r.w[7] = (r.w[7] & 0x0000ffffu) | ((cd | ((unsigned)plist.size() << 4) | (basebit << 8) | (3u << 14)) << 16);
          out.cells_packed++;
          continue;
        }
        const double x = lv.x0 + (ix + 0.5) * hcell, y = lv.y0 + (iy + 0.5) * hcell;
        int rs = 0;
        if (y >= ymin && y <= ymax && x <= xmax)
          for (int i : rowedges)
            if (poly_cross_ray(seg[i].sx - x, seg[i].sy - y, seg[i].ex - x, seg[i].ey - y)) rs++;
        out.cells[c].w[7] |= (rs % 2 == 0 ? 1u : 2u) << 30;
        out.cells_known++;
      }
    }
  }
  h.cx = 0.5 * (xmin + xmax); h.cy = 0.5 * (ymin + ymax);
  for (int l = 0; l < 3; ++l) {
    const PolyLevel &lv = h.lv[l];
    if (lv.nx != kPolyGrid || lv.ny != kPolyGrid) return false;
    h.linv[l] = lv.inv_h;
    h.lbase[l] = lv.base;
  }
  h.lr[0] = 0.5 * (L + 2.0 * margins[0]) * (1.0 - 1e-9);
  h.lr[1] = 0.5 * (L + 2.0 * margins[1]) * (1.0 - 1e-9);
  h.lr[2] = 0.5 * (L + 2.0 * 40.0 * L) * (1.0 - 1e-9);
  h.pad2 = 0;
  return true;
}
}  // namespace svsdf
