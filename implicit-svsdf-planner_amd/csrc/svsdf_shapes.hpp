// svsdf_shapes.hpp -- gfx950 device code: the 16 analytic 2-D robot-shape SDFs + Polygon.
//
// Behavioural spec: reference src/utils/include/utils/Shape.hpp (SHP) getonlySDF(pos_rel)
// one-argument overloads and the DEFINE_USEFUL_FUNCTION finite-difference gradient macro
// (SHP:34-53).  Operation order follows the cited lines so that, compiled with
// -ffp-contract=off, each formula rounds like the reference's x86-64 build (no FMA).
// Written from scratch for HIP: shape id is a template parameter so every kernel is
// specialised per shape (no divergent dispatch in the inner loop).
#pragma once
#include <hip/hip_runtime.h>

#include "svsdf_polygon.hpp"

namespace svsdf {

constexpr double kPI = 3.14159265358979323846;  // SHP:31

enum ShapeId : int {
  kUnevenCapsule = 0, kCutDisk, kTrapezoid, kRhombus, kStar, kTunnel, kHorseshoe, kHeart,
  kOrientedVesica, kRoundedCross, kRoundedX, kBigX, kMoon, kPie, kPie2, kArc, kPolygon,
  kShapeCount,
  // Kernel-template value only (never a svsdf_shape_id): Polygon whose edge array sits at the START of the block's dynamic
  // LDS (k_solve / k_round of outlines of <= kPolyLdsMaxVerts vertices).  A compile-time LDS address lets the compiler
  // use ds_read for the per-edge loads; a run-time choice between an LDS and a global pointer would make them flat loads.
  kPolygonLds = kShapeCount,
  kKernelShapeCount
};
constexpr int kPolyLdsMaxVerts = 1024;   // 48 KB of LDS

// every kernel's `extern __shared__` array aliases the start of the block's dynamic LDS
extern __shared__ double svsdf_dyn_lds[];
template <int SHAPE>
constexpr bool is_polygon() { return SHAPE == kPolygon || SHAPE == kPolygonLds; }
// doubles at the start of the dynamic LDS that hold the Polygon's edges (kPolygonLds kernels), else 0
template <int SHAPE>
__device__ __forceinline__ size_t poly_lds_doubles(int nverts) { return SHAPE == kPolygonLds ? (size_t)kPolyEdgeDoubles * (size_t)nverts : 0; }

// Host-prepared constants.  The reference evaluates cos/sin member initialisers with the host
// libm at construction (SHP:855, 1237, 1278, 1320); we do the same on the host and pass them in.
struct ShapeParams {
  double tx, ty;              // trans                       SHP:286
  double r00, r01, r10, r11;  // Rotate                      SHP:287-292
  double c0x, c0y;            // shape-specific (cos,sin) pair: horseshoe c / pie c / arc sc
  double r_bound;             // conservative circumradius about the body-frame origin (culling)
  int identity;               // trans == 0 and Rotate == I (poly_params = 0): the transform is exact identity
  int nverts;                 // Polygon: outline vertices
  const PolyAccel *accel;     // Polygon: device pointer to the outline's candidate lists (svsdf_polygon.hpp)
  const PolyEdge *edges;      // Polygon: the outline's edges in global memory
};

// std::max / std::min.  Strict builds keep the compare+select form; the default uses v_max_f64 /
// v_min_f64, which return the same value for every non-NaN input (only the sign of a zero result
// can differ when the operands are zeros of opposite sign, which no formula here distinguishes).
#ifdef SVSDF_SELECT_MINMAX
__device__ __forceinline__ double dmax(double a, double b) { return (a < b) ? b : a; }
__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; }
#else
__device__ __forceinline__ double dmax(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ double dmin(double a, double b) { return __builtin_fmin(a, b); }
#endif
__device__ __forceinline__ double clipd(double v, double lo, double hi) { return dmax(dmin(v, hi), lo); }

// a / b for a shape CONSTANT b (round 5): the quotient the division instruction sequence rounds to, from rb = RN(1 / b) (folded
// at compile time) by poly_quot's two residual steps -- five full-rate operations instead of the division's thirteen with
// a quarter-rate v_rcp_f64 (svsdf_polygon.hpp: Markstein's theorem; the same function the Polygon edges use, checked against
// the division on adversarial operands by tests/test_polygon_accel.py and, for these constants, tests/test_const_div.py).
// |a| outside [1e-150, 1e150] (a zero, a NaN, an infinity) takes the division itself, wave-uniformly.
#ifndef SVSDF_CONST_DIV
#define SVSDF_CONST_DIV 1
#endif
__device__ __forceinline__ double div_const(double a, double b, double rb) {
#if SVSDF_CONST_DIV
  bool inr;
  double q = poly_quot(a, b, rb, inr);
  if (SVSDF_WAVE_ANY(!inr)) q = inr ? q : a / b;
  return q;
#else
  (void)rb;
  return a / b;
#endif
}
__device__ __forceinline__ double norm2(double x, double y) { return sqrt(x * x + y * y); }

// SHP:531-543
__device__ __forceinline__ double sdf_uneven_capsule(double px, double py) {
  const double r1 = 2.0, r2 = 1.0, h = 5.0;
  px = fabs(px);
  const double b = (r1 - r2) / h;
  const double a = sqrt(1.0 - b * b);
  const double k = px * (-b) + py * a;
  if (k < 0.0) return norm2(px, py) - r1;
  if (k > a * h) return norm2(px - 0.0, py - h) - r2;
  return (px * a + py * b) - r1;
}

// SHP:698-711
__device__ __forceinline__ double sdf_cut_disk(double px, double py) {
  const double r = 5.0, h = 2.0;
  const double w = sqrt(r * r - h * h);
  px = fabs(px);
  const double s = dmax((h - r) * px * px + w * w * (h + r - 2.0 * py), h * px - w * py);
  return (s < 0.0) ? norm2(px, py) - r : (px < w) ? h - py : norm2(px - w, py - h);
}

// SHP:754-767
__device__ __forceinline__ double sdf_trapezoid(double px, double py) {
  constexpr double r1 = 1.0, r2 = 3.0, he = 2.0;
  constexpr double k1x = r2, k1y = he, k2x = r2 - r1, k2y = 2.0 * he;
  px = fabs(px);
  const double cax = dmax(0.0, px - ((py < 0.0) ? r1 : r2));
  const double cay = fabs(py) - he;
  constexpr double den = k2x * k2x + k2y * k2y;
  const double c = clipd(div_const((k1x - px) * k2x + (k1y - py) * k2y, den, 1.0 / den), 0.0, 1.0);
  const double cbx = (px - k1x) + k2x * c;
  const double cby = (py - k1y) + k2y * c;
  const double s = (cbx < 0.0 && cay < 0.0) ? -1.0 : 1.0;
  return s * sqrt(dmin(cax * cax + cay * cay, cbx * cbx + cby * cby));
}

// SHP:809-826
__device__ __forceinline__ double sdf_rhombus(double px, double py) {
  constexpr double bx = 1.0, by = 4.5;
  px = fabs(px);
  py = fabs(py);
  const double mbx = bx - 2.0 * px, mby = by - 2.0 * py;
  constexpr double dp = bx * bx + by * by;
  const double h = clipd(div_const(mbx * bx - mby * by, dp, 1.0 / dp), -1.0, 1.0);
  const double bhx = 0.5 * bx, bhy = 0.5 * by;
  const double vhx = 1.0 - h, vhy = 1.0 + h;
  const double d = norm2(px - bhx * vhx, py - bhy * vhy);
  const double e = px * by + py * bx - bx * by;
  const double sign_term = (__double2hiint(e) < 0) ? -1.0 : 1.0;  // std::signbit
  return d * sign_term;
}

// SHP:584-601
__device__ __forceinline__ double sdf_star(double px, double py) {
  constexpr double r = 2.8, rf = 0.6;
  constexpr double k1x = 0.809016994375, k1y = -0.587785252292;
  const double k2x = -k1x, k2y = k1y;
  px = fabs(px);
  double s = 2.0 * dmax(k1x * px + k1y * py, 0.0);
  px -= s * k1x;
  py -= s * k1y;
  s = 2.0 * dmax(k2x * px + k2y * py, 0.0);
  px -= s * k2x;
  py -= s * k2y;
  px = fabs(px);
  py -= r;
  constexpr double bax = rf * (-k1y) - 0.0;
  constexpr double bay = rf * k1x - 1.0;
  constexpr double bb = bax * bax + bay * bay;
  const double h = clipd(div_const(px * bax + py * bay, bb, 1.0 / bb), 0.0, r);
  const double dx = px - bax * h, dy = py - bay * h;
  return norm2(dx, dy) * copysign(1.0, py * bax - px * bay);
}

// SHP:642-658
__device__ __forceinline__ double sdf_tunnel(double px, double py) {
  const double whx = 2.5, why = 1.5;
  px = fabs(px);
  py = -py;
  double qx = px - whx;
  const double qy = py - why;
  const double m = dmax(qx, 0.0);
  const double d1 = m * m + qy * qy;
  qx = (py > 0.0) ? qx : sqrt(px * px + py * py) - whx;
  const double n = dmax(qy, 0.0);
  const double d2 = qx * qx + n * n;
  const double d = sqrt(dmin(d1, d2));
  return (dmax(qx, qy) < 0.0) ? -d : d;
}

// SHP:870-891
__device__ __forceinline__ double sdf_horseshoe(double px, double py, double cx, double cy) {
  const double r = 1.5, wx = 1.55, wy = 0.20;
  px = fabs(px);
  const double l = norm2(px, py);
  double nx = -cx * px + cy * py;
  double ny = cy * px + cx * py;
  const double pxr = nx;
  if (pxr <= 0 && ny <= 0) nx = l * copysign(1.0, -cx);
  if (pxr <= 0) ny = l;
  nx = nx - wx;
  ny = fabs(ny - r) - wy;
  return norm2(dmax(nx, 0.0), dmax(ny, 0.0)) + dmin(0.0, dmax(nx, ny));
}

// SHP:939-952
__device__ __forceinline__ double sdf_heart(double px, double py) {
  px = px / 4.0;
  py = py / 4.0;
  px = fabs(px);
  if (py + px > 1.0) {
    const double ax = px - 0.25, ay = py - 0.75;
    return 4 * (sqrt(ax * ax + ay * ay) - sqrt(2.0) / 4.0);
  }
  const double bx = px - 0.0, by = py - 1.0;
  const double value1 = bx * bx + by * by;
  const double temp = dmax(px + py, 0.0);
  const double cx = px - 0.5 * temp, cy = py - 0.5 * temp;
  const double value2 = cx * cx + cy * cy;
  return 4 * (sqrt(dmin(value1, value2)) * copysign(1.0, px - py));
}

// SHP:988-994 (w = 3) and SHP:1024-1030 (bigX, w = 5)
__device__ __forceinline__ double sdf_rounded_x(double px, double py, double w) {
  const double r = 0.25;
  const double ax = fabs(px), ay = fabs(py);
  const double m = (ax + ay > w) ? (w * 0.5) : (ax + ay) * 0.5;
  return norm2(ax - m, ay - m) - r;
}

// SHP:1062-1075
__device__ __forceinline__ double sdf_rounded_cross(double px, double py) {
  const double h = 1.0;
  px = px / 2.0;
  py = py / 2.0;
  const double k = 0.5 * (h + 1.0 / h);
  const double ax = fabs(px), ay = fabs(py);
  if (ax < 1.0 && ay < ax * (k - h) + h) {
    const double ux = ax - 1, uy = ay - k;
    return 2 * (k - sqrt(ux * ux + uy * uy));
  } else {
    const double ux = ax - 0, uy = ay - h;
    const double vx = ax - 1, vy = ay - 0;
    return 2 * sqrt(dmin(ux * ux + uy * uy, vx * vx + vy * vy));
  }
}

// SHP:1115-1146
__device__ __forceinline__ double sdf_oriented_vesica(double px, double py) {
  const double ax = 2, ay = 4, bx = -2, by = -4, w = 0.8;
  const double r = 0.5 * norm2(bx - ax, by - ay);
  const double d = 0.5 * (r * r - w * w) / w;
  const double vx = (bx - ax) / r, vy = (by - ay) / r;
  const double cx = 0.5 * (bx + ax), cy = 0.5 * (by + ay);
  const double ux = px - cx, uy = py - cy;
  const double qx = 0.5 * fabs(vy * ux + vx * uy);
  const double qy = 0.5 * fabs((-vx) * ux + vy * uy);
  double hx, hy, hz;
  if (r * qx < d * (qy - r)) { hx = 0.0; hy = r; hz = 0.0; }
  else { hx = -d; hy = 0.0; hz = d + w; }
  return 1.0 * (norm2(qx - hx, qy - hy) - hz);
}

// SHP:1202-1214
__device__ __forceinline__ double sdf_moon(double qx, double qy) {
  const double d = 0.8, ra = 3.0, rb = 2.4;
  qy = fabs(qy);
  const double a = (ra * ra - rb * rb + d * d) / (2.0 * d);
  const double b = sqrt(dmax(ra * ra - a * a, 0.0));
  const bool condition = d * (qx * b - qy * a) > d * d * dmax(b - qy, 0.0);
  const double dist1 = norm2(qx - a, qy - b);
  const double dist2 = dmax(norm2(qx, qy) - ra, -norm2(qx - d, qy - 0.0) + rb);
  return condition ? dist1 : dist2;
}

// SHP:1253-1260 (sdPie) and SHP:1294-1301 (sdPie2)
__device__ __forceinline__ double sdf_pie(double px, double py, double cx, double cy) {
  const double r = 3.0;
  px = fabs(px);
  const double l = norm2(px, py) - r;
  const double k = clipd(px * cx + py * cy, 0.0, r);
  const double m = norm2(px - cx * k, py - cy * k);
  return dmax(l, m * copysign(1.0, cy * px - cx * py));
}

// SHP:1334-1343
__device__ __forceinline__ double sdf_arc(double px, double py, double scx, double scy) {
  const double ra = 2.3333, rb = 0.5;
  px = fabs(px);
  const bool condition = scy * px > scx * py;
  const double dist1 = norm2(px - scx * ra, py - scy * ra);
  const double dist2 = fabs(norm2(px, py) - ra);
  return (condition ? dist1 : dist2) - rb;
}

// Polygon (SHP:1352-1531; no trans/Rotate, as the reference): svsdf_polygon.hpp -- the reference's loop over all edges,
// restricted to the edges that can matter for the query's cell / slab (same bits).
// The shape formula proper, on the shape-local point (after trans / Rotate).
template <int SHAPE>
__device__ __forceinline__ double shape_core(const ShapeParams &sp, double px, double py) {
  if constexpr (SHAPE == kUnevenCapsule) return sdf_uneven_capsule(px, py);
  else if constexpr (SHAPE == kCutDisk) return sdf_cut_disk(px, py);
  else if constexpr (SHAPE == kTrapezoid) return sdf_trapezoid(px, py);
  else if constexpr (SHAPE == kRhombus) return sdf_rhombus(px, py);
  else if constexpr (SHAPE == kStar) return sdf_star(px, py);
  else if constexpr (SHAPE == kTunnel) return sdf_tunnel(px, py);
  else if constexpr (SHAPE == kHorseshoe) return sdf_horseshoe(px, py, sp.c0x, sp.c0y);
  else if constexpr (SHAPE == kHeart) return sdf_heart(px, py);
  else if constexpr (SHAPE == kOrientedVesica) return sdf_oriented_vesica(px, py);
  else if constexpr (SHAPE == kRoundedCross) return sdf_rounded_cross(px, py);
  else if constexpr (SHAPE == kRoundedX) return sdf_rounded_x(px, py, 3.0);
  else if constexpr (SHAPE == kBigX) return sdf_rounded_x(px, py, 5.0);
  else if constexpr (SHAPE == kMoon) return sdf_moon(px, py);
  else if constexpr (SHAPE == kPie || SHAPE == kPie2) return sdf_pie(px, py, sp.c0x, sp.c0y);
  else return sdf_arc(px, py, sp.c0x, sp.c0y);
}

// getonlySDF(pos_rel): ((pos_rel - trans) * Rotate).head(2) then the shape formula.
template <int SHAPE>
__device__ __forceinline__ double shape_sdf(const ShapeParams &sp, double x, double y) {
  if constexpr (SHAPE == kPolygonLds) {
    return poly_sdf<false>(*sp.accel, reinterpret_cast<const PolyEdge *>(svsdf_dyn_lds), x, y, nullptr, nullptr);
  } else if constexpr (SHAPE == kPolygon) {
    return poly_sdf<false>(*sp.accel, sp.edges, x, y, nullptr, nullptr);
  } else {
    double px = x, py = y;
    if (!sp.identity) {  // wave-uniform; with trans = 0, Rotate = I the products below are exact no-ops
      const double dx = x - sp.tx, dy = y - sp.ty;
      px = dx * sp.r00 + dy * sp.r10;
      py = dx * sp.r01 + dy * sp.r11;
    }
    return shape_core<SHAPE>(sp, px, py);
  }
}

// getonlySDF(pos_rel, R_obj) (2-argument overloads, e.g. SHP:545-559): ((pos_rel - trans) * Rotate *
// R_obj).head(2) with R_obj = AngleAxisd(yaw, Z), i.e. the row vector times [[c,-s],[s,c]].  Not defined
// for Polygon in the reference (SHP:1477 does not override the Matrix3d virtual).
template <int SHAPE>
__device__ __forceinline__ double shape_sdf_rot(const ShapeParams &sp, double x, double y, double c, double s) {
  static_assert(!is_polygon<SHAPE>(), "Polygon has no (pos_rel, R_obj) overload");
  const double dx = x - sp.tx, dy = y - sp.ty;
  const double qx = dx * sp.r00 + dy * sp.r10;
  const double qy = dx * sp.r01 + dy * sp.r11;
  const double px = qx * c + qy * s;
  const double py = qx * (-s) + qy * c;
  return shape_core<SHAPE>(sp, px, py);
}

// getonlyGrad1: central FD, dx = 1e-6 (SHP:35-53); Polygon: analytic (SHP:1505-1531).
template <int SHAPE>
__device__ __forceinline__ void shape_grad(const ShapeParams &sp, double x, double y, double &gx,
                                           double &gy) {
  if constexpr (is_polygon<SHAPE>()) {
    double cx, cy;
    const double sd = poly_sdf<true>(*sp.accel, sp.edges, x, y, &cx, &cy);
    double vx = x - cx, vy = y - cy;
    const double z = vx * vx + vy * vy;
    if (z > 0.0) { const double n = sqrt(z); vx = vx / n; vy = vy / n; }
    if (__double2hiint(sd) < 0) { vx = -vx; vy = -vy; }  // odd crossing count <=> -dis_min
    gx = vx;
    gy = vy;
  } else {
    const double dx = 0.000001;
    double t0 = x, t1 = y;
    t0 -= dx;
    double sdfold = shape_sdf<SHAPE>(sp, t0, t1);
    t0 += 2 * dx;
    const double gradx = shape_sdf<SHAPE>(sp, t0, t1) - sdfold;
    t0 = x;
    t1 -= dx;
    sdfold = shape_sdf<SHAPE>(sp, t0, t1);
    t1 += 2 * dx;
    const double grady = shape_sdf<SHAPE>(sp, t0, t1) - sdfold;
    gx = gradx / (2 * dx);
    gy = grady / (2 * dx);
  }
}

}  // namespace svsdf
