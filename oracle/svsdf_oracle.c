/*
 * svsdf_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See svsdf_oracle.h for scope, citation abbreviations and the "parity unpinned" note.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).  The
 * reference is built -O3 without -march / -ffast-math (src/planner_algorithm/
 * CMakeLists.txt:4), so x86-64 GCC emits no FMA; -ffp-contract=off keeps that here.
 */
#define _GNU_SOURCE   /* sincos() */
#include "svsdf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.14159265358979323846 /* SHP:31 */

struct orc_ctx {
  orc_shape shape;
  orc_traj traj;
  double safety_hor, weight_p, rho;
  double head[9], tail[9]; /* 3x3 column-major: col0 = pos, col1 = vel, col2 = acc */
  orc_counters cnt;
  int have_duration;
  int trig_mode; /* 0: libm sin/cos/atan2 (the reference's arithmetic); 1: the ROCm device library's algorithms;
                    2: libm results moved by -1 / 0 / +1 ulp, pseudo-randomly per argument (orc_set_trig_perturb) */
  unsigned long long trig_seed;
};

/* thread-local work counters, folded into ctx->cnt by the entry points */
static _Thread_local orc_counters tl_cnt;

/* optional trace of the descents (studies only, tools/experiments/wave_model.py): per gradientDescent call one 32-byte
 * record -- [0] passes (<= 31 recorded), [k] the accepted ladder index of pass k (1 .. 29; 0: none accepted) */
static unsigned char *g_gd_trace = NULL;
static size_t g_gd_trace_cap = 0, g_gd_trace_n = 0;
void orc_set_gd_trace(unsigned char *buf, size_t cap_records) { g_gd_trace = buf; g_gd_trace_cap = cap_records; g_gd_trace_n = 0; }
size_t orc_gd_trace_count(void) { return g_gd_trace_n; }

static inline double dmax(double a, double b) { return (a < b) ? b : a; } /* std::max */
static inline double dmin(double a, double b) { return (b < a) ? b : a; } /* std::min */
static inline double clipd(double v, double lo, double hi) { return dmax(dmin(v, hi), lo); }
static inline double norm2(double x, double y) { return sqrt(x * x + y * y); }

/* ------------------------------------------------------------------------- */
/* shapes                                                                     */
/* ------------------------------------------------------------------------- */
static const char *k_shape_names[ORC_SHAPE_COUNT] = {
    "sdUnevenCapsule", "sdCutDisk", "sdTrapezoid", "sdRhombus", "star", "sdTunnel",
    "sdHorseshoe", "sdHeart", "sdOrientedVesica", "sdRoundedCross", "sdRoundedX", "bigX",
    "sdMoon", "sdPie", "sdPie2", "sdArc", "Polygon"};

int orc_shape_id_from_name(const char *name) {
  for (int i = 0; i < ORC_SHAPE_Polygon; ++i) /* Polygon is not in the registry (SWM:187-235) */
    if (strcmp(name, k_shape_names[i]) == 0) return i;
  return -1;
}
const char *orc_shape_name(int id) {
  return (id >= 0 && id < ORC_SHAPE_COUNT) ? k_shape_names[id] : "?";
}

void orc_shape_init(orc_shape *s, int id, const double poly_params[3], const double *poly_xy,
                    int nverts) {
  memset(s, 0, sizeof(*s));
  s->id = id;
  /* SHP:286-292 */
  s->tx = poly_params ? poly_params[0] : 0.0;
  s->ty = poly_params ? poly_params[1] : 0.0;
  double yaw = (poly_params ? poly_params[2] : 0.0) * ORC_PI / 180.0;
  /* std::cos(yaw) / std::sin(yaw) in the reference (SHP:289-292); its g++ -O3 build merges the pair into one sincos()
   * call (cse_sincos), whose results differ from cos() / sin() in the last bit for some arguments: explicit here, so
   * that this file does not depend on what the compiler merges */
  {
    double sn_, cs_;
    sincos(yaw, &sn_, &cs_);
    s->r00 = cs_; s->r01 = -sn_; s->r10 = sn_; s->r11 = cs_;
  }
  s->hs_cx = cos(20.5);  s->hs_cy = sin(20.5);      /* SHP:855 (radians) */
  s->pie_cx = cos(43.0); s->pie_cy = sin(43.0);     /* SHP:1237 */
  s->pie2_cx = cos(1.0); s->pie2_cy = sin(1.0);     /* SHP:1278 */
  s->arc_scx = sin(20.0); s->arc_scy = cos(20.0);   /* SHP:1320 */
  if (id == ORC_SHAPE_Polygon) {
    if (poly_xy && nverts >= 3) {
      if (nverts > ORC_MAX_POLY_VERTS) nverts = ORC_MAX_POLY_VERTS;
      s->nverts = nverts;
      for (int i = 0; i < nverts; ++i) { s->vx[i] = poly_xy[2 * i]; s->vy[i] = poly_xy[2 * i + 1]; }
    } else {
      /* fallback rectangle SWM:363-369 */
      static const double rect[8] = {6, -0.1, 6, 0.1, -6, 0.1, -6, -0.1};
      s->nverts = 4;
      for (int i = 0; i < 4; ++i) { s->vx[i] = rect[2 * i]; s->vy[i] = rect[2 * i + 1]; }
    }
    for (int i = 0; i < s->nverts; ++i) s->next[i] = (i + 1) % s->nverts;   /* one closed chain, SHP:1452-1454 */
  }
}

int orc_shape_set_loops(orc_shape *s, const int *loop_sizes, int nloops) {
  if (!s || s->id != ORC_SHAPE_Polygon || !loop_sizes || nloops < 1) return -1;
  int tot = 0;
  for (int k = 0; k < nloops; ++k) { if (loop_sizes[k] < 3) return -1; tot += loop_sizes[k]; }
  if (tot != s->nverts) return -1;
  int b = 0;
  for (int k = 0; k < nloops; ++k) {
    for (int i = 0; i < loop_sizes[k]; ++i) s->next[b + i] = b + (i + 1) % loop_sizes[k];
    b += loop_sizes[k];
  }
  return 0;
}

/* ((pos_rel - trans) * Rotate).head(2): row vector times matrix (e.g. SHP:586).  The z
 * term is (z - 0) * Rotate(2, j) = finite * 0 and only adds a signed zero. */
static inline void shape_local(const orc_shape *s, double x, double y, double *px, double *py) {
  double dx = x - s->tx, dy = y - s->ty;
  *px = dx * s->r00 + dy * s->r10;
  *py = dx * s->r01 + dy * s->r11;
}

/* SHP:531-543 */
static double sdf_uneven_capsule(double px, double py) {
  const double r1 = 2.0, r2 = 1.0, h = 5.0;
  px = fabs(px);
  double b = (r1 - r2) / h;
  double a = sqrt(1.0 - b * b);
  double k = px * (-b) + py * a;
  if (k < 0.0) return norm2(px, py) - r1;
  if (k > a * h) return norm2(px - 0.0, py - h) - r2;
  return (px * a + py * b) - r1;
}

/* SHP:698-711 */
static double sdf_cut_disk(double px, double py) {
  const double r = 5.0, h = 2.0;
  const double w = sqrt(r * r - h * h);
  px = fabs(px);
  double s = dmax((h - r) * px * px + w * w * (h + r - 2.0 * py), h * px - w * py);
  return (s < 0.0) ? norm2(px, py) - r : (px < w) ? h - py : norm2(px - w, py - h);
}

/* SHP:754-767 */
static double sdf_trapezoid(double px, double py) {
  const double r1 = 1.0, r2 = 3.0, he = 2.0;
  const double k1x = r2, k1y = he;
  const double k2x = r2 - r1, k2y = 2.0 * he;
  px = fabs(px);
  double cax = dmax(0.0, px - ((py < 0.0) ? r1 : r2));
  double cay = fabs(py) - he;
  double c = clipd(((k1x - px) * k2x + (k1y - py) * k2y) / (k2x * k2x + k2y * k2y), 0.0, 1.0);
  double cbx = (px - k1x) + k2x * c;
  double cby = (py - k1y) + k2y * c;
  double s = (cbx < 0.0 && cay < 0.0) ? -1.0 : 1.0;
  return s * sqrt(dmin(cax * cax + cay * cay, cbx * cbx + cby * cby));
}

/* SHP:809-826 */
static double sdf_rhombus(double px, double py) {
  const double bx = 1.0, by = 4.5;
  px = fabs(px);
  py = fabs(py);
  double mbx = bx - 2.0 * px, mby = by - 2.0 * py;
  double dp = bx * bx + by * by;
  double h = clipd((mbx * bx - mby * by) / dp, -1.0, 1.0);
  double bhx = 0.5 * bx, bhy = 0.5 * by;
  double vhx = 1.0 - h, vhy = 1.0 + h;
  double d = norm2(px - bhx * vhx, py - bhy * vhy);
  double sign_term = signbit(px * by + py * bx - bx * by) ? -1.0 : 1.0;
  return d * sign_term;
}

/* SHP:584-601 */
static double sdf_star(double px, double py) {
  const double r = 2.8, rf = 0.6;
  const double k1x = 0.809016994375, k1y = -0.587785252292;
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
  double bax = rf * (-k1y) - 0.0;
  double bay = rf * k1x - 1.0;
  double h = clipd((px * bax + py * bay) / (bax * bax + bay * bay), 0.0, r);
  double dx = px - bax * h, dy = py - bay * h;
  return norm2(dx, dy) * copysign(1.0, py * bax - px * bay);
}

/* SHP:642-658 */
static double sdf_tunnel(double px, double py) {
  const double whx = 2.5, why = 1.5;
  px = fabs(px);
  py = -py;
  double qx = px - whx, qy = py - why;
  double m = dmax(qx, 0.0);
  double d1 = m * m + qy * qy;             /* std::pow(x, 2) == x*x */
  qx = (py > 0.0) ? qx : sqrt(px * px + py * py) - whx;
  double n = dmax(qy, 0.0);
  double d2 = qx * qx + n * n;
  double d = sqrt(dmin(d1, d2));
  return (dmax(qx, qy) < 0.0) ? -d : d;
}

/* SHP:870-891 */
static double sdf_horseshoe(const orc_shape *s, double px, double py) {
  const double r = 1.5, wx = 1.55, wy = 0.20;
  const double cx = s->hs_cx, cy = s->hs_cy;
  px = fabs(px);
  double l = norm2(px, py);
  double nx = -cx * px + cy * py;
  double ny = cy * px + cx * py;
  double pxr = nx;
  if (pxr <= 0 && ny <= 0) nx = l * copysign((double)1.0f, -cx); /* std::copysign(float, double) promotes to double */
  if (pxr <= 0) ny = l;
  nx = nx - wx;
  ny = fabs(ny - r) - wy;
  return norm2(dmax(nx, 0.0), dmax(ny, 0.0)) + dmin(0.0, dmax(nx, ny));
}

/* SHP:939-952 */
static double sdf_heart(double px, double py) {
  px = px / 4.0;
  py = py / 4.0;
  px = fabs(px);
  if (py + px > 1.0) {
    double ax = px - 0.25, ay = py - 0.75;
    return 4 * (sqrt(ax * ax + ay * ay) - sqrt(2.0) / 4.0);
  }
  double bx = px - 0.0, by = py - 1.0;
  double value1 = bx * bx + by * by;
  double temp = dmax(px + py, 0.0);
  double cx = px - 0.5 * temp, cy = py - 0.5 * temp;
  double value2 = cx * cx + cy * cy;
  return 4 * (sqrt(dmin(value1, value2)) * copysign(1.0, px - py));
}

/* SHP:988-994 (w=3) and SHP:1024-1030 (bigX, w=5) */
static double sdf_rounded_x(double px, double py, double w) {
  const double r = 0.25;
  double ax = fabs(px), ay = fabs(py);
  double m = (ax + ay > w) ? (w * 0.5f) : (ax + ay) * 0.5f;
  return norm2(ax - m, ay - m) - r;
}

/* SHP:1062-1075 */
static double sdf_rounded_cross(double px, double py) {
  const double h = 1.0;
  px = px / 2.0;
  py = py / 2.0;
  double k = 0.5 * (h + 1.0 / h);
  double ax = fabs(px), ay = fabs(py);
  if (ax < 1.0 && ay < ax * (k - h) + h) {
    double ux = ax - 1, uy = ay - k;
    return 2 * (k - sqrt(ux * ux + uy * uy));
  } else {
    double ux = ax - 0, uy = ay - h;
    double vx = ax - 1, vy = ay - 0;
    return 2 * sqrt(dmin(ux * ux + uy * uy, vx * vx + vy * vy));
  }
}

/* SHP:1115-1146 */
static double sdf_oriented_vesica(double px, double py) {
  const double ax = 2, ay = 4, bx = -2, by = -4, w = 0.8;
  px = px / 1.0;
  py = py / 1.0;
  double r = 0.5 * norm2(bx - ax, by - ay);
  double d = 0.5 * (r * r - w * w) / w;
  double vx = (bx - ax) / r, vy = (by - ay) / r;
  double cx = 0.5 * (bx + ax), cy = 0.5 * (by + ay);
  double ux = px - cx, uy = py - cy;
  double qx = 0.5 * fabs(vy * ux + vx * uy);
  double qy = 0.5 * fabs((-vx) * ux + vy * uy);
  double hx, hy, hz;
  if (r * qx < d * (qy - r)) { hx = 0.0; hy = r; hz = 0.0; }
  else { hx = -d; hy = 0.0; hz = d + w; }
  return 1.0 * (norm2(qx - hx, qy - hy) - hz);
}

/* SHP:1202-1214 */
static double sdf_moon(double qx, double qy) {
  const double d = 0.8, ra = 3.0, rb = 2.4;
  qy = fabs(qy);
  double a = (ra * ra - rb * rb + d * d) / (2.0 * d);
  double b = sqrt(dmax(ra * ra - a * a, 0.0));
  int condition = d * (qx * b - qy * a) > d * d * dmax(b - qy, 0.0);
  double dist1 = norm2(qx - a, qy - b);
  double dist2 = dmax(norm2(qx, qy) - ra, -norm2(qx - d, qy - 0.0) + rb);
  return condition ? dist1 : dist2;
}

/* SHP:1253-1260 (sdPie) and SHP:1294-1301 (sdPie2) */
static double sdf_pie(double px, double py, double cx, double cy) {
  const double r = 3.0;
  px = fabs(px);
  double l = norm2(px, py) - r;
  double k = clipd(px * cx + py * cy, 0.0, r);
  double m = norm2(px - cx * k, py - cy * k);
  return dmax(l, m * copysign((double)1.0f, cy * px - cx * py));
}

/* SHP:1334-1343 */
static double sdf_arc(const orc_shape *s, double px, double py) {
  const double ra = 2.3333, rb = 0.5;
  const double scx = s->arc_scx, scy = s->arc_scy;
  px = fabs(px);
  int condition = scy * px > scx * py;
  double dist1 = norm2(px - scx * ra, py - scy * ra);
  double dist2 = fabs(norm2(px, py) - ra);
  return (condition ? dist1 : dist2) - rb;
}

/* Polygon: SHP:1370-1401 (edge helpers), SHP:1448-1476 (getonlySDF).  NB: no trans/Rotate. */
static int poly_cross_ray(double sx, double sy, double ex, double ey, double qx, double qy) {
  double s2x = sx - qx, s2y = sy - qy, e2x = ex - qx, e2y = ey - qy;
  double theta_s = atan2(s2y, s2x);
  double theta_e = atan2(e2y, e2x);
  theta_s = (theta_s < 0.0) ? (theta_s + 2 * ORC_PI) : theta_s;
  theta_e = (theta_e < 0.0) ? (theta_e + 2 * ORC_PI) : theta_e;
  double d1 = fabs(theta_s - theta_e);
  return (d1 < ORC_PI) ? 0 : 1;
}
static double poly_dis2seg(double sx, double sy, double ex, double ey, double px, double py,
                           double *cx, double *cy) {
  double vx = ex - sx, vy = ey - sy;
  double wx = px - sx, wy = py - sy;
  double t = (wx * vx + wy * vy) / (vx * vx + vy * vy);
  if (t < 0.0) t = 0.0;
  else if (t > 1.0) t = 1.0;
  *cx = sx + t * vx;
  *cy = sy + t * vy;
  return norm2(px - *cx, py - *cy);
}
static double sdf_polygon(const orc_shape *s, double x, double y, double *cminx, double *cminy) {
  double dis_min = 1e9, cx = 0, cy = 0, mx = 0, my = 0;
  int rs = 0;
  for (int i = 0; i < s->nverts; ++i) {
    int j = s->next[i];   /* (i + 1) % nverts for the reference's single chain */
    double dis = poly_dis2seg(s->vx[i], s->vy[i], s->vx[j], s->vy[j], x, y, &cx, &cy);
    if (dis < dis_min) { dis_min = dis; mx = cx; my = cy; }
    if (poly_cross_ray(s->vx[i], s->vy[i], s->vx[j], s->vy[j], x, y)) rs++;
  }
  if (cminx) { *cminx = mx; *cminy = my; }
  return (rs % 2 == 0) ? dis_min : -dis_min;
}

double orc_shape_sdf(const orc_shape *s, double x, double y) {
  tl_cnt.shape_evals++;
  if (s->id == ORC_SHAPE_Polygon) return sdf_polygon(s, x, y, NULL, NULL);
  double px, py;
  shape_local(s, x, y, &px, &py);
  switch (s->id) {
    case ORC_SHAPE_sdUnevenCapsule: return sdf_uneven_capsule(px, py);
    case ORC_SHAPE_sdCutDisk: return sdf_cut_disk(px, py);
    case ORC_SHAPE_sdTrapezoid: return sdf_trapezoid(px, py);
    case ORC_SHAPE_sdRhombus: return sdf_rhombus(px, py);
    case ORC_SHAPE_star: return sdf_star(px, py);
    case ORC_SHAPE_sdTunnel: return sdf_tunnel(px, py);
    case ORC_SHAPE_sdHorseshoe: return sdf_horseshoe(s, px, py);
    case ORC_SHAPE_sdHeart: return sdf_heart(px, py);
    case ORC_SHAPE_sdOrientedVesica: return sdf_oriented_vesica(px, py);
    case ORC_SHAPE_sdRoundedCross: return sdf_rounded_cross(px, py);
    case ORC_SHAPE_sdRoundedX: return sdf_rounded_x(px, py, 3.0);
    case ORC_SHAPE_bigX: return sdf_rounded_x(px, py, 5.0);
    case ORC_SHAPE_sdMoon: return sdf_moon(px, py);
    case ORC_SHAPE_sdPie: return sdf_pie(px, py, s->pie_cx, s->pie_cy);
    case ORC_SHAPE_sdPie2: return sdf_pie(px, py, s->pie2_cx, s->pie2_cy);
    case ORC_SHAPE_sdArc: return sdf_arc(s, px, py);
    default: return 1e9;
  }
}

/* getonlySDF(pos_rel, R_obj) (the 2-argument overloads, e.g. SHP:545-559, 603-616): identical bodies
 * to the 1-argument forms after Pos_rel = ((pos_rel - trans) * Rotate * R_obj).head(2), with
 * R_obj = AngleAxisd(yaw, Z) = [[c,-s,0],[s,c,0],[0,0,1]] -> row vector times R_obj.
 * Polygon has no override with this signature (SHP:1477 takes a Matrix2d), so the reference's
 * virtual call lands in the empty base body (SHP:267): not defined, returns NaN here. */
double orc_shape_sdf_rot(const orc_shape *s, double x, double y, double yaw) {
  if (s->id == ORC_SHAPE_Polygon) return NAN;
  double qx, qy;
  shape_local(s, x, y, &qx, &qy);
  double c = cos(yaw), sn = sin(yaw);
  double px = qx * c + qy * sn;
  double py = qx * (-sn) + qy * c;
  tl_cnt.shape_evals++;
  switch (s->id) {
    case ORC_SHAPE_sdUnevenCapsule: return sdf_uneven_capsule(px, py);
    case ORC_SHAPE_sdCutDisk: return sdf_cut_disk(px, py);
    case ORC_SHAPE_sdTrapezoid: return sdf_trapezoid(px, py);
    case ORC_SHAPE_sdRhombus: return sdf_rhombus(px, py);
    case ORC_SHAPE_star: return sdf_star(px, py);
    case ORC_SHAPE_sdTunnel: return sdf_tunnel(px, py);
    case ORC_SHAPE_sdHorseshoe: return sdf_horseshoe(s, px, py);
    case ORC_SHAPE_sdHeart: return sdf_heart(px, py);
    case ORC_SHAPE_sdOrientedVesica: return sdf_oriented_vesica(px, py);
    case ORC_SHAPE_sdRoundedCross: return sdf_rounded_cross(px, py);
    case ORC_SHAPE_sdRoundedX: return sdf_rounded_x(px, py, 3.0);
    case ORC_SHAPE_bigX: return sdf_rounded_x(px, py, 5.0);
    case ORC_SHAPE_sdMoon: return sdf_moon(px, py);
    case ORC_SHAPE_sdPie: return sdf_pie(px, py, s->pie_cx, s->pie_cy);
    case ORC_SHAPE_sdPie2: return sdf_pie(px, py, s->pie2_cx, s->pie2_cy);
    case ORC_SHAPE_sdArc: return sdf_arc(s, px, py);
    default: return 1e9;
  }
}

/* getonlyGrad1: FD macro SHP:35-53 for the analytic shapes, analytic for Polygon SHP:1505-1531 */
void orc_shape_grad(const orc_shape *s, double x, double y, double g[2]) {
  if (s->id == ORC_SHAPE_Polygon) {
    double cx, cy;
    double sd = sdf_polygon(s, x, y, &cx, &cy);
    tl_cnt.shape_evals++;
    double vx = x - cx, vy = y - cy;
    double z = vx * vx + vy * vy; /* .normalized(): divide iff squaredNorm > 0 */
    if (z > 0.0) { double n = sqrt(z); vx = vx / n; vy = vy / n; }
    if (sd < 0.0 || (sd == 0.0 && signbit(sd))) { vx = -vx; vy = -vy; } /* rs odd <=> -dis_min */
    g[0] = vx;
    g[1] = vy;
    return;
  }
  const double dx = 0.000001;
  double t0 = x, t1 = y;
  t0 -= dx;
  double sdfold = orc_shape_sdf(s, t0, t1);
  t0 += 2 * dx;
  double gradx = orc_shape_sdf(s, t0, t1) - sdfold;
  t0 = x;
  t1 -= dx;
  sdfold = orc_shape_sdf(s, t0, t1);
  t1 += 2 * dx;
  double grady = orc_shape_sdf(s, t0, t1) - sdfold;
  g[0] = gradx / (2 * dx);
  g[1] = grady / (2 * dx);
}

/* ------------------------------------------------------------------------- */
/* trajectory                                                                 */
/* ------------------------------------------------------------------------- */
/* TRJ:498-516: mutates t to the local time */
static inline int traj_locate(const orc_traj *tr, double *t) {
  int N = tr->N, idx;
  double dur;
  if (tr->cum_locate) {
    /* diagnostic "device arithmetic" mode (orc_set_trig_mode): the HIP kernels locate the piece on cumulative
     * start times S_i = T_0 + ... + T_{i-1} (summed left to right) and take s = t - S_i, one rounding instead of
     * the i roundings of the loop below.  Same result whenever the partial sums are exact (e.g. the equal 2.5 s
     * pieces of every BASELINE config), otherwise within i ulp(t) of it. */
    double S = 0.0;
    for (idx = 0; idx < N - 1 && *t > S + tr->T[idx]; idx++) S += tr->T[idx];
    *t = *t - S;
    return idx;
  }
  for (idx = 0; idx < N && *t > (dur = tr->T[idx]); idx++) *t -= dur;
  if (idx == N) { idx--; *t += tr->T[idx]; }
  return idx;
}
/* Piece::getPos TRJ:104-114 (Piece stores the s^5 column first, MNC:521-525; here
 * c[k] = coefficient of s^k so the loop runs k = 0..5 with tn = s^k) */
static inline void piece_pos(const double *c, double t, double out[3]) {
  double p0 = 0.0, p1 = 0.0, p2 = 0.0, tn = 1.0;
  for (int k = 0; k <= 5; ++k) {
    p0 += tn * c[k * 3 + 0];
    p1 += tn * c[k * 3 + 1];
    p2 += tn * c[k * 3 + 2];
    tn *= t;
  }
  out[0] = p0; out[1] = p1; out[2] = p2;
}
/* Piece::getVel TRJ:116-128 */
static inline void piece_vel(const double *c, double t, double out[3]) {
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, tn = 1.0;
  int n = 1;
  for (int k = 1; k <= 5; ++k) {
    double w = n * tn;
    v0 += w * c[k * 3 + 0];
    v1 += w * c[k * 3 + 1];
    v2 += w * c[k * 3 + 2];
    tn *= t;
    n++;
  }
  out[0] = v0; out[1] = v1; out[2] = v2;
}
static inline void traj_pos(const orc_traj *tr, double t, double out[3]) {
  int i = traj_locate(tr, &t);
  piece_pos(tr->c + (size_t)i * 18, t, out);
}
static inline void traj_vel(const orc_traj *tr, double t, double out[3]) {
  int i = traj_locate(tr, &t);
  piece_vel(tr->c + (size_t)i * 18, t, out);
}
void orc_traj_pos(const orc_ctx *ctx, double t, double out[3]) { traj_pos(&ctx->traj, t, out); }
void orc_traj_vel(const orc_ctx *ctx, double t, double out[3]) { traj_vel(&ctx->traj, t, out); }
double orc_traj_duration(const orc_ctx *ctx) { return ctx->traj.traj_duration; }

/* ---- optional "device trig" mode (diagnostic) ------------------------------------------------
 * The reference's run-time transcendentals on this path are sin/cos of the yaw and of the GSIP
 * sample angles and one atan2 (SWM:95); every other operation is IEEE (+, -, *, /, sqrt, compare).
 * glibc and the ROCm device library both stay below 1 ulp but do not round identically, which is
 * the ONLY reason the HIP path and this oracle differ in the last bits (and, on flat stretches of
 * SDF(t), in t*).  trig_mode = 1 evaluates these three functions with the device library's published
 * algorithms (ROCm 7.2 ocml: __ocml_sincos_f64 small-argument path = trigredsmall + sincosred2,
 * __ocml_atan2_f64 = atanred on min/max with quadrant fix-up), written in C with exact fma(); the
 * HIP path must then agree bit for bit per point (tests/test_gpu_parity.py).  Mode 0 is the oracle
 * of record. */
static void dev_sincos(double a, double *sn, double *cs) {
  const double ax = fabs(a);
  if (!(ax < 0x1p30)) { *sn = sin(a); *cs = cos(a); return; }
  const double r = rint(ax * 0x1.45f306dc9c883p-1);
  const double t4 = fma(r, -0x1.921fb54442d18p+0, ax);
  const double t5 = fma(r, -0x1.1a62633145c00p-54, t4);
  const double t6 = r * 0x1.1a62633145c00p-54;
  const double t8 = fma(r, 0x1.1a62633145c00p-54, -t6);
  const double t9 = t4 - t6;
  const double t10 = t4 - t9;
  const double t11 = t10 - t6;
  const double t12 = t9 - t5;
  const double t13 = t12 + t11;
  const double t14 = t13 - t8;
  const double t15 = fma(r, -0x1.b839a252049c0p-104, t14);
  const double x = t5 + t15;
  const double y = t15 - (x - t5);
  const int q = (int)r & 3;
  const double s = x * x;
  const double h = s * 0.5;
  const double u5 = 1.0 - h;
  const double u7 = (1.0 - u5) - h;
  const double s2 = s * s;
  double pc = fma(s, -0x1.907db46cc5e42p-37, 0x1.1eeb69037ab78p-29);
  pc = fma(s, pc, -0x1.27e4fa17f65f6p-22);
  pc = fma(s, pc, 0x1.a01a019f4ec90p-16);
  pc = fma(s, pc, -0x1.6c16c16c16967p-10);
  pc = fma(s, pc, 0x1.5555555555555p-5);
  const double ny = -y;
  const double c15 = fma(x, ny, u7);
  const double c16 = fma(s2, pc, c15);
  const double cv = u5 + c16;
  double ps = fma(s, 0x1.5e0b2f9a43bb8p-33, -0x1.ae600b42fdfa7p-26);
  ps = fma(s, ps, 0x1.71de3796cde01p-19);
  ps = fma(s, ps, -0x1.a01a019e83e5cp-13);
  ps = fma(s, ps, 0x1.1111111110bb3p-7);
  const double xs = x * (-s);
  const double s25 = fma(xs, ps, y * 0.5);
  const double s26 = fma(s, s25, ny);
  const double s27 = fma(xs, -0x1.5555555555555p-3, s26);
  const double sv = x - s27;
  double so = (q & 1) ? cv : sv;
  double co = (q & 1) ? -sv : cv;
  if (q > 1) { so = -so; co = -co; }
  if (signbit(a)) so = -so;
  *sn = so; *cs = co;
}
static double dev_atan2(double y, double x) {
  const double ay = fabs(y), ax = fabs(x);
  const double mx = fmax(ax, ay), mn = fmin(ax, ay);
  const double v = mn / mx;
  /* __ocmlpriv_atanred_f64 */
  const double s = v * v;
  double p = fma(s, 0x1.ba404b5e68a13p-17, -0x1.3e260bd3237f4p-13);
  p = fma(s, p, 0x1.b2bb069efb384p-11);
  p = fma(s, p, -0x1.7952daf56de9bp-9);
  p = fma(s, p, 0x1.d6d43a595c56fp-8);
  p = fma(s, p, -0x1.c6ea4a57d9582p-7);
  p = fma(s, p, 0x1.67e295f08b19fp-6);
  p = fma(s, p, -0x1.e9ae6fc27006ap-6);
  p = fma(s, p, 0x1.2c15b5711927ap-5);
  p = fma(s, p, -0x1.59976e82d3ff0p-5);
  p = fma(s, p, 0x1.82d5d6ef28734p-5);
  p = fma(s, p, -0x1.ae5ce6a214619p-5);
  p = fma(s, p, 0x1.e1bb48427b883p-5);
  p = fma(s, p, -0x1.110e48b207f05p-4);
  p = fma(s, p, 0x1.3b13657b87036p-4);
  p = fma(s, p, -0x1.745d119378e4fp-4);
  p = fma(s, p, 0x1.c71c717e1913cp-4);
  p = fma(s, p, -0x1.2492492376b7dp-3);
  p = fma(s, p, 0x1.99999999952ccp-3);
  p = fma(s, p, -0x1.5555555555523p-2);
  const double a0 = fma(v, s * p, v);
  /* quadrant fix-up of __ocml_atan2_f64 */
  const double pio2 = 0x1.921fb54442d18p+0, pi = 0x1.921fb54442d18p+1;
  const int xneg = signbit(x) ? 1 : 0;
  double t = (ax < ay) ? (pio2 - a0) : a0;
  t = xneg ? (pi - t) : t;
  if (y == 0.0) t = xneg ? pi : 0.0;
  if (isinf(ax) && isinf(ay)) t = xneg ? 0x1.2d97c7f3321d2p+1 : 0x1.921fb54442d18p-1;
  if (isnan(x) || isnan(y)) t = NAN;
  return copysign(t, y);
}
/* trig_mode 2: "some other libm".  glibc's sin / cos / atan2 are accurate to < 1 ulp but not correctly rounded; another
 * build of the reference (another libm, another compiler's vector math) may return either neighbour.  The result is
 * moved by k in {-1, 0, +1} ulp, k a hash of (argument bits, seed): deterministic per argument, so the oracle stays a
 * function.  Sensitivity bracket of tools/fuzz_parity.py: how far does the reference's own answer move under a <= 1 ulp
 * change of these three functions? */
static inline double ulp_nudge(double v, unsigned long long key) {
  key ^= key >> 33; key *= 0xff51afd7ed558ccdULL; key ^= key >> 33; key *= 0xc4ceb9fe1a85ec53ULL; key ^= key >> 33;
  const int k = (int)(key % 3ULL) - 1;
  if (k == 0 || !isfinite(v)) return v;
  return nextafter(v, k > 0 ? INFINITY : -INFINITY);
}
static inline unsigned long long dbits(double a) { unsigned long long u; memcpy(&u, &a, 8); return u; }
static inline void trig_sincos(const orc_ctx *ctx, double a, double *sn, double *cs) {
  if (ctx->trig_mode == 1) dev_sincos(a, sn, cs);
  else {
    sincos(a, sn, cs);   /* Eigen's AngleAxis computes sin(angle), cos(angle): one sincos() call in a g++ -O3 build */
    if (ctx->trig_mode == 2) {
      *sn = ulp_nudge(*sn, dbits(a) ^ ctx->trig_seed);
      *cs = ulp_nudge(*cs, dbits(a) ^ (ctx->trig_seed * 0x9e3779b97f4a7c15ULL + 1ULL));
    }
  }
}
static inline double trig_atan2(const orc_ctx *ctx, double y, double x) {
  if (ctx->trig_mode == 1) return dev_atan2(y, x);
  const double v = atan2(y, x);
  return ctx->trig_mode == 2 ? ulp_nudge(v, dbits(y) ^ (dbits(x) * 0x9e3779b97f4a7c15ULL) ^ ctx->trig_seed) : v;
}
void orc_set_trig_perturb(orc_ctx *ctx, unsigned long long seed) {
  /* libm trig moved by <= 1 ulp (seeded), the reference's piece location: a third oracle for the sensitivity bracket */
  ctx->trig_mode = 2;
  ctx->trig_seed = seed * 0x9e3779b97f4a7c15ULL + 0x632be59bd9b4e019ULL;
  ctx->traj.cum_locate = 0;
}
void orc_set_modes(orc_ctx *ctx, int trig_mode, int cum_locate) {
  /* the two diagnostic switches separately: device-library trig (1) and cumulative piece location (1) */
  ctx->trig_mode = trig_mode ? 1 : 0;
  ctx->traj.cum_locate = cum_locate ? 1 : 0;
}
void orc_set_trig_mode(orc_ctx *ctx, int mode) {
  ctx->trig_mode = mode ? 1 : 0;
  ctx->traj.cum_locate = ctx->trig_mode;
}

/* ------------------------------------------------------------------------- */
/* SDF at a time stamp, argmin over t                                         */
/* ------------------------------------------------------------------------- */
/* getSDFAtTimeStamp<false> SWM:741-750 = getStateOnTrajStamp SWM:465-474 + posEva2Rel SWM:521-526.
 * Rt = AngleAxisd(yaw, Z) = [[c,-s,0],[s,c,0],[0,0,1]]; p_rel = Rt^T (p - xt). */
static inline double sdf_at_time(const orc_ctx *ctx, double px, double py, double t) {
  double xt[3];
  traj_pos(&ctx->traj, t, xt);
  double yaw = xt[2];
  double s, c;
  trig_sincos(ctx, yaw, &s, &c);
  double dx = px - xt[0], dy = py - xt[1];
  double rx = c * dx + s * dy;
  double ry = (-s) * dx + c * dy;
  tl_cnt.sdf_evals++;
  return orc_shape_sdf(&ctx->shape, rx, ry);
}
double orc_sdf_at_time(orc_ctx *ctx, double px, double py, double t) {
  return sdf_at_time(ctx, px, py, t);
}

/* Batch of raw body-frame shape evaluations: getonlySDF(pos_rel) and getonlyGrad1(pos_rel) of the context's shape
 * (no trajectory involved); sdf_out / grad_out (2 per point) may be NULL.  Test convenience only. */
void orc_shape_eval_batch(orc_ctx *ctx, const double *xy, size_t P, double *sdf_out, double *grad_out) {
  for (size_t i = 0; i < P; ++i) {
    if (sdf_out) sdf_out[i] = orc_shape_sdf(&ctx->shape, xy[2 * i], xy[2 * i + 1]);
    if (grad_out) orc_shape_grad(&ctx->shape, xy[2 * i], xy[2 * i + 1], grad_out + 2 * i);
  }
}

/* getGradPrelAtTimeStamp<false> SWM:779-788 */
static inline void grad_prel_at_time(const orc_ctx *ctx, double px, double py, double t,
                                     double g[2]) {
  double xt[3];
  traj_pos(&ctx->traj, t, xt);
  double yaw = xt[2];
  double s, c;
  trig_sincos(ctx, yaw, &s, &c);
  double dx = px - xt[0], dy = py - xt[1];
  double rx = c * dx + s * dy;
  double ry = (-s) * dx + c * dy;
  orc_shape_grad(&ctx->shape, rx, ry, g);
}

/* getSDF_DOTAtTimeStamp<false> SWM:799-806 (the analytic form below it is unreachable) */
static inline double sdf_dot_at_time(const orc_ctx *ctx, double px, double py, double t) {
  double t1 = dmax(0.0, t - 0.000001);
  double t2 = dmin(ctx->traj.traj_duration, t + 0.000001);
  double sdf1 = sdf_at_time(ctx, px, py, t1);
  double sdf2 = sdf_at_time(ctx, px, py, t2);
  return (sdf2 - sdf1) * 500000;
}

/* choiceTInit<false>(pos_eva, dt) SWM:538-581 */
static double choice_t_init(const orc_ctx *ctx, double px, double py, double dt) {
  double min_dis = 1e9, dis = 1e9, time_seed = 0.0;
  int pricision_layers = 4, current_layer = 1;
  double loop_terminal = ctx->traj.traj_duration;
  double t = 0.0;
  while (current_layer <= pricision_layers) {
    if (current_layer == 1) t = 0.0;
    if (current_layer > 1) {
      t = dmax(0.0, time_seed - 10 * dt);
      loop_terminal = dmin(ctx->traj.traj_duration, time_seed + 10 * dt);
    }
    for (; t <= loop_terminal; t += dt) {
      dis = sdf_at_time(ctx, px, py, t);
      if (dis < min_dis) { time_seed = t; min_dis = dis; }
    }
    dt *= 0.1;
    current_layer += 1;
  }
  return time_seed;
}

/* gradientDescent SWM:1249-1325 (momentum unused; asserts compiled out in Release) */
static void gradient_descent(const orc_ctx *ctx, double t_min, double t_max, const double x0,
                             double *fx, double *x, double px, double py) {
  int max_iter = 1000;
  double alpha = 0.01, tau = alpha, g = 0.0, tol = 1e-16;
  *x = x0;
  double projection = 0, change = 0, prev_x = 10000000.0;
  int iter = 0, stop = 0;
  double x_candidate, fx_candidate;
  g = 100.0;
  int passes = 0;
  unsigned char rec[32] = {0};
  while (iter < max_iter && !stop && fabs(*x - prev_x) > tol) {
    passes++;
    if (iter == 0) *fx = sdf_at_time(ctx, px, py, *x);
    g = sdf_dot_at_time(ctx, px, py, *x);
    tau = alpha;
    prev_x = *x;
    for (int div = 1; div < 30; div++) {
      iter = iter + 1;
      tl_cnt.gd_trials++;
      projection = *x;
      g = sdf_dot_at_time(ctx, px, py, projection);
      change = -tau * ((int)(g > 0) - (g < 0));
      x_candidate = *x + change;
      x_candidate = dmax(dmin(x_candidate, t_max), t_min);
      fx_candidate = sdf_at_time(ctx, px, py, x_candidate);
      if ((fx_candidate - *fx) < 0) {
        *x = x_candidate;
        *fx = fx_candidate;
        if (passes < 32) rec[passes] = (unsigned char)div;
        break;
      }
      tau = 0.5 * tau;
      if (div == 29) stop = 1;
    }
  }
  if (g_gd_trace) {
    size_t slot;
#ifdef _OPENMP
#pragma omp atomic capture
#endif
    slot = g_gd_trace_n++;
    if (slot < g_gd_trace_cap) {
      rec[0] = (unsigned char)(passes < 31 ? passes : 31);
      memcpy(g_gd_trace + 32 * slot, rec, 32);
    }
  }
  tl_cnt.gd_passes += passes;
  if (passes > tl_cnt.gd_max_passes) tl_cnt.gd_max_passes = passes;
  tl_cnt.gd_pass_hist[passes / 4 < 31 ? passes / 4 : 31]++;
}

/* getSDFofSweptVolume<false,true> SWM:844-866 */
static double sdf_swept(const orc_ctx *ctx, double px, double py, double *time_seed_f,
                        double grad[3]) {
  double t_star = 0.0, sdf_star = 0.0, dtime = 0.15;
  double ts = choice_t_init(ctx, px, py, dtime);
  double tmin_ = dmax(0.0, ts - 3.4);
  double tmax_ = dmin(ts + 3.4, ctx->traj.traj_duration);
  gradient_descent(ctx, tmin_, tmax_, ts, &sdf_star, &t_star, px, py);
  double g2[2];
  grad_prel_at_time(ctx, px, py, t_star, g2);
  grad[0] = g2[0]; grad[1] = g2[1]; grad[2] = 0.0;
  *time_seed_f = t_star;
  tl_cnt.solves++;
  return sdf_star;
}
double orc_sdf_swept(orc_ctx *ctx, double px, double py, double *t_star, double grad[3]) {
  return sdf_swept(ctx, px, py, t_star, grad);
}

/* SampleSet2D::getElements SWM:60-71 (one ring: rk = 1.0 only since rk_res = 1.5) */
#define ORC_MAX_ELEMS 64
static int sample_elements(double theta0, double theta_res, double rk0, double rk_res,
                           double *rk_out, double *th_out) {
  int n = 0;
  for (double rk = rk0; rk > 0; rk -= rk_res)
    for (double theta = theta0; theta < theta0 + 2 * ORC_PI; theta += theta_res) {
      if (n < ORC_MAX_ELEMS) { rk_out[n] = rk; th_out[n] = theta; }
      n++;
    }
  return n < ORC_MAX_ELEMS ? n : ORC_MAX_ELEMS;
}

/* getTrueSDFofSweptVolume<true> SWM:916-1018 */
static double true_sdf(const orc_ctx *ctx, double px, double py, double *time_seed_f,
                       double grad_prel[3]) {
  const orc_traj *tr = &ctx->traj;
  double argmin_dis = sdf_swept(ctx, px, py, time_seed_f, grad_prel);
  if (argmin_dis > 0) return argmin_dis; /* outside case */

  tl_cnt.interior_points++;
  double r0 = 10;
  double vel[3];
  traj_vel(tr, *time_seed_f, vel);
#define VNORM(v) sqrt((v)[0] * (v)[0] + (v)[1] * (v)[1] + (v)[2] * (v)[2])
  if (VNORM(vel) < 0.01) {
    if (*time_seed_f < 0.1) {
      for (double t_scan = *time_seed_f; t_scan <= tr->traj_duration; t_scan += 0.1) {
        traj_vel(tr, t_scan, vel);
        if (VNORM(vel) >= 0.01) break;
      }
    } else if (*time_seed_f > tr->traj_duration - 0.1) {
      for (double t_scan = *time_seed_f; t_scan >= 0; t_scan -= 0.1) {
        traj_vel(tr, t_scan, vel);
        if (VNORM(vel) >= 0.01) break;
      }
    }
  }
#undef VNORM
  /* SampleSet2D::initSet SWM:73-103 */
  double cxr = px, cyr = py, r = r0;
  double theta0 = trig_atan2(ctx, vel[0], -vel[1]);
  if (theta0 < 0) theta0 += 2 * ORC_PI;
  double theta_res = ORC_PI + 0.1, rk_res = 1.5, rk0 = 1.0;

  double el_rk[ORC_MAX_ELEMS], el_th[ORC_MAX_ELEMS];
  double star_rk = 0.0, star_th = 0.0; /* yk_star (uninitialised in the reference until first hit) */
  double r_star, max_g, cur_g, real_t_star = *time_seed_f;
  int iter = 1;
  while (1) {
    max_g = -100000;
    int Y_size = sample_elements(theta0, theta_res, rk0, rk_res, el_rk, el_th);
    for (int i = 0; i < Y_size; i++) {
      /* CircleCoord2D::getPosition SWM:36-39 */
      double sth, cth;
      trig_sincos(ctx, el_th[i], &sth, &cth);
      double ykx = cxr + el_rk[i] * r * cth;
      double yky = cyr + el_rk[i] * r * sth;
      cur_g = sdf_swept(ctx, ykx, yky, time_seed_f, grad_prel);
      if (cur_g > max_g) {
        max_g = cur_g;
        real_t_star = *time_seed_f;
        star_rk = el_rk[i];
        star_th = el_th[i];
      }
    }
    r_star = r - max_g;
    r = r_star;
    if (iter > 8) break;
    if (fabs(max_g) < 0.1) break;
    /* expandSet(2, theta*) SWM:105-110 */
    theta_res /= (2 + 1);
    theta_res = dmax(0.3, theta_res);
    theta0 = star_th;
    iter++;
  }
  double sst, cst;
  trig_sincos(ctx, star_th, &sst, &cst);
  double corx = cxr + star_rk * r_star * cst;
  double cory = cyr + star_rk * r_star * sst;
  double gx = corx - px, gy = cory - py;
  double z = gx * gx + gy * gy; /* normalize(): divide iff squaredNorm > 0 */
  if (z > 0.0) { double n = sqrt(z); gx = gx / n; gy = gy / n; }
  grad_prel[0] = gx; grad_prel[1] = gy; grad_prel[2] = 0.0;
  *time_seed_f = real_t_star;
  return -r_star;
}
double orc_true_sdf(orc_ctx *ctx, double px, double py, double *t_star, double grad[3]) {
  return true_sdf(ctx, px, py, t_star, grad);
}

/* ------------------------------------------------------------------------- */
/* per-point penalty + reduction                                              */
/* ------------------------------------------------------------------------- */
/* smoothedL1 BEO:316-340 */
int orc_smoothed_l1(double x, double mu, double *f, double *df) {
  if (x < 0.0) return 0;
  else if (x > mu) { *f = x - 0.5 * mu; *df = 1.0; return 1; }
  else {
    const double xdmu = x / mu;
    const double sqrxdmu = xdmu * xdmu;
    const double mumxd2 = mu - 0.5 * x;
    *f = mumxd2 * sqrxdmu * xdmu;
    *df = sqrxdmu * ((-0.5) * xdmu + 3.0 * mumxd2 / mu);
    return 1;
  }
}

typedef struct {
  int piece;
  double gdC[18]; /* [k*3 + d] */
  double gdT;
  double pena;
  double sdf, tstar;
} point_contrib;

/* loop body of BEO:786-865 (everything before the critical section) */
static void point_contribution(const orc_ctx *ctx, double px, double py, point_contrib *out) {
  const orc_traj *tr = &ctx->traj;
  const double weightPos = ctx->weight_p;
  double gradp_rel[3];
  double time_star = 0.0;
  double sdf_value = true_sdf(ctx, px, py, &time_star, gradp_rel);
  double time_local = time_star;
  int i = traj_locate(tr, &time_local);
  const double *c = tr->c + (size_t)i * 18;
  double s1 = time_local, s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
  double beta0[6] = {1.0, s1, s2, s3, s4, s5};
  double beta1[6] = {0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4};
  double pos[3] = {0, 0, 0}, vel[3] = {0, 0, 0};
  for (int d = 0; d < 3; ++d) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < 6; ++k) { a += c[k * 3 + d] * beta0[k]; b += c[k * 3 + d] * beta1[k]; }
    pos[d] = a; vel[d] = b;
  }
  double yaw = pos[2];
  double sy, cy; /* rotate = AngleAxisd(yaw, Z) */
  trig_sincos(ctx, yaw, &sy, &cy);
  pos[2] = 0.0;
  if (sdf_value < 0) { /* BEO:832: gradp_rel = rotate^T * gradp_rel */
    double gx = cy * gradp_rel[0] + sy * gradp_rel[1];
    double gy = (-sy) * gradp_rel[0] + cy * gradp_rel[1];
    gradp_rel[0] = gx; gradp_rel[1] = gy;
  }
  /* grad_cost_p_sw BEO:1031-1066 (St = I) */
  double costp = 0.0, gradp[2] = {0, 0}, grad_yaw = 0.0;
  double sdf_cost = -1.0, sdf_out_grad = 0.0;
  orc_smoothed_l1(ctx->safety_hor - sdf_value, 0.01, &sdf_cost, &sdf_out_grad);
  /* sdf_grad = -L' * ( -(St^-1)^T * rotate * gradp_rel ) */
  double mrx = (-cy) * gradp_rel[0] + (sy) * gradp_rel[1];
  double mry = (-sy) * gradp_rel[0] + (-cy) * gradp_rel[1];
  double sgx = -sdf_out_grad * mrx, sgy = -sdf_out_grad * mry;
  if (sdf_cost > 0) {
    costp += sdf_cost;
    gradp[0] += sgx; gradp[1] += sgy;
    double dx = px - pos[0], dy = py - pos[1];
    /* VR_theta^T * p_minus_x, VR_theta = [[-s,-c,0],[c,-s,0],[0,0,1]] */
    double v0 = (-sy) * dx + (cy) * dy;
    double v1 = (-cy) * dx + (-sy) * dy;
    grad_yaw = (-sdf_out_grad * gradp_rel[0]) * v0 + (-sdf_out_grad * gradp_rel[1]) * v1;
  }
  double gPx = 0.0, gPy = 0.0, gYaw = 0.0, pena = 0.0;
  if (costp > 0) {
    gPx += weightPos * gradp[0];
    gPy += weightPos * gradp[1];
    gYaw += weightPos * grad_yaw;
    pena += weightPos * costp;
  }
  out->piece = i;
  for (int k = 0; k < 6; ++k) {
    out->gdC[k * 3 + 0] = beta0[k] * gPx;
    out->gdC[k * 3 + 1] = beta0[k] * gPy;
    out->gdC[k * 3 + 2] = beta0[k] * gYaw;
  }
  out->gdT = -((gPx * vel[0] + gPy * vel[1]) + gYaw * vel[2]);
  out->pena = pena;
  out->sdf = sdf_value;
  out->tstar = time_star;
}

static void fold_counters(orc_ctx *ctx) {
#ifdef _OPENMP
#pragma omp critical(orc_cnt)
#endif
  {
    ctx->cnt.sdf_evals += tl_cnt.sdf_evals;
    ctx->cnt.shape_evals += tl_cnt.shape_evals;
    ctx->cnt.solves += tl_cnt.solves;
    ctx->cnt.interior_points += tl_cnt.interior_points;
    ctx->cnt.gd_trials += tl_cnt.gd_trials;
    ctx->cnt.gd_passes += tl_cnt.gd_passes;
    if (tl_cnt.gd_max_passes > ctx->cnt.gd_max_passes) ctx->cnt.gd_max_passes = tl_cnt.gd_max_passes;
    for (int i = 0; i < 32; ++i) ctx->cnt.gd_pass_hist[i] += tl_cnt.gd_pass_hist[i];
  }
  memset(&tl_cnt, 0, sizeof(tl_cnt));
}

void orc_get_counters(const orc_ctx *ctx, orc_counters *out) { *out = ctx->cnt; }

void orc_query(orc_ctx *ctx, const double *xyz, size_t P, int nthreads, double *sdf,
               double *tstar, double *grad_xy) {
  memset(&ctx->cnt, 0, sizeof(ctx->cnt));
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
    memset(&tl_cnt, 0, sizeof(tl_cnt));
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
    for (long long k = 0; k < (long long)P; ++k) {
      double g[3], ts = 0.0;
      double v = true_sdf(ctx, xyz[3 * k], xyz[3 * k + 1], &ts, g);
      if (sdf) sdf[k] = v;
      if (tstar) tstar[k] = ts;
      if (grad_xy) { grad_xy[2 * k] = g[0]; grad_xy[2 * k + 1] = g[1]; }
    }
    fold_counters(ctx);
  }
}

void orc_penalty(orc_ctx *ctx, const double *xyz, size_t P, int nthreads, int sum_mode,
                 double *cost, double *gradT, double *gradC, double *sdf, double *tstar,
                 double *pcost) {
  const int N = ctx->traj.N;
  memset(&ctx->cnt, 0, sizeof(ctx->cnt));
  if (nthreads < 1) nthreads = 1;
  point_contrib *pc = (point_contrib *)malloc(sizeof(point_contrib) * (P ? P : 1));
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
    memset(&tl_cnt, 0, sizeof(tl_cnt));
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
    for (long long k = 0; k < (long long)P; ++k)
      point_contribution(ctx, xyz[3 * k], xyz[3 * k + 1], &pc[k]); /* z forced to 0 (BEO:791) */
    fold_counters(ctx);
  }
  /* reduction of BEO:855-863 in index order */
  if (sum_mode == 0) {
    for (size_t k = 0; k < P; ++k) {
      const point_contrib *q = &pc[k];
      *cost += q->pena;
      for (int kk = 0; kk < 6; ++kk)
        for (int d = 0; d < 3; ++d) gradC[(size_t)d * 6 * N + 6 * q->piece + kk] += q->gdC[kk * 3 + d];
      for (int j = 0; j < q->piece; ++j) gradT[j] += q->gdT;
    }
  } else {
    long double lc = 0.0L;
    long double *lC = (long double *)calloc((size_t)18 * N, sizeof(long double));
    long double *lH = (long double *)calloc((size_t)N + 1, sizeof(long double));
    for (size_t k = 0; k < P; ++k) {
      const point_contrib *q = &pc[k];
      lc += q->pena;
      for (int kk = 0; kk < 6; ++kk)
        for (int d = 0; d < 3; ++d) lC[(size_t)d * 6 * N + 6 * q->piece + kk] += q->gdC[kk * 3 + d];
      lH[q->piece] += q->gdT;
    }
    *cost += (double)lc;
    for (size_t i = 0; i < (size_t)18 * N; ++i) gradC[i] += (double)lC[i];
    long double suf = 0.0L; /* gradT[j] += sum_{piece > j} gdT */
    for (int j = N - 1; j >= 0; --j) { gradT[j] += (double)suf; suf += lH[j]; }
    free(lC);
    free(lH);
  }
  for (size_t k = 0; k < P; ++k) {
    if (sdf) sdf[k] = pc[k].sdf;
    if (tstar) tstar[k] = pc[k].tstar;
    if (pcost) pcost[k] = pc[k].pena;
  }
  free(pc);
}

/* ------------------------------------------------------------------------- */
/* context                                                                    */
/* ------------------------------------------------------------------------- */
orc_ctx *orc_create(int shape_id, const double poly_params[3], const double *poly_xy, int nverts,
                    double safety_hor, double weight_p, double rho, const double head_state[9],
                    const double tail_state[9]) {
  orc_ctx *ctx = (orc_ctx *)calloc(1, sizeof(orc_ctx));
  orc_shape_init(&ctx->shape, shape_id, poly_params, poly_xy, nverts);
  ctx->safety_hor = safety_hor;
  ctx->weight_p = weight_p;
  ctx->rho = rho;
  if (head_state) memcpy(ctx->head, head_state, sizeof(double) * 9);
  if (tail_state) memcpy(ctx->tail, tail_state, sizeof(double) * 9);
  return ctx;
}
void orc_destroy(orc_ctx *ctx) {
  if (!ctx) return;
  free(ctx->traj.T);
  free(ctx->traj.c);
  free(ctx);
}

int orc_set_polygon_loops(orc_ctx *ctx, const int *loop_sizes, int nloops) {
  return ctx ? orc_shape_set_loops(&ctx->shape, loop_sizes, nloops) : -1;
}

/* minco.getTrajectory MNC:515-528 + SweptVolumeManager::updateTraj SWM:376-385 */
void orc_set_traj(orc_ctx *ctx, int N, const double *coeffs_colmajor, const double *T) {
  orc_traj *tr = &ctx->traj;
  if (tr->N != N) {
    free(tr->T); free(tr->c);
    tr->T = (double *)malloc(sizeof(double) * N);
    tr->c = (double *)malloc(sizeof(double) * 18 * N);
    tr->N = N;
  }
  for (int i = 0; i < N; ++i) {
    tr->T[i] = T[i];
    for (int k = 0; k < 6; ++k)
      for (int d = 0; d < 3; ++d)
        tr->c[((size_t)i * 6 + k) * 3 + d] = coeffs_colmajor[(size_t)d * 6 * N + 6 * i + k];
  }
  double td = 0.0; /* getTotalDuration TRJ:410-419 */
  for (int i = 0; i < N; ++i) td += T[i];
  if (td < 3 * 1e2 || !ctx->have_duration) { /* first call always sets (member is uninitialised in the reference) */
    tr->traj_duration = td;
    ctx->have_duration = 1;
  }
}

/* ------------------------------------------------------------------------- */
/* MINCO S3NU (MNC:397-655) with BandedSystem (MNC:43-198), tau/xi maps, a14   */
/* ------------------------------------------------------------------------- */
typedef struct {
  int n, lb, ub;
  double *d;
} banded;
#define BA(A, i, j) ((A)->d[((i) - (j) + (A)->ub) * (A)->n + (j)])

static void banded_create(banded *A, int n, int p, int q) {
  A->n = n; A->lb = p; A->ub = q;
  A->d = (double *)calloc((size_t)n * (p + q + 1), sizeof(double));
}
static void banded_factorize(banded *A) { /* MNC:96-128 */
  int n = A->n;
  for (int k = 0; k <= n - 2; k++) {
    int iM = (k + A->lb < n - 1) ? k + A->lb : n - 1;
    double cVl = BA(A, k, k);
    for (int i = k + 1; i <= iM; i++)
      if (BA(A, i, k) != 0.0) BA(A, i, k) /= cVl;
    int jM = (k + A->ub < n - 1) ? k + A->ub : n - 1;
    for (int j = k + 1; j <= jM; j++) {
      cVl = BA(A, k, j);
      if (cVl != 0.0)
        for (int i = k + 1; i <= iM; i++)
          if (BA(A, i, k) != 0.0) BA(A, i, j) -= BA(A, i, k) * cVl;
    }
  }
}
/* b: n x 3 row-major here (b[i*3+d]) */
static void banded_solve(const banded *A, double *b) { /* MNC:133-163 */
  int n = A->n;
  for (int j = 0; j <= n - 1; j++) {
    int iM = (j + A->lb < n - 1) ? j + A->lb : n - 1;
    for (int i = j + 1; i <= iM; i++)
      if (BA(A, i, j) != 0.0)
        for (int d = 0; d < 3; ++d) b[i * 3 + d] -= BA(A, i, j) * b[j * 3 + d];
  }
  for (int j = n - 1; j >= 0; j--) {
    for (int d = 0; d < 3; ++d) b[j * 3 + d] /= BA(A, j, j);
    int iM = (0 > j - A->ub) ? 0 : j - A->ub;
    for (int i = iM; i <= j - 1; i++)
      if (BA(A, i, j) != 0.0)
        for (int d = 0; d < 3; ++d) b[i * 3 + d] -= BA(A, i, j) * b[j * 3 + d];
  }
}
static void banded_solve_adj(const banded *A, double *b) { /* MNC:168-197 */
  int n = A->n;
  for (int j = 0; j <= n - 1; j++) {
    for (int d = 0; d < 3; ++d) b[j * 3 + d] /= BA(A, j, j);
    int iM = (j + A->ub < n - 1) ? j + A->ub : n - 1;
    for (int i = j + 1; i <= iM; i++)
      if (BA(A, j, i) != 0.0)
        for (int d = 0; d < 3; ++d) b[i * 3 + d] -= BA(A, j, i) * b[j * 3 + d];
  }
  for (int j = n - 1; j >= 0; j--) {
    int iM = (0 > j - A->lb) ? 0 : j - A->lb;
    for (int i = iM; i <= j - 1; i++)
      if (BA(A, j, i) != 0.0)
        for (int d = 0; d < 3; ++d) b[i * 3 + d] -= BA(A, j, i) * b[j * 3 + d];
  }
}

typedef struct {
  int N;
  banded A;
  double *b; /* 6N x 3 row-major */
  double *T1, *T2, *T3, *T4, *T5;
} minco_s3;

static void minco_free(minco_s3 *m) {
  free(m->A.d); free(m->b); free(m->T1); free(m->T2); free(m->T3); free(m->T4); free(m->T5);
}

/* setConditions MNC:418-433 + setParameters MNC:435-513.  head/tail: 3x3 col-major,
 * inPs: 3 x (N-1) col-major. */
static void minco_set(minco_s3 *m, const double *head, const double *tail, int N,
                      const double *inPs, const double *ts) {
  m->N = N;
  banded_create(&m->A, 6 * N, 6, 6);
  m->b = (double *)calloc((size_t)6 * N * 3, sizeof(double));
  m->T1 = (double *)malloc(sizeof(double) * N); m->T2 = (double *)malloc(sizeof(double) * N);
  m->T3 = (double *)malloc(sizeof(double) * N); m->T4 = (double *)malloc(sizeof(double) * N);
  m->T5 = (double *)malloc(sizeof(double) * N);
  for (int i = 0; i < N; ++i) {
    m->T1[i] = ts[i];
    m->T2[i] = m->T1[i] * m->T1[i];
    m->T3[i] = m->T2[i] * m->T1[i];
    m->T4[i] = m->T2[i] * m->T2[i];
    m->T5[i] = m->T4[i] * m->T1[i];
  }
  banded *A = &m->A;
  double *b = m->b;
  const double *T1 = m->T1, *T2 = m->T2, *T3 = m->T3, *T4 = m->T4, *T5 = m->T5;
  BA(A, 0, 0) = 1.0; BA(A, 1, 1) = 1.0; BA(A, 2, 2) = 2.0;
  for (int d = 0; d < 3; ++d) { b[0 * 3 + d] = head[0 * 3 + d]; b[1 * 3 + d] = head[1 * 3 + d]; b[2 * 3 + d] = head[2 * 3 + d]; }
  for (int i = 0; i < N - 1; i++) {
    BA(A, 6 * i + 3, 6 * i + 3) = 6.0;
    BA(A, 6 * i + 3, 6 * i + 4) = 24.0 * T1[i];
    BA(A, 6 * i + 3, 6 * i + 5) = 60.0 * T2[i];
    BA(A, 6 * i + 3, 6 * i + 9) = -6.0;
    BA(A, 6 * i + 4, 6 * i + 4) = 24.0;
    BA(A, 6 * i + 4, 6 * i + 5) = 120.0 * T1[i];
    BA(A, 6 * i + 4, 6 * i + 10) = -24.0;
    BA(A, 6 * i + 5, 6 * i) = 1.0;
    BA(A, 6 * i + 5, 6 * i + 1) = T1[i];
    BA(A, 6 * i + 5, 6 * i + 2) = T2[i];
    BA(A, 6 * i + 5, 6 * i + 3) = T3[i];
    BA(A, 6 * i + 5, 6 * i + 4) = T4[i];
    BA(A, 6 * i + 5, 6 * i + 5) = T5[i];
    BA(A, 6 * i + 6, 6 * i) = 1.0;
    BA(A, 6 * i + 6, 6 * i + 1) = T1[i];
    BA(A, 6 * i + 6, 6 * i + 2) = T2[i];
    BA(A, 6 * i + 6, 6 * i + 3) = T3[i];
    BA(A, 6 * i + 6, 6 * i + 4) = T4[i];
    BA(A, 6 * i + 6, 6 * i + 5) = T5[i];
    BA(A, 6 * i + 6, 6 * i + 6) = -1.0;
    BA(A, 6 * i + 7, 6 * i + 1) = 1.0;
    BA(A, 6 * i + 7, 6 * i + 2) = 2 * T1[i];
    BA(A, 6 * i + 7, 6 * i + 3) = 3 * T2[i];
    BA(A, 6 * i + 7, 6 * i + 4) = 4 * T3[i];
    BA(A, 6 * i + 7, 6 * i + 5) = 5 * T4[i];
    BA(A, 6 * i + 7, 6 * i + 7) = -1.0;
    BA(A, 6 * i + 8, 6 * i + 2) = 2.0;
    BA(A, 6 * i + 8, 6 * i + 3) = 6 * T1[i];
    BA(A, 6 * i + 8, 6 * i + 4) = 12 * T2[i];
    BA(A, 6 * i + 8, 6 * i + 5) = 20 * T3[i];
    BA(A, 6 * i + 8, 6 * i + 8) = -2.0;
    for (int d = 0; d < 3; ++d) b[(6 * i + 5) * 3 + d] = inPs[i * 3 + d];
  }
  BA(A, 6 * N - 3, 6 * N - 6) = 1.0;
  BA(A, 6 * N - 3, 6 * N - 5) = T1[N - 1];
  BA(A, 6 * N - 3, 6 * N - 4) = T2[N - 1];
  BA(A, 6 * N - 3, 6 * N - 3) = T3[N - 1];
  BA(A, 6 * N - 3, 6 * N - 2) = T4[N - 1];
  BA(A, 6 * N - 3, 6 * N - 1) = T5[N - 1];
  BA(A, 6 * N - 2, 6 * N - 5) = 1.0;
  BA(A, 6 * N - 2, 6 * N - 4) = 2 * T1[N - 1];
  BA(A, 6 * N - 2, 6 * N - 3) = 3 * T2[N - 1];
  BA(A, 6 * N - 2, 6 * N - 2) = 4 * T3[N - 1];
  BA(A, 6 * N - 2, 6 * N - 1) = 5 * T4[N - 1];
  BA(A, 6 * N - 1, 6 * N - 4) = 2;
  BA(A, 6 * N - 1, 6 * N - 3) = 6 * T1[N - 1];
  BA(A, 6 * N - 1, 6 * N - 2) = 12 * T2[N - 1];
  BA(A, 6 * N - 1, 6 * N - 1) = 20 * T3[N - 1];
  for (int d = 0; d < 3; ++d) {
    b[(6 * N - 3) * 3 + d] = tail[0 * 3 + d];
    b[(6 * N - 2) * 3 + d] = tail[1 * 3 + d];
    b[(6 * N - 1) * 3 + d] = tail[2 * 3 + d];
  }
  banded_factorize(A);
  banded_solve(A, b);
}

#define ROWDOT(b, r1, r2) (((b)[(r1) * 3] * (b)[(r2) * 3] + (b)[(r1) * 3 + 1] * (b)[(r2) * 3 + 1]) + (b)[(r1) * 3 + 2] * (b)[(r2) * 3 + 2])

static double minco_energy(const minco_s3 *m) { /* MNC:530-543 */
  double energy = 0.0;
  const double *b = m->b;
  for (int i = 0; i < m->N; i++) {
    energy += 36.0 * ROWDOT(b, 6 * i + 3, 6 * i + 3) * m->T1[i] +
              144.0 * ROWDOT(b, 6 * i + 4, 6 * i + 3) * m->T2[i] +
              192.0 * ROWDOT(b, 6 * i + 4, 6 * i + 4) * m->T3[i] +
              240.0 * ROWDOT(b, 6 * i + 5, 6 * i + 3) * m->T3[i] +
              720.0 * ROWDOT(b, 6 * i + 5, 6 * i + 4) * m->T4[i] +
              720.0 * ROWDOT(b, 6 * i + 5, 6 * i + 5) * m->T5[i];
  }
  return energy;
}
/* gdC: 6N x 3 row-major.  MNC:550-567 */
static void minco_energy_grad_c(const minco_s3 *m, double *gdC) {
  const double *b = m->b;
  for (int i = 0; i < m->N; i++)
    for (int d = 0; d < 3; ++d) {
      double b3 = b[(6 * i + 3) * 3 + d], b4 = b[(6 * i + 4) * 3 + d], b5 = b[(6 * i + 5) * 3 + d];
      gdC[(6 * i + 5) * 3 + d] = 240.0 * b3 * m->T3[i] + 720.0 * b4 * m->T4[i] + 1440.0 * b5 * m->T5[i];
      gdC[(6 * i + 4) * 3 + d] = 144.0 * b3 * m->T2[i] + 384.0 * b4 * m->T3[i] + 720.0 * b5 * m->T4[i];
      gdC[(6 * i + 3) * 3 + d] = 72.0 * b3 * m->T1[i] + 144.0 * b4 * m->T2[i] + 240.0 * b5 * m->T3[i];
      gdC[(6 * i + 0) * 3 + d] = 0.0;
      gdC[(6 * i + 1) * 3 + d] = 0.0;
      gdC[(6 * i + 2) * 3 + d] = 0.0;
    }
}
static void minco_energy_grad_t(const minco_s3 *m, double *gdT) { /* MNC:569-582 */
  const double *b = m->b;
  for (int i = 0; i < m->N; i++)
    gdT[i] = 36.0 * ROWDOT(b, 6 * i + 3, 6 * i + 3) +
             288.0 * ROWDOT(b, 6 * i + 4, 6 * i + 3) * m->T1[i] +
             576.0 * ROWDOT(b, 6 * i + 4, 6 * i + 4) * m->T2[i] +
             720.0 * ROWDOT(b, 6 * i + 5, 6 * i + 3) * m->T2[i] +
             2880.0 * ROWDOT(b, 6 * i + 5, 6 * i + 4) * m->T3[i] +
             3600.0 * ROWDOT(b, 6 * i + 5, 6 * i + 5) * m->T4[i];
}
/* propogateGrad MNC:584-654.  partialGradByCoeffs 6N x 3 row-major (consumed as adjGrad). */
static void minco_propagate(const minco_s3 *m, double *adj, const double *pgT, double *gradP,
                            double *gradTimes) {
  const int N = m->N;
  const double *b = m->b, *T1 = m->T1, *T2 = m->T2, *T3 = m->T3, *T4 = m->T4;
  banded_solve_adj(&m->A, adj);
  for (int i = 0; i < N - 1; i++)
    for (int d = 0; d < 3; ++d) gradP[i * 3 + d] = adj[(6 * i + 5) * 3 + d];
  double B1[6][3], B2[3][3];
  for (int i = 0; i < N - 1; i++) {
    for (int d = 0; d < 3; ++d) {
      const double c1 = b[(i * 6 + 1) * 3 + d], c2 = b[(i * 6 + 2) * 3 + d], c3 = b[(i * 6 + 3) * 3 + d],
                   c4 = b[(i * 6 + 4) * 3 + d], c5 = b[(i * 6 + 5) * 3 + d];
      B1[2][d] = -(c1 + 2.0 * T1[i] * c2 + 3.0 * T2[i] * c3 + 4.0 * T3[i] * c4 + 5.0 * T4[i] * c5);
      B1[3][d] = B1[2][d];
      B1[4][d] = -(2.0 * c2 + 6.0 * T1[i] * c3 + 12.0 * T2[i] * c4 + 20.0 * T3[i] * c5);
      B1[5][d] = -(6.0 * c3 + 24.0 * T1[i] * c4 + 60.0 * T2[i] * c5);
      B1[0][d] = -(24.0 * c4 + 120.0 * T1[i] * c5);
      B1[1][d] = -120.0 * c5;
    }
    double s = 0.0; /* B1.cwiseProduct(adjGrad.block<6,3>(6i+3,0)).sum(): column-major traversal */
    for (int d = 0; d < 3; ++d)
      for (int r = 0; r < 6; ++r) s += B1[r][d] * adj[(6 * i + 3 + r) * 3 + d];
    gradTimes[i] = s;
  }
  for (int d = 0; d < 3; ++d) {
    const double c1 = b[(6 * N - 5) * 3 + d], c2 = b[(6 * N - 4) * 3 + d], c3 = b[(6 * N - 3) * 3 + d],
                 c4 = b[(6 * N - 2) * 3 + d], c5 = b[(6 * N - 1) * 3 + d];
    B2[0][d] = -(c1 + 2.0 * T1[N - 1] * c2 + 3.0 * T2[N - 1] * c3 + 4.0 * T3[N - 1] * c4 + 5.0 * T4[N - 1] * c5);
    B2[1][d] = -(2.0 * c2 + 6.0 * T1[N - 1] * c3 + 12.0 * T2[N - 1] * c4 + 20.0 * T3[N - 1] * c5);
    B2[2][d] = -(6.0 * c3 + 24.0 * T1[N - 1] * c4 + 60.0 * T2[N - 1] * c5);
  }
  double s = 0.0;
  for (int d = 0; d < 3; ++d)
    for (int r = 0; r < 3; ++r) s += B2[r][d] * adj[(6 * N - 3 + r) * 3 + d];
  gradTimes[N - 1] = s;
  for (int i = 0; i < N; ++i) gradTimes[i] += pgT[i];
}

void orc_minco_coeffs(const double head_state[9], const double tail_state[9], int N,
                      const double *inPs, const double *T, double *coeffs_colmajor) {
  minco_s3 m;
  minco_set(&m, head_state, tail_state, N, inPs, T);
  for (int r = 0; r < 6 * N; ++r)
    for (int d = 0; d < 3; ++d) coeffs_colmajor[(size_t)d * 6 * N + r] = m.b[r * 3 + d];
  minco_free(&m);
}

void orc_forward_T(const double *tau, double *T, int N) { /* BEO:213-226 */
  for (int i = 0; i < N; i++) {
    double temp = tau[i];
    T[i] = temp > 0.0 ? ((0.5 * temp + 1.0) * temp + 1.0) : 1.0 / ((0.5 * temp - 1.0) * (temp) + 1.0);
  }
}
void orc_backward_T(const double *T, double *tau, int N) { /* BEO:228-241 */
  for (int i = 0; i < N; i++)
    tau[i] = T[i] > 1.0 ? (sqrt(2.0 * T[i] - 1.0) - 1.0) : (1.0 - sqrt(2.0 / T[i] - 1.0));
}

/* costFunctionLmbmParallel BEO:344-408 */
double orc_cost_function(orc_ctx *ctx, const double *xyz, size_t P, int nthreads, const double *x,
                         double *g, int n, double *costs3) {
  const int N = (n + 3) / 4; /* n = N + 3(N-1) */
  const int dimTau = N;
  double *T = (double *)malloc(sizeof(double) * N);
  orc_forward_T(x, T, dimTau);
  const double *inPs = x + dimTau; /* forwardP: P.col(i) = xi[3i..3i+3) */
  minco_s3 m;
  minco_set(&m, ctx->head, ctx->tail, N, inPs, T);
  double cost = minco_energy(&m);
  double energy_cost = cost;
  double *pgC = (double *)malloc(sizeof(double) * 18 * N); /* row-major */
  double *pgT = (double *)malloc(sizeof(double) * N);
  minco_energy_grad_c(&m, pgC);
  minco_energy_grad_t(&m, pgT);
  /* getTrajectory + updateTraj */
  double *cm = (double *)malloc(sizeof(double) * 18 * N);
  double *gC = (double *)malloc(sizeof(double) * 18 * N); /* col-major accumulators */
  for (int r = 0; r < 6 * N; ++r)
    for (int d = 0; d < 3; ++d) {
      cm[(size_t)d * 6 * N + r] = m.b[r * 3 + d];
      gC[(size_t)d * 6 * N + r] = pgC[r * 3 + d];
    }
  orc_set_traj(ctx, N, cm, T);
  orc_penalty(ctx, xyz, P, nthreads, 1, &cost, pgT, gC, NULL, NULL, NULL);
  for (int r = 0; r < 6 * N; ++r)
    for (int d = 0; d < 3; ++d) pgC[r * 3 + d] = gC[(size_t)d * 6 * N + r];
  double pos_cost = cost - energy_cost;
  double *gradP = (double *)malloc(sizeof(double) * 3 * (N > 1 ? N - 1 : 1));
  double *gradTimes = (double *)malloc(sizeof(double) * N);
  minco_propagate(&m, pgC, pgT, gradP, gradTimes);
  double tsum = 0.0;
  for (int i = 0; i < N; ++i) tsum += T[i];
  cost += ctx->rho * tsum;
  for (int i = 0; i < N; ++i) gradTimes[i] += ctx->rho;
  if (costs3) { costs3[0] = pos_cost; costs3[1] = cost - pos_cost; costs3[2] = cost; }
  /* backwardGradT BEO:268-289 */
  for (int i = 0; i < dimTau; i++) {
    if (x[i] > 0) g[i] = gradTimes[i] * (x[i] + 1.0);
    else {
      double denSqrt = (0.5 * x[i] - 1.0) * x[i] + 1.0;
      g[i] = gradTimes[i] * (1.0 - x[i]) / (denSqrt * denSqrt);
    }
  }
  /* backwardGradP BEO:303-314 */
  for (int i = 0; i < N - 1; ++i)
    for (int d = 0; d < 3; ++d) g[dimTau + 3 * i + d] = gradP[i * 3 + d];
  free(T); free(pgC); free(pgT); free(cm); free(gC); free(gradP); free(gradTimes);
  minco_free(&m);
  return cost;
}

/* ------------------------------------------------------------------------- */
/* query-point producer                                                       */
/* ------------------------------------------------------------------------- */
typedef struct {
  double res, bmin[3], bmax[3];
  int X, Y, Z;
  double *grid; /* counts, then 0/1 */
} orc_grid;

static int grid_in_map(const orc_grid *g, const double p[3]) { /* Gridmap3D.cpp:43-71 */
  for (int d = 0; d < 3; ++d)
    if (p[d] < g->bmin[d] || p[d] > g->bmax[d]) return 0;
  return 1;
}
static void grid_index(const orc_grid *g, const double p[3], int id[3]) { /* Gridmap3D.cpp:137-177 */
  if (!grid_in_map(g, p)) { id[0] = id[1] = id[2] = 0; return; }
  int ix = (int)floor((p[0] - g->bmin[0]) / g->res);
  int iy = (int)floor((p[1] - g->bmin[1]) / g->res);
  int iz = (int)floor((p[2] - g->bmin[2]) / g->res);
  if (ix < 0) ix = 0;
  if (ix >= g->X) ix = g->X - 1;
  if (iy < 0) ix = 0; /* sic */
  if (iy >= g->Y) iy = g->Y - 1;
  if (iz < 0) ix = 0; /* sic */
  if (iz >= g->Z) iz = g->Z - 1;
  id[0] = ix; id[1] = iy; id[2] = iz;
}
static int grid_addr(const orc_grid *g, int i, int j, int k) { return i * g->Y * g->Z + j * g->Z + k; }
static int grid_occupied(const orc_grid *g, int i, int j, int k) { /* Gridmap3D.cpp:239-283 */
  if (i < 0 || i >= g->X || j < 0 || j >= g->Y || k < 0 || k >= g->Z) return 1;
  return g->grid[grid_addr(g, i, j, k)] == 0 ? 0 : 1;
}
static void proj_in_map(const orc_grid *g, double p[3]) { /* PCSmap_manager.h:128-135 */
  for (int d = 0; d < 3; ++d) {
    if (p[d] < g->bmin[d]) p[d] = g->bmin[d];
    if (p[d] > g->bmax[d]) p[d] = g->bmax[d];
  }
}
static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }

size_t orc_map_points(const float *cloud, size_t n, double resolution, int sta_threshold,
                      const double *centres, size_t m, const double halfbd[3], double *out_xyz,
                      size_t cap, int dims_out[3]) {
  orc_grid g;
  g.res = resolution;
  for (int d = 0; d < 3; ++d) { g.bmin[d] = 999999999; g.bmax[d] = -999999999; }
  for (size_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      if (cloud[3 * i + d] > g.bmax[d]) g.bmax[d] = cloud[3 * i + d];
      if (cloud[3 * i + d] < g.bmin[d]) g.bmin[d] = cloud[3 * i + d];
    }
  g.X = (int)ceil((g.bmax[0] - g.bmin[0]) / g.res); /* Gridmap3D.cpp:29-31 */
  g.Y = (int)ceil((g.bmax[1] - g.bmin[1]) / g.res);
  g.Z = (int)ceil((g.bmax[2] - g.bmin[2]) / g.res);
  if (dims_out) { dims_out[0] = g.X; dims_out[1] = g.Y; dims_out[2] = g.Z; }
  const size_t total = (size_t)g.X * g.Y * g.Z;
  g.grid = (double *)calloc(total ? total : 1, sizeof(double));
  for (size_t i = 0; i < n; ++i) {
    double p[3] = {cloud[3 * i], cloud[3 * i + 1], cloud[3 * i + 2]};
    int id[3];
    grid_index(&g, p, id);
    g.grid[grid_addr(&g, id[0], id[1], id[2])]++;
  }
  for (size_t a = 0; a < total; ++a) g.grid[a] = (g.grid[a] >= sta_threshold) ? 1 : 0;
  /* plan_manager.cpp:156-167 */
  int *ids = (int *)malloc(sizeof(int) * (total ? total : 1));
  unsigned char *seen = (unsigned char *)calloc(total ? total : 1, 1);
  size_t nid = 0;
  double last[3] = {999, 999, 999};
  for (size_t c = 0; c < m; ++c) {
    double c1[3], c2[3], l1[3], l2[3];
    for (int d = 0; d < 3; ++d) {
      c1[d] = centres[3 * c + d] - halfbd[d]; c2[d] = centres[3 * c + d] + halfbd[d];
      l1[d] = last[d] - halfbd[d]; l2[d] = last[d] + halfbd[d];
    }
    proj_in_map(&g, c1); proj_in_map(&g, c2); proj_in_map(&g, l1); proj_in_map(&g, l2);
    int i1[3], i2[3], j1[3], j2[3];
    grid_index(&g, c1, i1); grid_index(&g, c2, i2); grid_index(&g, l1, j1); grid_index(&g, l2, j2);
    for (int i = i1[0]; i <= i2[0]; i++)
      for (int j = i1[1]; j <= i2[1]; j++)
        for (int k = i1[2]; k <= i2[2]; k++)
          if (i > j2[0] || i < j1[0] || j > j2[1] || j < j1[1] || k > j2[2] || k < j1[2])
            if (grid_occupied(&g, i, j, k)) {
              const int uid = k * g.X * g.Y + j * g.X + i; /* unifiedID, PCSmap_manager.h:118-125 */
              if (!seen[uid]) { seen[uid] = 1; ids[nid++] = uid; }
            }
    for (int d = 0; d < 3; ++d) last[d] = centres[3 * c + d];
  }
  qsort(ids, nid, sizeof(int), cmp_int);
  for (size_t q = 0; q < nid && q < cap; ++q) {
    const int uid = ids[q];
    const int i = uid % g.X, j = (uid / g.X) % g.Y, k = uid / (g.X * g.Y);
    out_xyz[3 * q + 0] = (i + 0.5) * g.res + g.bmin[0]; /* getGridCubeCenter, Gridmap3D.cpp:184-195 */
    out_xyz[3 * q + 1] = (j + 0.5) * g.res + g.bmin[1];
    out_xyz[3 * q + 2] = (k + 0.5) * g.res + g.bmin[2];
  }
  free(ids); free(seen); free(g.grid);
  return nid;
}

/* ------------------------------------------------------------------------- */
/* front-end continuous collision check + shape kernels  */
/* ------------------------------------------------------------------------- */
/* SweptVolumeManager::checkSubSWCollision SWM:1171-1211: for every obstacle point scan the
 * linearly interpolated pose kt = 0, 0.02, ... (accumulated adds, kt <= 1.0) and fail on the
 * first negative SDF.  Returns 1 (= reference `true`, edge is free) or 0. */
int orc_check_sub_sw_collision(orc_ctx *ctx, const double father[3], const double child[3],
                               const double *pts_xy, size_t n) {
  const double dt = 0.02;
  for (size_t i = 0; i < n; ++i) {
    double min_sdf = 1e9, temp_sdf = 1e8;
    for (double kt = 0.0; kt <= 1.0; kt += dt) {
      double lx = kt * child[0] + (1 - kt) * father[0];
      double ly = kt * child[1] + (1 - kt) * father[1];
      double yaw = kt * child[2] + (1 - kt) * father[2];
      double s = sin(yaw), c = cos(yaw);
      double dx = pts_xy[2 * i] - lx, dy = pts_xy[2 * i + 1] - ly;
      double rx = c * dx + s * dy;
      double ry = (-s) * dx + c * dy;
      temp_sdf = orc_shape_sdf(&ctx->shape, rx, ry);
      if (temp_sdf < min_sdf) min_sdf = temp_sdf;
      if (min_sdf < 0) return 0;
    }
  }
  return 1;
}

/* BasicShape::initShape SHP:386-430 + byteShapeKernel::generateByteKernel SHP:194-216.
 * map_out: kernel_count x ks x ks bools (a-major: map[a*ks+b]); bytes_out: kernel_count x
 * ks*((ks+7)/8) bytes, bit b of row a = or_mask[b%8] (0x80 >> (b%8)) of byte a*bpl + b/8;
 * yaw_out: kernel_count yaws of the accumulated loop `for (yaw=-PI; yaw<PI; yaw+=yaw_res)`.
 * Returns the number of kernels the reference loop would produce (it can exceed kernel_count by
 * one through rounding, which overflows the reference's arrays; only kernel_count are written). */
int orc_shape_kernels(orc_ctx *ctx, int ks, int kernel_count, double resu, double safemargin,
                      unsigned char *map_out, unsigned char *bytes_out, double *yaw_out) {
  static const unsigned char or_mask[8] = {0x80, 0x40, 0x20, 0x10, 0x08, 0x04, 0x02, 0x01}; /* SHP:95 */
  const double PI_ = 3.14159265358979323846; /* SHP:31 */
  int size_side = (int)(0.5 * (ks - 1));
  int bpl = (ks + 7) / 8;
  double yaw_res = 2 * PI_ / kernel_count;
  int ind = 0;
  for (double yaw = -PI_; yaw < PI_; yaw += yaw_res, ind++) {
    if (ind >= kernel_count) continue;
    if (yaw_out) yaw_out[ind] = yaw;
    unsigned char *m = map_out + (size_t)ind * ks * ks;
    unsigned char *bm = bytes_out ? bytes_out + (size_t)ind * ks * bpl : NULL;
    if (bm) memset(bm, 0, (size_t)ks * bpl);
    for (int a = 0; a < ks; ++a)
      for (int b = 0; b < ks; ++b) {
        double x = resu * a - size_side * resu;
        double y = resu * b - size_side * resu;
        double sdf = orc_shape_sdf_rot(&ctx->shape, x, y, yaw);
        unsigned char occ = (sdf <= safemargin) ? 1 : 0;
        m[a * ks + b] = occ;
        if (bm && occ) bm[a * bpl + b / 8] |= or_mask[b % 8];
      }
  }
  return ind;
}
