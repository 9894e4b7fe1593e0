// poly_host.cpp -- test harness: runs the product's Polygon evaluation (csrc/svsdf_polygon.hpp, the very functions the
// gfx950 kernels inline) on the HOST so that the candidate-list logic can be checked bit for bit against the oracle's
// plain loop over all edges without a GPU.  Built by tests/test_polygon_accel.py:
//   hipcc -x hip --cuda-host-only -O2 -ffp-contract=off -shared -fPIC poly_host.cpp -o libpoly_host.so
#include <cstddef>

#include "../../implicit-svsdf-planner_amd/csrc/svsdf_polygon.hpp"

using namespace svsdf;

#include <chrono>
#include <cstring>

static int g_refine = SVSDF_POLY_REFINE;
static std::vector<int> g_loops;   // loop sizes of the outlines given to the calls below (empty: one loop)

static bool build(const double *xy, int n, PolyAccelHost &h) {
  return build_poly_accel(xy, n, h, 128, 256, 256, g_refine, 32, g_loops.empty() ? nullptr : g_loops.data(), (int)g_loops.size());
}

extern "C" {

// outline of several closed loops (sizes add up to n) for the calls below; nloops = 0: back to one loop
void polyhost_set_loops(const int *sizes, int nloops) { g_loops.assign(sizes, sizes + (nloops > 0 ? nloops : 0)); }

// FNV-1a over everything build_poly_accel produces (edges, cell / slab records, long lists, the header's numbers) and the
// wall time of the build
int polyhost_build_hash(const double *xy, int n, unsigned long long *hash_out, double *ms_out, long long *sizes_out) {
  PolyAccelHost h;
  const auto t0 = std::chrono::steady_clock::now();
  if (!build(xy, n, h)) return 1;
  *ms_out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  unsigned long long f = 1469598103934665603ull;
  auto add = [&](const void *p, size_t bytes) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < bytes; ++i) { f ^= b[i]; f *= 1099511628211ull; }
  };
  add(h.edges.data(), h.edges.size() * sizeof(PolyEdge));
  add(h.cells.data(), h.cells.size() * sizeof(PolyRec));
  add(h.slabs.data(), h.slabs.size() * sizeof(PolyRec));
  add(h.over.data(), h.over.size() * sizeof(unsigned short));
  PolyAccel hdr = h.hdr;
  hdr.edges = nullptr; hdr.cells = nullptr; hdr.slabs = nullptr; hdr.over = nullptr;
  add(&hdr, sizeof hdr);
  *hash_out = f;
  if (sizes_out) { sizes_out[0] = (long long)h.edges.size(); sizes_out[1] = (long long)h.cells.size(); sizes_out[2] = (long long)h.over.size(); sizes_out[3] = (long long)h.cand_total; }
  return 0;
}

// second-pass resolution of the candidate lists for the calls below (1 = first pass only)
void polyhost_set_refine(int r) { g_refine = r; }

// stats_out[0..5]: fine-grid cells, coarse-grid cells, candidates (all cells), largest candidate list, entries in
// long lists, largest slab list
int polyhost_eval(const double *xy, int n, const double *pts, size_t P, double *sdf_out, double *sdfc_out,
                  double *closest_out, long long *stats_out) {
  PolyAccelHost h;
  if (!build(xy, n, h)) return 1;
  PolyAccel pa = h.hdr;
  pa.edges = h.edges.data();
  pa.cells = h.cells.data();
  pa.slabs = h.slabs.data();
  pa.over = h.over.data();
  for (size_t i = 0; i < P; ++i) {
    const double x = pts[2 * i], y = pts[2 * i + 1];
    if (sdf_out) sdf_out[i] = poly_sdf<false>(pa, pa.edges, x, y, nullptr, nullptr);
    if (sdfc_out) sdfc_out[i] = poly_sdf<true>(pa, pa.edges, x, y, closest_out + 2 * i, closest_out + 2 * i + 1);
  }
  if (stats_out) {
    stats_out[0] = (long long)pa.lv[0].nx * pa.lv[0].ny;
    stats_out[1] = (long long)pa.lv[1].nx * pa.lv[1].ny;
    stats_out[2] = (long long)h.cand_total;
    stats_out[3] = (long long)h.cand_max;
    stats_out[4] = (long long)h.over.size();
    stats_out[5] = (long long)h.slab_max;
  }
  return 0;
}

// which list a query uses: 0 fine grid, 1 coarse grid, 2 far grid, 3 full loop; and how many edges it visits
int polyhost_visits(const double *xy, int n, const double *pts, size_t P, int *level_out, int *count_out) {
  PolyAccelHost h;
  if (!build(xy, n, h)) return 1;
  for (size_t i = 0; i < P; ++i) {
    unsigned base = 0;
    const int cell = poly_locate(h.hdr, pts[2 * i], pts[2 * i + 1], base);
    if (cell < 0) { level_out[i] = 3; count_out[i] = n; continue; }
    const int lvl = (base == h.hdr.lv[0].base) ? 0 : (base == h.hdr.lv[1].base) ? 1 : 2;
    level_out[i] = lvl;
    { const unsigned w7 = h.cells[h.hdr.lv[lvl].base + cell].w[7]; count_out[i] = (int)((w7 >> 30) == 3u ? ((w7 >> 16) & 0xfu) : ((w7 >> 16) & kPolyCountMask)); }
  }
  return 0;
}

// how many edges the crossing-parity loop of a query visits (0 when no edge can cross its ray)
int polyhost_parity_visits(const double *xy, int n, const double *pts, size_t P, int *count_out) {
  PolyAccelHost h;
  if (!build(xy, n, h)) return 1;
  const PolyAccel &pa = h.hdr;
  for (size_t i = 0; i < P; ++i) {
    const double x = pts[2 * i], y = pts[2 * i + 1];
    unsigned base = 0;
    const int cell = poly_locate(pa, x, y, base);
    const bool ray = y >= pa.ymin - pa.tol && y <= pa.ymax + pa.tol && x <= pa.xmax + pa.tol;
    const double fs = (y - pa.ymin) * pa.slab_inv_h;
    const int slab = !(fs >= 0.0) ? 0 : (fs >= (double)pa.nslab) ? pa.nslab - 1 : (int)fs;
    const double fx = (x - pa.xmin) * pa.xb_inv_h;
    const int xb = !(fx >= 0.0) ? 0 : (fx >= (double)pa.nxb) ? pa.nxb - 1 : (int)fx;
    count_out[2 * i] = ray ? (int)((h.slabs[slab * pa.nxb].w[7] >> 16) & kPolyCountMask) : 0;   // the whole slab
    const unsigned w7 = (cell >= 0) ? h.cells[base + cell].w[7] : 0u;
    const unsigned pstate = w7 >> 30;
    count_out[2 * i + 1] = (pstate == 3u) ? (int)((w7 >> 20) & 0xfu)
                         : (ray && pstate == 0u) ? (int)((h.slabs[slab * pa.nxb + xb].w[7] >> 16) & kPolyCountMask) : 0;   // as evaluated
  }
  return 0;
}

// poly_quot (through the same range test as poly_sdf) vs the division: q_out = what dis2Seg's t is computed from,
// used_out = 1 where the refinement (not the division) produced it
void polyhost_quot(const double *a, const double *b, size_t m, double *q_out, int *used_out) {
  for (size_t i = 0; i < m; ++i) {
    bool inrange;
    const double q = poly_quot(a[i], b[i], (b[i] > 0.0) ? 1.0 / b[i] : 0.0, inrange);
    const bool ok = inrange && b[i] >= 1e-100 && b[i] <= 1e100;   // PolyAccel::div_ok
    q_out[i] = ok ? q : a[i] / b[i];
    used_out[i] = ok ? 1 : 0;
  }
}

// the grid cell poly_locate gives a query, as the rectangle the host built that cell's lists for (before its enlargement):
// rect_out[4 i ..] = x0, y0, x1, y1 (NaN when the query is outside all levels); level_out: 0 fine, 1 coarse, 2 far, 3 none
int polyhost_cell_rect(const double *xy, int n, const double *pts, size_t P, double *rect_out, int *level_out) {
  PolyAccelHost h;
  if (!build(xy, n, h)) return 1;
  for (size_t i = 0; i < P; ++i) {
    unsigned base = 0;
    const int cell = poly_locate(h.hdr, pts[2 * i], pts[2 * i + 1], base);
    if (cell < 0) { level_out[i] = 3; for (int k = 0; k < 4; ++k) rect_out[4 * i + k] = std::nan(""); continue; }
    const int l = (base == h.hdr.lv[0].base) ? 0 : (base == h.hdr.lv[1].base) ? 1 : 2;
    const PolyLevel &lv = h.hdr.lv[l];
    const int ix = cell % lv.nx, iy = cell / lv.nx;
    const double hc = 1.0 / lv.inv_h;
    level_out[i] = l;
    rect_out[4 * i] = lv.x0 + ix * hc; rect_out[4 * i + 1] = lv.y0 + iy * hc;
    rect_out[4 * i + 2] = lv.x0 + (ix + 1) * hc; rect_out[4 * i + 3] = lv.y0 + (iy + 1) * hc;
  }
  return 0;
}

}  // extern "C"
