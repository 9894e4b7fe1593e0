// poly_host.cpp -- test harness: runs the product's Polygon evaluation (csrc/svsdf_polygon.hpp, the very functions the
// gfx950 kernels inline) on the HOST so that the candidate-list logic can be checked bit for bit against the oracle's
// plain loop over all edges without a GPU.  Built by tests/test_polygon_accel.py:
//   hipcc -x hip --cuda-host-only -O2 -ffp-contract=off -shared -fPIC poly_host.cpp -o libpoly_host.so
#include <cstddef>

#include "../../implicit-svsdf-planner_amd/csrc/svsdf_polygon.hpp"

using namespace svsdf;

extern "C" {

// stats_out[0..5]: fine-grid cells, coarse-grid cells, candidates (all cells), largest candidate list, entries in
// long lists, largest slab list
int polyhost_eval(const double *xy, int n, const double *pts, size_t P, double *sdf_out, double *sdfc_out,
                  double *closest_out, long long *stats_out) {
  PolyAccelHost h;
  if (!build_poly_accel(xy, n, h)) return 1;
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

// which list a query uses: 0 fine grid, 1 coarse grid, 2 full loop; and how many edges it visits
int polyhost_visits(const double *xy, int n, const double *pts, size_t P, int *level_out, int *count_out) {
  PolyAccelHost h;
  if (!build_poly_accel(xy, n, h)) return 1;
  for (size_t i = 0; i < P; ++i) {
    int cell = poly_cell(h.hdr.lv[0], pts[2 * i], pts[2 * i + 1]), lvl = 0;
    if (cell < 0) { cell = poly_cell(h.hdr.lv[1], pts[2 * i], pts[2 * i + 1]); lvl = 1; }
    if (cell < 0) { level_out[i] = 2; count_out[i] = n; continue; }
    level_out[i] = lvl;
    count_out[i] = (int)(h.cells[h.hdr.lv[lvl].base + cell].w[0] & 0xffffu);
  }
  return 0;
}

}  // extern "C"
