// svsdf_extras.hip -- C ABI of the rows around the hot path: batched front-end collision check and
// shape byte kernels (SWM:1171-1211, SHP:386-430), query-point producer (PCSmap_manager.cpp:88-210), mesh outline, the
// swept volume's outline and its extrusion (SWM:321-336), the in-repo L-BFGS driver (lbfgs.hpp:290-438).
#include "svsdf_ctx.hpp"
#include "svsdf_lbfgs.hpp"
#include "svsdf_mesh.hpp"
#include "svsdf_contour.hpp"
#include "svsdf_points.hpp"

using namespace svsdf;
using namespace svsdf_impl;

extern "C" {

// ---- front end -----------------------------------------------------------------
int svsdf_check_sub_sw_collision(svsdf_ctx *ctx, size_t n_edges, const double *father_states,
                                 const double *child_states, const size_t *pts_offset, const double *pts_xy,
                                 unsigned char *free_out) {
  if (!ctx || ctx->host_only) return fail(ctx, SVSDF_ERR_NO_DEVICE, "svsdf_check_sub_sw_collision: no device context");
  if (!ctx->subs.empty()) {
    const int r = svsdf_check_sub_sw_collision(ctx->subs[0], n_edges, father_states, child_states, pts_offset, pts_xy, free_out);
    if (r) ctx->err = ctx->subs[0]->err;
    return r;
  }
  if (n_edges == 0) return SVSDF_OK;
  if (!father_states || !child_states || !pts_offset || !free_out)
    return fail(ctx, SVSDF_ERR_INVALID, "svsdf_check_sub_sw_collision: null argument");
  const size_t total = pts_offset[n_edges];
  if (pts_offset[0] != 0 || (total && !pts_xy))
    return fail(ctx, SVSDF_ERR_INVALID, "svsdf_check_sub_sw_collision: bad offsets");
  size_t max_pts = 0;
  for (size_t e = 0; e < n_edges; ++e) {
    if (pts_offset[e + 1] < pts_offset[e]) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_check_sub_sw_collision: offsets not monotone");
    max_pts = std::max(max_pts, pts_offset[e + 1] - pts_offset[e]);
  }
  if (n_edges > 0x7fffffffu || (max_pts + kSubswPoints - 1) / kSubswPoints > 65535u)
    return fail(ctx, SVSDF_ERR_INVALID, "svsdf_check_sub_sw_collision: batch too large");
  // kt = 0, 0.02, ... by accumulated adds while kt <= 1.0 (SWM:1189)
  double kt_tab[kMaxKt];
  int nkt = 0;
  for (double kt = 0.0; kt <= 1.0 && nkt < kMaxKt; kt += 0.02) kt_tab[nkt++] = kt;
  if (total == 0) { std::memset(free_out, 1, n_edges); return SVSDF_OK; }
  HIPCHK(hipSetDevice(ctx->device));
  // one packed upload: [father 3E | child 3E | kt 64 | offsets E+1 (u64) | pts 2T] through a pinned staging buffer
  const size_t need = 6 * n_edges + kMaxKt + (n_edges + 1) + 2 * total;
  if (need > ctx->fe_cap) {
    const size_t cap = need + need / 2;
    int rc = dev_alloc(ctx, &ctx->d_fe, cap);
    if (rc) return rc;
    if (ctx->h_fe) { (void)hipHostFree(ctx->h_fe); ctx->h_fe = nullptr; }
    HIPCHK(hipHostMalloc((void **)&ctx->h_fe, cap * sizeof(double)));
    ctx->fe_cap = cap;
  }
  if (n_edges > ctx->fe_edges_cap) {
    int rc = dev_alloc(ctx, &ctx->d_fe_flag, 2 * n_edges);
    if (rc) return rc;
    ctx->fe_edges_cap = 2 * n_edges;
    ctx->h_fe_flag.resize(2 * n_edges);
  }
  double *h = ctx->h_fe;
  std::memcpy(h, father_states, 3 * n_edges * sizeof(double));
  std::memcpy(h + 3 * n_edges, child_states, 3 * n_edges * sizeof(double));
  std::memcpy(h + 6 * n_edges, kt_tab, kMaxKt * sizeof(double));
  unsigned long long *h_offs = reinterpret_cast<unsigned long long *>(h + 6 * n_edges + kMaxKt);
  for (size_t e = 0; e <= n_edges; ++e) h_offs[e] = pts_offset[e];
  std::memcpy(h + 6 * n_edges + kMaxKt + n_edges + 1, pts_xy, 2 * total * sizeof(double));
  double *d_father = ctx->d_fe, *d_child = d_father + 3 * n_edges, *d_kt = d_child + 3 * n_edges;
  unsigned long long *d_offs = reinterpret_cast<unsigned long long *>(d_kt + kMaxKt);
  double *d_pts = d_kt + kMaxKt + n_edges + 1;
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(ctx->d_fe, h, need * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(ctx->d_fe_flag, 0, n_edges * sizeof(int), st));   // hit flags: 1 = some sdf < 0
  const dim3 grid((unsigned)n_edges, (unsigned)((max_pts + kSubswPoints - 1) / kSubswPoints));
  (void)launch_k_subsw(ctx->cfg.shape_id, grid, st, ctx->sp, d_father, d_child, d_offs, d_pts, d_kt, nkt, ctx->d_fe_flag);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(ctx->h_fe_flag.data(), ctx->d_fe_flag, n_edges * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (size_t e = 0; e < n_edges; ++e) free_out[e] = ctx->h_fe_flag[e] ? 0 : 1;
  return SVSDF_OK;
}

int svsdf_shape_kernels(svsdf_ctx *ctx, int kernel_size, int kernel_count, double kernel_resolution,
                        double safemargin, unsigned char *map_out, unsigned char *bytes_out, double *yaw_out,
                        int *loop_count) {
  if (!ctx || ctx->host_only) return fail(ctx, SVSDF_ERR_NO_DEVICE, "svsdf_shape_kernels: no device context");
  if (!ctx->subs.empty()) {
    const int r = svsdf_shape_kernels(ctx->subs[0], kernel_size, kernel_count, kernel_resolution, safemargin, map_out,
                                      bytes_out, yaw_out, loop_count);
    if (r) ctx->err = ctx->subs[0]->err;
    return r;
  }
  if (kernel_size <= 0 || kernel_count <= 0 || kernel_size > 4096 || kernel_count > 65536 || !map_out)
    return fail(ctx, SVSDF_ERR_INVALID, "svsdf_shape_kernels: bad argument");
  if (ctx->cfg.shape_id == SVSDF_SHAPE_Polygon)
    return fail(ctx, SVSDF_ERR_INVALID,
                "svsdf_shape_kernels: Polygon has no getonlySDF(pos_rel, Matrix3d) in the reference (Shape.hpp:1477)");
  // yaw table: for (yaw = -PI; yaw < PI; yaw += yaw_res) (SHP:400-401); PI macro of SHP:31
  const double PI_ = 3.14159265358979323846;
  const double yaw_res = 2 * PI_ / kernel_count;
  std::vector<double> yaws;
  int ind = 0;
  for (double yaw = -PI_; yaw < PI_; yaw += yaw_res, ind++)
    if (ind < kernel_count) yaws.push_back(yaw);
  if (loop_count) *loop_count = ind;
  const int count = (int)yaws.size();
  const int size_side = (int)(0.5 * (kernel_size - 1));
  const size_t cells = (size_t)kernel_size * kernel_size;
  HIPCHK(hipSetDevice(ctx->device));
  double *d_yaw = nullptr;
  unsigned char *d_map = nullptr;
  HIPCHK(hipMalloc((void **)&d_yaw, count * sizeof(double)));
  if (hipMalloc((void **)&d_map, cells * count) != hipSuccess) {
    (void)hipFree(d_yaw);
    return fail(ctx, SVSDF_ERR_HIP_BASE + (int)hipErrorOutOfMemory, "svsdf_shape_kernels: hipMalloc");
  }
  hipStream_t st = ctx->stream;
  hipError_t e1 = hipMemcpyAsync(d_yaw, yaws.data(), count * sizeof(double), hipMemcpyHostToDevice, st);
  const unsigned grid = (unsigned)((cells * count + kBlock - 1) / kBlock);
  (void)launch_k_shape_kernels(ctx->cfg.shape_id, grid, st, ctx->sp, kernel_size, count, kernel_resolution, size_side,
                               safemargin, d_yaw, d_map);
  hipError_t e2 = hipGetLastError();
  hipError_t e3 = hipMemcpyAsync(map_out, d_map, cells * count, hipMemcpyDeviceToHost, st);
  hipError_t e4 = hipStreamSynchronize(st);
  (void)hipFree(d_yaw);
  (void)hipFree(d_map);
  for (hipError_t e : {e1, e2, e3, e4})
    if (e != hipSuccess) return fail(ctx, SVSDF_ERR_HIP_BASE + (int)e, std::string("svsdf_shape_kernels: ") + hipGetErrorString(e));
  if (yaw_out) std::memcpy(yaw_out, yaws.data(), count * sizeof(double));
  if (bytes_out) {  // byteShapeKernel::generateByteKernel SHP:194-216, or_mask SHP:95
    const int bpl = (kernel_size + 7) / 8;
    std::memset(bytes_out, 0, (size_t)count * kernel_size * bpl);
    for (int k = 0; k < count; ++k)
      for (int a = 0; a < kernel_size; ++a)
        for (int b = 0; b < kernel_size; ++b)
          if (map_out[(size_t)k * cells + (size_t)a * kernel_size + b])
            bytes_out[((size_t)k * kernel_size + a) * bpl + b / 8] |= (unsigned char)(0x80u >> (b % 8));
  }
  return SVSDF_OK;
}

// ---- query-point producer (host) -----------------------------------------------------------------
struct svsdf_map {
  svsdf_host::OccupancyMap m;
};

svsdf_map *svsdf_map_create(const float *xyz, size_t n, double resolution, int sta_threshold) {
  if ((!xyz && n) || !(resolution > 0.0)) return nullptr;
  svsdf_map *mp = new svsdf_map();
  mp->m.build(xyz, n, resolution, sta_threshold);
  return mp;
}
void svsdf_map_destroy(svsdf_map *map) { delete map; }
int svsdf_map_info(const svsdf_map *map, int dims[3], double bmin[3], double bmax[3], size_t *occupied) {
  if (!map) return SVSDF_ERR_INVALID;
  for (int d = 0; d < 3; ++d) {
    if (dims) dims[d] = map->m.dims()[d];
    if (bmin) bmin[d] = map->m.bmin()[d];
    if (bmax) bmax[d] = map->m.bmax()[d];
  }
  if (occupied) *occupied = map->m.occupied_count();
  return SVSDF_OK;
}
int svsdf_map_gather(const svsdf_map *map, const double *centres_xyz, size_t ncentres, const double halfbd[3],
                     double *out_xyz, size_t capacity, size_t *count) {
  if (!map || (!centres_xyz && ncentres) || !halfbd || !count) return SVSDF_ERR_INVALID;
  std::vector<double> pts;
  map->m.gather(centres_xyz, ncentres, halfbd, pts);
  *count = pts.size() / 3;
  if (out_xyz) {
    if (capacity < *count) return SVSDF_ERR_INVALID;
    std::copy(pts.begin(), pts.end(), out_xyz);
  }
  return SVSDF_OK;
}
int svsdf_pcd_read_ascii(const char *path, float *xyz, size_t capacity, size_t *n) {
  if (!path || !n) return SVSDF_ERR_INVALID;
  std::vector<float> v;
  if (!svsdf_host::read_pcd_ascii(path, v)) return SVSDF_ERR_INVALID;
  *n = v.size() / 3;
  if (xyz) {
    if (capacity < *n) return SVSDF_ERR_INVALID;
    std::copy(v.begin(), v.end(), xyz);
  }
  return SVSDF_OK;
}

// ---- mesh shapes (host) ------------------------------------------------------------------------------
static int outline_out(const std::vector<double> &xy, double *xy_out, size_t capacity_verts, size_t *count) {
  *count = xy.size() / 2;
  if (xy_out) {
    if (capacity_verts < *count) return SVSDF_ERR_INVALID;
    std::copy(xy.begin(), xy.end(), xy_out);
  }
  return SVSDF_OK;
}
int svsdf_mesh_outline(const double *V, size_t nv, const int *F, size_t nf, double z0, double *xy_out,
                       size_t capacity_verts, size_t *count, int *loops) {
  if (!V || !F || !count || !std::isfinite(z0)) return SVSDF_ERR_INVALID;
  std::vector<double> xy;
  if (!svsdf_host::mesh_outline(V, nv, F, nf, z0, xy, loops)) return fail(nullptr, SVSDF_ERR_INVALID, "svsdf_mesh_outline: no closed cross-section at z0");
  return outline_out(xy, xy_out, capacity_verts, count);
}
static int section_out(const std::vector<double> &xy, const std::vector<int> &sizes, double *xy_out, size_t capacity_verts,
                       size_t *n_verts, int *loop_sizes, size_t capacity_loops, size_t *n_loops) {
  *n_verts = xy.size() / 2;
  *n_loops = sizes.size();
  if ((xy_out == nullptr) != (loop_sizes == nullptr)) return SVSDF_ERR_INVALID;
  if (xy_out) {
    if (capacity_verts < *n_verts || capacity_loops < *n_loops) return SVSDF_ERR_INVALID;
    std::copy(xy.begin(), xy.end(), xy_out);
    std::copy(sizes.begin(), sizes.end(), loop_sizes);
  }
  return SVSDF_OK;
}
int svsdf_mesh_section(const double *V, size_t nv, const int *F, size_t nf, double z0, double *xy_out, size_t capacity_verts,
                       size_t *n_verts, int *loop_sizes, size_t capacity_loops, size_t *n_loops) {
  if (!V || !F || !n_verts || !n_loops || !std::isfinite(z0)) return SVSDF_ERR_INVALID;
  std::vector<double> xy;
  std::vector<int> sizes;
  if (!svsdf_host::mesh_section(V, nv, F, nf, z0, xy, sizes)) return fail(nullptr, SVSDF_ERR_INVALID, "svsdf_mesh_section: no closed cross-section at z0");
  return section_out(xy, sizes, xy_out, capacity_verts, n_verts, loop_sizes, capacity_loops, n_loops);
}
int svsdf_mesh_section_obj(const char *obj_path, double z0, double *xy_out, size_t capacity_verts, size_t *n_verts,
                           int *loop_sizes, size_t capacity_loops, size_t *n_loops) {
  if (!obj_path || !n_verts || !n_loops || !std::isfinite(z0)) return SVSDF_ERR_INVALID;
  std::vector<double> V, xy;
  std::vector<int> F, sizes;
  if (!svsdf_host::read_obj(obj_path, V, F)) return fail(nullptr, SVSDF_ERR_INVALID, std::string("svsdf_mesh_section_obj: cannot read ") + obj_path);
  if (!svsdf_host::mesh_section(V.data(), V.size() / 3, F.data(), F.size() / 3, z0, xy, sizes))
    return fail(nullptr, SVSDF_ERR_INVALID, "svsdf_mesh_section_obj: no closed cross-section at z0");
  return section_out(xy, sizes, xy_out, capacity_verts, n_verts, loop_sizes, capacity_loops, n_loops);
}
int svsdf_mesh_outline_obj(const char *obj_path, double z0, double *xy_out, size_t capacity_verts, size_t *count,
                           int *loops) {
  if (!obj_path || !count || !std::isfinite(z0)) return SVSDF_ERR_INVALID;
  std::vector<double> V, xy;
  std::vector<int> F;
  if (!svsdf_host::read_obj(obj_path, V, F)) return fail(nullptr, SVSDF_ERR_INVALID, std::string("svsdf_mesh_outline_obj: cannot read ") + obj_path);
  if (!svsdf_host::mesh_outline(V.data(), V.size() / 3, F.data(), F.size() / 3, z0, xy, loops))
    return fail(nullptr, SVSDF_ERR_INVALID, "svsdf_mesh_outline_obj: no closed cross-section at z0");
  return outline_out(xy, xy_out, capacity_verts, count);
}

// ---- swept-volume outline (what sw_calculate.cpp / SWM:321-336 produce for visual validation) ----------
int svsdf_swept_outline(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, double cell, double margin,
                        double *xy_out, size_t capacity_verts, size_t *n_verts, int *loop_sizes, size_t capacity_loops,
                        size_t *n_loops, svsdf_outline_stats *stats_out) {
  if (!ctx || !coeffs || !T || !n_verts || !n_loops) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_swept_outline: null argument");
  if (N < 1 || N > kMaxPieces || !(cell > 0.0) || !std::isfinite(cell) || !std::isfinite(margin))
    return fail(ctx, SVSDF_ERR_INVALID, "svsdf_swept_outline: N, cell or margin out of range");
  if ((xy_out == nullptr) != (loop_sizes == nullptr))
    return fail(ctx, SVSDF_ERR_INVALID, "svsdf_swept_outline: xy_out and loop_sizes must both be given (fill) or both be NULL (size query)");
  const svsdf_ctx *base = ctx->subs.empty() ? ctx : ctx->subs[0];
  if (base->host_only) return fail(ctx, SVSDF_ERR_NO_DEVICE, "host-only context: no device entry points");
  // the size query and the fill that follows it carry the same arguments: the second call copies the first one's result
  std::vector<double> key;
  key.reserve(19 * (size_t)N + 3);
  key.push_back((double)N); key.push_back(cell); key.push_back(margin);
  key.insert(key.end(), coeffs, coeffs + 18 * (size_t)N);
  key.insert(key.end(), T, T + N);
  auto deliver = [&](const std::vector<double> &xy, const std::vector<int> &loops, const svsdf_outline_stats &st) -> int {
    *n_verts = xy.size() / 2;
    *n_loops = loops.size();
    if (stats_out) *stats_out = st;
    if (xy_out && loop_sizes) {
      if (capacity_verts < xy.size() / 2 || capacity_loops < loops.size())
        return fail(ctx, SVSDF_ERR_INVALID, "svsdf_swept_outline: output capacity too small (query with xy_out = NULL first)");
      std::copy(xy.begin(), xy.end(), xy_out);
      std::copy(loops.begin(), loops.end(), loop_sizes);
    }
    return SVSDF_OK;
  };
  if (ctx->ol_valid && ctx->ol_key.size() == key.size() && std::memcmp(ctx->ol_key.data(), key.data(), key.size() * sizeof(double)) == 0)
    return deliver(ctx->ol_xy, ctx->ol_loops, ctx->ol_stats);
  // bounding box of the path (body origin), grown by the shape's bound radius: the swept volume lies inside
  double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
  for (int i = 0; i < N; ++i) {
    if (!(T[i] > 0.0) || !std::isfinite(T[i])) return fail(ctx, SVSDF_ERR_NONFINITE, "svsdf_swept_outline: bad duration");
    for (int q = 0; q <= 32; ++q) {
      const double sl = T[i] * (double)q / 32.0;
      for (int d = 0; d < 2; ++d) {
        double v = 0.0;
        for (int k = 5; k >= 0; --k) v = v * sl + coeffs[(size_t)d * 6 * N + (size_t)i * 6 + k];
        if (!std::isfinite(v)) return fail(ctx, SVSDF_ERR_NONFINITE, "svsdf_swept_outline: non-finite trajectory");
        lo[d] = std::min(lo[d], v); hi[d] = std::max(hi[d], v);
      }
    }
  }
  // (a quintic between samples 1/32 of a piece apart can leave the sampled box by a little: one more bound radius)
  const double grow = 2.0 * base->r_bound + std::max(margin, 0.0) + 4.0 * cell;
  svsdf_host::ContourGrid g;
  g.h = cell;
  g.levels = 4;
  g.x0 = lo[0] - grow; g.y0 = lo[1] - grow;
  const double wx = (hi[0] - lo[0]) + 2.0 * grow, wy = (hi[1] - lo[1]) + 2.0 * grow;
  if (wx / cell > 1e6 || wy / cell > 1e6) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_swept_outline: more than 1e6 cells per side");
  g.nx = (long long)std::ceil(wx / cell); g.ny = (long long)std::ceil(wy / cell);
  // a private single-device context with the same shape and weights: the caller's resident cloud stays as it is
  svsdf_config c = base->cfg;
  c.n_devices = 0; c.rank = 0; c.world_size = 1; c.combine = SVSDF_COMBINE_AUTO; c.device = base->device;
  c.polygon_xy = base->poly_xy.empty() ? nullptr : base->poly_xy.data();
  c.polygon_nverts = (int)(base->poly_xy.size() / 2);
  c.polygon_loop_sizes = base->poly_loops.empty() ? nullptr : base->poly_loops.data();
  c.polygon_nloops = (int)base->poly_loops.size();
  svsdf_ctx *tmp = svsdf_create(&c);
  if (!tmp) return fail(ctx, SVSDF_ERR_INVALID, std::string("svsdf_swept_outline: ") + svsdf_last_error_string(nullptr));
  std::vector<double> xyz, sdf;
  std::vector<long long> idx;
  const svsdf_host::FieldEval eval = [&](const std::vector<double> &xy, std::vector<double> &val) -> int {
    const size_t P = xy.size() / 2;
    xyz.resize(3 * P);
    for (size_t k = 0; k < P; ++k) { xyz[3 * k] = xy[2 * k]; xyz[3 * k + 1] = xy[2 * k + 1]; xyz[3 * k + 2] = 0.0; }
    int rc = svsdf_set_points(tmp, xyz.data(), P);
    if (rc) return rc;
    if (svsdf_num_points(tmp) != P) return SVSDF_ERR_INVALID;
    sdf.resize(P); idx.resize(P);
    rc = swept_field(tmp, N, coeffs, T, sdf.data());
    if (rc) return rc;
    rc = svsdf_shard_indices(tmp, idx.data());
    if (rc) return rc;
    val.assign(P, 0.0);
    for (size_t k = 0; k < P; ++k) {
      if (!std::isfinite(sdf[k])) return SVSDF_ERR_NONFINITE;   // (a NaN would silently read as "outside")
      val[(size_t)idx[k]] = sdf[k];
    }
    return 0;
  };
  std::vector<double> xy;
  std::vector<int> loops;
  svsdf_host::ContourStats st;
  const int rc = svsdf_host::swept_contour(g, eval, 1.5, xy, loops, &st);
  const std::string tmp_err = rc ? svsdf_last_error_string(tmp) : "";
  svsdf_destroy(tmp);
  if (rc) return fail(ctx, rc > 0 ? rc : SVSDF_ERR_INVALID, "svsdf_swept_outline: evaluation failed: " + tmp_err);
  svsdf_outline_stats so{};
  so.nodes_evaluated = st.nodes_evaluated; so.dense_nodes = st.dense_nodes;
  so.cells_marched = st.cells_marched; so.batches = st.batches; so.open_chains = st.open_chains;
  ctx->ol_key.swap(key);
  ctx->ol_xy = xy;
  ctx->ol_loops = loops;
  ctx->ol_stats = so;
  ctx->ol_valid = true;
  return deliver(ctx->ol_xy, ctx->ol_loops, ctx->ol_stats);
}

int svsdf_outline_extrude(const double *xy, const int *loop_sizes, size_t n_loops, double z0, double z1, int caps,
                          double *V_out, size_t capacity_verts, size_t *n_verts, int *F_out, size_t capacity_tris,
                          size_t *n_tris) {
  if (!n_verts || !n_tris || !std::isfinite(z0) || !std::isfinite(z1)) return SVSDF_ERR_INVALID;
  if (n_loops == 0) { *n_verts = 0; *n_tris = 0; return SVSDF_OK; }   // an empty outline extrudes to an empty surface
  if (!xy || !loop_sizes) return SVSDF_ERR_INVALID;
  for (size_t l = 0; l < n_loops; ++l)
    if (loop_sizes[l] < 3) return SVSDF_ERR_INVALID;
  std::vector<double> V;
  std::vector<int> F;
  svsdf_host::extrude_outline(xy, loop_sizes, n_loops, z0, z1, caps != 0, V, F);
  *n_verts = V.size() / 3;
  *n_tris = F.size() / 3;
  if (!V_out || !F_out) return SVSDF_OK;
  if (capacity_verts < V.size() / 3 || capacity_tris < F.size() / 3) return SVSDF_ERR_INVALID;
  std::copy(V.begin(), V.end(), V_out);
  std::copy(F.begin(), F.end(), F_out);
  return SVSDF_OK;
}

}  // extern "C"

// ---- optimizer driver (host) ---------------------------------------------------
void svsdf_lbfgs_params_default(svsdf_lbfgs_params *p) {
  if (!p) return;
  p->mem_size = 8; p->g_epsilon = 1.0e-5; p->past = 3; p->delta = 1.0e-6; p->max_iterations = 0;
  p->max_linesearch = 64; p->min_step = 1.0e-20; p->max_step = 1.0e+20; p->f_dec_coeff = 1.0e-4;
  p->s_curv_coeff = 0.9; p->cautious_factor = 1.0e-6; p->machine_prec = 1.0e-16;
}

int svsdf_lbfgs_minimize(int n, double *x, svsdf_evaluate_t eval, void *instance, svsdf_progress_t progress,
                         void *progress_user, const svsdf_lbfgs_params *params, double *final_cost,
                         int *iterations, int *evaluations) {
  if (!x || !eval) return SVSDF_LBFGSERR_INVALIDPARAMETERS;
  svsdf_lbfgs_params p;
  if (params) p = *params; else svsdf_lbfgs_params_default(&p);
  const svsdf_host::LbfgsResult r = svsdf_host::lbfgs_minimize(n, x, eval, instance, progress, progress_user, p);
  if (final_cost) *final_cost = r.fx;
  if (iterations) *iterations = r.iterations;
  if (evaluations) *evaluations = r.evaluations;
  return r.status;
}

int svsdf_optimize_traj(svsdf_ctx *ctx, double *x, int n, const svsdf_lbfgs_params *params,
                        svsdf_progress_t progress, void *progress_user, double *final_cost, int *iterations,
                        int *evaluations) {
  if (!ctx || !x || n < 1 || (n + 3) % 4 != 0) {
    fail(ctx, SVSDF_ERR_INVALID, "svsdf_optimize_traj: n must be N + 3(N-1)");
    return SVSDF_LBFGSERR_INVALIDPARAMETERS;
  }
  // As in the reference, the side outputs (svsdf_last_costs, MINCO state) are those of the LAST callback
  // evaluation -- after a failed line search that is a trial point, not the returned x -- and the objective is
  // history dependent once a trial's total duration reaches 300 s (stale traj_duration, sw_manager.hpp:380-384):
  // the value reported is the one the driver accepted, no re-evaluation is made here.
  const int rc = svsdf_lbfgs_minimize(n, x, svsdf_lmbm_evaluate, ctx, progress, progress_user, params, final_cost,
                                      iterations, evaluations);
  return rc;
}
