// svsdf_capi.hip -- the C ABI of include/svsdf_c.h for the hot path: context creation (shape constants the reference
// evaluates with libm at construction, SHP:281-294 / :855 / :1237 / :1278 / :1320), resident points, the inner operator
// addSaftyPenaOnSweptVolumeParallelTrueSDF (BEO:774-869), the full callback costFunctionLmbmParallel (BEO:344-408), launch
// plan and statistics.  No CPU fallback: without a HIP device every compute entry fails.
#include "svsdf_ctx.hpp"

using namespace svsdf;
using namespace svsdf_impl;

namespace {

// Circumradius about the shape-local origin of each registered shape, from the constants of its SDF
// (csrc/svsdf_shapes.hpp, i.e. Shape.hpp:531-1476).  All these SDFs are exact distance functions, so
// sdf(q) >= |q| - R holds for every q with R = this radius (+ the shape offset).
double shape_circumradius(int shape_id, const double *poly_xy, int nverts) {
  switch (shape_id) {
    case SVSDF_SHAPE_sdUnevenCapsule: return 6.0;                              // cap r2 = 1 centred at (0, h = 5)
    case SVSDF_SHAPE_sdCutDisk: return 5.0;                                    // disk radius r
    case SVSDF_SHAPE_sdTrapezoid: return std::sqrt(3.0 * 3.0 + 2.0 * 2.0);     // corner (r2, he)
    case SVSDF_SHAPE_sdRhombus: return 4.5;                                    // vertex (0, b.y)
    case SVSDF_SHAPE_star: return 2.8;                                         // outer tips at r
    case SVSDF_SHAPE_sdTunnel: return std::sqrt(2.5 * 2.5 + 1.5 * 1.5);        // box corner (wh.x, wh.y) vs arch radius wh.x
    case SVSDF_SHAPE_sdHorseshoe: return std::hypot(1.5 + 0.20, 1.55);         // far corner of a leg: (r + w.y, w.x)
    case SVSDF_SHAPE_sdHeart: return 4.0 * (std::sqrt(0.25 * 0.25 + 0.75 * 0.75) + std::sqrt(2.0) / 4.0);  // lobe circle
    case SVSDF_SHAPE_sdOrientedVesica: return std::sqrt(2.0 * 2.0 + 4.0 * 4.0);  // tips a, b
    case SVSDF_SHAPE_sdRoundedCross: return 2.0;                               // tips (1, 0), (0, h) scaled by 2
    case SVSDF_SHAPE_sdRoundedX: return 3.0 / std::sqrt(2.0) + 0.25;           // arm end (w/2, w/2) + r
    case SVSDF_SHAPE_bigX: return 5.0 / std::sqrt(2.0) + 0.25;
    case SVSDF_SHAPE_sdMoon: return 3.0;                                       // outer disk ra
    case SVSDF_SHAPE_sdPie: return 3.0;
    case SVSDF_SHAPE_sdPie2: return 3.0;
    case SVSDF_SHAPE_sdArc: return 2.3333 + 0.5;                               // ra + rb
    default: {
      double r = 0.0;
      for (int i = 0; i < nverts; ++i) r = std::max(r, std::hypot(poly_xy[2 * i], poly_xy[2 * i + 1]));
      return r;
    }
  }
}

}  // namespace

extern "C" {


void svsdf_config_default(svsdf_config *cfg) {
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->shape_id = SVSDF_SHAPE_star;
  cfg->safety_hor = 0.7;  // src/plan_manager/config/star.yaml
  cfg->weight_p = 60.0;
  cfg->rho = 3.8;
  cfg->device = -1;
  cfg->rank = 0;
  cfg->world_size = 1;
}

int svsdf_shape_id_from_inputdata(const char *inputdata) {
  if (!inputdata) return SVSDF_SHAPE_Polygon;
  std::string s(inputdata);
  const size_t start = s.find_last_of('/') + 1;  // npos + 1 == 0
  const size_t end = s.find_last_of('.');
  const std::string stem = s.substr(start, end == std::string::npos ? std::string::npos : end - start);
  for (int i = 0; i < SVSDF_SHAPE_Polygon; ++i)
    if (stem == kShapeNames[i]) return i;
  return SVSDF_SHAPE_Polygon;
}

const char *svsdf_shape_name(int id) { return (id >= 0 && id < SVSDF_SHAPE_COUNT) ? kShapeNames[id] : "?"; }

const char *svsdf_last_error_string(const svsdf_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

svsdf_ctx *svsdf_create(const svsdf_config *cfg) {
  if (!cfg || cfg->shape_id < 0 || cfg->shape_id >= SVSDF_SHAPE_COUNT || cfg->world_size < 1 ||
      cfg->rank < 0 || cfg->rank >= cfg->world_size) {
    g_last_error = "svsdf_create: invalid config";
    return nullptr;
  }
  if (cfg->n_devices < 0 || cfg->n_devices > SVSDF_MAX_DEVICES || cfg->combine < 0 || cfg->combine > SVSDF_COMBINE_RCCL) {
    g_last_error = "svsdf_create: n_devices out of range [0, 8] or unknown combine mode";
    return nullptr;
  }
  if (cfg->flags & SVSDF_FLAG_HOST_ONLY) {
    svsdf_ctx *h = new svsdf_ctx();
    h->cfg = *cfg;
    h->cfg.polygon_xy = nullptr;
    h->cfg.polygon_loop_sizes = nullptr;
    h->host_only = true;
    return h;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_last_error = "svsdf_create: no HIP device (this library has no CPU fallback)";
    return nullptr;
  }
  // (one device + RCCL combine is accepted too: a 1-rank communicator, for measuring the collective's fixed cost)
  if (cfg->n_devices >= 2 || (cfg->n_devices == 1 && cfg->combine == SVSDF_COMBINE_RCCL)) return create_group(cfg, ndev);
  svsdf_ctx *ctx = new svsdf_ctx();
  ctx->cfg = *cfg;
  ctx->cfg.polygon_xy = nullptr;
  ctx->cfg.polygon_loop_sizes = nullptr;
  int dev = cfg->device;
  if (dev < 0) (void)hipGetDevice(&dev);
  ctx->device = dev;
  auto bail = [&](const std::string &m) -> svsdf_ctx * {
    g_last_error = m;
    svsdf_destroy(ctx);
    return nullptr;
  };
  if (hipSetDevice(dev) != hipSuccess) return bail("hipSetDevice failed");
  {
    StreamSet ss;   // from the process-wide pool: created once per device, never destroyed (svsdf_ctx.hpp)
    if (!acquire_streams(dev, ss)) return bail("hipStreamCreate failed");
    ctx->stream = ss.main;
    for (int b = 0; b < kMaxBatches; ++b) ctx->bstream[b] = ss.batch[b];
    ctx->stream_slot = ss.slot;
  }
  // shape constants, evaluated with the host libm exactly where the reference does (SHP:281-294,
  // :855, :1237, :1278, :1320)
  ShapeParams &sp = ctx->sp;
  sp.tx = cfg->poly_params[0];
  sp.ty = cfg->poly_params[1];
  const double yaw = cfg->poly_params[2] * kPI / 180.0;
  // Rotate (SHP:287-292) is written with std::cos(yaw) / std::sin(yaw); g++ -O3 -- the reference's build -- merges such a
  // pair into ONE call of sincos(), and glibc's sincos is not bit-identical with its cos / sin (yaw = -72.42 deg: cos one
  // ulp apart).  So: sincos, explicitly (the oracle does the same; found by the device-arithmetic fuzz, round 4).
  {
    double sn_ = 0.0, cs_ = 1.0;
    ::sincos(yaw, &sn_, &cs_);
    sp.r00 = cs_; sp.r01 = -sn_; sp.r10 = sn_; sp.r11 = cs_;
  }
  switch (cfg->shape_id) {
    case SVSDF_SHAPE_sdHorseshoe: sp.c0x = std::cos(20.5); sp.c0y = std::sin(20.5); break;
    case SVSDF_SHAPE_sdPie: sp.c0x = std::cos(43.0); sp.c0y = std::sin(43.0); break;
    case SVSDF_SHAPE_sdPie2: sp.c0x = std::cos(1.0); sp.c0y = std::sin(1.0); break;
    case SVSDF_SHAPE_sdArc: sp.c0x = std::sin(20.0); sp.c0y = std::cos(20.0); break;
    default: sp.c0x = 0.0; sp.c0y = 0.0; break;
  }
  sp.r_bound = 0.0;
  sp.identity = (sp.tx == 0.0 && sp.ty == 0.0 && sp.r00 == 1.0 && sp.r01 == 0.0 && sp.r10 == 0.0 && sp.r11 == 1.0) ? 1 : 0;
  sp.nverts = 0;
  sp.accel = nullptr;
  sp.edges = nullptr;
  if (cfg->shape_id == SVSDF_SHAPE_Polygon) {
    std::vector<double> &v = ctx->poly_xy;
    std::vector<int> &loops = ctx->poly_loops;
    loops.clear();
    if (cfg->polygon_xy && cfg->polygon_nverts >= 3) {
      const int nl = (cfg->polygon_loop_sizes && cfg->polygon_nloops >= 2) ? cfg->polygon_nloops : 0;
      if ((long long)cfg->polygon_nverts + nl > SVSDF_MAX_POLY_VERTS)
        return bail("svsdf_create: polygon_nverts (+ one entry per loop) exceeds SVSDF_MAX_POLY_VERTS (" + std::to_string(SVSDF_MAX_POLY_VERTS) + ")");
      v.assign(cfg->polygon_xy, cfg->polygon_xy + 2 * (size_t)cfg->polygon_nverts);
      if (nl) {
        long long tot = 0;
        for (int k = 0; k < nl; ++k) { if (cfg->polygon_loop_sizes[k] < 3) return bail("svsdf_create: a polygon loop needs at least 3 vertices"); tot += cfg->polygon_loop_sizes[k]; }
        if (tot != cfg->polygon_nverts) return bail("svsdf_create: polygon_loop_sizes do not add up to polygon_nverts");
        loops.assign(cfg->polygon_loop_sizes, cfg->polygon_loop_sizes + nl);
      }
    } else {
      v = {6, -0.1, 6, 0.1, -6, 0.1, -6, -0.1};  // SWM:363-369
    }
    // candidate lists of the outline (svsdf_polygon.hpp), then one upload: the header's pointers are device addresses
    PolyAccelHost pa;
    const int ngf = 128, ngc = 256;   // grid cells per side, fine / coarse
    if (!build_poly_accel(v.data(), (int)(v.size() / 2), pa, ngf, ngc, 256, SVSDF_POLY_REFINE, 32, loops.empty() ? nullptr : loops.data(), (int)loops.size()))
      return bail("svsdf_create: polygon outline rejected (non-finite vertex?)");
    auto align = [](size_t o) { return (o + 63) & ~(size_t)63; };
    const size_t o_edges = align(sizeof(PolyAccel));
    const size_t o_cell = align(o_edges + pa.edges.size() * sizeof(PolyEdge));
    const size_t o_slab = align(o_cell + pa.cells.size() * sizeof(PolyRec));
    const size_t o_over = align(o_slab + pa.slabs.size() * sizeof(PolyRec));
    const size_t total = align(o_over + pa.over.size() * sizeof(unsigned short));
    if (hipMalloc((void **)&ctx->d_poly, total) != hipSuccess) return bail("hipMalloc polygon failed");
    std::vector<unsigned char> blob(total, 0);
    pa.hdr.edges = reinterpret_cast<const PolyEdge *>(ctx->d_poly + o_edges);
    pa.hdr.cells = reinterpret_cast<const PolyRec *>(ctx->d_poly + o_cell);
    pa.hdr.slabs = reinterpret_cast<const PolyRec *>(ctx->d_poly + o_slab);
    pa.hdr.over = reinterpret_cast<const unsigned short *>(ctx->d_poly + o_over);
    std::memcpy(blob.data(), &pa.hdr, sizeof(PolyAccel));
    std::memcpy(blob.data() + o_edges, pa.edges.data(), pa.edges.size() * sizeof(PolyEdge));
    std::memcpy(blob.data() + o_cell, pa.cells.data(), pa.cells.size() * sizeof(PolyRec));
    std::memcpy(blob.data() + o_slab, pa.slabs.data(), pa.slabs.size() * sizeof(PolyRec));
    if (!pa.over.empty()) std::memcpy(blob.data() + o_over, pa.over.data(), pa.over.size() * sizeof(unsigned short));
    if (hipMemcpy(ctx->d_poly, blob.data(), total, hipMemcpyHostToDevice) != hipSuccess)
      return bail("hipMemcpy polygon failed");
    sp.nverts = (int)pa.edges.size();   // entries of the edge array: the vertices + one closing copy per loop of a multi-loop outline
    sp.accel = reinterpret_cast<const PolyAccel *>(ctx->d_poly);
    sp.edges = pa.hdr.edges;
    // the solve / round kernels keep outlines of up to 1024 edges (48 KB) in LDS in front of the pose table
    ctx->poly_lds = sp.nverts <= kPolyLdsMaxVerts;
    if (const char *e = std::getenv("SVSDF_POLY_LDS")) ctx->poly_lds = ctx->poly_lds && std::atoi(e) != 0;
    ctx->cfg.polygon_nverts = (int)(v.size() / 2);
    ctx->cfg.polygon_nloops = (int)loops.size();
  }
  ctx->G_env = 0;
  if (const char *e = std::getenv("SVSDF_G")) { const int g = std::atoi(e); if (g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 32) { ctx->G_env = g; ctx->G = g; ctx->G_late = std::max(g, 8); } }
  if (const char *e = std::getenv("SVSDF_G_LATE")) { const int g = std::atoi(e); if (g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 32) { ctx->G_late = g; ctx->G_late_env = g; } }
  if (const char *e = std::getenv("SVSDF_PRUNE")) ctx->prune = std::atoi(e) != 0;
  if (const char *e = std::getenv("SVSDF_BATCHES")) ctx->want_batches = (std::string(e) == "measure") ? -1 : std::max(0, std::min(std::atoi(e), (int)kMaxBatches));
  if (const char *e = std::getenv("SVSDF_SELECT_DELTA")) { ctx->select_delta = std::atof(e); ctx->select_env = true; }
  {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && n > 0) ctx->n_cu = n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device) == hipSuccess && n > 0) ctx->lds_limit = (size_t)n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeWallClockRate, ctx->device) == hipSuccess && n > 0) ctx->wall_clock_khz = (double)n;
  }
  if (const char *e = std::getenv("SVSDF_CULL")) { ctx->cull = std::atoi(e) != 0; ctx->cull2 = std::atoi(e) >= 2; }   // 0 none, 1 circle bound only, 2 both (default)
  if (const char *e = std::getenv("SVSDF_ROUND_BPC")) ctx->round_blocks_per_cu = std::max(1, std::min(std::atoi(e), 16));
  if (const char *e = std::getenv("SVSDF_SCAN_ANCHORS")) ctx->scan_anchors = std::atoi(e) != 0;
  if (const char *e = std::getenv("SVSDF_TAIL_DUO")) ctx->tail_duo = std::atoi(e) != 0;
  if (const char *e = std::getenv("SVSDF_TAIL_LOCAL")) ctx->tail_local = std::atoi(e) != 0;
  if (const char *e = std::getenv("SVSDF_TAIL_LATENCY")) ctx->tail_latency = std::atoi(e) != 0;
  if (const char *e = std::getenv("SVSDF_TAIL")) ctx->tail_mode = (std::string(e) == "off") ? -2 : (std::string(e) == "auto") ? -1 : std::max(0, std::atoi(e));
  if (const char *e = std::getenv("SVSDF_UB_FULL")) { ctx->ub_full = std::atoi(e) != 0; ctx->ub_lazy = std::atoi(e) == 2; ctx->ub_anchor = std::atoi(e) == 3; ctx->ub_env = true; }
  if (const char *e = std::getenv("SVSDF_PROFILE")) ctx->profile = std::atoi(e) != 0;
  if (const char *e = std::getenv("SVSDF_PIECE_TIME")) {   // exact | fast | auto (default)
    ctx->cfg.flags &= ~(SVSDF_FLAG_EXACT_PIECE_TIME | SVSDF_FLAG_FAST_PIECE_TIME);
    if (std::string(e) == "exact") ctx->cfg.flags |= SVSDF_FLAG_EXACT_PIECE_TIME;
    else if (std::string(e) == "fast") ctx->cfg.flags |= SVSDF_FLAG_FAST_PIECE_TIME;
  }
  for (int b = 0; b < kMaxBatches; ++b) {
    if (hipEventCreateWithFlags(&ctx->ev_done[b], hipEventDisableTiming) != hipSuccess)
      return bail("event creation failed");
  }
  if (hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming) != hipSuccess) return bail("event creation failed");
  if (hipMalloc((void **)&ctx->d_traj, sizeof(TrajDev)) != hipSuccess ||
      hipMalloc((void **)&ctx->d_ctl, kMaxBatches * sizeof(BatchCtl) + 8 * 8 * (kMaxIter + 4)) != hipSuccess ||
      hipMalloc((void **)&ctx->d_nonfinite, sizeof(int)) != hipSuccess ||
      hipMalloc((void **)&ctx->d_sums, kOutPartial * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&ctx->d_out, kOutDoubles * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void **)&ctx->h_out, kOutDoubles * sizeof(double), hipHostMallocDefault) != hipSuccess ||
      hipMalloc((void **)&ctx->d_ticket, sizeof(unsigned)) != hipSuccess)
    return bail("device allocation failed");
  if (hipMemsetAsync(ctx->d_ticket, 0, sizeof(unsigned), ctx->stream) != hipSuccess) return bail("hipMemset failed");
  {  // the pinned result buffer as the device sees it: k_reduce's last block writes the evaluation's result there directly
     // (SVSDF_HOST_WRITE=0: a device-to-host copy command instead, as before round 5)
    void *dp = nullptr;
    const char *e = std::getenv("SVSDF_HOST_WRITE");
    if (!(e && std::atoi(e) == 0) && hipHostGetDevicePointer(&dp, ctx->h_out, 0) == hipSuccess) ctx->h_out_dev = (double *)dp;
    (void)hipGetLastError();
  }
  // (on the context's own stream: it is non-blocking, a null-stream memset would not be ordered with its kernels)
  if (hipMemsetAsync(ctx->d_ctl, 0, kMaxBatches * sizeof(BatchCtl) + 8 * 8 * (kMaxIter + 4), ctx->stream) != hipSuccess) return bail("hipMemset failed");
  {  // shape bound radius R with sdf_shape(q) >= |q| - R for every q: the shape's circumradius about the body origin
     // (analytic, per shape; shape_circumradius above) plus the length of its offset (Shape.hpp:281-294).  The polar
     // sample of |q| - sdf(q) (k_rbound, out to 60 m) is kept as a self-check of that bound, not as its source.
    const double r0 = shape_circumradius(cfg->shape_id, ctx->poly_xy.data(), (int)(ctx->poly_xy.size() / 2));
    const double analytic = (r0 + std::hypot(sp.tx, sp.ty)) * (1.0 + 1e-12) + 1e-6;
    if (hipMemsetAsync(ctx->d_out, 0, 2 * sizeof(double), ctx->stream) != hipSuccess) return bail("hipMemset failed");
    const int nrad = 512, nang = 4096;
    const unsigned grid = (unsigned)((nrad * nang + kBlock - 1) / kBlock);
    (void)launch_k_rbound(cfg->shape_id, grid, ctx->stream, ctx->sp, 60.0, nrad, nang, ctx->d_out);
    double rbl[2] = {0.0, 0.0};
    if (hipMemcpyAsync(rbl, ctx->d_out, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
      return bail("shape bound kernel failed");
    const double rb = rbl[0];
    ctx->r_bound_sampled = rb;
    // 1-Lipschitz check of the shape SDF (k_rbound): the value-based second cull and the anchor bound mode are exact
    // only for such a function; a shape that fails (none of the 17 does) runs without both.  It is a SAMPLED
    // sanity check (every node of a polar grid against four neighbours), not a proof: what makes the property true is
    // that all 17 shapes are exact distance functions (tools/experiments/shape_lipschitz_check.py is the denser CPU study).
    ctx->lipschitz_excess = rbl[1];
    const char *nl = std::getenv("SVSDF_ASSUME_NOT_LIPSCHITZ");   // (test switch; "0" leaves the check's verdict alone)
    if (rbl[1] > 0.0 || (nl && std::atoi(nl) != 0)) { ctx->lipschitz_ok = false; ctx->cull2 = false; }
    if (rb > analytic)
      return bail("svsdf_create: sampled shape bound " + std::to_string(rb) + " exceeds the analytic circumradius " +
                  std::to_string(analytic) + " (internal error: the pruning bound would be unsafe)");
    ctx->r_bound = analytic;
    ctx->sp.r_bound = ctx->r_bound;
  }
  return ctx;
}

void svsdf_destroy(svsdf_ctx *ctx) {
  if (!ctx) return;
  if (ctx->host_only) { delete ctx; return; }
  if (!ctx->subs.empty() || !ctx->workers.empty()) {
    destroy_group_resources(ctx);
    for (svsdf_ctx *s : ctx->subs) svsdf_destroy(s);
    delete ctx;
    return;
  }
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  void *bufs[] = {ctx->d_poly, ctx->d_px, ctx->d_py, ctx->d_traj, ctx->d_in, ctx->d_pose, ctx->d_chunks,
                  ctx->d_sdf, ctx->d_t, ctx->d_res_sdf, ctx->d_res_t, ctx->d_res_gx, ctx->d_res_gy, ctx->gs.pt,
                  ctx->gs.r, ctx->gs.theta0, ctx->gs.theta_res, ctx->gs.iter, ctx->gs.nsamp, ctx->gs.phase, ctx->gs.req,
                  ctx->gs.list[0], ctx->gs.list[1], ctx->gs.solve, ctx->gs.sqx, ctx->gs.sqy, ctx->gs.sqth,
                  ctx->gs.sq_ub, ctx->gs.sq_k, ctx->gs.sq_sdf, ctx->gs.sq_t, ctx->d_ctl,
                  ctx->d_block_partials, ctx->d_sums,
                  ctx->d_out, ctx->d_nonfinite, ctx->d_fe, ctx->d_fe_flag, ctx->d_ticket};
  for (void *p : bufs)
    if (p) (void)hipFree(p);
  if (ctx->h_in) (void)hipHostFree(ctx->h_in);
  if (ctx->h_out) (void)hipHostFree(ctx->h_out);
  if (ctx->h_fe) (void)hipHostFree(ctx->h_fe);
  for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->ev_prep) (void)hipEventDestroy(ctx->ev_prep);
  for (int b = 0; b < kMaxBatches; ++b)
    if (ctx->ev_done[b]) (void)hipEventDestroy(ctx->ev_done[b]);
  if (ctx->stream_slot >= 0) {   // the streams go back to the pool (idle: the device was synchronised above)
    StreamSet ss;
    ss.main = ctx->stream;
    for (int b = 0; b < kMaxBatches; ++b) ss.batch[b] = ctx->bstream[b];
    ss.slot = ctx->stream_slot;
    release_streams(ctx->device, ss);
  }
  delete ctx;
}

int svsdf_set_points(svsdf_ctx *ctx, const double *xyz_aos, size_t P) {
  if (!ctx || (!xyz_aos && P)) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_set_points: null argument");
  return set_points_host(ctx, xyz_aos, P);
}

int svsdf_set_points_device(svsdf_ctx *ctx, const double *d_xyz_aos, size_t P) {
  if (!ctx || (!d_xyz_aos && P)) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_set_points_device: null argument");
  if (ctx->host_only) return fail(ctx, SVSDF_ERR_NO_DEVICE, "host-only context: no device entry points");
  if (P > 0xffffffffull) return fail(ctx, SVSDF_ERR_INVALID, "too many points");
  if (ctx->subs.empty()) {   // the cloud never leaves the device: keys, radix sort and stripe gather run there
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = upload_shard_device(ctx, d_xyz_aos, P, ctx->cfg.rank, ctx->cfg.world_size);
    ctx->setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
  }
  // multi-device context: planned once where the cloud lives (devices[0]), stripes handed to the other devices
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = upload_group_device(ctx, d_xyz_aos, P);
  ctx->setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (svsdf_ctx *s : ctx->subs) s->setup_ms = ctx->setup_ms;
  return rc;
}

size_t svsdf_num_points(const svsdf_ctx *ctx) { return ctx ? ctx->P : 0; }

int svsdf_shard_indices(const svsdf_ctx *ctx, long long *idx_out) {
  if (!ctx || !idx_out) return SVSDF_ERR_INVALID;
  std::copy(ctx->shard_idx.begin(), ctx->shard_idx.end(), idx_out);
  return SVSDF_OK;
}

int svsdf_eval_penalty_partial(svsdf_ctx *ctx, int N, const double *coeffs, const double *T,
                               double **d_partial, size_t *partial_len) {
  if (!ctx || !coeffs || !T) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_eval_penalty_partial: null argument");
  int rc = run_pipeline(ctx, N, coeffs, T);
  if (rc) return rc;
  double *dp = ctx->d_out;
  if (!ctx->subs.empty()) {
    // multi-process x multi-device: the node-local sum goes back to device 0 for the caller's collective
    svsdf_ctx *s0 = ctx->subs[0];
    dp = (ctx->combine == SVSDF_COMBINE_RCCL) ? ctx->d_red[0] : s0->d_out;
    if (ctx->combine != SVSDF_COMBINE_RCCL) {
      HIPCHK(hipSetDevice(s0->device));
      HIPCHK(hipMemcpy(dp, ctx->comb.data(), (19 * (size_t)N + 1) * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  if (d_partial) *d_partial = dp;
  if (partial_len) *partial_len = 19 * (size_t)N + 1;
  return SVSDF_OK;
}

int svsdf_accumulate_partial(svsdf_ctx *ctx, int N, const double *partial_host, double *cost,
                             double *gradT, double *gradC) {
  if (!partial_host || !cost || !gradT || !gradC || N < 1 || N > kMaxPieces)  // ctx may be NULL (pure host)
    return fail(ctx, SVSDF_ERR_INVALID, "svsdf_accumulate_partial: invalid argument");
  for (int e = 0; e < 19 * N + 1; ++e)
    if (!std::isfinite(partial_host[e])) return fail(ctx, SVSDF_ERR_NONFINITE, "non-finite partial");
  accumulate(N, partial_host, cost, gradT, gradC);
  return SVSDF_OK;
}

int svsdf_sum_partials(const double *partials, int G, size_t len, double *out) {
  if (!partials || !out || G < 1) return SVSDF_ERR_INVALID;
  for (size_t e = 0; e < len; ++e) {
    double a = partials[e];
    for (int k = 1; k < G; ++k) a += partials[(size_t)k * len + e];
    out[e] = a;
  }
  return SVSDF_OK;
}

int svsdf_set_conditions(svsdf_ctx *ctx, const double head_state[9], const double tail_state[9]) {
  if (!ctx || !head_state || !tail_state) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_set_conditions: null argument");
  for (int i = 0; i < 9; ++i)
    if (!std::isfinite(head_state[i]) || !std::isfinite(tail_state[i]))
      return fail(ctx, SVSDF_ERR_NONFINITE, "svsdf_set_conditions: non-finite boundary state");
  std::copy(head_state, head_state + 9, ctx->cfg.head_state);
  std::copy(tail_state, tail_state + 9, ctx->cfg.tail_state);
  return SVSDF_OK;
}

int svsdf_eval_penalty(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, double *cost,
                       double *gradT, double *gradC) {
  if (!ctx || !coeffs || !T || !cost || !gradT || !gradC)
    return fail(ctx, SVSDF_ERR_INVALID, "svsdf_eval_penalty: null argument");
  int rc = run_pipeline(ctx, N, coeffs, T);
  if (rc) return rc;
  return svsdf_accumulate_partial(ctx, N, ctx->h_partial, cost, gradT, gradC);
}

int svsdf_query_points(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, double *sdf,
                       double *tstar, double *grad_xy) {
  if (!ctx || !coeffs || !T) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_query_points: null argument");
  if (!ctx->subs.empty()) {   // per device, concatenated in svsdf_shard_indices order
    std::vector<size_t> off(ctx->subs.size() + 1, 0);
    for (size_t k = 0; k < ctx->subs.size(); ++k) off[k + 1] = off[k] + ctx->subs[k]->P;
    const int rcg = group_run(ctx, [&](int k) -> int {
      return svsdf_query_points(ctx->subs[k], N, coeffs, T, sdf ? sdf + off[k] : nullptr, tstar ? tstar + off[k] : nullptr,
                                grad_xy ? grad_xy + 2 * off[k] : nullptr);
    });
    if (!rcg) merge_stats(ctx);
    return rcg;
  }
  if (ctx->host_only) return fail(ctx, SVSDF_ERR_NO_DEVICE, "host-only context: no device entry points");
  if (!ctx->points_set) return fail(ctx, SVSDF_ERR_NO_POINTS, "svsdf_set_points has not been called");
  if (ctx->P == 0) return SVSDF_OK;
  int rc = evaluate_points(ctx, N, coeffs, T, /*allow_cull=*/false, /*with_partial=*/false);  // per-point outputs need every solve
  if (rc) return rc;
  fill_mode_stats(ctx);
  const size_t P = ctx->P;
  if (sdf) HIPCHK(hipMemcpy(sdf, ctx->d_res_sdf, P * sizeof(double), hipMemcpyDeviceToHost));
  if (tstar) HIPCHK(hipMemcpy(tstar, ctx->d_res_t, P * sizeof(double), hipMemcpyDeviceToHost));
  if (grad_xy) {
    std::vector<double> gx(P), gy(P);
    HIPCHK(hipMemcpy(gx.data(), ctx->d_res_gx, P * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(gy.data(), ctx->d_res_gy, P * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t j = 0; j < P; ++j) { grad_xy[2 * j] = gx[j]; grad_xy[2 * j + 1] = gy[j]; }
  }
  return SVSDF_OK;
}

long long svsdf_debug_sincos_mismatches(svsdf_ctx *ctx, double lo, double hi, int n) {
  if (!ctx || ctx->host_only || n < 2) return -1;
  if (!ctx->subs.empty()) return svsdf_debug_sincos_mismatches(ctx->subs[0], lo, hi, n);
  return sincos_mismatches(ctx, lo, hi, n);
}

#ifdef SVSDF_SITE_STATS
// diagnostic builds only (tools/site_stats.py): k_solve's per-site execution / lane counters of the last evaluation
int svsdf_debug_site_stats(svsdf_ctx *ctx, unsigned long long out[26]) {
  if (!ctx || ctx->host_only || !ctx->subs.empty()) return SVSDF_ERR_INVALID;
  if (hipSetDevice(ctx->device) != hipSuccess) return SVSDF_ERR_HIP_BASE;
  std::vector<BatchCtl> hc(kMaxBatches);
  if (hipMemcpy(hc.data(), ctx->d_ctl, sizeof(BatchCtl) * kMaxBatches, hipMemcpyDeviceToHost) != hipSuccess) return SVSDF_ERR_HIP_BASE;
  for (int i = 0; i < 26; ++i) out[i] = 0;
  for (const BatchCtl &b : hc)
    for (const StatSlot &sl : b.stat)
      for (int i = 0; i < 26; ++i) {
        if (i == 22 || i == 23) out[i] = std::max(out[i], (unsigned long long)sl.pad[i]);   // maxima over the waves (k_tail)
        else out[i] += sl.pad[i];
      }
  return SVSDF_OK;
}
#endif

int svsdf_debug_sdf_at(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, size_t n, const double *points_xy,
                       const double *t, double *out8) {
  if (!ctx || !coeffs || !T || (n && (!points_xy || !t || !out8))) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_debug_sdf_at: null argument");
  if (ctx->host_only) return fail(ctx, SVSDF_ERR_NO_DEVICE, "host-only context: no device entry points");
  if (!ctx->subs.empty()) return svsdf_debug_sdf_at(ctx->subs[0], N, coeffs, T, n, points_xy, t, out8);
  if (n == 0) return SVSDF_OK;
  if (n > 0x7fffffffull) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_debug_sdf_at: too many queries");
  return debug_sdf_at(ctx, N, coeffs, T, n, points_xy, t, out8);
}

int svsdf_set_profiling(svsdf_ctx *ctx, int enable) {
  if (!ctx) return SVSDF_ERR_INVALID;
  if (!ctx->subs.empty()) {
    int rc = SVSDF_OK;
    for (svsdf_ctx *s : ctx->subs) { const int r = svsdf_set_profiling(s, enable); if (r && !rc) rc = r; }
    ctx->profile = enable != 0 && enable != 3;
    ctx->profile_span = enable == 3;
    return rc;
  }
  // enable == 3 (round 6): the evaluation's device span only (stats.device_ms from the two events every evaluation records
  // anyway).  The per-launch events of levels 1 / 2 -- two per kernel, ~ 60 per evaluation -- are not free: a 500 k-point
  // stripe measures 3.85 ms with them and 3.67 ms without (the same evaluation by the host's clock).
  ctx->profile = enable != 0 && enable != 3;
  ctx->profile_span = enable == 3;
  if (ctx->host_only) return SVSDF_OK;
  // enable == 2: also run the point batches one after the other (one batch) while profiling, so that every launch's
  // duration is its own cost and not stretched by the kernels of the other batches it normally overlaps with
  if (enable == 2 && ctx->nbatch > 1 && ctx->points_set) {
    ctx->saved_nbatch = ctx->nbatch;
    HIPCHK(hipSetDevice(ctx->device));
    return set_batches(ctx, 1);
  }
  if (enable != 2 && ctx->saved_nbatch > 0 && ctx->points_set) {
    const int nb = ctx->saved_nbatch;
    ctx->saved_nbatch = 0;
    HIPCHK(hipSetDevice(ctx->device));
    return set_batches(ctx, nb);
  }
  return SVSDF_OK;
}

int svsdf_shape_bound(const svsdf_ctx *ctx, double out2[2]) {
  if (!ctx || !out2) return SVSDF_ERR_INVALID;
  const svsdf_ctx *c = ctx->subs.empty() ? ctx : ctx->subs[0];
  out2[0] = c->r_bound;
  out2[1] = c->r_bound_sampled;
  return SVSDF_OK;
}

int svsdf_shape_selfcheck(const svsdf_ctx *ctx, double out3[3]) {
  if (!ctx || !out3) return SVSDF_ERR_INVALID;
  const svsdf_ctx *c = ctx->subs.empty() ? ctx : ctx->subs[0];
  out3[0] = c->r_bound;
  out3[1] = c->r_bound_sampled;
  out3[2] = c->lipschitz_ok ? 0.0 : std::max(c->lipschitz_excess, 1e-300);
  return SVSDF_OK;
}

int svsdf_get_plan(const svsdf_ctx *ctx, svsdf_plan *out) {
  if (!ctx || !out) return SVSDF_ERR_INVALID;
  const svsdf_ctx *c = ctx->subs.empty() ? ctx : ctx->subs[0];
  out->bound_mode = bound_mode_of(c);
  out->batches = (c->saved_nbatch > 0) ? c->saved_nbatch : c->nbatch;
  out->lanes_per_query = c->G;
  out->tail_iter = (c->tail_mode == -2) ? -2 : (c->tail_mode >= 0) ? c->tail_mode : (c->have_prev_nactive ? choose_tail_iter(c) : SVSDF_PLAN_AUTO);
  out->settled = ((c->ub_env || c->ub_tune > 0) && c->bt_state == 0 && c->an_state == 0 && c->lz_state == 0 && c->have_prev_nsolve) ? 1 : 0;
  return SVSDF_OK;
}

int svsdf_set_plan(svsdf_ctx *ctx, const svsdf_plan *plan) {
  if (!ctx || !plan) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_set_plan: null argument");
  const int g = plan->lanes_per_query;
  if (plan->bound_mode < SVSDF_PLAN_AUTO || plan->bound_mode > 3 || plan->batches < -2 || plan->batches == 0 || plan->batches > kMaxBatches ||
      !(g == SVSDF_PLAN_AUTO || g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 32) || plan->tail_iter < -2 || plan->tail_iter >= kMaxIter)
    return fail(ctx, SVSDF_ERR_INVALID, "svsdf_set_plan: field out of range");
  if (!ctx->subs.empty()) {
    int rc = SVSDF_OK;
    for (svsdf_ctx *s : ctx->subs) { const int r = svsdf_set_plan(s, plan); if (r && !rc) { rc = r; ctx->err = s->err; } }
    return rc;
  }
  if (ctx->host_only) return SVSDF_OK;
  HIPCHK(hipSetDevice(ctx->device));
  // bound mode: a change invalidates the launch widths on record (they belong to the other mode's solve counts)
  if (plan->bound_mode == SVSDF_PLAN_AUTO) {
    if (ctx->ub_env) { ctx->ub_env = false; ctx->ub_tune = 0; ctx->have_prev_nsolve = false; ctx->have_prev_nactive = false; }
  } else {
    const bool full = plan->bound_mode != 0, lazy = plan->bound_mode == 2, anchor = plan->bound_mode == 3;
    if (!ctx->ub_env || full != ctx->ub_full || lazy != ctx->ub_lazy || anchor != ctx->ub_anchor) { ctx->have_prev_nsolve = false; ctx->have_prev_nactive = false; ctx->ub_tune = 0; }
    ctx->ub_env = true; ctx->ub_full = full; ctx->ub_lazy = lazy; ctx->ub_anchor = anchor; ctx->an_state = 0; ctx->lz_state = 0;
  }
  ctx->want_batches = (plan->batches == SVSDF_PLAN_AUTO) ? 0 : (plan->batches == -2) ? -1 : plan->batches;
  ctx->bt_state = 0;
  if (ctx->want_batches <= 0) ctx->ub_tune = 0;   // rule / measurement run again after the next evaluation
  else if (ctx->points_set) {
    if (ctx->saved_nbatch > 0) ctx->saved_nbatch = ctx->want_batches;
    else if (ctx->nbatch != ctx->want_batches) { const int rc = set_batches(ctx, ctx->want_batches); if (rc) return rc; }
  }
  ctx->G_env = (g == SVSDF_PLAN_AUTO) ? 0 : g;
  if (ctx->G_env) { ctx->G = g; if (!ctx->G_late_env) ctx->G_late = std::max(g, 8); }
  else if (ctx->points_set) {
    const size_t Ps = ctx->P;
    ctx->G = default_lanes(ctx, Ps);
    if (!ctx->G_late_env) ctx->G_late = std::max(ctx->G, 8);
  }
  ctx->tail_mode = (plan->tail_iter == SVSDF_PLAN_AUTO) ? -1 : plan->tail_iter;
  return SVSDF_OK;
}

int svsdf_set_combine(svsdf_ctx *ctx, int combine) {
  if (!ctx || ctx->subs.empty()) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_set_combine: not a multi-device context");
  if (combine != SVSDF_COMBINE_HOST && combine != SVSDF_COMBINE_RCCL) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_set_combine: unknown mode");
  if (combine == SVSDF_COMBINE_RCCL) {
    const std::string e = group_init_rccl(ctx);
    if (!e.empty()) return fail(ctx, SVSDF_ERR_RCCL, "svsdf_set_combine: " + e);
  }
  ctx->combine = combine;
  return SVSDF_OK;
}

int svsdf_group_info(const svsdf_ctx *ctx, int *n_devices, int *combine, int *rccl_ranks) {
  if (!ctx) return SVSDF_ERR_INVALID;
  if (n_devices) *n_devices = ctx->subs.empty() ? 1 : (int)ctx->subs.size();
  if (combine) *combine = ctx->subs.empty() ? SVSDF_COMBINE_HOST : ctx->combine;
  if (rccl_ranks) {
    *rccl_ranks = rccl_comm_count(ctx);   // asked of the communicator itself (ncclCommCount); 0: none
  }
  return SVSDF_OK;
}

int svsdf_group_stripe(const svsdf_ctx *ctx, int k, int *device, size_t *points, svsdf_stats *stats, svsdf_plan *plan) {
  if (!ctx) return SVSDF_ERR_INVALID;
  const int G = ctx->subs.empty() ? 1 : (int)ctx->subs.size();
  if (k < 0 || k >= G) return SVSDF_ERR_INVALID;
  const svsdf_ctx *s = ctx->subs.empty() ? ctx : ctx->subs[k];
  if (device) *device = s->device;
  if (points) *points = s->P;
  if (stats) *stats = s->stats;
  if (plan) return svsdf_get_plan(s, plan);
  return SVSDF_OK;
}

int svsdf_set_group_serial(svsdf_ctx *ctx, int serial) {
  if (!ctx || ctx->subs.empty()) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_set_group_serial: not a multi-device context");
  ctx->group_serial = serial != 0;
  return SVSDF_OK;
}

int svsdf_last_stats(const svsdf_ctx *ctx, svsdf_stats *out) {
  if (!ctx || !out) return SVSDF_ERR_INVALID;
  *out = ctx->stats;
  return SVSDF_OK;
}

// ---- host MINCO helpers --------------------------------------------------------------------------
int svsdf_minco_coeffs(const double head_state[9], const double tail_state[9], int N, const double *inPs,
                       const double *T, double *coeffs) {
  if (!head_state || !tail_state || !T || !coeffs || N < 1 || (N > 1 && !inPs)) return SVSDF_ERR_INVALID;
  svsdf_host::MincoS3 m;
  m.set_conditions(head_state, tail_state, N);
  m.set_parameters(inPs, T);
  m.coeffs_colmajor(coeffs);
  return SVSDF_OK;
}

void svsdf_forward_T(const double *tau, double *T, int N) {
  for (int i = 0; i < N; ++i) T[i] = svsdf_host::tau_to_T(tau[i]);
}
void svsdf_backward_T(const double *T, double *tau, int N) {
  for (int i = 0; i < N; ++i) tau[i] = svsdf_host::T_to_tau(T[i]);
}

// ---- full optimizer callback (BEO:344-408) -------------------------------------------------------
static int lmbm_prepare(svsdf_ctx *ctx, const double *x, int n) {
  if (!ctx || !x || n < 1 || (n + 3) % 4 != 0) return fail(ctx, SVSDF_ERR_INVALID, "n must be 4N - 3");
  const int N = (n + 3) / 4;
  if (N > kMaxPieces) return fail(ctx, SVSDF_ERR_INVALID, "N > 128");
  ctx->xlast.assign(x, x + n);
  ctx->T.resize(N);
  for (int i = 0; i < N; ++i) ctx->T[i] = svsdf_host::tau_to_T(x[i]);  // forwardT
  ctx->minco.set_conditions(ctx->cfg.head_state, ctx->cfg.tail_state, N);
  ctx->minco.set_parameters(x + N, ctx->T.data());                      // forwardP is a reshape
  ctx->energy_cost = ctx->minco.energy();
  ctx->pgC.resize(18 * (size_t)N);
  ctx->pgT.resize(N);
  ctx->minco.energy_grad_coeffs(ctx->pgC.data());
  ctx->minco.energy_grad_times(ctx->pgT.data());
  ctx->cm.resize(18 * (size_t)N);
  ctx->minco.coeffs_colmajor(ctx->cm.data());
  return SVSDF_OK;
}

static double lmbm_complete(svsdf_ctx *ctx, const double *partial, const double *x, double *g, int n) {
  const int N = (n + 3) / 4;
  double cost = ctx->energy_cost;
  ctx->gC.assign(18 * (size_t)N, 0.0);
  for (int r = 0; r < 6 * N; ++r)
    for (int c = 0; c < 3; ++c) ctx->gC[(size_t)c * 6 * N + r] = ctx->pgC[r * 3 + c];
  accumulate(N, partial, &cost, ctx->pgT.data(), ctx->gC.data());
  for (int r = 0; r < 6 * N; ++r)
    for (int c = 0; c < 3; ++c) ctx->pgC[r * 3 + c] = ctx->gC[(size_t)c * 6 * N + r];
  const double pos_cost = cost - ctx->energy_cost;
  ctx->gradq.assign(3 * (size_t)std::max(1, N - 1), 0.0);
  ctx->gradT.assign(N, 0.0);
  ctx->minco.propagate(ctx->pgC.data(), ctx->pgT.data(), ctx->gradq.data(), ctx->gradT.data());
  double tsum = 0.0;
  for (int i = 0; i < N; ++i) tsum += ctx->T[i];
  cost += ctx->cfg.rho * tsum;
  ctx->costs3[0] = pos_cost;
  ctx->costs3[1] = cost - pos_cost;
  ctx->costs3[2] = cost;
  for (int i = 0; i < N; ++i) g[i] = svsdf_host::grad_T_to_tau(x[i], ctx->gradT[i] + ctx->cfg.rho);
  for (int i = 0; i + 1 < N; ++i)
    for (int c = 0; c < 3; ++c) g[N + 3 * i + c] = ctx->gradq[i * 3 + c];
  return cost;
}

int svsdf_shard_plan(const double *xyz_aos, size_t P, int rank, int world_size, int flags,
                     long long *idx_out, size_t *count_out) {
  if ((!xyz_aos && P) || rank < 0 || world_size < 1 || rank >= world_size) return SVSDF_ERR_INVALID;
  std::vector<long long> idx;
  shard_plan(xyz_aos, P, rank, world_size, flags, idx);
  if (count_out) *count_out = idx.size();
  if (idx_out) std::copy(idx.begin(), idx.end(), idx_out);
  return SVSDF_OK;
}

int svsdf_lmbm_prepare(svsdf_ctx *ctx, const double *x, int n, double *coeffs_out, double *T_out) {
  int rc = lmbm_prepare(ctx, x, n);
  if (rc) return rc;
  const int N = (n + 3) / 4;
  if (coeffs_out) std::copy(ctx->cm.begin(), ctx->cm.end(), coeffs_out);
  if (T_out) std::copy(ctx->T.begin(), ctx->T.begin() + N, T_out);
  return SVSDF_OK;
}

int svsdf_lmbm_begin(svsdf_ctx *ctx, const double *x, int n, double **d_partial, size_t *partial_len) {
  int rc = lmbm_prepare(ctx, x, n);
  if (rc) return rc;
  const int N = (n + 3) / 4;
  return svsdf_eval_penalty_partial(ctx, N, ctx->cm.data(), ctx->T.data(), d_partial, partial_len);
}

// Uses the x given to the matching svsdf_lmbm_begin; the caller passes the same n.
double svsdf_lmbm_finish(svsdf_ctx *ctx, const double *partial_host, double *g, int n) {
  if (!ctx || !partial_host || !g || (int)ctx->xlast.size() != n) return std::numeric_limits<double>::infinity();
  const int N = (n + 3) / 4;
  const std::vector<double> &x = ctx->xlast;
  for (int e = 0; e < 19 * N + 1; ++e)
    if (!std::isfinite(partial_host[e])) {
      std::fill(g, g + n, 0.0);
      fail(ctx, SVSDF_ERR_NONFINITE, "non-finite partial");
      return std::numeric_limits<double>::infinity();
    }
  return lmbm_complete(ctx, partial_host, x.data(), g, n);
}

double svsdf_lmbm_evaluate(void *vctx, const double *x, double *g, const int n) {
  svsdf_ctx *ctx = (svsdf_ctx *)vctx;
  const double inf = std::numeric_limits<double>::infinity();
  if (g && n > 0) std::fill(g, g + n, 0.0);
  if (!ctx || !x || !g) return inf;
  if (lmbm_prepare(ctx, x, n)) return inf;
  const int N = (n + 3) / 4;
  if (run_pipeline(ctx, N, ctx->cm.data(), ctx->T.data())) return inf;
  const size_t plen = 19 * (size_t)N + 1;
  for (size_t e = 0; e < plen; ++e)
    if (!std::isfinite(ctx->h_partial[e])) return inf;
  return lmbm_complete(ctx, ctx->h_partial, x, g, n);
}

int svsdf_last_costs(const svsdf_ctx *ctx, double costs3[3]) {
  if (!ctx || !costs3) return SVSDF_ERR_INVALID;
  costs3[0] = ctx->costs3[0]; costs3[1] = ctx->costs3[1]; costs3[2] = ctx->costs3[2];
  return SVSDF_OK;
}

}  // extern "C"
