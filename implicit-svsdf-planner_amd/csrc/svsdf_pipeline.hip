// svsdf_pipeline.hip -- device pipeline of ONE device behind the C ABI (include/svsdf_c.h): kernel dispatch, launchers,
// trajectory upload (SweptVolumeManager::updateTraj, SWM:376-385), one evaluation of
// addSaftyPenaOnSweptVolumeParallelTrueSDF (BEO:774-869) as a chain of launches per point batch, launch-plan rules, and
// the query-point upload (sort, stripes).  This translation unit also compiles the shape-independent kernels of
// svsdf_kernels.hpp (SVSDF_API_TU); the shape-templated ones live in svsdf_shape_slice.hip.
#include <hipcub/hipcub.hpp>

#define SVSDF_API_TU
#include "svsdf_ctx.hpp"

using namespace svsdf;

namespace svsdf_impl {

thread_local std::string g_last_error;

int fail(svsdf_ctx *ctx, int code, const std::string &msg) {
  g_last_error = msg;
  if (ctx) ctx->err = msg;
  return code;
}

const char *kShapeNames[SVSDF_SHAPE_COUNT] = {
    "sdUnevenCapsule", "sdCutDisk", "sdTrapezoid", "sdRhombus", "star", "sdTunnel",
    "sdHorseshoe", "sdHeart", "sdOrientedVesica", "sdRoundedCross", "sdRoundedX", "bigX",
    "sdMoon", "sdPie", "sdPie2", "sdArc", "Polygon"};
inline uint32_t part1by1(uint32_t x) {
  x &= 0x0000ffff;
  x = (x ^ (x << 8)) & 0x00ff00ff;
  x = (x ^ (x << 4)) & 0x0f0f0f0f;
  x = (x ^ (x << 2)) & 0x33333333;
  x = (x ^ (x << 1)) & 0x55555555;
  return x;
}

// process-wide stream pool (svsdf_ctx.hpp): per device, sets of 1 + kMaxBatches non-blocking streams
namespace {
struct PoolEntry { StreamSet set; bool in_use = false; };
std::mutex g_pool_mutex;
std::vector<std::vector<PoolEntry>> g_pool;   // [device][slot]
}  // namespace

bool acquire_streams(int device, StreamSet &out) {
  std::lock_guard<std::mutex> lk(g_pool_mutex);
  if (device < 0) return false;
  if ((size_t)device >= g_pool.size()) g_pool.resize((size_t)device + 1);
  std::vector<PoolEntry> &pool = g_pool[(size_t)device];
  for (size_t k = 0; k < pool.size(); ++k)
    if (!pool[k].in_use) {   // lowest free slot first
      // A pooled set outlives its contexts on purpose (never destroyed: see svsdf_ctx.hpp) -- but not a hipDeviceReset(),
      // after which its handles dangle.  An idle stream answers hipStreamQuery with success; anything but
      // success / not-ready means the handle is dead: the slot gets a fresh set, in the same fixed order.
      bool alive = true;
      {
        const hipError_t q = hipStreamQuery(pool[k].set.main);
        if (q != hipSuccess && q != hipErrorNotReady) { alive = false; (void)hipGetLastError(); }
      }
      if (!alive) {
        StreamSet fresh;
        fresh.slot = (int)k;
        bool ok = hipStreamCreateWithFlags(&fresh.main, hipStreamNonBlocking) == hipSuccess;
        for (int b = 0; b < kMaxBatches && ok; ++b) ok = hipStreamCreateWithFlags(&fresh.batch[b], hipStreamNonBlocking) == hipSuccess;
        if (!ok) return false;
        pool[k].set = fresh;
      }
      pool[k].in_use = true; out = pool[k].set; return true;
    }
  PoolEntry e;
  e.set.slot = (int)pool.size();
  bool ok = hipStreamCreateWithFlags(&e.set.main, hipStreamNonBlocking) == hipSuccess;
  for (int b = 0; b < kMaxBatches && ok; ++b) ok = hipStreamCreateWithFlags(&e.set.batch[b], hipStreamNonBlocking) == hipSuccess;
  if (!ok) {
    if (e.set.main) (void)hipStreamDestroy(e.set.main);
    for (int b = 0; b < kMaxBatches; ++b) if (e.set.batch[b]) (void)hipStreamDestroy(e.set.batch[b]);
    return false;
  }
  e.in_use = true;
  pool.push_back(e);
  out = e.set;
  return true;
}

void release_streams(int device, StreamSet &s) {
  std::lock_guard<std::mutex> lk(g_pool_mutex);
  if (device >= 0 && (size_t)device < g_pool.size() && s.slot >= 0 && (size_t)s.slot < g_pool[(size_t)device].size())
    g_pool[(size_t)device][(size_t)s.slot].in_use = false;
  s = StreamSet{};
}

// SVSDF_DEBUG_SYNC=1 (diagnosis only): every launch of the chain is followed by a stream synchronisation and named on stderr
// BEFORE it runs, so that a device fault (which aborts the process at the next synchronisation) is pinned to one launch.
static bool debug_sync_on() { static const bool on = [] { const char *e = std::getenv("SVSDF_DEBUG_SYNC"); return e && std::atoi(e) != 0; }(); return on; }
static void debug_sync(hipStream_t st, const char *what, int a, int b) {
  if (!debug_sync_on()) return;
  std::fprintf(stderr, "[svsdf] launched %s (%d, %d) ... ", what, a, b);
  const hipError_t e = hipStreamSynchronize(st);
  std::fprintf(stderr, "%s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
}

size_t next_event(svsdf_ctx *ctx) {
  if (ctx->ev_used == ctx->ev_pool.size()) {
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    ctx->ev_pool.push_back(e);
  }
  return ctx->ev_used++;
}

// ---- kernel dispatch over the shape id: the shape-templated kernels live in four translation units
// (svsdf_shape_slice.hip, shapes with id % 4 == slice); a development build (-DSVSDF_FAST_BUILD) holds only star /
// sdHorseshoe / sdHeart / Polygon and serves every other id with the Polygon kernels
#define SVSDF_SLICE_DISPATCH(NAME, ...)                      \
  switch (shape % kNSlices) {                                \
    case 0: return NAME##_s0(shape, __VA_ARGS__);            \
    case 1: return NAME##_s1(shape, __VA_ARGS__);            \
    case 2: return NAME##_s2(shape, __VA_ARGS__);            \
    default: return NAME##_s3(shape, __VA_ARGS__);           \
  }
int compiled_shape(int shape) {
#ifdef SVSDF_FAST_BUILD
  return (shape == 4 || shape == 6 || shape == 7 || shape == kPolygonLds) ? shape : 16;
#else
  return shape;
#endif
}
}  // namespace svsdf_impl
namespace svsdf {
bool launch_k_solve(int shape, int G, unsigned grid, unsigned block, size_t lds, hipStream_t st, const SolveLaunch &a) {
  shape = svsdf_impl::compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_solve, G, grid, block, lds, st, a)
}
bool launch_k_round(int shape, int lp, int mode, unsigned grid, size_t lds, hipStream_t st, const RoundLaunch &a) {
  shape = svsdf_impl::compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_round, lp, mode, grid, lds, st, a)
}
bool launch_k_classify(int shape, unsigned grid, size_t lds, hipStream_t st, const ClassifyLaunch &a) {
  shape = svsdf_impl::compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_classify, grid, lds, st, a)
}
bool launch_k_tail(int shape, int mode, unsigned grid, size_t lds, hipStream_t st, const TailLaunch &a) {
  shape = svsdf_impl::compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_tail, mode, grid, lds, st, a)
}
bool launch_k_rbound(int shape, unsigned grid, hipStream_t st, ShapeParams sp, double rmax, int nrad, int nang, double *out) {
  shape = svsdf_impl::compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_rbound, grid, st, sp, rmax, nrad, nang, out)
}
bool launch_k_subsw(int shape, dim3 grid, hipStream_t st, ShapeParams sp, const double *father, const double *child,
                    const unsigned long long *offs, const double *pts, const double *kt, int nkt, int *flag) {
  shape = svsdf_impl::compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_subsw, grid, st, sp, father, child, offs, pts, kt, nkt, flag)
}
bool launch_k_debug_sdf_at(int shape, unsigned grid, size_t lds, hipStream_t st, const TrajDev *traj, ShapeParams sp,
                           const double *pxy, const double *t, int n, double *out) {
  shape = svsdf_impl::compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_debug_sdf_at, grid, lds, st, traj, sp, pxy, t, n, out)
}
bool launch_k_shape_kernels(int shape, unsigned grid, hipStream_t st, ShapeParams sp, int ks, int count, double resu,
                            int size_side, double safemargin, const double *yaw, unsigned char *map) {
  shape = svsdf_impl::compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_shape_kernels, grid, st, sp, ks, count, resu, size_side, safemargin, yaw, map)
}
}  // namespace svsdf
namespace svsdf_impl {

// Persistent-grid launcher of the argmin kernel.  max_queries bounds the (possibly device-side)
// query count and sizes the grid; surplus blocks exit before touching LDS.
size_t table_lds_doubles(const svsdf_ctx *ctx) {
  return 4 * (size_t)ctx->K + 4 * (size_t)((ctx->K + kChunk - 1) / kChunk);
}

void launch_solve(svsdf_ctx *ctx, int G, hipStream_t st, const QuerySet &qs, long long max_queries, double *out_sdf,
                  double *out_t, BatchCtl *ctl, int work_idx,
                  double cull_thresh = std::numeric_limits<double>::infinity()) {
  const double *d_tk = ctx->d_in + 19 * (size_t)ctx->N;
  // a launch of at most one query per SIMD with 32-lane groups (the main solve of a reference-scale cloud): one query per
  // wave, so that the descents take the fused pass (k_solve, `prune` bit 1)
  const bool solo = G == 32 && !ctx->G_env && max_queries <= (long long)ctx->n_cu * 4;
  const long long lanes = std::max<long long>(max_queries * (solo ? 64 : G), 64);
  // Every block stages the pose table + chunk bounds + trajectory into LDS (13 KB at 16 pieces x 2.5 s, 24 KB at 32):
  // with one wave per block that caps the CU at 160 KB / lds waves -- 6 at C3, half of what the kernel's 141 VGPRs
  // allow (3 waves per SIMD) -- so the block grows until LDS no longer binds (measured at C3: 10.7 -> 9.2 ms).
  // A Polygon's edges go in front of the tables while the whole block still fits the device's per-block LDS; a long
  // trajectory (large pose table) falls back to the kernel variant that reads the edges from global memory.
  const size_t wlds = ladder_lds_bytes(G);   // per wave: the descent state of its 64 / G groups
  bool poly_lds = ctx->poly_lds;
  size_t lds = 0, lds_total = 0;
  int blk = ctx->block;
  for (int attempt = 0; attempt < 2; ++attempt) {
    lds = (table_lds_doubles(ctx) + (size_t)traj_lds_doubles(ctx->N) + (poly_lds ? (size_t)svsdf::kPolyEdgeDoubles * (size_t)ctx->sp.nverts : 0)) * sizeof(double);
    blk = ctx->block;
    if (!ctx->block_env) blk = ((lds + wlds) * 12 <= 160 * 1024) ? 64 : ((lds + 2 * wlds) * 6 <= 160 * 1024) ? 128 : 256;
    lds_total = ((lds + 15) & ~(size_t)15) + (size_t)(blk / 64) * wlds;
    if (lds_total <= ctx->lds_limit || !poly_lds) break;
    poly_lds = false;
  }
  if (lds_total > ctx->lds_limit) {
    if (ctx->launch_err.empty()) ctx->launch_err = "trajectory too long for the LDS pose table (" + std::to_string(lds_total) + " B of " + std::to_string(ctx->lds_limit) + " B per block)";
    return;
  }
  const unsigned grid = (unsigned)std::min<long long>((lanes + blk - 1) / blk, (long long)(256 * ctx->waves_per_cu * 64) / blk);
  size_t e0 = 0, e1 = 0;
  if (ctx->profile) { e0 = next_event(ctx); (void)hipEventRecord(ctx->ev_pool[e0], st); }
  // (the second, value-based cull runs with the first one: main points of an evaluation that may cull)
  const bool cull2 = std::isfinite(cull_thresh) && ctx->cull2;
  const double *d_rot = d_tk + ctx->K + (ctx->K + kChunk - 1) / kChunk;
  const SolveLaunch a{ctx->d_traj, d_tk, ctx->d_pose, ctx->d_chunks, ctx->sp, qs, out_sdf, out_t, ctx->prune | (solo ? 2 : 0), ctl, work_idx, cull_thresh,
                      cull2 ? d_rot : nullptr, ctx->slack_max};
  if (!launch_k_solve(poly_lds ? (int)kPolygonLds : ctx->cfg.shape_id, G, grid, (unsigned)blk, lds_total, st, a) && ctx->launch_err.empty())
    ctx->launch_err = "k_solve: shape not compiled into this build";
  debug_sync(st, "k_solve", G, work_idx);
  if (ctx->profile) {
    e1 = next_event(ctx);
    (void)hipEventRecord(ctx->ev_pool[e1], st);
    ctx->refine_events.emplace_back(e0, e1);
  }
  ctx->stats.solve_launches++;
}

void launch_round(svsdf_ctx *ctx, hipStream_t st, int b, int it) {
  const int mode = bound_mode_of(ctx);   // k_round MODE: cheap / full / lazy / anchor bound
  const bool scans = mode != 0;
  const long long pts = std::max(1, ctx->bcount[b]);
  bool poly_lds = ctx->poly_lds;
  size_t lds = (table_lds_doubles(ctx) + (poly_lds ? (size_t)svsdf::kPolyEdgeDoubles * (size_t)ctx->sp.nverts : 0)) * sizeof(double);
  if (lds + 16384 > ctx->lds_limit && poly_lds) {   // (k_round's static tables: < 16 KB) edges from global memory instead
    poly_lds = false;
    lds = table_lds_doubles(ctx) * sizeof(double);
  }
  if (lds + 16384 > ctx->lds_limit) {
    if (ctx->launch_err.empty()) ctx->launch_err = "trajectory too long for the LDS pose table (k_round)";
    return;
  }
  // late iterations hold few points and are latency-bound: request every sample there, which
  // avoids supplementary iterations at no cost in time
  // the seed bound of the scanning modes is tight: a narrow band selects (ctx->scan_delta: 0.002 m, solve-all from
  // iteration 7); the chunk bound of the cheap mode needs 0.1 m / iteration 5.  SVSDF_SELECT_DELTA overrides both.
  const double sel = (scans && !ctx->select_env) ? ctx->scan_delta : ctx->select_delta;
  const int all_it = (scans && !ctx->all_iter_env) ? 7 : ctx->delta_all_iter;
  const double delta = (it >= all_it) ? 1e300 : sel;
  const double band_delta = (it >= all_it) ? 1e300 : ctx->select_delta;   // lazy mode: cheap-bound band that gets scanned
  // iterations 0 and 1 are (almost always) GSIP rounds 1 and 2 with 2 and 6 samples: 8 lanes per point; later rounds have
  // 18-21 samples: 32 lanes per point (either handles any count).  Iteration 2 (18 samples, still every interior point
  // active: throughput, not latency) also runs faster with 8 lanes and three sample passes per point -- measured round 3,
  // SVSDF_ROUND_LP8_ITERS 2 / 3 / 4 / 6 / all: C3 6.15 / 5.97 / 6.12 / 6.33 / 6.45 ms, NS 7.49 / 7.24 / 7.21 / 7.31 / 7.58
  // (round 6, re-measured with the balanced launches: 2 / 3 / 4 / 6 iterations -- C3 5.80 / 5.63 / 5.67 / 5.87 ms, C4 6.26 / 5.84 /
  // 5.88 / 6.00; the lazy mode, whose launches scan few samples per point, gains from a fourth: NS 7.45 / 7.17 / 7.03 / 7.05)
  const int lp = (it < ctx->round_lp8_iters + (mode == 2 ? 1 : 0)) ? 8 : 32;
  const unsigned grid = (unsigned)std::min<long long>((pts * lp + kRoundBlock - 1) / kRoundBlock, (long long)ctx->n_cu * ctx->round_blocks_per_cu);
  const RoundLaunch a{ctx->d_traj, ctx->d_pose, ctx->d_chunks, ctx->sp, ctx->d_px, ctx->d_py, ctx->gs, ctx->icap, it, delta,
                      band_delta, ctx->d_res_sdf, ctx->d_res_t, ctx->d_res_gx, ctx->d_res_gy, ctx->d_ctl + b, ctx->round_list | ((ctx->scan_anchors && ctx->lipschitz_ok) ? 16 : 0)};
  size_t e0 = 0, e1 = 0;
  if (ctx->profile) { e0 = next_event(ctx); (void)hipEventRecord(ctx->ev_pool[e0], st); }
  if (!launch_k_round(poly_lds ? (int)kPolygonLds : ctx->cfg.shape_id, lp, mode, grid, lds, st, a) && ctx->launch_err.empty())
    ctx->launch_err = "k_round: shape not compiled into this build";
  debug_sync(st, "k_round", lp * 10 + mode, it);
  if (ctx->profile) {
    e1 = next_event(ctx);
    (void)hipEventRecord(ctx->ev_pool[e1], st);
    ctx->round_events.emplace_back(e0, e1);
  }
}

// LDS one k_tail block asks for (Polygon edges in LDS or not), and whether the device grants it
size_t tail_lds_bytes(const svsdf_ctx *ctx, bool poly_lds, bool local) {
  const size_t lds_tables = ((table_lds_doubles(ctx) + (size_t)traj_lds_doubles(ctx->N) + (poly_lds ? (size_t)svsdf::kPolyEdgeDoubles * (size_t)ctx->sp.nverts : 0)) * sizeof(double) + 15) & ~(size_t)15;
  return lds_tables + (kTailBlock / 64) * (local ? kTailWaveLds : kTailWaveLds - kTailLocalBytes);
}

// k_tail: every GSIP iteration of batch b from `it0` on, in one launch (two points per wave, solved in the wave).
void launch_tail(svsdf_ctx *ctx, hipStream_t st, int b, int it0) {
  const int mode = bound_mode_of(ctx);
  const bool scans = mode != 0;
  const double sel = (scans && !ctx->select_env) ? ctx->scan_delta : ctx->select_delta;
  const int all_it = (scans && !ctx->all_iter_env) ? 7 : ctx->delta_all_iter;
  const int all_after = (ctx->tail_all_after < 0) ? std::max(0, all_it - it0) : ctx->tail_all_after;
  // the active count lives on the device; the grid is sized by what the previous evaluation had there (surplus blocks
  // exit at once, missing ones are made up for by the waves' work fetch)
  long long pts = std::max(1, ctx->bcount[b]);
  if (ctx->have_prev_nactive && it0 <= ctx->prev_tail_iter && it0 < kMaxIter)
    pts = std::min<long long>(pts, ctx->prev_nactive[it0] / std::max(1, ctx->nbatch) * 5 / 4 + 64);
  bool poly_lds = ctx->poly_lds;
  // the wave-local copy of the GSIP state only where it is used: the whole loop in the tail (it0 == 0) with the switch on;
  // when it does not fit, the state stays in global memory (same bits)
  bool local = ctx->tail_local && it0 == 0;
  size_t lds = 0;
  for (int attempt = 0; attempt < 4; ++attempt) {
    lds = tail_lds_bytes(ctx, poly_lds, local);
    if (lds <= ctx->lds_limit) break;
    if (local) local = false;
    else if (poly_lds) poly_lds = false;
    else break;
  }
  if (lds > ctx->lds_limit) {
    if (ctx->launch_err.empty()) ctx->launch_err = "trajectory too long for the LDS pose table (k_tail)";
    return;
  }
  // one point per wave while that still fits the chip's wave slots (3 per SIMD): the launch is a latency chain then
  const int ppw = (pts <= (long long)ctx->n_cu * 12) ? 1 : 2;
  const long long per_block = (kTailBlock / 64) * ppw;
  const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((pts + per_block - 1) / per_block, (long long)ctx->n_cu * 3));
  const double *d_tk = ctx->d_in + 19 * (size_t)ctx->N;
  const TailLaunch a{ctx->d_traj, d_tk, ctx->d_pose, ctx->d_chunks, ctx->sp, ctx->d_px, ctx->d_py, ctx->gs, ctx->icap, it0, mode,
                     sel, ctx->select_delta, all_after, ppw, ctx->d_res_sdf, ctx->d_res_t, ctx->d_res_gx, ctx->d_res_gy,
                     ctx->d_ctl + b, ctx->round_list | (local ? 4 : 0) | (ctx->tail_duo ? 0 : 8) | ((ctx->scan_anchors && ctx->lipschitz_ok) ? 16 : 0), ctx->prune,
                     // a wave slot for every point at two waves per SIMD: the instantiation without scratch (k_tail, kTailLatencyWaves)
                     (ctx->tail_latency && ppw == 1 && pts <= (long long)ctx->n_cu * 4 * svsdf::kTailLatencyWaves) ? 1 : 0};
  size_t e0 = 0, e1 = 0;
  if (ctx->profile) { e0 = next_event(ctx); (void)hipEventRecord(ctx->ev_pool[e0], st); }
  if (!launch_k_tail(poly_lds ? (int)kPolygonLds : ctx->cfg.shape_id, mode, grid, lds, st, a) && ctx->launch_err.empty())
    ctx->launch_err = "k_tail: shape not compiled into this build";
  debug_sync(st, "k_tail", mode, it0);
  if (ctx->profile) {
    e1 = next_event(ctx);
    (void)hipEventRecord(ctx->ev_pool[e1], st);
    ctx->tail_events.emplace_back(e0, e1);
  }
  ctx->stats.tail_launches++;
}

// Iteration the fused tail starts at in this evaluation.  Measured (round 4, profiles/r04_tail_*): the launch chain packs the
// solves of all points 32 to a wave and runs k_round at 4 waves per SIMD -- once a launch holds more points than the chip
// keeps in flight it has several times the tail's throughput, and its last launches take 5 - 50 us each, so placing the tail
// late in the chain of a 100 k - 1 M cloud only costs time (C2 + 6 %, C3 + 2 %, NS + 3 % with a threshold of 4096 points).
// A small cloud is a pure latency chain of ~ 20 launches: there the WHOLE GSIP loop runs in the tail (it0 = 0), as long as
// its interior points fit one generation of waves (two per wave, 12 waves per CU: 6144 on this part).  C1 workload, chain
// -> tail: 3 k points 0.68 -> 0.54 ms, 10 k (4.7 k interior) 0.89 -> 0.71, 13 k (6.1 k) 0.93 -> 0.75, 20 k (9.4 k) 1.09 ->
// 1.06, 30 k (14 k) 1.24 -> 1.56 (tools/tail_scan.py).  SVSDF_TAIL / svsdf_set_plan pin an iteration or turn the tail
// off.  Any choice gives the same bits.
int choose_tail_iter(const svsdf_ctx *ctx) {
  if (ctx->tail_mode == -2) return -1;
  //  a trajectory whose tables leave no room for the tail's per-wave state runs the launch chain, which needs
  // less LDS per block, instead of failing the evaluation (128 pieces on a device with 64 KB per block)
  if (tail_lds_bytes(ctx, false, false) > ctx->lds_limit) return -1;
  if (ctx->tail_mode >= 0) return std::min(ctx->tail_mode, (int)kMaxIter - 2);
  if (!ctx->have_prev_nactive) return -1;   // first evaluation of a point set: the interior count is not known yet
  const long long below = ctx->tail_below > 0 ? ctx->tail_below : (long long)ctx->n_cu * 24;
  return (ctx->prev_nactive[0] <= below) ? 0 : -1;
}

void launch_classify(svsdf_ctx *ctx, hipStream_t st, int b) {
  const unsigned grid = (unsigned)std::min<long long>(((long long)ctx->bcount[b] + kBlock - 1) / kBlock, 2048);
  const size_t lds = (size_t)traj_lds_doubles(ctx->N) * sizeof(double);
  const ClassifyLaunch a{ctx->d_traj, ctx->sp, ctx->d_px, ctx->d_py, ctx->d_sdf, ctx->d_t, ctx->d_res_sdf, ctx->d_res_t,
                         ctx->d_res_gx, ctx->d_res_gy, ctx->gs, ctx->d_ctl + b, &ctx->d_ctl->n_int, (int)ctx->icap};
  (void)launch_k_classify(ctx->cfg.shape_id, grid, lds, st, a);
  debug_sync(st, "k_classify", b, 0);
}

// Upload (coeffs, T), update traj_duration like SweptVolumeManager::updateTraj (SWM:376-385),
// build the layer-1 time grid exactly like the reference's accumulating loop (SWM:567).
int upload_traj(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, bool clear_nonfinite = false) {
  if (N < 1 || N > kMaxPieces) return fail(ctx, SVSDF_ERR_INVALID, "N out of range [1, 128]");
  double td = 0.0;
  for (int i = 0; i < N; ++i) td += T[i];  // Trajectory::getTotalDuration (TRJ:410-419)
  if (!(td == td) || std::isinf(td)) return fail(ctx, SVSDF_ERR_NONFINITE, "non-finite duration");
  // forwardT (BEO:213-226) only produces positive durations and Trajectory asserts t_max > 0; a non-positive
  // piece would leave the scan table empty
  for (int i = 0; i < N; ++i)
    if (!(T[i] > 0.0)) return fail(ctx, SVSDF_ERR_INVALID, "piece durations must be positive");
  if (td < 3 * 1e2 || !ctx->have_duration) {
    ctx->traj_duration = td;
    ctx->have_duration = true;
  }
  const double dur = ctx->traj_duration;
  size_t K = 0;
  for (double t = 0.0; t <= dur; t += 0.15) ++K;
  if (K < 1 || K > 16000) return fail(ctx, SVSDF_ERR_INVALID, "trajectory duration out of range for the scan table");
  const size_t need = 19 * (size_t)N + K + 2 * ((K + kChunk - 1) / kChunk);  // coeffs | T | tk | chunk slack | chunk yaw allowance
  if (need > ctx->in_cap) {
    const size_t cap = need + 4096;
    if (ctx->h_in) (void)hipHostFree(ctx->h_in);
    ctx->h_in = nullptr;
    HIPCHK(hipHostMalloc((void **)&ctx->h_in, cap * sizeof(double), hipHostMallocDefault));
    int rc = dev_alloc(ctx, &ctx->d_in, cap);
    if (rc) return rc;
    ctx->in_cap = cap;
  }
  if (K > ctx->pose_cap) {
    int rc = dev_alloc(ctx, &ctx->d_pose, K + 1024);
    if (rc) return rc;
    rc = dev_alloc(ctx, &ctx->d_chunks, 2 * ((K + 1024) / kChunk + 2));   // chunk records, then the anchor table (ChunkAnchor, k_prep)
    if (rc) return rc;
    ctx->pose_cap = K + 1024;
  }
  std::memcpy(ctx->h_in, coeffs, sizeof(double) * 18 * N);
  std::memcpy(ctx->h_in + 18 * N, T, sizeof(double) * N);
  {
    double *tk = ctx->h_in + 19 * N;
    size_t k = 0;
    for (double t = 0.0; t <= dur; t += 0.15) tk[k++] = t;
  }
  for (size_t i = 0; i < 19 * (size_t)N; ++i)
    if (!std::isfinite(ctx->h_in[i])) return fail(ctx, SVSDF_ERR_NONFINITE, "non-finite trajectory input");
  {
    // Exact cull (k_solve, main points): a point is skipped when  min_c(|p - c_c| - rb_c - slack_c) > safety_hor.
    // Chunk c holds the table times t_k0 .. t_k1; every t of [t_k0 - h, t_k1 + h] lies within h of one of them
    // (h = half the table spacing; the last chunk also covers (t_last, dur]), so |x(t) - x(t_k)| <= V_c * h with
    // V_c a rigorous bound of the planar speed on that interval: the quartic velocity polynomials lie in the
    // convex hull of their Bernstein coefficients on the sub-interval.  Then sdf(t) >= |p - x(t)| - R_shape >=
    // |p - c_c| - rb_c - V_c * h for every continuous t, whatever local minimum the reference's search returns,
    // and smoothedL1 (BEO:316-340) is inactive.  Only in the regular regime (traj_duration not stale).
    const size_t nch = (K + kChunk - 1) / kChunk;
    double *slack = ctx->h_in + 19 * (size_t)N + K;
    double *rot = slack + nch;   // W_c * h: yaw-rate allowance of the value-based second cull (scan_layer1)
    ctx->slack_max = 0.0;
    const double *tk = ctx->h_in + 19 * (size_t)N;
    ctx->cull_ok = (td == dur) && K >= 1;
    std::vector<double> S(N + 1, 0.0);
    for (int i = 0; i < N; ++i) S[i + 1] = S[i] + T[i];
    for (size_t c = 0; c < nch; ++c) {
      slack[c] = std::numeric_limits<double>::infinity();
      rot[c] = std::numeric_limits<double>::infinity();
      if (!ctx->cull_ok) continue;
      const size_t k0 = c * kChunk, k1 = std::min(k0 + kChunk, K) - 1;
      const double h = (c + 1 == nch) ? std::max(0.0751, dur - tk[k1]) : 0.0751;
      const double ta = std::max(0.0, tk[k0] - h), tb = std::min(td, tk[k1] + h);
      double v2 = 0.0, wmax = 0.0;
      for (int i = 0; i < N; ++i) {
        const double a = std::max(ta, S[i]) - S[i], b = std::min(tb, S[i + 1]) - S[i];  // local times in piece i
        if (!(b >= a)) continue;
        const double w = b - a;
        double bound[3] = {0.0, 0.0, 0.0};
        for (int d = 0; d < 3; ++d) {
          double p[5];  // velocity in s: p[k] = (k+1) c_{k+1}
          for (int k = 0; k < 5; ++k) p[k] = (k + 1) * coeffs[(size_t)d * 6 * N + 6 * i + k + 1];
          for (int j = 0; j < 4; ++j)          // Taylor shift s = a + s' (repeated synthetic division)
            for (int k = 3; k >= j; --k) p[k] += a * p[k + 1];
          double wp = 1.0;
          for (int k = 0; k < 5; ++k) { p[k] *= wp; wp *= w; }   // s' = w u, u in [0, 1]
          static const double binom4[5] = {1, 4, 6, 4, 1};
          for (int j = 0; j <= 4; ++j) {
            double bj = 0.0, cjq = 1.0;  // C(j, q)
            for (int q = 0; q <= j; ++q) { bj += cjq / binom4[q] * p[q]; cjq = cjq * (j - q) / (q + 1); }
            bound[d] = std::max(bound[d], std::fabs(bj));
          }
        }
        v2 = std::max(v2, std::hypot(bound[0], bound[1]));
        wmax = std::max(wmax, bound[2]);
      }
      const double sl = v2 * (1.0 + 1e-9) * h + 1e-9;
      if (std::isfinite(sl)) { slack[c] = sl; ctx->slack_max = std::max(ctx->slack_max, sl); }
      const double rl = wmax * (1.0 + 1e-9) * h + 1e-12;
      if (std::isfinite(rl)) rot[c] = rl;
    }
  }
  {
    // Piece-local time: the reference subtracts the durations one after the other from t
    // (TRJ:498-516); t - (T_0 + ... + T_{i-1}) in one subtraction is the same number only when every operation
    // involved is exact.  That is guaranteed when all durations are coarse dyadic numbers (multiples of 2^-20 below
    // 2^20, e.g. the 2.5 s of every BASELINE config): then all partial sums and all differences with any t < 2^30
    // are exact in both forms.  Otherwise (an optimiser's durations are generic doubles) the faithful chain runs.
    bool coarse = dur < 1073741824.0;
    double tmin = std::numeric_limits<double>::infinity();
    for (int i = 0; i < N; ++i) {
      const double v = std::ldexp(T[i], 20);
      coarse = coarse && T[i] < 1048576.0 && v == std::floor(v);
      tmin = std::min(tmin, T[i]);
    }
    const int f = ctx->cfg.flags;
    int mode = (f & SVSDF_FLAG_EXACT_PIECE_TIME) ? 1 : (f & SVSDF_FLAG_FAST_PIECE_TIME) ? 0 : (coarse ? 0 : 1);
    if (mode == 1 && !(tmin >= 1e-6)) mode = 2;
    ctx->piece_time_mode = mode;
    ctx->stats_piece_time = mode;
  }
  ctx->N = N;
  ctx->K = (int)K;
  HIPCHK(hipMemcpyAsync(ctx->d_in, ctx->h_in, need * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const size_t lds = (size_t)traj_lds_doubles(N) * sizeof(double);
  // one block of 1024 threads (round 6; 256 before): a pose per thread for the ~ 600 poses of a reference-scale trajectory
  // instead of three one after the other, and four waves per SIMD to hide the chain's latencies behind
  hipLaunchKernelGGL(k_prep, dim3(1), dim3(1024), lds, ctx->stream, ctx->d_in, N, dur, (int)K,
                     ctx->piece_time_mode, ctx->d_traj,
                     ctx->d_pose, ctx->d_chunks, ctx->r_bound, ctx->d_ctl, ctx->nbatch, clear_nonfinite ? ctx->d_nonfinite : nullptr);
  return SVSDF_OK;
}


// The stream batch b's chain runs on: its own, or -- with a single batch -- the main stream (round 5: a one-batch evaluation
// used to hop main -> batch -> main through two events, ~12 us each on the device timeline; a reference-scale callback is
// ~20 launches of a few us)
hipStream_t batch_stream(const svsdf_ctx *ctx, int b) { return ctx->nbatch > 1 ? ctx->bstream[b] : ctx->stream; }

int join_batches(svsdf_ctx *ctx) {
  if (!ctx->launch_err.empty()) {   // a launch that could not be made: a clear error instead of a raw HIP launch failure
    const std::string m = ctx->launch_err;
    ctx->launch_err.clear();
    for (int b = 0; b < ctx->nbatch; ++b) (void)hipStreamSynchronize(batch_stream(ctx, b));
    return fail(ctx, SVSDF_ERR_INVALID, m);
  }
  if (ctx->nbatch > 1)
    for (int b = 0; b < ctx->nbatch; ++b) {
      HIPCHK(hipEventRecord(ctx->ev_done[b], ctx->bstream[b]));
      HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_done[b], 0));
    }
  HIPCHK(hipGetLastError());
  return SVSDF_OK;
}

// GSIP pipeline per batch:  R0 S0 R1 S1 ... S(k-1) Rk   with
//   R_i = k_round(i): close the rounds whose samples S_(i-1) solved, open the next ones, select
//   S_i = k_solve over the samples R_i selected.
// enqueue_solve_round(i) enqueues S_i then R_(i+1).
void enqueue_solve_round(svsdf_ctx *ctx, int it, bool tail_next = false) {
  for (int b = 0; b < ctx->nbatch; ++b) {
    hipStream_t st = batch_stream(ctx, b);
    BatchCtl *ctl = ctx->d_ctl + b;
    QuerySet q{};
    q.qx = ctx->gs.sqx; q.qy = ctx->gs.sqy; q.count_ptr = &ctl->n_solve[it];
    q.slots = ctx->gs.solve + (size_t)ctx->bstart[b] * kMaxSlots; q.n_outer = 1;
    if (ctx->ub_full) { q.seed_k = ctx->gs.sq_k; q.seed_d = ctx->gs.sq_ub; }
    int G = (it >= ctx->late_iter) ? ctx->G_late : ctx->G;
    // Successive callbacks of one optimisation see almost the same trajectory: iteration `it` of the previous
    // evaluation tells how many solves this launch will hold.  Few solves = a latency-bound launch (<= ~1 wave per
    // SIMD): widen the groups to shorten the dependent chain; many solves = throughput: keep the shard's width even
    // in late iterations.  Any width gives the same bits, so a wrong guess only costs time.
    if (!ctx->G_env && !ctx->G_late_env && ctx->have_prev_nsolve) {
      const long long n = ctx->prev_nsolve[it];   // summed over the batches: they run this iteration concurrently
      G = ctx->G;
      if (n < ctx->wide32_below) G = std::max(G, 32);
      else if (n < ctx->wide16_below) G = std::max(G, 16);
      else if (n < ctx->wide8_below) G = std::max(G, 8);
    }
    launch_solve(ctx, G, st, q, (long long)ctx->bcount[b] * kMaxSlots, ctx->gs.sq_sdf, ctx->gs.sq_t, ctl, it + 1);
    if (tail_next) launch_tail(ctx, st, b, it + 1);
    else launch_round(ctx, st, b, it + 1);
  }
}

// Enqueue the device pipeline up to the per-point results of getTrueSDFofSweptVolume (res_*).
// No host synchronisation: batches run on their own streams, joined back onto ctx->stream.
int enqueue_queries(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, bool allow_cull) {
  if (ctx->host_only) return fail(ctx, SVSDF_ERR_NO_DEVICE, "host-only context: no device entry points");
  if (!ctx->points_set) return fail(ctx, SVSDF_ERR_NO_POINTS, "svsdf_set_points has not been called");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->ev_used = 0;
  ctx->refine_events.clear();
  ctx->round_events.clear();
  ctx->tail_events.clear();
  ctx->stats = svsdf_stats{};
  ctx->stats.points = ctx->P;
  const size_t e_begin = next_event(ctx);
  (void)hipEventRecord(ctx->ev_pool[e_begin], ctx->stream);
  int rc = upload_traj(ctx, N, coeffs, T, /*clear_nonfinite=*/true);   // (k_prep also clears the non-finite counter)
  if (rc) return rc;
  const bool fork = ctx->nbatch > 1;   // one batch: the whole chain stays on the main stream (no cross-stream hand-offs)
  if (fork) HIPCHK(hipEventRecord(ctx->ev_prep, ctx->stream));
  const int m = ctx->tail_iter = choose_tail_iter(ctx);
  for (int b = 0; b < ctx->nbatch; ++b) {
    hipStream_t st = batch_stream(ctx, b);
    BatchCtl *ctl = ctx->d_ctl + b;
    if (fork) HIPCHK(hipStreamWaitEvent(st, ctx->ev_prep, 0));
    QuerySet qm{};
    qm.qx = ctx->d_px; qm.qy = ctx->d_py; qm.stride = 0; qm.count_ptr = nullptr;
    qm.count_fixed = ctx->bcount[b]; qm.list = nullptr; qm.base = ctx->bstart[b]; qm.n_outer = 1;
    const double cull_thresh = (allow_cull && ctx->cull && ctx->cull_ok) ? ctx->cfg.safety_hor + 1e-9 : std::numeric_limits<double>::infinity();
    launch_solve(ctx, ctx->G, st, qm, ctx->bcount[b], ctx->d_sdf, ctx->d_t, ctl, 0, cull_thresh);
    launch_classify(ctx, st, b);
    if (m == 0) launch_tail(ctx, st, b, 0);
    else launch_round(ctx, st, b, 0);
  }
  if (m >= 0) {
    // launch chain up to iteration m, everything after it in k_tail: nothing is ever left pending
    for (int it = 0; it < m; ++it) enqueue_solve_round(ctx, it, it == m - 1);
    ctx->it_done = kMaxIter + 1;   // (an index whose pending-solve count is always zero)
  } else {
    for (int it = 0; it < ctx->first_iters; ++it) enqueue_solve_round(ctx, it);
    ctx->it_done = ctx->first_iters;
  }
  return join_batches(ctx);
}

// The swept-volume implicit function alone: getSDFofSweptVolume<false,true> (SWM:844-866) = min over t of the shape SDF,
// for every resident point, in the sorted shard order -- the main solve without cull, classification and GSIP rounds
// (svsdf_swept_outline: the zero set only needs the sign and a Lipschitz value next to it).
int swept_field(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, double *sdf_sorted) {
  if (ctx->host_only) return fail(ctx, SVSDF_ERR_NO_DEVICE, "host-only context: no device entry points");
  if (!ctx->points_set) return fail(ctx, SVSDF_ERR_NO_POINTS, "svsdf_set_points has not been called");
  if (ctx->P == 0) return SVSDF_OK;
  HIPCHK(hipSetDevice(ctx->device));
  ctx->ev_used = 0;
  ctx->refine_events.clear();
  ctx->round_events.clear();
  ctx->tail_events.clear();
  ctx->stats = svsdf_stats{};
  ctx->stats.points = ctx->P;
  int rc = upload_traj(ctx, N, coeffs, T);
  if (rc) return rc;
  if (ctx->nbatch > 1) HIPCHK(hipEventRecord(ctx->ev_prep, ctx->stream));
  for (int b = 0; b < ctx->nbatch; ++b) {
    hipStream_t st = batch_stream(ctx, b);
    if (ctx->nbatch > 1) HIPCHK(hipStreamWaitEvent(st, ctx->ev_prep, 0));
    QuerySet qm{};
    qm.qx = ctx->d_px; qm.qy = ctx->d_py; qm.stride = 0; qm.count_ptr = nullptr;
    qm.count_fixed = ctx->bcount[b]; qm.list = nullptr; qm.base = ctx->bstart[b]; qm.n_outer = 1;
    launch_solve(ctx, ctx->G, st, qm, ctx->bcount[b], ctx->d_sdf, ctx->d_t, ctx->d_ctl + b, 0);
  }
  rc = join_batches(ctx);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(sdf_sorted, ctx->d_sdf, ctx->P * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVSDF_OK;
}

// assemble + reduce + k_finish on the main stream, one D2H of [partial | counters], sync.
int reduce_and_read(svsdf_ctx *ctx, bool with_partial) {
  const int N = ctx->N;
  bool host_written = false;
  if (with_partial) {
    const unsigned grid = (unsigned)std::min<size_t>((ctx->P + kBlock - 1) / kBlock, 512);   // (256 / 1024 / 2048 blocks: within 1 %, round 6)
    const size_t plen = 19 * (size_t)N + 1;
    if ((size_t)grid * plen > ctx->block_partials_cap) {
      int rc = dev_alloc(ctx, &ctx->d_block_partials, (size_t)grid * plen);
      if (rc) return rc;
      ctx->block_partials_cap = (size_t)grid * plen;
    }
    // one accumulator row per wave; a cloud that fits ONE block keeps its points' terms in LDS instead (assemble_body)
    const size_t lds = ((size_t)traj_lds_doubles(N) + std::max<size_t>((kBlock / 64) * plen, grid == 1 ? svsdf::assemble_small_doubles(N) : 0)) * sizeof(double);
    if (lds > ctx->lds_limit)   // (98 KB at 128 pieces: fits gfx950's 160 KB; a clear error instead of a raw launch failure elsewhere)
      return fail(ctx, SVSDF_ERR_INVALID, "trajectory too long for the reduction's LDS accumulator rows (" + std::to_string(lds) + " B of " +
                  std::to_string(ctx->lds_limit) + " B per block)");
    // small clouds (<= 1024 points): assembly, block reduction, fixed-order final sum, suffix sum and counters in ONE
    // launch whose last block writes the result to the device buffer and straight into the pinned host buffer (k_reduce);
    // larger ones: the assembly, then k_final (one wave per entry) and k_finish as launches of their own
    const int fuse = grid <= 4 ? 1 : 0;
    hipLaunchKernelGGL(k_reduce, dim3(grid), dim3(kBlock), lds, ctx->stream, ctx->d_traj, ctx->d_px, ctx->d_py,
                       (int)ctx->P, ctx->d_res_sdf, ctx->d_res_t, ctx->d_res_gx, ctx->d_res_gy,
                       ctx->cfg.safety_hor, ctx->cfg.weight_p, ctx->d_block_partials, ctx->d_nonfinite, ctx->d_out,
                       ctx->d_ctl, ctx->nbatch, ctx->it_done, (int)kOutPartial, (int)kOutDoubles, ctx->d_ticket, ctx->h_out_dev, fuse);
    if (!fuse) {
      hipLaunchKernelGGL(k_final, dim3((unsigned)plen), dim3(64), 0, ctx->stream, ctx->d_block_partials, (int)grid, ctx->d_sums);
      hipLaunchKernelGGL(k_finish, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->d_sums, N, ctx->d_out, ctx->d_ctl,
                         ctx->nbatch, ctx->it_done, ctx->d_nonfinite,
                         reinterpret_cast<unsigned long long *>(ctx->d_out + kOutPartial), ctx->h_out_dev, (int)kOutDoubles);
    }
    host_written = ctx->h_out_dev != nullptr;
  } else {
    HIPCHK(hipMemsetAsync(ctx->d_sums, 0, kOutPartial * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->d_sums, N, ctx->d_out, ctx->d_ctl,
                       ctx->nbatch, ctx->it_done, ctx->d_nonfinite,
                       reinterpret_cast<unsigned long long *>(ctx->d_out + kOutPartial), ctx->h_out_dev, (int)kOutDoubles);
    host_written = ctx->h_out_dev != nullptr;
  }
  ctx->e_end = next_event(ctx);
  (void)hipEventRecord(ctx->ev_pool[ctx->e_end], ctx->stream);
  if (!host_written) HIPCHK(hipMemcpyAsync(ctx->h_out, ctx->d_out, kOutDoubles * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  return SVSDF_OK;
}

int alloc_interior_buffers(svsdf_ctx *ctx, size_t cap);   // below

// Finish an evaluation: reduce, read back, and -- rare slow path -- run more GSIP iterations when
// some points needed more than `first_iters` (rounds + supplementary solves), then reduce again.
int finish(svsdf_ctx *ctx, bool with_partial) {
  int rc = reduce_and_read(ctx, with_partial);
  if (rc) return rc;
  const unsigned long long *st = reinterpret_cast<const unsigned long long *>(ctx->h_out + kOutPartial);
  while (st[5] > 0 && ctx->it_done < kMaxIter) {  // solves requested by the last k_round are pending
    const int it1 = std::min(ctx->it_done + 3, (int)kMaxIter);
    for (int it = ctx->it_done; it < it1; ++it) enqueue_solve_round(ctx, it);
    ctx->it_done = it1;
    if ((rc = join_batches(ctx))) return rc;
    HIPCHK(hipMemsetAsync(ctx->d_nonfinite, 0, sizeof(int), ctx->stream));
    if ((rc = reduce_and_read(ctx, with_partial))) return rc;
  }
  if (st[5] > 0) return fail(ctx, SVSDF_ERR_INVALID, "GSIP iterations exhausted (internal limit)");
  {
    const size_t n_int = (size_t)st[11 + 2 * kMaxIter];
    if (n_int > ctx->icap) {   // surplus interior points were dropped: grow and repeat (same bits once everything fits)
      rc = alloc_interior_buffers(ctx, std::min(ctx->P, n_int + n_int / 8 + 4096));
      return rc ? rc : kRepeat;
    }
    if (!ctx->icap_fitted && ctx->P >= 100000 && ctx->icap > 2 * (n_int + n_int / 8 + 4096)) {
      // first evaluation of a large cloud: give back what the interior count does not need (contents are per evaluation)
      rc = alloc_interior_buffers(ctx, n_int + n_int / 8 + 4096);
      if (rc) return rc;
    }
    ctx->icap_fitted = true;
  }
  ctx->stats.solves = st[0];
  ctx->stats.sdf_evals = st[1];
  ctx->stats.scan_evals = st[2];
  ctx->stats.interior_points = st[3];
  ctx->stats.gsip_samples = st[6];
  ctx->stats.gsip_iterations = (unsigned)st[7];
  ctx->stats.culled_points = st[8];
  ctx->stats.round_scan_evals = st[9 + kMaxIter];
  ctx->stats.speculative_evals = st[10 + kMaxIter];
  ctx->stats.batches = ctx->nbatch;
  {
    const unsigned long long c = st[12 + 2 * kMaxIter], r = st[13 + 2 * kMaxIter];   // clock probe (BatchCtl::clk)
    ctx->stats.shader_clock_mhz = r ? (double)c / (double)r * ctx->wall_clock_khz * 1e-3 : 0.0;
  }
  for (int i = 0; i < kMaxIter; ++i) ctx->prev_nsolve[i] = (long long)st[9 + i];
  ctx->have_prev_nsolve = true;
  for (int i = 0; i < kMaxIter; ++i) ctx->prev_nactive[i] = (long long)st[11 + kMaxIter + i];
  ctx->prev_tail_iter = ctx->tail_iter;
  ctx->have_prev_nactive = true;
  ctx->stats.tail_iter = ctx->tail_iter;
  ctx->stats.tail_points = (ctx->tail_iter >= 0 && ctx->tail_iter < kMaxIter) ? (unsigned long long)st[11 + kMaxIter + ctx->tail_iter] : 0ull;
  // next evaluation enqueues as many iterations as this one needed (+1); the slow path above
  // covers an underestimate
  if (ctx->adaptive_iters && ctx->tail_iter < 0) ctx->first_iters = std::max(2, std::min((int)st[7] + 1, (int)kMaxIter));
  if (ctx->profile_span && !ctx->profile) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ctx->ev_pool[0], ctx->ev_pool[ctx->e_end]);
    ctx->stats.device_ms = ms;
  }
  if (ctx->profile) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ctx->ev_pool[0], ctx->ev_pool[ctx->e_end]);
    ctx->stats.device_ms = ms;
    // k_solve time of the evaluation: the point batches run concurrently on their own streams, so the launches'
    // [start, stop] intervals (device clock, relative to the evaluation's first event) are merged -- solve_ms is the
    // time during which at least one k_solve launch was executing, solve_ms_sum the plain sum of the launch durations
    // (what a kernel trace adds up)
    auto merged = [&](const std::vector<std::pair<size_t, size_t>> &evs, double &uni, double &sum) {
      sum = 0.0;
      std::vector<std::pair<float, float>> iv;
      for (const auto &pr : evs) {
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, ctx->ev_pool[0], ctx->ev_pool[pr.first]) == hipSuccess &&
            hipEventElapsedTime(&b, ctx->ev_pool[0], ctx->ev_pool[pr.second]) == hipSuccess && b >= a) {
          iv.emplace_back(a, b);
          sum += b - a;
        }
      }
      std::sort(iv.begin(), iv.end());
      uni = 0.0;
      float cur_a = 0.f, cur_b = -1.f;
      for (const auto &x : iv) {
        if (x.first > cur_b) { if (cur_b >= cur_a) uni += cur_b - cur_a; cur_a = x.first; cur_b = x.second; }
        else cur_b = std::max(cur_b, x.second);
      }
      if (cur_b >= cur_a) uni += cur_b - cur_a;
    };
    merged(ctx->refine_events, ctx->stats.solve_ms, ctx->stats.solve_ms_sum);
    merged(ctx->round_events, ctx->stats.round_ms, ctx->stats.round_ms_sum);
    merged(ctx->tail_events, ctx->stats.tail_ms, ctx->stats.tail_ms_sum);
  }
  if (st[4]) return fail(ctx, SVSDF_ERR_NONFINITE, "non-finite per-point result on the device");
  return SVSDF_OK;
}


// One evaluation up to the per-point results (and the partial): enqueue, finish.
int evaluate_points(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, bool allow_cull, bool with_partial) {
  for (int attempt = 0; attempt < 3; ++attempt) {
    int rc = enqueue_queries(ctx, N, coeffs, T, allow_cull);
    if (rc) return rc;
    rc = finish(ctx, with_partial);
    if (rc != kRepeat) return rc;   // kRepeat: the interior arrays were too small and have been grown
  }
  return fail(ctx, SVSDF_ERR_INVALID, "interior capacity could not be settled (internal error)");
}

void fill_mode_stats(svsdf_ctx *ctx) {
  ctx->stats.gsip_bound_mode = bound_mode_of(ctx);
  ctx->stats.piece_time_exact = ctx->stats_piece_time;
  ctx->stats.bound_mode_decided = (ctx->ub_env || ctx->ub_tune > 0) ? 1 : 0;
  ctx->stats.plan_settled = (ctx->stats.bound_mode_decided && ctx->bt_state == 0 && ctx->an_state == 0 && ctx->lz_state == 0 && ctx->have_prev_nsolve) ? 1 : 0;
  ctx->stats.bound_ratio = ctx->ub_ratio;
  ctx->stats.n_devices = 1;
  ctx->stats.combine = SVSDF_COMBINE_HOST;
  ctx->stats.combine_ms = 0.0;
  ctx->stats.setup_ms = ctx->setup_ms;
}

// Whole device pipeline of ONE device; leaves [cost, gradC, gradT] (19N+1 doubles) in d_out / h_out.
int run_pipeline_leaf(svsdf_ctx *ctx, int N, const double *coeffs, const double *T) {
  if (ctx->host_only) return fail(ctx, SVSDF_ERR_NO_DEVICE, "host-only context: no device entry points");
  if (!ctx->points_set) return fail(ctx, SVSDF_ERR_NO_POINTS, "svsdf_set_points has not been called");
  if (ctx->P == 0) {
    // an obstacle-free window (or an empty stripe): the reference's loop over parallel_points_num == 0 adds
    // nothing (BEO:785) and the callback returns energy + rho * sum(T).  The trajectory is still validated and
    // traj_duration updated (SWM:376-385); the partial is zero on the host and on the device (collectives).
    if (N < 1 || N > kMaxPieces) return fail(ctx, SVSDF_ERR_INVALID, "N out of range [1, 128]");
    double td = 0.0;
    for (int i = 0; i < N; ++i) {
      if (!std::isfinite(T[i])) return fail(ctx, SVSDF_ERR_NONFINITE, "non-finite duration");
      if (!(T[i] > 0.0)) return fail(ctx, SVSDF_ERR_INVALID, "piece durations must be positive");
      td += T[i];
    }
    for (size_t i = 0; i < 18 * (size_t)N; ++i)
      if (!std::isfinite(coeffs[i])) return fail(ctx, SVSDF_ERR_NONFINITE, "non-finite trajectory input");
    if (td < 3 * 1e2 || !ctx->have_duration) { ctx->traj_duration = td; ctx->have_duration = true; }
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemsetAsync(ctx->d_out, 0, kOutDoubles * sizeof(double), ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::memset(ctx->h_out, 0, kOutDoubles * sizeof(double));
    ctx->N = N;
    ctx->stats = svsdf_stats{};
    fill_mode_stats(ctx);
    ctx->h_partial = ctx->h_out;
    return SVSDF_OK;
  }
  // GSIP upper-bound mode.  The cheap bound (8 table poses of the nearest chunk) is enough for some shapes (star:
  // a quarter of the samples get solved, full scans only add work), useless for others (sdHorseshoe: 87 % of the
  // samples land in the selection band; with the sample's own table scan as the bound 5.3 -> 1.4 solves per
  // point, 2.4x overall).  Both modes return the same bits, so the choice only costs time.  Rule (deterministic,
  // one evaluation): the first evaluation after a new point set runs with the cheap bound; if it had to solve
  // more than half of the GSIP samples it emitted, every later evaluation scans.  (Round 1 timed three
  // evaluations per mode with the wall clock; the rule reproduces its choices on C1-C5 without the five extra
  // evaluations and without depending on the box.)
  const bool deciding = !ctx->ub_env && ctx->ub_tune == 0;
  if (deciding) { ctx->ub_full = false; ctx->ub_lazy = false; ctx->ub_anchor = false; ctx->an_state = 0; ctx->lz_state = 0; }
  // Anchor trial (full-scan shapes only): one evaluation in the full mode and one in the anchor mode, compared by their
  // table-evaluation COUNTERS (deterministic: no timing) -- the anchor mode stays when it saves at least 28 % of them
  // (sdHeart - 32 %: k_round - 27 %, evaluation - 12 %; Polygon - 25 %: a wash; sdHorseshoe - 10 %: + 5 %, its extra passes
  // cost more than the skipped scans; profiles/r04_anchor_ab.txt).  Both modes return the same bits.
  const int an_trial = (ctx->ub_env || ctx->saved_nbatch > 0) ? 0 : ctx->an_state;
  if (an_trial == 1) ctx->ub_anchor = false;
  if (an_trial == 2) ctx->ub_anchor = true;
  // Lazy trial (round 5; small clouds of a cheap-bound shape only, see the deciding block below): this evaluation runs in
  // the lazy mode and is judged by its GSIP solve COUNT against the cheap bound's (deterministic, no timing).
  const int lz_trial = (ctx->ub_env || ctx->saved_nbatch > 0) ? 0 : ctx->lz_state;
  if (lz_trial == 1) { ctx->ub_full = true; ctx->ub_lazy = true; }
  // Batch count (DESIGN.md "concurrent point batches").  Default: a RULE -- 3 batches from 80 k points per device, 1 below
  // (round 5; rounds 3 - 4: from 400 k in a scanning mode only; what the measurements -- 1 / 2 / 3 / 4 at NS, C3, C4 and at
  // 60 k ... 500 k points -- chose on every box; with the main stream that is at most 4 streams, the HIP runtime's default
  // number of hardware queues) -- so that the plan is the same on every run and settled after the deciding evaluation.  svsdf_set_plan
  // (batches = -1) / SVSDF_BATCHES=measure ask for a measurement instead: after one evaluation that learns the launch
  // widths, every candidate count (1, 4 [2 in the lazy mode], 3) runs three evaluations, timed with HIP events on the
  // library's own stream (device time, not the host's wall clock), and the best median stays.  Every count computes the
  // same bits.  While svsdf_set_profiling(ctx, 2) holds the batches serialised, nothing is measured or changed.
  int rc = SVSDF_OK;
  const bool timing = ctx->bt_state >= 2 && ctx->saved_nbatch == 0;
  if (timing) rc = set_batches(ctx, ctx->bt_cand[ctx->bt_k]);
  if (rc == SVSDF_OK) rc = evaluate_points(ctx, N, coeffs, T, /*allow_cull=*/true, /*with_partial=*/true);
  if (rc == SVSDF_OK && ctx->bt_state == 1 && ctx->saved_nbatch == 0) {
    ctx->bt_state = 2;
    ctx->bt_k = 0;
    ctx->bt_rep = 0;
  } else if (rc == SVSDF_OK && timing) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ctx->ev_pool[0], ctx->ev_pool[ctx->e_end]);
    ctx->bt_samples[ctx->bt_rep++] = ms;
    if (ctx->bt_rep == 3) {
      std::sort(ctx->bt_samples, ctx->bt_samples + 3);
      ctx->bt_ms[ctx->bt_k] = ctx->bt_samples[1];
      ctx->bt_rep = 0;
      if (++ctx->bt_k == ctx->bt_ncand) {
        int best = 0;
        for (int k = 1; k < ctx->bt_ncand; ++k)
          if (ctx->bt_ms[k] < ctx->bt_ms[best]) best = k;
        rc = set_batches(ctx, ctx->bt_cand[best]);
        ctx->bt_state = 0;
      }
    }
  }
  if (rc == SVSDF_OK && an_trial == 1) {
    ctx->an_full_evals = ctx->stats.round_scan_evals;
    ctx->an_state = 2;
  } else if (rc == SVSDF_OK && an_trial == 2) {
    const bool keep = (double)ctx->stats.round_scan_evals <= 0.72 * (double)ctx->an_full_evals;
    if (keep != ctx->ub_anchor) { ctx->have_prev_nsolve = false; ctx->have_prev_nactive = false; }
    ctx->ub_anchor = keep;
    ctx->an_state = 0;
  }
  if (lz_trial == 1) {
    const unsigned long long main_solves = ctx->stats.points - ctx->stats.culled_points;
    const unsigned long long gs = ctx->stats.solves > main_solves ? ctx->stats.solves - main_solves : 0ull;
    const bool keep = rc == SVSDF_OK && (double)gs <= 0.76 * (double)ctx->lz_cheap_solves;
    if (!keep) { ctx->ub_full = false; ctx->ub_lazy = false; }
    ctx->have_prev_nsolve = false;   // (the widths on record belong to the deciding evaluation / to the rejected mode; the interior count stands)
    ctx->lz_state = 0;
  }
  if (rc == SVSDF_OK && deciding) {
    const unsigned long long main_solves = ctx->stats.points - ctx->stats.culled_points;
    const unsigned long long gs = ctx->stats.solves > main_solves ? ctx->stats.solves - main_solves : 0ull;
    ctx->ub_ratio = ctx->stats.gsip_samples ? (double)gs / (double)ctx->stats.gsip_samples : 0.0;
    // Polygon: an SDF evaluation costs several times an analytic shape's (candidate edges of the outline), a table
    // scan proportionally less of a solve, so scanning pays from a lower ratio
    const double thr = ctx->cfg.shape_id == SVSDF_SHAPE_Polygon ? 0.2 : ctx->ub_threshold;
    // below the threshold a large shard still gains from scanning -- but only the samples the cheap bound would have
    // had solved (lazy mode: NS, star / 16 pieces / 1 M points, 12.3 -> 11.4 ms; at 100 k points no gain)
    const bool large = ctx->P >= 400000;
    ctx->ub_full = ctx->ub_ratio > thr || large;
    ctx->ub_lazy = !(ctx->ub_ratio > thr);
    // A SMALL cloud of a cheap-bound shape (round 5): when the fused tail owns the whole GSIP loop (choose_tail_iter: at most
    // 24 interior points per CU) an evaluation is a latency chain -- a scan costs lanes that idle anyway, every solve it
    // saves is a dozen dependent steps.  Whether the lazy scans save enough depends on the workload, not on the ratio above
    // (star: 16 pieces / 1 k - 10 k points - 8 ... - 20 %, the reference's star map 398 -> 367 us; 8 pieces / 10 k points
    // + 15 %: profiles/r05_small_cloud_modes.txt), so the next evaluation TRIES the lazy mode and stays in it when its GSIP
    // solves drop to <= 76 % of this evaluation's (the cases above: 68 - 72 % against 83 - 87 %).  Same bits in every mode.
    const long long tail_below = ctx->tail_below > 0 ? ctx->tail_below : (long long)ctx->n_cu * 24;
    if (!ctx->ub_full && ctx->tail_mode != -2 && (long long)ctx->stats.interior_points <= tail_below && gs > 0) {
      ctx->lz_state = 1;
      ctx->lz_cheap_solves = gs;
    }
    if (ctx->ub_full && !ctx->ub_lazy && ctx->lipschitz_ok) ctx->an_state = 1;   // full scans pay: does the anchor variant pay more?
    if (ctx->ub_full) { ctx->have_prev_nsolve = false; ctx->have_prev_nactive = false; }   // the launch plan on record is the cheap-bound one
  }
  // the batch count follows once the bound mode is known (also when it was pinned)
  if (rc == SVSDF_OK && ctx->ub_tune == 0) {
    ctx->ub_tune = 1;
    const bool big = ctx->ub_full && ctx->P >= 400000;
    if (ctx->want_batches == 0) {
      // Round 5 (profiles/r05_batches_by_size.txt): 3 batches pay from ~ 80 k points per device in EVERY bound mode -- 60 k: + 2 ... 9 %,
      // 100 k: + 9 ... 14 %, 150 - 250 k: + 4 ... 16 % (star / sdHorseshoe / sdHeart; 2 batches always in between, 4 lose at
      // 500 k) -- not only from 400 k in the scanning modes (round 4's rule, measured at 1 M points only): the chain's ~ 20
      // dependent launches leave the chip draining at every step, and another batch's kernels fill those tails whatever the
      // bound mode.  Matters for BASELINE config 2 (100 k points) and for the stripes of a multi-GPU run (C4 / 8 = 500 k, NS / 8 = 125 k).
      const int nb = (ctx->P >= 80000) ? 3 : 1;
      if (ctx->saved_nbatch > 0) ctx->saved_nbatch = nb;        // (serialised for profiling: takes effect when that ends)
      else if (nb != ctx->nbatch) rc = set_batches(ctx, nb);
    } else if (ctx->want_batches < 0 && big) {
      ctx->bt_state = 1;
      ctx->bt_cand[0] = 1;
      ctx->bt_cand[1] = ctx->ub_lazy ? 2 : 4;
      ctx->bt_cand[2] = 3;
      ctx->bt_ncand = 3;
    }
  }
  fill_mode_stats(ctx);
  ctx->h_partial = ctx->h_out;
  return rc;
}


void accumulate(int N, const double *partial, double *cost, double *gradT, double *gradC) {
  *cost += partial[0];
  for (int e = 0; e < 18 * N; ++e) gradC[e] += partial[1 + e];
  for (int j = 0; j < N; ++j) gradT[j] += partial[1 + 18 * N + j];
}

// Interior-sized arrays: GSIP state per interior point and the 24 sample slots per interior point (the big ones: 24 x 52 B).
int alloc_interior_buffers(svsdf_ctx *ctx, size_t cap) {
  int rc = 0;
  cap = std::max<size_t>(cap, 1);
  if (cap * kMaxSlots > 0x7fffffffull) return fail(ctx, SVSDF_ERR_INVALID, "too many interior points per shard (max ~89M)");
  if ((rc = dev_alloc(ctx, &ctx->gs.pt, cap))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.r, cap))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.theta0, cap))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.theta_res, cap))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.iter, cap))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.nsamp, cap))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.phase, cap))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.req, cap))) return rc;
  const size_t S = cap * kMaxSlots;
  if ((rc = dev_alloc(ctx, &ctx->gs.sqx, S))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.sqy, S))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.sqth, S))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.sq_ub, S))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.sq_k, S))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.sq_sdf, S))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.sq_t, S))) return rc;
  ctx->icap = cap;
  return SVSDF_OK;
}

int alloc_point_buffers(svsdf_ctx *ctx, size_t P) {
  int rc = 0;
  if ((rc = dev_alloc(ctx, &ctx->d_px, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_py, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_sdf, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_t, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_res_sdf, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_res_t, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_res_gx, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_res_gy, P))) return rc;
  // per batch, in the batch's range of the point arrays: the two active lists (compact interior indices) and the
  // solve list (up to kMaxSlots requests per point of the batch)
  if ((rc = dev_alloc(ctx, &ctx->gs.list[0], P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.list[1], P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.solve, P * kMaxSlots))) return rc;
  // the interior-sized arrays start at half the cloud (a corridor cloud has 25 - 40 % interior points) and are fitted
  // to the measured count after the first evaluation; an evaluation that needs more grows them and runs again
  ctx->icap_fitted = false;
  size_t cap0 = std::min(P, std::max<size_t>(P / 2, 4096));
  if (const char *e = std::getenv("SVSDF_ICAP_INIT")) cap0 = std::max<size_t>(1, (size_t)std::atoll(e));   // tests: force the grow-and-repeat path
  return alloc_interior_buffers(ctx, cap0);
}

// Morton order of the cloud (pure host): original indices sorted by (Morton code of the quantised xy, input index).
// The 64 lanes of a wave then hold spatially adjacent points (similar t*, similar iteration counts, same
// interior/exterior class); the sum is order-independent.
void morton_order(const double *xyz, size_t P, int flags, std::vector<long long> &order) {
  order.resize(P);
  std::iota(order.begin(), order.end(), 0ll);
  if ((flags & SVSDF_FLAG_KEEP_INPUT_ORDER) || P <= 1) return;
  double xmin = std::numeric_limits<double>::infinity(), xmax = -xmin, ymin = xmin, ymax = -xmin;
  for (size_t i = 0; i < P; ++i) {
    const double x = xyz[3 * i], y = xyz[3 * i + 1];
    if (x < xmin) xmin = x;
    if (x > xmax) xmax = x;
    if (y < ymin) ymin = y;
    if (y > ymax) ymax = y;
  }
  const double ext = std::max(std::max(xmax - xmin, ymax - ymin), 1e-12);
  std::vector<uint64_t> key(P);
  for (size_t i = 0; i < P; ++i) {
    const double fx = (xyz[3 * i] - xmin) / ext, fy = (xyz[3 * i + 1] - ymin) / ext;
    const uint32_t qx = (uint32_t)std::min(65535.0, std::max(0.0, fx * 65535.0));
    const uint32_t qy = (uint32_t)std::min(65535.0, std::max(0.0, fy * 65535.0));
    key[i] = ((uint64_t)(part1by1(qx) | (part1by1(qy) << 1)) << 32) | (uint64_t)(i & 0xffffffffu);
  }
  // LSD radix sort of the 32 Morton bits, 4 stable 8-bit passes over keys that start in index order (the low 32
  // bits carry the index and are never a sort digit)
  std::vector<uint64_t> tmp(P);
  for (int pass = 0; pass < 4; ++pass) {
    const int sh = 32 + 8 * pass;
    size_t cnt[257] = {0};
    for (size_t i = 0; i < P; ++i) ++cnt[((key[i] >> sh) & 0xffu) + 1];
    for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
    for (size_t i = 0; i < P; ++i) tmp[cnt[(key[i] >> sh) & 0xffu]++] = key[i];
    key.swap(tmp);
  }
  for (size_t i = 0; i < P; ++i) order[i] = (long long)(key[i] & 0xffffffffull);
}

// Stripe `rk` of `ws` of a Morton order: stripes, not blocks, so that every rank gets a spatially uniform
// subsample (interior points cost up to ~150x an exterior one and cluster in space).
void stripe_of(const std::vector<long long> &order, int rk, int ws, std::vector<long long> &out) {
  ws = std::max(1, ws);
  out.clear();
  out.reserve(order.size() / (size_t)ws + 1);
  for (size_t k = (size_t)rk; k < order.size(); k += (size_t)ws) out.push_back(order[k]);
}

// which original indices rank `rk` of `ws` owns, in device order (also exported as svsdf_shard_plan)
void shard_plan(const double *xyz, size_t P, int rk, int ws, int flags, std::vector<long long> &out) {
  if (P > 0xffffffffull) { out.clear(); return; }
  std::vector<long long> order;
  morton_order(xyz, P, flags, order);
  stripe_of(order, rk, ws, out);
}

// Split the resident shard into nb contiguous batches of the sorted cloud, each running the whole launch chain on its
// own stream.  Large shards in the full-scan GSIP mode use 4: that chain alternates k_round (table scans) and k_solve
// launches of similar weight, ~10 dependent pairs, each ending in a tail where the chip drains; another batch's
// kernels fill those tails (1 M points: sdHorseshoe / 32 pieces 9.3 -> 8.6 ms, sdHeart 11.3 -> 10.5 ms).  In the
// cheap-bound mode (star) the solves dominate and splitting buys nothing (12.0 vs 12.0-12.7 ms); small shards are
// latency-bound chains, splitting only adds launches.  Any split gives the same bits.
int set_batches(svsdf_ctx *ctx, int nb) {
  nb = std::max(1, std::min(nb, kMaxBatches));
  const size_t Ps = ctx->P;
  ctx->nbatch = nb;
  std::vector<BatchCtl> hc(kMaxBatches);
  std::memset(hc.data(), 0, sizeof(BatchCtl) * kMaxBatches);
  // (Round 6 measured batches that are STRIPES of the Morton order -- points b, b + nb, ... -- instead of contiguous parts:
  // their launch chains then weigh the same and end together, where the contiguous thirds of C3 end 300 us apart.  It is 3 - 5 %
  // SLOWER (C3 5.60 -> 5.77 ms, a 500 k-point C4 stripe 3.61 -> 3.79): identical chains run in lockstep, all in k_round or all
  // in k_solve at the same time, and the overlap of unlike phases is what the concurrent batches are for.)
  for (int b = 0; b < nb; ++b) {
    const size_t s = Ps * (size_t)b / nb, e = Ps * (size_t)(b + 1) / nb;
    ctx->bstart[b] = (int)s;
    ctx->bcount[b] = (int)(e - s);
    hc[b].start = (int)s;
    hc[b].count = (int)(e - s);
  }
  HIPCHK(hipMemcpy(ctx->d_ctl, hc.data(), sizeof(BatchCtl) * kMaxBatches, hipMemcpyHostToDevice));
  return SVSDF_OK;
}

// Plan of a cloud ON THE DEVICE: d_xyz (AoS, P x 3 doubles, on the planner's device) -> bounding box -> Morton keys ->
// radix sort (hipcub, all 64 bits of (Morton code << 32 | input index): the same order as the host planner's stable
// sort by Morton code; a partial bit range [32, 64) came back unsorted).  Same key formula as the host planner
// (svsdf_shard_plan), so both give the same order (tests/test_points_upload_gpu.py).  1 M points: ~2 ms on the device
// (+ ~4 ms PCIe when the cloud comes from host memory) against 29 ms for round 1's host radix sort.
// The plan is made ONCE per cloud; every context of a multi-device group then takes its stripe from it (take_stripe).
void CloudPlan::release() {
  (void)hipSetDevice(device);
  for (void *q : {(void *)d_part, (void *)d_keys, (void *)d_keys2, d_tmp})
    if (q) (void)hipFree(q);
  d_part = nullptr; d_keys = d_keys2 = nullptr; d_tmp = nullptr; sorted = nullptr;
}

int plan_cloud(svsdf_ctx *ctx, const double *d_xyz, size_t P, CloudPlan &plan) {
  plan.device = ctx->device;
  plan.d_xyz = d_xyz;
  plan.P = P;
  if (P > 0x7fffffffull) return fail(ctx, SVSDF_ERR_INVALID, "too many points (max 2^31 - 1 per call)");
  if (P == 0) return SVSDF_OK;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
#define UPCHK(expr)                                                                                                  \
  do {                                                                                                               \
    hipError_t e_ = (expr);                                                                                          \
    if (e_ != hipSuccess) { plan.release(); return fail(ctx, SVSDF_ERR_HIP_BASE + (int)e_, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
  } while (0)
  const unsigned grid = (unsigned)std::min<size_t>((P + kBlock - 1) / kBlock, 1024);
  UPCHK(hipMalloc((void **)&plan.d_part, (size_t)grid * 4 * sizeof(double)));
  UPCHK(hipMalloc((void **)&plan.d_keys, P * sizeof(unsigned long long)));
  UPCHK(hipMalloc((void **)&plan.d_keys2, P * sizeof(unsigned long long)));
  UPCHK(hipMemsetAsync(ctx->d_nonfinite, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_points_bbox, dim3(grid), dim3(kBlock), 0, st, d_xyz, P, plan.d_part, ctx->d_nonfinite);
  std::vector<double> hp((size_t)grid * 4);
  int bad = 0;
  UPCHK(hipMemcpyAsync(hp.data(), plan.d_part, hp.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  UPCHK(hipMemcpyAsync(&bad, ctx->d_nonfinite, sizeof(int), hipMemcpyDeviceToHost, st));
  UPCHK(hipStreamSynchronize(st));
  if (bad) { plan.release(); return fail(ctx, SVSDF_ERR_NONFINITE, "non-finite query point"); }
  double xmin = hp[0], xmax = hp[1], ymin = hp[2], ymax = hp[3];
  for (unsigned b = 1; b < grid; ++b) {
    xmin = std::min(xmin, hp[4 * b]); xmax = std::max(xmax, hp[4 * b + 1]);
    ymin = std::min(ymin, hp[4 * b + 2]); ymax = std::max(ymax, hp[4 * b + 3]);
  }
  const double ext = std::max(std::max(xmax - xmin, ymax - ymin), 1e-12);
  const int keep = ((ctx->cfg.flags & SVSDF_FLAG_KEEP_INPUT_ORDER) || P <= 1) ? 1 : 0;
  hipLaunchKernelGGL(k_points_keys, dim3(grid), dim3(kBlock), 0, st, d_xyz, P, xmin, ymin, ext, keep, plan.d_keys);
  plan.sorted = plan.d_keys;
  if (!keep) {
    size_t tmp_bytes = 0;
    UPCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, plan.d_keys, plan.d_keys2, (int)P, 0, 64, st));
    UPCHK(hipMalloc(&plan.d_tmp, std::max<size_t>(tmp_bytes, 16)));
    UPCHK(hipcub::DeviceRadixSort::SortKeys(plan.d_tmp, tmp_bytes, plan.d_keys, plan.d_keys2, (int)P, 0, 64, st));
    plan.sorted = plan.d_keys2;
  }
  UPCHK(hipStreamSynchronize(st));
  UPCHK(hipGetLastError());
#undef UPCHK
  return SVSDF_OK;
}

// Context `ctx` takes stripe (rk, ws) of a planned cloud: gather into its SoA arrays + original indices.  The planner
// may be another context (multi-device group): then the stripe is gathered on the planner's device into a staging
// buffer and handed over with hipMemcpyPeer (over xGMI between two GPUs; the same call when both are one device).
int take_stripe(svsdf_ctx *ctx, svsdf_ctx *planner, const CloudPlan &plan, int rk, int ws) {
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());
  ws = std::max(1, ws);
  const size_t P = plan.P;
  // until the new cloud is completely in place the context holds none (a failed upload must not leave the old point
  // count with re-allocated buffers)
  ctx->P = 0;
  ctx->points_set = false;
  ctx->saved_nbatch = 0;
  const size_t Ps = (P > (size_t)rk) ? (P - (size_t)rk + (size_t)ws - 1) / (size_t)ws : 0;
  if (Ps * kMaxSlots > 0x7fffffffull) return fail(ctx, SVSDF_ERR_INVALID, "too many points per shard (max ~89M)");
  ctx->shard_idx.assign(Ps, 0ll);
  int rc = alloc_point_buffers(ctx, Ps);
  if (rc) return rc;
  if (Ps > 0) {
    const bool local = planner == ctx;
    HIPCHK(hipSetDevice(planner->device));
    hipStream_t st = planner->stream;
    long long *d_idx = nullptr;
    double *d_sx = nullptr, *d_sy = nullptr;
    auto cleanup = [&]() {
      (void)hipSetDevice(planner->device);
      for (void *q : {(void *)d_idx, (void *)d_sx, (void *)d_sy})
        if (q) (void)hipFree(q);
    };
#define UPCHK(expr)                                                                                                  \
  do {                                                                                                               \
    hipError_t e_ = (expr);                                                                                          \
    if (e_ != hipSuccess) { cleanup(); return fail(ctx, SVSDF_ERR_HIP_BASE + (int)e_, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
  } while (0)
    UPCHK(hipMalloc((void **)&d_idx, Ps * sizeof(long long)));
    if (!local) {
      UPCHK(hipMalloc((void **)&d_sx, Ps * sizeof(double)));
      UPCHK(hipMalloc((void **)&d_sy, Ps * sizeof(double)));
    }
    const unsigned g2 = (unsigned)std::min<size_t>((Ps + kBlock - 1) / kBlock, 1024);
    hipLaunchKernelGGL(k_points_gather, dim3(g2), dim3(kBlock), 0, st, plan.d_xyz, plan.sorted, P, rk, ws, Ps,
                       local ? ctx->d_px : d_sx, local ? ctx->d_py : d_sy, d_idx);
    UPCHK(hipMemcpyAsync(ctx->shard_idx.data(), d_idx, Ps * sizeof(long long), hipMemcpyDeviceToHost, st));
    UPCHK(hipStreamSynchronize(st));
    UPCHK(hipGetLastError());
    if (!local) {
      UPCHK(hipMemcpyPeer(ctx->d_px, ctx->device, d_sx, planner->device, Ps * sizeof(double)));
      UPCHK(hipMemcpyPeer(ctx->d_py, ctx->device, d_sy, planner->device, Ps * sizeof(double)));
    }
#undef UPCHK
    cleanup();
  }
  HIPCHK(hipSetDevice(ctx->device));
  ctx->P = Ps;
  ctx->points_set = true;
  // batches: contiguous ranges of the sorted shard, pipelined on separate streams (one until the GSIP bound mode is
  // known, see set_batches)
  int rcb = set_batches(ctx, ctx->want_batches > 0 ? ctx->want_batches : 1);   // (rule / measurement: after the bound mode, run_pipeline_leaf)
  if (rcb) return rcb;
  // lanes per query: an evaluation is a chain of ~10 dependent solve launches, each a chain of ~100 dependent
  // group steps.  Small shards cannot fill the GPU and are pure latency: wide groups shorten the chains
  // (32 lanes: a whole halving ladder / scan layer per step).  Large shards are throughput: narrow groups waste
  // fewer lanes (2 since round 3: the ladders share the wave's lanes anyway, so the width only shapes the scan layers and
  // the derivative).  Measured crossovers (tools/latency.py, tools/sweep.py): 3e3, 2e4, 3e5 points.
  ctx->have_prev_nsolve = false;
  ctx->have_prev_nactive = false;
  ctx->prev_tail_iter = -1;
  ctx->ub_tune = 0;
  ctx->bt_state = 0;
  ctx->ub_ratio = 0.0;
  if (!ctx->ub_env) { ctx->ub_full = false; ctx->ub_lazy = false; ctx->ub_anchor = false; }
  ctx->an_state = 0;
  ctx->lz_state = 0;
  if (!ctx->G_env) {
    // (Polygon: 4 -- its 2-lane kernel spills 52 registers under the 3-waves cap; C5 36.5 vs 34.8 ms)
    ctx->G = default_lanes(ctx, Ps);
    if (!ctx->G_late_env) ctx->G_late = std::max(ctx->G, 8);
  }
  return SVSDF_OK;
}

// single-device context: plan + its own stripe
int upload_shard_device(svsdf_ctx *ctx, const double *d_xyz, size_t P, int rk, int ws) {
  CloudPlan plan;
  int rc = plan_cloud(ctx, d_xyz, P, plan);
  if (rc) { ctx->P = 0; ctx->points_set = false; return rc; }
  rc = take_stripe(ctx, ctx, plan, rk, ws);
  plan.release();
  return rc;
}

// svsdf_debug_sdf_at: upload the trajectory (same path as an evaluation: k_prep, piece-time mode), evaluate on the device
int debug_sdf_at(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, size_t n, const double *pxy, const double *t,
                 double *out8) {
  if (n == 0) return SVSDF_OK;
  if (n > 0x7fffffffull) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_debug_sdf_at: too many queries");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->ev_used = 0;
  const size_t e0 = next_event(ctx);
  (void)hipEventRecord(ctx->ev_pool[e0], ctx->stream);
  int rc = upload_traj(ctx, N, coeffs, T);
  if (rc) return rc;
  double *d_p = nullptr, *d_t = nullptr, *d_o = nullptr;
  auto cleanup = [&]() { for (void *q : {(void *)d_p, (void *)d_t, (void *)d_o}) if (q) (void)hipFree(q); };
  hipError_t e = hipMalloc((void **)&d_p, 2 * n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void **)&d_t, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void **)&d_o, 8 * n * sizeof(double));
  if (e == hipSuccess) e = hipMemcpyAsync(d_p, pxy, 2 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_t, t, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    const size_t lds = (size_t)traj_lds_doubles(N) * sizeof(double);
    if (!launch_k_debug_sdf_at(ctx->cfg.shape_id, (unsigned)((n + 63) / 64), lds, ctx->stream, ctx->d_traj, ctx->sp, d_p, d_t, (int)n, d_o)) {
      cleanup();
      return fail(ctx, SVSDF_ERR_INVALID, "svsdf_debug_sdf_at: shape not compiled into this build");
    }
    e = hipGetLastError();   // a failed launch is this function's error, not the next evaluation's
    if (e == hipSuccess) e = hipMemcpyAsync(out8, d_o, 8 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  cleanup();
  if (e != hipSuccess) return fail(ctx, SVSDF_ERR_HIP_BASE + (int)e, std::string("svsdf_debug_sdf_at: ") + hipGetErrorString(e));
  return SVSDF_OK;
}

// mismatch count of the kernels' inlined sincos against the device library's (svsdf_debug_sincos_mismatches)
long long sincos_mismatches(svsdf_ctx *ctx, double lo, double hi, int n) {
  if (hipSetDevice(ctx->device) != hipSuccess) return -1;
  unsigned long long *d = reinterpret_cast<unsigned long long *>(ctx->d_out);
  if (hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream) != hipSuccess) return -1;
  hipLaunchKernelGGL(k_sincos_check, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, lo, hi, n, d);
  unsigned long long h = 0;
  if (hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return -1;
  return (long long)h;
}

}  // namespace svsdf_impl
