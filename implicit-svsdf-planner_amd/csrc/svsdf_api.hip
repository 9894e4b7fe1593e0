// svsdf_api.hip -- C-ABI host layer (include/svsdf_c.h) over the gfx950 kernels.
//
// Mirrors, for this one path, the state and call order of the reference's
// TrajOptimizer::costFunctionLmbmParallel (BEO:344-408) and
// addSaftyPenaOnSweptVolumeParallelTrueSDF (BEO:774-869); see include/svsdf_c.h for the
// per-entry-point citations.  No CPU fallback: without a HIP device every compute entry fails.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/svsdf_c.h"
#define SVSDF_API_TU   // this translation unit compiles the shape-independent kernels of svsdf_kernels.hpp
#include "svsdf_launch.hpp"
#include "svsdf_lbfgs.hpp"
#include "svsdf_mesh.hpp"
#include "svsdf_contour.hpp"
#include "svsdf_minco.hpp"
#include "svsdf_points.hpp"

using namespace svsdf;

namespace {
thread_local std::string g_last_error;
}

// One host thread per device of an in-process multi-GPU context.
struct Worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = true, quit = false;
  int rc = 0;
  Worker() {
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(m);
      for (;;) {
        cv.wait(lk, [this] { return has_job || quit; });
        if (quit) return;
        std::function<int()> f = std::move(job);
        has_job = false;
        lk.unlock();
        const int r = f();
        lk.lock();
        rc = r;
        done = true;
        cv.notify_all();
      }
    });
  }
  void post(std::function<int()> f) {
    std::lock_guard<std::mutex> lk(m);
    job = std::move(f);
    has_job = true;
    done = false;
    cv.notify_all();
  }
  int wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [this] { return done; });
    return rc;
  }
  ~Worker() {
    { std::lock_guard<std::mutex> lk(m); quit = true; cv.notify_all(); }
    if (th.joinable()) th.join();
  }
};

struct svsdf_ctx {
  svsdf_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;               // main stream: upload, prep, assemble, readback
  hipStream_t bstream[kMaxBatches] = {};      // one stream per point batch
  hipEvent_t ev_prep = nullptr, ev_done[kMaxBatches] = {};
  ShapeParams sp{};
  bool poly_lds = false;             // Polygon: k_solve / k_round run their kPolygonLds variants (edges at the start of LDS)
  unsigned char *d_poly = nullptr;   // Polygon: one device blob [PolyAccel | edges | cell records | slab records | long lists]
  std::vector<double> poly_xy;       // Polygon: the outline as given (host copy)
  std::string err;

  // points: this rank's shard, Morton-sorted, split into nbatch contiguous batches
  size_t P = 0;
  bool points_set = false;           // svsdf_set_points was called (P may be 0: an obstacle-free window)
  std::vector<long long> shard_idx;  // original index of shard element j
  double *d_px = nullptr, *d_py = nullptr;
  int nbatch = 1;
  int bstart[kMaxBatches] = {}, bcount[kMaxBatches] = {};
  BatchCtl *d_ctl = nullptr;

  // trajectory
  TrajDev *d_traj = nullptr;
  double *d_in = nullptr;  // device staging: coeffs (18N) | T (N) | tk (K)
  double *h_in = nullptr;  // pinned mirror
  size_t in_cap = 0;       // doubles
  Pose *d_pose = nullptr;
  Chunk *d_chunks = nullptr;
  size_t pose_cap = 0;
  double r_bound = 0.0;    // shape bound radius for the layer-1 chunk pruning (analytic circumradius + offset)
  double r_bound_sampled = 0.0;  // max over a polar grid of |q| - sdf(q): self-check, must not exceed r_bound
  double traj_duration = 0.0;
  bool have_duration = false;
  bool host_only = false;  // SVSDF_FLAG_HOST_ONLY: MINCO / callback host logic only, no device
  int N = 0, K = 0;
  int piece_time_mode = 0;     // this trajectory: 0 cumulative form (exactly equivalent here), 1 / 2 faithful chain
  int stats_piece_time = 0;

  // tuning (env SVSDF_G / SVSDF_G_LATE / SVSDF_PRUNE / SVSDF_BLOCK / SVSDF_BATCHES)
  int G = 0 /* 0 = by shard size */, G_late = 8, prune = 1, block = 64, want_batches = 0, waves_per_cu = 16;
  bool block_env = false;      // env SVSDF_BLOCK pins the solve kernel's block size (default: by LDS footprint)
  int late_iter = 4, first_iters = 12, it_done = 0, round_lp8_iters = 3, delta_all_iter = 5;
  int n_cu = 256;
  bool adaptive_iters = true;
  bool ub_full = false;        // k_round scans every new GSIP sample (seed = tightest layer-1 bound, reused by k_solve)
  bool ub_lazy = false;        // with ub_full: only the samples in the cheap-bound band are scanned (k_round MODE 2)
  bool ub_env = false;         // env SVSDF_UB_FULL=0/1/2 pins the mode, otherwise run_pipeline decides after one evaluation
  int ub_tune = 0;             // evaluations since the point set changed that took part in the decision (0 or 1)
  double ub_ratio = 0.0;       // GSIP solves / GSIP samples of the deciding (cheap-bound) evaluation
  double ub_threshold = 0.5;   // env SVSDF_UB_RATIO (analytic shapes; Polygon 0.2)
  bool ub_thr_env = false;
  int round_list = 3;          // k_round: per-point candidate-chunk lists (env SVSDF_ROUND_LIST: bit 0 scans, bit 1 cheap bound use them; 0: all chunks; same results)
  bool cull = true;            // exact cull of provably inactive points in the main solve (env SVSDF_CULL=0 disables)
  bool cull_ok = false;        // this trajectory: duration not stale, slack table valid
  int G_env = 0, G_late_env = 0;
  // batch count of a large shard in the scanning bound modes: chosen by timing real evaluations (any split gives the
  // same bits): 0 idle / done, 1 next evaluation learns the launch plan with one batch, 2.. timing candidate bt_k
  int bt_state = 0, bt_k = 0, bt_rep = 0, bt_ncand = 0, bt_cand[3] = {1, 1, 1};
  double bt_ms[3] = {0, 0, 0}, bt_samples[3] = {0, 0, 0};
  long long prev_nsolve[kMaxIter] = {};  // solves per GSIP iteration of the previous evaluation (same point set)
  bool have_prev_nsolve = false;
  // fused GSIP tail (k_tail): all iterations from tail_iter on in one launch per batch
  int tail_mode = -1;                    // -1: by the previous evaluation's active counts, -2: off (launch chain only), >= 0: pinned
  long long tail_below = 16384;          // auto: the whole GSIP loop runs in k_tail when the shard has at most this many interior points
  int tail_all_after = 1 << 30;          // steps of a point inside k_tail after which every sample is requested (-1: like the chain)
  int tail_iter = -1;                    // this evaluation: iteration the tail starts at (-1: none)
  long long prev_nactive[kMaxIter] = {}; // active GSIP points per iteration of the previous evaluation, up to its tail
  int prev_tail_iter = -1;
  bool have_prev_nactive = false;
  long long wide32_below = 2000, wide16_below = 5000, wide8_below = 40000;  // env SVSDF_WIDE32 / SVSDF_WIDE16 / SVSDF_WIDE8
  bool select_env = false, all_iter_env = false;
  double select_delta = 0.1;  // k_round: solve the samples whose upper bound is within this of the best one first

  // per-point / per-sub-query buffers
  double *d_sdf = nullptr, *d_t = nullptr;
  double *d_res_sdf = nullptr, *d_res_t = nullptr, *d_res_gx = nullptr, *d_res_gy = nullptr;
  GsipState gs{};
  double *d_block_partials = nullptr;
  size_t block_partials_cap = 0;  // doubles
  double *d_sums = nullptr;       // 19 * kMaxPieces + 1
  double *d_out = nullptr;        // [partial (19 * kMaxPieces + 1) | 8 x u64 stats]
  double *h_out = nullptr;        // pinned mirror
  int *d_nonfinite = nullptr;
  int h_nonfinite = 0;
  size_t e_end = 0;

  // front-end batches (row f3): growing device scratch [father | child | pts | kt] + offsets + flags
  double *d_fe = nullptr, *h_fe = nullptr;   // device buffer + pinned staging mirror
  size_t fe_cap = 0;                          // doubles
  int *d_fe_flag = nullptr;
  std::vector<int> h_fe_flag;
  size_t fe_edges_cap = 0;

  // profiling
  bool profile = false;  // per-launch HIP events (env SVSDF_PROFILE=1 or svsdf_set_profiling)
  int saved_nbatch = 0;  // svsdf_set_profiling(ctx, 2): the batch split to restore
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<std::pair<size_t, size_t>> refine_events;  // k_solve launches: (start, stop) indices into ev_pool
  std::vector<std::pair<size_t, size_t>> round_events;   // k_round launches
  std::vector<std::pair<size_t, size_t>> tail_events;    // k_tail launches
  svsdf_stats stats{};

  // in-process multi-GPU group (svsdf_config::n_devices > 1): this context then owns no device state of its
  // own, only the host-side callback state below; subs[k] is the single-device context of stripe k
  std::vector<svsdf_ctx *> subs;
  std::vector<std::unique_ptr<Worker>> workers;   // one host thread per sub-context
  int combine = 0;                  // SVSDF_COMBINE_HOST / SVSDF_COMBINE_RCCL (resolved)
  std::vector<void *> comms;        // ncclComm_t per sub-context (RCCL combine)
  std::vector<double *> d_red;      // per sub-context all-reduce output (RCCL combine)
  double *h_red = nullptr;          // pinned: reduced partial read back from subs[0]
  std::vector<double> comb;         // host-combined [cost | gradC | gradT]
  const double *h_partial = nullptr;  // where the last evaluation's summed partial lives on the host
  double combine_ms = 0.0, setup_ms = 0.0;

  // svsdf_swept_outline: the last result, so that the documented query-then-fill protocol runs the extraction once
  std::vector<double> ol_key, ol_xy;
  std::vector<int> ol_loops;
  svsdf_outline_stats ol_stats{};
  bool ol_valid = false;
  size_t lds_limit = 65536;         // dynamic LDS a block may ask for on this device
  std::string launch_err;           // a launch that could not be made (LDS budget, shape not compiled); reported by join_batches

  // full-callback state (TrajOptimizer members BEO:44-60)
  svsdf_host::MincoS3 minco;
  std::vector<double> T, pgC, pgT, cm, gC, gradq, gradT, xlast;
  double energy_cost = 0.0;
  double costs3[3] = {0, 0, 0};
};

namespace {

constexpr size_t kOutPartial = 19 * kMaxPieces + 1;
constexpr size_t kOutDoubles = kOutPartial + 11 + 2 * kMaxIter;   // partial | 9 counters | solves per iteration | round scans | speculative | active per iteration

int fail(svsdf_ctx *ctx, int code, const std::string &msg) {
  g_last_error = msg;
  if (ctx) ctx->err = msg;
  return code;
}

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(ctx, SVSDF_ERR_HIP_BASE + (int)e_, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

template <typename T>
int dev_alloc(svsdf_ctx *ctx, T **p, size_t count) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (count == 0) count = 1;
  HIPCHK(hipMalloc((void **)p, count * sizeof(T)));
  return SVSDF_OK;
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
}  // namespace
namespace svsdf {
bool launch_k_solve(int shape, int G, unsigned grid, unsigned block, size_t lds, hipStream_t st, const SolveLaunch &a) {
  shape = compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_solve, G, grid, block, lds, st, a)
}
bool launch_k_round(int shape, int lp, int mode, unsigned grid, size_t lds, hipStream_t st, const RoundLaunch &a) {
  shape = compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_round, lp, mode, grid, lds, st, a)
}
bool launch_k_classify(int shape, unsigned grid, size_t lds, hipStream_t st, const ClassifyLaunch &a) {
  shape = compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_classify, grid, lds, st, a)
}
bool launch_k_tail(int shape, int mode, unsigned grid, size_t lds, hipStream_t st, const TailLaunch &a) {
  shape = compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_tail, mode, grid, lds, st, a)
}
bool launch_k_rbound(int shape, unsigned grid, hipStream_t st, ShapeParams sp, double rmax, int nrad, int nang, double *out) {
  shape = compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_rbound, grid, st, sp, rmax, nrad, nang, out)
}
bool launch_k_subsw(int shape, dim3 grid, hipStream_t st, ShapeParams sp, const double *father, const double *child,
                    const unsigned long long *offs, const double *pts, const double *kt, int nkt, int *flag) {
  shape = compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_subsw, grid, st, sp, father, child, offs, pts, kt, nkt, flag)
}
bool launch_k_shape_kernels(int shape, unsigned grid, hipStream_t st, ShapeParams sp, int ks, int count, double resu,
                            int size_side, double safemargin, const double *yaw, unsigned char *map) {
  shape = compiled_shape(shape);
  SVSDF_SLICE_DISPATCH(launch_k_shape_kernels, grid, st, sp, ks, count, resu, size_side, safemargin, yaw, map)
}
}  // namespace svsdf
namespace {

// Persistent-grid launcher of the argmin kernel.  max_queries bounds the (possibly device-side)
// query count and sizes the grid; surplus blocks exit before touching LDS.
size_t table_lds_doubles(const svsdf_ctx *ctx) {
  return 4 * (size_t)ctx->K + 4 * (size_t)((ctx->K + kChunk - 1) / kChunk);
}

void launch_solve(svsdf_ctx *ctx, int G, hipStream_t st, const QuerySet &qs, long long max_queries, double *out_sdf,
                  double *out_t, BatchCtl *ctl, int work_idx,
                  double cull_thresh = std::numeric_limits<double>::infinity()) {
  const double *d_tk = ctx->d_in + 19 * (size_t)ctx->N;
  const long long lanes = std::max<long long>(max_queries * G, 64);
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
    lds = (table_lds_doubles(ctx) + (size_t)traj_lds_doubles(ctx->N) + (poly_lds ? 5 * (size_t)ctx->sp.nverts : 0)) * sizeof(double);
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
  const SolveLaunch a{ctx->d_traj, d_tk, ctx->d_pose, ctx->d_chunks, ctx->sp, qs, out_sdf, out_t, ctx->prune, ctl, work_idx, cull_thresh};
  if (!launch_k_solve(poly_lds ? (int)kPolygonLds : ctx->cfg.shape_id, G, grid, (unsigned)blk, lds_total, st, a) && ctx->launch_err.empty())
    ctx->launch_err = "k_solve: shape not compiled into this build";
  if (ctx->profile) {
    e1 = next_event(ctx);
    (void)hipEventRecord(ctx->ev_pool[e1], st);
    ctx->refine_events.emplace_back(e0, e1);
  }
  ctx->stats.solve_launches++;
}

void launch_round(svsdf_ctx *ctx, hipStream_t st, int b, int it) {
  const int mode = ctx->ub_full ? (ctx->ub_lazy ? 2 : 1) : 0;   // k_round MODE: cheap / full / lazy bound
  const bool scans = mode != 0;
  const long long pts = std::max(1, ctx->bcount[b]);
  bool poly_lds = ctx->poly_lds;
  size_t lds = (table_lds_doubles(ctx) + (poly_lds ? 5 * (size_t)ctx->sp.nverts : 0)) * sizeof(double);
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
  // the seed bound of the scanning modes is tight: a narrow band selects (measured optimum 0.01 m, solve-all from
  // iteration 7); the chunk bound of the cheap mode needs 0.1 m / iteration 5.  Env values override both.
  const double sel = (scans && !ctx->select_env) ? 0.01 : ctx->select_delta;
  const int all_it = (scans && !ctx->all_iter_env) ? 7 : ctx->delta_all_iter;
  const double delta = (it >= all_it) ? 1e300 : sel;
  const double band_delta = (it >= all_it) ? 1e300 : ctx->select_delta;   // lazy mode: cheap-bound band that gets scanned
  // iterations 0 and 1 are (almost always) GSIP rounds 1 and 2 with 2 and 6 samples: 8 lanes per point; later rounds have
  // 18-21 samples: 32 lanes per point (either handles any count).  Iteration 2 (18 samples, still every interior point
  // active: throughput, not latency) also runs faster with 8 lanes and three sample passes per point -- measured round 3,
  // SVSDF_ROUND_LP8_ITERS 2 / 3 / 4 / 6 / all: C3 6.15 / 5.97 / 6.12 / 6.33 / 6.45 ms, NS 7.49 / 7.24 / 7.21 / 7.31 / 7.58
  const int lp = (it < ctx->round_lp8_iters) ? 8 : 32;
  const unsigned grid = (unsigned)std::min<long long>((pts * lp + kRoundBlock - 1) / kRoundBlock, (long long)(256 * 16 * 64) / kRoundBlock);
  const RoundLaunch a{ctx->d_traj, ctx->d_pose, ctx->d_chunks, ctx->sp, ctx->d_px, ctx->d_py, ctx->gs, ctx->P, it, delta,
                      band_delta, ctx->d_res_sdf, ctx->d_res_t, ctx->d_res_gx, ctx->d_res_gy, ctx->d_ctl + b, ctx->round_list};
  size_t e0 = 0, e1 = 0;
  if (ctx->profile) { e0 = next_event(ctx); (void)hipEventRecord(ctx->ev_pool[e0], st); }
  if (!launch_k_round(poly_lds ? (int)kPolygonLds : ctx->cfg.shape_id, lp, mode, grid, lds, st, a) && ctx->launch_err.empty())
    ctx->launch_err = "k_round: shape not compiled into this build";
  if (ctx->profile) {
    e1 = next_event(ctx);
    (void)hipEventRecord(ctx->ev_pool[e1], st);
    ctx->round_events.emplace_back(e0, e1);
  }
}

// k_tail: every GSIP iteration of batch b from `it0` on, in one launch (two points per wave, solved in the wave).
void launch_tail(svsdf_ctx *ctx, hipStream_t st, int b, int it0) {
  const int mode = ctx->ub_full ? (ctx->ub_lazy ? 2 : 1) : 0;
  const bool scans = mode != 0;
  const double sel = (scans && !ctx->select_env) ? 0.01 : ctx->select_delta;
  const int all_it = (scans && !ctx->all_iter_env) ? 7 : ctx->delta_all_iter;
  const int all_after = (ctx->tail_all_after < 0) ? std::max(0, all_it - it0) : ctx->tail_all_after;
  // the active count lives on the device; the grid is sized by what the previous evaluation had there (surplus blocks
  // exit at once, missing ones are made up for by the waves' work fetch)
  long long pts = std::max(1, ctx->bcount[b]);
  if (ctx->have_prev_nactive && it0 <= ctx->prev_tail_iter && it0 < kMaxIter)
    pts = std::min<long long>(pts, ctx->prev_nactive[it0] / std::max(1, ctx->nbatch) * 5 / 4 + 64);
  bool poly_lds = ctx->poly_lds;
  size_t lds = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const size_t lds_tables = ((table_lds_doubles(ctx) + (size_t)traj_lds_doubles(ctx->N) + (poly_lds ? 5 * (size_t)ctx->sp.nverts : 0)) * sizeof(double) + 15) & ~(size_t)15;
    lds = lds_tables + (kTailBlock / 64) * kTailWaveLds;
    if (lds <= ctx->lds_limit || !poly_lds) break;
    poly_lds = false;
  }
  if (lds > ctx->lds_limit) {
    if (ctx->launch_err.empty()) ctx->launch_err = "trajectory too long for the LDS pose table (k_tail)";
    return;
  }
  const long long per_block = (kTailBlock / 64) * 2;
  const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((pts + per_block - 1) / per_block, (long long)ctx->n_cu * 3));
  const double *d_tk = ctx->d_in + 19 * (size_t)ctx->N;
  const TailLaunch a{ctx->d_traj, d_tk, ctx->d_pose, ctx->d_chunks, ctx->sp, ctx->d_px, ctx->d_py, ctx->gs, ctx->P, it0, mode,
                     sel, ctx->select_delta, all_after, ctx->d_res_sdf, ctx->d_res_t, ctx->d_res_gx, ctx->d_res_gy,
                     ctx->d_ctl + b, ctx->round_list, ctx->prune};
  size_t e0 = 0, e1 = 0;
  if (ctx->profile) { e0 = next_event(ctx); (void)hipEventRecord(ctx->ev_pool[e0], st); }
  if (!launch_k_tail(poly_lds ? (int)kPolygonLds : ctx->cfg.shape_id, mode, grid, lds, st, a) && ctx->launch_err.empty())
    ctx->launch_err = "k_tail: shape not compiled into this build";
  if (ctx->profile) {
    e1 = next_event(ctx);
    (void)hipEventRecord(ctx->ev_pool[e1], st);
    ctx->tail_events.emplace_back(e0, e1);
  }
  ctx->stats.tail_launches++;
}

// Iteration the fused tail starts at in this evaluation.  Measured (round 4, profiles/r04_tail_*): the launch chain packs the
// solves of all points 32 to a wave and runs k_round at 4 waves per SIMD -- wherever a launch still holds more points than
// the chip keeps in flight at two per wave it has several times the tail's throughput, and its last launches take 5 - 50 us
// each, so at 100 k - 1 M points the tail only costs time (C2 + 6 %, C3 + 2 %, NS + 3 % with the threshold at 4096 points).
// A small cloud is a pure latency chain of ~ 20 launches: there the whole GSIP loop runs in the tail (it0 = 0; C1, 10 k
// points: 0.90 -> 0.76 ms).  Rule: every GSIP iteration in k_tail when the previous evaluation of this point set had at
// most tail_below interior points, the launch chain otherwise; SVSDF_TAIL pins an iteration or turns the tail off.  Any
// choice gives the same bits.
int choose_tail_iter(const svsdf_ctx *ctx) {
  if (ctx->tail_mode == -2) return -1;
  if (ctx->tail_mode >= 0) return std::min(ctx->tail_mode, (int)kMaxIter - 2);
  if (!ctx->have_prev_nactive) return -1;   // first evaluation of a point set: the interior count is not known yet
  return (ctx->prev_nactive[0] <= ctx->tail_below) ? 0 : -1;
}

void launch_classify(svsdf_ctx *ctx, hipStream_t st, int b) {
  const unsigned grid = (unsigned)std::min<long long>(((long long)ctx->bcount[b] + kBlock - 1) / kBlock, 2048);
  const size_t lds = (size_t)traj_lds_doubles(ctx->N) * sizeof(double);
  const ClassifyLaunch a{ctx->d_traj, ctx->sp, ctx->d_px, ctx->d_py, ctx->d_sdf, ctx->d_t, ctx->d_res_sdf, ctx->d_res_t,
                         ctx->d_res_gx, ctx->d_res_gy, ctx->gs, ctx->d_ctl + b};
  (void)launch_k_classify(ctx->cfg.shape_id, grid, lds, st, a);
}

// Upload (coeffs, T), update traj_duration like SweptVolumeManager::updateTraj (SWM:376-385),
// build the layer-1 time grid exactly like the reference's accumulating loop (SWM:567).
int upload_traj(svsdf_ctx *ctx, int N, const double *coeffs, const double *T) {
  if (N < 1 || N > kMaxPieces) return fail(ctx, SVSDF_ERR_INVALID, "N out of range [1, 64]");
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
  const size_t need = 19 * (size_t)N + K + (K + kChunk - 1) / kChunk;  // coeffs | T | tk | chunk slack
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
    rc = dev_alloc(ctx, &ctx->d_chunks, (K + 1024) / kChunk + 2);
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
    const double *tk = ctx->h_in + 19 * (size_t)N;
    ctx->cull_ok = (td == dur) && K >= 1;
    std::vector<double> S(N + 1, 0.0);
    for (int i = 0; i < N; ++i) S[i + 1] = S[i] + T[i];
    for (size_t c = 0; c < nch; ++c) {
      slack[c] = std::numeric_limits<double>::infinity();
      if (!ctx->cull_ok) continue;
      const size_t k0 = c * kChunk, k1 = std::min(k0 + kChunk, K) - 1;
      const double h = (c + 1 == nch) ? std::max(0.0751, dur - tk[k1]) : 0.0751;
      const double ta = std::max(0.0, tk[k0] - h), tb = std::min(td, tk[k1] + h);
      double v2 = 0.0;
      for (int i = 0; i < N; ++i) {
        const double a = std::max(ta, S[i]) - S[i], b = std::min(tb, S[i + 1]) - S[i];  // local times in piece i
        if (!(b >= a)) continue;
        const double w = b - a;
        double bound[2] = {0.0, 0.0};
        for (int d = 0; d < 2; ++d) {
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
      }
      const double sl = v2 * (1.0 + 1e-9) * h + 1e-9;
      if (std::isfinite(sl)) slack[c] = sl;
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
  hipLaunchKernelGGL(k_prep, dim3(1), dim3(kBlock), lds, ctx->stream, ctx->d_in, N, dur, (int)K,
                     ctx->piece_time_mode, ctx->d_traj,
                     ctx->d_pose, ctx->d_chunks, ctx->r_bound, ctx->d_ctl, ctx->nbatch);
  return SVSDF_OK;
}


int join_batches(svsdf_ctx *ctx) {
  if (!ctx->launch_err.empty()) {   // a launch that could not be made: a clear error instead of a raw HIP launch failure
    const std::string m = ctx->launch_err;
    ctx->launch_err.clear();
    for (int b = 0; b < ctx->nbatch; ++b) (void)hipStreamSynchronize(ctx->bstream[b]);
    return fail(ctx, SVSDF_ERR_INVALID, m);
  }
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
    hipStream_t st = ctx->bstream[b];
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
  int rc = upload_traj(ctx, N, coeffs, T);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(ctx->d_nonfinite, 0, sizeof(int), ctx->stream));
  HIPCHK(hipEventRecord(ctx->ev_prep, ctx->stream));
  const int m = ctx->tail_iter = choose_tail_iter(ctx);
  for (int b = 0; b < ctx->nbatch; ++b) {
    hipStream_t st = ctx->bstream[b];
    BatchCtl *ctl = ctx->d_ctl + b;
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_prep, 0));
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
  HIPCHK(hipEventRecord(ctx->ev_prep, ctx->stream));
  for (int b = 0; b < ctx->nbatch; ++b) {
    hipStream_t st = ctx->bstream[b];
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_prep, 0));
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
  if (with_partial) {
    const unsigned grid = (unsigned)std::min<size_t>((ctx->P + kBlock - 1) / kBlock, 512);
    const size_t plen = 19 * (size_t)N + 1;
    if ((size_t)grid * plen > ctx->block_partials_cap) {
      int rc = dev_alloc(ctx, &ctx->d_block_partials, (size_t)grid * plen);
      if (rc) return rc;
      ctx->block_partials_cap = (size_t)grid * plen;
    }
    const size_t lds = ((size_t)traj_lds_doubles(N) + (kBlock / 64) * plen) * sizeof(double);  // one accumulator row per wave
    hipLaunchKernelGGL(k_assemble, dim3(grid), dim3(kBlock), lds, ctx->stream, ctx->d_traj, ctx->d_px, ctx->d_py,
                       (int)ctx->P, ctx->d_res_sdf, ctx->d_res_t, ctx->d_res_gx, ctx->d_res_gy,
                       ctx->cfg.safety_hor, ctx->cfg.weight_p, ctx->d_block_partials, ctx->d_nonfinite);
    hipLaunchKernelGGL(k_final, dim3((unsigned)plen), dim3(64), 0, ctx->stream, ctx->d_block_partials, (int)grid,
                       ctx->d_sums);
  } else {
    HIPCHK(hipMemsetAsync(ctx->d_sums, 0, kOutPartial * sizeof(double), ctx->stream));
  }
  hipLaunchKernelGGL(k_finish, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->d_sums, N, ctx->d_out, ctx->d_ctl,
                     ctx->nbatch, ctx->it_done, ctx->d_nonfinite,
                     reinterpret_cast<unsigned long long *>(ctx->d_out + kOutPartial));
  ctx->e_end = next_event(ctx);
  (void)hipEventRecord(ctx->ev_pool[ctx->e_end], ctx->stream);
  HIPCHK(hipMemcpyAsync(ctx->h_out, ctx->d_out, kOutDoubles * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  return SVSDF_OK;
}

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

int set_batches(svsdf_ctx *ctx, int nb);   // below

// One evaluation up to the per-point results (and the partial): enqueue, finish.
int evaluate_points(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, bool allow_cull, bool with_partial) {
  int rc = enqueue_queries(ctx, N, coeffs, T, allow_cull);
  if (rc) return rc;
  return finish(ctx, with_partial);
}

void fill_mode_stats(svsdf_ctx *ctx) {
  ctx->stats.gsip_bound_mode = ctx->ub_full ? (ctx->ub_lazy ? 2 : 1) : 0;
  ctx->stats.piece_time_exact = ctx->stats_piece_time;
  ctx->stats.bound_mode_decided = (ctx->ub_env || ctx->ub_tune > 0) ? 1 : 0;
  ctx->stats.plan_settled = (ctx->stats.bound_mode_decided && ctx->bt_state == 0 && ctx->have_prev_nsolve) ? 1 : 0;
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
    if (N < 1 || N > kMaxPieces) return fail(ctx, SVSDF_ERR_INVALID, "N out of range [1, 64]");
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
  if (deciding) { ctx->ub_full = false; ctx->ub_lazy = false; }
  // Batch count (large shards in the scanning modes; DESIGN.md "concurrent point batches").  Default: a RULE -- 3 batches in
  // a scanning bound mode from 400 k points per device, 1 otherwise (what the measurements of rounds 3 - 4 -- 1 / 2 / 3 / 4 at NS, C3, C4 --
  // chose on every box; with the main stream that is at most 4 streams, the HIP runtime's default number of hardware
  // queues) -- so that the plan is the same on every run and settled after the deciding evaluation.  svsdf_set_plan
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
    if (ctx->ub_full) { ctx->have_prev_nsolve = false; ctx->have_prev_nactive = false; }   // the launch plan on record is the cheap-bound one
  }
  // the batch count follows once the bound mode is known (also when it was pinned)
  if (rc == SVSDF_OK && ctx->ub_tune == 0) {
    ctx->ub_tune = 1;
    const bool big = ctx->ub_full && ctx->P >= 400000;
    if (ctx->want_batches == 0) {
      const int nb = big ? 3 : 1;
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

int run_pipeline(svsdf_ctx *ctx, int N, const double *coeffs, const double *T);  // leaf or group (below)

void accumulate(int N, const double *partial, double *cost, double *gradT, double *gradC) {
  *cost += partial[0];
  for (int e = 0; e < 18 * N; ++e) gradC[e] += partial[1 + e];
  for (int j = 0; j < N; ++j) gradT[j] += partial[1 + 18 * N + j];
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
  if ((rc = dev_alloc(ctx, &ctx->gs.pt, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.r, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.theta0, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.theta_res, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.iter, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.nsamp, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.phase, P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.list[0], P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.list[1], P))) return rc;

  const size_t S = P * kMaxSlots;
  if ((rc = dev_alloc(ctx, &ctx->gs.solve, S))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.sqx, S))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.sqy, S))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.sqth, S))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.sq_ub, S))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.sq_k, S))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.sq_sdf, S))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->gs.sq_t, S))) return rc;
  return SVSDF_OK;
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
struct CloudPlan {
  int device = 0;
  const double *d_xyz = nullptr;
  size_t P = 0;
  double *d_part = nullptr;
  unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
  const unsigned long long *sorted = nullptr;
  void *d_tmp = nullptr;
  void release() {
    (void)hipSetDevice(device);
    for (void *q : {(void *)d_part, (void *)d_keys, (void *)d_keys2, d_tmp})
      if (q) (void)hipFree(q);
    d_part = nullptr; d_keys = d_keys2 = nullptr; d_tmp = nullptr; sorted = nullptr;
  }
};

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
  if (!ctx->ub_env) { ctx->ub_full = false; ctx->ub_lazy = false; }
  if (!ctx->G_env) {
    // (Polygon: 4 -- its 2-lane kernel spills 52 registers under the 3-waves cap; C5 36.5 vs 34.8 ms)
    ctx->G = (Ps < 3000) ? 32 : (Ps < 20000) ? 16 : (Ps < 300000) ? 8 : (ctx->cfg.shape_id == (int)kPolygon) ? 4 : 2;
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

// multi-device group: the cloud (on device subs[0]) is planned ONCE there, every sub-context takes its stripe
int upload_group_device(svsdf_ctx *ctx, const double *d_xyz, size_t P) {
  const int G = (int)ctx->subs.size();
  svsdf_ctx *s0 = ctx->subs[0];
  CloudPlan plan;
  int rc = plan_cloud(s0, d_xyz, P, plan);
  for (int k = 0; k < G && !rc; ++k) {
    rc = take_stripe(ctx->subs[k], s0, plan, ctx->cfg.rank * G + k, ctx->cfg.world_size * G);
    if (rc) {
      ctx->err = "device " + std::to_string(ctx->subs[k]->device) + " (stripe " + std::to_string(k) + "): " + ctx->subs[k]->err;
      g_last_error = ctx->err;
    }
  }
  if (rc && ctx->err.empty()) ctx->err = s0->err;
  plan.release();
  ctx->P = 0;
  ctx->shard_idx.clear();
  for (svsdf_ctx *s : ctx->subs) {
    if (rc) { s->P = 0; s->points_set = false; }
    ctx->P += s->P;
    ctx->shard_idx.insert(ctx->shard_idx.end(), s->shard_idx.begin(), s->shard_idx.end());
  }
  ctx->points_set = rc == SVSDF_OK;
  return rc;
}

// ---- in-process multi-GPU group ---------------------------------------------------------------------
// One host thread per device: kernel launches of the devices are issued concurrently (an evaluation is ~25
// launches per device) and every thread keeps its device current.
// run f(k) for every sub-context on its worker thread; first non-zero return code wins
int group_run(svsdf_ctx *ctx, const std::function<int(int)> &f) {
  const int G = (int)ctx->subs.size();
  for (int k = 0; k < G; ++k) ctx->workers[k]->post([&f, k] { return f(k); });
  int rc = SVSDF_OK;
  for (int k = 0; k < G; ++k) {
    const int r = ctx->workers[k]->wait();
    if (r && !rc) {
      rc = r;
      ctx->err = "device " + std::to_string(ctx->subs[k]->device) + " (stripe " + std::to_string(k) + "): " + ctx->subs[k]->err;
      g_last_error = ctx->err;
    }
  }
  return rc;
}

// RCCL entry points, resolved lazily (dlopen) so that single-GPU users never load the library
struct RcclApi {
  void *h = nullptr;
  int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
  int (*CommDestroy)(void *comm) = nullptr;
  int (*AllReduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t st) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*CommCount)(void *comm, int *count) = nullptr;
  bool load() {
    if (h) return true;
    // RCCL must come from the same ROCm tree as the HIP/HSA runtime this library is bound to: its init dlopen()s
    // "libhsa-runtime64.so" by file name, and a copy from another tree (PyTorch-ROCm ships its own under torch/lib)
    // is a second, uninitialised HSA instance ("no ROCm-capable device").  So: the librccl next to our libamdhip64
    // first, then whatever the process already holds, then the default search path.
    {
      Dl_info di;
      if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &di) && di.dli_fname) {
        std::string dir(di.dli_fname);
        const size_t k = dir.rfind('/');
        if (k != std::string::npos) {
          dir.resize(k);
          for (const char *name : {"/librccl.so.1", "/librccl.so"}) {
            h = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
            if (h) break;
          }
        }
      }
    }
    if (!h)
      for (const char *name : {"librccl.so.1", "librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
      }
    if (!h)
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
      }
    if (!h) return false;
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(h, "ncclAllReduce"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(h, "ncclCommCount"));
    return CommInitAll && CommDestroy && AllReduce;
  }
};
RcclApi g_rccl;
constexpr int kNcclFloat64 = 8, kNcclSum = 0;   // ncclDataType_t / ncclRedOp_t values of rccl.h (ncclDouble, ncclSum)

void merge_stats(svsdf_ctx *ctx) {
  svsdf_stats t{};
  for (svsdf_ctx *s : ctx->subs) {
    const svsdf_stats &a = s->stats;
    t.points += a.points; t.interior_points += a.interior_points; t.solves += a.solves;
    t.gsip_samples += a.gsip_samples; t.sdf_evals += a.sdf_evals; t.scan_evals += a.scan_evals;
    t.round_scan_evals += a.round_scan_evals; t.speculative_evals += a.speculative_evals;
    t.round_ms = std::max(t.round_ms, a.round_ms); t.round_ms_sum = std::max(t.round_ms_sum, a.round_ms_sum);
    t.batches = std::max(t.batches, a.batches);
    t.culled_points += a.culled_points;
    t.device_ms = std::max(t.device_ms, a.device_ms); t.solve_ms = std::max(t.solve_ms, a.solve_ms);
    t.solve_ms_sum = std::max(t.solve_ms_sum, a.solve_ms_sum);
    t.solve_launches = std::max(t.solve_launches, a.solve_launches);
    t.tail_launches = std::max(t.tail_launches, a.tail_launches);
    t.tail_iter = std::max(t.tail_iter, a.tail_iter);
    t.tail_points += a.tail_points;
    t.tail_ms = std::max(t.tail_ms, a.tail_ms); t.tail_ms_sum = std::max(t.tail_ms_sum, a.tail_ms_sum);
    t.gsip_iterations = std::max(t.gsip_iterations, a.gsip_iterations);
    t.gsip_bound_mode = std::max(t.gsip_bound_mode, a.gsip_bound_mode);
    t.piece_time_exact = std::max(t.piece_time_exact, a.piece_time_exact);
    t.bound_ratio = std::max(t.bound_ratio, a.bound_ratio);
  }
  t.bound_mode_decided = 1;
  t.plan_settled = 1;
  for (svsdf_ctx *s : ctx->subs) { t.bound_mode_decided &= s->stats.bound_mode_decided; t.plan_settled &= s->stats.plan_settled; }
  t.n_devices = (int)ctx->subs.size();
  t.combine = ctx->combine;
  t.combine_ms = ctx->combine_ms;
  t.setup_ms = ctx->setup_ms;
  ctx->stats = t;
}

int run_pipeline_group(svsdf_ctx *ctx, int N, const double *coeffs, const double *T) {
  const int G = (int)ctx->subs.size();
  const size_t plen = 19 * (size_t)N + 1;
  const bool rccl = ctx->combine == SVSDF_COMBINE_RCCL;
  int rc = group_run(ctx, [&](int k) -> int {
    svsdf_ctx *s = ctx->subs[k];
    int r = run_pipeline_leaf(s, N, coeffs, T);
    if (!rccl) return r;
    // EVERY device thread joins the collective, also after a local failure (non-finite result, exhausted GSIP
    // iterations, invalid trajectory ...): a thread that returned early would leave the others blocked in the all-reduce
    // for ever.  A failed stripe contributes a NaN-poisoned partial (all bits set), so no rank can mistake the sum for a
    // result; the error is reported after the synchronisation.
    (void)hipSetDevice(s->device);
    if (r) (void)hipMemsetAsync(s->d_out, 0xFF, plen * sizeof(double), s->stream);
    const int e = g_rccl.AllReduce(s->d_out, ctx->d_red[k], plen, kNcclFloat64, kNcclSum, ctx->comms[k], s->stream);
    if (e && !r) r = fail(s, SVSDF_ERR_RCCL, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"));
    if (k == 0 && hipMemcpyAsync(ctx->h_red, ctx->d_red[0], plen * sizeof(double), hipMemcpyDeviceToHost, s->stream) != hipSuccess && !r)
      r = fail(s, SVSDF_ERR_RCCL, "read-back of the reduced partial failed");
    if (hipStreamSynchronize(s->stream) != hipSuccess && !r) r = fail(s, SVSDF_ERR_RCCL, "stream sync after ncclAllReduce failed");
    return r;
  });
  if (rc) return rc;
  const auto t0 = std::chrono::steady_clock::now();
  ctx->comb.resize(kOutPartial);
  if (rccl) {
    std::copy(ctx->h_red, ctx->h_red + plen, ctx->comb.begin());
  } else {
    // fixed-order host sum of G pinned partials (G x 5 KB): deterministic, no extra launch or sync
    for (size_t e = 0; e < plen; ++e) {
      double a = ctx->subs[0]->h_out[e];
      for (int k = 1; k < G; ++k) a += ctx->subs[k]->h_out[e];
      ctx->comb[e] = a;
    }
  }
  ctx->combine_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx->N = N;
  ctx->h_partial = ctx->comb.data();
  merge_stats(ctx);
  return SVSDF_OK;
}

int run_pipeline(svsdf_ctx *ctx, int N, const double *coeffs, const double *T) {
  return ctx->subs.empty() ? run_pipeline_leaf(ctx, N, coeffs, T) : run_pipeline_group(ctx, N, coeffs, T);
}

// Stage the host cloud on the device (one H2D of the AoS array) and plan + gather there; a multi-device group stages
// and plans it once, on its first device.
int set_points_host(svsdf_ctx *ctx, const double *xyz, size_t P) {
  if (ctx->host_only) return fail(ctx, SVSDF_ERR_NO_DEVICE, "host-only context: no device entry points");
  if (P > 0xffffffffull) return fail(ctx, SVSDF_ERR_INVALID, "too many points");
  const auto t0 = std::chrono::steady_clock::now();
  svsdf_ctx *s0 = ctx->subs.empty() ? ctx : ctx->subs[0];
  HIPCHK(hipSetDevice(s0->device));
  double *d_xyz = nullptr;
  if (P) {
    HIPCHK(hipMalloc((void **)&d_xyz, 3 * P * sizeof(double)));
    const hipError_t e = hipMemcpy(d_xyz, xyz, 3 * P * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d_xyz); return fail(ctx, SVSDF_ERR_HIP_BASE + (int)e, "upload of the query points failed"); }
  }
  const int rc = ctx->subs.empty() ? upload_shard_device(ctx, d_xyz, P, ctx->cfg.rank, ctx->cfg.world_size)
                                   : upload_group_device(ctx, d_xyz, P);
  if (d_xyz) { (void)hipSetDevice(s0->device); (void)hipFree(d_xyz); }
  ctx->setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (svsdf_ctx *s : ctx->subs) s->setup_ms = ctx->setup_ms;
  return rc;
}

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

// RCCL side of an in-process group: one communicator rank per sub-context (ncclCommInitAll over the group's devices),
// an all-reduce output buffer per device and one pinned read-back buffer.  Returns an error text, empty on success.
std::string group_init_rccl(svsdf_ctx *g) {
  if (!g->comms.empty()) return "";
  const int G = (int)g->subs.size();
  std::vector<int> devs(G);
  bool distinct = true;
  for (int k = 0; k < G; ++k) {
    devs[k] = g->subs[k]->device;
    for (int j = 0; j < k; ++j) distinct = distinct && devs[j] != devs[k];
  }
  if (!distinct) return "SVSDF_COMBINE_RCCL needs distinct devices (one communicator rank per GPU)";
  if (!g_rccl.load()) return "librccl.so could not be loaded (SVSDF_COMBINE_RCCL)";
  g->comms.assign(G, nullptr);
  {
    const hipError_t stale = hipGetLastError();   // RCCL's init treats any pending (sticky-until-read) HIP error as its own
    if (stale != hipSuccess && std::getenv("SVSDF_DEBUG")) std::fprintf(stderr, "[svsdf] cleared pending HIP error before ncclCommInitAll: %s\n", hipGetErrorString(stale));
  }
  const int e = g_rccl.CommInitAll(g->comms.data(), G, devs.data());
  if (e) { g->comms.clear(); return std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"); }
  g->d_red.assign(G, nullptr);
  for (int k = 0; k < G; ++k)
    if (hipSetDevice(devs[k]) != hipSuccess || hipMalloc((void **)&g->d_red[k], kOutPartial * sizeof(double)) != hipSuccess)
      return "allocation of the all-reduce buffer failed";
  if (!g->h_red && hipHostMalloc((void **)&g->h_red, kOutPartial * sizeof(double), hipHostMallocDefault) != hipSuccess)
    return "pinned allocation failed";
  return "";
}

// In-process multi-GPU context: one single-device sub-context (and one host thread) per entry of cfg->devices.
svsdf_ctx *create_group(const svsdf_config *cfg, int ndev) {
  const int G = cfg->n_devices;
  for (int k = 0; k < G; ++k) {
    if (cfg->devices[k] < 0 || cfg->devices[k] >= ndev) {
      g_last_error = "svsdf_create: devices[" + std::to_string(k) + "] is not a visible HIP device";
      return nullptr;
    }
  }
  svsdf_ctx *g = new svsdf_ctx();
  g->cfg = *cfg;
  g->cfg.polygon_xy = nullptr;
  g->device = cfg->devices[0];
  g->combine = cfg->combine == SVSDF_COMBINE_RCCL ? SVSDF_COMBINE_RCCL : SVSDF_COMBINE_HOST;
  if (const char *e = std::getenv("SVSDF_COMBINE")) g->combine = (std::string(e) == "rccl") ? SVSDF_COMBINE_RCCL : SVSDF_COMBINE_HOST;
  auto bail = [&](const std::string &m) -> svsdf_ctx * {
    g_last_error = m;
    svsdf_destroy(g);
    return nullptr;
  };
  for (int k = 0; k < G; ++k) {
    svsdf_config c = *cfg;
    c.n_devices = 0;
    c.device = cfg->devices[k];
    c.rank = cfg->rank * G + k;
    c.world_size = cfg->world_size * G;
    svsdf_ctx *s = svsdf_create(&c);
    if (!s) return bail("svsdf_create: device " + std::to_string(c.device) + ": " + g_last_error);
    g->subs.push_back(s);
    g->workers.emplace_back(new Worker());
  }
  g->r_bound = g->subs[0]->r_bound;
  if (g->combine == SVSDF_COMBINE_RCCL) {
    const std::string e = group_init_rccl(g);
    if (!e.empty()) return bail("svsdf_create: " + e);
  }
  return g;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
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
  int dev = cfg->device;
  if (dev < 0) (void)hipGetDevice(&dev);
  ctx->device = dev;
  auto bail = [&](const std::string &m) -> svsdf_ctx * {
    g_last_error = m;
    svsdf_destroy(ctx);
    return nullptr;
  };
  if (hipSetDevice(dev) != hipSuccess) return bail("hipSetDevice failed");
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate failed");
  // shape constants, evaluated with the host libm exactly where the reference does (SHP:281-294,
  // :855, :1237, :1278, :1320)
  ShapeParams &sp = ctx->sp;
  sp.tx = cfg->poly_params[0];
  sp.ty = cfg->poly_params[1];
  const double yaw = cfg->poly_params[2] * kPI / 180.0;
  sp.r00 = std::cos(yaw); sp.r01 = -std::sin(yaw); sp.r10 = std::sin(yaw); sp.r11 = std::cos(yaw);
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
    if (cfg->polygon_xy && cfg->polygon_nverts >= 3) {
      if (cfg->polygon_nverts > SVSDF_MAX_POLY_VERTS)
        return bail("svsdf_create: polygon_nverts exceeds SVSDF_MAX_POLY_VERTS (" + std::to_string(SVSDF_MAX_POLY_VERTS) + ")");
      v.assign(cfg->polygon_xy, cfg->polygon_xy + 2 * (size_t)cfg->polygon_nverts);
    } else {
      v = {6, -0.1, 6, 0.1, -6, 0.1, -6, -0.1};  // SWM:363-369
    }
    // candidate lists of the outline (svsdf_polygon.hpp), then one upload: the header's pointers are device addresses
    PolyAccelHost pa;
    int ngf = 128, ngc = 256;   // grid cells per side, fine / coarse (env SVSDF_POLY_GRID="f,c": experiments)
    if (const char *e = std::getenv("SVSDF_POLY_GRID")) {
      int a = 0, b = 0;
      if (std::sscanf(e, "%d,%d", &a, &b) == 2 && a >= 8 && a <= 1024 && b >= 8 && b <= 1024) { ngf = a; ngc = b; }
    }
    if (!build_poly_accel(v.data(), (int)(v.size() / 2), pa, ngf, ngc)) return bail("svsdf_create: polygon outline rejected (non-finite vertex?)");
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
    sp.nverts = (int)(v.size() / 2);
    sp.accel = reinterpret_cast<const PolyAccel *>(ctx->d_poly);
    sp.edges = pa.hdr.edges;
    // the solve / round kernels keep outlines of up to 1024 edges (40 KB) in LDS in front of the pose table
    ctx->poly_lds = sp.nverts <= kPolyLdsMaxVerts;
    if (const char *e = std::getenv("SVSDF_POLY_LDS")) ctx->poly_lds = ctx->poly_lds && std::atoi(e) != 0;
    ctx->cfg.polygon_nverts = sp.nverts;
  }
  ctx->G_env = 0;
  if (const char *e = std::getenv("SVSDF_G")) { const int g = std::atoi(e); if (g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 32) { ctx->G_env = g; ctx->G = g; ctx->G_late = std::max(g, 8); } }
  if (const char *e = std::getenv("SVSDF_G_LATE")) { const int g = std::atoi(e); if (g == 1 || g == 2 || g == 4 || g == 8 || g == 16 || g == 32) { ctx->G_late = g; ctx->G_late_env = g; } }
  if (const char *e = std::getenv("SVSDF_PRUNE")) ctx->prune = std::atoi(e) != 0;
  if (const char *e = std::getenv("SVSDF_BLOCK")) { const int b = std::atoi(e); ctx->block = (b == 128 || b == 256) ? b : 64; ctx->block_env = true; }
  if (const char *e = std::getenv("SVSDF_BATCHES")) ctx->want_batches = (std::string(e) == "measure") ? -1 : std::max(0, std::min(std::atoi(e), (int)kMaxBatches));
  if (const char *e = std::getenv("SVSDF_WAVES_PER_CU")) ctx->waves_per_cu = std::max(1, std::min(std::atoi(e), 32));
  if (const char *e = std::getenv("SVSDF_DELTA_ALL_ITER")) { ctx->delta_all_iter = std::atoi(e); ctx->all_iter_env = true; }
  if (const char *e = std::getenv("SVSDF_ROUND_LP8_ITERS")) ctx->round_lp8_iters = std::atoi(e);
  if (const char *e = std::getenv("SVSDF_SELECT_DELTA")) { ctx->select_delta = std::atof(e); ctx->select_env = true; }
  {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && n > 0) ctx->n_cu = n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device) == hipSuccess && n > 0) ctx->lds_limit = (size_t)n;
  }
  if (const char *e = std::getenv("SVSDF_FIRST_ITERS")) { ctx->first_iters = std::max(1, std::min(std::atoi(e), (int)kMaxIter)); ctx->adaptive_iters = false; }
  if (const char *e = std::getenv("SVSDF_LATE_ITER")) ctx->late_iter = std::atoi(e);
  if (const char *e = std::getenv("SVSDF_CULL")) ctx->cull = std::atoi(e) != 0;
  if (const char *e = std::getenv("SVSDF_TAIL")) ctx->tail_mode = (std::string(e) == "off") ? -2 : (std::string(e) == "auto") ? -1 : std::max(0, std::atoi(e));
  if (const char *e = std::getenv("SVSDF_TAIL_BELOW")) ctx->tail_below = std::atoll(e);
  if (const char *e = std::getenv("SVSDF_TAIL_ALL_AFTER")) ctx->tail_all_after = std::atoi(e);
  if (const char *e = std::getenv("SVSDF_ROUND_LIST")) ctx->round_list = std::atoi(e);
  if (const char *e = std::getenv("SVSDF_UB_FULL")) { ctx->ub_full = std::atoi(e) != 0; ctx->ub_lazy = std::atoi(e) == 2; ctx->ub_env = true; }
  if (const char *e = std::getenv("SVSDF_UB_RATIO")) { ctx->ub_threshold = std::atof(e); ctx->ub_thr_env = true; }
  if (const char *e = std::getenv("SVSDF_WIDE32")) ctx->wide32_below = std::atoll(e);
  if (const char *e = std::getenv("SVSDF_WIDE16")) ctx->wide16_below = std::atoll(e);
  if (const char *e = std::getenv("SVSDF_WIDE8")) ctx->wide8_below = std::atoll(e);
  if (const char *e = std::getenv("SVSDF_PROFILE")) ctx->profile = std::atoi(e) != 0;
  if (const char *e = std::getenv("SVSDF_PIECE_TIME")) {   // exact | fast | auto (default)
    ctx->cfg.flags &= ~(SVSDF_FLAG_EXACT_PIECE_TIME | SVSDF_FLAG_FAST_PIECE_TIME);
    if (std::string(e) == "exact") ctx->cfg.flags |= SVSDF_FLAG_EXACT_PIECE_TIME;
    else if (std::string(e) == "fast") ctx->cfg.flags |= SVSDF_FLAG_FAST_PIECE_TIME;
  }
  for (int b = 0; b < kMaxBatches; ++b) {
    if (hipStreamCreateWithFlags(&ctx->bstream[b], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_done[b], hipEventDisableTiming) != hipSuccess)
      return bail("stream/event creation failed");
  }
  if (hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming) != hipSuccess) return bail("event creation failed");
  if (hipMalloc((void **)&ctx->d_traj, sizeof(TrajDev)) != hipSuccess ||
      hipMalloc((void **)&ctx->d_ctl, kMaxBatches * sizeof(BatchCtl) + 8 * 8 * (kMaxIter + 4)) != hipSuccess ||
      hipMalloc((void **)&ctx->d_nonfinite, sizeof(int)) != hipSuccess ||
      hipMalloc((void **)&ctx->d_sums, kOutPartial * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&ctx->d_out, kOutDoubles * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void **)&ctx->h_out, kOutDoubles * sizeof(double), hipHostMallocDefault) != hipSuccess)
    return bail("device allocation failed");
  // (on the context's own stream: it is non-blocking, a null-stream memset would not be ordered with its kernels)
  if (hipMemsetAsync(ctx->d_ctl, 0, kMaxBatches * sizeof(BatchCtl) + 8 * 8 * (kMaxIter + 4), ctx->stream) != hipSuccess) return bail("hipMemset failed");
  {  // shape bound radius R with sdf_shape(q) >= |q| - R for every q: the shape's circumradius about the body origin
     // (analytic, per shape; shape_circumradius above) plus the length of its offset (Shape.hpp:281-294).  The polar
     // sample of |q| - sdf(q) (k_rbound, out to 60 m) is kept as a self-check of that bound, not as its source.
    const double r0 = shape_circumradius(cfg->shape_id, ctx->poly_xy.data(), sp.nverts);
    const double analytic = (r0 + std::hypot(sp.tx, sp.ty)) * (1.0 + 1e-12) + 1e-6;
    if (hipMemsetAsync(ctx->d_out, 0, sizeof(double), ctx->stream) != hipSuccess) return bail("hipMemset failed");
    const int nrad = 512, nang = 4096;
    const unsigned grid = (unsigned)((nrad * nang + kBlock - 1) / kBlock);
    (void)launch_k_rbound(cfg->shape_id, grid, ctx->stream, ctx->sp, 60.0, nrad, nang, ctx->d_out);
    double rb = 0.0;
    if (hipMemcpyAsync(&rb, ctx->d_out, sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
      return bail("shape bound kernel failed");
    ctx->r_bound_sampled = rb;
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
    ctx->workers.clear();   // joins the threads
    for (size_t k = 0; k < ctx->comms.size(); ++k)
      if (ctx->comms[k] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comms[k]);
    for (size_t k = 0; k < ctx->d_red.size(); ++k)
      if (ctx->d_red[k]) { (void)hipSetDevice(ctx->subs[k]->device); (void)hipFree(ctx->d_red[k]); }
    if (ctx->h_red) (void)hipHostFree(ctx->h_red);
    for (svsdf_ctx *s : ctx->subs) svsdf_destroy(s);
    delete ctx;
    return;
  }
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  void *bufs[] = {ctx->d_poly, ctx->d_px, ctx->d_py, ctx->d_traj, ctx->d_in, ctx->d_pose, ctx->d_chunks,
                  ctx->d_sdf, ctx->d_t, ctx->d_res_sdf, ctx->d_res_t, ctx->d_res_gx, ctx->d_res_gy, ctx->gs.pt,
                  ctx->gs.r, ctx->gs.theta0, ctx->gs.theta_res, ctx->gs.iter, ctx->gs.nsamp, ctx->gs.phase,
                  ctx->gs.list[0], ctx->gs.list[1], ctx->gs.solve, ctx->gs.sqx, ctx->gs.sqy, ctx->gs.sqth,
                  ctx->gs.sq_ub, ctx->gs.sq_k, ctx->gs.sq_sdf, ctx->gs.sq_t, ctx->d_ctl,
                  ctx->d_block_partials, ctx->d_sums,
                  ctx->d_out, ctx->d_nonfinite, ctx->d_fe, ctx->d_fe_flag};
  for (void *p : bufs)
    if (p) (void)hipFree(p);
  if (ctx->h_in) (void)hipHostFree(ctx->h_in);
  if (ctx->h_out) (void)hipHostFree(ctx->h_out);
  if (ctx->h_fe) (void)hipHostFree(ctx->h_fe);
  for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->ev_prep) (void)hipEventDestroy(ctx->ev_prep);
  for (int b = 0; b < kMaxBatches; ++b) {
    if (ctx->ev_done[b]) (void)hipEventDestroy(ctx->ev_done[b]);
    if (ctx->bstream[b]) (void)hipStreamDestroy(ctx->bstream[b]);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
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

#ifdef SVSDF_SITE_STATS
// diagnostic builds only (tools/site_stats.py): k_solve's per-site execution / lane counters of the last evaluation
extern "C" int svsdf_debug_site_stats(svsdf_ctx *ctx, unsigned long long out[20]) {
  if (!ctx || ctx->host_only || !ctx->subs.empty()) return SVSDF_ERR_INVALID;
  if (hipSetDevice(ctx->device) != hipSuccess) return SVSDF_ERR_HIP_BASE;
  std::vector<BatchCtl> hc(kMaxBatches);
  if (hipMemcpy(hc.data(), ctx->d_ctl, sizeof(BatchCtl) * kMaxBatches, hipMemcpyDeviceToHost) != hipSuccess) return SVSDF_ERR_HIP_BASE;
  for (int i = 0; i < 20; ++i) out[i] = 0;
  for (const BatchCtl &b : hc)
    for (const StatSlot &sl : b.stat)
      for (int i = 0; i < 20; ++i) out[i] += sl.pad[i];
  return SVSDF_OK;
}
#endif

int svsdf_set_profiling(svsdf_ctx *ctx, int enable) {
  if (!ctx) return SVSDF_ERR_INVALID;
  if (!ctx->subs.empty()) {
    int rc = SVSDF_OK;
    for (svsdf_ctx *s : ctx->subs) { const int r = svsdf_set_profiling(s, enable); if (r && !rc) rc = r; }
    ctx->profile = enable != 0;
    return rc;
  }
  ctx->profile = enable != 0;
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

int svsdf_get_plan(const svsdf_ctx *ctx, svsdf_plan *out) {
  if (!ctx || !out) return SVSDF_ERR_INVALID;
  const svsdf_ctx *c = ctx->subs.empty() ? ctx : ctx->subs[0];
  out->bound_mode = c->ub_full ? (c->ub_lazy ? 2 : 1) : 0;
  out->batches = (c->saved_nbatch > 0) ? c->saved_nbatch : c->nbatch;
  out->lanes_per_query = c->G;
  out->tail_iter = (c->tail_mode == -2) ? -2 : (c->tail_mode >= 0) ? c->tail_mode : (c->have_prev_nactive ? choose_tail_iter(c) : SVSDF_PLAN_AUTO);
  out->settled = ((c->ub_env || c->ub_tune > 0) && c->bt_state == 0 && c->have_prev_nsolve) ? 1 : 0;
  return SVSDF_OK;
}

int svsdf_set_plan(svsdf_ctx *ctx, const svsdf_plan *plan) {
  if (!ctx || !plan) return fail(ctx, SVSDF_ERR_INVALID, "svsdf_set_plan: null argument");
  const int g = plan->lanes_per_query;
  if (plan->bound_mode < SVSDF_PLAN_AUTO || plan->bound_mode > 2 || plan->batches < -2 || plan->batches == 0 || plan->batches > kMaxBatches ||
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
    const bool full = plan->bound_mode != 0, lazy = plan->bound_mode == 2;
    if (!ctx->ub_env || full != ctx->ub_full || lazy != ctx->ub_lazy) { ctx->have_prev_nsolve = false; ctx->have_prev_nactive = false; ctx->ub_tune = 0; }
    ctx->ub_env = true; ctx->ub_full = full; ctx->ub_lazy = lazy;
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
    ctx->G = (Ps < 3000) ? 32 : (Ps < 20000) ? 16 : (Ps < 300000) ? 8 : (ctx->cfg.shape_id == (int)kPolygon) ? 4 : 2;
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
    *rccl_ranks = 0;   // no communicator
    if (!ctx->comms.empty() && ctx->comms[0] && g_rccl.CommCount) {
      int n = 0;
      if (g_rccl.CommCount(ctx->comms[0], &n) == 0) *rccl_ranks = n;   // asked of the communicator itself
    }
  }
  return SVSDF_OK;
}

int svsdf_last_stats(const svsdf_ctx *ctx, svsdf_stats *out) {
  if (!ctx || !out) return SVSDF_ERR_INVALID;
  *out = ctx->stats;
  return SVSDF_OK;
}

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
  if (N > kMaxPieces) return fail(ctx, SVSDF_ERR_INVALID, "N > 64");
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
