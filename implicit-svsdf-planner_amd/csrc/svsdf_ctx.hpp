// svsdf_ctx.hpp -- internal state of the C-ABI host layer, shared by its translation units (round 4 split of svsdf_api.hip):
//   svsdf_pipeline.hip  device pipeline of ONE device: launchers, trajectory upload, evaluation, point upload / sort / stripes
//   svsdf_group.hip     in-process multi-GPU group: worker threads, host / RCCL combine
//   svsdf_capi.hip      the C ABI of include/svsdf_c.h for the hot path: contexts, points, evaluation, plan, full callback
//   svsdf_extras.hip    C ABI of the rows around the path: front end, map / mesh helpers, swept outline, L-BFGS driver
// Nothing here is part of the public interface.
#pragma once
#include <hip/hip_runtime.h>

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
#include "svsdf_launch.hpp"
#include "svsdf_minco.hpp"

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
  hipStream_t bstream[svsdf::kMaxBatches] = {};      // one stream per point batch
  int stream_slot = -1;                       // which set of the process-wide stream pool these are (acquire_streams)
  hipEvent_t ev_prep = nullptr, ev_done[svsdf::kMaxBatches] = {};
  svsdf::ShapeParams sp{};
  bool poly_lds = false;             // Polygon: k_solve / k_round run their kPolygonLds variants (edges at the start of LDS)
  unsigned char *d_poly = nullptr;   // Polygon: one device blob [PolyAccel | edges | cell records | slab records | long lists]
  std::vector<double> poly_xy;       // Polygon: the outline as given (host copy)
  std::vector<int> poly_loops;       // Polygon: vertices per closed loop (empty: one loop)
  std::string err;

  // points: this rank's shard, Morton-sorted, split into nbatch contiguous batches
  size_t P = 0;
  bool points_set = false;           // svsdf_set_points was called (P may be 0: an obstacle-free window)
  std::vector<long long> shard_idx;  // original index of shard element j
  double *d_px = nullptr, *d_py = nullptr;
  int nbatch = 1;
  int bstart[svsdf::kMaxBatches] = {}, bcount[svsdf::kMaxBatches] = {};
  svsdf::BatchCtl *d_ctl = nullptr;

  // trajectory
  svsdf::TrajDev *d_traj = nullptr;
  double *d_in = nullptr;  // device staging: coeffs (18N) | T (N) | tk (K)
  double *h_in = nullptr;  // pinned mirror
  size_t in_cap = 0;       // doubles
  svsdf::Pose *d_pose = nullptr;
  svsdf::Chunk *d_chunks = nullptr;
  size_t pose_cap = 0;
  double r_bound = 0.0;    // shape bound radius for the layer-1 chunk pruning (analytic circumradius + offset)
  double r_bound_sampled = 0.0;  // max over a polar grid of |q| - sdf(q): self-check, must not exceed r_bound
  double lipschitz_excess = 0.0; // largest |f(q') - f(q)| - |q' - q| found by the same grid (k_rbound); must be <= 0
  bool lipschitz_ok = true;      // false: the shape SDF is not 1-Lipschitz -> no value-based cull, no anchor bound mode
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
  int round_blocks_per_cu = 4;   // k_round: blocks per CU the grid is capped at (its blocks fetch further iterations themselves); env SVSDF_ROUND_BPC
  double wall_clock_khz = 100000.0;   // hipDeviceAttributeWallClockRate: rate of the constant counter of the clock probe
  bool adaptive_iters = true;
  bool ub_full = false;        // k_round scans every new GSIP sample (seed = tightest layer-1 bound, reused by k_solve)
  bool ub_lazy = false;        // with ub_full: only the samples in the cheap-bound band are scanned (k_round MODE 2)
  bool ub_anchor = false;      // with ub_full && !ub_lazy: anchor scans (k_round MODE 3: every third sample, the rest by their Lipschitz bound)
  int an_state = 0;            // anchor trial: 0 decided / idle, 1 next evaluation counts the full mode's table evaluations, 2 the anchor mode's
  unsigned long long an_full_evals = 0;
  int lz_state = 0;            // lazy-scan trial of a small cloud: 0 idle, 1 the next evaluation runs in the lazy mode and is judged by its GSIP solves
  unsigned long long lz_cheap_solves = 0;
  bool ub_env = false;         // env SVSDF_UB_FULL=0/1/2 pins the mode, otherwise run_pipeline decides after one evaluation
  int ub_tune = 0;             // evaluations since the point set changed that took part in the decision (0 or 1)
  double ub_ratio = 0.0;       // GSIP solves / GSIP samples of the deciding (cheap-bound) evaluation
  double ub_threshold = 0.5;   // env SVSDF_UB_RATIO (analytic shapes; Polygon 0.2)
  bool ub_thr_env = false;
  bool scan_anchors = true;    // seed scans of k_round / k_tail evaluate the anchors of the circle test's survivors first (ChunkAnchor; env SVSDF_SCAN_ANCHORS=0: off; needs lipschitz_ok; same results)
  int round_list = 3;          // k_round: per-point candidate-chunk lists (env SVSDF_ROUND_LIST: bit 0 scans, bit 1 cheap bound use them; 0: all chunks; same results)
  bool cull = true;            // exact cull of provably inactive points in the main solve (env SVSDF_CULL=0 disables)
  bool cull_ok = false;        // this trajectory: duration not stale, slack table valid
  bool cull2 = true;           // second, value-based exact cull after the table scan (env SVSDF_CULL=1 turns only this one off)
  double slack_max = 0.0;      // max over the chunks of the linear continuous-path allowance
  int G_env = 0, G_late_env = 0;
  // batch count of a large shard in the scanning bound modes: chosen by timing real evaluations (any split gives the
  // same bits): 0 idle / done, 1 next evaluation learns the launch plan with one batch, 2.. timing candidate bt_k
  int bt_state = 0, bt_k = 0, bt_rep = 0, bt_ncand = 0, bt_cand[3] = {1, 1, 1};
  double bt_ms[3] = {0, 0, 0}, bt_samples[3] = {0, 0, 0};
  long long prev_nsolve[svsdf::kMaxIter] = {};  // solves per GSIP iteration of the previous evaluation (same point set)
  bool have_prev_nsolve = false;
  // fused GSIP tail (k_tail): all iterations from tail_iter on in one launch per batch
  int tail_mode = -1;                    // -1: by the previous evaluation's active counts, -2: off (launch chain only), >= 0: pinned
  long long tail_below = 0;              // auto: the whole GSIP loop runs in k_tail when the shard has at most this many interior
                                         // points; 0 = what one generation of waves holds (24 per CU: 6144 on 256 CUs)
  int tail_all_after = 1 << 30;          // steps of a point inside k_tail after which every sample is requested (-1: like the chain)
  int tail_iter = -1;                    // this evaluation: iteration the tail starts at (-1: none)
  bool tail_duo = true;                  // k_tail with one point per wave: both half-waves own the point and share its seed scans (env SVSDF_TAIL_DUO=0: off)
  bool tail_latency = true;              // small launches of k_tail (a wave slot per point at two waves per SIMD) use the instantiation that keeps its ~ 240 VGPRs (env SVSDF_TAIL_LATENCY=0: the 168-VGPR one everywhere; same results)
  bool tail_local = true;                // k_tail from iteration 0 keeps its points' GSIP state and samples in the wave's LDS (env SVSDF_TAIL_LOCAL=0: global arrays)
  long long prev_nactive[svsdf::kMaxIter] = {}; // active GSIP points per iteration of the previous evaluation, up to its tail
  int prev_tail_iter = -1;
  bool have_prev_nactive = false;
  long long wide32_below = 2000, wide16_below = 5000, wide8_below = 40000;  // env SVSDF_WIDE32 / SVSDF_WIDE16 / SVSDF_WIDE8
  bool select_env = false, all_iter_env = false;
  double scan_delta = 0.002;   // the same band in the scanning bound modes, where the bound is the sample's own table minimum (round 6 sweep, 0 ... 0.01:
                               // sdHeart / anchor mode - 3.5 % at 0.001 - 0.002 m -- 7 % fewer solves --, C3 / NS / C5 / C2 within +- 0.5 %; 0.01 m until round 5)
  double select_delta = 0.1;  // k_round: solve the samples whose upper bound is within this of the best one first

  // per-point / per-sub-query buffers
  double *d_sdf = nullptr, *d_t = nullptr;
  double *d_res_sdf = nullptr, *d_res_t = nullptr, *d_res_gx = nullptr, *d_res_gy = nullptr;
  svsdf::GsipState gs{};
  size_t icap = 0;                // interior capacity: entries of the per-interior-point arrays, stride of the sample arrays
  bool icap_fitted = false;       // capacity already fitted to a measured interior count of this point set
  double *d_block_partials = nullptr;
  size_t block_partials_cap = 0;  // doubles
  double *d_sums = nullptr;       // 19 * kMaxPieces + 1
  double *d_out = nullptr;        // [partial (19 * kMaxPieces + 1) | 8 x u64 stats]
  double *h_out = nullptr;        // pinned mirror
  double *h_out_dev = nullptr;    // the same buffer as the device addresses it (k_reduce writes the result there); null: copy
  unsigned *d_ticket = nullptr;   // k_reduce: finished-block counter (zero between launches)
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
  bool profile_span = false;     // svsdf_set_profiling(ctx, 3): device_ms only -- the span between the evaluation's first and last event, no per-launch events
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
  double combine_ms = 0.0, setup_ms = 0.0, fanout_ms = 0.0;
  bool group_serial = false;        // svsdf_set_group_serial: the stripes' evaluations one after the other (diagnostic)

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

namespace svsdf_impl {

constexpr size_t kOutPartial = 19 * svsdf::kMaxPieces + 1;
constexpr size_t kOutDoubles = kOutPartial + 14 + 2 * svsdf::kMaxIter;   // partial | 9 counters | solves per iteration | round scans | speculative | active per iteration | interior found | clock probe (2)
constexpr int kRepeat = -12345;   // finish(): more interior points than capacity -- the arrays were grown, repeat the evaluation

extern thread_local std::string g_last_error;
extern const char *kShapeNames[SVSDF_SHAPE_COUNT];
int fail(svsdf_ctx *ctx, int code, const std::string &msg);

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

// Lanes per query of the main solve by shard size: small shards are latency chains (wide groups shorten them), large ones
// throughput (narrow groups waste fewer lanes; the descent's ladders share the wave anyway).  Crossovers re-measured in
// round 4 after the shared ladders (profiles/r04_lanes_sweep.txt: 100 k points 8 -> 4 lanes - 5 %, 200 k 8 -> 2 lanes - 7 %).
// (Polygon: never below 4 -- its 2-lane kernel spills under the 3-waves register cap.)
inline int bound_mode_of(const svsdf_ctx *c) { return c->ub_full ? (c->ub_lazy ? 2 : ((c->ub_anchor && c->lipschitz_ok) ? 3 : 1)) : 0; }
inline int default_lanes(const svsdf_ctx *ctx, size_t Ps) {
  const int g = (Ps < 3000) ? 32 : (Ps < 20000) ? 16 : (Ps < 75000) ? 8 : (Ps < 150000) ? 4 : 2;
  return (ctx->cfg.shape_id == (int)svsdf::kPolygon) ? std::max(g, 4) : g;
}

// ---- svsdf_pipeline.hip
// Streams come from a process-wide pool and go back to it (round 5).  A context used to create its 9 streams and destroy
// them with itself; the NEXT context of the process then ran its concurrent point batches 7 % slower -- same kernels, same
// plan, same serialised times (tools/state_probe.py: a workload after any destroyed context 7.6 - 7.7 ms, beside a live
// one or in a fresh process 7.1 ms): the runtime maps streams to hardware queues when they are created and tears the
// queues down with the last stream, so the second generation of streams overlaps differently.  Pooled streams are created
// once per device in a fixed order and never destroyed: every context that is alone on its device gets set 0, i.e. the
// stream-to-queue mapping of a fresh process.
// Limits: (a) the "fresh process" mapping holds for a context that is ALONE on its device -- two live contexts
// get sets 0 and 1; (b) the pool is process-wide static state and its streams are never destroyed, so it must not be
// relied on across hipDeviceReset(): acquire_streams probes a pooled set with hipStreamQuery and re-creates a dead one,
// which covers a reset BETWEEN contexts, not one under a live context (that invalidates the context itself).
struct StreamSet { hipStream_t main = nullptr; hipStream_t batch[svsdf::kMaxBatches] = {}; int slot = -1; };
bool acquire_streams(int device, StreamSet &out);
void release_streams(int device, StreamSet &s);
int set_batches(svsdf_ctx *ctx, int nb);
int choose_tail_iter(const svsdf_ctx *ctx);
int swept_field(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, double *sdf_sorted);
int evaluate_points(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, bool allow_cull, bool with_partial);
void fill_mode_stats(svsdf_ctx *ctx);
int run_pipeline_leaf(svsdf_ctx *ctx, int N, const double *coeffs, const double *T);
void accumulate(int N, const double *partial, double *cost, double *gradT, double *gradC);
void shard_plan(const double *xyz, size_t P, int rk, int ws, int flags, std::vector<long long> &out);
struct CloudPlan {
  int device = 0;
  const double *d_xyz = nullptr;
  size_t P = 0;
  double *d_part = nullptr;
  unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
  const unsigned long long *sorted = nullptr;
  void *d_tmp = nullptr;
  void release();
};
int plan_cloud(svsdf_ctx *ctx, const double *d_xyz, size_t P, CloudPlan &plan);
int take_stripe(svsdf_ctx *ctx, svsdf_ctx *planner, const CloudPlan &plan, int rk, int ws);
int upload_shard_device(svsdf_ctx *ctx, const double *d_xyz, size_t P, int rk, int ws);
long long sincos_mismatches(svsdf_ctx *ctx, double lo, double hi, int n);
int debug_sdf_at(svsdf_ctx *ctx, int N, const double *coeffs, const double *T, size_t n, const double *pxy, const double *t, double *out8);
int site_stats(svsdf_ctx *ctx, unsigned long long out[20]);
// ---- svsdf_group.hip
int upload_group_device(svsdf_ctx *ctx, const double *d_xyz, size_t P);
int group_run(svsdf_ctx *ctx, const std::function<int(int)> &f);
void merge_stats(svsdf_ctx *ctx);
int run_pipeline(svsdf_ctx *ctx, int N, const double *coeffs, const double *T);   // leaf or group
int set_points_host(svsdf_ctx *ctx, const double *xyz, size_t P);
std::string group_init_rccl(svsdf_ctx *g);
svsdf_ctx *create_group(const svsdf_config *cfg, int ndev);
void destroy_group_resources(svsdf_ctx *ctx);
int rccl_comm_count(const svsdf_ctx *ctx);

}  // namespace svsdf_impl
