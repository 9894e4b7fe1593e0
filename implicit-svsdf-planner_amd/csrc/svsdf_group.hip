// svsdf_group.hip -- in-process multi-GPU context (svsdf_config::n_devices): one single-device sub-context and one host
// thread per device, the cloud planned once and handed out in stripes, the partials summed on the host in a fixed order
// or by an RCCL all-reduce (ncclCommInitAll, one rank per device).  Reference shape: one process, one optimizer
// (plan_manager.cpp:168-199), one sum at the end of the OpenMP loop (BEO:855-863).
#include "svsdf_ctx.hpp"

using namespace svsdf;

namespace svsdf_impl {

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
  using clk = std::chrono::steady_clock;
  std::vector<clk::time_point> ts(G), te(G);
  auto job = [&f, &ts, &te](int k) { ts[k] = clk::now(); const int r = f(k); te[k] = clk::now(); return r; };
  int rc = SVSDF_OK;
  auto collect = [&](int k, int r) {
    if (r && !rc) {
      rc = r;
      ctx->err = "device " + std::to_string(ctx->subs[k]->device) + " (stripe " + std::to_string(k) + "): " + ctx->subs[k]->err;
      g_last_error = ctx->err;
    }
  };
  double wake = 0.0, join = 0.0;
  auto ms = [](clk::duration d) { return std::chrono::duration<double, std::milli>(d).count(); };
  if (ctx->group_serial) {
    // diagnostic (svsdf_set_group_serial): the stripes one after the other, so that on a box with fewer GPUs than stripes
    // every stripe's device time is its own -- what it would take on a GPU of its own
    for (int k = 0; k < G; ++k) {
      const clk::time_point t_post = clk::now();
      ctx->workers[k]->post([&job, k] { return job(k); });
      collect(k, ctx->workers[k]->wait());
      wake += ms(ts[k] - t_post);
      join += ms(clk::now() - te[k]);
    }
  } else {
    const clk::time_point t_post = clk::now();
    for (int k = 0; k < G; ++k) ctx->workers[k]->post([&job, k] { return job(k); });
    for (int k = 0; k < G; ++k) collect(k, ctx->workers[k]->wait());
    const clk::time_point t_done = clk::now();
    clk::time_point last_start = ts[0], last_end = te[0];
    for (int k = 1; k < G; ++k) { last_start = std::max(last_start, ts[k]); last_end = std::max(last_end, te[k]); }
    wake = ms(last_start - t_post);
    join = ms(t_done - last_end);
  }
  ctx->fanout_ms = wake + join;
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
    if (a.shader_clock_mhz > 0.0) t.shader_clock_mhz = (t.shader_clock_mhz > 0.0) ? std::min(t.shader_clock_mhz, a.shader_clock_mhz) : a.shader_clock_mhz;
  }
  t.bound_mode_decided = 1;
  t.plan_settled = 1;
  for (svsdf_ctx *s : ctx->subs) { t.bound_mode_decided &= s->stats.bound_mode_decided; t.plan_settled &= s->stats.plan_settled; }
  t.n_devices = (int)ctx->subs.size();
  t.combine = ctx->combine;
  t.combine_ms = ctx->combine_ms;
  t.fanout_ms = ctx->fanout_ms;
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
  //  a failure from here on must not leave a half-initialised group behind: svsdf_set_combine's early-out above
  // would take the populated `comms` for a ready communicator and run the all-reduce into null buffers
  auto undo = [&](const char *m) -> std::string {
    for (int k = 0; k < G; ++k)
      if (g->comms[k] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g->comms[k]);
    g->comms.clear();
    for (size_t k = 0; k < g->d_red.size(); ++k)
      if (g->d_red[k]) { (void)hipSetDevice(devs[k]); (void)hipFree(g->d_red[k]); }
    g->d_red.clear();
    return m;
  };
  g->d_red.assign(G, nullptr);
  for (int k = 0; k < G; ++k)
    if (hipSetDevice(devs[k]) != hipSuccess || hipMalloc((void **)&g->d_red[k], kOutPartial * sizeof(double)) != hipSuccess)
      return undo("allocation of the all-reduce buffer failed");
  if (!g->h_red && hipHostMalloc((void **)&g->h_red, kOutPartial * sizeof(double), hipHostMallocDefault) != hipSuccess)
    return undo("pinned allocation failed");
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
  g->cfg.polygon_loop_sizes = nullptr;
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

// svsdf_destroy of a group: threads, communicators, all-reduce buffers (the sub-contexts are destroyed by the caller)
void destroy_group_resources(svsdf_ctx *ctx) {
  ctx->workers.clear();   // joins the threads
  for (size_t k = 0; k < ctx->comms.size(); ++k)
    if (ctx->comms[k] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comms[k]);
  for (size_t k = 0; k < ctx->d_red.size(); ++k)
    if (ctx->d_red[k]) { (void)hipSetDevice(ctx->subs[k]->device); (void)hipFree(ctx->d_red[k]); }
  if (ctx->h_red) (void)hipHostFree(ctx->h_red);
}

// rank count of the group's communicator as RCCL reports it (0: none)
int rccl_comm_count(const svsdf_ctx *ctx) {
  int n = 0;
  if (!ctx->comms.empty() && ctx->comms[0] && g_rccl.CommCount && g_rccl.CommCount(ctx->comms[0], &n) == 0) return n;
  return 0;
}

}  // namespace svsdf_impl
