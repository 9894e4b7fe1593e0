// Micro-benchmark (development): latency of the accesses a persistent task-queue kernel hands data over with on gfx950 --
// dependent plain loads vs relaxed agent-scope atomic loads (sc1), a returning atomicAdd, and a write-through store +
// completion wait -- for one wave alone and for every wave of a full-machine grid.
//   hipcc --offload-arch=gfx950 -O3 -o coh_latency coh_latency.hip && ./coh_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void k_chase(const unsigned long long *next, unsigned long long *ctr, unsigned long long *sink, int iters, int mode,
                        unsigned long long n, long long *out) {
  const int lane = threadIdx.x & 63;
  const unsigned long long wave = (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) / 64;
  unsigned long long p = (wave * 7919ull + lane * 104729ull) % n;
  const long long t0 = clock64();
  if (mode == 0) {
    for (int i = 0; i < iters; ++i) p = next[p];
  } else if (mode == 1) {
    for (int i = 0; i < iters; ++i) p = __hip_atomic_load(next + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (mode == 2) {   // returning atomic on a per-wave address (lane 0)
    for (int i = 0; i < iters; ++i)
      if (lane == 0) p += atomicAdd(ctr + (wave % 4096) * 16, 1ull) & 1ull;
  } else if (mode == 3) {   // write-through store + wait for completion
    for (int i = 0; i < iters; ++i) {
      __hip_atomic_store(sink + (p + i) % n, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
  } else if (mode == 4) {   // agent-scope fence (cache-wide write-back + invalidate)
    for (int i = 0; i < iters; ++i) {
      sink[(p + i) % n] = p;
      __threadfence();
    }
  } else if (mode == 5) {   // returning atomic on ONE shared address
    for (int i = 0; i < iters; ++i)
      if (lane == 0) p += atomicAdd(ctr, 1ull) & 1ull;
  }
  const long long t1 = clock64();
  if (lane == 0) out[wave] = t1 - t0;
  if (p == 0xdeadbeefdeadbeefull) sink[0] = p;
}

int main() {
  const unsigned long long n = 1ull << 24;   // 128 MB of 8-byte links
  std::vector<unsigned long long> h(n);
  std::iota(h.begin(), h.end(), 0ull);
  std::mt19937_64 rng(1);
  for (unsigned long long i = n - 1; i > 0; --i) std::swap(h[i], h[rng() % i]);   // Sattolo: one cycle
  unsigned long long *d_next, *d_ctr, *d_sink;
  long long *d_out;
  hipMalloc(&d_next, n * 8); hipMalloc(&d_sink, n * 8); hipMalloc(&d_ctr, 4096 * 16 * 8); hipMalloc(&d_out, 8 * 65536);
  hipMemcpy(d_next, h.data(), n * 8, hipMemcpyHostToDevice);
  hipMemset(d_ctr, 0, 4096 * 16 * 8);
  const char *names[6] = {"plain dependent load", "agent-scope atomic load", "returning atomicAdd (own address)",
                          "write-through store + wait", "plain store + agent fence", "returning atomicAdd (one address)"};
  int dev_clock = 0;
  hipDeviceGetAttribute(&dev_clock, hipDeviceAttributeClockRate, 0);
  for (int waves : {1, 3072}) {
    for (int mode = 0; mode < 6; ++mode) {
      const int iters = (mode == 4 && waves > 1) ? 20 : 200;
      const int blocks = waves;   // one wave per block
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k_chase, dim3(blocks), dim3(64), 0, 0, d_next, d_ctr, d_sink, 10, mode, n, d_out);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_chase, dim3(blocks), dim3(64), 0, 0, d_next, d_ctr, d_sink, iters, mode, n, d_out);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> o(waves);
      hipMemcpy(o.data(), d_out, waves * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (auto v : o) avg += (double)v; avg /= waves;
      std::printf("waves %4d  %-36s  %8.3f us per op (kernel)  %10.0f ticks per op (clock64)\n", waves, names[mode],
                  1e3 * ms / iters, avg / iters);
    }
  }
  std::printf("clock rate attribute: %d kHz\n", dev_clock);
  return 0;
}
