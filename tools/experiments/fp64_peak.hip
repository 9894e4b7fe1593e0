// fp64_peak.hip -- what the FP64 vector pipes of this GPU deliver, measured: the denominators of bench.py's `roofline.fp64`
// (78.6 TFLOP/s = v_fma_f64 at full rate on 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz x 2 flop; the product is built
// -ffp-contract=off, so its own ceiling is the un-fused rate, 1 flop per lane-op: 39.3 TFLOP/s).
//   hipcc --offload-arch=gfx950 -O3 -o fp64_peak.bin tools/experiments/fp64_peak.hip && ./fp64_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>   // 0: fma chains, 1: mul + add chains (separate instructions), 2: add chains
__global__ void __launch_bounds__(256) k_peak(double *out, double a, double b, int iters) {
  double x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = (double)(threadIdx.x + k) * 1e-3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 0) x[k] = __builtin_fma(x[k], a, b);
      else if (MODE == 1) { double m; asm volatile("v_mul_f64 %0, %1, %2" : "=v"(m) : "v"(x[k]), "v"(a)); asm volatile("v_add_f64 %0, %1, %2" : "=v"(x[k]) : "v"(m), "v"(b)); }
      else asm volatile("v_add_f64 %0, %1, %2" : "=v"(x[k]) : "v"(x[k]), "v"(b));
    }
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  if (s == 123.456) out[0] = s;
}

template <int MODE>
static void run(const char *name, double flop_per_op_slot) {
  double *d;
  hipMalloc(&d, 8);
  const int iters = 1 << 14, blocks = 256 * 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_peak<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0.999999, 1e-9, 64);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_peak<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0.999999, 1e-9, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * 256 * iters * 8;
  std::printf("%-28s %8.3f ms  %7.2f T lane-ops/s  %7.2f TFLOP/s\n", name, ms, ops * ((MODE == 1) ? 2 : 1) / ms / 1e9, ops * flop_per_op_slot / ms / 1e9);
  hipFree(d);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  std::printf("%s, %d CUs, %.0f MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1e3);
  run<0>("v_fma_f64 (2 flop/op)", 2.0);
  run<1>("v_mul_f64 + v_add_f64", 2.0);
  run<2>("v_add_f64", 1.0);
  return 0;
}
