// lat_probe.hip -- dependent-issue latencies on gfx950 that shape the piece-time chain (round 4):
// v_add_f64 chain, ds_read_b64 pointer chase, s_load pointer chase (scalar cache hit), one wave alone.
// build: hipcc --offload-arch=gfx950 -O3 -o lat_probe lat_probe.hip ; run: ./lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const __attribute__((address_space(4))) int *cint;
__global__ void k_add(double *out, double a, double b, long long *cyc, int n) {
  double r = a + threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    r = r - b; r = r - b; r = r - b; r = r - b; r = r - b; r = r - b; r = r - b; r = r - b;
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds(int *out, long long *cyc, int n) {
  __shared__ int tab[256];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) tab[i] = (i * 37 + 11) & 255;
  __syncthreads();
  int p = threadIdx.x & 255;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) { p = tab[p]; p = tab[p]; p = tab[p]; p = tab[p]; p = tab[p]; p = tab[p]; p = tab[p]; p = tab[p]; }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = p;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_sload(const int *tab_g, int *out, long long *cyc, int n) {
  cint tab = (cint)tab_g;
  int p = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) { p = tab[p]; p = tab[p]; p = tab[p]; p = tab[p]; p = tab[p]; p = tab[p]; p = tab[p]; p = tab[p]; }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = p;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double *d; long long *c; int *io, *tab;
  hipMalloc(&d, 8 * 1024); hipMalloc(&c, 64); hipMalloc(&io, 4 * 1024); hipMalloc(&tab, 4 * 256);
  int h[256]; for (int i = 0; i < 256; ++i) h[i] = (i * 37 + 11) & 255;
  hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
  long long hc = 0; const int n = 2000;
  for (int waves = 1; waves <= 4; waves *= 2) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_add, dim3(1), dim3(64 * waves), 0, 0, d, 1e9, 1.0, c, n);
    hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    printf("v_add_f64 dependent chain, %d wave(s) in the block: %.2f cycles/op (shader clock counter)\n", waves, (double)hc / (8.0 * n));
  }
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, io, c, n);
  hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
  printf("ds_read_b32 pointer chase, 1 wave: %.2f cycles/op\n", (double)hc / (8.0 * n));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_sload, dim3(1), dim3(64), 0, 0, tab, io, c, n);
  hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
  printf("s_load_dword pointer chase (scalar cache hit), 1 wave: %.2f cycles/op\n", (double)hc / (8.0 * n));
  return 0;
}
