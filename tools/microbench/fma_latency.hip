// fma_latency.hip -- diagnostic: issue interval of v_fma_f64 for a single wave per SIMD as a function of the number of
// independent dependency chains (1 chain = pure latency).  hipcc --offload-arch=gfx950 -O3 fma_latency.hip -o fma_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, int iters) {
  double a[NCH];
  for (int i = 0; i < NCH; i++) a[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  const double m = 1.0000001, c = 1e-12;
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 64 / NCH; r++)
#pragma unroll
      for (int i = 0; i < NCH; i++) a[i] = __builtin_fma(a[i], m, c);
  }
  const long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < NCH; i++) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NCH> void run(int waves_per_cu) {
  const int grid = 256 * waves_per_cu, iters = 4000;
  double *out; long long *cyc;
  hipMalloc(&out, grid * 64 * sizeof(double)); hipMalloc(&cyc, grid * sizeof(long long));
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<NCH>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
  long long h; hipMemcpy(&h, cyc + grid / 2, sizeof h, hipMemcpyDeviceToHost);
  printf("chains %d, waves/CU %2d: %.2f s_memtime ticks per FMA per wave\n", NCH, waves_per_cu, (double)h / (64.0 * iters));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w : {4, 8}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); run<16>(w); }
  return 0;
}
