// clocks.hip -- diagnostic: shader clock under f64 FMA load and issue cost of v_fma_f64 on this part.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/clocks.hip -o tools/microbench/clocks && tools/microbench/clocks
// clock64() = s_memtime, wall_clock64() = s_memrealtime (constant 100 MHz).  Each wave runs `iters` rounds of
// 64 independent-chain FMAs (8 chains x 8) and stamps both counters around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(64) void k_fma(double *out, long long *stamps, int iters) {
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  const double m = 1.0000001, c = 1e-12;
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = __builtin_fma(a[i], m, c);
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  printf("CUs %d, clockRate %d kHz\n", ncu, prop.clockRate);
  for (int wpc : {1, 4, 8, 16}) {
    const int grid = ncu * wpc, iters = 20000;
    double *out; long long *st;
    hipMalloc(&out, grid * 64 * sizeof(double));
    hipMalloc(&st, grid * 2 * sizeof(long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_fma, dim3(grid), dim3(64), 0, 0, out, st, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid * 2);
    hipMemcpy(h.data(), st, grid * 2 * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> ratio, wall;
    for (int i = 0; i < grid; i++) { ratio.push_back((double)h[2 * i] / (double)h[2 * i + 1]); wall.push_back((double)h[2 * i + 1]); }
    std::sort(ratio.begin(), ratio.end()); std::sort(wall.begin(), wall.end());
    const double wall_s = wall[grid / 2] * 1e-8;           // 100 MHz ticks
    const double fma_per_wave = 64.0 * iters;
    printf("waves/CU %2d: kernel %.3f ms; s_memtime/s_memrealtime median %.3f (=> %.0f MHz if s_memtime is the shader clock); "
           "wave loop %.3f ms => %.2f ns per wave-FMA per wave, %.2f ns per FMA issued on a SIMD (x waves/SIMD %.2f)\n",
           wpc, ms, ratio[grid / 2], 100.0 * ratio[grid / 2], wall_s * 1e3, wall_s * 1e9 / fma_per_wave,
           wall_s * 1e9 / fma_per_wave / std::max(1.0, wpc / 4.0), std::max(1.0, wpc / 4.0));
    printf("             f64 rate %.1f TFLOP/s\n", 2.0 * 64 * fma_per_wave * grid / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(st);
  }
  return 0;
}
