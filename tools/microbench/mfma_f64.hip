// mfma_f64.hip -- diagnostic: what the f64 matrix pipe of gfx950 gives next to the f64 vector pipe.
//   hipcc --offload-arch=gfx950 -O3 mfma_f64.hip -o mfma_f64 && ./mfma_f64
// One 512-thread workgroup per CU = two waves on each SIMD (waves w and w + 4 share a SIMD).  Each half of the workgroup is
// given a ROLE: idle, a stream of v_fma_f64 (8 independent chains), a stream of v_mfma_f64_4x4x4_4b_f64 or of
// v_mfma_f64_16x16x4_f64 (NCH independent accumulators), or one wave that interleaves one matrix instruction with K vector
// FMAs.  Every wave times itself with the shader clock (s_memtime) and the constant 100 MHz counter, so the table gives
// cycles per instruction per wave, the clock the part really ran at under that load, and the f64 rate of the whole chip.
// Questions: (1) cycles per matrix instruction, alone and with two waves per SIMD; (2) does a matrix stream in one wave
// run BESIDE a vector stream in the other wave of the same SIMD (both at their own rates) or do they share one issue slot;
// (3) can ONE wave overlap its own matrix instructions with independent vector FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
enum Role { IDLE = 0, VALU = 1, MFMA4 = 2, MFMA16 = 3, MIX4 = 4, MIX16 = 5 };

__device__ inline long long wall100() { return wall_clock64(); }

template <int NCH>
__device__ void run_valu(int iters, double &sink) {
  double a[NCH];
  for (int i = 0; i < NCH; i++) a[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  const double m = 1.0000001, c = 1e-12;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 64 / NCH; r++)
#pragma unroll
      for (int i = 0; i < NCH; i++) a[i] = __builtin_fma(a[i], m, c);
  }
  for (int i = 0; i < NCH; i++) sink += a[i];
}
template <int NCH>
__device__ void run_mfma4(int iters, double &sink) {
  double acc[NCH];
  for (int i = 0; i < NCH; i++) acc[i] = 1e-9 * (threadIdx.x + i);
  const double a = 1.0 + 1e-7 * threadIdx.x, b = 1e-3;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 64 / NCH; r++)
#pragma unroll
      for (int i = 0; i < NCH; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  for (int i = 0; i < NCH; i++) sink += acc[i];
}
template <int NCH>
__device__ void run_mfma16(int iters, double &sink) {
  v4d acc[NCH];
  for (int i = 0; i < NCH; i++) acc[i] = v4d{1e-9 * threadIdx.x, 0., 0., 1e-9 * i};
  const double a = 1.0 + 1e-7 * threadIdx.x, b = 1e-3;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 64 / NCH; r++)
#pragma unroll
      for (int i = 0; i < NCH; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  for (int i = 0; i < NCH; i++) sink += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
}
// one wave: 16 x { one matrix instruction, KV independent vector FMAs }
template <int KV, bool BIG>
__device__ void run_mix(int iters, double &sink) {
  double acc4[4];
  v4d acc16[4];
  double v[8];
  for (int i = 0; i < 4; i++) { acc4[i] = 1e-9 * (threadIdx.x + i); acc16[i] = v4d{1e-9 * threadIdx.x, 0., 0., 1e-9 * i}; }
  for (int i = 0; i < 8; i++) v[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  const double a = 1.0 + 1e-7 * threadIdx.x, b = 1e-3, m = 1.0000001, c = 1e-12;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if (BIG) acc16[r % 4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[r % 4], 0, 0, 0);
      else acc4[r % 4] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[r % 4], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < KV; k++) v[(r * KV + k) % 8] = __builtin_fma(v[(r * KV + k) % 8], m, c);
    }
  }
  for (int i = 0; i < 4; i++) sink += acc4[i] + acc16[i].x + acc16[i].w;
  for (int i = 0; i < 8; i++) sink += v[i];
}

template <int KV>
__global__ __launch_bounds__(512) void k(double *out, long long *cyc, int role_lo, int role_hi, int nch, int iters) {
  const int wave = threadIdx.x >> 6, role = wave < 4 ? role_lo : role_hi;
  double sink = 0.;
  __syncthreads();
  const long long t0 = clock64(), w0 = wall100();
  if (role == VALU) run_valu<8>(iters, sink);
  else if (role == MFMA4) { if (nch == 1) run_mfma4<1>(iters, sink); else if (nch == 2) run_mfma4<2>(iters, sink); else if (nch == 4) run_mfma4<4>(iters, sink); else run_mfma4<8>(iters, sink); }
  else if (role == MFMA16) { if (nch == 1) run_mfma16<1>(iters, sink); else if (nch == 2) run_mfma16<2>(iters, sink); else if (nch == 4) run_mfma16<4>(iters, sink); else run_mfma16<8>(iters, sink); }
  else if (role == MIX4) run_mix<KV, false>(iters, sink);
  else if (role == MIX16) run_mix<KV, true>(iters, sink);
  const long long t1 = clock64(), w1 = wall100();
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = sink;
  if ((threadIdx.x & 63) == 0) { cyc[((size_t)blockIdx.x * 8 + wave) * 2] = t1 - t0; cyc[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = w1 - w0; }
}

static const char *rname(int r) { const char *n[] = {"idle", "v_fma_f64", "mfma 4x4x4_4b", "mfma 16x16x4", "mix 4x4x4", "mix 16x16x4"}; return n[r]; }
// f64 FMAs one wave instruction performs
static double fmas(int r, int kv) { return r == VALU ? 64. : r == MFMA4 ? 256. : r == MFMA16 ? 1024. : r == MIX4 ? (256. + 64. * kv) / (1 + kv) : r == MIX16 ? (1024. + 64. * kv) / (1 + kv) : 0.; }

template <int KV> void run(int lo, int hi, int nch) {
  const int grid = 256, iters = 3000;
  double *out; long long *cyc;
  hipMalloc(&out, (size_t)grid * 512 * sizeof(double)); hipMalloc(&cyc, (size_t)grid * 16 * sizeof(long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {   // the last repetition is the one reported (clock settled under this load)
    hipEventRecord(e0); hipLaunchKernelGGL(k<KV>, dim3(grid), dim3(512), 0, 0, out, cyc, lo, hi, nch, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<long long> h((size_t)grid * 16);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double c[2] = {0, 0}, w[2] = {0, 0};
  for (int b = 0; b < grid; b++) for (int wv = 0; wv < 8; wv++) { c[wv / 4] += h[(b * 8 + wv) * 2]; w[wv / 4] += h[(b * 8 + wv) * 2 + 1]; }
  const double n = grid * 4.0;
  // instructions per wave: 64 per iteration (mix: 16 matrix + 16 KV vector)
  auto per = [&](int role, int half) { const double inst = iters * (role >= MIX4 ? 16.0 * (1 + KV) : 64.0); return role == IDLE ? 0. : c[half] / n / inst; };
  const double ghz = lo != IDLE ? (c[0] / n) / (w[0] / n * 10.0) : (c[1] / n) / (w[1] / n * 10.0);   // shader cycles per ns
  const double inst_lo = lo == IDLE ? 0 : iters * (lo >= MIX4 ? 16.0 * (1 + KV) : 64.0), inst_hi = hi == IDLE ? 0 : iters * (hi >= MIX4 ? 16.0 * (1 + KV) : 64.0);
  const double tf = (inst_lo * fmas(lo, KV) + inst_hi * fmas(hi, KV)) * 4 * grid * 2 / (ms * 1e-3) * 1e-12;
  printf("%-14s | %-14s | chains %d kv %d | %7.2f | %7.2f | %5.2f GHz | %7.3f ms | %6.1f TFLOP/s\n", rname(lo), rname(hi), nch, KV,
         per(lo, 0), per(hi, 1), ghz, ms, tf);
  hipFree(out); hipFree(cyc); hipEventDestroy(e0); hipEventDestroy(e1);
}
int main() {
  printf("waves 0-3       | waves 4-7       |               | cyc/inst lo | hi | clock | kernel | chip f64 rate\n");
  // (1) each stream alone (one wave per SIMD) and doubled (two waves per SIMD)
  run<0>(VALU, IDLE, 8); run<0>(VALU, VALU, 8);
  for (int nch : {1, 2, 4, 8}) run<0>(MFMA4, IDLE, nch);
  run<0>(MFMA4, MFMA4, 4); run<0>(MFMA4, MFMA4, 8);
  for (int nch : {1, 2, 4, 8}) run<0>(MFMA16, IDLE, nch);
  run<0>(MFMA16, MFMA16, 4);
  // (2) matrix stream beside the other wave's vector stream
  run<0>(MFMA4, VALU, 4); run<0>(MFMA4, VALU, 8); run<0>(MFMA16, VALU, 4);
  // (3) one wave interleaving its own matrix and vector instructions
  run<1>(MIX4, IDLE, 4); run<2>(MIX4, IDLE, 4); run<4>(MIX4, IDLE, 4); run<4>(MIX4, MIX4, 4);
  run<4>(MIX16, IDLE, 4); run<8>(MIX16, IDLE, 4); run<16>(MIX16, IDLE, 4); run<16>(MIX16, MIX16, 4);
  return 0;
}
