// lds_rate.hip -- diagnostic: LDS throughput per CU for the access shapes of the pencil kernel.
//   hipcc --offload-arch=gfx950 -O3 lds_rate.hip -o lds_rate && ./lds_rate
// Every wave issues N LDS instructions of one kind on conflict-free addresses (lane * width); 8 waves per CU
// (2 per SIMD), one workgroup of 64 per wave as in the kernel.  Reports cycles of the CU's LDS per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, int iters) {
  __shared__ __attribute__((aligned(16))) double buf[2304];   // 18 KB: 8 of these fit a CU, as in the kernel
  const int lane = threadIdx.x;
  for (int i = lane; i < 2304; i += 64) buf[i] = i;
  __syncthreads();
  typedef __attribute__((address_space(3))) double lds_d;
  volatile lds_d *p = (volatile lds_d *)buf + (KIND == 1 ? 2 * lane : lane);
  double acc = 0.;
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) {        // 16 x ds_read_b64, consecutive lanes 8 bytes apart, immediate offsets
#pragma unroll
      for (int j = 0; j < 16; j++) acc += p[j * 64];
    } else if (KIND == 1) { // 16 x ds_read_b128 (two doubles per lane)
#pragma unroll
      for (int j = 0; j < 16; j++) {
        v2d v;
        asm volatile("ds_read_b128 %0, %1 offset:%2\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(lds_d *)p), "n"(0));
        acc += v.x + v.y;
      }
    } else if (KIND == 2) { // 16 x ds_write_b64
#pragma unroll
      for (int j = 0; j < 16; j++) p[j * 64] = acc + j;
    } else if (KIND == 4) { // 16 x ds_write2_b64: two doubles per lane, 200 B apart (a strided pencil's outputs)
#pragma unroll
      for (int j = 0; j < 16; j++)
        asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" : : "v"((unsigned)(size_t)(lds_d *)((volatile lds_d *)buf + lane)), "v"(acc), "v"(acc + 1.0), "n"(0), "n"(25) : "memory");
    } else if (KIND == 5) { // 16 x ds_write_b128, 16 bytes per lane
#pragma unroll
      for (int j = 0; j < 16; j++) {
        v2d v; v.x = acc; v.y = acc + j;
        asm volatile("ds_write_b128 %0, %1" : : "v"((unsigned)(size_t)(lds_d *)((volatile lds_d *)buf + 2 * lane)), "v"(v) : "memory");
      }
    } else if (KIND == 6) { // 16 x ds_read2_b64
#pragma unroll
      for (int j = 0; j < 16; j++) {
        v2d v;
        asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(lds_d *)((volatile lds_d *)buf + lane)), "n"(0), "n"(25));
        acc += v.x + v.y;
      }
    } else {                // pencil-like: 5 reads at stride 40 B per lane (i-direction pencil, lanes 200 B apart)
      volatile lds_d *q = (volatile lds_d *)buf + (lane % 25) * 5 + (lane / 25) * 125;
#pragma unroll
      for (int j = 0; j < 15; j++) acc += q[(j % 5) + (j / 5) * 375];
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 64 + lane] = acc;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND> void run(const char *name, int per_iter) {
  const int grid = 256 * 8, iters = 2000;
  double *out; long long *cyc;
  hipMalloc(&out, grid * 64 * sizeof(double)); hipMalloc(&cyc, grid * sizeof(long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, out, cyc, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  const double inst_per_cu = 8.0 * iters * per_iter;
  printf("%-34s kernel %.3f ms: %.2f ns of CU time per LDS instruction (%.1f cycles at 2.3 GHz)\n", name, ms,
         ms * 1e6 / inst_per_cu, ms * 1e6 / inst_per_cu * 2.3);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("ds_read_b64, unit stride", 16);
  run<1>("ds_read_b128 (+wait each)", 16);
  run<2>("ds_write_b64, unit stride", 16);
  run<3>("ds_read_b64, pencil pattern", 15);
  run<4>("ds_write2_b64 (2 x 8 B per lane)", 16);
  run<5>("ds_write_b128 (16 B per lane)", 16);
  run<6>("ds_read2_b64 (+wait each)", 16);
  return 0;
}
