// qstream.hip -- how fast can persistent waves stream the q-point data in the fused kernel's access
// pattern, with no compute?  Variants of the layout / load width, same bytes.
//   hipcc --offload-arch=gfx950 -O3 -o qstream qstream.hip && ./qstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int NC = 19;  // 10 geometry + 9 state components per point

// A: [elem][comp][125] doubles, two point slots per lane (q = lane, lane + 64): the shipped layout
__global__ void kA(const double *q, double *out, int nelem, int depth) {
  const int lane = threadIdx.x, nw = gridDim.x;
  double acc = 0.;
  for (int e = blockIdx.x; e < nelem; e += nw) {
    const double *b = q + (size_t)e * NC * 125;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int p = lane + 64 * s < 125 ? lane + 64 * s : 124;
#pragma unroll
      for (int c = 0; c < NC; c++) acc += b[c * 125 + p];
    }
  }
  if (acc == 1.2345e-300) out[blockIdx.x * 64 + lane] = acc;
}
// B: [elem][comp][128] (padded: every wave-load is an aligned 512-byte run)
__global__ void kB(const double *q, double *out, int nelem, int depth) {
  const int lane = threadIdx.x, nw = gridDim.x;
  double acc = 0.;
  for (int e = blockIdx.x; e < nelem; e += nw) {
    const double *b = q + (size_t)e * NC * 128;
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
      for (int c = 0; c < NC; c++) acc += b[c * 128 + lane + 64 * s];
    }
  }
  if (acc == 1.2345e-300) out[blockIdx.x * 64 + lane] = acc;
}
// C: [elem][slot][comp pair][64 lanes][2]: 16 bytes per lane per load (dwordx4), 1 KB per wave-load
__global__ void kC(const double2 *q, double *out, int nelem, int depth) {
  const int lane = threadIdx.x, nw = gridDim.x;
  double acc = 0.;
  for (int e = blockIdx.x; e < nelem; e += nw) {
    const double2 *b = q + (size_t)e * 10 * 128;
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
      for (int c = 0; c < 10; c++) { const double2 v = b[(s * 10 + c) * 64 + lane]; acc += v.x + v.y; }
    }
  }
  if (acc == 1.2345e-300) out[blockIdx.x * 64 + lane] = acc;
}
// D: plain streaming read, 16 bytes per lane, grid-stride (the ceiling)
__global__ void kD(const double2 *q, double *out, size_t n2) {
  double acc = 0.;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { const double2 v = q[i]; acc += v.x + v.y; }
  if (acc == 1.2345e-300) out[threadIdx.x] = acc;
}

int main() {
  const int nelem = 99000;
  const size_t bytes = (size_t)nelem * 20 * 128 * 8;  // big enough for every variant
  double *q, *out;
  CHK(hipMalloc(&q, bytes)); CHK(hipMalloc(&out, 1 << 22));
  CHK(hipMemset(q, 0, bytes));
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  auto time = [&](const char *name, double gb, auto launch) {
    launch(); launch();
    hipEventRecord(a); for (int i = 0; i < 10; i++) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-58s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, gb / (ms * 1e-3));
  };
  for (int wpc : {4, 8, 16, 32}) {
    const int grid = 256 * wpc;
    char nm[128];
    snprintf(nm, sizeof nm, "A [e][19][125], 8 B/lane, %2d waves/CU", wpc);
    time(nm, nelem * 19.0 * 125 * 8 / 1e9, [&] { hipLaunchKernelGGL(kA, dim3(grid), dim3(64), 0, 0, q, out, nelem, 0); });
    snprintf(nm, sizeof nm, "B [e][19][128] padded, 8 B/lane, %2d waves/CU", wpc);
    time(nm, nelem * 19.0 * 128 * 8 / 1e9, [&] { hipLaunchKernelGGL(kB, dim3(grid), dim3(64), 0, 0, q, out, nelem, 0); });
    snprintf(nm, sizeof nm, "C [e][2][10][64][2], 16 B/lane, %2d waves/CU", wpc);
    time(nm, nelem * 20.0 * 128 * 8 / 1e9, [&] { hipLaunchKernelGGL(kC, dim3(grid), dim3(64), 0, 0, (const double2 *)q, out, nelem, 0); });
  }
  time("D plain 16 B/lane grid-stride read, 2048 x 256", bytes / 1e9, [&] { hipLaunchKernelGGL(kD, dim3(2048), dim3(256), 0, 0, (const double2 *)q, out, bytes / 16); });
  return 0;
}
