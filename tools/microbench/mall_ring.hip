// mall_ring.hip -- diagnostic: does a SMALL buffer that is rewritten and re-read every few hundred microseconds stay in the 256 MB
// memory-side cache (MALL / Infinity Cache) while a large stream passes through it?  (Round 4: an E-vector RING reused by the segments
// of a pipelined apply instead of one full-size E-vector.)
//   hipcc --offload-arch=gfx950 -O3 mall_ring.hip -o mall_ring && ./mall_ring
// Per iteration, in one stream: W writes the buffer R (S bytes), T reads `stream_mb` MB of a 4 GB array (a different part each time: the
// stored-state stream of an apply), A reads R back.  Timed per kernel with events; reported as GB/s of W and A against S, for R = one
// buffer reused every iteration ("ring") and for R = a different S-byte slice of a 3 GB array each iteration ("fresh": what a full-size
// E-vector is).  If the ring lives in the MALL, A(ring) runs well above the HBM rate and A(fresh) at it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ void k_write(v2d *p, size_t n, double v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v2d{v, v + i};
}
__global__ void k_read(const v2d *p, size_t n, double *out) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { v2d v = p[i]; s += v.x + v.y; }
  if (s == 1.2345e-300) out[0] = s;
}
int main() {
  const size_t GB = 1ull << 30, MB = 1ull << 20;
  char *T, *F; double *out;
  if (hipMalloc(&T, 4 * GB) != hipSuccess || hipMalloc(&F, 3 * GB) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { puts("alloc failed"); return 1; }
  hipMemset(T, 1, 4 * GB); hipMemset(F, 1, 3 * GB);
  hipEvent_t e[4]; for (auto &x : e) hipEventCreate(&x);
  const int grid = 256 * 8, block = 256, iters = 12;
  printf("%8s %10s | %-28s | %-28s\n", "S (MB)", "stream MB", "ring: write GB/s, read GB/s", "fresh: write GB/s, read GB/s");
  for (size_t stream_mb : {300, 900}) for (size_t S : {8 * MB, 16 * MB, 32 * MB, 64 * MB, 128 * MB, 233 * MB}) {
    double res[2][2] = {{0, 0}, {0, 0}};
    for (int mode = 0; mode < 2; mode++) {
      double tw = 0, tr = 0; int cnt = 0;
      for (int it = 0; it < iters; it++) {
        char *R = mode == 0 ? F : F + ((size_t)it * S) % (3 * GB - S) / 256 * 256;
        char *Tp = T + ((size_t)it * stream_mb * MB) % (4 * GB - stream_mb * MB);
        hipEventRecord(e[0]);
        hipLaunchKernelGGL(k_write, dim3(grid), dim3(block), 0, 0, (v2d *)R, S / 16, 1.0 + it);
        hipEventRecord(e[1]);
        hipLaunchKernelGGL(k_read, dim3(grid), dim3(block), 0, 0, (const v2d *)Tp, stream_mb * MB / 16, out);
        hipEventRecord(e[2]);
        hipLaunchKernelGGL(k_read, dim3(grid), dim3(block), 0, 0, (const v2d *)R, S / 16, out);
        hipEventRecord(e[3]);
        hipEventSynchronize(e[3]);
        float a, b; hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[2], e[3]);
        if (it >= 4) { tw += a; tr += b; cnt++; }
      }
      res[mode][0] = S / (tw / cnt * 1e-3) * 1e-9; res[mode][1] = S / (tr / cnt * 1e-3) * 1e-9;
    }
    printf("%8zu %10zu | %12.0f %12.0f    | %12.0f %12.0f\n", S / MB, stream_mb, res[0][0], res[0][1], res[1][0], res[1][1]);
  }
  return 0;
}
