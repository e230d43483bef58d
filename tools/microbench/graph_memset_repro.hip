// graph_memset_repro.hip -- does a recorded hipMemsetAsync node keep its order against neighbouring kernel nodes on replay,
// when eager launches of the same kernels happen between capture and replay?  (the library records zero-fills
// as a fill KERNEL because a memset node was seen to lose its ordering; this is the library-free check of that claim.)
//
// Recorded on a capturing stream:  k_set(buf, 7) ; hipMemsetAsync(buf, 0) ; k_add(buf, 1)   -> every word must be 1.
// If the memset ran BEFORE k_set the words are 8, if AFTER k_add they are 0.
// build: hipcc --offload-arch=gfx950 -O3 -o graph_memset_repro graph_memset_repro.hip ; run: ./graph_memset_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_set(double *p, size_t n, double v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void k_add(double *p, size_t n, double v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += v; }

int main() {
  const size_t sizes[] = {1 << 10, 1 << 16, 1 << 22, 19537320};
  hipStream_t s, cap;
  CHECK(hipStreamCreate(&s));
  CHECK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  int bad_total = 0;
  for (size_t n : sizes) {
    double *buf;
    CHECK(hipMalloc(&buf, n * sizeof(double)));
    for (int eager_between = 0; eager_between < 2; eager_between++) {
      hipGraph_t g; hipGraphExec_t ge;
      CHECK(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
      hipLaunchKernelGGL(k_set, dim3(1024), dim3(256), 0, cap, buf, n, 7.0);
      CHECK(hipMemsetAsync(buf, 0, n * sizeof(double), cap));
      hipLaunchKernelGGL(k_add, dim3(1024), dim3(256), 0, cap, buf, n, 1.0);
      CHECK(hipStreamEndCapture(cap, &g));
      CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      int bad = 0;
      std::vector<double> h(n);
      for (int rep = 0; rep < 20; rep++) {
        if (eager_between) {   // eager launches of the same kernels and an eager memset on the stream the graph is launched on
          hipLaunchKernelGGL(k_set, dim3(1024), dim3(256), 0, s, buf, n, 3.0);
          CHECK(hipMemsetAsync(buf, 0, n * sizeof(double), s));
          hipLaunchKernelGGL(k_add, dim3(1024), dim3(256), 0, s, buf, n, 5.0);
        }
        CHECK(hipGraphLaunch(ge, s));
        CHECK(hipMemcpyAsync(h.data(), buf, n * sizeof(double), hipMemcpyDeviceToHost, s));
        CHECK(hipStreamSynchronize(s));
        size_t wrong = 0;
        for (size_t i = 0; i < n; i++) wrong += (h[i] != 1.0);
        if (wrong) { bad++; if (bad <= 2) printf("   n=%zu eager_between=%d rep %d: %zu words wrong (first value %g)\n", n, eager_between, rep, wrong, h[0]); }
      }
      printf("n=%9zu  eager launches between replays: %d   replays with a wrong result: %d of 20\n", n, eager_between, bad);
      bad_total += bad;
      CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
    }
    CHECK(hipFree(buf));
  }
  printf("%s\n", bad_total ? "REPRODUCED: a recorded memset node lost its order" : "not reproduced: memset nodes kept their order in every replay");
  return 0;
}
