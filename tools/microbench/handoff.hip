// handoff.hip -- diagnostic: device-side cost of handing work from one stream to another, three ways.
//   hipcc --offload-arch=gfx950 -O3 handoff.hip -o handoff && ./handoff
// A chain of 2 N short kernels alternates between two streams; each kernel stamps the 100 MHz wall clock at its start and end.
// Hand-over: (0) none: one stream (baseline); (1) hipEventRecord + hipStreamWaitEvent (events without timing);
// (2) hipStreamWriteValue64 + hipStreamWaitValue64 on device memory; (3) the same on hipMallocSignalMemory.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k(long long *stamp, int idx, int spin) {
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * idx] = wall_clock64();
  for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(64);
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * idx + 1] = wall_clock64();
}
// the same chain on ONE stream with a kernel whose argument block is NB bytes (the fused kernel's is ~5 KB)
template <int NB> struct Blob { char b[NB]; };
template <int NB> __global__ void kbig(long long *stamp, int idx, int spin, Blob<NB> blob) {
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * idx] = wall_clock64() + (blob.b[idx % NB] & 0);
  for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(64);
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * idx + 1] = wall_clock64();
}
template <int NB> int run_big(int grid, int lds) {
  const int N = 200, spin = 100;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  long long *stamp; CK(hipMalloc(&stamp, sizeof(long long) * 4 * N)); CK(hipMemset(stamp, 0, sizeof(long long) * 4 * N));
  Blob<NB> blob; memset(&blob, 0, sizeof blob);
  CK(hipDeviceSynchronize());
  for (int i = 0; i < 2 * N; i++) hipLaunchKernelGGL(kbig<NB>, dim3(grid), dim3(64), lds, s, stamp, i, spin, blob);
  CK(hipDeviceSynchronize());
  std::vector<long long> h(4 * N);
  CK(hipMemcpy(h.data(), stamp, sizeof(long long) * 4 * N, hipMemcpyDeviceToHost));
  double gap = 0; int n = 0;
  for (int i = 2 * N / 4; i < 2 * N - 1; i++) { gap += (h[2 * (i + 1)] - h[2 * i + 1]) * 0.01; n++; }
  printf("grid %5d  one stream, %4d-byte argument block, %5d B of LDS per workgroup: gap to the next kernel %.2f us\n", grid, NB, lds, gap / n);
  return 0;
}
int run(int mode, int grid) {
  const int N = 200, spin = 100;
  hipStream_t s[2];
  CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  long long *stamp; CK(hipMalloc(&stamp, sizeof(long long) * 4 * N)); CK(hipMemset(stamp, 0, sizeof(long long) * 4 * N));
  std::vector<hipEvent_t> ev(2 * N);
  for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  uint64_t *flag = nullptr;
  if (mode == 2) { CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64)); }
  if (mode == 3) { CK(hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory)); CK(hipMemset(flag, 0, 8)); }
  CK(hipDeviceSynchronize());
  for (int i = 0; i < 2 * N; i++) {
    hipStream_t cur = mode == 0 ? s[0] : s[i & 1], nxt = mode == 0 ? s[0] : s[(i + 1) & 1];
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, cur, stamp, i, spin);
    if (mode == 1) { CK(hipEventRecord(ev[i], cur)); CK(hipStreamWaitEvent(nxt, ev[i], 0)); }
    if (mode >= 2) { CK(hipStreamWriteValue64(cur, flag, (uint64_t)(i + 1), 0)); CK(hipStreamWaitValue64(nxt, flag, (uint64_t)(i + 1), hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull)); }
  }
  CK(hipDeviceSynchronize());
  std::vector<long long> h(4 * N);
  CK(hipMemcpy(h.data(), stamp, sizeof(long long) * 4 * N, hipMemcpyDeviceToHost));
  double gap = 0, dur = 0; int n = 0;
  for (int i = 2 * N / 4; i < 2 * N - 1; i++) { gap += (h[2 * (i + 1)] - h[2 * i + 1]) * 0.01; dur += (h[2 * i + 1] - h[2 * i]) * 0.01; n++; }
  const char *names[] = {"one stream", "events", "write/wait value (hipMalloc)", "write/wait value (signal memory)"};
  printf("grid %5d  %-34s kernel %.1f us, gap to the next kernel %.2f us\n", grid, names[mode], dur / n, gap / n);
  return 0;
}
// visibility: kernel A (stream 0) fills a 256 MB buffer with the iteration number from every CU, kernel B (stream 1) checks it and counts
// mismatches, B hands back to the next A -- through write / wait value on device memory only (no event anywhere).
__global__ void fill(uint64_t *b, size_t n, uint64_t v) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = v + i; }
__global__ void check(const uint64_t *b, size_t n, uint64_t v, unsigned long long *bad) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += b[(i * 7919) % n] != v + (i * 7919) % n;
  if (c) atomicAdd(bad, c);
}
int visibility() {
  const size_t n = 32u << 20; const int N = 100;
  hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  uint64_t *b, *flag; unsigned long long *bad;
  CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64)); CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
  CK(hipDeviceSynchronize());
  uint64_t seq = 0;
  for (int i = 0; i < N; i++) {
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, s[0], b, n, (uint64_t)i * 1000003);
    CK(hipStreamWriteValue64(s[0], flag, ++seq, 0)); CK(hipStreamWaitValue64(s[1], flag, seq, hipStreamWaitValueGte, ~0ull));
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, s[1], b, n, (uint64_t)i * 1000003, bad);
    CK(hipStreamWriteValue64(s[1], flag + 1, seq, 0)); CK(hipStreamWaitValue64(s[0], flag + 1, seq, hipStreamWaitValueGte, ~0ull));
  }
  CK(hipDeviceSynchronize());
  unsigned long long h = 0; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
  printf("visibility across a write/wait-value hand-over: %llu mismatches in %d x %zu checked entries\n", h, N, n);
  return 0;
}
// does hipExtAnyOrderLaunch let a kernel start beside its predecessor in the SAME stream?  (hip_ext.h: "not supported on AMD GFX9xx boards")
int any_order() {
  const int N = 50, spin = 100;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  long long *stamp; CK(hipMalloc(&stamp, sizeof(long long) * 4 * N)); CK(hipMemset(stamp, 0, sizeof(long long) * 4 * N));
  CK(hipDeviceSynchronize());
  for (int i = 0; i < 2 * N; i++)
    hipExtLaunchKernelGGL(k, dim3(64), dim3(64), 0, s, nullptr, nullptr, (i & 1) ? hipExtAnyOrderLaunch : 0, stamp, i, spin);
  CK(hipDeviceSynchronize());
  std::vector<long long> h(4 * N);
  CK(hipMemcpy(h.data(), stamp, sizeof(long long) * 4 * N, hipMemcpyDeviceToHost));
  double ov = 0; int n = 0;
  for (int i = 10; i < 2 * N - 1; i += 2) { ov += (h[2 * (i + 1)] - h[2 * i + 1]) * 0.01; n++; }   // start of the any-order kernel minus end of its predecessor
  printf("any-order launch on one stream: the second kernel starts %.2f us after the first one ENDS (negative = beside it)\n", ov / n);
  return 0;
}
int main() {
  if (any_order()) printf("any-order test failed to run\n");
  if (visibility()) printf("visibility test failed to run\n");
  int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  for (int grid : {1, 2048}) { run_big<64>(grid, 0); run_big<2048>(grid, 0); run_big<3900>(grid, 0); run_big<64>(grid, 18496); run_big<3900>(grid, 18496); }
  for (int grid : {1, 2048}) for (int m = 0; m < 4; m++) if (run(m, grid)) printf("mode %d failed\n", m);
  return 0;
}
