// cu_mask.hip -- which compute units does a stream created with hipExtStreamCreateWithCUMask use on the MI355X (8 XCDs x 32 CUs)?
// Every one-wave workgroup of a large grid records (XCC_ID, HW_ID); the host counts the distinct (xcc, se, sh, cu) tuples per
// mask.  Used to lay out the mask that keeps a few CUs free for RCCL's kernel beside the persistent fused grid.
//   hipcc --offload-arch=gfx950 -O2 cu_mask.hip -o cu_mask && ./cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_where(unsigned *out, int spin) {
  unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF;      // hwreg(HW_REG_XCC_ID, 0, 4)
  unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);             // hwreg(HW_REG_HW_ID, 0, 32)
  for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(64);          // stay resident so that the grid spreads over the allowed CUs
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static int run(const char *name, const std::vector<uint32_t> &mask, unsigned *d_out, int nblk) {
  hipStream_t s;
  CHK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  CHK(hipMemsetAsync(d_out, 0xFF, sizeof(unsigned) * 2 * nblk, s));
  hipLaunchKernelGGL(k_where, dim3(nblk), dim3(64), 0, s, d_out, 200);
  CHK(hipStreamSynchronize(s));
  std::vector<unsigned> h(2 * nblk);
  CHK(hipMemcpy(h.data(), d_out, sizeof(unsigned) * 2 * nblk, hipMemcpyDeviceToHost));
  std::set<unsigned> cus; int per_xcc[16] = {0};
  std::set<unsigned> per[16];
  for (int b = 0; b < nblk; b++) {
    const unsigned xcc = h[2 * b], hw = h[2 * b + 1];
    const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;   // gfx9 HW_ID layout
    const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    cus.insert(key); per[xcc & 15].insert(key);
  }
  int bits = 0; for (uint32_t w : mask) bits += __builtin_popcount(w);
  printf("%-34s bits set %3d -> distinct CUs used %3zu; per XCC:", name, bits, cus.size());
  for (int x = 0; x < 8; x++) printf(" %zu", per[x].size());
  printf("\n");
  (void)per_xcc;
  CHK(hipStreamDestroy(s));
  return 0;
}

int main() {
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount, nw = (ncu + 31) / 32, nblk = 16384;
  printf("%s: %d CUs\n", p.gcnArchName, ncu);
  unsigned *d_out; CHK(hipMalloc((void **)&d_out, sizeof(unsigned) * 2 * nblk));
  std::vector<uint32_t> all(nw, 0xFFFFFFFFu);
  if (run("all", all, d_out, nblk)) return 1;
  { std::vector<uint32_t> m(nw, 0u); m[0] = 1u; if (run("bit 0 only", m, d_out, nblk)) return 1; }
  { std::vector<uint32_t> m(nw, 0u); m[0] = 0xFFu; if (run("bits 0-7", m, d_out, nblk)) return 1; }
  { std::vector<uint32_t> m(nw, 0u); m[0] = 0xFFFFFFFFu; if (run("bits 0-31", m, d_out, nblk)) return 1; }
  { std::vector<uint32_t> m(all); m[nw - 1] &= 0x00FFFFFFu; if (run("all but the last 8 bits", m, d_out, nblk)) return 1; }
  { std::vector<uint32_t> m(all); m[0] &= ~0xFFu; if (run("all but bits 0-7", m, d_out, nblk)) return 1; }
  { std::vector<uint32_t> m(all); for (int w = 0; w < nw; w++) m[w] &= ~1u; if (run("all but bit 0 of every word", m, d_out, nblk)) return 1; }
  { std::vector<uint32_t> m(all); m[0] &= ~0xFFFFu; if (run("all but bits 0-15", m, d_out, nblk)) return 1; }
  return 0;
}
