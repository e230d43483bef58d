// xcd_handoff.hip -- which same-XCD producer -> consumer hand-off forms are sound inside ONE launch on gfx950?
//
// Question behind the gated assembly (csrc/kernels_misc.hip, k_assemble_gated): a wave stores data with PLAIN stores, waits
// for them (s_waitcnt vmcnt(0)) and bumps a counter with an agent-scope atomic add; another wave ON THE SAME XCD polls the
// counter and then reads the data.  Which poll (sc1 load / returning atomic) sees the adds, and which data loads (plain /
// sc1) see the stores, when the consumer's L1 and the XCD's L2 already hold OLD copies of both lines?
//
// Every wave is producer of its own slot and consumer of its partner's (ticket ^ 1 within its XCD, tickets handed out
// per XCD from HW_REG_XCC_ID), for ROUNDS rounds with new values each round.  Before polling, the consumer reads the
// partner's slot (old values -> L1/L2 warm) and the flag.  Counted per mode: polls that timed out, stale words.
//
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_handoff xcd_handoff.hip ;  run: ./xcd_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int SLOT = 256;      // doubles per slot (2 KB)
constexpr int MAXW = 128;      // waves per XCD at most
constexpr int ROUNDS = 200;

struct Stat { unsigned long long timeouts, stale, reads, no_partner, poll_iters, cross; };

__device__ __forceinline__ unsigned poll_value(unsigned *p, int mode) {
  if (mode == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // global_load sc1
  if (mode == 1) return atomicCAS(p, 0xFFFFFFFFu, 0xFFFFFFFFu);                                      // a real returning atomic
  return *(volatile unsigned *)p;                                                                    // plain load
}

// poll_mode: 0 sc1 load, 1 CAS, 2 plain;  read_mode: 0 plain, 1 sc1;  cross: partner on the NEXT XCD instead of the same one
__global__ __launch_bounds__(64) void k_handoff(double *data, unsigned *flag, unsigned *ticket, unsigned *nres, Stat *st,
                                                int poll_mode, int read_mode, int cross, int nwaves_expected) {
  const int lane = threadIdx.x;
  const int xcd = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);
  unsigned t = 0;
  if (lane == 0) t = atomicAdd(ticket + xcd * 32, 1u);
  t = __builtin_amdgcn_readfirstlane(t);
  if (t >= MAXW) return;
  // wait (bounded) until every wave of the grid has its ticket, so that partners exist
  if (lane == 0) {
    atomicAdd(nres, 1u);
    for (int i = 0; i < (1 << 16) && atomicCAS(nres, 0xFFFFFFFFu, 0xFFFFFFFFu) < (unsigned)nwaves_expected; i++) __builtin_amdgcn_s_sleep(16);
  }
  const unsigned nx_mine = atomicCAS(ticket + xcd * 32, 0xFFFFFFFFu, 0xFFFFFFFFu);
  const int pxcd = cross ? (xcd + 1) & 7 : xcd;
  const unsigned nx_p = atomicCAS(ticket + pxcd * 32, 0xFFFFFFFFu, 0xFFFFFFFFu);
  const unsigned partner = cross ? t : (t ^ 1u);
  (void)nx_mine;
  double *mine = data + ((size_t)xcd * MAXW + t) * SLOT, *theirs = data + ((size_t)pxcd * MAXW + partner) * SLOT;
  unsigned *myflag = flag + (xcd * MAXW + t) * 32, *pflag = flag + (pxcd * MAXW + partner) * 32;
  const bool have_partner = partner < nx_p && partner < MAXW;
  unsigned long long timeouts = 0, stale = 0, reads = 0, iters = 0;
  for (int r = 1; r <= ROUNDS; r++) {
    // consumer side first: warm L1 / L2 with the partner's OLD slot and flag
    double warm = 0.;
    if (have_partner) {
      for (int i = lane; i < SLOT; i += 64) warm += theirs[i];
      warm += (double)*(volatile unsigned *)pflag;
    }
    // producer: plain stores, wait for their acknowledgement, agent-scope add
    for (int i = lane; i < SLOT; i += 64) mine[i] = (double)(r * 1000 + (int)t) + 1e-3 * i + (warm == 12345.678 ? 1. : 0.);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) atomicAdd(myflag, 1u);
    if (!have_partner) continue;
    // consumer: poll (lane 0), then read
    int ok = 1;
    if (lane == 0) {
      int i = 0;
      while (poll_value(pflag, poll_mode) < (unsigned)r) {
        if (++i > 20000) { ok = 0; break; }
        __builtin_amdgcn_s_sleep(4);
      }
      iters += i;
    }
    ok = __builtin_amdgcn_readfirstlane(ok);
    if (!ok) { timeouts++; continue; }
    for (int i = lane; i < SLOT; i += 64) {
      const double v = read_mode ? __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : theirs[i];
      const double want = (double)(r * 1000 + (int)partner) + 1e-3 * i;
      reads++;
      if (v != want) stale++;
    }
  }
  atomicAdd(&st->stale, stale); atomicAdd(&st->reads, reads);
  if (lane == 0) { atomicAdd(&st->timeouts, timeouts); atomicAdd(&st->poll_iters, iters); if (!have_partner) atomicAdd(&st->no_partner, 1ull); }
}

int main() {
  const int grid = 512;
  double *data; unsigned *flag, *ticket, *nres; Stat *st;
  CHECK(hipMalloc(&data, sizeof(double) * 8 * MAXW * SLOT));
  CHECK(hipMalloc(&flag, sizeof(unsigned) * 8 * MAXW * 32));
  CHECK(hipMalloc(&ticket, sizeof(unsigned) * 8 * 32));
  CHECK(hipMalloc(&nres, sizeof(unsigned)));
  CHECK(hipMalloc(&st, sizeof(Stat)));
  const char *pn[] = {"sc1-load poll", "CAS poll", "plain poll"}, *rn[] = {"plain data loads", "sc1 data loads"};
  for (int cross = 0; cross < 2; cross++)
    for (int pm = 0; pm < 3; pm++)
      for (int rm = 0; rm < 2; rm++) {
        CHECK(hipMemset(data, 0, sizeof(double) * 8 * MAXW * SLOT));
        CHECK(hipMemset(flag, 0, sizeof(unsigned) * 8 * MAXW * 32));
        CHECK(hipMemset(ticket, 0, sizeof(unsigned) * 8 * 32));
        CHECK(hipMemset(nres, 0, sizeof(unsigned)));
        CHECK(hipMemset(st, 0, sizeof(Stat)));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_handoff, dim3(grid), dim3(64), 0, 0, data, flag, ticket, nres, st, pm, rm, cross, grid);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        Stat h; std::vector<unsigned> tk(8 * 32);
        CHECK(hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(tk.data(), ticket, sizeof(unsigned) * 8 * 32, hipMemcpyDeviceToHost));
        printf("%s partner | %-14s | %-16s | %7.2f ms | timeouts %llu of %d | stale words %llu of %llu | no partner %llu | avg poll iters %.1f | waves per XCD",
               cross ? "NEXT-XCD" : "same-XCD", pn[pm], rn[rm], ms, h.timeouts, grid * ROUNDS, h.stale, h.reads, h.no_partner,
               (double)h.poll_iters / (grid * ROUNDS));
        for (int x = 0; x < 8; x++) printf(" %u", tk[x * 32]);
        printf("\n");
        fflush(stdout);
      }
  return 0;
}
