// kernel_fused_grad.hpp -- the whole CeedOperatorApply of the residual / Jacobian
// operators (setuplibceed.c:517-542, :817-839) in ONE launch:
//   E-vector gather (offsets, Dirichlet flags) -> sum-factorised interpolation to
//   the Gauss points -> collocated gradient -> pointwise physics in registers
//   (q-point data streamed once, coalesced, prefetched one point-slot ahead) ->
//   collocated gradient^T -> interpolation^T -> f64 atomic scatter-add.
// Nothing but the L-vectors and the q-point data touches HBM: no E-vector, no
// quadrature-point intermediate is ever written out.
//
// Wave-per-element, barrier-free (v2).  A workgroup is ONE wave64 and owns its
// element(s) outright, so every LDS write -> read dependency stays inside one
// wave, where the LDS queue is in-order: no cross-wave barrier is ever needed and
// the waves of a CU never wait for each other (v1, two waves per element around
// ten workgroup barriers, spent 54 % of its wave-cycles in SQ_WAIT_ANY:
// profiles/r01_pmc_v1.txt).  Each lane owns SLOTS quadrature points of the element
// (q = lane + 64 s): Q=5 -> 125 points on 2 x 64 slots (97.7 % of lane-slots
// busy), Q=4 -> one point per lane, Q=7 -> 6 slots; small elements share a wave
// (Q=3: two elements, Q=2: eight).
#pragma once
#include "kernels_common.hpp"
#include "qfunctions_device.hpp"

namespace cps {

template <int Q> struct WaveGeom {
  static constexpr int Q3 = Q * Q * Q;
  static constexpr int TPE = Q3 <= 32 ? next_pow2(Q3) : ((Q3 + 63) / 64) * 64;  // point slots per element
  static constexpr int EPW = TPE >= 64 ? 1 : 64 / TPE;                           // elements per wave
  static constexpr int SLOTS = TPE >= 64 ? TPE / 64 : 1;                         // point slots per lane
  static constexpr int SLAB = 12 * Q3;                                           // doubles of LDS per element
};

// Phase boundary inside ONE wave.  The hardware executes a wave's LDS instructions in
// order, so a ds_write followed by another lane's ds_read of that address needs no wait
// and no s_barrier -- only the COMPILER must not move LDS accesses across the boundary.
// A wavefront-scope fence does exactly that and emits no instruction; in particular it
// does not drain vmcnt, so the q-point loads issued at kernel entry stay in flight across
// the whole interpolation (a __syncthreads() here cost 30 % of the wave's lifetime in
// `s_waitcnt vmcnt(0)`: tools/stamp_profile.py, round 1).
CPS_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int P, int Q, int QF>
__global__ __launch_bounds__(64) void k_fused_grad(const BasisTables tab, const FusedGradArgs a) {
  using G = WaveGeom<Q>;
  constexpr int Q3 = G::Q3, P3 = P * P * P, TPE = G::TPE, EPW = G::EPW, SLOTS = G::SLOTS;
  constexpr bool ST_IN = QFTraits<QF>::state_in, ST_OUT = QFTraits<QF>::state_out;
  static_assert(P <= Q, "interpolation to at least as many points as nodes");

  __shared__ double sB[Q * P];
  __shared__ double sD[Q * Q];
  __shared__ double slab[EPW][G::SLAB];

#ifdef CPS_STAMPS  // diagnostic build: where does a wave's lifetime go?  (never in the product build)
  unsigned long long stamp_[8]; int nst_ = 0;
#define CPS_STAMP() do { __builtin_amdgcn_sched_barrier(0); stamp_[nst_++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CPS_STAMP() do {} while (0)
#endif
  CPS_STAMP();
  const int lane = threadIdx.x;
  const int el = EPW > 1 ? lane / TPE : 0;
  const int q0 = EPW > 1 ? lane % TPE : lane;  // slot s handles point q0 + 64 s
  const int e = blockIdx.x * EPW + el;
  const bool live = e < a.nelem;
  // Element slab: RA, RB, RC = 3*Q3 work regions; RG = 9*Q3 region for the physics output,
  // spanning RB, RC and a further 3*Q3 (it is written only once RB / RC are dead, and is
  // disjoint from RA, which holds the interpolated field while the physics runs).
  double *RA = slab[el], *RB = RA + 3 * Q3, *RC = RA + 6 * Q3, *RG = RA + 3 * Q3;

  // All global loads are unconditional and in-bounds (indices clamped, results selected
  // afterwards): no exec-mask branches around loads, so a wave's loads issue back to back.
  const int ec = live ? e : a.nelem - 1;
  double qd[10], st[9];
  auto load_point = [&](int q) {
#ifdef CPS_ABLATE_QDATA  // timing-only build: no q-point stream (WRONG results)
    for (int c = 0; c < 10; c++) qd[c] = (c == 1 || c == 5 || c == 9 || c == 0) ? 1.0 + 1e-3 * q : 1e-3 * c;
    for (int c = 0; c < 9; c++) st[c] = 1e-3 * (c + lane);
    return;
#endif
    const int qc = q < Q3 ? q : Q3 - 1;
    const double *qp = a.qdata + (size_t)ec * 10 * Q3 + qc;
#pragma unroll
    for (int c = 0; c < 10; c++) qd[c] = qp[c * Q3];
    if constexpr (ST_IN) {
      const double *sp = a.state_in + (size_t)ec * 9 * Q3 + qc;
#pragma unroll
      for (int c = 0; c < 9; c++) st[c] = sp[c * Q3];
    }
  };

  // ---- gather: offsets of every slot first, then all x loads, then slot 0's q-point data ---
  // (vmcnt retires in order: the q-point loads go LAST so that waiting for x leaves them in
  // flight across the interpolation).
  uint32_t off[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int n = q0 + 64 * s;
    off[s] = a.offsets[(size_t)ec * P3 + (n < P3 ? n : P3 - 1)];
  }
  double xin[SLOTS][3];
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const uint32_t base = off[s] & OFF_MASK;
#pragma unroll
    for (int c = 0; c < 3; c++) xin[s][c] = a.x[base + c];
  }
  load_point(q0);  // slot 0's q-point data: in flight across the interpolation
  for (int i = lane; i < Q * P; i += 64) sB[i] = tab.interp[i];
  for (int i = lane; i < Q * Q; i += 64) sD[i] = tab.colo[i];
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int n = q0 + 64 * s;
    if (n < P3) {
      const uint32_t fl = (a.mask_in && live) ? (off[s] >> OFF_FLAG_SHIFT) : (live ? 0u : 7u);
#pragma unroll
      for (int c = 0; c < 3; c++) RA[c * P3 + n] = ((fl >> c) & 1u) ? 0. : xin[s][c];
    }
  }
  wave_sync();
  CPS_STAMP();  // 1: gather landed in LDS

  // ---- B: nodes -> points (x, y, z passes) -----------------------------------------------
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {  // x: RA[c][k][j][i] -> RB[c][k][j][i']
    const int q = q0 + 64 * s;
    if (q < P * P * Q) {
      const int i = q % Q, kj = q / Q;
      double b[P];
#pragma unroll
      for (int m = 0; m < P; m++) b[m] = sB[i * P + m];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double *u = RA + c * P3 + kj * P;
        double t = 0.;
#pragma unroll
        for (int m = 0; m < P; m++) t += b[m] * u[m];
        RB[c * Q3 + kj * Q + i] = t;
      }
    }
  }
  wave_sync();
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {  // y: RB[c][k][j][i'] -> RC[c][k][j'][i']
    const int q = q0 + 64 * s;
    if (q < P * Q * Q) {
      const int i = q % Q, j = (q / Q) % Q, k = q / (Q * Q);
      double b[P];
#pragma unroll
      for (int m = 0; m < P; m++) b[m] = sB[j * P + m];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double *u = RB + c * Q3 + k * (P * Q) + i;
        double t = 0.;
#pragma unroll
        for (int m = 0; m < P; m++) t += b[m] * u[m * Q];
        RC[c * Q3 + (k * Q + j) * Q + i] = t;
      }
    }
  }
  wave_sync();
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {  // z: RC[c][k][j'][i'] -> RA[c][k'][j'][i']
    const int q = q0 + 64 * s;
    if (q < Q3) {
      const int ji = q % (Q * Q), k = q / (Q * Q);
      double b[P];
#pragma unroll
      for (int m = 0; m < P; m++) b[m] = sB[k * P + m];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double *u = RC + c * Q3 + ji;
        double t = 0.;
#pragma unroll
        for (int m = 0; m < P; m++) t += b[m] * u[m * Q * Q];
        RA[c * Q3 + q] = t;
      }
    }
  }
  wave_sync();
  CPS_STAMP();  // 2: interpolated

  // ---- collocated gradient + physics, one point slot at a time ---------------------------
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = q0 + 64 * s;
    const int qi = q % Q, qj = (q / Q) % Q, qk = q / (Q * Q);
    double ug[9], dv[9], sto[9];
    if (q < Q3) {
      double d0[Q], d1[Q], d2[Q];
#pragma unroll
      for (int m = 0; m < Q; m++) { d0[m] = sD[qi * Q + m]; d1[m] = sD[qj * Q + m]; d2[m] = sD[qk * Q + m]; }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double *w = RA + c * Q3;
        double s0 = 0., s1 = 0., s2 = 0.;
#pragma unroll
        for (int m = 0; m < Q; m++) {
          s0 += d0[m] * w[(qk * Q + qj) * Q + m];
          s1 += d1[m] * w[(qk * Q + m) * Q + qi];
          s2 += d2[m] * w[(m * Q + qj) * Q + qi];
        }
        ug[0 * 3 + c] = s0; ug[1 * 3 + c] = s1; ug[2 * 3 + c] = s2;
      }
    }
    if (live && q < Q3) {
      qf_point<QF>(Phys{a.nu, a.E, a.lambda, a.TwoMu}, ug, qd, st, dv, sto);
      if constexpr (ST_OUT) {
        double *sp = a.state_out + (size_t)e * 9 * Q3 + q;
#pragma unroll
        for (int c = 0; c < 9; c++) sp[c * Q3] = sto[c];
      }
    } else {
#pragma unroll
      for (int c = 0; c < 9; c++) dv[c] = 0.;
    }
    if (s + 1 < SLOTS) load_point(q + 64);  // next slot's q-point data
    if (q < Q3) {
#pragma unroll
      for (int c = 0; c < 9; c++) RG[c * Q3 + q] = dv[c];
    }
  }
  wave_sync();
  CPS_STAMP();  // 3: physics done

  // ---- collocated gradient^T: RG[d][c] -> RA[c]  (RA is dead: every slot has read it) ------
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = q0 + 64 * s;
    if (q < Q3) {
      const int qi = q % Q, qj = (q / Q) % Q, qk = q / (Q * Q);
      double d0[Q], d1[Q], d2[Q];
#pragma unroll
      for (int m = 0; m < Q; m++) { d0[m] = sD[m * Q + qi]; d1[m] = sD[m * Q + qj]; d2[m] = sD[m * Q + qk]; }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double *g0 = RG + (0 * 3 + c) * Q3, *g1 = RG + (1 * 3 + c) * Q3, *g2 = RG + (2 * 3 + c) * Q3;
        double t = 0.;
#pragma unroll
        for (int m = 0; m < Q; m++) {
          t += d0[m] * g0[(qk * Q + qj) * Q + m];
          t += d1[m] * g1[(qk * Q + m) * Q + qi];
          t += d2[m] * g2[(m * Q + qj) * Q + qi];
        }
        RA[c * Q3 + q] = t;
      }
    }
  }
  wave_sync();
  CPS_STAMP();  // 4: gradient^T done

  // ---- B^T: points -> nodes (z^T, y^T, x^T), then scatter-add -------------------------------
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {  // z^T: RA[c][k'][j'][i'] -> RB[c][k][j'][i']
    const int q = q0 + 64 * s;
    if (q < P * Q * Q) {
      const int ji = q % (Q * Q), k = q / (Q * Q);
      double b[Q];
#pragma unroll
      for (int m = 0; m < Q; m++) b[m] = sB[m * P + k];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double *u = RA + c * Q3 + ji;
        double t = 0.;
#pragma unroll
        for (int m = 0; m < Q; m++) t += b[m] * u[m * Q * Q];
        RB[c * Q3 + k * Q * Q + ji] = t;
      }
    }
  }
  wave_sync();
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {  // y^T: RB[c][k][j'][i'] -> RC[c][k][j][i']
    const int q = q0 + 64 * s;
    if (q < P * P * Q) {
      const int i = q % Q, j = (q / Q) % P, k = q / (Q * P);
      double b[Q];
#pragma unroll
      for (int m = 0; m < Q; m++) b[m] = sB[m * P + j];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double *u = RB + c * Q3 + k * Q * Q + i;
        double t = 0.;
#pragma unroll
        for (int m = 0; m < Q; m++) t += b[m] * u[m * Q];
        RC[c * Q3 + (k * P + j) * Q + i] = t;
      }
    }
  }
  wave_sync();
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {  // x^T + scatter
    const int n = q0 + 64 * s;
    if (live && n < P3) {
      const int i = n % P, kj = n / P;
      double b[Q];
#pragma unroll
      for (int m = 0; m < Q; m++) b[m] = sB[m * P + i];
      const uint32_t base = off[s] & OFF_MASK;
      const uint32_t fl = a.mask_out ? (off[s] >> OFF_FLAG_SHIFT) : 0u;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double *u = RC + c * Q3 + kj * Q;
        double t = 0.;
#pragma unroll
        for (int m = 0; m < Q; m++) t += b[m] * u[m];
#if defined(CPS_ABLATE_ATOMICS)   // timing-only build: plain store instead of the atomic (WRONG results)
        if (!((fl >> c) & 1u)) a.y[base + c] = t;
#elif defined(CPS_ABLATE_SCATTER)  // timing-only build: no scatter at all
        asm volatile("" ::"v"(t));
#else
        if (!((fl >> c) & 1u)) atomic_add_f64(a.y + base + c, t);
#endif
      }
    }
  }
  CPS_STAMP();  // 5: atomics issued
#ifdef CPS_STAMPS
  if (a.stamps && lane == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    stamp_[nst_++] = __builtin_amdgcn_s_memtime();  // 6: all memory ops retired
    for (int i = 0; i < nst_; i++) a.stamps[(size_t)blockIdx.x * 8 + i] = stamp_[i];
  }
#endif
}

template <int P, int Q, int QF>
hipError_t launch_fused_grad_t(const BasisTables &t, const FusedGradArgs &a, hipStream_t s) {
  using G = WaveGeom<Q>;
  if (a.nelem <= 0) return hipSuccess;
  const int grid = (a.nelem + G::EPW - 1) / G::EPW;
  hipLaunchKernelGGL((k_fused_grad<P, Q, QF>), dim3(grid), dim3(64), 0, s, t, a);
  return hipGetLastError();
}

}  // namespace cps
