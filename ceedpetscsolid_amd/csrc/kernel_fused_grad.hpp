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

// All LDS traffic of a wave is in-order; in a single-wave workgroup this only keeps
// the compiler from moving LDS accesses across a phase boundary.
CPS_DEV void wave_sync() { __syncthreads(); }

template <int P, int Q, int QF>
__global__ __launch_bounds__(64) void k_fused_grad(const BasisTables tab, const FusedGradArgs a) {
  using G = WaveGeom<Q>;
  constexpr int Q3 = G::Q3, P3 = P * P * P, TPE = G::TPE, EPW = G::EPW, SLOTS = G::SLOTS;
  constexpr bool ST_IN = QFTraits<QF>::state_in, ST_OUT = QFTraits<QF>::state_out;
  static_assert(P <= Q, "interpolation to at least as many points as nodes");

  __shared__ double sB[Q * P];
  __shared__ double sD[Q * Q];
  __shared__ double slab[EPW][G::SLAB];

  const int lane = threadIdx.x;
  const int el = EPW > 1 ? lane / TPE : 0;
  const int q0 = EPW > 1 ? lane % TPE : lane;  // slot s handles point q0 + 64 s
  const int e = blockIdx.x * EPW + el;
  const bool live = e < a.nelem;
  // Element slab: RA, RB, RC = 3*Q3 work regions; RG = 9*Q3 region for the physics output,
  // spanning RB, RC and a further 3*Q3 (it is written only once RB / RC are dead, and is
  // disjoint from RA, which holds the interpolated field while the physics runs).
  double *RA = slab[el], *RB = RA + 3 * Q3, *RC = RA + 6 * Q3, *RG = RA + 3 * Q3;

  double qd[10], st[9];
  auto load_point = [&](int q) {
    if (live && q < Q3) {
      const double *qp = a.qdata + (size_t)e * 10 * Q3 + q;
#pragma unroll
      for (int c = 0; c < 10; c++) qd[c] = qp[c * Q3];
      if constexpr (ST_IN) {
        const double *sp = a.state_in + (size_t)e * 9 * Q3 + q;
#pragma unroll
        for (int c = 0; c < 9; c++) st[c] = sp[c * Q3];
      }
    }
  };

  // ---- gather ---------------------------------------------------------------------------
  uint32_t off[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int n = q0 + 64 * s;
    off[s] = 0;
    if (n < P3) {
      double xin[3] = {0., 0., 0.};
      if (live) {
        off[s] = a.offsets[(size_t)e * P3 + n];
        const uint32_t base = off[s] & OFF_MASK;
        const uint32_t fl = a.mask_in ? (off[s] >> OFF_FLAG_SHIFT) : 0u;
#pragma unroll
        for (int c = 0; c < 3; c++) xin[c] = ((fl >> c) & 1u) ? 0. : a.x[base + c];
      }
#pragma unroll
      for (int c = 0; c < 3; c++) RA[c * P3 + n] = xin[c];
    }
  }
  load_point(q0);  // slot 0's q-point data: in flight across the interpolation
  for (int i = lane; i < Q * P; i += 64) sB[i] = tab.interp[i];
  for (int i = lane; i < Q * Q; i += 64) sD[i] = tab.colo[i];
  wave_sync();

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
        if (!((fl >> c) & 1u)) atomic_add_f64(a.y + base + c, t);
      }
    }
  }
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
