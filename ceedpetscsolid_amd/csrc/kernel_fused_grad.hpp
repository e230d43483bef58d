// kernel_fused_grad.hpp -- the whole CeedOperatorApply of the residual / Jacobian
// operators (setuplibceed.c:517-542, :817-839) in ONE launch:
//   E-vector gather (offsets, Dirichlet flags) -> sum-factorised interpolation to
//   the Gauss points -> collocated gradient -> pointwise physics in registers
//   (q-point data streamed once, coalesced, issued before the basis work so HBM
//   latency hides under it) -> collocated gradient^T -> interpolation^T ->
//   f64 atomic scatter-add into the L-vector.
// Nothing but the L-vectors and the q-point data touches HBM: no E-vector, no
// quadrature-point intermediate is ever written out.
#pragma once
#include "kernels_common.hpp"
#include "qfunctions_device.hpp"

namespace cps {

template <int P, int Q, int QF>
__global__ __launch_bounds__(Geom<Q>::BLOCK) void k_fused_grad(const BasisTables tab,
                                                                const FusedGradArgs a) {
  using G = Geom<Q>;
  constexpr int Q3 = G::Q3, P3 = P * P * P, TPE = G::TPE, EPB = G::EPB, BLOCK = G::BLOCK;
  constexpr bool ST_IN = QFTraits<QF>::state_in, ST_OUT = QFTraits<QF>::state_out;
  static_assert(P <= Q, "interpolation to at least as many points as nodes");

  __shared__ double sB[Q * P];
  __shared__ double sD[Q * Q];
  __shared__ double slab[EPB][9 * Q3];

  const int tid = threadIdx.x;
  const int el = tid / TPE, q = tid % TPE;
  const int e = blockIdx.x * EPB + el;
  const bool live = e < a.nelem;
  double *R0 = slab[el], *R1 = R0 + 3 * Q3, *R2 = R0 + 6 * Q3;

  // ---- gather (issued first: its results are needed first) -----------------
  uint32_t off = 0;
  double xin[3] = {0., 0., 0.};
  const bool node = live && q < P3;
  if (node) {
    off = a.offsets[(size_t)e * P3 + q];
    const uint32_t base = off & OFF_MASK;
    const uint32_t fl = a.mask_in ? (off >> OFF_FLAG_SHIFT) : 0u;
#pragma unroll
    for (int c = 0; c < 3; c++) xin[c] = ((fl >> c) & 1u) ? 0. : a.x[base + c];
  }
  // ---- q-point data prefetch: stays in flight across the basis phase --------
  double qd[10], st[9];
  const bool pt = live && q < Q3;
  if (pt) {
    const double *qp = a.qdata + (size_t)e * 10 * Q3 + q;
#pragma unroll
    for (int c = 0; c < 10; c++) qd[c] = qp[c * Q3];
    if constexpr (ST_IN) {
      const double *sp = a.state_in + (size_t)e * 9 * Q3 + q;
#pragma unroll
      for (int c = 0; c < 9; c++) st[c] = sp[c * Q3];
    }
  }
  stage_table<Q * P, BLOCK>(tab.interp, sB);
  stage_table<Q * Q, BLOCK>(tab.colo, sD);
  if (q < P3) {
#pragma unroll
    for (int c = 0; c < 3; c++) R0[c * P3 + q] = xin[c];
  }
  __syncthreads();

  // ---- B: nodes -> points ---------------------------------------------------
  double u[3];
  interp_forward<P, Q>(q, R0, R1, R2, sB, u);
  if (q < Q3) {
#pragma unroll
    for (int c = 0; c < 3; c++) R0[c * Q3 + q] = u[c];
  }
  __syncthreads();

  // ---- collocated gradient + physics ---------------------------------------
  const int qi = q % Q, qj = (q / Q) % Q, qk = q / (Q * Q);
  double dv[9];
  if (q < Q3) {
    double ug[9];
    {
      double d0[Q], d1[Q], d2[Q];
#pragma unroll
      for (int m = 0; m < Q; m++) { d0[m] = sD[qi * Q + m]; d1[m] = sD[qj * Q + m]; d2[m] = sD[qk * Q + m]; }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double *w = R0 + c * Q3;
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
    double sto[9];
    if (pt) {
      qf_point<QF>(Phys{a.nu, a.E}, ug, qd, st, dv, sto);
      if constexpr (ST_OUT) {
        double *sp = a.state_out + (size_t)e * 9 * Q3 + q;
#pragma unroll
        for (int c = 0; c < 9; c++) sp[c * Q3] = sto[c];
      }
    } else {
#pragma unroll
      for (int c = 0; c < 9; c++) dv[c] = 0.;
    }
  }
  __syncthreads();
  if (q < Q3) {
#pragma unroll
    for (int c = 0; c < 9; c++) R0[c * Q3 + q] = dv[c];
  }
  __syncthreads();

  // ---- collocated gradient^T -------------------------------------------------
  double w3[3];
  if (q < Q3) {
    double d0[Q], d1[Q], d2[Q];
#pragma unroll
    for (int m = 0; m < Q; m++) { d0[m] = sD[m * Q + qi]; d1[m] = sD[m * Q + qj]; d2[m] = sD[m * Q + qk]; }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double *g0 = R0 + (0 * 3 + c) * Q3, *g1 = R0 + (1 * 3 + c) * Q3, *g2 = R0 + (2 * 3 + c) * Q3;
      double s = 0.;
#pragma unroll
      for (int m = 0; m < Q; m++) {
        s += d0[m] * g0[(qk * Q + qj) * Q + m];
        s += d1[m] * g1[(qk * Q + m) * Q + qi];
        s += d2[m] * g2[(m * Q + qj) * Q + qi];
      }
      w3[c] = s;
    }
  }
  __syncthreads();
  if (q < Q3) {
#pragma unroll
    for (int c = 0; c < 3; c++) R0[c * Q3 + q] = w3[c];
  }
  __syncthreads();

  // ---- B^T: points -> nodes, then scatter-add --------------------------------
  double v[3];
  interp_transpose<P, Q>(q, R0, R1, R2, sB, v);
  if (node) {
    const uint32_t base = off & OFF_MASK;
    const uint32_t fl = a.mask_out ? (off >> OFF_FLAG_SHIFT) : 0u;
#pragma unroll
    for (int c = 0; c < 3; c++)
      if (!((fl >> c) & 1u)) atomic_add_f64(a.y + base + c, v[c]);
  }
}

template <int P, int Q, int QF>
hipError_t launch_fused_grad_t(const BasisTables &t, const FusedGradArgs &a, hipStream_t s) {
  using G = Geom<Q>;
  if (a.nelem <= 0) return hipSuccess;
  const int grid = (a.nelem + G::EPB - 1) / G::EPB;
  hipLaunchKernelGGL((k_fused_grad<P, Q, QF>), dim3(grid), dim3(G::BLOCK), 0, s, t, a);
  return hipGetLastError();
}

}  // namespace cps
