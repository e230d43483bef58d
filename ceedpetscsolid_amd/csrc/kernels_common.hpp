// kernels_common.hpp -- device building blocks shared by the gfx950 kernels.
//
// `Geom<Q>` is the point-per-lane mapping of the SET-UP and TRANSFER kernels (kernels_misc.hip:
// k_setup_geo, k_transfer, k_diag): an element owns TPE lanes (Q^3 rounded up to a whole number of
// waves, or to a power of two when several elements share a wave) and a workgroup owns EPB elements;
// their 1-D contractions (interp_forward / interp_transpose below) go through an LDS element slab with
// workgroup barriers.  The operator-apply kernels (kernel_fused_pencil.hpp, kernel_fused_grad.hpp) have
// their own barrier-free wave-level mappings.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.hpp"

namespace cps {

#define CPS_DEV static __device__ __forceinline__

constexpr int cpow3(int n) { return n * n * n; }
constexpr int next_pow2(int n) { int p = 1; while (p < n) p *= 2; return p; }

template <int Q> struct Geom {
  static constexpr int Q3 = cpow3(Q);
  static constexpr int TPE = Q3 <= 32 ? next_pow2(Q3) : ((Q3 + 63) / 64) * 64;
#ifndef CPS_BLOCK_TARGET
#define CPS_BLOCK_TARGET 256
#endif
  static constexpr int EPB = TPE >= CPS_BLOCK_TARGET ? 1 : CPS_BLOCK_TARGET / TPE;
  static constexpr int BLOCK = TPE * EPB;
};

CPS_DEV void atomic_add_f64(double *p, double v) {
  // hardware f64 atomic (global_atomic_add_f64); no CAS loop
  unsafeAtomicAdd(p, v);
}

// --- forward interpolation P^3 nodes -> Q^3 points, three components --------
// in : R0[c][P^3] (x fastest)          out: val[c] at this lane's point q (if q < Q^3)
// scratch R1, R2 (each >= 3*Q^3).  sB = interp table B[q][p] in LDS.
// Every lane of the workgroup must call (contains barriers).
template <int P, int Q>
CPS_DEV void interp_forward(int q, const double *R0, double *R1, double *R2, const double *sB,
                            double val[3]) {
  constexpr int Q3 = Q * Q * Q;
  // x: [k][j][i] -> [k][j][i']
  if (q < P * P * Q) {
    const int i = q % Q, kj = q / Q;
    double b[P];
#pragma unroll
    for (int m = 0; m < P; m++) b[m] = sB[i * P + m];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double *u = R0 + c * (P * P * P) + kj * P;
      double s = 0.;
#pragma unroll
      for (int m = 0; m < P; m++) s += b[m] * u[m];
      R1[c * Q3 + kj * Q + i] = s;
    }
  }
  __syncthreads();
  // y: [k][j][i'] -> [k][j'][i']
  if (q < P * Q * Q) {
    const int i = q % Q, j = (q / Q) % Q, k = q / (Q * Q);
    double b[P];
#pragma unroll
    for (int m = 0; m < P; m++) b[m] = sB[j * P + m];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double *u = R1 + c * Q3 + k * (P * Q) + i;
      double s = 0.;
#pragma unroll
      for (int m = 0; m < P; m++) s += b[m] * u[m * Q];
      R2[c * Q3 + (k * Q + j) * Q + i] = s;
    }
  }
  __syncthreads();
  // z: [k][j'][i'] -> [k'][j'][i']
  if (q < Q3) {
    const int ji = q % (Q * Q), k = q / (Q * Q);
    double b[P];
#pragma unroll
    for (int m = 0; m < P; m++) b[m] = sB[k * P + m];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double *u = R2 + c * Q3 + ji;
      double s = 0.;
#pragma unroll
      for (int m = 0; m < P; m++) s += b[m] * u[m * Q * Q];
      val[c] = s;
    }
  }
}

// --- transpose of the above: Q^3 point values -> P^3 node values -------------
// in : R0[c][Q^3]        out: val[c] for node n = q (if q < P^3)
template <int P, int Q>
CPS_DEV void interp_transpose(int q, const double *R0, double *R1, double *R2, const double *sB,
                              double val[3]) {
  constexpr int Q3 = Q * Q * Q;
  // z^T: [k'][j'][i'] -> [k][j'][i']
  if (q < P * Q * Q) {
    const int ji = q % (Q * Q), k = q / (Q * Q);
    double b[Q];
#pragma unroll
    for (int m = 0; m < Q; m++) b[m] = sB[m * P + k];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double *u = R0 + c * Q3 + ji;
      double s = 0.;
#pragma unroll
      for (int m = 0; m < Q; m++) s += b[m] * u[m * Q * Q];
      R1[c * Q3 + k * Q * Q + ji] = s;
    }
  }
  __syncthreads();
  // y^T: [k][j'][i'] -> [k][j][i']
  if (q < P * P * Q) {
    const int i = q % Q, j = (q / Q) % P, k = q / (Q * P);
    double b[Q];
#pragma unroll
    for (int m = 0; m < Q; m++) b[m] = sB[m * P + j];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double *u = R1 + c * Q3 + k * Q * Q + i;
      double s = 0.;
#pragma unroll
      for (int m = 0; m < Q; m++) s += b[m] * u[m * Q];
      R2[c * Q3 + (k * P + j) * Q + i] = s;
    }
  }
  __syncthreads();
  // x^T: [k][j][i'] -> [k][j][i]
  if (q < P * P * P) {
    const int i = q % P, kj = q / P;
    double b[Q];
#pragma unroll
    for (int m = 0; m < Q; m++) b[m] = sB[m * P + i];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double *u = R2 + c * Q3 + kj * Q;
      double s = 0.;
#pragma unroll
      for (int m = 0; m < Q; m++) s += b[m] * u[m];
      val[c] = s;
    }
  }
}

// copy a small table from the kernarg segment into LDS
template <int N, int BLOCK>
CPS_DEV void stage_table(const double *src, double *dst) {
  for (int i = threadIdx.x; i < N; i += BLOCK) dst[i] = src[i];
}

}  // namespace cps
