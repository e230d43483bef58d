// kernels_coord.hip -- set-up operators driven by the nodal coordinates (run once per solve, not on the
// hot path; written for clarity, one workgroup per element):
//   opSetupForce (setuplibceed.c:555-583): x --INTERP(basisx, 2 -> Q)--> SetupConstantForce / SetupMMSForce
//                 with qdata --INTERP^T(basisu)--> force vector
//   opTrue       (setuplibceed.c:608-623): x --INTERP(basisxtrue, 2 -> P, GLL)--> MMSTrueSoln --NONE--> nodes
//                 (summed over the elements sharing a node; the caller divides by the multiplicity, :626-636)
#include "kernels_common.hpp"

namespace cps {

// u = 1e-8 (e^{2x} sin 3y cos 4z, e^{3y} sin 4z cos 2x, e^{4z} sin 2x cos 3y)      (manufacturedTrue.h)
CPS_DEV void qf_mms_true(const double *p, double *u) {
  u[0] = exp(2 * p[0]) * sin(3 * p[1]) * cos(4 * p[2]) / 1e8;
  u[1] = exp(3 * p[1]) * sin(4 * p[2]) * cos(2 * p[0]) / 1e8;
  u[2] = exp(4 * p[2]) * sin(2 * p[0]) * cos(3 * p[1]) / 1e8;
}
// f = -div sigma(u_true) for the stress LinElasF applies (linElas.h:133-139: the Voigt shear factor on the
// TENSOR strain, i.e. sigma_ij = mu e_ij off the diagonal), times w detJ        (manufacturedForce.h:62-101)
CPS_DEV void qf_mms_force(double nu, double E, const double *p, double wdetJ, double *f) {
  const double mu = E / (2 * (1 + nu)), lambda = E * nu / ((1 + nu) * (1 - 2 * nu));
  const double k[3] = {2., 3., 4.};
  double g[3] = {0., 0., 0.};
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const int s = (c + 1) % 3, t = (c + 2) % 3;
    const double e = exp(k[c] * p[c]);
    const double ss = sin(k[s] * p[s]), cs = cos(k[s] * p[s]);
    const double st = sin(k[t] * p[t]), ct = cos(k[t] * p[t]);
    const double u = e * ss * ct;
    g[c] += (lambda + 2 * mu) * k[c] * k[c] * u - (mu / 2) * (k[s] * k[s] + k[t] * k[t]) * u;
    g[s] += (lambda + mu / 2) * k[c] * k[s] * e * cs * ct;
    g[t] -= (lambda + mu / 2) * k[c] * k[t] * e * ss * st;
  }
#pragma unroll
  for (int c = 0; c < 3; c++) f[c] = -g[c] * wdetJ / 1e8;
}

// mode 0: SetupConstantForce (ctx = direction), 1: SetupMMSForce (ctx = nu, E), 2: MMSTrueSoln
__global__ __launch_bounds__(512) void k_coord_op(const CoordOpArgs a) {
  extern __shared__ double sh[];
  const int Q = a.Q, Q3 = Q * Q * Q, P = a.Pout, P3 = P * P * P;
  double *sx = sh, *sf = sh + 24;     // 8 coordinate nodes x 3; f[c][q]
  const int e = blockIdx.x, t = threadIdx.x;
  if (t < 8) {
    const uint32_t base = a.off_x[(size_t)e * 8 + t] & OFF_MASK;
    for (int c = 0; c < 3; c++) sx[c * 8 + t] = a.xcoord[base + c];
  }
  __syncthreads();
  if (t < Q3) {
    const int i = t % Q, j = (t / Q) % Q, k = t / (Q * Q);
    double p[3] = {0., 0., 0.};
    for (int cc = 0; cc < 2; cc++)
      for (int b = 0; b < 2; b++)
        for (int aa = 0; aa < 2; aa++) {
          const double w = a.bx[i * 2 + aa] * a.bx[j * 2 + b] * a.bx[k * 2 + cc];
          for (int c = 0; c < 3; c++) p[c] += w * sx[c * 8 + aa + 2 * b + 4 * cc];
        }
    double f[3];
    if (a.mode == 2) qf_mms_true(p, f);
    else {
      const double wdetJ = a.qdata[(size_t)e * 10 * Q3 + t];
      if (a.mode == 1) qf_mms_force(a.ctx[0], a.ctx[1], p, wdetJ, f);
      else for (int c = 0; c < 3; c++) f[c] = a.ctx[c] * wdetJ;
    }
    for (int c = 0; c < 3; c++) sf[c * Q3 + t] = f[c];
  }
  __syncthreads();
  if (t < P3) {
    const uint32_t off = a.off_u[(size_t)e * P3 + t], base = off & OFF_MASK;
    double v[3] = {0., 0., 0.};
    if (a.mode == 2) {  // collocated output (Q == Pout)
      for (int c = 0; c < 3; c++) v[c] = sf[c * Q3 + t];
    } else {            // INTERP^T with the displacement basis B[q][p]
      const int i = t % P, j = (t / P) % P, k = t / (P * P);
      for (int qk = 0; qk < Q; qk++)
        for (int qj = 0; qj < Q; qj++) {
          const double wjk = a.bu[qk * P + k] * a.bu[qj * P + j];
          for (int qi = 0; qi < Q; qi++) {
            const double w = wjk * a.bu[qi * P + i];
            const int q = (qk * Q + qj) * Q + qi;
            for (int c = 0; c < 3; c++) v[c] += w * sf[c * Q3 + q];
          }
        }
    }
    for (int c = 0; c < 3; c++) atomic_add_f64(a.y + base + c, v[c]);
  }
}

hipError_t launch_coord_op(const CoordOpArgs &a, hipStream_t s) {
  if (a.nelem <= 0) return hipSuccess;
  const int Q3 = a.Q * a.Q * a.Q, P3 = a.Pout * a.Pout * a.Pout;
  int nt = Q3 > P3 ? Q3 : P3;
  nt = ((nt + 63) / 64) * 64;
  if (nt > 512 || a.Q > MAXN1D || a.Pout > MAXN1D) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_coord_op, dim3(a.nelem), dim3(nt), sizeof(double) * (24 + 3 * Q3), s, a);
  return hipGetLastError();
}

}  // namespace cps
