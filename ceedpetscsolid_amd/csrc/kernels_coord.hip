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

// ===========================================================================
// opEnergy (setuplibceed.c:651-670, matops.c:247-296): u --GRAD(basisu)--> *Energy with qdata
// --INTERP^T(basisEnergy, 1 component)--> energy L-vector; its sum is the strain energy.  Post-processing,
// run once per solve: one workgroup per element, direct (not sum-factorised) tensor evaluation.
// ===========================================================================
namespace cps {

CPS_DEV double log1p_series4_e(double x) {  // hyperSS.h:43-55
  const double y = x / (2. + x), y2 = y * y;
  return 2. * (y + y2 * y / 3. + y2 * y2 * y / 5. + y2 * y2 * y2 * y / 7.);
}
CPS_DEV double log1p_series4_shifted_e(double x) {  // hyperFS.h:45-67
  const double left = sqrt(2.) / 2 - 1, right = sqrt(2.) - 1;
  double sum = 0;
  if (x < left) { sum -= log(2.) / 2; x = 1 + 2 * x; }
  else if (right < x) { sum += log(2.) / 2; x = (x - 1) / 2; }
  const double y = x / (2. + x), y2 = y * y;
  return sum + 2. * (y + y2 * y / 3. + y2 * y2 * y / 5. + y2 * y2 * y2 * y / 7.);
}
// model 0: LinElasEnergy (linElas.h:285-370), 1: HyperSSEnergy (hyperSS.h:326-412), 2: HyperFSEnergy
// (hyperFS.h:469-553), restated as written (including the `strain_vol * mu` term of the first two)
// returns the energy DENSITY; `dg` (optional) receives pressure, tr(strain), tr(strain^2), J of the diagnostic
// QFunctions (linElas.h:376-480, hyperSS.h:418-523, hyperFS.h:559-662)
CPS_DEV double qf_energy(int model, double nu, double E, const double *ug, const double *qd, double *dg) {
  const double TwoMu = E / (1 + nu), mu = TwoMu / 2, Kbulk = E / (3 * (1 - 2 * nu)), lambda = (3 * Kbulk - TwoMu) / 3;
  double g[3][3];  // grad u [component][derivative] = sum_m du[c][m] dXdx[m][k], ug[(d*3+c)] = du[c][d]
  for (int c = 0; c < 3; c++)
    for (int k = 0; k < 3; k++) {
      double s = 0;
      for (int m = 0; m < 3; m++) s += qd[1 + 3 * m + k] * ug[m * 3 + c];
      g[c][k] = s;
    }
  double en;
  if (model == 2) {
    const int J[6] = {0, 1, 2, 1, 0, 0}, K[6] = {0, 1, 2, 2, 2, 1};
    double w[6];
    for (int m = 0; m < 6; m++) {
      double s = g[J[m]][K[m]] + g[K[m]][J[m]];
      for (int n = 0; n < 3; n++) s += g[n][J[m]] * g[n][K[m]];
      w[m] = s;
    }
    const double detCm1 = w[0] * (w[1] * w[2] - w[3] * w[3]) + w[5] * (w[4] * w[3] - w[5] * w[2]) +
                          w[4] * (w[5] * w[3] - w[4] * w[1]) + w[0] + w[1] + w[2] + w[0] * w[1] + w[0] * w[2] +
                          w[1] * w[2] - w[5] * w[5] - w[4] * w[4] - w[3] * w[3];
    const double logj = log1p_series4_shifted_e(detCm1) / 2.;
    en = lambda * logj * logj / 2. - mu * logj + mu * (w[0] + w[1] + w[2]) / 2.;
    if (dg) {
      dg[0] = -lambda * logj;
      dg[1] = (w[0] + w[1] + w[2]) / 2.;
      dg[2] = (w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + 2. * (w[3] * w[3] + w[4] * w[4] + w[5] * w[5])) / 4.;
      dg[3] = sqrt(detCm1 + 1);
    }
  } else {
    const double e01 = (g[0][1] + g[1][0]) / 2., e02 = (g[0][2] + g[2][0]) / 2., e12 = (g[1][2] + g[2][1]) / 2.;
    const double sv = (g[0][0] + g[0][0]) / 2. + (g[1][1] + g[1][1]) / 2. + (g[2][2] + g[2][2]) / 2.;
    const double shear = (e01 * e01 + e02 * e02 + e12 * e12) * 2 * mu;
    const double llv = model == 1 ? log1p_series4_e(sv) : 0.;
    en = model == 0 ? lambda * sv * sv / 2. + sv * mu + shear : lambda * (1 + sv) * (llv - 1) + sv * mu + shear;
    if (dg) {
      const double e00 = (g[0][0] + g[0][0]) / 2., e11 = (g[1][1] + g[1][1]) / 2., e22 = (g[2][2] + g[2][2]) / 2.;
      dg[0] = model == 0 ? -lambda * sv : -lambda * llv;
      dg[1] = sv;
      dg[2] = e00 * e00 + e11 * e11 + e22 * e22 + 2. * (e01 * e01 + e02 * e02 + e12 * e12);
      dg[3] = 1 + sv;
    }
  }
  return en;
}

__global__ __launch_bounds__(512) void k_energy_op(const EnergyOpArgs a) {
  extern __shared__ double sh[];
  const int Q = a.Q, Q3 = Q * Q * Q, P = a.P, P3 = P * P * P;
  double *su = sh, *se = sh + 3 * P3;  // u[c][n]; energy[q]
  const int e = blockIdx.x, t = threadIdx.x;
  if (t < P3) {
    const uint32_t base = a.off_u[(size_t)e * P3 + t] & OFF_MASK;
    for (int c = 0; c < 3; c++) su[c * P3 + t] = a.u[base + c];
  }
  __syncthreads();
  if (t < Q3) {
    const int i = t % Q, j = (t / Q) % Q, k = t / (Q * Q);
    double ug[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.}, uv[3] = {0., 0., 0.};
    for (int cc = 0; cc < P; cc++)
      for (int b = 0; b < P; b++) {
        const double bk = a.interp[k * P + cc], gk = a.grad[k * P + cc], bj = a.interp[j * P + b], gj = a.grad[j * P + b];
        for (int aa = 0; aa < P; aa++) {
          const double bi = a.interp[i * P + aa], gi = a.grad[i * P + aa];
          const double w0 = gi * bj * bk, w1 = bi * gj * bk, w2 = bi * bj * gk;
          const int n = (cc * P + b) * P + aa;
          for (int c = 0; c < 3; c++) {
            const double v = su[c * P3 + n];
            ug[0 * 3 + c] += w0 * v; ug[1 * 3 + c] += w1 * v; ug[2 * 3 + c] += w2 * v;
            uv[c] += bi * bj * bk * v;
          }
        }
      }
    double qd[10];
    for (int c = 0; c < 10; c++) qd[c] = a.qdata[(size_t)e * 10 * Q3 + (size_t)c * Q3 + t];
    if (a.diag) {   // opDiagnostic: 8 fields, collocated with the points, summed over the elements sharing a node
      double dg[4];
      const double en = qf_energy(a.model, a.nu, a.E, ug, qd, dg);
      double *dst = a.y + (a.off_e[(size_t)e * Q3 + t] & OFF_MASK);
      for (int c = 0; c < 3; c++) atomic_add_f64(dst + c, uv[c]);
      for (int c = 0; c < 4; c++) atomic_add_f64(dst + 3 + c, dg[c]);
      atomic_add_f64(dst + 7, en);
    } else {
      se[t] = qf_energy(a.model, a.nu, a.E, ug, qd, nullptr) * qd[0];
    }
  }
  if (a.diag) return;
  __syncthreads();
  if (t < P3) {
    const int i = t % P, j = (t / P) % P, k = t / (P * P);
    double v = 0.;
    for (int qk = 0; qk < Q; qk++)
      for (int qj = 0; qj < Q; qj++) {
        const double wjk = a.interp_e[qk * P + k] * a.interp_e[qj * P + j];
        for (int qi = 0; qi < Q; qi++) v += wjk * a.interp_e[qi * P + i] * se[(qk * Q + qj) * Q + qi];
      }
    atomic_add_f64(a.y + (a.off_e[(size_t)e * P3 + t] & OFF_MASK), v);
  }
}

hipError_t launch_energy_op(const EnergyOpArgs &a, hipStream_t s) {
  if (a.nelem <= 0) return hipSuccess;
  const int Q3 = a.Q * a.Q * a.Q, P3 = a.P * a.P * a.P;
  int nt = Q3 > P3 ? Q3 : P3;
  nt = ((nt + 63) / 64) * 64;
  if (nt > 512 || a.Q > MAXN1D || a.P > MAXN1D) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_energy_op, dim3(a.nelem), dim3(nt), sizeof(double) * (3 * P3 + Q3), s, a);
  return hipGetLastError();
}

}  // namespace cps
