// qfunctions_device.hpp -- pointwise physics as gfx950 device functors.
//
// Device restatement of the reference's QFunctions (qfunctions/common.h,
// linElas.h, hyperSS.h, hyperFS.h).  Upstream GPU backends JIT the source named
// by the "file:Name" locator (setuplibceed.c:49-53); here the name after ':' is
// mapped to one of these precompiled functors (see ceed_api.cpp: resolve_qf).
//
// Register conventions inside the fused kernels (one quadrature point per lane):
//   ug[d*3 + c] = d u_c / d xi_d          (GRAD input,  linElas.h:62-71)
//   qd[0] = w detJ, qd[1 + 3r + s] = dXdx[r][s]      (common.h:84-96)
//   st[3c + k] = d u_c / d x_k            (stored state, hyperFS.h:215-220)
//   dv[k*3 + c]                            (GRAD output, linElas.h:148-153)
// The log1p series are the reference's own (hyperSS.h:43-55, hyperFS.h:45-67):
// libm's log1p differs by up to 3e-8 and would break the 1e-10 parity bar.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace cps {

// {nu, E} as in elasticity.h:33-36, plus the Lame constants derived from them ON THE HOST with
// the reference's formulas (hyperSS.h:79-81; IEEE division is correctly rounded on both sides,
// so the values are bit-identical) to keep f64 divisions out of the per-point code.
struct Phys { double nu, E, lambda, TwoMu; };

#define CPS_DEV static __device__ __forceinline__

// SW ("swept" elements, FusedGradArgs::geo_swept): the element is a prism -- x and y are bilinear in two reference directions, z is
// linear in the third -- and the fused kernel hands the reference directions over in the order (in-plane, in-plane, sweep), so that
// dXdx[m][k] vanishes wherever exactly one of m, k is 2: five of its nine entries are left and the two products below take 15
// multiply-adds instead of 27 (the sums run over the same terms in the same order; the dropped ones are exact zeros).
constexpr bool swept_zero(int m, int k) { return (m < 2) != (k < 2); }
// g[c][k] = sum_m dXdx[m][k] du[c][m], du[c][m] = ug[m*3+c]   (linElas.h:90-95)
template <bool SW = false>
CPS_DEV void physical_grad(const double *ug, const double *qd, double g[3][3]) {
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double s = 0.;
#pragma unroll
      for (int m = 0; m < 3; m++)
        if (!(SW && swept_zero(m, k))) s += qd[1 + 3 * m + k] * ug[m * 3 + c];
      g[c][k] = s;
    }
}
// dv[k*3+c] = sum_m dXdx[k][m] T[c][m] wdetJ                  (linElas.h:148-153)
template <bool SW = false>
CPS_DEV void pull_back(const double T[3][3], const double *qd, double *dv) {
  // the reference scales every product by wdetJ; scaling the sum once differs by <= 2 ulp
  const double wdetJ = qd[0];
  if constexpr (SW) {   // five entries: scale THEM once instead of the nine sums
    double K[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int m = 0; m < 3; m++) K[k][m] = swept_zero(k, m) ? 0. : qd[1 + 3 * k + m] * wdetJ;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        double s = 0.;
#pragma unroll
        for (int m = 0; m < 3; m++)
          if (!swept_zero(k, m)) s += K[k][m] * T[c][m];
        dv[k * 3 + c] = s;
      }
    return;
  }
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double s = 0.;
#pragma unroll
      for (int m = 0; m < 3; m++)
        if (!(SW && swept_zero(k, m))) s += qd[1 + 3 * k + m] * T[c][m];
      dv[k * 3 + c] = s * wdetJ;
    }
}

// ---- linear elasticity (linElas.h:97-145; note the reference's shear terms
// are ss*(1-2nu)*e_ij/2 with the TENSOR strain e_ij) -------------------------
template <bool SW = false>
CPS_DEV void qf_linelas(const Phys ph, const double *ug, const double *qd, double *dv) {
  double g[3][3], sig[3][3];
  physical_grad<SW>(ug, qd, g);
  const double nu = ph.nu;
  const double ss = ph.E / ((1 + nu) * (1 - 2 * nu));
  const double e00 = g[0][0], e11 = g[1][1], e22 = g[2][2];
  const double e12 = (g[1][2] + g[2][1]) / 2., e02 = (g[0][2] + g[2][0]) / 2., e01 = (g[0][1] + g[1][0]) / 2.;
  sig[0][0] = ss * ((1 - nu) * e00 + nu * e11 + nu * e22);
  sig[1][1] = ss * (nu * e00 + (1 - nu) * e11 + nu * e22);
  sig[2][2] = ss * (nu * e00 + nu * e11 + (1 - nu) * e22);
  sig[1][2] = sig[2][1] = ss * (1 - 2 * nu) * e12 * 0.5;
  sig[0][2] = sig[2][0] = ss * (1 - 2 * nu) * e02 * 0.5;
  sig[0][1] = sig[1][0] = ss * (1 - 2 * nu) * e01 * 0.5;
  pull_back<SW>(sig, qd, dv);
}

// ---- Neo-Hookean small strain (hyperSS.h) ---------------------------------
CPS_DEV double log1p_series4(double x) {  // hyperSS.h:43-55
  double y = x / (2. + x);
  const double y2 = y * y;
  double sum = y;
  y *= y2; sum += y * (1. / 3);
  y *= y2; sum += y * (1. / 5);
  y *= y2; sum += y * (1. / 7);
  return 2 * sum;
}
CPS_DEV void lame(const Phys ph, double &lambda, double &TwoMu) {  // hyperSS.h:79-81
  TwoMu = ph.TwoMu;
  lambda = ph.lambda;
}
template <bool SW = false>
CPS_DEV void qf_hyperss_f(const Phys ph, const double *ug, const double *qd, double *dv, double *st) {
  double lambda, TwoMu, g[3][3], sig[3][3];
  lame(ph, lambda, TwoMu);
  physical_grad<SW>(ug, qd, g);
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int k = 0; k < 3; k++) st[3 * c + k] = g[c][k];
  const double llv = log1p_series4(g[0][0] + g[1][1] + g[2][2]);
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++)
      sig[a][b] = TwoMu * ((g[a][b] + g[b][a]) / 2.) + (a == b ? lambda * llv : 0.);
  pull_back<SW>(sig, qd, dv);
}
template <bool SW = false>
CPS_DEV void qf_hyperss_df(const Phys ph, const double *dug, const double *qd, const double *st, double *dv) {
  double lambda, TwoMu, dg[3][3], ds[3][3];
  lame(ph, lambda, TwoMu);
  physical_grad<SW>(dug, qd, dg);
  const double lambda_bar = lambda / (1 + (st[0] + st[4] + st[8]));  // hyperSS.h:294-295
  const double ltr = lambda_bar * (dg[0][0] + dg[1][1] + dg[2][2]);
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++)
      ds[a][b] = TwoMu * ((dg[a][b] + dg[b][a]) / 2.) + (a == b ? ltr : 0.);
  pull_back<SW>(ds, qd, dv);
}

// ---- Neo-Hookean finite strain (hyperFS.h) --------------------------------
CPS_DEV double log1p_series4_shifted(double x) {  // hyperFS.h:45-67
  const double sqrt2 = 1.4142135623730951;       // == sqrt(2.) in IEEE double
  const double ln2h = 0.6931471805599453 / 2;    // == log(2.)/2
  const double left = sqrt2 / 2 - 1, right = sqrt2 - 1;
  double sum = 0.;
  if (x < left) { sum -= ln2h; x = 1 + 2 * x; }
  else if (right < x) { sum += ln2h; x = (x - 1) / 2; }
  double y = x / (2. + x);
  const double y2 = y * y;
  sum += y;
  y *= y2; sum += y * (1. / 3);
  y *= y2; sum += y * (1. / 5);
  y *= y2; sum += y * (1. / 7);
  return 2 * sum;
}
// Symmetric 3x3 kept as 6 scalars in the reference's packing (hyperFS.h:91):
// 0:(0,0) 1:(1,1) 2:(2,2) 3:(1,2) 4:(0,2) 5:(0,1)
#define CPS_SYM(w, a, b) ((a) == (b) ? w[a] : w[6 - (a) - (b)])
struct FSState { double S[6], Ci[6], llnj; };
// FAST_S: S = mu I + (llnj - mu) C^-1, algebraically equal to the reference's
// llnj C^-1 + mu C^-1 E2 (since C^-1 E2 = I - C^-1).  Used ONLY by the Jacobian, where S enters
// through grad(du) S next to the O(mu) term F dS, so its cancellation error (~1e-16 mu absolute) is
// far inside the 1e-10 bar; the residual keeps the reference's cancellation-free form.
template <bool FAST_S>
CPS_DEV void fs_state(double lambda, double mu, const double g[3][3], FSState &s) {  // hyperFS.h:85-142
  constexpr int J[6] = {0, 1, 2, 1, 0, 0}, K[6] = {0, 1, 2, 2, 2, 1};
  double E2[6];
#pragma unroll
  for (int m = 0; m < 6; m++) {
    double t = g[J[m]][K[m]] + g[K[m]][J[m]];
#pragma unroll
    for (int n = 0; n < 3; n++) t += g[n][J[m]] * g[n][K[m]];
    E2[m] = t;
  }
  const double detCm1 =  // hyperFS.h:72-80
      E2[0] * (E2[1] * E2[2] - E2[3] * E2[3]) + E2[5] * (E2[4] * E2[3] - E2[5] * E2[2]) +
      E2[4] * (E2[5] * E2[3] - E2[4] * E2[1]) + E2[0] + E2[1] + E2[2] + E2[0] * E2[1] +
      E2[0] * E2[2] + E2[1] * E2[2] - E2[5] * E2[5] - E2[4] * E2[4] - E2[3] * E2[3];
  const double C00 = 1 + E2[0], C11 = 1 + E2[1], C22 = 1 + E2[2], C12 = E2[3], C02 = E2[4], C01 = E2[5];
  const double A[6] = {C11 * C22 - C12 * C12, C00 * C22 - C02 * C02, C00 * C11 - C01 * C01,
                       C02 * C01 - C00 * C12, C01 * C12 - C02 * C11, C02 * C12 - C01 * C22};
  const double rden = 1. / (detCm1 + 1.);  // one reciprocal instead of six divisions (<= 1 ulp apart)
#pragma unroll
  for (int m = 0; m < 6; m++) s.Ci[m] = A[m] * rden;
  s.llnj = lambda * log1p_series4_shifted(detCm1) / 2.;
  if constexpr (FAST_S) {
    const double f = s.llnj - mu;
#pragma unroll
    for (int m = 0; m < 6; m++) s.S[m] = f * s.Ci[m] + (m < 3 ? mu : 0.);
  } else {
#pragma unroll
    for (int m = 0; m < 6; m++) {
      double t = s.llnj * s.Ci[m];
#pragma unroll
      for (int n = 0; n < 3; n++) t += mu * CPS_SYM(s.Ci, J[m], n) * CPS_SYM(E2, n, K[m]);
      s.S[m] = t;
    }
  }
}
CPS_DEV void fs_lame(const Phys ph, double &lambda, double &mu) {  // hyperFS.h:164-167
  double TwoMu;
  lame(ph, lambda, TwoMu);
  mu = TwoMu / 2;
}
CPS_DEV void fs_derived_state(double lambda, double mu, const double g[3][3], double *ds);
// ds (may be null, wave-uniform): the derived state of the tangent, written beside grad u (fs_derived_state)
template <bool SW = false>
CPS_DEV void qf_hyperfs_f(const Phys ph, const double *ug, const double *qd, double *dv, double *st, double *ds = nullptr) {
  double lambda, mu, g[3][3], P[3][3];
  fs_lame(ph, lambda, mu);
  physical_grad<SW>(ug, qd, g);
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int k = 0; k < 3; k++) st[3 * c + k] = g[c][k];
  if (ds) fs_derived_state(lambda, mu, g, ds);
  FSState s;
  fs_state<false>(lambda, mu, g, s);
#pragma unroll
  for (int a = 0; a < 3; a++)  // P = F S, F = I + grad u   (hyperFS.h:262-268)
#pragma unroll
    for (int b = 0; b < 3; b++) {
      double t = 0.;
#pragma unroll
      for (int m = 0; m < 3; m++) t += (g[a][m] + (a == m ? 1. : 0.)) * CPS_SYM(s.S, m, b);
      P[a][b] = t;
    }
  pull_back<SW>(P, qd, dv);
}
// 1/x by v_rcp_f64 + two Newton steps (<= 1 ulp for the normal, well-scaled arguments met here).
CPS_DEV double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.), r);
  return __builtin_fma(r, __builtin_fma(-x, r, 1.), r);
}
// The reference's series (hyperFS.h:45-67) with its division done by rcp_nr.
CPS_DEV double log1p_series4_shifted_fast(double x) {
  const double sqrt2 = 1.4142135623730951, ln2h = 0.6931471805599453 / 2;
  const double left = sqrt2 / 2 - 1, right = sqrt2 - 1;
  const bool lo = x < left, hi = right < x;
  double sum = lo ? -ln2h : (hi ? ln2h : 0.);
  x = lo ? 1 + 2 * x : (hi ? (x - 1) * 0.5 : x);
  double y = x * rcp_nr(2. + x);
  const double y2 = y * y;
  sum += y;
  y *= y2; sum += y * (1. / 3);
  y *= y2; sum += y * (1. / 5);
  y *= y2; sum += y * (1. / 7);
  return 2 * sum;
}
// Tangent of the finite-strain model (HyperFSdF, hyperFS.h:286-464) in its SPATIAL form.  The reference
// evaluates  dP = grad(du) S + F dS,  S = mu I + f C^-1 (f = lambda ln J - mu),
// dS = lambda (C^-1:dE) C^-1 - 2 f C^-1 dE C^-1,  dE = sym(grad(du)^T F).  With h = grad(du) F^-1 one has
// dE = F^T sym(h) F, C^-1 = F^-1 F^-T, C^-1:dE = tr h, and the sum collapses to
//     dP = mu grad(du) + (lambda tr(h) I - f h^T) F^-T,
// the same linear map with ~45 % fewer flops per point and no symmetric 6-packs to keep live (equal to the reference's
// evaluation to rounding: ~1e-15 relative for the conditioning of F met in elasticity; the 1e-10 parity tests
// cover it).  ln J uses the reference's own series on det C - 1 = J^2 - 1.  F^-1 = A / J is never formed: A enters
// unscaled and 1/J^2 is folded into the two scalars.
template <bool SW = false>
CPS_DEV void qf_hyperfs_df(const Phys ph, const double *dug, const double *qd, const double *st, double *dv) {
  double lambda, mu, dg[3][3], F[3][3], A[3][3], h[3][3], M[3][3], dP[3][3];
  fs_lame(ph, lambda, mu);
  physical_grad<SW>(dug, qd, dg);
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int k = 0; k < 3; k++) F[c][k] = st[3 * c + k] + (c == k ? 1. : 0.);
#pragma unroll
  for (int r = 0; r < 3; r++)  // A = adj(F): F^-1 = A / det F
#pragma unroll
    for (int s = 0; s < 3; s++) {
      const int s1 = (s + 1) % 3, s2 = (s + 2) % 3, r1 = (r + 1) % 3, r2 = (r + 2) % 3;
      A[r][s] = F[s1][r1] * F[s2][r2] - F[s1][r2] * F[s2][r1];
    }
  const double Jdet = F[0][0] * A[0][0] + F[0][1] * A[1][0] + F[0][2] * A[2][0];
  const double rJ = rcp_nr(Jdet), rJ2 = rJ * rJ;
  const double llnj = lambda * log1p_series4_shifted_fast(__builtin_fma(Jdet, Jdet, -1.)) * 0.5;  // hyperFS.h:130-131
  const double c2 = (llnj - mu) * rJ2;
#pragma unroll
  for (int a = 0; a < 3; a++)  // h J = grad(du) A
#pragma unroll
    for (int n = 0; n < 3; n++) {
      double t = 0.;
#pragma unroll
      for (int m = 0; m < 3; m++) t += dg[a][m] * A[m][n];
      h[a][n] = t;
    }
  const double c1trh = lambda * rJ2 * (h[0][0] + h[1][1] + h[2][2]);
#pragma unroll
  for (int a = 0; a < 3; a++)  // M J^2 = lambda tr(h) I - f h^T
#pragma unroll
    for (int m = 0; m < 3; m++) M[a][m] = (a == m ? c1trh : 0.) - c2 * h[m][a];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) {
      double t = mu * dg[a][b];
#pragma unroll
      for (int m = 0; m < 3; m++) t += M[a][m] * A[b][m];
      dP[a][b] = t;
    }
  pull_back<SW>(dP, qd, dv);
}
// DERIVED STATE (index.rst:458-464 discusses the same storage / recompute trade): what the tangent
// above needs of the state is F^-1 (nine numbers) and f = lambda ln J - mu (one); forming them from the stored grad u costs
// an adjugate, a determinant, a reciprocal and the log series at every point of every Jacobian apply.  The residual kernel
// can write them once per Newton step beside grad u (ten doubles per point instead of nine to read afterwards):
//   ds[3 r + s] = F^-1[r][s],  ds[9] = lambda ln J - mu.
CPS_DEV void fs_derived_state(double lambda, double mu, const double g[3][3], double *ds) {
  double F[3][3], A[3][3];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int k = 0; k < 3; k++) F[c][k] = g[c][k] + (c == k ? 1. : 0.);
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int s = 0; s < 3; s++) {
      const int s1 = (s + 1) % 3, s2 = (s + 2) % 3, r1 = (r + 1) % 3, r2 = (r + 2) % 3;
      A[r][s] = F[s1][r1] * F[s2][r2] - F[s1][r2] * F[s2][r1];
    }
  const double Jdet = F[0][0] * A[0][0] + F[0][1] * A[1][0] + F[0][2] * A[2][0];
  const double rJ = rcp_nr(Jdet);
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int s = 0; s < 3; s++) ds[3 * r + s] = A[r][s] * rJ;
  ds[9] = lambda * log1p_series4_shifted_fast(__builtin_fma(Jdet, Jdet, -1.)) * 0.5 - mu;
}
// dP = mu grad(du) + (lambda tr(h) I - f h^T) F^-T,  h = grad(du) F^-1, from the derived state
template <bool SW = false>
CPS_DEV void qf_hyperfs_df_ds(const Phys ph, const double *dug, const double *qd, const double *ds, double *dv) {
  double lambda, mu, dg[3][3], h[3][3], dP[3][3];
  fs_lame(ph, lambda, mu);
  physical_grad<SW>(dug, qd, dg);
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int n = 0; n < 3; n++) {
      double t = 0.;
#pragma unroll
      for (int m = 0; m < 3; m++) t += dg[a][m] * ds[3 * m + n];
      h[a][n] = t;
    }
  const double ltrh = lambda * (h[0][0] + h[1][1] + h[2][2]), f = ds[9];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) {
      double t = mu * dg[a][b] + ltrh * ds[3 * b + a];       // (lambda tr h I) F^-T
#pragma unroll
      for (int m = 0; m < 3; m++) t -= f * h[m][a] * ds[3 * b + m];
      dP[a][b] = t;
    }
  pull_back<SW>(dP, qd, dv);
}
// The reference's own evaluation order (kept for A/B and as documentation of the map above).
template <bool SW = false>
CPS_DEV void qf_hyperfs_df_reference_form(const Phys ph, const double *dug, const double *qd, const double *st, double *dv) {
  constexpr int J[6] = {0, 1, 2, 1, 0, 0}, K[6] = {0, 1, 2, 2, 2, 1};
  double lambda, mu, dg[3][3], g[3][3], F[3][3];
  fs_lame(ph, lambda, mu);
  physical_grad<SW>(dug, qd, dg);
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      g[c][k] = st[3 * c + k];
      F[c][k] = g[c][k] + (c == k ? 1. : 0.);
    }
  FSState s;
  fs_state<true>(lambda, mu, g, s);
  double dE[6];  // sym(grad(du)^T F)   (hyperFS.h:381-389); on the diagonal the two products coincide
#pragma unroll
  for (int m = 0; m < 6; m++) {
    double t = 0.;
    if (J[m] == K[m]) {
#pragma unroll
      for (int n = 0; n < 3; n++) t += dg[n][J[m]] * F[n][J[m]];
    } else {
#pragma unroll
      for (int n = 0; n < 3; n++) t += dg[n][J[m]] * F[n][K[m]] + F[n][J[m]] * dg[n][K[m]];
      t *= 0.5;
    }
    dE[m] = t;
  }
  // C^-1 : dE  (symmetric: diagonal + twice the off-diagonal)
  const double CidE = s.Ci[0] * dE[0] + s.Ci[1] * dE[1] + s.Ci[2] * dE[2] +
                      2. * (s.Ci[3] * dE[3] + s.Ci[4] * dE[4] + s.Ci[5] * dE[5]);
  double dECi[3][3], dS[6], dP[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) {
      double t = 0.;
#pragma unroll
      for (int m = 0; m < 3; m++) t += CPS_SYM(dE, a, m) * CPS_SYM(s.Ci, m, b);
      dECi[a][b] = t;
    }
  const double llnj_m2 = 2. * (s.llnj - mu), lCidE = lambda * CidE;
#pragma unroll
  for (int m = 0; m < 6; m++) {  // dS = lambda (C^-1:dE) C^-1 - 2 (llnj - mu) C^-1 dE C^-1: symmetric, 6 entries
    double t = 0.;
#pragma unroll
    for (int n = 0; n < 3; n++) t += CPS_SYM(s.Ci, J[m], n) * dECi[n][K[m]];
    dS[m] = lCidE * s.Ci[m] - llnj_m2 * t;  // hyperFS.h:438-442
  }
#pragma unroll
  for (int a = 0; a < 3; a++)  // dP = grad(du) S + F dS    (hyperFS.h:444-451)
#pragma unroll
    for (int b = 0; b < 3; b++) {
      double t = 0.;
#pragma unroll
      for (int m = 0; m < 3; m++) t += dg[a][m] * CPS_SYM(s.S, m, b) + F[a][m] * CPS_SYM(dS, m, b);
      dP[a][b] = t;
    }
  pull_back<SW>(dP, qd, dv);
}

// ---- geometry (common.h:47-101).  Jg[d*3+c] = d x_c / d xi_d ---------------
CPS_DEV void qf_setup_geo(const double *Jg, double w, double *qd) {
  double adj[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int s = 0; s < 3; s++) {
      const int a = (s + 1) % 3, b = (s + 2) % 3, c = (r + 1) % 3, d = (r + 2) % 3;
      // J[row][col] = d x_row / d xi_col = Jg[col*3 + row]
      adj[r][s] = Jg[c * 3 + a] * Jg[d * 3 + b] - Jg[d * 3 + a] * Jg[c * 3 + b];
    }
  const double detJ = Jg[0] * adj[0][0] + Jg[1] * adj[0][1] + Jg[2] * adj[0][2];
  qd[0] = w * detJ;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int s = 0; s < 3; s++) qd[1 + 3 * r + s] = adj[r][s] / detJ;
}

// The same with one reciprocal (rcp_nr, <= 1 ulp) instead of nine divisions: used where the factors are RECOMPUTED per
// point inside the fused kernel (FusedGradArgs::geo) rather than read back.
CPS_DEV void qf_setup_geo_rcp(const double *Jg, double w, double *qd) {
  double adj[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int s = 0; s < 3; s++) {
      const int a = (s + 1) % 3, b = (s + 2) % 3, c = (r + 1) % 3, d = (r + 2) % 3;
      adj[r][s] = Jg[c * 3 + a] * Jg[d * 3 + b] - Jg[d * 3 + a] * Jg[c * 3 + b];
    }
  const double detJ = Jg[0] * adj[0][0] + Jg[1] * adj[0][1] + Jg[2] * adj[0][2];
  const double rdet = rcp_nr(detJ);
  qd[0] = w * detJ;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int s = 0; s < 3; s++) qd[1 + 3 * r + s] = adj[r][s] * rdet;
}

// ---- uniform dispatch used by the fused kernels ----------------------------
// HAS_STATE_IN : Jacobians of the non-linear models read the stored gradu
// HAS_STATE_OUT: their residuals write it
// nstate: components of the state read per point (9: grad u; 10: the derived state of the finite-strain tangent)
template <int QF> struct QFTraits;
template <> struct QFTraits<QF_LINELAS>    { static constexpr bool state_in = false, state_out = false; static constexpr int nstate = 9; };
template <> struct QFTraits<QF_HYPERSS_F>  { static constexpr bool state_in = false, state_out = true;  static constexpr int nstate = 9; };
template <> struct QFTraits<QF_HYPERSS_DF> { static constexpr bool state_in = true,  state_out = false; static constexpr int nstate = 9; };
template <> struct QFTraits<QF_HYPERFS_F>  { static constexpr bool state_in = false, state_out = true;  static constexpr int nstate = 9; };
template <> struct QFTraits<QF_HYPERFS_DF> { static constexpr bool state_in = true,  state_out = false; static constexpr int nstate = 9; };
template <> struct QFTraits<QF_HYPERFS_DF_DS> { static constexpr bool state_in = true, state_out = false; static constexpr int nstate = 10; };

template <int QF, bool SW = false>
CPS_DEV void qf_point(const Phys ph, const double *ug, const double *qd, const double *st_in,
                      double *dv, double *st_out, double *derived_out = nullptr) {
  if constexpr (QF == QF_LINELAS) qf_linelas<SW>(ph, ug, qd, dv);
  else if constexpr (QF == QF_HYPERSS_F) qf_hyperss_f<SW>(ph, ug, qd, dv, st_out);
  else if constexpr (QF == QF_HYPERSS_DF) qf_hyperss_df<SW>(ph, ug, qd, st_in, dv);
  else if constexpr (QF == QF_HYPERFS_F) qf_hyperfs_f<SW>(ph, ug, qd, dv, st_out, derived_out);
  else if constexpr (QF == QF_HYPERFS_DF_DS) qf_hyperfs_df_ds<SW>(ph, ug, qd, st_in, dv);
#ifdef CPS_FS_REFERENCE_FORM  // A/B builds only
  else if constexpr (QF == QF_HYPERFS_DF) qf_hyperfs_df_reference_form<SW>(ph, ug, qd, st_in, dv);
#else
  else if constexpr (QF == QF_HYPERFS_DF) qf_hyperfs_df<SW>(ph, ug, qd, st_in, dv);
#endif
}

}  // namespace cps
