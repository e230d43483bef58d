/*
 * oracle_qfunctions.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's pointwise physics ("QFunctions"), in the
 * libCEED user-callback form `int f(void *ctx, CeedInt Q, in, out)` so the
 * oracle operator (oracle_ceed.c) can run either these or the reference's own
 * compiled callbacks (oracle/_ref) through the same pointer.
 *
 * Pinned by tests/test_oracle_qfunctions.py against golden vectors produced
 * from the reference headers themselves (oracle/gen_golden.py).
 *
 * Array conventions (reference qfunctions/linElas.h:43,62-71,148-153):
 *   GRAD input   in[0][(d*3 + c)*Q + i] = d u_c / d xi_d        (9 x Q)
 *   qdata        in[1][0*Q+i] = w*detJ, in[1][(1+3r+s)*Q+i] = dXdx[r][s]
 *   stored state gradu[(3*c + k)*Q + i] = d u_c / d x_k          (hyperFS.h:215-220)
 *   GRAD output  out[0][(k*3 + c)*Q + i]
 */
#include <ceed.h>
#include <math.h>
#include <string.h>

typedef struct { CeedScalar nu, E; } OraclePhysics; /* elasticity.h:33-36 */

typedef double m33[3][3];

/* ---- small helpers shared by every residual / Jacobian callback --------- */
static inline void load_ref_grad(const CeedScalar *ug, CeedInt Q, CeedInt i, m33 du) {
  for (int c = 0; c < 3; c++)
    for (int d = 0; d < 3; d++) du[c][d] = ug[(d * 3 + c) * Q + i];
}
static inline double load_qdata(const CeedScalar *qd, CeedInt Q, CeedInt i, m33 dXdx) {
  for (int r = 0; r < 3; r++)
    for (int s = 0; s < 3; s++) dXdx[r][s] = qd[(1 + 3 * r + s) * Q + i];
  return qd[i];
}
/* physical gradient g[c][k] = sum_m dXdx[m][k] du[c][m]   (linElas.h:90-95) */
static inline void to_physical(const m33 du, const m33 dXdx, m33 g) {
  for (int c = 0; c < 3; c++)
    for (int k = 0; k < 3; k++) {
      double s = 0;
      for (int m = 0; m < 3; m++) s += dXdx[m][k] * du[c][m];
      g[c][k] = s;
    }
}
/* out[k][c] = sum_m dXdx[k][m] T[c][m] wdetJ              (linElas.h:148-153) */
static inline void pull_back(const m33 T, const m33 dXdx, double wdetJ,
                             CeedScalar *dv, CeedInt Q, CeedInt i) {
  for (int c = 0; c < 3; c++)
    for (int k = 0; k < 3; k++) {
      double s = 0;
      for (int m = 0; m < 3; m++) s += dXdx[k][m] * T[c][m] * wdetJ;
      dv[(k * 3 + c) * Q + i] = s;
    }
}
static inline void sym_part(const m33 g, m33 e) {
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) e[a][b] = (g[a][b] + g[b][a]) / 2.;
}

/* ------------------------------------------------------------------------- */
/* SetupGeo (qfunctions/common.h:47-101): w detJ and the inverse Jacobian.   */
/* ------------------------------------------------------------------------- */
static int Oracle_SetupGeo(void *ctx, CeedInt Q, const CeedScalar *const *in,
                           CeedScalar *const *out) {
  (void)ctx;
  const CeedScalar *Jg = in[0], *w = in[1];
  CeedScalar *qd = out[0];
  for (CeedInt i = 0; i < Q; i++) {
    m33 J, adj; /* J[r][s] = d x_r / d xi_s */
    for (int r = 0; r < 3; r++)
      for (int s = 0; s < 3; s++) J[r][s] = Jg[(s * 3 + r) * Q + i];
    for (int r = 0; r < 3; r++)
      for (int s = 0; s < 3; s++) {
        int a = (s + 1) % 3, b = (s + 2) % 3, c = (r + 1) % 3, d = (r + 2) % 3;
        adj[r][s] = J[a][c] * J[b][d] - J[a][d] * J[b][c];
      }
    const double detJ = J[0][0] * adj[0][0] + J[1][0] * adj[0][1] + J[2][0] * adj[0][2];
    qd[i] = w[i] * detJ;
    for (int r = 0; r < 3; r++)
      for (int s = 0; s < 3; s++) qd[(1 + 3 * r + s) * Q + i] = adj[r][s] / detJ;
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Linear elasticity (qfunctions/linElas.h:39-158 F, :163-280 dF).           */
/* F and dF are the same linear map of the reference gradient.               */
/* ------------------------------------------------------------------------- */
static inline void hooke(double E, double nu, const m33 e, m33 sig) {
  const double ss = E / ((1 + nu) * (1 - 2 * nu));
  sig[0][0] = ss * ((1 - nu) * e[0][0] + nu * e[1][1] + nu * e[2][2]);
  sig[1][1] = ss * (nu * e[0][0] + (1 - nu) * e[1][1] + nu * e[2][2]);
  sig[2][2] = ss * (nu * e[0][0] + nu * e[1][1] + (1 - nu) * e[2][2]);
  sig[1][2] = sig[2][1] = ss * (1 - 2 * nu) * e[1][2] * 0.5;
  sig[0][2] = sig[2][0] = ss * (1 - 2 * nu) * e[0][2] * 0.5;
  sig[0][1] = sig[1][0] = ss * (1 - 2 * nu) * e[0][1] * 0.5;
}
static int Oracle_LinElas(void *ctx, CeedInt Q, const CeedScalar *const *in,
                          CeedScalar *const *out) {
  const OraclePhysics *ph = (const OraclePhysics *)ctx;
  for (CeedInt i = 0; i < Q; i++) {
    m33 du, dXdx, g, e, sig;
    load_ref_grad(in[0], Q, i, du);
    const double wdetJ = load_qdata(in[1], Q, i, dXdx);
    to_physical(du, dXdx, g);
    sym_part(g, e);
    hooke(ph->E, ph->nu, e, sig);
    pull_back(sig, dXdx, wdetJ, out[0], Q, i);
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Neo-Hookean, small strain (qfunctions/hyperSS.h:43-55,60-182,187-321).    */
/* ------------------------------------------------------------------------- */
/* 2*atanh-type series for log(1+x), 4 terms (hyperSS.h:43-55). */
static inline double log1p_series4(double x) {
  double y = x / (2. + x);
  const double y2 = y * y;
  double sum = y;
  y *= y2; sum += y / 3;
  y *= y2; sum += y / 5;
  y *= y2; sum += y / 7;
  return 2 * sum;
}
static inline void lame(const OraclePhysics *ph, double *lambda, double *TwoMu) {
  *TwoMu = ph->E / (1 + ph->nu);
  const double Kbulk = ph->E / (3 * (1 - 2 * ph->nu));
  *lambda = (3 * Kbulk - *TwoMu) / 3;
}
static int Oracle_HyperSSF(void *ctx, CeedInt Q, const CeedScalar *const *in,
                           CeedScalar *const *out) {
  double lambda, TwoMu;
  lame((const OraclePhysics *)ctx, &lambda, &TwoMu);
  CeedScalar *state = out[1];
  for (CeedInt i = 0; i < Q; i++) {
    m33 du, dXdx, g, e, sig;
    load_ref_grad(in[0], Q, i, du);
    const double wdetJ = load_qdata(in[1], Q, i, dXdx);
    to_physical(du, dXdx, g);
    for (int c = 0; c < 3; c++)
      for (int k = 0; k < 3; k++) state[(3 * c + k) * Q + i] = g[c][k];
    sym_part(g, e);
    const double llv = log1p_series4(e[0][0] + e[1][1] + e[2][2]);
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) sig[a][b] = TwoMu * e[a][b] + (a == b ? lambda * llv : 0.);
    /* reference order on the diagonal is lambda*llv + TwoMu*e (commutative) */
    pull_back(sig, dXdx, wdetJ, out[0], Q, i);
  }
  return 0;
}
static int Oracle_HyperSSdF(void *ctx, CeedInt Q, const CeedScalar *const *in,
                            CeedScalar *const *out) {
  double lambda, TwoMu;
  lame((const OraclePhysics *)ctx, &lambda, &TwoMu);
  const CeedScalar *state = in[2];
  for (CeedInt i = 0; i < Q; i++) {
    m33 ddu, dXdx, dg, de, dsig;
    load_ref_grad(in[0], Q, i, ddu);
    const double wdetJ = load_qdata(in[1], Q, i, dXdx);
    to_physical(ddu, dXdx, dg);
    sym_part(dg, de);
    const double strain_vol = state[0 * Q + i] + state[4 * Q + i] + state[8 * Q + i];
    const double lambda_bar = lambda / (1 + strain_vol);
    const double ltr = lambda_bar * (de[0][0] + de[1][1] + de[2][2]);
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) dsig[a][b] = TwoMu * de[a][b] + (a == b ? ltr : 0.);
    pull_back(dsig, dXdx, wdetJ, out[0], Q, i);
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Neo-Hookean, finite strain (qfunctions/hyperFS.h:45-67,72-80,85-142,      */
/* 147-281 F, 286-464 dF).                                                   */
/* ------------------------------------------------------------------------- */
/* Range-shifted series for log(1+x) (hyperFS.h:45-67): one halving or
 * doubling step brings J into (sqrt2/2, sqrt2) before the 4-term series. */
static inline double log1p_series4_shifted(double x) {
  const double left = sqrt(2.) / 2 - 1, right = sqrt(2.) - 1;
  double sum = 0;
  if (x < left) { sum -= log(2.) / 2; x = 1 + 2 * x; }
  else if (right < x) { sum += log(2.) / 2; x = (x - 1) / 2; }
  double y = x / (2. + x);
  const double y2 = y * y;
  sum += y;
  y *= y2; sum += y / 3;
  y *= y2; sum += y / 5;
  y *= y2; sum += y / 7;
  return 2 * sum;
}
/* Voigt-like packing used by the reference (hyperFS.h:91,99-102):
 * m = 0..5 <-> (0,0),(1,1),(2,2),(1,2),(0,2),(0,1). */
static const int VJ[6] = {0, 1, 2, 1, 0, 0}, VK[6] = {0, 1, 2, 2, 2, 1};
static inline void unpack6(const double w[6], m33 A) {
  for (int m = 0; m < 6; m++) A[VJ[m]][VK[m]] = A[VK[m]][VJ[m]] = w[m];
}
typedef struct { m33 S, Cinv; double llnj; } FSState;
/* S (2nd Piola-Kirchhoff), C^-1 and (lambda/2) log(det C) from grad u. */
static inline void fs_state(double lambda, double mu, const m33 g, FSState *st) {
  double E2w[6], Cinvw[6], Sw[6];
  for (int m = 0; m < 6; m++) {
    double s = g[VJ[m]][VK[m]] + g[VK[m]][VJ[m]];
    for (int n = 0; n < 3; n++) s += g[n][VJ[m]] * g[n][VK[m]];
    E2w[m] = s;
  }
  m33 E2, C;
  unpack6(E2w, E2);
  /* det C - 1 without cancellation (hyperFS.h:72-80) */
  const double detCm1 =
      E2w[0] * (E2w[1] * E2w[2] - E2w[3] * E2w[3]) + E2w[5] * (E2w[4] * E2w[3] - E2w[5] * E2w[2]) +
      E2w[4] * (E2w[5] * E2w[3] - E2w[4] * E2w[1]) + E2w[0] + E2w[1] + E2w[2] + E2w[0] * E2w[1] +
      E2w[0] * E2w[2] + E2w[1] * E2w[2] - E2w[5] * E2w[5] - E2w[4] * E2w[4] - E2w[3] * E2w[3];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) C[a][b] = E2[a][b] + (a == b ? 1. : 0.);
  /* adjugate of the symmetric C in the same packing (hyperFS.h:116-122) */
  const double A[6] = {C[1][1] * C[2][2] - C[1][2] * C[2][1], C[0][0] * C[2][2] - C[0][2] * C[2][0],
                       C[0][0] * C[1][1] - C[0][1] * C[1][0], C[0][2] * C[1][0] - C[0][0] * C[1][2],
                       C[0][1] * C[1][2] - C[0][2] * C[1][1], C[0][2] * C[2][1] - C[0][1] * C[2][2]};
  for (int m = 0; m < 6; m++) Cinvw[m] = A[m] / (detCm1 + 1.);
  unpack6(Cinvw, st->Cinv);
  st->llnj = lambda * log1p_series4_shifted(detCm1) / 2.;
  for (int m = 0; m < 6; m++) {
    double s = st->llnj * Cinvw[m];
    for (int n = 0; n < 3; n++) s += mu * st->Cinv[VJ[m]][n] * E2[n][VK[m]];
    Sw[m] = s;
  }
  unpack6(Sw, st->S);
}
static inline void fs_lame(const OraclePhysics *ph, double *lambda, double *mu) {
  double TwoMu;
  lame(ph, lambda, &TwoMu);
  *mu = TwoMu / 2;
}
static int Oracle_HyperFSF(void *ctx, CeedInt Q, const CeedScalar *const *in,
                           CeedScalar *const *out) {
  double lambda, mu;
  fs_lame((const OraclePhysics *)ctx, &lambda, &mu);
  CeedScalar *state = out[1];
  for (CeedInt i = 0; i < Q; i++) {
    m33 du, dXdx, g, F, P;
    FSState st;
    load_ref_grad(in[0], Q, i, du);
    const double wdetJ = load_qdata(in[1], Q, i, dXdx);
    to_physical(du, dXdx, g);
    for (int c = 0; c < 3; c++)
      for (int k = 0; k < 3; k++) {
        state[(3 * c + k) * Q + i] = g[c][k];
        F[c][k] = g[c][k] + (c == k ? 1. : 0.);
      }
    fs_state(lambda, mu, g, &st);
    for (int a = 0; a < 3; a++) /* P = F S */
      for (int b = 0; b < 3; b++) {
        double s = 0;
        for (int m = 0; m < 3; m++) s += F[a][m] * st.S[m][b];
        P[a][b] = s;
      }
    pull_back(P, dXdx, wdetJ, out[0], Q, i);
  }
  return 0;
}
static int Oracle_HyperFSdF(void *ctx, CeedInt Q, const CeedScalar *const *in,
                            CeedScalar *const *out) {
  double lambda, mu;
  fs_lame((const OraclePhysics *)ctx, &lambda, &mu);
  const CeedScalar *state = in[2];
  for (CeedInt i = 0; i < Q; i++) {
    m33 ddu, dXdx, dg, g, F, dE, dECinv, dS, dP;
    FSState st;
    double dEw[6];
    load_ref_grad(in[0], Q, i, ddu);
    const double wdetJ = load_qdata(in[1], Q, i, dXdx);
    to_physical(ddu, dXdx, dg);
    for (int c = 0; c < 3; c++)
      for (int k = 0; k < 3; k++) {
        g[c][k] = state[(3 * c + k) * Q + i];
        F[c][k] = g[c][k] + (c == k ? 1. : 0.);
      }
    fs_state(lambda, mu, g, &st);
    /* dE = sym(grad(du)^T F)  (hyperFS.h:381-395) */
    for (int m = 0; m < 6; m++) {
      double s = 0;
      for (int n = 0; n < 3; n++)
        s += (dg[n][VJ[m]] * F[n][VK[m]] + F[n][VJ[m]] * dg[n][VK[m]]) / 2.;
      dEw[m] = s;
    }
    unpack6(dEw, dE);
    double CinvdE = 0; /* C^-1 : dE */
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) CinvdE += st.Cinv[a][b] * dE[a][b];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        double s = 0;
        for (int m = 0; m < 3; m++) s += dE[a][m] * st.Cinv[m][b];
        dECinv[a][b] = s;
      }
    const double llnj_m = st.llnj - mu;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        double s = 0;
        for (int m = 0; m < 3; m++) s += st.Cinv[a][m] * dECinv[m][b];
        dS[a][b] = lambda * CinvdE * st.Cinv[a][b] - 2. * llnj_m * s;
      }
    for (int a = 0; a < 3; a++) /* dP = grad(du) S + F dS */
      for (int b = 0; b < 3; b++) {
        double s = 0;
        for (int m = 0; m < 3; m++) s += dg[a][m] * st.S[m][b] + F[a][m] * dS[m][b];
        dP[a][b] = s;
      }
    pull_back(dP, dXdx, wdetJ, out[0], Q, i);
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Forcing and manufactured solution (constantForce.h:39-70,                 */
/* manufacturedTrue.h:30-58, manufacturedForce.h:39-104).                    */
/* ------------------------------------------------------------------------- */
static int Oracle_SetupConstantForce(void *ctx, CeedInt Q, const CeedScalar *const *in,
                                     CeedScalar *const *out) {
  const CeedScalar *dir = (const CeedScalar *)ctx, *qd = in[1];
  for (CeedInt i = 0; i < Q; i++)
    for (int c = 0; c < 3; c++) out[0][c * Q + i] = dir[c] * qd[i];
  return 0;
}
/* u = 1e-8 (e^{2x} sin3y cos4z, e^{3y} sin4z cos2x, e^{4z} sin2x cos3y) */
static int Oracle_MMSTrueSoln(void *ctx, CeedInt Q, const CeedScalar *const *in,
                              CeedScalar *const *out) {
  (void)ctx;
  const CeedScalar *X = in[0];
  for (CeedInt i = 0; i < Q; i++) {
    const double x = X[i], y = X[Q + i], z = X[2 * Q + i];
    out[0][i] = exp(2 * x) * sin(3 * y) * cos(4 * z) / 1e8;
    out[0][Q + i] = exp(3 * y) * sin(4 * z) * cos(2 * x) / 1e8;
    out[0][2 * Q + i] = exp(4 * z) * sin(2 * x) * cos(3 * y) / 1e8;
  }
  return 0;
}
/* Forcing f = -div sigma(u_true), derived here from the true solution's second
 * derivatives for the stress the reference's LinElasF actually applies
 * (linElas.h:133-139): sigma_ii = lambda tr(e) + 2 mu e_ii but, off the
 * diagonal, sigma_ij = mu e_ij (the Voigt shear factor is applied to the tensor
 * strain, i.e. half the textbook shear stress).  The reference forcing
 * (manufacturedForce.h:62-101) is manufactured from that same operator:
 *   f_i = -[(lambda+2mu) u_i,ii + (mu/2)(u_i,jj + u_i,ll)
 *           + (lambda + mu/2)(u_j,ji + u_l,li)],   {i,j,l} = {x,y,z}.
 * Each u_c = exp(k_c p_c) sin(k_s p_s) cos(k_t p_t), cyclic s=c+1, t=c+2. */
static int Oracle_SetupMMSForce(void *ctx, CeedInt Q, const CeedScalar *const *in,
                                CeedScalar *const *out) {
  const OraclePhysics *ph = (const OraclePhysics *)ctx;
  const double mu = ph->E / (2 * (1 + ph->nu));
  const double lambda = ph->E * ph->nu / ((1 + ph->nu) * (1 - 2 * ph->nu));
  const double k[3] = {2., 3., 4.};
  const CeedScalar *X = in[0], *qd = in[1];
  for (CeedInt i = 0; i < Q; i++) {
    const double p[3] = {X[i], X[Q + i], X[2 * Q + i]};
    double f[3] = {0, 0, 0};
    for (int c = 0; c < 3; c++) {
      const int s = (c + 1) % 3, t = (c + 2) % 3;
      const double e = exp(k[c] * p[c]);
      const double ss = sin(k[s] * p[s]), cs = cos(k[s] * p[s]);
      const double st = sin(k[t] * p[t]), ct = cos(k[t] * p[t]);
      const double u = e * ss * ct;
      f[c] += (lambda + 2 * mu) * k[c] * k[c] * u - (mu / 2) * (k[s] * k[s] + k[t] * k[t]) * u;
      f[s] += (lambda + mu / 2) * k[c] * k[s] * e * cs * ct;  /* d_c d_s u_c */
      f[t] -= (lambda + mu / 2) * k[c] * k[t] * e * ss * st;  /* d_c d_t u_c */
    }
    for (int c = 0; c < 3; c++) out[0][c * Q + i] = -f[c] * qd[i] / 1e8;
  }
  return 0;
}

/* ---- strain energy (opEnergy, setuplibceed.c:651-670): energy density x w detJ ------------------
 * LinElasEnergy linElas.h:285-370, HyperSSEnergy hyperSS.h:326-412, HyperFSEnergy hyperFS.h:469-553;
 * restated AS WRITTEN, including the `strain_vol * mu` term of the first two. */
/* diag = 0: energy density x w detJ (1 output; inputs du, qdata);
 * diag = 1: the 8 diagnostic fields (inputs u, du, qdata): u (3), pressure, tr(strain), tr(strain^2), J, energy
 *           density -- LinElasDiagnostic linElas.h:376-480, HyperSSDiagnostic hyperSS.h:418-523, HyperFSDiagnostic
 *           hyperFS.h:559-662, as written */
static int oracle_energy(int model, int diag, void *ctx, CeedInt Q, const CeedScalar *const *in, CeedScalar *const *out) {
  const OraclePhysics *ph = (const OraclePhysics *)ctx;
  double lambda, TwoMu;
  lame(ph, &lambda, &TwoMu);
  const double mu = TwoMu / 2;
  const CeedScalar *ug = in[diag ? 1 : 0], *qd = in[diag ? 2 : 1];
  for (CeedInt i = 0; i < Q; i++) {
    m33 du, dXdx, g;
    load_ref_grad(ug, Q, i, du);
    const double wdetJ = load_qdata(qd, Q, i, dXdx);
    to_physical(du, dXdx, g);
    double en, press, tr1, tr2, J;
    if (model == 2) {
      double E2w[6];
      for (int m = 0; m < 6; m++) {
        double s = g[VJ[m]][VK[m]] + g[VK[m]][VJ[m]];
        for (int n = 0; n < 3; n++) s += g[n][VJ[m]] * g[n][VK[m]];
        E2w[m] = s;
      }
      const double detCm1 =
          E2w[0] * (E2w[1] * E2w[2] - E2w[3] * E2w[3]) + E2w[5] * (E2w[4] * E2w[3] - E2w[5] * E2w[2]) +
          E2w[4] * (E2w[5] * E2w[3] - E2w[4] * E2w[1]) + E2w[0] + E2w[1] + E2w[2] + E2w[0] * E2w[1] +
          E2w[0] * E2w[2] + E2w[1] * E2w[2] - E2w[5] * E2w[5] - E2w[4] * E2w[4] - E2w[3] * E2w[3];
      const double logj = log1p_series4_shifted(detCm1) / 2.;
      en = lambda * logj * logj / 2. - mu * logj + mu * (E2w[0] + E2w[1] + E2w[2]) / 2.;
      m33 E2;
      unpack6(E2w, E2);
      press = -lambda * logj;
      tr1 = (E2[0][0] + E2[1][1] + E2[2][2]) / 2.;
      tr2 = 0.;
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) tr2 += E2[a][b] * E2[b][a] / 4.;
      J = sqrt(detCm1 + 1);
    } else {
      m33 e;
      sym_part(g, e);
      const double sv = e[0][0] + e[1][1] + e[2][2];
      const double shear = (e[0][1] * e[0][1] + e[0][2] * e[0][2] + e[1][2] * e[1][2]) * 2 * mu;
      const double llv = model == 1 ? log1p_series4(sv) : 0.;
      if (model == 0) en = lambda * sv * sv / 2. + sv * mu + shear;
      else en = lambda * (1 + sv) * (llv - 1) + sv * mu + shear;
      press = model == 0 ? -lambda * sv : -lambda * llv;
      tr1 = sv;
      tr2 = 0.;
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) tr2 += e[a][b] * e[b][a];
      J = 1 + sv;
    }
    if (!diag) out[0][i] = en * wdetJ;
    else {
      for (int c = 0; c < 3; c++) out[0][c * Q + i] = in[0][c * Q + i];
      out[0][3 * Q + i] = press; out[0][4 * Q + i] = tr1; out[0][5 * Q + i] = tr2; out[0][6 * Q + i] = J; out[0][7 * Q + i] = en;
    }
  }
  return 0;
}
static int Oracle_LinElasEnergy(void *ctx, CeedInt Q, const CeedScalar *const *in, CeedScalar *const *out) { return oracle_energy(0, 0, ctx, Q, in, out); }
static int Oracle_HyperSSEnergy(void *ctx, CeedInt Q, const CeedScalar *const *in, CeedScalar *const *out) { return oracle_energy(1, 0, ctx, Q, in, out); }
static int Oracle_HyperFSEnergy(void *ctx, CeedInt Q, const CeedScalar *const *in, CeedScalar *const *out) { return oracle_energy(2, 0, ctx, Q, in, out); }
static int Oracle_LinElasDiagnostic(void *ctx, CeedInt Q, const CeedScalar *const *in, CeedScalar *const *out) { return oracle_energy(0, 1, ctx, Q, in, out); }
static int Oracle_HyperSSDiagnostic(void *ctx, CeedInt Q, const CeedScalar *const *in, CeedScalar *const *out) { return oracle_energy(1, 1, ctx, Q, in, out); }
static int Oracle_HyperFSDiagnostic(void *ctx, CeedInt Q, const CeedScalar *const *in, CeedScalar *const *out) { return oracle_energy(2, 1, ctx, Q, in, out); }

/* ------------------------------------------------------------------------- */
CEED_EXTERN CeedQFunctionUser OracleGetQFunction(const char *name) {
  static const struct { const char *n; CeedQFunctionUser f; } tab[] = {
      {"SetupGeo", Oracle_SetupGeo},
      {"LinElasF", Oracle_LinElas},
      {"LinElasdF", Oracle_LinElas},
      {"HyperSSF", Oracle_HyperSSF},
      {"HyperSSdF", Oracle_HyperSSdF},
      {"HyperFSF", Oracle_HyperFSF},
      {"HyperFSdF", Oracle_HyperFSdF},
      {"SetupConstantForce", Oracle_SetupConstantForce},
      {"SetupMMSForce", Oracle_SetupMMSForce},
      {"MMSTrueSoln", Oracle_MMSTrueSoln},
      {"LinElasEnergy", Oracle_LinElasEnergy},
      {"HyperSSEnergy", Oracle_HyperSSEnergy},
      {"HyperFSEnergy", Oracle_HyperFSEnergy},
      {"LinElasDiagnostic", Oracle_LinElasDiagnostic},
      {"HyperSSDiagnostic", Oracle_HyperSSDiagnostic},
      {"HyperFSDiagnostic", Oracle_HyperFSDiagnostic},
  };
  for (size_t i = 0; i < sizeof tab / sizeof tab[0]; i++)
    if (!strcmp(tab[i].n, name)) return tab[i].f;
  return NULL;
}
