// ceed_basis.cpp -- 1-D quadrature rules and the tensor H1 Lagrange basis tables; host side, set-up time
// only.  Reference call sites: src/setuplibceed.c:335-347 (basisu, basisx, basisEnergy, basisDiagnostic), :782-803 (level
// bases, GLL CtoF bases).  Pinned by tests/golden/basis_tables.npz (50-digit values, oracle/gen_tables_golden.py).
#include "ceed_impl.hpp"

using namespace cps;

static void legendre_pair(int n, double x, double *pn, double *pnm1) {
  double p0 = 1., p1 = x;
  if (n == 0) { *pn = 1.; *pnm1 = 0.; return; }
  for (int j = 2; j <= n; j++) {
    const double p2 = ((2. * j - 1.) * x * p1 - (j - 1.) * p0) / j;
    p0 = p1; p1 = p2;
  }
  *pn = p1; *pnm1 = p0;
}
extern "C" int CeedGaussQuadrature(CeedInt Q, CeedScalar *qref1d, CeedScalar *qweight1d) {
  for (int i = 0; i <= (Q - 1) / 2; i++) {
    double x = std::cos(M_PI * (2. * i + 1.) / (2. * Q)), pq, pqm1, dp;
    for (int it = 0; it < 100; it++) {
      legendre_pair(Q, x, &pq, &pqm1);
      dp = Q * (x * pq - pqm1) / (x * x - 1.);
      x -= pq / dp;
      if (it > 0 && std::fabs(pq) <= 10 * 2.220446049250313e-16) break;
    }
    legendre_pair(Q, x, &pq, &pqm1);
    dp = Q * (x * pq - pqm1) / (x * x - 1.);
    const double w = 2. / ((1. - x * x) * dp * dp);
    qref1d[i] = -x; qref1d[Q - 1 - i] = x;
    if (qweight1d) { qweight1d[i] = w; qweight1d[Q - 1 - i] = w; }
  }
  if (Q % 2) qref1d[Q / 2] = 0.;
  return 0;
}
extern "C" int CeedLobattoQuadrature(CeedInt Q, CeedScalar *qref1d, CeedScalar *qweight1d) {
  if (Q < 2) return ceed_error("Lobatto rule needs at least 2 points");
  const int n = Q - 1;
  qref1d[0] = -1.; qref1d[Q - 1] = 1.;
  if (qweight1d) qweight1d[0] = qweight1d[Q - 1] = 2. / (Q * (Q - 1.));
  for (int i = 1; i <= (Q - 1) / 2; i++) {
    double x = std::cos(M_PI * i / (double)n), pn, pnm1;
    for (int it = 0; it < 100; it++) {
      legendre_pair(n, x, &pn, &pnm1);
      const double dp = n * (x * pn - pnm1) / (x * x - 1.);
      const double d2p = (2. * x * dp - n * (n + 1.) * pn) / (1. - x * x);
      x -= dp / d2p;
      if (it > 0 && std::fabs(dp) <= 10 * 2.220446049250313e-16) break;
    }
    legendre_pair(n, x, &pn, &pnm1);
    const double w = 2. / (Q * (Q - 1.) * pn * pn);
    qref1d[i] = -x; qref1d[Q - 1 - i] = x;
    if (qweight1d) { qweight1d[i] = w; qweight1d[Q - 1 - i] = w; }
  }
  if (Q % 2) qref1d[Q / 2] = 0.;
  return 0;
}
// Fornberg's recurrence for the values and first derivatives at x of the P Lagrange polynomials on `nodes` -- the method
// libCEED itself uses; the test oracle evaluates the product formulas instead, so the two table generators
// are independent implementations (both are compared with the 50-digit fixture).
static void lagrange_fornberg(int P, const double *nodes, double x, double *val, double *der) {
  for (int j = 0; j < P; j++) val[j] = der[j] = 0.;
  double c1 = 1., c4 = nodes[0] - x;
  val[0] = 1.;
  for (int j = 1; j < P; j++) {
    double c2 = 1.;
    const double c5 = c4;
    c4 = nodes[j] - x;
    for (int k = 0; k < j; k++) {
      const double dx = nodes[j] - nodes[k];
      c2 *= dx;
      if (k == j - 1) {
        der[j] = c1 * (val[k] - c5 * der[k]) / c2;
        val[j] = -c1 * c5 * val[k] / c2;
      }
      der[k] = (c4 * der[k] - val[k]) / dx;
      val[k] = c4 * val[k] / dx;
    }
    c1 = c2;
  }
}
extern "C" int CeedBasisCreateTensorH1Lagrange(Ceed ceed, CeedInt dim, CeedInt ncomp, CeedInt P, CeedInt Q,
                                               CeedQuadMode qmode, CeedBasis *basis) {
  if (dim != 3) return ceed_error("only dim = 3 bases are supported");
  if (P < 2 || Q < 1 || P > MAXN1D || Q > MAXN1D) return ceed_error("basis sizes P=%d Q=%d outside [2,%d]", P, Q, MAXN1D);
  CeedBasis b = new CeedBasis_private;
  b->ceed = ceed; ceed_ref(ceed);
  b->dim = dim; b->ncomp = ncomp; b->P1d = P; b->Q1d = Q; b->qmode = qmode;
  b->interp1d.assign((size_t)P * Q, 0.); b->grad1d.assign((size_t)P * Q, 0.);
  b->qref1d.assign(Q, 0.); b->qweight1d.assign(Q, 0.); b->colo1d.assign((size_t)Q * Q, 0.);
  std::vector<double> nodes(P), tmp(Q);
  CHK(CeedLobattoQuadrature(P, nodes.data(), nullptr));
  if (qmode == CEED_GAUSS) CHK(CeedGaussQuadrature(Q, b->qref1d.data(), b->qweight1d.data()));
  else CHK(CeedLobattoQuadrature(Q, b->qref1d.data(), b->qweight1d.data()));
  for (int q = 0; q < Q; q++) {
    lagrange_fornberg(P, nodes.data(), b->qref1d[q], &b->interp1d[(size_t)q * P], &b->grad1d[(size_t)q * P]);
    // collocated derivative: Lagrange basis ON the quadrature points, differentiated there
    if (Q > 1) lagrange_fornberg(Q, b->qref1d.data(), b->qref1d[q], tmp.data(), &b->colo1d[(size_t)q * Q]);
  }
  *basis = b;
  return 0;
}
extern "C" int CeedBasisGetNumQuadraturePoints(CeedBasis b, CeedInt *Q) { *Q = b->Q1d * b->Q1d * b->Q1d; return 0; }
extern "C" int CeedBasisGetNumNodes(CeedBasis b, CeedInt *P) { *P = b->P1d * b->P1d * b->P1d; return 0; }
extern "C" int CeedBasisGetInterp1D(CeedBasis b, const CeedScalar **t) { *t = b->interp1d.data(); return 0; }
extern "C" int CeedBasisGetGrad1D(CeedBasis b, const CeedScalar **t) { *t = b->grad1d.data(); return 0; }
extern "C" int CeedBasisGetQWeights1D(CeedBasis b, const CeedScalar **t) { *t = b->qweight1d.data(); return 0; }
extern "C" int CeedBasisApply(CeedBasis, CeedInt, CeedTransposeMode, CeedEvalMode, CeedVector, CeedVector) {
  return ceed_error("standalone CeedBasisApply is not on the reference's path and is not provided by "
                    "/gpu/hip/mi355x: bases are applied inside the fused operator kernels");
}
extern "C" int CeedBasisDestroy(CeedBasis *basis) {
  if (!basis || !*basis) return 0;
  CeedBasis b = *basis;
  *basis = nullptr;
  if (b == CEED_BASIS_COLLOCATED) return 0;
  if (--b->refcount > 0) return 0;
  ceed_unref(b->ceed);
  delete b;
  return 0;
}
