/*
 * oracle_ceed.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * A plain-C CPU restatement of the libCEED semantics the reference
 * (ArashMehraban/CeedPetscSolid) relies on for its operator-apply path, behind
 * the same C ABI as the product (include/ceed.h), resource "/cpu/self/oracle".
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (ceedpetscsolid_amd/csrc) never does.
 *
 * What it restates, and from where:
 *   - libCEED itself is a third-party dependency that is ABSENT from
 *     /root/reference (Makefile:20 `CEED_DIR ?= ../..`, unpinned, API level
 *     ~v0.7).  Its published /cpu/self/ref algorithm is restated here from the
 *     semantics recorded in SURVEY.md Appendix A (A.1 Gauss, A.2 Lobatto,
 *     A.3 Lagrange tables, A.4 contraction, A.5 basis apply, A.6 restriction,
 *     A.7 operator apply, A.8 diagonal, A.9 identity QFunction), anchored on
 *     the reference's call sites (src/setuplibceed.c, src/matops.c, src/misc.c).
 *   - PARITY PIN: the pointwise physics (oracle_qfunctions.c) is pinned against
 *     the reference's own qfunctions/ *.h compiled in oracle/_ref.  The libCEED
 *     part has no golden vectors in the reference ("parity unpinned" against
 *     libCEED itself); it is pinned instead against independent analytic facts
 *     (numpy leggauss, polynomial exactness, adjointness, volume, FD-Jacobian).
 *
 * Single-threaded by default; OracleSetNumThreads(n) enables element-chunk
 * OpenMP threading of the per-element loop for the CPU baseline timing.
 */
#include <ceed.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_MAX_FIELDS 16

/* ------------------------------------------------------------------------- */
/* Error handling                                                             */
/* ------------------------------------------------------------------------- */
static int  g_err_return = 0;
static char g_err_msg[1024] = "";

static int oracle_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err_msg, sizeof g_err_msg, fmt, ap);
  va_end(ap);
  if (!g_err_return) {
    fprintf(stderr, "[oracle ceed] error: %s\n", g_err_msg);
    abort();
  }
  return 1;
}
#define CHK(x) do { int ierr_ = (x); if (ierr_) return ierr_; } while (0)

int CeedXSetErrorReturn(int enable) { g_err_return = enable; return 0; }
const char *CeedXLastError(void) { return g_err_msg; }

static int g_nthreads = 1;
CEED_EXTERN int OracleSetNumThreads(int n) {
  g_nthreads = n < 1 ? 1 : n;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Object layouts                                                             */
/* ------------------------------------------------------------------------- */
struct Ceed_private {
  int  refcount;
  char resource[64];
};

struct CeedVector_private {
  Ceed        ceed;
  int         refcount;
  CeedInt     length;
  CeedScalar *array;   /* current storage (owned or borrowed), or NULL */
  int         owned;   /* array was allocated/adopted by the vector    */
};

struct CeedElemRestriction_private {
  Ceed     ceed;
  int      refcount;
  CeedInt  nelem, elemsize, ncomp, compstride, lsize;
  CeedInt *offsets;    /* NULL for strided */
  CeedInt  strides[3]; /* node, comp, elem (strided only) */
};

struct CeedBasis_private {
  Ceed        ceed;
  int         refcount;
  CeedInt     dim, ncomp, P1d, Q1d;
  CeedScalar *interp1d, *grad1d, *qref1d, *qweight1d;
};

typedef struct {
  char         name[64];
  CeedInt      size;
  CeedEvalMode emode;
} QFField;

struct CeedQFunction_private {
  Ceed              ceed;
  int               refcount;
  CeedQFunctionUser f;
  char              source[256];
  void             *ctx;
  size_t            ctxsize;
  int               identity;
  CeedInt           identity_size;
  int               nin, nout;
  QFField           in[ORACLE_MAX_FIELDS], out[ORACLE_MAX_FIELDS];
};

typedef struct {
  int                 set;
  CeedElemRestriction rstr;
  CeedBasis           basis;
  CeedVector          vec;
} OpField;

struct CeedOperator_private {
  Ceed          ceed;
  int           refcount;
  CeedQFunction qf;
  OpField       in[ORACLE_MAX_FIELDS], out[ORACLE_MAX_FIELDS];
  int           composite, nsub;
  CeedOperator  sub[ORACLE_MAX_FIELDS];
  /* harness extensions (CeedX*): Dirichlet masks and the transfer fine-side scale */
  unsigned char *mask_in, *mask_out;
  CeedInt       mask_in_len, mask_out_len;
  int           mask_mode;
  CeedVector    scale;
  unsigned char *priority;   /* split-phase apply (CeedXOperatorSetOverlapSplit) */
  CeedInt       priority_len;
};

/* Sentinels: distinct addresses that are never dereferenced. */
static struct CeedVector_private          s_vec_active, s_vec_none;
static struct CeedElemRestriction_private s_rstr_none;
static struct CeedBasis_private           s_basis_colloc;
static struct CeedQFunction_private       s_qf_none;
static CeedRequest                        s_req_immediate, s_req_ordered;

const CeedVector          CEED_VECTOR_ACTIVE        = &s_vec_active;
const CeedVector          CEED_VECTOR_NONE          = &s_vec_none;
const CeedElemRestriction CEED_ELEMRESTRICTION_NONE = &s_rstr_none;
const CeedBasis           CEED_BASIS_COLLOCATED     = &s_basis_colloc;
const CeedQFunction       CEED_QFUNCTION_NONE       = &s_qf_none;
CeedRequest *const        CEED_REQUEST_IMMEDIATE    = &s_req_immediate;
CeedRequest *const        CEED_REQUEST_ORDERED      = &s_req_ordered;
const CeedInt             CEED_STRIDES_BACKEND[3]   = {-1, -1, -1};
const char *const         CeedMemTypes[]            = {"host", "device"};

/* ------------------------------------------------------------------------- */
/* Ceed                                                                       */
/* ------------------------------------------------------------------------- */
int CeedInit(const char *resource, Ceed *ceed) {
  if (!resource || strncmp(resource, "/cpu/self", 9))
    return oracle_error("oracle backend serves /cpu/self* only, got '%s'",
                        resource ? resource : "(null)");
  *ceed = calloc(1, sizeof **ceed);
  (*ceed)->refcount = 1;
  snprintf((*ceed)->resource, sizeof (*ceed)->resource, "/cpu/self/oracle");
  return 0;
}
static void ceed_ref(Ceed c) { c->refcount++; }
static void ceed_unref(Ceed c) { if (--c->refcount == 0) free(c); }
int CeedDestroy(Ceed *ceed) {
  if (!ceed || !*ceed) return 0;
  ceed_unref(*ceed);
  *ceed = NULL;
  return 0;
}
int CeedGetResource(Ceed ceed, const char **resource) {
  *resource = ceed->resource;
  return 0;
}
int CeedGetPreferredMemType(Ceed ceed, CeedMemType *type) {
  (void)ceed;
  *type = CEED_MEM_HOST;
  return 0;
}
int CeedXSetStream(Ceed ceed, void *s) { (void)ceed; (void)s; return 0; }
int CeedXSynchronize(Ceed ceed) { (void)ceed; return 0; }
/* the CPU oracle has no device queue to record: capture is refused, callers run eagerly */
typedef struct CeedXGraph_private *CeedXGraph;
int CeedXGraphBeginCapture(Ceed ceed) { (void)ceed; return 1; }
int CeedXGraphEndCapture(Ceed ceed, CeedXGraph *g) { (void)ceed; (void)g; return 1; }
int CeedXGraphIsStale(CeedXGraph graph, int *stale) { (void)graph; *stale = 0; return 0; }
int CeedXGraphLaunch(CeedXGraph g) { (void)g; return 1; }
int CeedXGraphDestroy(CeedXGraph *g) { if (g) *g = 0; return 0; }

/* ------------------------------------------------------------------------- */
/* CeedVector                                                                 */
/* ------------------------------------------------------------------------- */
int CeedVectorCreate(Ceed ceed, CeedInt length, CeedVector *vec) {
  CeedVector v = calloc(1, sizeof *v);
  v->ceed = ceed; ceed_ref(ceed);
  v->refcount = 1;
  v->length = length;
  *vec = v;
  return 0;
}
static int vec_host_only(CeedMemType m) {
  return m == CEED_MEM_HOST ? 0
         : oracle_error("oracle backend has no device memory");
}
static void vec_release(CeedVector v) {
  if (v->owned) free(v->array);
  v->array = NULL; v->owned = 0;
}
static void vec_ensure(CeedVector v) {
  if (!v->array) {
    v->array = calloc((size_t)(v->length > 0 ? v->length : 1), sizeof(CeedScalar));
    v->owned = 1;
  }
}
int CeedVectorSetArray(CeedVector v, CeedMemType mtype, CeedCopyMode cmode,
                       CeedScalar *array) {
  CHK(vec_host_only(mtype));
  switch (cmode) {
  case CEED_COPY_VALUES:
    if (!v->owned) { v->array = NULL; }
    vec_ensure(v);
    if (array) memcpy(v->array, array, sizeof(CeedScalar) * (size_t)v->length);
    break;
  case CEED_USE_POINTER:
    vec_release(v);
    v->array = array; v->owned = 0;
    break;
  case CEED_OWN_POINTER:
    vec_release(v);
    v->array = array; v->owned = 1;
    break;
  }
  return 0;
}
int CeedVectorTakeArray(CeedVector v, CeedMemType mtype, CeedScalar **array) {
  CHK(vec_host_only(mtype));
  if (array) *array = v->array;
  v->array = NULL; v->owned = 0;
  return 0;
}
int CeedVectorSetValue(CeedVector v, CeedScalar value) {
  vec_ensure(v);
  for (CeedInt i = 0; i < v->length; i++) v->array[i] = value;
  return 0;
}
int CeedVectorSyncArray(CeedVector v, CeedMemType mtype) {
  (void)v; return vec_host_only(mtype);
}
int CeedVectorGetArray(CeedVector v, CeedMemType mtype, CeedScalar **array) {
  CHK(vec_host_only(mtype));
  vec_ensure(v);
  *array = v->array;
  return 0;
}
int CeedVectorGetArrayRead(CeedVector v, CeedMemType mtype,
                           const CeedScalar **array) {
  CHK(vec_host_only(mtype));
  vec_ensure(v);
  *array = v->array;
  return 0;
}
int CeedVectorRestoreArray(CeedVector v, CeedScalar **array) {
  (void)v; if (array) *array = NULL; return 0;
}
int CeedVectorRestoreArrayRead(CeedVector v, const CeedScalar **array) {
  (void)v; if (array) *array = NULL; return 0;
}
int CeedVectorGetLength(CeedVector v, CeedInt *length) {
  *length = v->length; return 0;
}
int CeedVectorReciprocal(CeedVector v) {
  vec_ensure(v);
  for (CeedInt i = 0; i < v->length; i++)
    if (fabs(v->array[i]) > 1e-300) v->array[i] = 1. / v->array[i];
  return 0;
}
int CeedVectorDestroy(CeedVector *vec) {
  if (!vec || !*vec) return 0;
  CeedVector v = *vec;
  *vec = NULL;
  if (v == CEED_VECTOR_ACTIVE || v == CEED_VECTOR_NONE) return 0;
  if (--v->refcount > 0) return 0;
  vec_release(v);
  ceed_unref(v->ceed);
  free(v);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* CeedElemRestriction                                           */
/* ------------------------------------------------------------------------- */
int CeedElemRestrictionCreate(Ceed ceed, CeedInt nelem, CeedInt elemsize,
                              CeedInt ncomp, CeedInt compstride, CeedInt lsize,
                              CeedMemType mtype, CeedCopyMode cmode,
                              const CeedInt *offsets,
                              CeedElemRestriction *rstr) {
  CHK(vec_host_only(mtype));
  (void)cmode; /* always copied; the reference uses COPY_VALUES (:236) */
  size_t n = (size_t)nelem * (size_t)elemsize;
  for (size_t i = 0; i < n; i++) {
    long last = (long)offsets[i] + (long)(ncomp - 1) * compstride;
    if (offsets[i] < 0 || last >= lsize)
      return oracle_error("restriction offset %zu = %d out of range [0,%d)", i,
                          offsets[i], lsize);
  }
  CeedElemRestriction r = calloc(1, sizeof *r);
  r->ceed = ceed; ceed_ref(ceed);
  r->refcount = 1;
  r->nelem = nelem; r->elemsize = elemsize; r->ncomp = ncomp;
  r->compstride = compstride; r->lsize = lsize;
  r->offsets = malloc(sizeof(CeedInt) * (n ? n : 1));
  memcpy(r->offsets, offsets, sizeof(CeedInt) * n);
  *rstr = r;
  return 0;
}
int CeedElemRestrictionCreateStrided(Ceed ceed, CeedInt nelem, CeedInt elemsize,
                                     CeedInt ncomp, CeedInt lsize,
                                     const CeedInt strides[3],
                                     CeedElemRestriction *rstr) {
  CeedElemRestriction r = calloc(1, sizeof *r);
  r->ceed = ceed; ceed_ref(ceed);
  r->refcount = 1;
  r->nelem = nelem; r->elemsize = elemsize; r->ncomp = ncomp;
  r->compstride = 0; r->lsize = lsize;
  if (strides[0] < 0) { /* CEED_STRIDES_BACKEND: CPU choice is [e][c][n] */
    r->strides[0] = 1; r->strides[1] = elemsize; r->strides[2] = elemsize * ncomp;
  } else {
    memcpy(r->strides, strides, sizeof r->strides);
  }
  if ((long)nelem * elemsize * ncomp > lsize)
    return oracle_error("strided restriction larger than its L-vector");
  *rstr = r;
  return 0;
}
int CeedElemRestrictionCreateVector(CeedElemRestriction r, CeedVector *lvec,
                                    CeedVector *evec) {
  if (lvec) CHK(CeedVectorCreate(r->ceed, r->lsize, lvec));
  if (evec) CHK(CeedVectorCreate(r->ceed, r->nelem * r->elemsize * r->ncomp, evec));
  return 0;
}
/* E-layout is [e][c][n] everywhere in the oracle. */
static void rstr_gather(CeedElemRestriction r, const CeedScalar *l, CeedScalar *e) {
  const CeedInt S = r->elemsize, C = r->ncomp;
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_nthreads) if (g_nthreads > 1) schedule(static)
#endif
  for (CeedInt el = 0; el < r->nelem; el++)
    for (CeedInt c = 0; c < C; c++)
      for (CeedInt n = 0; n < S; n++) {
        size_t li = r->offsets
                    ? (size_t)r->offsets[(size_t)el * S + n] + (size_t)c * r->compstride
                    : (size_t)n * r->strides[0] + (size_t)c * r->strides[1] +
                      (size_t)el * r->strides[2];
        e[((size_t)el * C + c) * S + n] = l[li];
      }
}
static void rstr_scatter_add(CeedElemRestriction r, const CeedScalar *e, CeedScalar *l) {
  const CeedInt S = r->elemsize, C = r->ncomp;
#ifdef _OPENMP
  if (g_nthreads > 1) {   /* threaded runs (bench baseline, full-size tests): element chunks, atomic adds (order free: ~1e-16) */
#pragma omp parallel for num_threads(g_nthreads) schedule(static)
    for (CeedInt el = 0; el < r->nelem; el++)
      for (CeedInt c = 0; c < C; c++)
        for (CeedInt n = 0; n < S; n++) {
          size_t li = r->offsets
                      ? (size_t)r->offsets[(size_t)el * S + n] + (size_t)c * r->compstride
                      : (size_t)n * r->strides[0] + (size_t)c * r->strides[1] +
                        (size_t)el * r->strides[2];
#pragma omp atomic
          l[li] += e[((size_t)el * C + c) * S + n];
        }
    return;
  }
#endif
  for (CeedInt el = 0; el < r->nelem; el++)
    for (CeedInt c = 0; c < C; c++)
      for (CeedInt n = 0; n < S; n++) {
        size_t li = r->offsets
                    ? (size_t)r->offsets[(size_t)el * S + n] + (size_t)c * r->compstride
                    : (size_t)n * r->strides[0] + (size_t)c * r->strides[1] +
                      (size_t)el * r->strides[2];
        l[li] += e[((size_t)el * C + c) * S + n];
      }
}
int CeedElemRestrictionApply(CeedElemRestriction r, CeedTransposeMode tmode,
                             CeedVector u, CeedVector ru, CeedRequest *request) {
  (void)request;
  vec_ensure(u); vec_ensure(ru);
  if (tmode == CEED_NOTRANSPOSE) rstr_gather(r, u->array, ru->array);
  else rstr_scatter_add(r, u->array, ru->array);
  return 0;
}
int CeedElemRestrictionGetMultiplicity(CeedElemRestriction r, CeedVector mult) {
  size_t ne = (size_t)r->nelem * r->elemsize * r->ncomp;
  CeedScalar *ones = malloc(sizeof(CeedScalar) * (ne ? ne : 1));
  for (size_t i = 0; i < ne; i++) ones[i] = 1.;
  CHK(CeedVectorSetValue(mult, 0.));
  rstr_scatter_add(r, ones, mult->array);
  free(ones);
  return 0;
}
int CeedElemRestrictionDestroy(CeedElemRestriction *rstr) {
  if (!rstr || !*rstr) return 0;
  CeedElemRestriction r = *rstr;
  *rstr = NULL;
  if (r == CEED_ELEMRESTRICTION_NONE) return 0;
  if (--r->refcount > 0) return 0;
  free(r->offsets);
  ceed_unref(r->ceed);
  free(r);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Quadrature rules and Lagrange tables                    */
/* ------------------------------------------------------------------------- */
/* Legendre P_n(x) and P_{n-1}(x) by the three-term recurrence. */
static void legendre_pair(int n, double x, double *pn, double *pnm1) {
  double p0 = 1., p1 = x;
  if (n == 0) { *pn = 1.; *pnm1 = 0.; return; }
  for (int j = 2; j <= n; j++) {
    double p2 = ((2. * j - 1.) * x * p1 - (j - 1.) * p0) / j;
    p0 = p1; p1 = p2;
  }
  *pn = p1; *pnm1 = p0;
}
int CeedGaussQuadrature(CeedInt Q, CeedScalar *qref1d, CeedScalar *qweight1d) {
  /* Newton on P_Q from Chebyshev guesses; w = 2 / ((1-x^2) P_Q'(x)^2). */
  for (int i = 0; i <= (Q - 1) / 2; i++) {
    double x = cos(M_PI * (2. * i + 1.) / (2. * Q)), pq, pqm1, dp = 1.;
    for (int it = 0; it < 100; it++) {
      legendre_pair(Q, x, &pq, &pqm1);
      dp = Q * (x * pq - pqm1) / (x * x - 1.);
      x -= pq / dp;
      if (it > 0 && fabs(pq) <= 10 * 2.220446049250313e-16) break;
    }
    legendre_pair(Q, x, &pq, &pqm1);
    dp = Q * (x * pq - pqm1) / (x * x - 1.);
    double w = 2. / ((1. - x * x) * dp * dp);
    qref1d[i] = -x; qref1d[Q - 1 - i] = x;
    if (qweight1d) { qweight1d[i] = w; qweight1d[Q - 1 - i] = w; }
  }
  if (Q % 2) qref1d[Q / 2] = 0.;
  return 0;
}
int CeedLobattoQuadrature(CeedInt Q, CeedScalar *qref1d, CeedScalar *qweight1d) {
  /* Endpoints +-1 and the roots of P'_{Q-1}; w = 2 / (Q(Q-1) P_{Q-1}(x)^2). */
  if (Q < 2) return oracle_error("Lobatto rule needs at least 2 points");
  const int n = Q - 1;
  qref1d[0] = -1.; qref1d[Q - 1] = 1.;
  if (qweight1d) qweight1d[0] = qweight1d[Q - 1] = 2. / (Q * (Q - 1.));
  for (int i = 1; i <= (Q - 1) / 2; i++) {
    double x = cos(M_PI * i / (double)n), pn, pnm1;
    for (int it = 0; it < 100; it++) {
      legendre_pair(n, x, &pn, &pnm1);
      double dp = n * (x * pn - pnm1) / (x * x - 1.);
      double d2p = (2. * x * dp - n * (n + 1.) * pn) / (1. - x * x);
      x -= dp / d2p;
      if (it > 0 && fabs(dp) <= 10 * 2.220446049250313e-16) break;
    }
    legendre_pair(n, x, &pn, &pnm1);
    double w = 2. / (Q * (Q - 1.) * pn * pn);
    qref1d[i] = -x; qref1d[Q - 1 - i] = x;
    if (qweight1d) { qweight1d[i] = w; qweight1d[Q - 1 - i] = w; }
  }
  if (Q % 2) qref1d[Q / 2] = 0.;
  return 0;
}
/* Lagrange basis on `nodes` evaluated (value and derivative) at `x`:
 * barycentric-free direct products, O(P^2) per point, exact to a few ulp. */
static void lagrange_at(int P, const double *nodes, double x, double *val, double *der) {
  for (int j = 0; j < P; j++) {
    double v = 1., d = 0.;
    for (int m = 0; m < P; m++) {
      if (m == j) continue;
      double inv = 1. / (nodes[j] - nodes[m]);
      d = d * (x - nodes[m]) * inv + v * inv;
      v *= (x - nodes[m]) * inv;
    }
    val[j] = v; der[j] = d;
  }
}
int CeedBasisCreateTensorH1Lagrange(Ceed ceed, CeedInt dim, CeedInt ncomp,
                                    CeedInt P, CeedInt Q, CeedQuadMode qmode,
                                    CeedBasis *basis) {
  if (dim != 3) return oracle_error("oracle basis supports dim 3 only");
  if (P < 2 || Q < 1) return oracle_error("bad basis sizes P=%d Q=%d", P, Q);
  CeedBasis b = calloc(1, sizeof *b);
  b->ceed = ceed; ceed_ref(ceed);
  b->refcount = 1;
  b->dim = dim; b->ncomp = ncomp; b->P1d = P; b->Q1d = Q;
  b->interp1d = calloc((size_t)P * Q, sizeof(double));
  b->grad1d = calloc((size_t)P * Q, sizeof(double));
  b->qref1d = calloc((size_t)Q, sizeof(double));
  b->qweight1d = calloc((size_t)Q, sizeof(double));
  double *nodes = calloc((size_t)P, sizeof(double));
  CHK(CeedLobattoQuadrature(P, nodes, NULL));
  if (qmode == CEED_GAUSS) CHK(CeedGaussQuadrature(Q, b->qref1d, b->qweight1d));
  else CHK(CeedLobattoQuadrature(Q, b->qref1d, b->qweight1d));
  for (int q = 0; q < Q; q++)
    lagrange_at(P, nodes, b->qref1d[q], &b->interp1d[q * P], &b->grad1d[q * P]);
  free(nodes);
  *basis = b;
  return 0;
}
int CeedBasisGetNumQuadraturePoints(CeedBasis b, CeedInt *Q) {
  *Q = b->Q1d * b->Q1d * b->Q1d; return 0;
}
int CeedBasisGetNumNodes(CeedBasis b, CeedInt *P) {
  *P = b->P1d * b->P1d * b->P1d; return 0;
}
int CeedBasisGetInterp1D(CeedBasis b, const CeedScalar **t) { *t = b->interp1d; return 0; }
int CeedBasisGetGrad1D(CeedBasis b, const CeedScalar **t) { *t = b->grad1d; return 0; }
int CeedBasisGetQWeights1D(CeedBasis b, const CeedScalar **t) { *t = b->qweight1d; return 0; }
int CeedBasisDestroy(CeedBasis *basis) {
  if (!basis || !*basis) return 0;
  CeedBasis b = *basis;
  *basis = NULL;
  if (b == CEED_BASIS_COLLOCATED) return 0;
  if (--b->refcount > 0) return 0;
  free(b->interp1d); free(b->grad1d); free(b->qref1d); free(b->qweight1d);
  ceed_unref(b->ceed);
  free(b);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Tensor contraction and basis apply                       */
/* ------------------------------------------------------------------------- */
/* v[a][j][c] (+)= sum_b t[j][b] u[a][b][c];  t is J x B (row-major) or, when
 * `ttrans`, stored B x J and read transposed. */
static void contract(int A, int B, int C, int J, const double *t, int ttrans,
                     int add, const double *u, double *v) {
  if (!add) memset(v, 0, sizeof(double) * (size_t)A * J * C);
  for (int a = 0; a < A; a++)
    for (int j = 0; j < J; j++)
      for (int b = 0; b < B; b++) {
        double tjb = ttrans ? t[b * J + j] : t[j * B + b];
        for (int c = 0; c < C; c++)
          v[((size_t)a * J + j) * C + c] += tjb * u[((size_t)a * B + b) * C + c];
      }
}
static int ipow(int b, int e) { int r = 1; while (e-- > 0) r *= b; return r; }

/* One element.  NOTRANSPOSE: u[ncomp][P^3] -> v[ncomp][Q^3] (INTERP) or
 * v[dim][ncomp][Q^3] (GRAD).  TRANSPOSE: the reverse, OVERWRITING u-side out. */
static void basis_apply_elem(CeedBasis bs, CeedTransposeMode tmode, CeedEvalMode emode,
                             const double *in, double *out, double *tmp0, double *tmp1) {
  const int dim = bs->dim, nc = bs->ncomp, P = bs->P1d, Q = bs->Q1d;
  const int tr = tmode == CEED_TRANSPOSE;
  const int Bdim = tr ? Q : P, Jdim = tr ? P : Q; /* contracted / produced */
  const int nout = nc * ipow(Jdim, dim);
  if (emode == CEED_EVAL_INTERP) {
    const double *src = in;
    int pre = nc * ipow(Bdim, dim - 1), post = 1;
    for (int d = 0; d < dim; d++) {
      double *dst = d == dim - 1 ? out : (d % 2 ? tmp1 : tmp0);
      contract(pre, Bdim, post, Jdim, bs->interp1d, tr, 0, src, dst);
      src = dst; pre /= Bdim; post *= Jdim;
    }
  } else if (emode == CEED_EVAL_GRAD) {
    const int nqc = nc * ipow(Q, dim);
    if (tr) memset(out, 0, sizeof(double) * (size_t)nout);
    for (int p = 0; p < dim; p++) {
      const double *src = tr ? in + (size_t)p * nqc : in;
      int pre = nc * ipow(Bdim, dim - 1), post = 1;
      for (int d = 0; d < dim; d++) {
        const double *t = d == p ? bs->grad1d : bs->interp1d;
        int last = d == dim - 1;
        double *dst = last ? (tr ? out : out + (size_t)p * nqc) : (d % 2 ? tmp1 : tmp0);
        contract(pre, Bdim, post, Jdim, t, tr, last && tr, src, dst);
        src = dst; pre /= Bdim; post *= Jdim;
      }
    }
  }
}
static void basis_weights(CeedBasis bs, double *w) {
  const int Q = bs->Q1d;
  for (int k = 0; k < Q; k++)
    for (int j = 0; j < Q; j++)
      for (int i = 0; i < Q; i++)
        w[(k * Q + j) * Q + i] = bs->qweight1d[i] * bs->qweight1d[j] * bs->qweight1d[k];
}
static size_t basis_tmp_len(CeedBasis bs) {
  int m = bs->P1d > bs->Q1d ? bs->P1d : bs->Q1d;
  return (size_t)bs->ncomp * ipow(m, bs->dim);
}
int CeedBasisApply(CeedBasis bs, CeedInt nelem, CeedTransposeMode tmode,
                   CeedEvalMode emode, CeedVector u, CeedVector v) {
  const int dim = bs->dim, nc = bs->ncomp;
  const size_t np = (size_t)nc * ipow(bs->P1d, dim), nq = (size_t)nc * ipow(bs->Q1d, dim);
  const size_t qmul = emode == CEED_EVAL_GRAD ? (size_t)dim : 1;
  vec_ensure(v);
  if (emode == CEED_EVAL_WEIGHT) {
    size_t Q3 = (size_t)ipow(bs->Q1d, dim);
    for (CeedInt e = 0; e < nelem; e++) basis_weights(bs, v->array + e * Q3);
    return 0;
  }
  vec_ensure(u);
  double *t0 = malloc(sizeof(double) * basis_tmp_len(bs));
  double *t1 = malloc(sizeof(double) * basis_tmp_len(bs));
  for (CeedInt e = 0; e < nelem; e++) {
    if (tmode == CEED_NOTRANSPOSE)
      basis_apply_elem(bs, tmode, emode, u->array + e * np, v->array + e * nq * qmul, t0, t1);
    else
      basis_apply_elem(bs, tmode, emode, u->array + e * nq * qmul, v->array + e * np, t0, t1);
  }
  free(t0); free(t1);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* CeedQFunction                                                              */
/* ------------------------------------------------------------------------- */
CEED_EXTERN CeedQFunctionUser OracleGetQFunction(const char *name);

int CeedQFunctionCreateInterior(Ceed ceed, CeedInt vlength, CeedQFunctionUser f,
                                const char *source, CeedQFunction *qf) {
  (void)vlength;
  CeedQFunction q = calloc(1, sizeof *q);
  q->ceed = ceed; ceed_ref(ceed);
  q->refcount = 1;
  snprintf(q->source, sizeof q->source, "%s", source ? source : "");
  q->f = f;
  if (!q->f) { /* resolve the restated QFunction by the name after ':' */
    const char *colon = source ? strrchr(source, ':') : NULL;
    q->f = OracleGetQFunction(colon ? colon + 1 : (source ? source : ""));
    if (!q->f) {
      free(q);
      return oracle_error("no QFunction callback and unknown name '%s'",
                          source ? source : "(null)");
    }
  }
  *qf = q;
  return 0;
}
int CeedQFunctionCreateIdentity(Ceed ceed, CeedInt size, CeedEvalMode inmode,
                                CeedEvalMode outmode, CeedQFunction *qf) {
  CeedQFunction q = calloc(1, sizeof *q);
  q->ceed = ceed; ceed_ref(ceed);
  q->refcount = 1;
  q->identity = 1; q->identity_size = size;
  snprintf(q->source, sizeof q->source, "Identity");
  *qf = q;
  CHK(CeedQFunctionAddInput(q, "input", size, inmode));
  CHK(CeedQFunctionAddOutput(q, "output", size, outmode));
  return 0;
}
static int qf_add(QFField *arr, int *n, const char *name, CeedInt size, CeedEvalMode em) {
  if (*n >= ORACLE_MAX_FIELDS) return oracle_error("too many QFunction fields");
  snprintf(arr[*n].name, sizeof arr[*n].name, "%s", name);
  arr[*n].size = size; arr[*n].emode = em;
  (*n)++;
  return 0;
}
int CeedQFunctionAddInput(CeedQFunction qf, const char *name, CeedInt size, CeedEvalMode em) {
  return qf_add(qf->in, &qf->nin, name, size, em);
}
int CeedQFunctionAddOutput(CeedQFunction qf, const char *name, CeedInt size, CeedEvalMode em) {
  if (em == CEED_EVAL_WEIGHT) return oracle_error("WEIGHT is not an output mode");
  return qf_add(qf->out, &qf->nout, name, size, em);
}
int CeedQFunctionSetContext(CeedQFunction qf, void *ctx, size_t ctxsize) {
  qf->ctx = ctx; qf->ctxsize = ctxsize;
  return 0;
}
int CeedQFunctionDestroy(CeedQFunction *qf) {
  if (!qf || !*qf) return 0;
  CeedQFunction q = *qf;
  *qf = NULL;
  if (q == CEED_QFUNCTION_NONE) return 0;
  if (--q->refcount > 0) return 0;
  ceed_unref(q->ceed);
  free(q);
  return 0;
}
static int qf_call(CeedQFunction qf, CeedInt Q, const CeedScalar *const *in,
                   CeedScalar *const *out) {
  if (qf->identity) {
    memcpy(out[0], in[0], sizeof(CeedScalar) * (size_t)qf->identity_size * Q);
    return 0;
  }
  return qf->f(qf->ctx, Q, in, out);
}

/* ------------------------------------------------------------------------- */
/* CeedOperator                                             */
/* ------------------------------------------------------------------------- */
int CeedOperatorCreate(Ceed ceed, CeedQFunction qf, CeedQFunction dqf,
                       CeedQFunction dqfT, CeedOperator *op) {
  (void)dqf; (void)dqfT;
  CeedOperator o = calloc(1, sizeof *o);
  o->ceed = ceed; ceed_ref(ceed);
  o->refcount = 1;
  o->qf = qf; qf->refcount++;
  *op = o;
  return 0;
}
int CeedCompositeOperatorCreate(Ceed ceed, CeedOperator *op) {
  CeedOperator o = calloc(1, sizeof *o);
  o->ceed = ceed; ceed_ref(ceed);
  o->refcount = 1; o->composite = 1;
  *op = o;
  return 0;
}
int CeedCompositeOperatorAddSub(CeedOperator comp, CeedOperator sub) {
  if (!comp->composite) return oracle_error("not a composite operator");
  if (comp->nsub >= ORACLE_MAX_FIELDS) return oracle_error("too many sub-operators");
  comp->sub[comp->nsub++] = sub; sub->refcount++;
  return 0;
}
int CeedOperatorSetField(CeedOperator op, const char *name, CeedElemRestriction r,
                         CeedBasis b, CeedVector v) {
  if (op->composite) return oracle_error("cannot set a field on a composite operator");
  OpField *f = NULL;
  for (int i = 0; i < op->qf->nin && !f; i++)
    if (!strcmp(op->qf->in[i].name, name)) f = &op->in[i];
  for (int i = 0; i < op->qf->nout && !f; i++)
    if (!strcmp(op->qf->out[i].name, name)) f = &op->out[i];
  if (!f) return oracle_error("QFunction has no field named '%s'", name);
  f->set = 1; f->rstr = r; f->basis = b; f->vec = v;
  if (r != CEED_ELEMRESTRICTION_NONE) r->refcount++;
  if (b != CEED_BASIS_COLLOCATED) b->refcount++;
  if (v != CEED_VECTOR_ACTIVE && v != CEED_VECTOR_NONE) v->refcount++;
  return 0;
}
int CeedOperatorDestroy(CeedOperator *op) {
  if (!op || !*op) return 0;
  CeedOperator o = *op;
  *op = NULL;
  if (--o->refcount > 0) return 0;
  if (o->composite) {
    for (int i = 0; i < o->nsub; i++) { CeedOperator s = o->sub[i]; CeedOperatorDestroy(&s); }
  } else {
    for (int k = 0; k < 2; k++) {
      OpField *arr = k ? o->out : o->in;
      int n = k ? o->qf->nout : o->qf->nin;
      for (int i = 0; i < n; i++) {
        if (!arr[i].set) continue;
        CeedElemRestriction r = arr[i].rstr; CeedBasis b = arr[i].basis; CeedVector v = arr[i].vec;
        CeedElemRestrictionDestroy(&r); CeedBasisDestroy(&b); CeedVectorDestroy(&v);
      }
    }
    CeedQFunction q = o->qf; CeedQFunctionDestroy(&q);
  }
  free(o->mask_in); free(o->mask_out); free(o->priority);
  CeedVectorDestroy(&o->scale);
  ceed_unref(o->ceed);
  free(o);
  return 0;
}

typedef struct { CeedInt nelem, nqpts; } OpDims;

static int op_dims(CeedOperator op, OpDims *d) {
  d->nelem = -1; d->nqpts = -1;
  for (int k = 0; k < 2; k++) {
    OpField *arr = k ? op->out : op->in;
    QFField *qa = k ? op->qf->out : op->qf->in;
    int n = k ? op->qf->nout : op->qf->nin;
    for (int i = 0; i < n; i++) {
      if (!arr[i].set) return oracle_error("operator field '%s' not set", qa[i].name);
      if (arr[i].rstr != CEED_ELEMRESTRICTION_NONE) {
        if (d->nelem >= 0 && d->nelem != arr[i].rstr->nelem)
          return oracle_error("restrictions disagree on the element count");
        d->nelem = arr[i].rstr->nelem;
      }
      if (arr[i].basis != CEED_BASIS_COLLOCATED) {
        CeedInt q; CeedBasisGetNumQuadraturePoints(arr[i].basis, &q);
        if (d->nqpts >= 0 && d->nqpts != q)
          return oracle_error("bases disagree on the quadrature point count");
        d->nqpts = q;
      }
    }
  }
  if (d->nqpts < 0) /* all collocated: points are the nodes */
    for (int i = 0; i < op->qf->nin; i++)
      if (op->in[i].rstr != CEED_ELEMRESTRICTION_NONE) d->nqpts = op->in[i].rstr->elemsize;
  if (d->nelem < 0 || d->nqpts < 0) return oracle_error("cannot size the operator");
  return 0;
}

static int op_apply_single(CeedOperator op, CeedVector in, CeedVector out, int add) {
  CeedQFunction qf = op->qf;
  OpDims dm;
  CHK(op_dims(op, &dm));
  const CeedInt nelem = dm.nelem, Q = dm.nqpts;
  const int nin = qf->nin, nout = qf->nout;
  CeedScalar *ein[ORACLE_MAX_FIELDS] = {0}, *eout[ORACLE_MAX_FIELDS] = {0};
  CeedVector vin[ORACLE_MAX_FIELDS] = {0}, vout[ORACLE_MAX_FIELDS] = {0};
  size_t esz_in[ORACLE_MAX_FIELDS] = {0}, esz_out[ORACLE_MAX_FIELDS] = {0};
  double *wts = NULL;

  /* 1. outputs are overwritten: zero them (the active one unless ApplyAdd). */
  for (int i = 0; i < nout; i++) {
    vout[i] = op->out[i].vec == CEED_VECTOR_ACTIVE ? out : op->out[i].vec;
    if (!vout[i] || vout[i] == CEED_VECTOR_NONE)
      return oracle_error("output field '%s' has no vector", qf->out[i].name);
    vec_ensure(vout[i]);
    if (!(add && op->out[i].vec == CEED_VECTOR_ACTIVE))
      memset(vout[i]->array, 0, sizeof(CeedScalar) * (size_t)vout[i]->length);
    CeedElemRestriction r = op->out[i].rstr;
    esz_out[i] = (size_t)r->elemsize * r->ncomp;
    eout[i] = calloc(esz_out[i] * nelem + 1, sizeof(CeedScalar));
  }
  /* 2. gather the inputs to E-vectors. */
  for (int i = 0; i < nin; i++) {
    if (qf->in[i].emode == CEED_EVAL_WEIGHT) {
      if (!wts) { wts = malloc(sizeof(double) * (size_t)Q); basis_weights(op->in[i].basis, wts); }
      continue;
    }
    vin[i] = op->in[i].vec == CEED_VECTOR_ACTIVE ? in : op->in[i].vec;
    if (!vin[i] || vin[i] == CEED_VECTOR_NONE)
      return oracle_error("input field '%s' has no vector", qf->in[i].name);
    vec_ensure(vin[i]);
    CeedElemRestriction r = op->in[i].rstr;
    if (vin[i]->length < r->lsize)
      return oracle_error("input field '%s': vector shorter than the restriction's L-size",
                          qf->in[i].name);
    esz_in[i] = (size_t)r->elemsize * r->ncomp;
    ein[i] = malloc(sizeof(CeedScalar) * (esz_in[i] * nelem + 1));
    rstr_gather(r, vin[i]->array, ein[i]);
  }
  /* 3. element loop: basis -> QFunction -> basis^T. */
  int fail = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(g_nthreads) if (g_nthreads > 1)
#endif
  {
    double *qin[ORACLE_MAX_FIELDS] = {0}, *qout[ORACLE_MAX_FIELDS] = {0};
    double *t0 = NULL, *t1 = NULL;
    size_t tl = 1;
    for (int i = 0; i < nin; i++) {
      if (op->in[i].basis != CEED_BASIS_COLLOCATED && qf->in[i].emode != CEED_EVAL_WEIGHT) {
        qin[i] = malloc(sizeof(double) * (size_t)qf->in[i].size * Q);
        if (basis_tmp_len(op->in[i].basis) > tl) tl = basis_tmp_len(op->in[i].basis);
      }
    }
    for (int i = 0; i < nout; i++)
      if (qf->out[i].emode != CEED_EVAL_NONE) {
        qout[i] = malloc(sizeof(double) * (size_t)qf->out[i].size * Q);
        if (basis_tmp_len(op->out[i].basis) > tl) tl = basis_tmp_len(op->out[i].basis);
      }
    t0 = malloc(sizeof(double) * tl); t1 = malloc(sizeof(double) * tl);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (CeedInt e = 0; e < nelem; e++) {
      const CeedScalar *pin[ORACLE_MAX_FIELDS];
      CeedScalar *pout[ORACLE_MAX_FIELDS];
      for (int i = 0; i < nin; i++) {
        switch (qf->in[i].emode) {
        case CEED_EVAL_WEIGHT: pin[i] = wts; break;
        case CEED_EVAL_NONE: pin[i] = ein[i] + e * esz_in[i]; break;
        default:
          basis_apply_elem(op->in[i].basis, CEED_NOTRANSPOSE, qf->in[i].emode,
                           ein[i] + e * esz_in[i], qin[i], t0, t1);
          pin[i] = qin[i];
        }
      }
      for (int i = 0; i < nout; i++)
        pout[i] = qf->out[i].emode == CEED_EVAL_NONE ? eout[i] + e * esz_out[i] : qout[i];
      if (qf_call(qf, Q, pin, pout)) fail = 1;
      for (int i = 0; i < nout; i++)
        if (qf->out[i].emode != CEED_EVAL_NONE)
          basis_apply_elem(op->out[i].basis, CEED_TRANSPOSE, qf->out[i].emode, qout[i],
                           eout[i] + e * esz_out[i], t0, t1);
    }
    for (int i = 0; i < ORACLE_MAX_FIELDS; i++) { free(qin[i]); free(qout[i]); }
    free(t0); free(t1);
  }
  /* 4. scatter-add the outputs (element order, deterministic). */
  for (int i = 0; i < nout; i++) rstr_scatter_add(op->out[i].rstr, eout[i], vout[i]->array);
  for (int i = 0; i < ORACLE_MAX_FIELDS; i++) { free(ein[i]); free(eout[i]); }
  free(wts);
  return fail ? oracle_error("QFunction returned an error") : 0;
}

/* The harness extensions, restated on the host around the plain apply:
 * masked inputs read as zero, the fine side of a transfer scaled by the
 * multiplicity reciprocal, masked output rows dropped. */
static int op_apply_ext(CeedOperator op, CeedVector in, CeedVector out, int add) {
  const int has_ext = (op->mask_mode && (op->mask_in || op->mask_out)) || op->scale;
  if (!has_ext) return op_apply_single(op, in, out, add);
  const int prolong = op->qf->identity && op->qf->in[0].emode == CEED_EVAL_INTERP;
  const int restrict_ = op->qf->identity && !prolong;
  CeedVector xin, yout;
  vec_ensure(in); vec_ensure(out);
  CHK(CeedVectorCreate(op->ceed, in->length, &xin));
  CHK(CeedVectorCreate(op->ceed, out->length, &yout));
  CHK(CeedVectorSetArray(xin, CEED_MEM_HOST, CEED_COPY_VALUES, in->array));
  vec_ensure(yout);
  if ((op->mask_mode & 1) && op->mask_in)
    for (CeedInt i = 0; i < in->length && i < op->mask_in_len; i++)
      if (op->mask_in[i]) xin->array[i] = 0.;
  if (op->scale && restrict_) {
    vec_ensure(op->scale);
    for (CeedInt i = 0; i < in->length; i++) xin->array[i] *= op->scale->array[i];
  }
  CHK(op_apply_single(op, xin, yout, 0));
  if (op->scale && prolong) {
    vec_ensure(op->scale);
    for (CeedInt i = 0; i < out->length; i++) yout->array[i] *= op->scale->array[i];
  }
  const unsigned char *mo = op->mask_out ? op->mask_out : op->mask_in;
  const CeedInt mol = op->mask_out ? op->mask_out_len : op->mask_in_len;
  if ((op->mask_mode & 2) && mo)
    for (CeedInt i = 0; i < out->length && i < mol; i++)
      if (mo[i]) yout->array[i] = 0.;
  for (CeedInt i = 0; i < out->length; i++) out->array[i] = (add ? out->array[i] : 0.) + yout->array[i];
  CeedVectorDestroy(&xin); CeedVectorDestroy(&yout);
  return 0;
}

int CeedOperatorApply(CeedOperator op, CeedVector in, CeedVector out, CeedRequest *request) {
  (void)request;
  if (op->composite) {
    vec_ensure(out);
    memset(out->array, 0, sizeof(CeedScalar) * (size_t)out->length);
    for (int i = 0; i < op->nsub; i++) CHK(op_apply_ext(op->sub[i], in, out, 1));
    return 0;
  }
  return op_apply_ext(op, in, out, 0);
}
int CeedOperatorApplyAdd(CeedOperator op, CeedVector in, CeedVector out, CeedRequest *request) {
  (void)request;
  if (op->composite) {
    for (int i = 0; i < op->nsub; i++) CHK(op_apply_ext(op->sub[i], in, out, 1));
    return 0;
  }
  return op_apply_ext(op, in, out, 1);
}

/* diag(B^T D B): D from the QFunction applied to unit inputs. */
int CeedOperatorLinearAssembleDiagonal(CeedOperator op, CeedVector assembled, CeedRequest *request) {
  (void)request;
  if (op->composite) return oracle_error("diagonal of a composite operator not supported");
  CeedQFunction qf = op->qf;
  OpDims dm;
  CHK(op_dims(op, &dm));
  const CeedInt nelem = dm.nelem, Q = dm.nqpts;
  int ai = -1, ao = -1;
  for (int i = 0; i < qf->nin; i++)
    if (op->in[i].vec == CEED_VECTOR_ACTIVE) { if (ai >= 0) return oracle_error("one active input expected"); ai = i; }
  for (int i = 0; i < qf->nout; i++)
    if (op->out[i].vec == CEED_VECTOR_ACTIVE) { if (ao >= 0) return oracle_error("one active output expected"); ao = i; }
  if (ai < 0 || ao < 0) return oracle_error("operator has no active field pair");
  CeedBasis bi = op->in[ai].basis, bo = op->out[ao].basis;
  CeedElemRestriction ri = op->in[ai].rstr, ro = op->out[ao].rstr;
  if (bi == CEED_BASIS_COLLOCATED || bo == CEED_BASIS_COLLOCATED || ri != ro)
    return oracle_error("diagonal assembly needs the same basis-backed restriction in and out");
  const int nc = ri->ncomp, P3 = ri->elemsize, dim = bi->dim;
  const CeedEvalMode emi = qf->in[ai].emode, emo = qf->out[ao].emode;
  const int ndi = emi == CEED_EVAL_GRAD ? dim : 1, ndo = emo == CEED_EVAL_GRAD ? dim : 1;
  const int sin = qf->in[ai].size, sout = qf->out[ao].size;
  if (sin != ndi * nc || sout != ndo * nc) return oracle_error("active field sizes do not match ncomp*dim");

  /* Dense single-component basis matrices M_d[q][n], d < nd. */
  struct CeedBasis_private b1 = *bi; b1.ncomp = 1;
  double *t0 = malloc(sizeof(double) * basis_tmp_len(bi)), *t1 = malloc(sizeof(double) * basis_tmp_len(bi));
  double *Mi = calloc((size_t)ndi * Q * P3, sizeof(double)), *Mo;
  double *unit = calloc((size_t)P3, sizeof(double)), *col = malloc(sizeof(double) * (size_t)dim * Q);
  for (int n = 0; n < P3; n++) {
    unit[n] = 1.;
    basis_apply_elem(&b1, CEED_NOTRANSPOSE, emi, unit, col, t0, t1);
    for (int d = 0; d < ndi; d++)
      for (int q = 0; q < Q; q++) Mi[((size_t)d * Q + q) * P3 + n] = col[d * Q + q];
    unit[n] = 0.;
  }
  if (bo == bi && emo == emi) Mo = Mi;
  else {
    struct CeedBasis_private b2 = *bo; b2.ncomp = 1;
    Mo = calloc((size_t)ndo * Q * P3, sizeof(double));
    for (int n = 0; n < P3; n++) {
      unit[n] = 1.;
      basis_apply_elem(&b2, CEED_NOTRANSPOSE, emo, unit, col, t0, t1);
      for (int d = 0; d < ndo; d++)
        for (int q = 0; q < Q; q++) Mo[((size_t)d * Q + q) * P3 + n] = col[d * Q + q];
      unit[n] = 0.;
    }
  }
  /* Passive inputs as E-vectors. */
  CeedScalar *ein[ORACLE_MAX_FIELDS] = {0};
  size_t esz[ORACLE_MAX_FIELDS] = {0};
  double *wts = NULL;
  for (int i = 0; i < qf->nin; i++) {
    if (i == ai) continue;
    if (qf->in[i].emode == CEED_EVAL_WEIGHT) { wts = malloc(sizeof(double) * Q); basis_weights(op->in[i].basis, wts); continue; }
    if (qf->in[i].emode != CEED_EVAL_NONE) return oracle_error("passive inputs must be EVAL_NONE for diagonal assembly");
    CeedVector v = op->in[i].vec; vec_ensure(v);
    esz[i] = (size_t)op->in[i].rstr->elemsize * op->in[i].rstr->ncomp;
    ein[i] = malloc(sizeof(CeedScalar) * (esz[i] * nelem + 1));
    rstr_gather(op->in[i].rstr, v->array, ein[i]);
  }
  double *uin = calloc((size_t)sin * Q, sizeof(double));
  double *D = malloc(sizeof(double) * (size_t)sout * sin * Q);
  double *qo[ORACLE_MAX_FIELDS] = {0};
  for (int i = 0; i < qf->nout; i++) qo[i] = malloc(sizeof(double) * (size_t)qf->out[i].size * Q);
  double *ediag = calloc((size_t)nelem * nc * P3 + 1, sizeof(double));
  for (CeedInt e = 0; e < nelem; e++) {
    const CeedScalar *pin[ORACLE_MAX_FIELDS];
    for (int i = 0; i < qf->nin; i++)
      pin[i] = i == ai ? uin : (qf->in[i].emode == CEED_EVAL_WEIGHT ? wts : ein[i] + e * esz[i]);
    for (int s = 0; s < sin; s++) {
      for (int q = 0; q < Q; q++) uin[(size_t)s * Q + q] = 1.;
      if (qf_call(qf, Q, pin, qo)) return oracle_error("QFunction returned an error");
      for (int o = 0; o < sout; o++)
        for (int q = 0; q < Q; q++) D[((size_t)o * sin + s) * Q + q] = qo[ao][(size_t)o * Q + q];
      for (int q = 0; q < Q; q++) uin[(size_t)s * Q + q] = 0.;
    }
    for (int c = 0; c < nc; c++)
      for (int n = 0; n < P3; n++) {
        double acc = 0.;
        for (int dou = 0; dou < ndo; dou++)
          for (int din = 0; din < ndi; din++) {
            const double *Dq = &D[((size_t)(dou * nc + c) * sin + (din * nc + c)) * Q];
            for (int q = 0; q < Q; q++)
              acc += Mo[((size_t)dou * Q + q) * P3 + n] * Dq[q] * Mi[((size_t)din * Q + q) * P3 + n];
          }
        ediag[((size_t)e * nc + c) * P3 + n] = acc;
      }
  }
  vec_ensure(assembled);
  memset(assembled->array, 0, sizeof(CeedScalar) * (size_t)assembled->length);
  rstr_scatter_add(ro, ediag, assembled->array);
  if ((op->mask_mode & 2) && op->mask_in)
    for (CeedInt i = 0; i < assembled->length && i < op->mask_in_len; i++)
      if (op->mask_in[i]) assembled->array[i] = 0.;
  for (int i = 0; i < ORACLE_MAX_FIELDS; i++) { free(ein[i]); free(qo[i]); }
  if (Mo != Mi) free(Mo);
  free(Mi); free(unit); free(col); free(t0); free(t1); free(wts); free(uin); free(D); free(ediag);
  return 0;
}

/* Extensions that only make sense on the device backend: accepted as no-ops
 * or restated on the host so one harness drives both libraries. */
int CeedXOperatorGetKernelName(CeedOperator op, const char **name) { (void)op; *name = "oracle"; return 0; }
int CeedXOperatorSetDirichletMaskMode(CeedOperator op, CeedMemType mtype, const unsigned char *mask_in,
                                      CeedInt lsize_in, const unsigned char *mask_out, CeedInt lsize_out,
                                      int mode) {
  CHK(vec_host_only(mtype));
  free(op->mask_in); free(op->mask_out);
  op->mask_in = op->mask_out = NULL; op->mask_mode = 0;
  if (mask_in) {
    op->mask_in = malloc((size_t)lsize_in + 1); memcpy(op->mask_in, mask_in, (size_t)lsize_in);
    op->mask_in_len = lsize_in;
  }
  if (mask_out) {
    op->mask_out = malloc((size_t)lsize_out + 1); memcpy(op->mask_out, mask_out, (size_t)lsize_out);
    op->mask_out_len = lsize_out;
  }
  if (mask_in || mask_out) op->mask_mode = mode ? mode : 3;
  return 0;
}
int CeedXOperatorSetDirichletMask(CeedOperator op, CeedMemType mtype, const unsigned char *mask, CeedInt lsize) {
  return CeedXOperatorSetDirichletMaskMode(op, mtype, mask, lsize, NULL, 0, 3);
}
/* Split-phase apply restated on the host: both phases evaluate the whole operator; phase 0 keeps
 * the priority entries (others zero), phase 1 fills in the remaining entries. */
int CeedXOperatorSetOverlapSplit(CeedOperator op, CeedInt n_leading_elems, const unsigned char *priority, CeedInt lsize) {
  (void)n_leading_elems;
  free(op->priority); op->priority = NULL; op->priority_len = 0;
  if (priority) {
    op->priority = malloc((size_t)lsize + 1); memcpy(op->priority, priority, (size_t)lsize);
    op->priority_len = lsize;
  }
  return 0;
}
int CeedXOperatorApplyPhase(CeedOperator op, CeedVector in, CeedVector out, int phase) {
  if (!op->priority) return oracle_error("CeedXOperatorSetOverlapSplit was not called");
  CeedVector tmp;
  vec_ensure(out);
  CHK(CeedVectorCreate(op->ceed, out->length, &tmp));
  CHK(CeedOperatorApply(op, in, tmp, CEED_REQUEST_IMMEDIATE));
  for (CeedInt i = 0; i < out->length; i++) {
    const int pr = i < op->priority_len && op->priority[i];
    if (phase == 0) out->array[i] = pr ? tmp->array[i] : 0.;
    else if (!pr) out->array[i] = tmp->array[i];
  }
  CeedVectorDestroy(&tmp);
  return 0;
}
int CeedXOperatorSetFineScale(CeedOperator op, CeedVector scale) {
  CeedVectorDestroy(&op->scale);
  if (scale && scale != CEED_VECTOR_NONE) { op->scale = scale; scale->refcount++; }
  return 0;
}
int CeedXVectorPointwiseMult(CeedVector w, CeedVector x, CeedVector y) {
  vec_ensure(w); vec_ensure(x); vec_ensure(y);
  for (CeedInt i = 0; i < w->length; i++) w->array[i] = x->array[i] * y->array[i];
  return 0;
}
int CeedXVectorAXPBY(CeedVector y, double a, CeedVector x, double b) {
  vec_ensure(x); vec_ensure(y);
  for (CeedInt i = 0; i < y->length; i++) y->array[i] = a * x->array[i] + (b == 0. ? 0. : b * y->array[i]);
  return 0;
}
int CeedXVectorWAXPBY(CeedVector w, double a, CeedVector x, double b, CeedVector y) {
  vec_ensure(x); vec_ensure(y); vec_ensure(w);
  if (x->length != w->length || y->length != w->length) return oracle_error("CeedXVectorWAXPBY: vector lengths differ");
  for (CeedInt i = 0; i < w->length; i++) w->array[i] = a * x->array[i] + b * y->array[i];
  return 0;
}
int CeedXVectorChebyshevStart(CeedVector x, CeedVector d, CeedVector r, CeedVector b, CeedVector t, CeedVector dinv,
                              double c1, int assign_x) {
  vec_ensure(x); vec_ensure(d); vec_ensure(r); vec_ensure(b); vec_ensure(dinv);
  const int have_t = t && t != CEED_VECTOR_NONE;
  if (have_t) vec_ensure(t);
  if (b == r || b == x || b == d) return oracle_error("CeedXVectorChebyshevStart: the right-hand side must be a vector of its own");
  for (CeedInt i = 0; i < x->length; i++) {
    double ri = b->array[i];
    if (have_t) ri -= t->array[i];
    r->array[i] = ri;
    const double di = c1 * dinv->array[i] * ri;
    d->array[i] = di;
    x->array[i] = assign_x ? di : x->array[i] + di;
  }
  return 0;
}
int CeedXVectorChebyshevStep(CeedVector x, CeedVector d, CeedVector r, CeedVector b, CeedVector t, CeedVector dinv,
                             double c1, double c2, int assign_x) {
  vec_ensure(x); vec_ensure(d); vec_ensure(b); vec_ensure(dinv);
  const int have_t = t && t != CEED_VECTOR_NONE, have_r = r && r != CEED_VECTOR_NONE;
  if (have_t) vec_ensure(t);
  if (have_r) vec_ensure(r);
  if (b == x || b == d || (have_r && b == r)) return oracle_error("CeedXVectorChebyshevStep: the right-hand side must be a vector of its own");
  for (CeedInt i = 0; i < x->length; i++) {
    double ri = b->array[i];
    if (have_t) ri -= t->array[i];
    if (have_r) r->array[i] = ri;
    double di = (c1 * dinv->array[i]) * ri;
    if (c2 != 0.) di = fma(c2, d->array[i], di);
    d->array[i] = di;
    x->array[i] = assign_x ? di : x->array[i] + di;
  }
  return 0;
}
int CeedXVectorChebyshevUpdate(CeedVector x, CeedVector d, CeedVector r, CeedVector t, CeedVector dinv,
                               double c1, double c2, int assign_x) {
  vec_ensure(x); vec_ensure(d); vec_ensure(r); vec_ensure(dinv);
  const int have_t = t && t != CEED_VECTOR_NONE;
  if (have_t) vec_ensure(t);
  for (CeedInt i = 0; i < x->length; i++) {
    if (have_t) r->array[i] -= t->array[i];
    const double di = c1 * dinv->array[i] * r->array[i] + (c2 == 0. ? 0. : c2 * d->array[i]);
    d->array[i] = di;
    x->array[i] = assign_x ? di : x->array[i] + di;
  }
  return 0;
}
int CeedXClockProbe(Ceed ceed, int spin_us, double *ghz) { (void)ceed; (void)spin_us; *ghz = 0.; return 0; }   /* (no device clock on the CPU) */
/* The fused forms of the product (include/ceed.h), restated as what they fuse: the apply, then its consumer. */
int CeedXOperatorApplyChebyshev(CeedOperator op, CeedVector in, CeedVector t, CeedVector x, CeedVector d, CeedVector r, CeedVector b,
                                CeedVector dinv, double c1, double c2, int assign_x) {
  int ierr = CeedOperatorApply(op, in, t, CEED_REQUEST_IMMEDIATE);
  if (ierr) return ierr;
  if (b && b != CEED_VECTOR_NONE) return CeedXVectorChebyshevStep(x, d, r, b, t, dinv, c1, c2, assign_x);
  return CeedXVectorChebyshevUpdate(x, d, r, t, dinv, c1, c2, assign_x);
}
int CeedXOperatorApplyResidual(CeedOperator op, CeedVector in, CeedVector t, CeedVector b, CeedVector w) {
  int ierr = CeedOperatorApply(op, in, t, CEED_REQUEST_IMMEDIATE);
  if (ierr) return ierr;
  return CeedXVectorWAXPBY(w, 1.0, b, -1.0, t);
}
int CeedXVectorDot(CeedVector x, CeedVector y, CeedVector weight, double *result) {
  vec_ensure(x); vec_ensure(y);
  double s = 0.;
  if (weight && weight != CEED_VECTOR_NONE) {
    vec_ensure(weight);
    for (CeedInt i = 0; i < x->length; i++) s += weight->array[i] * x->array[i] * y->array[i];
  } else {
    for (CeedInt i = 0; i < x->length; i++) s += x->array[i] * y->array[i];
  }
  *result = s;
  return 0;
}
int CeedXVectorDotTo(CeedVector x, CeedVector y, CeedVector weight, CeedVector scalars, CeedInt idx) {
  vec_ensure(scalars);
  if (idx < 0 || idx >= scalars->length) return oracle_error("CeedXVectorDotTo: index out of range");
  return CeedXVectorDot(x, y, weight, &scalars->array[idx]);
}
int CeedXScalarDivide(CeedVector scalars, CeedInt dst, CeedInt num, CeedInt den, double scale) {
  vec_ensure(scalars);
  const CeedInt n = scalars->length;
  if (dst < 0 || dst >= n || num < 0 || num >= n || den >= n) return oracle_error("CeedXScalarDivide: index out of range");
  const double d = den < 0 ? 1. : scalars->array[den];
  scalars->array[dst] = (den < 0 || d > 0.) ? scale * scalars->array[num] / d : 0.;
  return 0;
}
int CeedXVectorAXPBYScalars(CeedVector y, CeedVector scalars, CeedInt ia, double sa, CeedVector x, CeedInt ib, double sb) {
  vec_ensure(x); vec_ensure(y); vec_ensure(scalars);
  if (ia >= scalars->length || ib >= scalars->length || x == y || x->length != y->length) return oracle_error("CeedXVectorAXPBYScalars: bad arguments");
  const double a = sa * (ia < 0 ? 1. : scalars->array[ia]), b = sb * (ib < 0 ? 1. : scalars->array[ib]);
  for (CeedInt i = 0; i < y->length; i++) y->array[i] = a * x->array[i] + b * y->array[i];
  return 0;
}
int CeedXOperatorSetTiming(CeedOperator op, int enable) { (void)op; (void)enable; return 0; }
int CeedXOperatorGetTiming(CeedOperator op, double *ms, int64_t *launches) { (void)op; *ms = 0; *launches = 0; return 0; }
int CeedXOperatorGetLaunchInfo(CeedOperator op, int out[4]) { (void)op; out[0] = 1; out[1] = 1; out[2] = 0; out[3] = 0; return 0; }
/* Halo exchange: the oracle is a single-process CPU checker; the multi-rank tests exchange through torch.distributed
   (ceedpetscsolid_amd/halo.py, gloo).  Only the neighbour-free halo exists here. */
struct CeedXHalo_private { int nneigh; };
int CeedXCommGetUniqueId(Ceed ceed, char id[128]) { (void)ceed; (void)id; return oracle_error("no communicator in the CPU oracle"); }
int CeedXCommInit(Ceed ceed, int nranks, int rank, const char id[128]) { (void)ceed; (void)rank; (void)id; return nranks == 1 ? 0 : oracle_error("no communicator in the CPU oracle"); }
int CeedXCommDestroy(Ceed ceed) { (void)ceed; return 0; }
int CeedXCommGetSize(Ceed ceed, int *nranks, int *rank) { (void)ceed; if (nranks) *nranks = 0; if (rank) *rank = -1; return 0; }
int CeedXHaloCreate(Ceed ceed, CeedInt nneigh, const int *neigh_rank, const CeedInt *count, const CeedInt *const *index, CeedXHalo *halo) {
  (void)ceed; (void)neigh_rank; (void)count; (void)index;
  if (nneigh != 0) return oracle_error("the CPU oracle exchanges no halo (use ceedpetscsolid_amd.halo over gloo)");
  *halo = (CeedXHalo)calloc(1, sizeof(struct CeedXHalo_private));
  return 0;
}
int CeedXHaloStart(CeedXHalo halo, CeedVector y) { (void)halo; (void)y; return 0; }
int CeedXHaloFinish(CeedXHalo halo, CeedVector y) { (void)halo; (void)y; return 0; }
int CeedXHaloDestroy(CeedXHalo *halo) { if (halo && *halo) { free(*halo); *halo = NULL; } return 0; }
int CeedXOperatorApplyWithHalo(CeedOperator op, CeedVector in, CeedVector out, CeedXHalo halo) { (void)halo; return CeedOperatorApply(op, in, out, CEED_REQUEST_IMMEDIATE); }
int CeedXCommAllReduce(Ceed ceed, CeedVector v, CeedInt first, CeedInt n) { (void)ceed; (void)v; (void)first; (void)n; return 0; }   /* one process */

/* ---- assembled sparse operator (include/ceed.h, CeedXCsr*): plain CSR on the host ------------ */
struct CeedXCsr_private {
  CeedInt nrows, ncols, nnz, ncoo, n_unit;
  CeedInt *rowptr, *cols, *coo_slot, *unit_rows;
  double  *vals;
  struct CeedXCsr_private *src;        /* CeedXCsrSetSource */
  CeedInt *termptr, *term_slot;
  double  *term_w;
  int refs, fixed;                     /* fixed: values given to CeedXCsrCreateRect */
};
int CeedXCsrCreate(Ceed ceed, CeedInt nrows, const CeedInt *rowptr, const CeedInt *cols, CeedInt ncoo,
                   const CeedInt *coo_slot, CeedInt n_unit, const CeedInt *unit_rows, CeedXCsr *csr) {
  (void)ceed;
  if (nrows < 0 || ncoo < 0 || !rowptr) return oracle_error("CeedXCsrCreate: bad pattern");
  CeedXCsr A = calloc(1, sizeof *A);
  A->nrows = nrows; A->ncols = nrows; A->nnz = rowptr[nrows]; A->ncoo = ncoo; A->n_unit = n_unit; A->refs = 1;
  A->rowptr = malloc(sizeof(CeedInt) * (size_t)(nrows + 1)); memcpy(A->rowptr, rowptr, sizeof(CeedInt) * (size_t)(nrows + 1));
  A->cols = malloc(sizeof(CeedInt) * (size_t)(A->nnz + 1)); memcpy(A->cols, cols, sizeof(CeedInt) * (size_t)A->nnz);
  A->coo_slot = malloc(sizeof(CeedInt) * (size_t)(ncoo + 1)); memcpy(A->coo_slot, coo_slot, sizeof(CeedInt) * (size_t)ncoo);
  A->unit_rows = malloc(sizeof(CeedInt) * (size_t)(n_unit + 1)); if (n_unit) memcpy(A->unit_rows, unit_rows, sizeof(CeedInt) * (size_t)n_unit);
  A->vals = calloc((size_t)(A->nnz + 1), sizeof(double));
  for (CeedInt k = 0; k < ncoo; k++)
    if (coo_slot[k] >= A->nnz) { free(A); return oracle_error("CeedXCsrCreate: COO entry maps outside the pattern"); }
  *csr = A;
  return 0;
}
int CeedXCsrAssemble(CeedXCsr A, CeedVector coo_values) {
  vec_ensure(coo_values);
  if (coo_values->length < A->ncoo) return oracle_error("CeedXCsrAssemble: too few COO values");
  memset(A->vals, 0, sizeof(double) * (size_t)A->nnz);
  for (CeedInt k = 0; k < A->ncoo; k++)   /* ascending entry order per slot, as the device path */
    if (A->coo_slot[k] >= 0) A->vals[A->coo_slot[k]] += coo_values->array[k];
  for (CeedInt i = 0; i < A->n_unit; i++) {
    const CeedInt r = A->unit_rows[i];
    int found = 0;
    for (CeedInt k = A->rowptr[r]; k < A->rowptr[r + 1]; k++) if (A->cols[k] == r) { A->vals[k] = 1.; found = 1; }
    if (!found) return oracle_error("CeedXCsrAssemble: unit row %d has no diagonal entry", r);
  }
  return 0;
}
int CeedXCsrApply(CeedXCsr A, CeedVector x, CeedVector y) {
  vec_ensure(x); vec_ensure(y);
  if (x == y || x->length < A->ncols || y->length < A->nrows) return oracle_error("CeedXCsrApply: bad vectors");
  for (CeedInt r = 0; r < A->nrows; r++) {
    double a = 0.;
    for (CeedInt k = A->rowptr[r]; k < A->rowptr[r + 1]; k++) a += A->vals[k] * x->array[A->cols[k]];
    y->array[r] = a;
  }
  return 0;
}
int CeedXCsrGetDiagonal(CeedXCsr A, CeedVector d) {
  vec_ensure(d);
  for (CeedInt r = 0; r < A->nrows; r++) {
    d->array[r] = 0.;
    for (CeedInt k = A->rowptr[r]; k < A->rowptr[r + 1]; k++) if (A->cols[k] == r) d->array[r] = A->vals[k];
  }
  return 0;
}
/* the pieces of the aggregation hierarchy (include/ceed.h): plain loops */
int CeedXCsrCreateRect(Ceed ceed, CeedInt nrows, CeedInt ncols, const CeedInt *rowptr, const CeedInt *cols, const CeedScalar *vals,
                       CeedXCsr *csr) {
  (void)ceed;
  if (nrows < 0 || ncols < 0 || !rowptr) return oracle_error("CeedXCsrCreateRect: bad pattern");
  const CeedInt nnz = rowptr[nrows];
  for (CeedInt k = 0; k < nnz; k++) if (cols[k] < 0 || cols[k] >= ncols) return oracle_error("CeedXCsrCreateRect: column out of range");
  CeedXCsr A = calloc(1, sizeof *A);
  A->nrows = nrows; A->ncols = ncols; A->nnz = nnz; A->refs = 1;
  A->rowptr = malloc(sizeof(CeedInt) * (size_t)(nrows + 1)); memcpy(A->rowptr, rowptr, sizeof(CeedInt) * (size_t)(nrows + 1));
  A->cols = malloc(sizeof(CeedInt) * (size_t)(nnz + 1)); if (nnz) memcpy(A->cols, cols, sizeof(CeedInt) * (size_t)nnz);
  A->vals = calloc((size_t)(nnz + 1), sizeof(double));
  if (vals && nnz) memcpy(A->vals, vals, sizeof(double) * (size_t)nnz);
  A->fixed = vals != NULL;
  *csr = A;
  return 0;
}
typedef struct { CeedInt col, slot; double w; } ProdTerm;
static int prodterm_cmp(const void *x, const void *y) {
  const ProdTerm *a = x, *b = y;
  if (a->col != b->col) return a->col < b->col ? -1 : 1;
  return a->slot < b->slot ? -1 : (a->slot > b->slot);
}
/* C = left * right with one fixed and one variable operand (include/ceed.h): row-wise, terms of an entry by slot */
int CeedXCsrCreateProduct(CeedXCsr Lm, CeedXCsr Rm, int variable, int dense, CeedXCsr *csr) {
  if (!Lm || !Rm || Lm == Rm || (variable != 0 && variable != 1) || Lm->ncols != Rm->nrows) return oracle_error("CeedXCsrCreateProduct: bad operands");
  CeedXCsr V = variable == 0 ? Lm : Rm, F = variable == 0 ? Rm : Lm;
  if (!F->fixed) return oracle_error("CeedXCsrCreateProduct: the fixed operand must carry values from CeedXCsrCreateRect");
  const CeedInt nrows = Lm->nrows, ncols = Rm->ncols;
  if (dense && nrows != ncols) return oracle_error("CeedXCsrCreateProduct: a dense result must be square");
  size_t cap_row = 1024, cap_e = 1024, cap_t = 4096, ne = 0, nt = 0;
  ProdTerm *row = malloc(cap_row * sizeof *row);
  CeedInt *rp = calloc((size_t)nrows + 1, sizeof *rp), *cl = malloc(cap_e * sizeof *cl), *tp = malloc((cap_e + 1) * sizeof *tp);
  CeedInt *ts = malloc(cap_t * sizeof *ts);
  double *tw = malloc(cap_t * sizeof *tw);
  tp[0] = 0;
  for (CeedInt i = 0; i < nrows; i++) {
    size_t nr = 0;
    for (CeedInt a = Lm->rowptr[i]; a < Lm->rowptr[i + 1]; a++) {
      const CeedInt j = Lm->cols[a];
      for (CeedInt b = Rm->rowptr[j]; b < Rm->rowptr[j + 1]; b++) {
        if (nr == cap_row) { cap_row *= 2; row = realloc(row, cap_row * sizeof *row); }
        row[nr].col = Rm->cols[b];
        row[nr].slot = variable == 0 ? a : b;
        row[nr].w = variable == 0 ? F->vals[b] : F->vals[a];
        nr++;
      }
    }
    qsort(row, nr, sizeof *row, prodterm_cmp);
    size_t k = 0;
    for (CeedInt c = 0; dense ? c < ncols : k < nr; c++) {
      if (!dense) c = row[k].col;
      while (k < nr && row[k].col == c) {
        if (nt == cap_t) { cap_t *= 2; ts = realloc(ts, cap_t * sizeof *ts); tw = realloc(tw, cap_t * sizeof *tw); }
        ts[nt] = row[k].slot; tw[nt] = row[k].w; nt++; k++;
      }
      if (ne == cap_e) { cap_e *= 2; cl = realloc(cl, cap_e * sizeof *cl); tp = realloc(tp, (cap_e + 1) * sizeof *tp); }
      cl[ne] = c; ne++; tp[ne] = (CeedInt)nt;
    }
    rp[i + 1] = (CeedInt)ne;
  }
  free(row);
  CeedXCsr A = calloc(1, sizeof *A);
  A->nrows = nrows; A->ncols = ncols; A->nnz = (CeedInt)ne; A->refs = 1;
  A->rowptr = rp; A->cols = cl; A->termptr = tp; A->term_slot = ts; A->term_w = tw;
  A->vals = calloc(ne + 1, sizeof(double));
  A->src = V; V->refs++;
  *csr = A;
  return 0;
}
int CeedXCsrGetPattern(CeedXCsr A, CeedInt *nrows, CeedInt *ncols, CeedInt *nnz, const CeedInt **rowptr, const CeedInt **cols) {
  if (nrows) *nrows = A->nrows;
  if (ncols) *ncols = A->ncols;
  if (nnz) *nnz = A->nnz;
  if (rowptr) *rowptr = A->rowptr;
  if (cols) *cols = A->cols;
  return 0;
}
int CeedXCsrUpdate(CeedXCsr A) {
  if (!A->src) return oracle_error("CeedXCsrUpdate: not a product");
  for (CeedInt s = 0; s < A->nnz; s++) {
    double a = 0.;
    for (CeedInt k = A->termptr[s]; k < A->termptr[s + 1]; k++) a += A->term_w[k] * A->src->vals[A->term_slot[k]];
    A->vals[s] = a;
  }
  return 0;
}
int CeedXCsrGetValues(CeedXCsr A, CeedVector v) {
  vec_ensure(v);
  if (v->length < A->nnz) return oracle_error("CeedXCsrGetValues: vector too short");
  memcpy(v->array, A->vals, sizeof(double) * (size_t)A->nnz);
  return 0;
}
/* Cholesky A = L L^T, then A^-1 = L^-T L^-1 (a different algorithm from the device's Gauss-Jordan on purpose) */
int CeedXCsrInvertDenseSPD(CeedXCsr A) {
  const CeedInt n = A->nrows;
  if (A->ncols != n || (long long)A->nnz != (long long)n * n) return oracle_error("CeedXCsrInvertDenseSPD: the pattern is not full");
  for (CeedInt r = 0; r < n; r++) for (CeedInt c = 0; c < n; c++) if (A->cols[(size_t)r * n + c] != c) return oracle_error("CeedXCsrInvertDenseSPD: columns not in order");
  double *a = A->vals, *L = calloc((size_t)n * n + 1, sizeof(double)), *X = calloc((size_t)n * n + 1, sizeof(double));
  for (CeedInt j = 0; j < n; j++) {
    double d = a[(size_t)j * n + j];
    for (CeedInt k = 0; k < j; k++) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    if (!(d > 0.)) { free(L); free(X); return oracle_error("CeedXCsrInvertDenseSPD: pivot %d is not positive", j); }
    L[(size_t)j * n + j] = sqrt(d);
    for (CeedInt i = j + 1; i < n; i++) {
      double v = a[(size_t)i * n + j];
      for (CeedInt k = 0; k < j; k++) v -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = v / L[(size_t)j * n + j];
    }
  }
  for (CeedInt c = 0; c < n; c++) {            /* X = L^-1, column by column (lower triangular) */
    X[(size_t)c * n + c] = 1. / L[(size_t)c * n + c];
    for (CeedInt i = c + 1; i < n; i++) {
      double v = 0.;
      for (CeedInt k = c; k < i; k++) v -= L[(size_t)i * n + k] * X[(size_t)k * n + c];
      X[(size_t)i * n + c] = v / L[(size_t)i * n + i];
    }
  }
  for (CeedInt r = 0; r < n; r++)
    for (CeedInt c = 0; c <= r; c++) {
      double v = 0.;
      for (CeedInt k = r; k < n; k++) v += X[(size_t)k * n + r] * X[(size_t)k * n + c];
      a[(size_t)r * n + c] = v; a[(size_t)c * n + r] = v;
    }
  free(L); free(X);
  return 0;
}
int CeedXCsrDestroy(CeedXCsr *csr) {
  if (!csr || !*csr) return 0;
  CeedXCsr A = *csr;
  *csr = NULL;
  if (--A->refs > 0) return 0;
  CeedXCsr src = A->src;
  free(A->rowptr); free(A->cols); free(A->coo_slot); free(A->unit_rows); free(A->vals);
  free(A->termptr); free(A->term_slot); free(A->term_w); free(A);
  if (src) CeedXCsrDestroy(&src);
  return 0;
}
