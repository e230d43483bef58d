// solid_harness.cpp -- the reference's host side of the path, PETSc-free (see include/solid_harness.h).
//
// Function for function what the reference does above libCEED:
//   SetupLibceedFineLevel  src/setuplibceed.c:243-745  (restrictions, bases, qdata by opSetupGeo, opApply)
//   SetupLibceedLevel      src/setuplibceed.c:748-939  (opJacob per level, opProlong / opRestrict)
//   SetupProlongRestrictCtx src/misc.c:73-146          (multiplicity -> multVec)
//   ApplyLocalCeedOp ... GetDiag_Ceed                   src/matops.c
// Only include/ceed.h is used, so this file links against either ABI library.
#include <solid_harness.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// problemOptions[] (setuplibceed.c:41-107): names and "file:name" locators as CEED_QFUNCTION builds them
struct ProblemData { const char *apply, *applyLoc, *jacob, *jacobLoc; bool state; };
static const ProblemData problemOptions[3] = {
    {"LinElasF", "qfunctions/linElas.h:LinElasF", "LinElasdF", "qfunctions/linElas.h:LinElasdF", false},
    {"HyperSSF", "qfunctions/hyperSS.h:HyperSSF", "HyperSSdF", "qfunctions/hyperSS.h:HyperSSdF", true},
    {"HyperFSF", "qfunctions/hyperFS.h:HyperFSF", "HyperFSdF", "qfunctions/hyperFS.h:HyperFSdF", true},
};

struct Physics_private { CeedScalar nu, E; };  // elasticity.h:33-36

struct CeedData_private {  // elasticity.h:218-240 (the members the path uses)
  CeedElemRestriction Erestrictu = nullptr;
  CeedBasis basisu = nullptr, basisCtoF = nullptr;
  CeedQFunction qfJacob = nullptr;
  CeedOperator opJacob = nullptr, opProlong = nullptr, opRestrict = nullptr;
  CeedVector xceed = nullptr, yceed = nullptr, multVec = nullptr;
  CeedInt Ulocsz = 0;
  CeedXHalo halo = nullptr;   // several GPUs: the interface sum of this level's L-vectors (borrowed; SolidAppSetHalo)
};

struct SolidApp_private {
  Ceed ceed = nullptr;
  problemType problemChoice = ELAS_LIN;
  Physics_private phys{}, physSmoother{};
  bool useSmootherPhys = false;
  CeedInt numLevels = 0, qextra = 0, nelem = 0;
  std::vector<CeedInt> levelDegrees;
  std::vector<CeedData_private> data;
  // fine-level objects shared by every level (setuplibceed.c:833-839)
  CeedElemRestriction Erestrictx = nullptr, Erestrictqdi = nullptr, ErestrictGradui = nullptr;
  CeedBasis basisx = nullptr;
  CeedVector xcoord = nullptr, qdata = nullptr, gradu = nullptr;
  CeedQFunction qfApply = nullptr, qfRestrict = nullptr, qfProlong = nullptr;
  CeedOperator opApply = nullptr;
};

#define CHK(x) do { int ierr_ = (x); if (ierr_) return ierr_; } while (0)

// Dirichlet handling of this build: G->L with zeroed Xloc (matops.c:33,106) and L->G dropping
// constrained rows (:57) are folded into the operator.
static int SetDirichlet(CeedOperator op, const unsigned char *mIn, CeedInt nIn, const unsigned char *mOut,
                        CeedInt nOut, int mode) {
  return CeedXOperatorSetDirichletMaskMode(op, CEED_MEM_HOST, mIn, nIn, mOut, nOut, mode);
}

static int SetupLibceedFineLevel(SolidApp app, CeedInt nvert, const CeedScalar *coords, const CeedInt *cells,
                                 const CeedInt *offsets, const unsigned char *mask) {
  Ceed ceed = app->ceed;
  const CeedInt fineLevel = app->numLevels - 1, nelem = app->nelem;
  const CeedInt P = app->levelDegrees[fineLevel] + 1, Q = P + app->qextra, dim = 3, ncompu = 3, ncompx = 3;
  const CeedInt qdatasize = 10, nqpts = Q * Q * Q;
  const ProblemData &pd = problemOptions[app->problemChoice];
  CeedData_private &d = app->data[fineLevel];

  // -- restrictions (:279-318): coordinates (P=2), solution, q-point data in the backend's layout
  std::vector<CeedInt> xoff((size_t)nelem * 8);
  for (size_t i = 0; i < xoff.size(); i++) xoff[i] = cells[i] * ncompx;
  CHK(CeedElemRestrictionCreate(ceed, nelem, 8, ncompx, 1, nvert * ncompx, CEED_MEM_HOST, CEED_COPY_VALUES,
                                xoff.data(), &app->Erestrictx));
  CHK(CeedElemRestrictionCreate(ceed, nelem, P * P * P, ncompu, 1, d.Ulocsz, CEED_MEM_HOST, CEED_COPY_VALUES,
                                offsets, &d.Erestrictu));
  CHK(CeedElemRestrictionCreateStrided(ceed, nelem, nqpts, qdatasize, qdatasize * nelem * nqpts,
                                       CEED_STRIDES_BACKEND, &app->Erestrictqdi));
  if (pd.state)
    CHK(CeedElemRestrictionCreateStrided(ceed, nelem, nqpts, dim * ncompu, dim * ncompu * nelem * nqpts,
                                         CEED_STRIDES_BACKEND, &app->ErestrictGradui));
  // -- element coordinates (:323-329)
  CHK(CeedElemRestrictionCreateVector(app->Erestrictx, &app->xcoord, NULL));
  CHK(CeedVectorSetArray(app->xcoord, CEED_MEM_HOST, CEED_COPY_VALUES, (CeedScalar *)coords));
  // -- bases (:335-341)
  CHK(CeedBasisCreateTensorH1Lagrange(ceed, dim, ncompu, P, Q, CEED_GAUSS, &d.basisu));
  CHK(CeedBasisCreateTensorH1Lagrange(ceed, dim, ncompx, 2, Q, CEED_GAUSS, &app->basisx));
  // -- persistent vectors (:353-361)
  CHK(CeedVectorCreate(ceed, qdatasize * nelem * nqpts, &app->qdata));
  if (pd.state) {
    CHK(CeedVectorCreate(ceed, dim * ncompu * nelem * nqpts, &app->gradu));
    CHK(CeedVectorSetValue(app->gradu, 0.));
  }
  // -- geometric factors (:370-393)
  {
    CeedQFunction qfSetupGeo;
    CeedOperator opSetupGeo;
    CHK(CeedQFunctionCreateInterior(ceed, 1, NULL, "qfunctions/common.h:SetupGeo", &qfSetupGeo));
    CHK(CeedQFunctionAddInput(qfSetupGeo, "dx", ncompx * dim, CEED_EVAL_GRAD));
    CHK(CeedQFunctionAddInput(qfSetupGeo, "weight", 1, CEED_EVAL_WEIGHT));
    CHK(CeedQFunctionAddOutput(qfSetupGeo, "qdata", qdatasize, CEED_EVAL_NONE));
    CHK(CeedOperatorCreate(ceed, qfSetupGeo, CEED_QFUNCTION_NONE, CEED_QFUNCTION_NONE, &opSetupGeo));
    CHK(CeedOperatorSetField(opSetupGeo, "dx", app->Erestrictx, app->basisx, CEED_VECTOR_ACTIVE));
    CHK(CeedOperatorSetField(opSetupGeo, "weight", CEED_ELEMRESTRICTION_NONE, app->basisx, CEED_VECTOR_NONE));
    CHK(CeedOperatorSetField(opSetupGeo, "qdata", app->Erestrictqdi, CEED_BASIS_COLLOCATED, CEED_VECTOR_ACTIVE));
    CHK(CeedOperatorApply(opSetupGeo, app->xcoord, app->qdata, CEED_REQUEST_IMMEDIATE));
    CHK(CeedQFunctionDestroy(&qfSetupGeo));
    CHK(CeedOperatorDestroy(&opSetupGeo));
  }
  // -- local residual evaluator (:518-542)
  CHK(CeedQFunctionCreateInterior(ceed, 1, NULL, pd.applyLoc, &app->qfApply));
  CHK(CeedQFunctionAddInput(app->qfApply, "du", ncompu * dim, CEED_EVAL_GRAD));
  CHK(CeedQFunctionAddInput(app->qfApply, "qdata", qdatasize, CEED_EVAL_NONE));
  CHK(CeedQFunctionAddOutput(app->qfApply, "dv", ncompu * dim, CEED_EVAL_GRAD));
  if (pd.state) CHK(CeedQFunctionAddOutput(app->qfApply, "gradu", ncompu * dim, CEED_EVAL_NONE));
  CHK(CeedQFunctionSetContext(app->qfApply, &app->phys, sizeof(app->phys)));
  CHK(CeedOperatorCreate(ceed, app->qfApply, CEED_QFUNCTION_NONE, CEED_QFUNCTION_NONE, &app->opApply));
  CHK(CeedOperatorSetField(app->opApply, "du", d.Erestrictu, d.basisu, CEED_VECTOR_ACTIVE));
  CHK(CeedOperatorSetField(app->opApply, "qdata", app->Erestrictqdi, CEED_BASIS_COLLOCATED, app->qdata));
  CHK(CeedOperatorSetField(app->opApply, "dv", d.Erestrictu, d.basisu, CEED_VECTOR_ACTIVE));
  if (pd.state)  // the reference passes basisu to this EVAL_NONE field (:538-539); ignored
    CHK(CeedOperatorSetField(app->opApply, "gradu", app->ErestrictGradui, d.basisu, app->gradu));
  // residual: boundary values stay in Xloc (matops.c:70-71), constrained rows are dropped (:57)
  CHK(SetDirichlet(app->opApply, mask, d.Ulocsz, NULL, 0, 2));
  return 0;
}

static int SetupLibceedLevel(SolidApp app, CeedInt level, const CeedInt *offsets, const unsigned char *const *masks) {
  Ceed ceed = app->ceed;
  const CeedInt fineLevel = app->numLevels - 1, nelem = app->nelem;
  const CeedInt P = app->levelDegrees[level] + 1, Q = app->levelDegrees[fineLevel] + 1 + app->qextra;  // (:756-757)
  const CeedInt dim = 3, ncompu = 3, qdatasize = 10;
  const ProblemData &pd = problemOptions[app->problemChoice];
  CeedData_private &d = app->data[level];

  if (level != fineLevel) {  // (:771-784)
    CHK(CeedElemRestrictionCreate(ceed, nelem, P * P * P, ncompu, 1, d.Ulocsz, CEED_MEM_HOST, CEED_COPY_VALUES,
                                  offsets, &d.Erestrictu));
    CHK(CeedBasisCreateTensorH1Lagrange(ceed, dim, ncompu, P, Q, CEED_GAUSS, &d.basisu));
  }
  if (level != 0)  // (:799-803)
    CHK(CeedBasisCreateTensorH1Lagrange(ceed, dim, ncompu, app->levelDegrees[level - 1] + 1, P, CEED_GAUSS_LOBATTO,
                                        &d.basisCtoF));
  CHK(CeedVectorCreate(ceed, d.Ulocsz, &d.xceed));  // (:808-809)
  CHK(CeedVectorCreate(ceed, d.Ulocsz, &d.yceed));
  // -- Jacobian evaluator (:818-839): coarse nodes, FINE quadrature and q-point data
  CHK(CeedQFunctionCreateInterior(ceed, 1, NULL, pd.jacobLoc, &d.qfJacob));
  CHK(CeedQFunctionAddInput(d.qfJacob, "deltadu", ncompu * dim, CEED_EVAL_GRAD));
  CHK(CeedQFunctionAddInput(d.qfJacob, "qdata", qdatasize, CEED_EVAL_NONE));
  if (pd.state) CHK(CeedQFunctionAddInput(d.qfJacob, "gradu", ncompu * dim, CEED_EVAL_NONE));
  CHK(CeedQFunctionAddOutput(d.qfJacob, "deltadv", ncompu * dim, CEED_EVAL_GRAD));
  CHK(CeedQFunctionSetContext(d.qfJacob, &app->phys, sizeof(&app->phys)));  // sizeof(pointer), as at :826
  CHK(CeedOperatorCreate(ceed, d.qfJacob, CEED_QFUNCTION_NONE, CEED_QFUNCTION_NONE, &d.opJacob));
  CHK(CeedOperatorSetField(d.opJacob, "deltadu", d.Erestrictu, d.basisu, CEED_VECTOR_ACTIVE));
  CHK(CeedOperatorSetField(d.opJacob, "qdata", app->Erestrictqdi, CEED_BASIS_COLLOCATED, app->qdata));
  CHK(CeedOperatorSetField(d.opJacob, "deltadv", d.Erestrictu, d.basisu, CEED_VECTOR_ACTIVE));
  if (pd.state) CHK(CeedOperatorSetField(d.opJacob, "gradu", app->ErestrictGradui, CEED_BASIS_COLLOCATED, app->gradu));
  CHK(SetDirichlet(d.opJacob, masks[level], d.Ulocsz, NULL, 0, 3));
  // -- multiplicity (SetupProlongRestrictCtx, misc.c:115-143)
  CHK(CeedElemRestrictionCreateVector(d.Erestrictu, &d.multVec, NULL));
  CHK(CeedElemRestrictionGetMultiplicity(d.Erestrictu, d.multVec));
  CHK(CeedVectorReciprocal(d.multVec));
  // -- restriction and prolongation (:847-862)
  if (level != 0) {
    CeedData_private &c = app->data[level - 1];
    CHK(CeedOperatorCreate(ceed, app->qfRestrict, CEED_QFUNCTION_NONE, CEED_QFUNCTION_NONE, &d.opRestrict));
    CHK(CeedOperatorSetField(d.opRestrict, "input", d.Erestrictu, CEED_BASIS_COLLOCATED, CEED_VECTOR_ACTIVE));
    CHK(CeedOperatorSetField(d.opRestrict, "output", c.Erestrictu, d.basisCtoF, CEED_VECTOR_ACTIVE));
    CHK(CeedOperatorCreate(ceed, app->qfProlong, CEED_QFUNCTION_NONE, CEED_QFUNCTION_NONE, &d.opProlong));
    CHK(CeedOperatorSetField(d.opProlong, "input", c.Erestrictu, d.basisCtoF, CEED_VECTOR_ACTIVE));
    CHK(CeedOperatorSetField(d.opProlong, "output", d.Erestrictu, CEED_BASIS_COLLOCATED, CEED_VECTOR_ACTIVE));
    // VecPointwiseMult with multVec (matops.c:149,176) folded into the transfer kernels
    CHK(CeedXOperatorSetFineScale(d.opRestrict, d.multVec));
    CHK(CeedXOperatorSetFineScale(d.opProlong, d.multVec));
    CHK(SetDirichlet(d.opProlong, masks[level - 1], c.Ulocsz, masks[level], d.Ulocsz, 3));
    CHK(SetDirichlet(d.opRestrict, masks[level], d.Ulocsz, masks[level - 1], c.Ulocsz, 3));
  }
  return 0;
}

extern "C" int SolidAppCreate(Ceed ceed, problemType problem, double nu, double E, CeedInt numLevels,
                              const CeedInt *levelDegrees, CeedInt qextra, CeedInt nelem, CeedInt nvert,
                              const CeedScalar *coords, const CeedInt *cells, const CeedInt *const *offsets,
                              const CeedInt *lsizes, const unsigned char *const *masks, SolidApp *out) {
  if (problem < ELAS_LIN || problem > ELAS_HYPER_FS) {
    fprintf(stderr, "[solid harness] unknown problem type %d\n", (int)problem);
    return 1;
  }
  SolidApp app = new SolidApp_private;
  app->ceed = ceed;
  app->problemChoice = problem;
  app->phys = {nu, E};
  app->numLevels = numLevels; app->qextra = qextra; app->nelem = nelem;
  app->levelDegrees.assign(levelDegrees, levelDegrees + numLevels);
  app->data.resize(numLevels);
  for (CeedInt l = 0; l < numLevels; l++) app->data[l].Ulocsz = lsizes[l];
  // identity QFunctions for the transfer operators (elasticity.c:249-252)
  CHK(CeedQFunctionCreateIdentity(ceed, 3, CEED_EVAL_NONE, CEED_EVAL_INTERP, &app->qfRestrict));
  CHK(CeedQFunctionCreateIdentity(ceed, 3, CEED_EVAL_INTERP, CEED_EVAL_NONE, &app->qfProlong));
  const CeedInt fineLevel = numLevels - 1;
  CHK(SetupLibceedFineLevel(app, nvert, coords, cells, offsets[fineLevel], masks[fineLevel]));  // elasticity.c:262
  for (CeedInt l = 0; l < numLevels; l++) CHK(SetupLibceedLevel(app, l, offsets[l], masks));      // elasticity.c:269-281
  *out = app;
  return 0;
}

extern "C" int SolidAppDestroy(SolidApp *papp) {  // CeedDataDestroy, setuplibceed.c:129-191
  if (!papp || !*papp) return 0;
  SolidApp app = *papp;
  for (auto &d : app->data) {
    CeedVectorDestroy(&d.xceed); CeedVectorDestroy(&d.yceed); CeedVectorDestroy(&d.multVec);
    CeedElemRestrictionDestroy(&d.Erestrictu);
    CeedBasisDestroy(&d.basisu); CeedBasisDestroy(&d.basisCtoF);
    CeedQFunctionDestroy(&d.qfJacob);
    CeedOperatorDestroy(&d.opJacob); CeedOperatorDestroy(&d.opProlong); CeedOperatorDestroy(&d.opRestrict);
  }
  CeedVectorDestroy(&app->qdata); CeedVectorDestroy(&app->gradu); CeedVectorDestroy(&app->xcoord);
  CeedElemRestrictionDestroy(&app->Erestrictx); CeedElemRestrictionDestroy(&app->Erestrictqdi);
  CeedElemRestrictionDestroy(&app->ErestrictGradui);
  CeedBasisDestroy(&app->basisx);
  CeedQFunctionDestroy(&app->qfApply); CeedQFunctionDestroy(&app->qfRestrict); CeedQFunctionDestroy(&app->qfProlong);
  CeedOperatorDestroy(&app->opApply);
  delete app;
  *papp = nullptr;
  return 0;
}

// ---- src/matops.c ------------------------------------------------------------------------------
// This function uses libCEED to compute the local action of an operator (matops.c:26-60).  With
// L-layout vectors the DMGlobalToLocal / VecZeroEntries / DMLocalToGlobal bracket is the operator's
// Dirichlet handling; the SetArray/TakeArray borrowing of PETSc arrays (:40-50) has no counterpart
// because X and Y already are CeedVectors.
extern "C" int ApplyLocalCeedOp(SolidApp, CeedOperator op, CeedVector X, CeedVector Y) {
  return CeedOperatorApply(op, X, Y, CEED_REQUEST_IMMEDIATE);
}
// DMLocalToGlobal(ADD_VALUES) (matops.c:57,153,199,238) and the DMGlobalToLocal of the next use, across the GPUs of the
// node: one neighbour sum on the replicated interface entries (CeedXHalo*, RCCL); nothing to do on one GPU.
static int LocalToGlobalAdd(SolidApp app, CeedInt level, CeedVector Y) {
  CeedXHalo h = app->data[level].halo;
  if (!h) return 0;
  CHK(CeedXHaloStart(h, Y));
  return CeedXHaloFinish(h, Y);
}
// ApplyLocalCeedOp followed by DMLocalToGlobal(ADD_VALUES) (matops.c:46,57) as ONE library call when the level has a halo:
// the library overlaps the exchange with the interior elements if the operator carries a split (CeedXOperatorSetOverlapSplit),
// and otherwise applies, then exchanges -- the same numbers either way.
static int ApplyLocalThenGlobalAdd(SolidApp app, CeedInt level, CeedOperator op, CeedVector X, CeedVector Y) {
  CeedXHalo h = app->data[level].halo;
  if (!h) return ApplyLocalCeedOp(app, op, X, Y);
  return CeedXOperatorApplyWithHalo(op, X, Y, h);
}
// matops.c:63-79: X carries the boundary values inserted at the current load increment
extern "C" int FormResidual_Ceed(SolidApp app, CeedVector X, CeedVector Y) {
  return ApplyLocalThenGlobalAdd(app, app->numLevels - 1, app->opApply, X, Y);
}
// matops.c:98-112
extern "C" int ApplyJacobian_Ceed(SolidApp app, CeedInt level, CeedVector X, CeedVector Y) {
  return ApplyLocalThenGlobalAdd(app, level, app->data[level].opJacob, X, Y);
}
// matops.c:115-157 (level-1 -> level)
extern "C" int Prolong_Ceed(SolidApp app, CeedInt level, CeedVector Xc, CeedVector Yf) {
  CHK(CeedOperatorApply(app->data[level].opProlong, Xc, Yf, CEED_REQUEST_IMMEDIATE));
  return LocalToGlobalAdd(app, level, Yf);
}
// matops.c:160-203 (level -> level-1)
extern "C" int Restrict_Ceed(SolidApp app, CeedInt level, CeedVector Xf, CeedVector Yc) {
  CHK(CeedOperatorApply(app->data[level].opRestrict, Xf, Yc, CEED_REQUEST_IMMEDIATE));
  return LocalToGlobalAdd(app, level - 1, Yc);
}
// matops.c:206-244, including the context swap for -nu_smoother (:215-217, :231-232)
extern "C" int GetDiag_Ceed(SolidApp app, CeedInt level, CeedVector D) {
  CeedData_private &d = app->data[level];
  if (app->useSmootherPhys) CHK(CeedQFunctionSetContext(d.qfJacob, &app->physSmoother, sizeof(app->physSmoother)));
  CHK(CeedOperatorLinearAssembleDiagonal(d.opJacob, D, CEED_REQUEST_IMMEDIATE));
  if (app->useSmootherPhys) CHK(CeedQFunctionSetContext(d.qfJacob, &app->phys, sizeof(app->phys)));
  return LocalToGlobalAdd(app, level, D);
}
// Several GPUs: the halo of level `level` (created by the caller with CeedXHaloCreate from the partition's neighbour
// lists; borrowed, NULL clears).  From then on every matops function above ends with the interface sum.
extern "C" int SolidAppSetHalo(SolidApp app, CeedInt level, CeedXHalo halo) {
  if (level < 0 || level >= app->numLevels) return 1;
  app->data[level].halo = halo;
  return 0;
}
extern "C" int SolidAppSetSmootherNu(SolidApp app, double nu_smoother) {
  app->useSmootherPhys = nu_smoother >= 0.;
  app->physSmoother = {nu_smoother, app->phys.E};  // elasticity.c:189-199
  return 0;
}

extern "C" int SolidAppGetVectors(SolidApp app, CeedVector *qdata, CeedVector *gradu) {
  if (qdata) *qdata = app->qdata;
  if (gradu) *gradu = app->gradu;
  return 0;
}
extern "C" int SolidAppGetLevelOperators(SolidApp app, CeedInt level, CeedOperator *opJacob, CeedOperator *opProlong,
                                         CeedOperator *opRestrict) {
  if (level < 0 || level >= app->numLevels) return 1;
  if (opJacob) *opJacob = app->data[level].opJacob;
  if (opProlong) *opProlong = app->data[level].opProlong;
  if (opRestrict) *opRestrict = app->data[level].opRestrict;
  return 0;
}
extern "C" int SolidAppGetResidualOperator(SolidApp app, CeedOperator *opApply) { *opApply = app->opApply; return 0; }
extern "C" int SolidAppGetMultiplicityInverse(SolidApp app, CeedInt level, CeedVector *multinv) {
  *multinv = app->data[level].multVec;
  return 0;
}
