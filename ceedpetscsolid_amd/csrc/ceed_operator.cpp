// ceed_operator.cpp -- CeedQFunction, CeedOperator and the operator apply of the MI355X backend.
//
// A CeedOperator is lowered, at its first apply, to one hand-written gfx950 kernel family by matching its field
// signature against the operator graphs the reference builds:
//
//   fused_grad : GRAD active in, NONE qdata (+ NONE state in / out), GRAD active out
//                -> opApply (setuplibceed.c:517-542) and opJacob per level (:817-839)
//   setup_geo  : GRAD coords + WEIGHT -> NONE qdata             (:370-389)
//   prolong    : Identity, INTERP in -> NONE out                (:857-862)
//   restrict   : Identity, NONE in  -> INTERP out               (:849-854)
//   coord / energy : forcing, MMS, strain energy, diagnostics   (:555-737)
//
// There is NO host fallback: a graph outside these families or a QFunction without a device functor is a loud error.
#include "ceed_impl.hpp"

using namespace cps;

// ---------------------------------------------------------------------------
// CeedQFunction
// ---------------------------------------------------------------------------
static int resolve_qf(const std::string &name) {
  static const struct { const char *n; int k; } tab[] = {
      {"SetupGeo", QF_SETUP_GEO},    {"LinElasF", QF_LINELAS},       {"LinElasdF", QF_LINELAS},
      {"HyperSSF", QF_HYPERSS_F},    {"HyperSSdF", QF_HYPERSS_DF},   {"HyperFSF", QF_HYPERFS_F},
      {"HyperFSdF", QF_HYPERFS_DF},  {"SetupConstantForce", QF_CONST_FORCE}, {"SetupMMSForce", QF_MMS_FORCE},
      {"MMSTrueSoln", QF_MMS_TRUE},  {"LinElasEnergy", QF_ENERGY_LINELAS}, {"HyperSSEnergy", QF_ENERGY_HYPERSS},
      {"HyperFSEnergy", QF_ENERGY_HYPERFS}, {"LinElasDiagnostic", QF_DIAG_LINELAS}, {"HyperSSDiagnostic", QF_DIAG_HYPERSS},
      {"HyperFSDiagnostic", QF_DIAG_HYPERFS},
  };
  for (auto &t : tab) if (name == t.n) return t.k;
  return QF_NONE;
}
extern "C" int CeedQFunctionCreateInterior(Ceed ceed, CeedInt, CeedQFunctionUser f, const char *source,
                                           CeedQFunction *qf) {
  std::string src = source ? source : "";
  const size_t colon = src.rfind(':');
  std::string name = colon == std::string::npos ? src : src.substr(colon + 1);
  const int kind = resolve_qf(name);
  if (kind == QF_NONE)
    return ceed_error("QFunction '%s' has no gfx950 device functor in this backend (host callbacks are "
                      "never executed on /gpu/hip/mi355x)", src.c_str());
  CeedQFunction q = new CeedQFunction_private;
  q->ceed = ceed; ceed_ref(ceed);
  q->f = f; q->source = src; q->name = name; q->kind = kind;
  *qf = q;
  return 0;
}
extern "C" int CeedQFunctionCreateIdentity(Ceed ceed, CeedInt size, CeedEvalMode inmode, CeedEvalMode outmode,
                                           CeedQFunction *qf) {
  CeedQFunction q = new CeedQFunction_private;
  q->ceed = ceed; ceed_ref(ceed);
  q->name = q->source = "Identity"; q->kind = QF_IDENTITY; q->identity_size = size;
  q->in.push_back({"input", size, inmode});
  q->out.push_back({"output", size, outmode});
  *qf = q;
  return 0;
}
extern "C" int CeedQFunctionAddInput(CeedQFunction qf, const char *name, CeedInt size, CeedEvalMode em) {
  qf->in.push_back({name, size, em});
  return 0;
}
extern "C" int CeedQFunctionAddOutput(CeedQFunction qf, const char *name, CeedInt size, CeedEvalMode em) {
  if (em == CEED_EVAL_WEIGHT) return ceed_error("WEIGHT is not an output mode");
  qf->out.push_back({name, size, em});
  return 0;
}
extern "C" int CeedQFunctionSetContext(CeedQFunction qf, void *ctx, size_t ctxsize) {
  qf->ctx = ctx; qf->ctxsize = ctxsize;  // borrowed; re-read at every apply (matops.c:215-232)
  return 0;
}
extern "C" int CeedQFunctionDestroy(CeedQFunction *qf) {
  if (!qf || !*qf) return 0;
  CeedQFunction q = *qf;
  *qf = nullptr;
  if (q == CEED_QFUNCTION_NONE) return 0;
  if (--q->refcount > 0) return 0;
  ceed_unref(q->ceed);
  delete q;
  return 0;
}

// ---------------------------------------------------------------------------
// CeedOperator
// ---------------------------------------------------------------------------
extern "C" int CeedOperatorCreate(Ceed ceed, CeedQFunction qf, CeedQFunction, CeedQFunction, CeedOperator *op) {
  CeedOperator o = new CeedOperator_private;
  o->ceed = ceed; ceed_ref(ceed);
  o->qf = qf; qf->refcount++;
  o->in.resize(qf->in.size()); o->out.resize(qf->out.size());
  *op = o;
  return 0;
}
extern "C" int CeedCompositeOperatorCreate(Ceed ceed, CeedOperator *op) {
  CeedOperator o = new CeedOperator_private;
  o->ceed = ceed; ceed_ref(ceed);
  o->composite = true;
  *op = o;
  return 0;
}
extern "C" int CeedCompositeOperatorAddSub(CeedOperator comp, CeedOperator sub) {
  if (!comp->composite) return ceed_error("not a composite operator");
  comp->sub.push_back(sub); sub->refcount++;
  return 0;
}
extern "C" int CeedOperatorSetField(CeedOperator op, const char *name, CeedElemRestriction r, CeedBasis b, CeedVector v) {
  if (op->composite) return ceed_error("cannot set a field on a composite operator");
  if (op->in.size() != op->qf->in.size()) op->in.resize(op->qf->in.size());
  if (op->out.size() != op->qf->out.size()) op->out.resize(op->qf->out.size());
  OpField *f = nullptr;
  for (size_t i = 0; i < op->qf->in.size() && !f; i++) if (op->qf->in[i].name == name) f = &op->in[i];
  for (size_t i = 0; i < op->qf->out.size() && !f; i++) if (op->qf->out[i].name == name) f = &op->out[i];
  if (!f) return ceed_error("QFunction '%s' has no field named '%s'", op->qf->name.c_str(), name);
  f->set = true; f->rstr = r; f->basis = b; f->vec = v;
  if (r != CEED_ELEMRESTRICTION_NONE) r->refcount++;
  if (b != CEED_BASIS_COLLOCATED) b->refcount++;
  if (v != CEED_VECTOR_ACTIVE && v != CEED_VECTOR_NONE) v->refcount++;
  op->plan = PLAN_NONE;
  return 0;
}
static void op_free_flags(CeedOperator o) {
  if (o->d_off_flagged_out && o->d_off_flagged_out != o->d_off_flagged_in) (void)hipFree(o->d_off_flagged_out);
  if (o->d_off_flagged_in) (void)hipFree(o->d_off_flagged_in);
  o->d_off_flagged_in = o->d_off_flagged_out = nullptr;
  if (o->d_node_flags) (void)hipFree(o->d_node_flags);
  if (o->d_node_flags_ovl) (void)hipFree(o->d_node_flags_ovl);
  if (o->d_node_flags_shell) (void)hipFree(o->d_node_flags_shell);
  for (auto &pf : o->pipe_flags) ceed_retire(o->ceed, pf.second);   // (recorded graphs may still read them)
  o->pipe_flags.clear();
  o->d_node_flags = o->d_node_flags_ovl = o->d_node_flags_shell = nullptr;
  o->h_mask.clear();
  o->h_mask_fine.clear();
  ceed_retire(o->ceed, o->d_own_f); o->d_own_f = nullptr;     // (recorded graphs may still read it)
  o->mask_mode = 0;
}
extern "C" int CeedOperatorDestroy(CeedOperator *op) {
  if (!op || !*op) return 0;
  CeedOperator o = *op;
  *op = nullptr;
  if (--o->refcount > 0) return 0;
  if (o->composite) {
    for (CeedOperator s : o->sub) CeedOperatorDestroy(&s);
  } else {
    for (auto *arr : {&o->in, &o->out})
      for (OpField &f : *arr) {
        if (!f.set) continue;
        CeedElemRestrictionDestroy(&f.rstr); CeedBasisDestroy(&f.basis); CeedVectorDestroy(&f.vec);
      }
    CeedQFunctionDestroy(&o->qf);
  }
  op_free_flags(o);
  for (auto &pf : o->pack_folds) { ceed_retire(o->ceed, pf.d_ptr); ceed_retire(o->ceed, pf.d_slot); }
  o->ovl_csr.release();
  ceed_retire(o->ceed, o->d_w);
  CeedVectorDestroy(&o->scale);
  for (auto &ev : o->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  ceed_unref(o->ceed);
  delete o;
  return 0;
}

static void fill_tables(BasisTables &t, CeedBasis b) {
  memset(&t, 0, sizeof t);
  memcpy(t.interp, b->interp1d.data(), sizeof(double) * b->interp1d.size());
  memcpy(t.grad, b->grad1d.data(), sizeof(double) * b->grad1d.size());
  memcpy(t.colo, b->colo1d.data(), sizeof(double) * b->colo1d.size());
  memcpy(t.qw, b->qweight1d.data(), sizeof(double) * b->qweight1d.size());
}
// Even-odd form of one 1-D table (FusedGradArgs::eo).  M(o, m) = TR ? tab[m * LD + o] : tab[o * LD + m], NOUT x NIN,
// expected centro-symmetric (sgn = +1) or centro-antisymmetric (sgn = -1); false if it is not (to 1e-13).
static bool build_eo_table(const double *tab, int NOUT, int NIN, int LD, bool TR, int sgn, double *T) {
  auto M = [&](int o, int m) { return TR ? tab[m * LD + o] : tab[o * LD + m]; };
  double mx = 0.;
  for (int o = 0; o < NOUT; o++) for (int m = 0; m < NIN; m++) mx = std::max(mx, fabs(M(o, m)));
  for (int o = 0; o < NOUT; o++) for (int m = 0; m < NIN; m++)
    if (fabs(M(NOUT - 1 - o, NIN - 1 - m) - sgn * M(o, m)) > 1e-13 * mx) return false;
  const int HIN = NIN / 2, COUT = (NOUT + 1) / 2;
  if (COUT > 4 || HIN > 4 || 2 * COUT * HIN + COUT > 30) return false;   // table must stay within 60 SGPRs
  for (int i = 0; i < EO_TAB; i++) T[i] = 0.;
  for (int r = 0; r < COUT; r++) {
    for (int j = 0; j < HIN; j++) {
      T[r * HIN + j] = 0.5 * (M(r, j) + M(r, NIN - 1 - j));
      T[16 + r * HIN + j] = 0.5 * (M(r, j) - M(r, NIN - 1 - j));
    }
    if (NIN & 1) T[32 + r] = M(r, HIN);
  }
  return true;
}

// Match the operator's field signature against the supported kernel families.
static int op_plan(CeedOperator op) {
  if (op->plan != PLAN_NONE) return 0;
  CeedQFunction qf = op->qf;
  for (size_t i = 0; i < qf->in.size(); i++) if (!op->in[i].set) return ceed_error("operator field '%s' not set", qf->in[i].name.c_str());
  for (size_t i = 0; i < qf->out.size(); i++) if (!op->out[i].set) return ceed_error("operator field '%s' not set", qf->out[i].name.c_str());
  op->i_active = op->i_qdata = op->i_state = op->i_weight = op->o_active = op->o_state = op->o_qdata = -1;
  const int k = qf->kind;
  auto unsupported = [&](const char *why) {
    return ceed_error("operator with QFunction '%s' is outside the kernel families of /gpu/hip/mi355x: %s",
                      qf->name.c_str(), why);
  };
  if (k == QF_LINELAS || k == QF_HYPERSS_F || k == QF_HYPERSS_DF || k == QF_HYPERFS_F || k == QF_HYPERFS_DF) {
    // inputs: GRAD active (9) | NONE qdata (10) | [NONE state (9)]
    for (size_t i = 0; i < qf->in.size(); i++) {
      const QFField &f = qf->in[i];
      if (f.emode == CEED_EVAL_GRAD && op->in[i].vec == CEED_VECTOR_ACTIVE && f.size == 9 && op->i_active < 0) op->i_active = (int)i;
      else if (f.emode == CEED_EVAL_NONE && f.size == 10 && op->i_qdata < 0) op->i_qdata = (int)i;
      else if (f.emode == CEED_EVAL_NONE && f.size == 9 && op->i_state < 0) op->i_state = (int)i;
      else return unsupported("unexpected input field");
    }
    for (size_t i = 0; i < qf->out.size(); i++) {
      const QFField &f = qf->out[i];
      if (f.emode == CEED_EVAL_GRAD && op->out[i].vec == CEED_VECTOR_ACTIVE && f.size == 9 && op->o_active < 0) op->o_active = (int)i;
      else if (f.emode == CEED_EVAL_NONE && f.size == 9 && op->o_state < 0) op->o_state = (int)i;
      else return unsupported("unexpected output field");
    }
    if (op->i_active != 0 || op->i_qdata != 1) return unsupported("inputs must be (GRAD active, NONE qdata[, NONE state])");
    const bool st_in = (k == QF_HYPERSS_DF || k == QF_HYPERFS_DF), st_out = (k == QF_HYPERSS_F || k == QF_HYPERFS_F);
    if (st_in != (op->i_state >= 0) || st_out != (op->o_state >= 0) || op->o_active != 0)
      return unsupported("stored-state fields do not match the QFunction");
    OpField &ai = op->in[op->i_active], &ao = op->out[op->o_active], &qd = op->in[op->i_qdata];
    if (!is_offsets(ai.rstr) || ai.rstr != ao.rstr || ai.basis != ao.basis || ai.basis == CEED_BASIS_COLLOCATED)
      return unsupported("active input and output must share one offsets restriction and one basis");
    if (ai.rstr->ncomp != 3 || ai.rstr->compstride != 1) return unsupported("active fields must be 3 interlaced components");
    CeedBasis b = ai.basis;
    const int P = b->P1d, Q = b->Q1d, Q3 = Q * Q * Q;
    if (ai.rstr->elemsize != P * P * P) return unsupported("restriction element size is not P^3");
    if (!is_strided(qd.rstr) || qd.rstr->elemsize != Q3 || qd.rstr->ncomp != 10 || qd.rstr->nelem != ai.rstr->nelem)
      return unsupported("qdata must be a strided 10 x Q^3 field");
    if (st_in) { OpField &s = op->in[op->i_state]; if (!is_strided(s.rstr) || s.rstr->elemsize != Q3 || s.rstr->ncomp != 9) return unsupported("state input must be strided 9 x Q^3"); }
    if (st_out) { OpField &s = op->out[op->o_state]; if (!is_strided(s.rstr) || s.rstr->elemsize != Q3 || s.rstr->ncomp != 9) return unsupported("state output must be strided 9 x Q^3"); }
    fill_tables(op->tables, b);
    if (pencil_even_odd(Q)) {   // even-odd forms of the six products, built once here (not per apply)
      const BasisTables &t = op->tables;
      const bool ok = build_eo_table(t.interp, Q, P, P, false, +1, op->eo[0]) && build_eo_table(t.interp, P, Q, P, true, +1, op->eo[1]) &&
                      build_eo_table(t.colo, Q, Q, Q, false, -1, op->eo[2]) && build_eo_table(t.colo, Q, Q, Q, true, -1, op->eo[3]) &&
                      build_eo_table(t.grad, Q, P, P, false, -1, op->eo[4]) && build_eo_table(t.grad, P, Q, P, true, -1, op->eo[5]);
      if (!ok) return unsupported("the basis tables are not centro-symmetric (they are for every CeedBasisCreateTensorH1Lagrange basis)");
    }
    op->plan = PLAN_FUSED_GRAD;
    return 0;
  }
  if (k == QF_SETUP_GEO) {
    if (qf->in.size() != 2 || qf->out.size() != 1) return unsupported("SetupGeo takes (dx, weight) -> qdata");
    if (qf->in[0].emode != CEED_EVAL_GRAD || qf->in[1].emode != CEED_EVAL_WEIGHT || qf->out[0].emode != CEED_EVAL_NONE)
      return unsupported("SetupGeo eval modes must be GRAD, WEIGHT -> NONE");
    OpField &x = op->in[0], &qd = op->out[0];
    if (!is_offsets(x.rstr) || x.rstr->elemsize != 8 || x.rstr->ncomp != 3 || x.rstr->compstride != 1 || x.basis == CEED_BASIS_COLLOCATED || x.basis->P1d != 2)
      return unsupported("coordinates must be trilinear (P=2), 3 interlaced components (setuplibceed.c:279,339)");
    const int Q = x.basis->Q1d;
    if (!is_strided(qd.rstr) || qd.rstr->ncomp != 10 || qd.rstr->elemsize != Q * Q * Q) return unsupported("qdata must be strided 10 x Q^3");
    op->i_active = 0; op->i_weight = 1; op->o_qdata = 0;
    fill_tables(op->tables, x.basis);
    op->plan = PLAN_SETUP_GEO;
    return 0;
  }
  if (k == QF_IDENTITY) {
    if (qf->identity_size != 3) return unsupported("identity transfer operators carry 3 components");
    OpField &fi = op->in[0], &fo = op->out[0];
    const CeedEvalMode mi = qf->in[0].emode, mo = qf->out[0].emode;
    if (!is_offsets(fi.rstr) || !is_offsets(fo.rstr) || fi.rstr->nelem != fo.rstr->nelem) return unsupported("transfer needs offsets restrictions on both sides");
    if (fi.rstr->ncomp != 3 || fo.rstr->ncomp != 3 || fi.rstr->compstride != 1 || fo.rstr->compstride != 1) return unsupported("3 interlaced components expected");
    if (mi == CEED_EVAL_INTERP && mo == CEED_EVAL_NONE && fi.basis != CEED_BASIS_COLLOCATED && fo.basis == CEED_BASIS_COLLOCATED) {
      CeedBasis b = fi.basis;
      if (fi.rstr->elemsize != b->P1d * b->P1d * b->P1d || fo.rstr->elemsize != b->Q1d * b->Q1d * b->Q1d) return unsupported("prolongation sizes");
      fill_tables(op->tables, b);
      op->plan = PLAN_PROLONG;
    } else if (mi == CEED_EVAL_NONE && mo == CEED_EVAL_INTERP && fi.basis == CEED_BASIS_COLLOCATED && fo.basis != CEED_BASIS_COLLOCATED) {
      CeedBasis b = fo.basis;
      if (fo.rstr->elemsize != b->P1d * b->P1d * b->P1d || fi.rstr->elemsize != b->Q1d * b->Q1d * b->Q1d) return unsupported("restriction sizes");
      fill_tables(op->tables, b);
      op->plan = PLAN_RESTRICT;
    } else return unsupported("identity operator is neither INTERP->NONE nor NONE->INTERP");
    op->i_active = 0; op->o_active = 0;
    return 0;
  }
  if (k == QF_ENERGY_LINELAS || k == QF_ENERGY_HYPERSS || k == QF_ENERGY_HYPERFS) {
    // opEnergy (setuplibceed.c:651-670): (du GRAD active, qdata NONE) -> energy INTERP, 1 component
    if (qf->in.size() != 2 || qf->out.size() != 1) return unsupported("energy takes (du, qdata) -> energy");
    if (qf->in[0].emode != CEED_EVAL_GRAD || qf->in[0].size != 9 || qf->in[1].emode != CEED_EVAL_NONE || qf->in[1].size != 10 ||
        qf->out[0].emode != CEED_EVAL_INTERP || qf->out[0].size != 1)
      return unsupported("energy eval modes must be GRAD(9), NONE(10) -> INTERP(1)");
    OpField &u = op->in[0], &qd = op->in[1], &en = op->out[0];
    if (!is_offsets(u.rstr) || u.rstr->ncomp != 3 || u.rstr->compstride != 1 || u.basis == CEED_BASIS_COLLOCATED) return unsupported("displacement field");
    const int P = u.basis->P1d, Q = u.basis->Q1d;
    if (u.rstr->elemsize != P * P * P) return unsupported("restriction element size is not P^3");
    if (!is_strided(qd.rstr) || qd.rstr->ncomp != 10 || qd.rstr->elemsize != Q * Q * Q) return unsupported("qdata must be strided 10 x Q^3");
    if (!is_offsets(en.rstr) || en.rstr->ncomp != 1 || en.rstr->nelem != u.rstr->nelem || en.basis == CEED_BASIS_COLLOCATED ||
        en.basis->P1d * en.basis->P1d * en.basis->P1d != en.rstr->elemsize || en.basis->Q1d != Q || en.basis->P1d != P)
      return unsupported("energy field must be a 1-component field on the displacement's nodes and points");
    op->i_active = 0; op->i_qdata = 1; op->o_active = 0;
    op->plan = PLAN_ENERGY;
    return 0;
  }
  if (k == QF_DIAG_LINELAS || k == QF_DIAG_HYPERSS || k == QF_DIAG_HYPERFS) {
    // opDiagnostic (setuplibceed.c:712-737): (u INTERP, du GRAD, qdata NONE) -> diagnostic NONE, 8 components
    if (qf->in.size() != 3 || qf->out.size() != 1) return unsupported("diagnostic takes (u, du, qdata) -> diagnostic");
    if (qf->in[0].emode != CEED_EVAL_INTERP || qf->in[0].size != 3 || qf->in[1].emode != CEED_EVAL_GRAD || qf->in[1].size != 9 ||
        qf->in[2].emode != CEED_EVAL_NONE || qf->in[2].size != 10 || qf->out[0].emode != CEED_EVAL_NONE || qf->out[0].size != 8)
      return unsupported("diagnostic eval modes must be INTERP(3), GRAD(9), NONE(10) -> NONE(8)");
    OpField &u = op->in[0], &du = op->in[1], &qd = op->in[2], &dg = op->out[0];
    if (op->in[0].vec != CEED_VECTOR_ACTIVE || op->in[1].vec != CEED_VECTOR_ACTIVE || u.rstr != du.rstr || u.basis != du.basis)
      return unsupported("u and du must be the same active field");
    if (!is_offsets(u.rstr) || u.rstr->ncomp != 3 || u.rstr->compstride != 1 || u.basis == CEED_BASIS_COLLOCATED) return unsupported("displacement field");
    const int P = u.basis->P1d, Q = u.basis->Q1d;
    if (u.rstr->elemsize != P * P * P) return unsupported("restriction element size is not P^3");
    if (!is_strided(qd.rstr) || qd.rstr->ncomp != 10 || qd.rstr->elemsize != Q * Q * Q) return unsupported("qdata must be strided 10 x Q^3");
    if (!is_offsets(dg.rstr) || dg.rstr->ncomp != 8 || dg.rstr->compstride != 1 || dg.rstr->nelem != u.rstr->nelem ||
        dg.rstr->elemsize != Q * Q * Q || dg.basis != CEED_BASIS_COLLOCATED)
      return unsupported("diagnostic field must be 8 interlaced components collocated with the points");
    op->i_active = 0; op->i_qdata = 2; op->o_active = 0;
    op->plan = PLAN_ENERGY;
    return 0;
  }
  if (k == QF_CONST_FORCE || k == QF_MMS_FORCE || k == QF_MMS_TRUE) {
    // opSetupForce: (x INTERP, qdata NONE) -> force INTERP (setuplibceed.c:555-583); opTrue: x INTERP -> true_soln NONE (:608-623)
    const bool force = k != QF_MMS_TRUE;
    if (qf->in.size() != (force ? 2u : 1u) || qf->out.size() != 1) return unsupported("expected (x[, qdata]) -> one output");
    if (qf->in[0].emode != CEED_EVAL_INTERP || qf->in[0].size != 3 || qf->out[0].size != 3) return unsupported("x must be 3 components, INTERP");
    OpField &x = op->in[0], &o = op->out[0];
    if (!is_offsets(x.rstr) || x.rstr->elemsize != 8 || x.rstr->ncomp != 3 || x.rstr->compstride != 1 || x.basis == CEED_BASIS_COLLOCATED || x.basis->P1d != 2)
      return unsupported("coordinates must be trilinear (P=2), 3 interlaced components");
    if (!is_offsets(o.rstr) || o.rstr->ncomp != 3 || o.rstr->compstride != 1 || o.rstr->nelem != x.rstr->nelem) return unsupported("output must be an offsets restriction with 3 interlaced components");
    const int Q = x.basis->Q1d;
    if (force) {
      if (qf->in[1].emode != CEED_EVAL_NONE || qf->in[1].size != 10 || qf->out[0].emode != CEED_EVAL_INTERP) return unsupported("forcing takes qdata NONE and gives force INTERP");
      OpField &qd = op->in[1];
      if (!is_strided(qd.rstr) || qd.rstr->ncomp != 10 || qd.rstr->elemsize != Q * Q * Q) return unsupported("qdata must be strided 10 x Q^3");
      if (o.basis == CEED_BASIS_COLLOCATED || o.basis->Q1d != Q || o.rstr->elemsize != o.basis->P1d * o.basis->P1d * o.basis->P1d) return unsupported("force basis must share the quadrature of the coordinate basis");
      op->i_qdata = 1;
    } else {
      if (qf->out[0].emode != CEED_EVAL_NONE || o.basis != CEED_BASIS_COLLOCATED || o.rstr->elemsize != Q * Q * Q) return unsupported("true solution is collocated on the points of the coordinate basis");
    }
    op->i_active = 0; op->o_active = 0;
    op->plan = PLAN_COORD;
    return 0;
  }
  return unsupported("no kernel family");
}

static void lame_constants(double nu, double E, double *lambda, double *TwoMu) {
  // hyperSS.h:79-81 / hyperFS.h:164-167, evaluated once per apply on the host
  *TwoMu = E / (1 + nu);
  const double Kbulk = E / (3 * (1 - 2 * nu));
  *lambda = (3 * Kbulk - *TwoMu) / 3;
}
static int read_phys(CeedQFunction qf, double *nu, double *E) {
  // The reference passes sizeof(pointer) as the context size at setuplibceed.c:826; the
  // context is the 16-byte {nu, E} struct behind the pointer (elasticity.h:33-36).
  if (!qf->ctx) return ceed_error("QFunction '%s' needs its Physics context", qf->name.c_str());
  const double *p = (const double *)qf->ctx;
  *nu = p[0]; *E = p[1];
  return 0;
}

// a recording remembers each provenance buffer it read once (a V-cycle applies the same operators many times)
static void capture_dep(Ceed c, const GraphDep &d) {
  for (const GraphDep &e : c->capture_deps)
    if (e.v == d.v && e.geo == d.geo && e.derived == d.derived) return;
  d.v->refcount++;      // held from here: the vector may be destroyed by its creator before the recording ends
  c->capture_deps.push_back(d);
}

struct TimerScope {
  CeedOperator op; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
  TimerScope(CeedOperator o, hipStream_t st) : op(o), s(st) {
    if (op->timing && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, s);
  }
  ~TimerScope() {
    if (a && b) { (void)hipEventRecord(b, s); op->events.emplace_back(a, b); }
  }
};


// ---------------------------------------------------------------------------
// The residual / Jacobian operator: k_fused_pencil (+ k_assemble)
// ---------------------------------------------------------------------------
// Everything of one apply that does not depend on the element range of a launch.
struct FusedApply {
  FusedGradArgs a{};
  CeedElemRestriction r = nullptr;
  CeedBasis b = nullptr;
  int qfkind = 0;
  bool add = false, direct = false, split = false;
  CeedVector derived_for = nullptr;        // residual applies: the state vector whose derived state this apply writes
  double *py = nullptr;
  const CsrMap *M = nullptr;               // the transpose map of this apply (restriction's, shell, or the operator's split map)
  const unsigned char *flags = nullptr;    // Dirichlet flags per row of M (null: none)
};
static unsigned char *make_row_flags(CeedOperator op, CeedElemRestriction r, const std::vector<uint32_t> &node_off, int *err) {
  std::vector<unsigned char> fl(node_off.size(), 0);
  for (size_t i = 0; i < node_off.size(); i++)
    for (int c = 0; c < r->ncomp && c < 3; c++)
      if (op->h_mask[(size_t)node_off[i] + (size_t)c * r->compstride]) fl[i] |= (unsigned char)(1u << c);
  unsigned char *d = nullptr;
  if (hipMalloc((void **)&d, fl.size() ? fl.size() : 1) != hipSuccess ||
      hipMemcpy(d, fl.data(), fl.size(), hipMemcpyHostToDevice) != hipSuccess) { *err = ceed_error("device allocation of the Dirichlet row flags failed"); return nullptr; }
  return d;
}
// vectors, tables, geometry provenance, physics, transpose map, flags, scratch: the launch arguments of this apply
static int fused_prepare(CeedOperator op, CeedVector in, CeedVector out, bool add, bool split, FusedApply &F) {
  CeedQFunction qf = op->qf;
  OpField &ai = op->in[op->i_active];
  CeedElemRestriction r = ai.rstr;
  Ceed c = op->ceed;
  if (!in || in == CEED_VECTOR_NONE || !out || out == CEED_VECTOR_NONE) return ceed_error("active vectors required");
  if (in->length < r->lsize || out->length < r->lsize) return ceed_error("active vector shorter than the restriction's L-size");
  if (in == out) return ceed_error("in-place operator apply is not supported");
  FusedGradArgs &a = F.a;
  F.r = r; F.b = ai.basis; F.qfkind = qf->kind; F.add = add; F.split = split;
  double *px, *pq, *ps = nullptr;
  CHK(vec_dev(in, false, &px));
  CHK(vec_dev(out, true, &F.py));
  CHK(vec_dev(op->in[op->i_qdata].vec, false, &pq));
  a.offsets = op->d_off_flagged_in ? op->d_off_flagged_in : r->d_offsets;
  a.x = px; a.y = F.py; a.qdata = pq;
  CHK(read_phys(qf, &a.nu, &a.E));
  const int Q3 = ai.basis->Q1d * ai.basis->Q1d * ai.basis->Q1d;
  if (op->i_state >= 0) {
    CeedVector sv = op->in[op->i_state].vec;
    CHK(vec_dev(sv, false, &ps)); a.state_in = ps;
    // the derived state the residual kernel left beside grad u, if it still belongs to it (same elements, points, material)
    if (qf->kind == QF_HYPERFS_DF && c->opt.derived_state && pencil_derived_state(ai.basis->Q1d) && sv->derived_valid && sv->derived_nelem == r->nelem && sv->derived_Q3 == Q3 &&
        sv->derived_nu == a.nu && sv->derived_E == a.E) {
      F.qfkind = QF_HYPERFS_DF_DS;
      a.state_in = sv->derived;
      if (c->capturing) capture_dep(c, GraphDep{sv, nullptr, sv->derived});
    }
  }
  if (op->o_state >= 0) {
    CeedVector sv = op->out[op->o_state].vec;
    if (!sv || sv == CEED_VECTOR_NONE || sv == CEED_VECTOR_ACTIVE) return ceed_error("state output needs a passive vector");
    CHK(vec_dev(sv, true, &ps)); a.state_out = ps;  // every point is overwritten (and the derived state invalidated)
    if (qf->kind == QF_HYPERFS_F && c->opt.derived_state && pencil_derived_state(ai.basis->Q1d) && !split) {
      const size_t need = (size_t)r->nelem * 10 * Q3;
      if (sv->derived_len < need) {
        if (c->capturing) return ceed_error("evaluate the residual once before recording (derived-state buffer)");
        ceed_retire(c, sv->derived);
        sv->derived = nullptr; sv->derived_len = 0;
        HIPCHK(hipMalloc((void **)&sv->derived, sizeof(double) * need));
        sv->derived_len = need;
      }
      a.state_out2 = sv->derived;
      F.derived_for = sv;
    }
  }
  a.mask_in = (op->mask_mode & 1) ? 1 : 0; a.mask_out = (op->mask_mode & 2) ? 1 : 0;
  if (pencil_even_odd(ai.basis->Q1d)) memcpy(a.eo, op->eo, sizeof a.eo);
  {  // geometric factors recomputed in the kernel if the qdata vector still is what SetupGeo wrote on these elements
    CeedVector qv = op->in[op->i_qdata].vec;
    bool same_rule = qv->geo && qv->geo_nelem == r->nelem && qv->geo_Q == ai.basis->Q1d;
    for (int i = 0; same_rule && i < ai.basis->Q1d; i++)
      same_rule = qv->geo_qref[i] == ai.basis->qref1d[i] && qv->geo_qwt[i] == ai.basis->qweight1d[i];
    if (same_rule && c->opt.recompute_geo) {
      a.geo = qv->geo;
      a.geo_aff = qv->geo_aff;
      a.geo_swept = qv->geo_swept; a.geo_axis = qv->geo_axis;
      for (int i = 0; i < ai.basis->Q1d; i++) { a.qref[i] = qv->geo_qref[i]; a.qwt[i] = qv->geo_qwt[i]; }
      if (c->capturing) capture_dep(c, GraphDep{qv, qv->geo, nullptr});
    }
  }
  lame_constants(a.nu, a.E, &a.lambda, &a.TwoMu);
  a.waves_per_cu = c->opt.pencil_waves;
  if (split && (add || op->ovl_lead <= 0 || !op->ovl_csr.built))
    return ceed_error("split-phase apply needs CeedXOperatorSetOverlapSplit and overwrite mode");
  a.elem_begin = 0; a.nelem = r->nelem;
  // element-interior nodes straight to y: overwrite mode only (split maps are built to match, see SetOverlapSplit)
  F.direct = !add && c->opt.direct_interior && rstr_interior_private(r, ai.basis->P1d);
  a.direct = F.direct ? 1 : 0;
  // atomic-free, deterministic scatter: element results -> E-vector -> per-node sums
  unsigned char **flagsp;
  if (split) {
    F.M = &op->ovl_csr; flagsp = &op->d_node_flags_ovl;
    if ((F.M->nskipped > 0) != F.direct) return ceed_error("split-phase map and direct-store mode disagree");
  } else if (F.direct) {
    CHK(build_csr(r, r->csr_shell, nullptr, ai.basis->P1d));
    CHK(build_interior_list(r, ai.basis->P1d));
    F.M = &r->csr_shell; flagsp = &op->d_node_flags_shell;
  } else {
    CHK(build_csr(r, r->csr, nullptr));
    F.M = &r->csr; flagsp = &op->d_node_flags;
  }
  if (!*flagsp && !op->h_mask.empty()) {
    int err = 0;
    *flagsp = make_row_flags(op, r, F.M->h_node_off, &err);
    if (err) return err;
  }
  F.flags = (op->mask_mode & 2) ? *flagsp : nullptr;
  a.evec_stride = 3 * (F.direct ? evec_block_records(ai.basis->P1d) : r->elemsize);
  CHK(ceed_need_evec(c, (size_t)r->nelem * std::max((size_t)3 * (size_t)r->elemsize, (size_t)a.evec_stride)));   // (an aligned shell block may exceed P^3 records at small P)
  a.evec = c->evec;
  return 0;
}
// one launch of the fused kernel over elements [e0, e0 + ne)
static int fused_launch(CeedOperator op, const FusedApply &F, int e0, int ne, int wave_groups, hipStream_t s, const char **kname) {
  FusedGradArgs ak = F.a;
  ak.elem_begin = e0; ak.nelem = ne; ak.wave_groups = wave_groups;
  hipError_t e = launch_fused_grad(F.b->P1d, F.b->Q1d, F.qfkind, op->tables, ak, s, kname);
  if (e == hipErrorInvalidValue && !**kname)
    return ceed_error("no fused kernel instantiated for P=%d Q=%d QFunction %s", F.b->P1d, F.b->Q1d, op->qf->name.c_str());
  HIPCHK(e);
  op->geo_mode = F.a.geo_aff && F.a.geo ? 2 : (F.a.geo_swept && F.a.geo ? 3 : (F.a.geo ? 1 : 0));
  return 0;
}
static int assemble_rows(const FusedApply &F, int row0, int nrows, hipStream_t s, int max_blocks = 0, const HaloUnpackArgs *un = nullptr,
                         const HaloPackFold *pk = nullptr) {
  const CsrMap *M = F.M;
  HaloPackFold p0{nullptr, nullptr, nullptr};
  if (pk) p0 = HaloPackFold{pk->ptr + row0, pk->slot, pk->send};
  HIPCHK(launch_assemble(M->d_rowptr + row0, M->d_cols, M->d_node_off + row0, F.flags ? F.flags + row0 : nullptr, F.a.evec, F.py,
                         nrows, F.add ? 1 : 0, s, max_blocks, un, pk ? &p0 : nullptr));
  return 0;
}
// The pack of halo H folded into the launch that sums the rows of map M: per row the send slots of its node's entries.
// Built once per (map, halo); not ok (-> the separate pack kernel) if an entry of the halo is no row of the map.
static int get_pack_fold(CeedOperator op, CeedElemRestriction r, const CsrMap *M, CeedXHalo H, HaloPackFold *out, bool *ok) {
  for (auto &pf : op->pack_folds)
    if (pf.M == M && pf.H == H && pf.serial == H->serial) { *ok = pf.ok; *out = HaloPackFold{pf.d_ptr, pf.d_slot, H->send}; return 0; }
  CeedOperator_private::PackFold pf{M, H, H->serial, nullptr, nullptr, false};
  if (op->ceed->capturing) { *ok = false; return 0; }     // cold while recording: the separate pack kernel
  const int nn = M->nnodes;
  bool good = r->ncomp == 3 && r->compstride == 1 && (size_t)H->total < (1u << 30);
  std::vector<uint32_t> ptr((size_t)nn + 1, 0u), slot((size_t)(H->total ? H->total : 1));
  if (good) {
    // row of a node offset: the map's rows are distinct node offsets (ascending within each priority class): look up by sort
    std::vector<std::pair<uint32_t, uint32_t>> rows((size_t)nn);
    for (int i = 0; i < nn; i++) rows[(size_t)i] = {M->h_node_off[(size_t)i], (uint32_t)i};
    std::sort(rows.begin(), rows.end());
    std::vector<uint32_t> row_of((size_t)H->total);
    for (int k = 0; k < H->total && good; k++) {
      const uint32_t d = H->h_idx[(size_t)k], node = d - d % 3;
      auto it = std::lower_bound(rows.begin(), rows.end(), std::make_pair(node, 0u));
      if (it == rows.end() || it->first != node) good = false;
      else { row_of[(size_t)k] = it->second; ptr[(size_t)it->second + 1]++; }
    }
    if (good) {
      for (int i = 0; i < nn; i++) ptr[(size_t)i + 1] += ptr[(size_t)i];
      std::vector<uint32_t> cur(ptr.begin(), ptr.end() - 1);
      for (int k = 0; k < H->total; k++) slot[cur[row_of[(size_t)k]]++] = (uint32_t)k | ((H->h_idx[(size_t)k] % 3u) << 30);
    }
  }
  if (good) {
    HIPCHK(hipMalloc((void **)&pf.d_ptr, sizeof(uint32_t) * ptr.size()));
    HIPCHK(hipMalloc((void **)&pf.d_slot, sizeof(uint32_t) * slot.size()));
    HIPCHK(hipMemcpy(pf.d_ptr, ptr.data(), sizeof(uint32_t) * ptr.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pf.d_slot, slot.data(), sizeof(uint32_t) * slot.size(), hipMemcpyHostToDevice));
    pf.ok = true;
  }
  op->pack_folds.push_back(pf);
  *ok = pf.ok; *out = HaloPackFold{pf.d_ptr, pf.d_slot, H->send};
  return 0;
}

// Whole apply, pipelined form: segment k's fused kernel AND its rows on stream k % 2 -- the fused kernel of
// segment k + 1 sits in the other queue and fills the chip as the waves of segment k retire (no kernel boundary between
// fused kernels), every k_assemble but the last runs beside a fused kernel.  Every row is summed in contributor order by
// one thread, whatever the segment: bitwise the serial result.
static int apply_pipelined(CeedOperator op, const FusedApply &F, PipeMap *PM, const char **kname, const EpilogueArgs *ep = nullptr) {
  Ceed c = op->ceed;
  hipStream_t s = c->stream;
  CHK(ceed_need_side_stream(c));
  const unsigned char *fl = nullptr;
  if ((op->mask_mode & 2) && !op->h_mask.empty()) {   // Dirichlet flags in this map's row order (made once per map)
    for (auto &pf : op->pipe_flags) if (pf.first == PM) fl = pf.second;
    if (!fl) {
      if (c->capturing) return ceed_error("apply the operator once before recording (Dirichlet flags of the pipelined map)");
      int err = 0;
      unsigned char *d = make_row_flags(op, F.r, PM->h_node_off, &err);
      if (err) return err;
      op->pipe_flags.emplace_back(PM, d);
      fl = d;
    }
  }
  const int nseg = PM->nseg;
  op->launch_info[0] = nseg; op->launch_info[1] = 2; op->launch_info[2] = nseg;
  op->launch_info[3] = PM->elem_bound[nseg] - PM->elem_bound[nseg - 1];
  HIPCHK(hipEventRecord(c->ev_fork, s));
  HIPCHK(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
  for (int k = 0; k < nseg; k++) {
    hipStream_t sk = (k & 1) ? c->side_stream : s;     // (the FIRST segment on the operator's own stream; putting the last one there instead, so that the join is never waited for, measured 6-9 % slower at even segment counts)
    CHK(fused_launch(op, F, PM->elem_bound[k], PM->elem_bound[k + 1] - PM->elem_bound[k], 0, sk, kname));
    // the rows of segment k have contributors in EARLIER segments too (the nodes on the cut between two segments):
    // segment k - 1's fused kernel runs on the other stream, the ones before it precede one of the two in stream order
    if (!c->ev_seg[k]) HIPCHK(hipEventCreateWithFlags(&c->ev_seg[k], hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->ev_seg[k], sk));
    if (k >= 1) HIPCHK(hipStreamWaitEvent(sk, c->ev_seg[k - 1], 0));
    const int r0 = PM->row_bound[k], nr = PM->row_bound[k + 1] - r0;
    if (ep) {   // the consumer of the output in place of its store; the segment's own element-interior nodes ride along
      EpilogueArgs ek = *ep;
      const int nint = F.direct ? F.r->int_per_elem : 0;
      ek.int_off = nint ? F.r->d_int_off + (size_t)PM->elem_bound[k] * nint : nullptr;
      ek.n_int = (PM->elem_bound[k + 1] - PM->elem_bound[k]) * nint;
      HIPCHK(launch_assemble_epi(PM->d_rowptr + r0, PM->d_cols, PM->d_node_off + r0, fl ? fl + r0 : nullptr, F.a.evec, nr, ek, sk,
                                 k + 1 < nseg ? c->opt.pipe_blocks : 0));
    } else
    HIPCHK(launch_assemble(PM->d_rowptr + r0, PM->d_cols, PM->d_node_off + r0, fl ? fl + r0 : nullptr, F.a.evec, F.py, nr, 0,
                           sk, k + 1 < nseg ? c->opt.pipe_blocks : 0));
  }
  HIPCHK(hipEventRecord(c->ev_join, c->side_stream));
  HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
  return 0;
}

// phase -1: whole apply; phase 0 / 1: the two halves of a split-phase apply (CeedXOperatorApplyPhase), one after the other
// on the Ceed's stream.
// `H` (whole applies in overwrite mode only): the interface sum of the output follows IN ORDER on the same stream -- in the
// serial form the pack is folded into the rows' launch (HaloPackFold), then the RCCL group and the unpack-add launch.
static int apply_fused_grad(CeedOperator op, CeedVector in, CeedVector out, bool add, int phase, const char **kname, CeedXHalo H = nullptr) {
  Ceed c = op->ceed;
  hipStream_t s = c->stream;
  FusedApply F;
  const bool split = phase >= 0;
  CHK(fused_prepare(op, in, out, add, split, F));
  const CsrMap *M = F.M;
  if (!add && !M->full_cover && phase <= 0) CHK(dev_zero(c, F.py, (size_t)out->length));
  if (F.derived_for) {   // this (whole) residual apply also writes the tangent's derived state beside grad u: valid from here on in stream order
    CeedVector sv = F.derived_for;
    sv->derived_valid = true; sv->derived_nelem = F.r->nelem; sv->derived_Q3 = F.b->Q1d * F.b->Q1d * F.b->Q1d;
    sv->derived_nu = F.a.nu; sv->derived_E = F.a.E;
  }
  TimerScope ts(op, s);
  if (split) {
    const int lead = op->ovl_lead;
    if (phase == 0) { CHK(fused_launch(op, F, 0, lead, 0, s, kname)); CHK(assemble_rows(F, 0, M->nprio, s)); }
    else { CHK(fused_launch(op, F, lead, F.r->nelem - lead, 0, s, kname)); CHK(assemble_rows(F, M->nprio, M->nnodes - M->nprio, s)); }
    op->launch_info[0] = 1; op->launch_info[1] = 1; op->launch_info[2] = 1; op->launch_info[3] = phase == 0 ? lead : F.r->nelem - lead;
    op->launches++;
    return 0;
  }
  // pipelined assembly: whole applies in overwrite mode, large enough for two segments
  if (c->opt.pipe_segments != 0 && !add) {
    int waves = 0;            // persistent waves of a full launch of THIS kernel (LDS-limited from Q = 6 on)
    {
      FusedGradArgs aq = F.a;
      aq.query_waves = &waves;
      const char *nm = "";
      HIPCHK(launch_fused_grad(F.b->P1d, F.b->Q1d, F.qfkind, op->tables, aq, s, &nm));
      if (waves <= 0) return ceed_error("pipelined assembly: no persistent-wave count for P=%d Q=%d", F.b->P1d, F.b->Q1d);
    }
    const int per_elem = F.direct ? evec_block_records(F.b->P1d) : F.r->elemsize;
    PipeMap *PM = nullptr;
    const bool fs = F.qfkind == QF_HYPERFS_DF || F.qfkind == QF_HYPERFS_DF_DS || F.qfkind == QF_HYPERFS_F;
    const int mb = c->opt.pipe_mb > 0 ? c->opt.pipe_mb : (fs ? 160 : 90);     // MB of E-vector per segment (get_pipe)
    CHK(get_pipe(F.r, *M, pencil_group_elems(F.b->Q1d), per_elem, std::max(c->opt.pipe_segments, 0), waves, mb, &PM));
    if (PM && PM->nseg >= 2) {
      CHK(apply_pipelined(op, F, PM, kname));
      if (H) {
        CHK(halo_pack_and_send(H, F.py, s));
        CHK(halo_wait_arrivals(H, s));
        HIPCHK(launch_halo_unpack_add(halo_unpack_args(H), F.py, s));
      }
      op->launches++;
      return 0;
    }
  }
  op->launch_info[0] = 1; op->launch_info[1] = 1; op->launch_info[2] = 1; op->launch_info[3] = F.r->nelem;
  CHK(fused_launch(op, F, 0, F.r->nelem, 0, s, kname));
  HaloPackFold pk{nullptr, nullptr, nullptr};
  bool folded = false;
  if (H && c->opt.fold_pack) CHK(get_pack_fold(op, F.r, M, H, &pk, &folded));
  CHK(assemble_rows(F, 0, M->nnodes, s, 0, nullptr, folded ? &pk : nullptr));   // timed together with the fused kernel: the launches ARE the operator apply
  if (H) {
    if (folded) CHK(halo_send(H, s)); else CHK(halo_pack_and_send(H, F.py, s));
    CHK(halo_wait_arrivals(H, s));
    HIPCHK(launch_halo_unpack_add(halo_unpack_args(H), F.py, s));
  }
  op->launches++;
  return 0;
}

// The apply with its consumer fused behind it (CeedXOperatorApplyChebyshev / ApplyResidual): the launches of a whole apply in
// overwrite mode, with k_assemble_epi in place of k_assemble -- the shell rows' sums and the element-interior values the fused kernel
// stored into `t` go straight into the Chebyshev step (or the residual), y = t is never written or re-read as a whole.
// *fused = false (nothing launched): the apply is not of that shape (nodes without an element: full_cover) -- the caller runs the two
// steps one after the other.
static int apply_fused_epilogue(CeedOperator op, CeedVector in, CeedVector t, EpilogueArgs ep, const char **kname, bool *fused) {
  Ceed c = op->ceed;
  hipStream_t s = c->stream;
  FusedApply F;
  *fused = false;
  CHK(fused_prepare(op, in, t, false, false, F));
  const CsrMap *M = F.M;
  if (!M->full_cover) return 0;
  const int nint = F.direct ? F.r->int_per_elem : 0;
  if (F.direct && !F.r->d_int_off) return ceed_error("interior-node list of the restriction missing");
  *fused = true;
  ep.t = F.py;
  TimerScope ts(op, s);
  // Serial form by default: with the consumer in the epilogue the rows' launch is no longer light enough to hide beside the next
  // segment's fused kernel, and the fork / join costs inside a replayed graph -- V-cycle at config 4's size: serial 6.05 ms eager and
  // replayed, pipelined 6.10 eager / 6.83 replayed (profiles/r05_ab_experiments.txt item 10).  CEED_MI355X_EPI_PIPELINED=1: pipelined (tests).
  if (c->opt.epi_pipelined && c->opt.pipe_segments != 0) {
    int waves = 0;
    {
      FusedGradArgs aq = F.a;
      aq.query_waves = &waves;
      const char *nm = "";
      HIPCHK(launch_fused_grad(F.b->P1d, F.b->Q1d, F.qfkind, op->tables, aq, s, &nm));
      if (waves <= 0) return ceed_error("pipelined assembly: no persistent-wave count for P=%d Q=%d", F.b->P1d, F.b->Q1d);
    }
    const int per_elem = F.direct ? evec_block_records(F.b->P1d) : F.r->elemsize;
    PipeMap *PM = nullptr;
    const bool fs = F.qfkind == QF_HYPERFS_DF || F.qfkind == QF_HYPERFS_DF_DS || F.qfkind == QF_HYPERFS_F;
    const int mb = c->opt.pipe_mb > 0 ? c->opt.pipe_mb : (fs ? 160 : 90);
    CHK(get_pipe(F.r, *M, pencil_group_elems(F.b->Q1d), per_elem, std::max(c->opt.pipe_segments, 0), waves, mb, &PM));
    if (PM && PM->nseg >= 2) {
      CHK(apply_pipelined(op, F, PM, kname, &ep));
      op->launches++;
      return 0;
    }
  }
  op->launch_info[0] = 1; op->launch_info[1] = 1; op->launch_info[2] = 1; op->launch_info[3] = F.r->nelem;
  CHK(fused_launch(op, F, 0, F.r->nelem, 0, s, kname));
  ep.int_off = nint ? F.r->d_int_off : nullptr;
  ep.n_int = F.r->nelem * nint;
  HIPCHK(launch_assemble_epi(M->d_rowptr, M->d_cols, M->d_node_off, F.flags, F.a.evec, M->nnodes, ep, s));
  op->launches++;
  return 0;
}

// Split-phase apply WITH the interface sum, as one call (CeedXOperatorApplyWithHalo; the library-side form of
// ApplyLocalCeedOp + DMLocalToGlobal(ADD_VALUES), src/matops.c:46,57, on several GPUs).  Two chains:
//   Ceed's stream : fused kernel of the interface-touching elements -> their nodes' rows -> pack -> [RCCL on the comm stream]
//   side stream   : fused kernel of the interior elements (queued right behind the first: it fills the chip beside it and
//                   takes over the slots its waves free) -> the remaining rows + the arrivals of the exchange, ONE launch
// and the join.  Both chains are bitwise the whole apply followed by the exchange (same rows, same order, same sums).
static int apply_fused_with_halo(CeedOperator op, CeedVector in, CeedVector out, CeedXHalo H, const char **kname) {
  Ceed c = op->ceed;
  hipStream_t s = c->stream;
  FusedApply F;
  CHK(fused_prepare(op, in, out, false, true, F));
  if (H->in_flight) return ceed_error("CeedXOperatorApplyWithHalo: an exchange is already in flight");
  if (out->length < H->lsize_min) return ceed_error("CeedXOperatorApplyWithHalo: vector shorter than the halo's indices");
  const CsrMap *M = F.M;
  const int lead = op->ovl_lead, rest = F.r->nelem - lead;
  // Contract of this form: the arrivals are added by the SAME launch that overwrites the non-priority rows, and
  // the exchange starts when only the priority rows are complete -- so every entry of the halo must lie on a priority row of
  // the split map.  Checked once per (operator, halo) on the host.
  if (op->ovl_halo_checked != H->serial) {
    std::vector<uint32_t> prio(M->h_node_off.begin(), M->h_node_off.begin() + M->nprio);
    std::sort(prio.begin(), prio.end());
    for (uint32_t d : H->h_idx)
      if (!std::binary_search(prio.begin(), prio.end(), d - d % 3u))
        return ceed_error("CeedXOperatorApplyWithHalo: entry %u of the halo is not on a priority node of the operator's overlap split "
                          "(CeedXOperatorSetOverlapSplit): its partial sum would be exchanged before it is complete", d);
    op->ovl_halo_checked = H->serial;
  }
  if (!M->full_cover) CHK(dev_zero(c, F.py, (size_t)out->length));
  TimerScope ts(op, s);
  const CeedOptions &o = c->opt;
  const HaloUnpackArgs un = halo_unpack_args(H);
  op->launch_info[0] = 2; op->launch_info[1] = o.ovl_mode == 2 ? 2 : 1; op->launch_info[2] = 2; op->launch_info[3] = rest;
  if (o.ovl_mode != 2) {   // round 2's sequence on one stream
    CHK(fused_launch(op, F, 0, lead, 0, s, kname));
    CHK(assemble_rows(F, 0, M->nprio, s));
    CHK(halo_pack_and_send(H, F.py, s));
    CHK(fused_launch(op, F, lead, rest, 0, s, kname));
    CHK(halo_wait_arrivals(H, s));
    CHK(assemble_rows(F, M->nprio, M->nnodes - M->nprio, s, 0, &un));
    op->launches++;
    return 0;
  }
  CHK(ceed_need_side_stream(c));
  hipStream_t s1 = c->side_stream;
  HIPCHK(hipEventRecord(c->ev_fork, s));
  HIPCHK(hipStreamWaitEvent(s1, c->ev_fork, 0));
  CHK(fused_launch(op, F, 0, lead, o.ovl_groups0, s, kname));
  CHK(fused_launch(op, F, lead, rest, o.ovl_groups1, s1, kname));
  if (!c->ev_seg[0]) HIPCHK(hipEventCreateWithFlags(&c->ev_seg[0], hipEventDisableTiming));
  HIPCHK(hipEventRecord(c->ev_seg[0], s));             // the interface-touching elements also hold interior rows' contributors
  CHK(assemble_rows(F, 0, M->nprio, s));
  CHK(halo_pack_and_send(H, F.py, s));
  HIPCHK(hipStreamWaitEvent(s1, c->ev_seg[0], 0));
  CHK(halo_wait_arrivals(H, s1));
  CHK(assemble_rows(F, M->nprio, M->nnodes - M->nprio, s1, 0, &un));
  HIPCHK(hipEventRecord(c->ev_join, s1));
  HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
  op->launches++;
  return 0;
}

// ---------------------------------------------------------------------------
// The transfer operators in OWNER form (kernels_misc.hip, k_transfer)
// ---------------------------------------------------------------------------
// own_f[e][n] = offset | fine-side Dirichlet flags if element e is the FIRST (in element order) to hold fine node n, else
// 0xFFFFFFFF.  Set-up time, host; rebuilt when the operator's mask changes.
static int transfer_owner_map(CeedOperator op, CeedElemRestriction rf) {
  if (op->d_own_f) return 0;
  Ceed c = op->ceed;
  if (c->capturing) return ceed_error("first apply of a transfer operator during graph capture: apply it once before recording");
  const size_t n = rf->h_offsets.size();
  std::vector<uint32_t> own(n ? n : 1);
  std::vector<unsigned char> seen((size_t)rf->lsize, 0);
  const std::vector<unsigned char> &mk = op->h_mask_fine;
  size_t distinct = 0;
  for (size_t i = 0; i < n; i++) {
    const uint32_t o = (uint32_t)rf->h_offsets[i];
    if (seen[o]) { own[i] = 0xFFFFFFFFu; continue; }
    seen[o] = 1; distinct++;
    uint32_t f = 0;
    if (!mk.empty()) for (int k = 0; k < 3; k++) if (mk[(size_t)o + k]) f |= 1u << k;
    own[i] = o | (f << OFF_FLAG_SHIFT);
  }
  op->own_full_cover = distinct * 3 == (size_t)rf->lsize;
  HIPCHK(hipMalloc((void **)&op->d_own_f, sizeof(uint32_t) * own.size()));
  HIPCHK(hipMemcpy(op->d_own_f, own.data(), sizeof(uint32_t) * own.size(), hipMemcpyHostToDevice));
  return 0;
}
// w = (fine-side scale, CeedXOperatorSetFineScale, or 1) x (local multiplicity of the fine restriction) per fine dof; *w = null
// when every covered entry is 1 (the scale IS 1 / local multiplicity: one rank).  Recomputed when the scale vector was written
// since (CeedVector_private::version) -- with one host read of a counter, so never while recording.
static int transfer_weights(CeedOperator op, CeedElemRestriction rf, const double **w) {
  Ceed c = op->ceed;
  CeedVector sc = op->scale;
  const uint64_t ver = sc ? sc->version : 0;
  if (op->w_ready && op->w_scale == sc && op->w_version == ver) { *w = op->w_unit ? nullptr : op->d_w; return 0; }
  if (c->capturing)
    return ceed_error("transfer operator during graph capture: its fine-side scale was written since the last apply (or this is the first); "
                      "apply the operator once before recording");
  double *psc = nullptr;
  if (sc) CHK(vec_dev(sc, false, &psc));
  const size_t n = (size_t)rf->lsize;
  if (op->w_len < n) {
    ceed_retire(c, op->d_w); op->d_w = nullptr; op->w_len = 0;
    HIPCHK(hipMalloc((void **)&op->d_w, sizeof(double) * (n ? n : 1)));
    op->w_len = n;
  }
  int *d_cnt = nullptr, cnt = 1;
  HIPCHK(hipMalloc((void **)&d_cnt, sizeof(int)));
  HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(int), c->stream));
  CHK(dev_zero(c, op->d_w, n));
  HIPCHK(launch_multiplicity(rf->d_offsets, rf->nelem, rf->elemsize, rf->ncomp, rf->compstride, op->d_w, c->stream));
  HIPCHK(launch_transfer_weights(op->d_w, psc, n, d_cnt, c->stream));
  HIPCHK(hipMemcpyAsync(&cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));      // set-up time only
  (void)hipFree(d_cnt);
  op->w_unit = cnt == 0; op->w_scale = sc; op->w_version = ver; op->w_ready = true;
  if (op->w_unit) { ceed_retire(c, op->d_w); op->d_w = nullptr; op->w_len = 0; }     // (not needed again until the scale is rewritten: 8 B per fine dof given back)
  *w = op->w_unit ? nullptr : op->d_w;
  return 0;
}

static int op_apply_single(CeedOperator op, CeedVector in, CeedVector out, bool add) {
  CHK(op_plan(op));
  CeedQFunction qf = op->qf;
  hipStream_t s = op->ceed->stream;
  const char *kname = "";
  switch (op->plan) {
  case PLAN_FUSED_GRAD:
    CHK(apply_fused_grad(op, in, out, add, -1, &kname));
    break;
  case PLAN_SETUP_GEO: {
    OpField &x = op->in[0];
    if (!in || in->length < x.rstr->lsize) return ceed_error("coordinate vector too short");
    SetupGeoArgs a{};
    double *px, *pq;
    CHK(vec_dev(in, false, &px));
    CHK(vec_dev(out, true, &pq));
    a.off_x = x.rstr->d_offsets; a.xcoord = px; a.qdata = pq; a.nelem = x.rstr->nelem;
    if ((size_t)out->length < (size_t)a.nelem * 10 * x.basis->Q1d * x.basis->Q1d * x.basis->Q1d) return ceed_error("qdata vector too short");
    TimerScope ts(op, s);
    hipError_t e = launch_setup_geo(x.basis->Q1d, op->tables, a, s, &kname);
    if (e == hipErrorInvalidValue && !*kname) return ceed_error("no setup_geo kernel for Q=%d", x.basis->Q1d);
    HIPCHK(e);
    op->launches++;
    // provenance for the fused kernels: trilinear elements (coordinate basis P = 2) -> keep the map coefficients with
    // the qdata vector; operators reading this vector may then recompute the factors instead of streaming them
    if (op->ceed->opt.recompute_geo && !op->ceed->capturing && x.basis->P1d == 2 && x.rstr->elemsize == 8 && x.rstr->ncomp == 3 && x.rstr->compstride == 1) {
      HIPCHK(hipMalloc((void **)&out->geo, sizeof(double) * GEO_NCOEF * (size_t)a.nelem));
      HIPCHK(launch_geo_coeffs(a.off_x, px, out->geo, a.nelem, s));
      out->geo_nelem = a.nelem; out->geo_Q = x.basis->Q1d;
      if (op->ceed->opt.affine_geo) {   // all elements affine (box meshes)?  then dXdx and det J are per-ELEMENT constants
        int *d_cnt = nullptr, cnt = 1;
        HIPCHK(hipMalloc((void **)&out->geo_aff, sizeof(double) * GEO_NAFF * (size_t)a.nelem));
        HIPCHK(hipMalloc((void **)&d_cnt, sizeof(int)));
        HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(int), s));
        HIPCHK(launch_geo_affine(out->geo, out->geo_aff, a.nelem, d_cnt, s));
        HIPCHK(hipMemcpyAsync(&cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));      // set-up time only
        (void)hipFree(d_cnt);
        if (cnt != 0) { (void)hipFree(out->geo_aff); out->geo_aff = nullptr; }   // a mixed mesh takes the general recompute everywhere
      }
      if (!out->geo_aff && op->ceed->opt.swept_geo) {   // every element swept along ONE reference direction (extruded meshes)?
        int *d_cnt = nullptr, cnt[4] = {0, 0, 0, 1};
        HIPCHK(hipMalloc((void **)&out->geo_swept, sizeof(double) * GEO_NSWEPT * (size_t)a.nelem));
        HIPCHK(hipMalloc((void **)&d_cnt, 4 * sizeof(int)));
        HIPCHK(hipMemsetAsync(d_cnt, 0, 4 * sizeof(int), s));
        HIPCHK(launch_geo_swept(out->geo, out->geo_swept, a.nelem, d_cnt, -1, s));     // count: every direction an element qualifies for
        HIPCHK(hipMemcpyAsync(cnt, d_cnt, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));      // set-up time only
        (void)hipFree(d_cnt);
        int axis = -1;
        for (int d = 2; d >= 0; d--) if (cnt[d] == a.nelem) axis = d;      // a direction ALL elements share
        if (axis < 0) { (void)hipFree(out->geo_swept); out->geo_swept = nullptr; }   // no common direction or general hexes: the general recompute
        else { HIPCHK(launch_geo_swept(out->geo, out->geo_swept, a.nelem, nullptr, axis, s)); out->geo_axis = axis; }
      }
      for (int i = 0; i < x.basis->Q1d && i < MAXN1D; i++) { out->geo_qref[i] = x.basis->qref1d[i]; out->geo_qwt[i] = x.basis->qweight1d[i]; }
    }
    break;
  }
  case PLAN_PROLONG:
  case PLAN_RESTRICT: {
    const bool pro = op->plan == PLAN_PROLONG;
    CeedElemRestriction rc = pro ? op->in[0].rstr : op->out[0].rstr, rf = pro ? op->out[0].rstr : op->in[0].rstr;
    CeedBasis b = pro ? op->in[0].basis : op->out[0].basis;
    if (in == out) return ceed_error("in-place operator apply is not supported");
    if (in->length < (pro ? rc : rf)->lsize || out->length < (pro ? rf : rc)->lsize) return ceed_error("transfer vector too short");
    TransferArgs a{};
    double *px, *py;
    CHK(vec_dev(in, false, &px));
    CHK(vec_dev(out, true, &py));
    if (op->scale && op->scale->length < rf->lsize) return ceed_error("scale vector too short");
    // OWNER form (kernels_misc.hip): the fine nodes each element owns, and the weights (null: all 1, the one-rank case)
    CHK(transfer_owner_map(op, rf));
    CHK(transfer_weights(op, rf, &a.w_f));
    // the coarse side's flagged offsets: the input side of a prolongation, the output side of a restriction
    const uint32_t *fc = pro ? op->d_off_flagged_in : op->d_off_flagged_out;
    a.off_c = fc ? fc : rc->d_offsets;
    a.own_f = op->d_own_f;
    a.x = px; a.y = py; a.nelem = rc->nelem; a.add = add ? 1 : 0;
    const int m_in = (op->mask_mode & 1) ? 1 : 0, m_out = (op->mask_mode & 2) ? 1 : 0;
    a.mask_c = pro ? m_in : m_out; a.mask_f = pro ? m_out : m_in;
    if (pro) {
      // every fine node is stored by its owner: no E-vector, no sum
      if (!add && !op->own_full_cover) CHK(dev_zero(op->ceed, py, (size_t)out->length));
    } else {
      // deterministic scatter on the COARSE side (Pc^3 nodes per element): element results -> E-vector -> per-node sums in
      // element order over the coarse restriction's transpose map (masked entries travel as zeros)
      CHK(build_csr(rc, rc->csr, nullptr));
      CHK(ceed_need_evec(op->ceed, (size_t)rc->nelem * rc->ncomp * rc->elemsize));
      a.evec = op->ceed->evec;
      if (!add && !rc->csr.full_cover) CHK(dev_zero(op->ceed, py, (size_t)out->length));
    }
    TimerScope ts(op, s);
    hipError_t e = launch_transfer(b->P1d, b->Q1d, pro, op->tables, a, s, &kname);
    if (e == hipErrorInvalidValue && !*kname) return ceed_error("no transfer kernel for Pc=%d Pf=%d", b->P1d, b->Q1d);
    HIPCHK(e);
    if (!pro) HIPCHK(launch_assemble(rc->csr.d_rowptr, rc->csr.d_cols, rc->csr.d_node_off, nullptr, a.evec, py, rc->csr.nnodes, add ? 1 : 0, s));
    op->launches++;
    break;
  }
  case PLAN_ENERGY: {
    OpField &u = op->in[0], &en = op->out[0];
    if (!in || in->length < u.rstr->lsize || !out || out->length < en.rstr->lsize) return ceed_error("displacement / energy vector too short");
    EnergyOpArgs a{};
    double *pu, *py, *pq;
    CHK(vec_dev(in, false, &pu)); CHK(vec_dev(out, true, &py)); CHK(vec_dev(op->in[op->i_qdata].vec, false, &pq));
    a.off_u = u.rstr->d_offsets; a.u = pu; a.off_e = en.rstr->d_offsets; a.y = py; a.qdata = pq;
    a.nelem = u.rstr->nelem; a.Q = u.basis->Q1d; a.P = u.basis->P1d;
    const int kd = qf->kind;
    a.diag = (kd == QF_DIAG_LINELAS || kd == QF_DIAG_HYPERSS || kd == QF_DIAG_HYPERFS) ? 1 : 0;
    a.model = (kd == QF_ENERGY_LINELAS || kd == QF_DIAG_LINELAS) ? 0 : ((kd == QF_ENERGY_HYPERSS || kd == QF_DIAG_HYPERSS) ? 1 : 2);
    CHK(read_phys(qf, &a.nu, &a.E));
    memcpy(a.interp, u.basis->interp1d.data(), sizeof(double) * u.basis->interp1d.size());
    memcpy(a.grad, u.basis->grad1d.data(), sizeof(double) * u.basis->grad1d.size());
    if (!a.diag) memcpy(a.interp_e, en.basis->interp1d.data(), sizeof(double) * en.basis->interp1d.size());
    if (!add) CHK(dev_zero(op->ceed, py, (size_t)out->length));
    hipError_t e = launch_energy_op(a, s);
    if (e == hipErrorInvalidValue) return ceed_error("energy operator: Q=%d / P=%d outside the supported range", a.Q, a.P);
    HIPCHK(e);
    kname = a.diag ? (a.model == 0 ? "diagnostic_op<LinElasDiagnostic>" : (a.model == 1 ? "diagnostic_op<HyperSSDiagnostic>" : "diagnostic_op<HyperFSDiagnostic>"))
                   : (a.model == 0 ? "energy_op<LinElasEnergy>" : (a.model == 1 ? "energy_op<HyperSSEnergy>" : "energy_op<HyperFSEnergy>"));
    op->launches++;
    break;
  }
  case PLAN_COORD: {
    OpField &x = op->in[0], &o = op->out[0];
    if (!in || in->length < x.rstr->lsize || !out || out->length < o.rstr->lsize) return ceed_error("coordinate / output vector too short");
    CoordOpArgs a{};
    double *px, *py, *pq = nullptr;
    CHK(vec_dev(in, false, &px)); CHK(vec_dev(out, true, &py));
    a.off_x = x.rstr->d_offsets; a.xcoord = px; a.off_u = o.rstr->d_offsets; a.y = py;
    a.nelem = x.rstr->nelem; a.Q = x.basis->Q1d;
    a.mode = qf->kind == QF_CONST_FORCE ? 0 : (qf->kind == QF_MMS_FORCE ? 1 : 2);
    if (a.mode != 2) {
      CHK(vec_dev(op->in[1].vec, false, &pq)); a.qdata = pq;
      a.Pout = o.basis->P1d;
      memcpy(a.bu, o.basis->interp1d.data(), sizeof(double) * o.basis->interp1d.size());
      if (!qf->ctx) return ceed_error("QFunction '%s' needs its context", qf->name.c_str());
      const double *cx = (const double *)qf->ctx;   // pointer pass-through: forcing vector (3) or Physics {nu, E} (setuplibceed.c:563-566)
      for (int i = 0; i < (a.mode == 0 ? 3 : 2); i++) a.ctx[i] = cx[i];
    } else {
      a.Pout = a.Q;
    }
    memcpy(a.bx, x.basis->interp1d.data(), sizeof(double) * x.basis->interp1d.size());
    if (!add) CHK(dev_zero(op->ceed, py, (size_t)out->length));
    hipError_t e = launch_coord_op(a, s);
    if (e == hipErrorInvalidValue) return ceed_error("coordinate operator: Q=%d / P=%d outside the supported range", a.Q, a.Pout);
    HIPCHK(e);
    kname = a.mode == 2 ? "coord_op<MMSTrueSoln>" : (a.mode == 1 ? "coord_op<SetupMMSForce>" : "coord_op<SetupConstantForce>");
    op->launches++;
    break;
  }
  default: return ceed_error("operator has no plan");
  }
  op->kernel_name = kname;
  return 0;
}

extern "C" int CeedOperatorApply(CeedOperator op, CeedVector in, CeedVector out, CeedRequest *) {
  if (op->composite) {
    CHK(CeedVectorSetValue(out, 0.));
    for (CeedOperator s : op->sub) CHK(op_apply_single(s, in, out, true));
    return 0;
  }
  return op_apply_single(op, in, out, false);
}
extern "C" int CeedOperatorApplyAdd(CeedOperator op, CeedVector in, CeedVector out, CeedRequest *) {
  if (op->composite) { for (CeedOperator s : op->sub) CHK(op_apply_single(s, in, out, true)); return 0; }
  return op_apply_single(op, in, out, true);
}

extern "C" int CeedOperatorLinearAssembleDiagonal(CeedOperator op, CeedVector assembled, CeedRequest *) {
  if (op->composite) return ceed_error("diagonal of a composite operator not supported");
  CHK(op_plan(op));
  if (op->plan != PLAN_FUSED_GRAD || op->o_state >= 0) return ceed_error("diagonal assembly is provided for the Jacobian operators");
  CeedQFunction qf = op->qf;
  hipStream_t s = op->ceed->stream;
  OpField &ai = op->in[op->i_active];
  DiagArgs a{};
  double *pd, *pq, *ps = nullptr;
  CHK(vec_dev(assembled, true, &pd));
  CHK(vec_dev(op->in[op->i_qdata].vec, false, &pq));
  if (op->i_state >= 0) CHK(vec_dev(op->in[op->i_state].vec, false, &ps));
  if (assembled->length < ai.rstr->lsize) return ceed_error("diagonal vector too short");
  a.offsets = op->d_off_flagged_in ? op->d_off_flagged_in : ai.rstr->d_offsets;
  a.diag = pd; a.qdata = pq; a.state_in = ps; a.nelem = ai.rstr->nelem; a.mask_out = (op->mask_mode & 2) ? 1 : 0;
  CHK(read_phys(qf, &a.nu, &a.E));
  lame_constants(a.nu, a.E, &a.lambda, &a.TwoMu);
  CHK(dev_zero(op->ceed, pd, (size_t)assembled->length));  // overwrite semantics (matops.c:227)
  // deterministic: element contributions -> E-vector -> per-node sums in element order
  CHK(build_csr(ai.rstr, ai.rstr->csr, nullptr));
  CHK(ceed_need_evec(op->ceed, (size_t)ai.rstr->nelem * ai.rstr->ncomp * ai.rstr->elemsize));
  a.evec = op->ceed->evec;
  const char *kname = "";
  hipError_t e = launch_diag(ai.basis->P1d, ai.basis->Q1d, qf->kind, op->tables, a, s, &kname);
  if (e == hipErrorInvalidValue && !*kname) return ceed_error("no diagonal kernel for P=%d Q=%d %s", ai.basis->P1d, ai.basis->Q1d, qf->name.c_str());
  HIPCHK(e);
  HIPCHK(launch_assemble(ai.rstr->csr.d_rowptr, ai.rstr->csr.d_cols, ai.rstr->csr.d_node_off, nullptr, a.evec, pd,
                         ai.rstr->csr.nnodes, 0, s));
  return 0;
}

// ---------------------------------------------------------------------------
// extensions
// ---------------------------------------------------------------------------
// the instantiation of the last apply; for the fused operators also how the geometric factors were obtained
extern "C" int CeedXOperatorGetKernelName(CeedOperator op, const char **name) {
  if (op->plan == PLAN_FUSED_GRAD && !op->kernel_name.empty() && op->kernel_name.find(" [") == std::string::npos)
    op->kernel_name += op->geo_mode == 2 ? " [affine elements: dXdx per element]" : (op->geo_mode == 3 ? " [swept elements: 2 x 2 dXdx recomputed per point]" : (op->geo_mode == 1 ? " [dXdx recomputed per point]" : " [qdata read]"));
  *name = op->kernel_name.c_str();
  return 0;
}

static int make_flagged(CeedElemRestriction r, const unsigned char *mask, CeedInt lsize, uint32_t **dev) {
  if (lsize < r->lsize) return ceed_error("Dirichlet mask shorter than the L-vector");
  std::vector<uint32_t> fl(r->h_offsets.size());
  for (size_t i = 0; i < fl.size(); i++) {
    uint32_t o = (uint32_t)r->h_offsets[i], f = 0;
    for (int c = 0; c < r->ncomp && c < 3; c++) if (mask[(size_t)o + (size_t)c * r->compstride]) f |= 1u << c;
    fl[i] = o | (f << OFF_FLAG_SHIFT);
  }
  HIPCHK(hipMalloc((void **)dev, sizeof(uint32_t) * (fl.size() ? fl.size() : 1)));
  HIPCHK(hipMemcpy(*dev, fl.data(), sizeof(uint32_t) * fl.size(), hipMemcpyHostToDevice));
  return 0;
}
// mode: 1 = masked entries read as zero, 2 = masked rows dropped, 3 = both (default for mode 0)
extern "C" int CeedXOperatorSetDirichletMaskMode(CeedOperator op, CeedMemType mtype, const unsigned char *mask,
                                                 CeedInt lsize, const unsigned char *mask_out, CeedInt lsize_out, int mode) {
  if (op->composite) return ceed_error("set the mask on the sub-operators");
  CHK(op_plan(op));
  op_free_flags(op);
  if (!mask && !mask_out) return 0;
  if (mtype != CEED_MEM_HOST) return ceed_error("pass the Dirichlet mask in host memory (it is folded into the offsets once)");
  if (op->plan == PLAN_FUSED_GRAD) {
    CHK(make_flagged(op->in[op->i_active].rstr, mask, lsize, &op->d_off_flagged_in));
    op->d_off_flagged_out = op->d_off_flagged_in;
    op->h_mask.assign(mask, mask + lsize);
  } else if (op->plan == PLAN_PROLONG || op->plan == PLAN_RESTRICT) {
    if (!mask || !mask_out) return ceed_error("transfer operators need the input-side and the output-side mask");
    // the COARSE side's flags ride in its offsets (input of a prolongation, output of a restriction); the FINE side's in the
    // owner map (transfer_owner_map), rebuilt at the next apply
    const bool pro = op->plan == PLAN_PROLONG;
    if (pro) CHK(make_flagged(op->in[0].rstr, mask, lsize, &op->d_off_flagged_in));
    else CHK(make_flagged(op->out[0].rstr, mask_out, lsize_out, &op->d_off_flagged_out));
    CeedElemRestriction rf = pro ? op->out[0].rstr : op->in[0].rstr;
    if ((pro ? lsize_out : lsize) < rf->lsize) return ceed_error("Dirichlet mask shorter than the L-vector");
    const unsigned char *mf = pro ? mask_out : mask;
    op->h_mask_fine.assign(mf, mf + rf->lsize);
  } else return ceed_error("this operator takes no Dirichlet mask");
  op->mask_mode = mode ? mode : 3;
  return 0;
}
extern "C" int CeedXOperatorSetDirichletMask(CeedOperator op, CeedMemType mtype, const unsigned char *mask, CeedInt lsize) {
  return CeedXOperatorSetDirichletMaskMode(op, mtype, mask, lsize, nullptr, 0, 3);
}
// Fine-side multiplicity scale of the transfer operators (matops.c:149,176); NULL clears.
extern "C" int CeedXOperatorSetFineScale(CeedOperator op, CeedVector scale) {
  CeedVectorDestroy(&op->scale);
  op->w_ready = false;
  if (scale && scale != CEED_VECTOR_NONE) { op->scale = scale; scale->refcount++; }
  return 0;
}
extern "C" int CeedXOperatorSetTiming(CeedOperator op, int enable) {
  op->timing = enable != 0;
  for (auto &ev : op->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  op->events.clear(); op->ms_accum = 0.; op->launches = 0;
  return 0;
}
extern "C" int CeedXOperatorGetTiming(CeedOperator op, double *ms, int64_t *launches) {
  for (auto &ev : op->events) {
    float t = 0.f;
    HIPCHK(hipEventSynchronize(ev.second));
    HIPCHK(hipEventElapsedTime(&t, ev.first, ev.second));
    op->ms_accum += t;
    (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second);
  }
  op->events.clear();
  *ms = op->ms_accum; *launches = op->launches;
  return 0;
}

// Split-phase apply for communication overlap (the halo sum of matops.c:57 hidden under the interior
// elements): the first `n_leading_elems` elements must be the ONLY contributors of the nodes flagged in
// `priority` (one byte per L-vector entry, read at each node's first component).  Phase 0 computes those
// elements and finishes the flagged nodes; phase 1 does the rest.  Phase 0 then 1 == CeedOperatorApply.
extern "C" int CeedXOperatorSetOverlapSplit(CeedOperator op, CeedInt n_leading_elems, const unsigned char *priority,
                                            CeedInt lsize) {
  if (op->composite) return ceed_error("set the overlap split on the sub-operators");
  CHK(op_plan(op));
  if (op->plan != PLAN_FUSED_GRAD) return ceed_error("overlap split is provided for the residual / Jacobian operators");
  CeedElemRestriction r = op->in[op->i_active].rstr;
  op->ovl_csr.release();
  op->ovl_halo_checked = 0;
  if (op->d_node_flags_ovl) { (void)hipFree(op->d_node_flags_ovl); op->d_node_flags_ovl = nullptr; }
  op->ovl_lead = 0;
  if (!priority) return 0;
  if (lsize < r->lsize || n_leading_elems < 0 || n_leading_elems > r->nelem) return ceed_error("bad overlap split arguments");
  // check the contract: every contributor of a priority node is a leading element
  const size_t es = (size_t)r->elemsize;
  for (size_t i = 0; i < r->h_offsets.size(); i++)
    if (priority[(size_t)r->h_offsets[i]] && i / es >= (size_t)n_leading_elems)
      return ceed_error("element %zu touches a priority node but is not among the %d leading elements", i / es, n_leading_elems);
  const int P1 = op->in[op->i_active].basis->P1d;
  CHK(build_csr(r, op->ovl_csr, priority, (op->ceed->opt.direct_interior && rstr_interior_private(r, P1)) ? P1 : 0));
  op->ovl_lead = n_leading_elems;
  return 0;
}
extern "C" int CeedXOperatorApplyPhase(CeedOperator op, CeedVector in, CeedVector out, int phase) {
  if (op->composite) return ceed_error("split-phase apply of a composite operator is not supported");
  CHK(op_plan(op));
  if (op->plan != PLAN_FUSED_GRAD || (phase != 0 && phase != 1)) return ceed_error("bad split-phase apply");
  const char *kname = "";
  CHK(apply_fused_grad(op, in, out, false, phase, &kname));
  op->kernel_name = kname;
  return 0;
}

extern "C" int CeedXOperatorGetLaunchInfo(CeedOperator op, int out[4]) {
  for (int i = 0; i < 4; i++) out[i] = op->launch_info[i];
  return 0;
}
// The split-phase apply and the interface sum of its output in ONE call: phase 0, the exchange started, phase 1 beside it,
// the arrivals added (apply_fused_with_halo).  `halo` with no neighbours: a plain apply.
extern "C" int CeedXOperatorApplyWithHalo(CeedOperator op, CeedVector in, CeedVector out, CeedXHalo halo) {
  if (op->composite) return ceed_error("split-phase apply of a composite operator is not supported");
  CHK(op_plan(op));
  if (!halo || halo->nb.empty()) return CeedOperatorApply(op, in, out, CEED_REQUEST_IMMEDIATE);
  // Default (CeedOptions::ovl_mode 0): the whole apply, then pack / RCCL group / unpack-add, IN ORDER on the Ceed's stream.
  // Measured on the emulated rank 3 of 8: 13 200 hexes at p = 4 -- split-phase on two streams 119 us, split-phase on
  // one stream 112 us, whole apply + exchange through the communicator's stream 120 us; a hand-over between two streams costs
  // more than the exchange it would hide.
  // (while a graph is recorded always this form: RCCL's calls record correctly only in order on the capturing stream)
  if (op->plan == PLAN_FUSED_GRAD && (op->ceed->opt.ovl_mode == 0 || op->ceed->capturing || op->ovl_lead <= 0 || !op->ovl_csr.built)) {
    if (halo->in_flight) return ceed_error("CeedXOperatorApplyWithHalo: an exchange is already in flight");
    if (out->length < halo->lsize_min) return ceed_error("CeedXOperatorApplyWithHalo: vector shorter than the halo's indices");
    const char *kname = "";
    CHK(apply_fused_grad(op, in, out, false, -1, &kname, halo));
    op->kernel_name = kname;
    return 0;
  }
  if (op->plan != PLAN_FUSED_GRAD) {
    CHK(CeedOperatorApply(op, in, out, CEED_REQUEST_IMMEDIATE));
    CHK(CeedXHaloStart(halo, out));
    return CeedXHaloFinish(halo, out);
  }
  const char *kname = "";
  CHK(apply_fused_with_halo(op, in, out, halo, &kname));
  op->kernel_name = kname;
  return 0;
}

// ---------------------------------------------------------------------------
// The apply fused with its consumer (include/ceed.h; elasticity.c:539-552, 588-590)
// ---------------------------------------------------------------------------
static int epi_check(CeedOperator op, CeedVector in, CeedVector t, const char *who) {
  if (op->composite) return ceed_error("%s: not provided for composite operators", who);
  CHK(op_plan(op));
  if (op->plan != PLAN_FUSED_GRAD || op->o_state >= 0) return ceed_error("%s is provided for the Jacobian operators", who);
  if (!in || !t || in == t) return ceed_error("%s: the scratch vector t must be a vector of its own", who);
  return 0;
}
extern "C" int CeedXOperatorApplyChebyshev(CeedOperator op, CeedVector in, CeedVector t, CeedVector x, CeedVector d, CeedVector r,
                                           CeedVector b, CeedVector dinv, double c1, double c2, int assign_x) {
  CHK(epi_check(op, in, t, "CeedXOperatorApplyChebyshev"));
  const bool first = b && b != CEED_VECTOR_NONE, has_r = r && r != CEED_VECTOR_NONE;
  const CeedInt n = x->length;
  if (d->length != n || (has_r && r->length != n) || dinv->length != n || t->length != n || in->length != n || (first && b->length != n))
    return ceed_error("CeedXOperatorApplyChebyshev: vector lengths differ");
  if (!first && !has_r) return ceed_error("CeedXOperatorApplyChebyshev: a residual vector r or a right-hand side b is needed");
  if (first && ((has_r && b == r) || b == x || b == d)) return ceed_error("CeedXOperatorApplyChebyshev: the right-hand side must be a vector of its own");
  if (t == x || t == d || (has_r && t == r) || t == dinv || (first && t == b)) return ceed_error("CeedXOperatorApplyChebyshev: the scratch vector t aliases an operand");
  EpilogueArgs ep{};
  ep.kind = EPI_CHEB;
  double *pb = nullptr, *pi;
  CHK(vec_dev(dinv, false, &pi));
  if (first) CHK(vec_dev(b, false, &pb));
  if (has_r) CHK(vec_dev(r, true, &ep.r));
  CHK(vec_dev(d, true, &ep.d)); CHK(vec_dev(x, true, &ep.x));
  ep.r0 = pb; ep.dinv = pi; ep.c1 = c1; ep.c2 = c2; ep.assign_x = assign_x;
  const char *kname = "";
  bool fused = false;
  CHK(apply_fused_epilogue(op, in, t, ep, &kname, &fused));
  if (!fused) {     // a restriction that leaves L-vector entries without an element: the two steps, one after the other
    CHK(apply_fused_grad(op, in, t, false, -1, &kname));
    double *pt;
    CHK(vec_dev(t, false, &pt));
    HIPCHK(launch_cheb_update(ep.x, ep.d, ep.r, ep.r0, pt, ep.dinv, c1, c2, assign_x, (size_t)n, op->ceed->stream));
  }
  op->kernel_name = kname;
  return 0;
}
extern "C" int CeedXOperatorApplyResidual(CeedOperator op, CeedVector in, CeedVector t, CeedVector b, CeedVector w) {
  CHK(epi_check(op, in, t, "CeedXOperatorApplyResidual"));
  const CeedInt n = w->length;
  if (b->length != n || t->length != n || in->length != n) return ceed_error("CeedXOperatorApplyResidual: vector lengths differ");
  if (t == b || t == w) return ceed_error("CeedXOperatorApplyResidual: the scratch vector t aliases an operand");
  EpilogueArgs ep{};
  ep.kind = EPI_RESID;
  double *pb;
  CHK(vec_dev(b, false, &pb));
  CHK(vec_dev(w, true, &ep.w));
  ep.b = pb;
  const char *kname = "";
  bool fused = false;
  CHK(apply_fused_epilogue(op, in, t, ep, &kname, &fused));
  if (!fused) {
    CHK(apply_fused_grad(op, in, t, false, -1, &kname));
    double *pt;
    CHK(vec_dev(t, false, &pt));
    HIPCHK(launch_waxpby(ep.w, 1.0, pb, -1.0, pt, (size_t)n, op->ceed->stream));
  }
  op->kernel_name = kname;
  return 0;
}
