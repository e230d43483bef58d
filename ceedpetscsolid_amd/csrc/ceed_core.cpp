// ceed_core.cpp -- host side of the MI355X backend behind include/ceed.h: errors, the Ceed object and its options,
// hipGraph capture, CeedVector and the vector helpers that stand in for the PETSc Vec calls of src/matops.c.
//
// Resource "/gpu/hip/mi355x".  Objects are reference counted exactly as the reference expects (operators keep their
// qfunction / restrictions / bases / passive vectors alive after the creator destroys its handles, e.g.
// setuplibceed.c:392-393).  There is NO host fallback: a missing GPU is a loud error (CeedInit).
// ONE DEVICE PER PROCESS: the library caches device properties process-wide (one rank per GPU, as under torchrun / mpirun).
#include "ceed_impl.hpp"

using namespace cps;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static int g_err_return = 0;
static thread_local char g_err_msg[1024] = "";

int ceed_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err_msg, sizeof g_err_msg, fmt, ap);
  va_end(ap);
  if (!g_err_return) {
    fprintf(stderr, "[ceed mi355x] error: %s\n", g_err_msg);
    abort();
  }
  return 1;
}
extern "C" int CeedXSetErrorReturn(int enable) { g_err_return = enable; return 0; }
extern "C" const char *CeedXLastError(void) { return g_err_msg; }

// sentinels
static CeedVector_private s_vec_active, s_vec_none;
static CeedElemRestriction_private s_rstr_none;
static CeedBasis_private s_basis_colloc;
static CeedQFunction_private s_qf_none;
static CeedRequest s_req_immediate, s_req_ordered;
extern "C" {
const CeedVector CEED_VECTOR_ACTIVE = &s_vec_active;
const CeedVector CEED_VECTOR_NONE = &s_vec_none;
const CeedElemRestriction CEED_ELEMRESTRICTION_NONE = &s_rstr_none;
const CeedBasis CEED_BASIS_COLLOCATED = &s_basis_colloc;
const CeedQFunction CEED_QFUNCTION_NONE = &s_qf_none;
CeedRequest *const CEED_REQUEST_IMMEDIATE = &s_req_immediate;
CeedRequest *const CEED_REQUEST_ORDERED = &s_req_ordered;
const CeedInt CEED_STRIDES_BACKEND[3] = {-1, -1, -1};
const char *const CeedMemTypes[] = {"host", "device"};
}

// ---------------------------------------------------------------------------
// Ceed
// ---------------------------------------------------------------------------
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; }
static bool env_is(const char *name, const char *val) { const char *e = getenv(name); return e && !strcmp(e, val); }

extern "C" int CeedInit(const char *resource, Ceed *ceed) {
  if (!resource || strncmp(resource, "/gpu/hip", 8))
    return ceed_error("this library serves /gpu/hip/mi355x only (got '%s'); there is no CPU path",
                      resource ? resource : "(null)");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1)
    return ceed_error("no HIP device visible (%s): the MI355X backend cannot run",
                      e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  Ceed c = new Ceed_private;
  c->resource = "/gpu/hip/mi355x";
  HIPCHK(hipGetDevice(&c->device));
  // every switch is read HERE, once (ceed_impl.hpp, CeedOptions): nothing on an apply path touches the environment
  CeedOptions &o = c->opt;
  o.recompute_geo = !env_is("CEED_MI355X_GEO", "0");
  o.direct_interior = !env_is("CEED_MI355X_DIRECT", "0");
  o.affine_geo = !env_is("CEED_MI355X_AFFINE", "0");
  o.swept_geo = !env_is("CEED_MI355X_SWEPT", "0");
  o.derived_state = !env_is("CEED_MI355X_DERIVED", "0");
  if (env_is("CEED_MI355X_ASSEMBLE", "serial")) o.pipe_segments = 0;
  else { const int ps = env_int("CEED_MI355X_PIPE_SEGMENTS", 0); o.pipe_segments = ps >= 2 ? std::min(ps, 16) : -1; }
  o.pipe_blocks = env_int("CEED_MI355X_PIPE_BLOCKS", 0);
  o.pipe_mb = std::max(0, env_int("CEED_MI355X_PIPE_MB", o.pipe_mb));
  o.pipe_last_rounds = std::max(0, env_int("CEED_MI355X_PIPE_LAST", o.pipe_last_rounds));
  o.pipe_min_total_rounds = std::max(0, env_int("CEED_MI355X_PIPE_MIN_TOTAL", o.pipe_min_total_rounds));
  o.pipe_min_rounds = std::max(0, env_int("CEED_MI355X_PIPE_MIN_ROUNDS", o.pipe_min_rounds));
  o.pipe_debug = getenv("CEED_MI355X_PIPE_DEBUG") != nullptr;
  o.graph_memset = env_int("CEED_MI355X_GRAPH_MEMSET", 0) != 0;
  o.pencil_waves = std::max(0, env_int("CEED_MI355X_PENCIL_WAVES", 0));
  o.ovl_mode = env_int("CEED_MI355X_OVL_MODE", o.ovl_mode);
  o.ovl_groups0 = std::max(0, env_int("CEED_MI355X_OVL_G0", o.ovl_groups0));
  o.ovl_groups1 = std::max(0, env_int("CEED_MI355X_OVL_G1", o.ovl_groups1));
  o.comm_priority = env_int("CEED_MI355X_COMM_PRIO", o.comm_priority);
  o.comm_inline = env_int("CEED_MI355X_COMM_INLINE", o.comm_inline);
  o.fold_pack = env_int("CEED_MI355X_FOLD_PACK", o.fold_pack);
  o.spgemm_row = !env_is("CEED_MI355X_SPGEMM", "entry");
  o.spmv_stream = !env_is("CEED_MI355X_SPMV", "vector");
  o.epi_pipelined = env_int("CEED_MI355X_EPI_PIPELINED", 0) != 0;
  *ceed = c;
  return 0;
}
void ceed_ref(Ceed c) { c->refcount++; }
int (*g_rccl_comm_destroy)(void *) = nullptr;   // set when RCCL is bound (ceed_halo.cpp)
static void ceed_free_parked(Ceed c) {
  for (double *p : c->evec_parked) (void)hipFree(p);
  for (void *p : c->parked_misc) (void)hipFree(p);
  c->evec_parked.clear(); c->parked_misc.clear();
}
void ceed_retire(Ceed c, void *p) {
  if (!p) return;
  if (c->capturing || c->live_graphs > 0) c->parked_misc.push_back(p);
  else { (void)hipStreamSynchronize(c->stream); (void)hipFree(p); }
}
void ceed_unref(Ceed c) {
  if (--c->refcount > 0) return;
  if (c->comm) {   // the exchange's stream must have drained before the communicator goes
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (g_rccl_comm_destroy) g_rccl_comm_destroy(c->comm);
  }
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  if (c->capture_stream) (void)hipStreamDestroy(c->capture_stream);
  if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
  if (c->evec) (void)hipFree(c->evec);
  ceed_free_parked(c);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  for (hipEvent_t ev : c->ev_seg) if (ev) (void)hipEventDestroy(ev);
  if (c->d_scalar) (void)hipFree(c->d_scalar);
  if (c->h_scalar) (void)hipHostFree(c->h_scalar);
  delete c;
}
int ceed_need_side_stream(Ceed c) {
  if (c->side_stream) return 0;
  // (A side stream created with hipExtStreamCreateWithCUMask, leaving one or two CUs per XCD to RCCL's kernel, was measured in round 3:
  // every kernel on the masked queue ran 2-3x longer with 100-us gaps, 650-800 us per apply instead of 108: profiles/r03_ab_experiments.txt.)
  HIPCHK(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  return 0;
}
extern "C" int CeedDestroy(Ceed *ceed) {
  if (!ceed || !*ceed) return 0;
  ceed_unref(*ceed);
  *ceed = nullptr;
  return 0;
}
extern "C" int CeedGetResource(Ceed ceed, const char **resource) { *resource = ceed->resource.c_str(); return 0; }
extern "C" int CeedGetPreferredMemType(Ceed, CeedMemType *type) { *type = CEED_MEM_DEVICE; return 0; }
extern "C" int CeedXSetStream(Ceed ceed, void *s) { ceed->stream = (hipStream_t)s; return 0; }
extern "C" int CeedXSynchronize(Ceed ceed) {
  if (ceed->capturing) return ceed_error("CeedXSynchronize during graph capture");
  HIPCHK(hipStreamSynchronize(ceed->stream));
  return 0;
}
// Shader clock (GHz) while the work already queued on the Ceed's stream runs: a one-wave probe on a stream of its own counts shader
// cycles against the constant 100 MHz counter for `spin_us` microseconds.  Blocks the host until the probe (not the queue) is done.
extern "C" int CeedXClockProbe(Ceed ceed, int spin_us, double *ghz) {
  if (ceed->capturing) return ceed_error("CeedXClockProbe during graph capture");
  if (spin_us < 1 || spin_us > 1000000) return ceed_error("CeedXClockProbe: 1 us ... 1 s");
  hipStream_t ps = nullptr;
  long long *d = nullptr, h[2] = {0, 0};
  HIPCHK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  HIPCHK(hipMalloc((void **)&d, 2 * sizeof(long long)));
  HIPCHK(launch_clock_probe(d, spin_us, ps));
  HIPCHK(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ps));
  HIPCHK(hipStreamSynchronize(ps));
  (void)hipFree(d); (void)hipStreamDestroy(ps);
  *ghz = h[1] > 0 ? (double)h[0] / ((double)h[1] * 10.0) : 0.;     // cycles / (ticks x 10 ns) = GHz
  return 0;
}
// the references capture_dep() took on the vectors a recording depends on, given back when the recording is dropped
static void release_capture_deps(Ceed ceed) {
  std::vector<GraphDep> deps;
  deps.swap(ceed->capture_deps);
  for (GraphDep &d : deps) { CeedVector v = d.v; (void)CeedVectorDestroy(&v); }
}
extern "C" int CeedXGraphBeginCapture(Ceed ceed) {
  if (ceed->capturing) return ceed_error("graph capture already in progress");
  HIPCHK(hipStreamSynchronize(ceed->stream));
  if (!ceed->capture_stream) HIPCHK(hipStreamCreateWithFlags(&ceed->capture_stream, hipStreamNonBlocking));
  ceed->saved_stream = ceed->stream;
  ceed->stream = ceed->capture_stream;
  HIPCHK(hipStreamBeginCapture(ceed->stream, hipStreamCaptureModeRelaxed));
  ceed->capturing = true;
  release_capture_deps(ceed);
  return 0;
}
extern "C" int CeedXGraphEndCapture(Ceed ceed, CeedXGraph *graph) {
  if (!ceed->capturing) return ceed_error("no graph capture in progress");
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(ceed->stream, &g);
  ceed->stream = ceed->saved_stream;
  ceed->capturing = false;
  if (e != hipSuccess || !g) { release_capture_deps(ceed); return ceed_error("graph capture failed: %s", hipGetErrorString(e)); }
  CeedXGraph G = new CeedXGraph_private;
  G->deps.swap(ceed->capture_deps);                // (the references were taken when the dependencies were recorded: the vectors outlive the graph that checks them)
  G->ceed = ceed; G->graph = g;
  (void)hipGraphGetNodes(g, nullptr, &G->nodes);
  e = hipGraphInstantiate(&G->exec, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    for (GraphDep &d : G->deps) { CeedVector v = d.v; (void)CeedVectorDestroy(&v); }
    delete G;
    return ceed_error("hipGraphInstantiate: %s", hipGetErrorString(e));
  }
  ceed_ref(ceed);
  ceed->live_graphs++;
  *graph = G;
  return 0;
}
// 0: the recording still describes its vectors; 1: a qdata vector's geometry provenance was dropped; 2: a stored state's derived state
static int graph_stale(CeedXGraph G) {
  for (const GraphDep &d : G->deps) {
    if (d.geo && d.v->geo != d.geo) return 1;
    if (d.derived && !(d.v->derived_valid && d.v->derived == d.derived)) return 2;
  }
  return 0;
}
// The check CeedXGraphLaunch makes, without launching: on SEVERAL ranks a caller agrees on the outcome (e.g. a MAX all-reduce of
// *stale) BEFORE any rank replays a graph that holds RCCL sends / receives -- one rank refusing while its peers launch would hang
// the job.  The check itself is local to this rank.
extern "C" int CeedXGraphIsStale(CeedXGraph G, int *stale) { *stale = graph_stale(G); return 0; }
extern "C" int CeedXGraphLaunch(CeedXGraph G) {
  if (G->ceed->capturing) return ceed_error("CeedXGraphLaunch during graph capture");
  const int st = graph_stale(G);
  if (st == 1)
    return ceed_error("CeedXGraphLaunch: a qdata vector this graph's operators recompute the geometry of was overwritten after the recording "
                      "(its recorded kernels would still use the old element maps): record the graph again");
  if (st == 2)
    return ceed_error("CeedXGraphLaunch: the stored state a recorded HyperFSdF apply reads was overwritten outside the residual operator after "
                      "the recording (its derived state is no longer valid): record the graph again");
  HIPCHK(hipGraphLaunch(G->exec, G->ceed->stream));
  return 0;
}
extern "C" int CeedXGraphDestroy(CeedXGraph *graph) {
  if (!graph || !*graph) return 0;
  CeedXGraph G = *graph;
  (void)hipStreamSynchronize(G->ceed->stream);
  if (G->exec) (void)hipGraphExecDestroy(G->exec);
  if (G->graph) (void)hipGraphDestroy(G->graph);
  for (GraphDep &d : G->deps) { CeedVector v = d.v; (void)CeedVectorDestroy(&v); }
  if (--G->ceed->live_graphs == 0 && !G->ceed->capturing) ceed_free_parked(G->ceed);   // the stream was drained above
  ceed_unref(G->ceed);
  delete G;
  *graph = nullptr;
  return 0;
}

void vec_drop_geo(CeedVector v) {   // = "the vector is being written"
  v->version++;
  v->derived_valid = false;
  // retired, not freed: a recorded graph may hold these pointers in its kernel arguments (it refuses to replay --
  // CeedXGraphLaunch checks its GraphDeps -- but its nodes must never point at freed memory)
  ceed_retire(v->ceed, v->geo); ceed_retire(v->ceed, v->geo_aff); ceed_retire(v->ceed, v->geo_swept);
  v->geo = v->geo_aff = v->geo_swept = nullptr; v->geo_nelem = v->geo_Q = 0;
}

int ceed_need_evec(Ceed c, size_t len) {
  if (c->evec_len >= len) return 0;
  if (c->evec) {
    if (c->capturing || c->live_graphs > 0) c->evec_parked.push_back(c->evec);   // recorded nodes still point at it
    else { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->evec)); }
    c->evec = nullptr; c->evec_len = 0;
  }
  HIPCHK(hipMalloc((void **)&c->evec, sizeof(double) * len));
  c->evec_len = len;
  return 0;
}

// ---------------------------------------------------------------------------
// CeedVector: host and device mirrors with validity flags
// ---------------------------------------------------------------------------
static size_t vbytes(CeedVector v) { return sizeof(double) * (size_t)(v->length > 0 ? v->length : 1); }
// Zero `n` doubles on the Ceed's stream.  While a hipGraph is being recorded this is a fill KERNEL rather than a memset node
// (same cost).  Round 1 had blamed a wrong replay on recorded memset nodes losing their order; a library-free reproducer
// (tools/microbench/graph_memset_repro.hip) and this library with CEED_MI355X_GRAPH_MEMSET=1 both replay correctly: the
// cause was the scratch E-vector being re-allocated under recorded nodes (see ceed_need_evec).  CEED_MI355X_GRAPH_MEMSET=1
// records memset nodes instead (A/B: tools/graph_replay_check.py).
int dev_zero(Ceed c, double *p, size_t n) {
  if (!n) return 0;
  if (c->capturing && !c->opt.graph_memset) HIPCHK(launch_set_value(p, n, 0.0, c->stream));
  else HIPCHK(hipMemsetAsync(p, 0, sizeof(double) * n, c->stream));
  return 0;
}
static int vec_need_host(CeedVector v) {
  if (!v->h) { v->h = (double *)calloc(vbytes(v), 1); v->h_owned = true; }
  return 0;
}
static int vec_need_dev(CeedVector v) {
  if (!v->d) { HIPCHK(hipMalloc((void **)&v->d, vbytes(v))); v->d_owned = true; }
  return 0;
}
static int vec_sync_to(CeedVector v, CeedMemType m) {
  hipStream_t s = v->ceed->stream;
  if (m == CEED_MEM_HOST) {
    CHK(vec_need_host(v));
    if (!v->h_valid && v->d_valid) {
      if (v->ceed->capturing) return ceed_error("host access to a device vector during graph capture");
      HIPCHK(hipMemcpyAsync(v->h, v->d, sizeof(double) * (size_t)v->length, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
    }
    v->h_valid = true;
  } else {
    CHK(vec_need_dev(v));
    if (!v->d_valid && v->h_valid) {
      if (v->ceed->capturing) return ceed_error("host-to-device vector upload during graph capture");
      HIPCHK(hipMemcpyAsync(v->d, v->h, sizeof(double) * (size_t)v->length, hipMemcpyHostToDevice, s));
      HIPCHK(hipStreamSynchronize(s));  // the host buffer may be reused by the caller
    } else if (!v->d_valid && !v->h_valid) {
      CHK(dev_zero(v->ceed, v->d, (size_t)v->length));
    }
    v->d_valid = true;
  }
  return 0;
}
// device pointer for kernels; write=true invalidates the host mirror
int vec_dev(CeedVector v, bool write, double **p) {
  CHK(vec_sync_to(v, CEED_MEM_DEVICE));
  if (write) { v->h_valid = false; vec_drop_geo(v); }
  *p = v->d;
  return 0;
}
static void vec_drop_host(CeedVector v) { if (v->h_owned) free(v->h); v->h = nullptr; v->h_owned = false; v->h_valid = false; }
static void vec_drop_dev(CeedVector v) { if (v->d_owned && v->d) (void)hipFree(v->d); v->d = nullptr; v->d_owned = false; v->d_valid = false; }

extern "C" int CeedVectorCreate(Ceed ceed, CeedInt length, CeedVector *vec) {
  CeedVector v = new CeedVector_private;
  v->ceed = ceed; ceed_ref(ceed);
  v->length = length;
  *vec = v;
  return 0;
}
extern "C" int CeedVectorSetArray(CeedVector v, CeedMemType mtype, CeedCopyMode cmode, CeedScalar *array) {
  const size_t nb = sizeof(double) * (size_t)v->length;
  if (mtype == CEED_MEM_HOST) {
    if (cmode == CEED_COPY_VALUES) {
      if (!v->h_owned) v->h = nullptr;
      CHK(vec_need_host(v));
      if (array) memcpy(v->h, array, nb);
    } else {
      vec_drop_host(v);
      v->h = array; v->h_owned = (cmode == CEED_OWN_POINTER);
    }
    v->h_valid = true; v->d_valid = false; vec_drop_geo(v);
  } else {
    if (cmode == CEED_COPY_VALUES) {
      if (!v->d_owned) v->d = nullptr;
      CHK(vec_need_dev(v));
      if (array) HIPCHK(hipMemcpyAsync(v->d, array, nb, hipMemcpyDeviceToDevice, v->ceed->stream));
    } else {
      vec_drop_dev(v);
      v->d = array; v->d_owned = (cmode == CEED_OWN_POINTER);
    }
    v->d_valid = true; v->h_valid = false; vec_drop_geo(v);
  }
  return 0;
}
extern "C" int CeedVectorTakeArray(CeedVector v, CeedMemType mtype, CeedScalar **array) {
  if (mtype == CEED_MEM_HOST) {
    if (v->h || v->d_valid) CHK(vec_sync_to(v, CEED_MEM_HOST));
    if (array) *array = v->h;
    v->h = nullptr; v->h_owned = false; v->h_valid = false;
    if (!v->d_valid) vec_drop_geo(v);
  } else {
    if (v->d || v->h_valid) CHK(vec_sync_to(v, CEED_MEM_DEVICE));
    if (array) *array = v->d;
    v->d = nullptr; v->d_owned = false; v->d_valid = false;
    vec_drop_geo(v);
  }
  return 0;
}
extern "C" int CeedVectorSetValue(CeedVector v, CeedScalar value) {
  CHK(vec_need_dev(v));
  if (value == 0.) CHK(dev_zero(v->ceed, v->d, (size_t)v->length));
  else HIPCHK(launch_set_value(v->d, (size_t)v->length, value, v->ceed->stream));
  v->d_valid = true; v->h_valid = false; vec_drop_geo(v);
  return 0;
}
extern "C" int CeedVectorSyncArray(CeedVector v, CeedMemType mtype) { return vec_sync_to(v, mtype); }
extern "C" int CeedVectorGetArray(CeedVector v, CeedMemType mtype, CeedScalar **array) {
  CHK(vec_sync_to(v, mtype));
  if (mtype == CEED_MEM_HOST) { *array = v->h; v->d_valid = false; }
  else { *array = v->d; v->h_valid = false; }
  vec_drop_geo(v);   // write access
  return 0;
}
extern "C" int CeedVectorGetArrayRead(CeedVector v, CeedMemType mtype, const CeedScalar **array) {
  CHK(vec_sync_to(v, mtype));
  *array = mtype == CEED_MEM_HOST ? v->h : v->d;
  return 0;
}
extern "C" int CeedVectorRestoreArray(CeedVector, CeedScalar **array) { if (array) *array = nullptr; return 0; }
extern "C" int CeedVectorRestoreArrayRead(CeedVector, const CeedScalar **array) { if (array) *array = nullptr; return 0; }
extern "C" int CeedVectorGetLength(CeedVector v, CeedInt *length) { *length = v->length; return 0; }
extern "C" int CeedVectorReciprocal(CeedVector v) {
  double *p;
  CHK(vec_dev(v, true, &p));
  HIPCHK(launch_reciprocal(p, (size_t)v->length, v->ceed->stream));
  return 0;
}
extern "C" int CeedVectorDestroy(CeedVector *vec) {
  if (!vec || !*vec) return 0;
  CeedVector v = *vec;
  *vec = nullptr;
  if (v == CEED_VECTOR_ACTIVE || v == CEED_VECTOR_NONE) return 0;
  if (--v->refcount > 0) return 0;
  vec_drop_host(v); vec_drop_dev(v); vec_drop_geo(v);
  ceed_retire(v->ceed, v->derived);
  ceed_unref(v->ceed);
  delete v;
  return 0;
}

// ---------------------------------------------------------------------------
// vector helpers
// ---------------------------------------------------------------------------
// Vector helpers standing in for the PETSc Vec calls of src/matops.c on device data.
extern "C" int CeedXVectorPointwiseMult(CeedVector w, CeedVector x, CeedVector y) {
  double *pw, *px, *py;
  CHK(vec_dev(x, false, &px)); CHK(vec_dev(y, false, &py)); CHK(vec_dev(w, true, &pw));
  HIPCHK(launch_pointwise_mult(pw, px, py, (size_t)w->length, w->ceed->stream));
  return 0;
}
extern "C" int CeedXVectorAXPBY(CeedVector y, double a, CeedVector x, double b) {
  double *px, *py;
  CHK(vec_dev(x, false, &px)); CHK(vec_dev(y, true, &py));
  HIPCHK(launch_axpby(py, a, px, b, (size_t)y->length, y->ceed->stream));
  return 0;
}
extern "C" int CeedXVectorChebyshevUpdate(CeedVector x, CeedVector d, CeedVector r, CeedVector t, CeedVector dinv,
                                          double c1, double c2, int assign_x) {
  double *px, *pd, *pr, *pt = nullptr, *pi;
  const CeedInt n = x->length;
  if (d->length != n || r->length != n || dinv->length != n || (t && t != CEED_VECTOR_NONE && t->length != n))
    return ceed_error("CeedXVectorChebyshevUpdate: vector lengths differ");
  CHK(vec_dev(dinv, false, &pi));
  if (t && t != CEED_VECTOR_NONE) CHK(vec_dev(t, false, &pt));
  CHK(vec_dev(r, pt != nullptr, &pr)); CHK(vec_dev(d, true, &pd)); CHK(vec_dev(x, true, &px));
  HIPCHK(launch_cheb_update(px, pd, pr, nullptr, pt, pi, c1, c2, assign_x, (size_t)n, x->ceed->stream));
  return 0;
}
// first step of a Chebyshev sweep: r = b - t (t may be NULL), d = c1 dinv r, x = d or x + d -- no copy of b into r first
extern "C" int CeedXVectorChebyshevStart(CeedVector x, CeedVector d, CeedVector r, CeedVector b, CeedVector t, CeedVector dinv,
                                         double c1, int assign_x) {
  double *px, *pd, *pr, *pb, *pt = nullptr, *pi;
  const CeedInt n = x->length;
  if (d->length != n || r->length != n || b->length != n || dinv->length != n || (t && t != CEED_VECTOR_NONE && t->length != n))
    return ceed_error("CeedXVectorChebyshevStart: vector lengths differ");
  if (b == r || b == x || b == d) return ceed_error("CeedXVectorChebyshevStart: the right-hand side must be a vector of its own");
  CHK(vec_dev(dinv, false, &pi)); CHK(vec_dev(b, false, &pb));
  if (t && t != CEED_VECTOR_NONE) CHK(vec_dev(t, false, &pt));
  CHK(vec_dev(r, true, &pr)); CHK(vec_dev(d, true, &pd)); CHK(vec_dev(x, true, &px));
  HIPCHK(launch_cheb_update(px, pd, pr, pb, pt, pi, c1, 0., assign_x, (size_t)n, x->ceed->stream));
  return 0;
}
// One step with the residual RECOMPUTED from the right-hand side, as KSPCHEBYSHEV does (r = b - A x_k every iteration, not r -= A d):
// ri = b - t (t may be NULL);  d = c1 dinv ri + c2 d;  x = d or x + d;  ri is stored only if r is given.
extern "C" int CeedXVectorChebyshevStep(CeedVector x, CeedVector d, CeedVector r, CeedVector b, CeedVector t, CeedVector dinv,
                                        double c1, double c2, int assign_x) {
  double *px, *pd, *pr = nullptr, *pb, *pt = nullptr, *pi;
  const CeedInt n = x->length;
  const bool has_r = r && r != CEED_VECTOR_NONE, has_t = t && t != CEED_VECTOR_NONE;
  if (d->length != n || b->length != n || dinv->length != n || (has_r && r->length != n) || (has_t && t->length != n))
    return ceed_error("CeedXVectorChebyshevStep: vector lengths differ");
  if (b == x || b == d || (has_r && b == r)) return ceed_error("CeedXVectorChebyshevStep: the right-hand side must be a vector of its own");
  CHK(vec_dev(dinv, false, &pi)); CHK(vec_dev(b, false, &pb));
  if (has_t) CHK(vec_dev(t, false, &pt));
  if (has_r) CHK(vec_dev(r, true, &pr));
  CHK(vec_dev(d, true, &pd)); CHK(vec_dev(x, true, &px));
  HIPCHK(launch_cheb_update(px, pd, pr, pb, pt, pi, c1, c2, assign_x, (size_t)n, x->ceed->stream));
  return 0;
}
extern "C" int CeedXVectorWAXPBY(CeedVector w, double a, CeedVector x, double b, CeedVector y) {
  if (x->length != w->length || y->length != w->length) return ceed_error("CeedXVectorWAXPBY: vector lengths differ");
  double *px, *py, *pw;
  CHK(vec_dev(x, false, &px)); CHK(vec_dev(y, false, &py)); CHK(vec_dev(w, true, &pw));
  HIPCHK(launch_waxpby(pw, a, px, b, py, (size_t)w->length, w->ceed->stream));
  return 0;
}
extern "C" int CeedXVectorDot(CeedVector x, CeedVector y, CeedVector weight, double *result) {
  double *px, *py, *pw = nullptr, *dres;
  CHK(vec_dev(x, false, &px)); CHK(vec_dev(y, false, &py));
  if (weight && weight != CEED_VECTOR_NONE) CHK(vec_dev(weight, false, &pw));
  hipStream_t s = x->ceed->stream;
  if (x->ceed->capturing) return ceed_error("CeedXVectorDot during graph capture (it returns a host value)");
  if (!x->ceed->d_scalar) HIPCHK(hipMalloc((void **)&x->ceed->d_scalar, sizeof(double) * (1 + 2048)));  // result + per-block partials
  if (!x->ceed->h_scalar) HIPCHK(hipHostMalloc((void **)&x->ceed->h_scalar, sizeof(double), hipHostMallocDefault));
  dres = x->ceed->d_scalar;
  HIPCHK(launch_dot(px, py, pw, (size_t)x->length, dres, s));
  HIPCHK(hipMemcpyAsync(x->ceed->h_scalar, dres, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  *result = *x->ceed->h_scalar;
  return 0;
}
// Scalars that stay on the device: a CeedVector as a small register file, so that a Krylov recurrence with a fixed number
// of steps (the Lanczos eigenvalue estimate of the smoothers) runs without a host round trip per dot.  All recordable.
extern "C" int CeedXVectorDotTo(CeedVector x, CeedVector y, CeedVector weight, CeedVector scalars, CeedInt idx) {
  if (idx < 0 || idx >= scalars->length) return ceed_error("CeedXVectorDotTo: scalar %d of %d", idx, scalars->length);
  if (y->length != x->length) return ceed_error("CeedXVectorDotTo: vector lengths differ");
  double *px, *py, *pw = nullptr, *ps;
  CHK(vec_dev(x, false, &px)); CHK(vec_dev(y, false, &py)); CHK(vec_dev(scalars, true, &ps));
  if (weight && weight != CEED_VECTOR_NONE) CHK(vec_dev(weight, false, &pw));
  if (!x->ceed->d_scalar) {
    if (x->ceed->capturing) return ceed_error("CeedXVectorDotTo: take one dot product before recording (scratch allocation)");
    HIPCHK(hipMalloc((void **)&x->ceed->d_scalar, sizeof(double) * (1 + 2048)));
  }
  HIPCHK(launch_dot(px, py, pw, (size_t)x->length, x->ceed->d_scalar, x->ceed->stream, ps + idx));
  return 0;
}
extern "C" int CeedXScalarDivide(CeedVector scalars, CeedInt dst, CeedInt num, CeedInt den, double scale) {
  const CeedInt n = scalars->length;
  if (dst < 0 || dst >= n || num < 0 || num >= n || den >= n) return ceed_error("CeedXScalarDivide: index out of range");
  double *ps;
  CHK(vec_dev(scalars, true, &ps));
  HIPCHK(launch_scalar_div(ps, dst, num, den, scale, scalars->ceed->stream));
  return 0;
}
extern "C" int CeedXVectorAXPBYScalars(CeedVector y, CeedVector scalars, CeedInt ia, double sa, CeedVector x, CeedInt ib, double sb) {
  if (ia >= scalars->length || ib >= scalars->length) return ceed_error("CeedXVectorAXPBYScalars: index out of range");
  if (x->length != y->length || x == y) return ceed_error("CeedXVectorAXPBYScalars: bad vectors");
  double *px, *py, *ps;
  CHK(vec_dev(x, false, &px)); CHK(vec_dev(scalars, false, &ps)); CHK(vec_dev(y, true, &py));
  HIPCHK(launch_axpby_dev(py, ps, ia, sa, px, ib, sb, (size_t)y->length, y->ceed->stream));
  return 0;
}
