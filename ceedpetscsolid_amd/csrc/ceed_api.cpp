// ceed_api.cpp -- host side of the MI355X backend behind include/ceed.h.
//
// Resource "/gpu/hip/mi355x".  Objects are reference counted exactly as the
// reference expects (operators keep their qfunction / restrictions / bases /
// passive vectors alive after the creator destroys its handles, e.g.
// setuplibceed.c:392-393).  A CeedOperator is lowered, at its first apply, to one
// hand-written gfx950 kernel family by matching its field signature against the
// operator graphs the reference builds:
//
//   fused_grad : GRAD active in, NONE qdata (+ NONE state in / out), GRAD active out
//                -> opApply (setuplibceed.c:517-542) and opJacob per level (:817-839)
//   setup_geo  : GRAD coords + WEIGHT -> NONE qdata             (:370-389)
//   prolong    : Identity, INTERP in -> NONE out                (:857-862)
//   restrict   : Identity, NONE in  -> INTERP out               (:849-854)
//
// There is NO host fallback: a graph outside these families, a QFunction without
// a device functor, or a missing GPU is a loud error.
#include <ceed.h>
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"

using namespace cps;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static int g_err_return = 0;
static thread_local char g_err_msg[1024] = "";

static int ceed_error(const char *fmt, ...) {
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
#define CHK(x) do { int ierr_ = (x); if (ierr_) return ierr_; } while (0)
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
  return ceed_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

extern "C" int CeedXSetErrorReturn(int enable) { g_err_return = enable; return 0; }
extern "C" const char *CeedXLastError(void) { return g_err_msg; }

// ---------------------------------------------------------------------------
// object layouts
// ---------------------------------------------------------------------------
struct Ceed_private {
  int refcount = 1;
  std::string resource;
  hipStream_t stream = nullptr;
  int device = 0;
  // scratch E-vector shared by the operators of this Ceed (applies are serialised on `stream`)
  double *evec = nullptr;
  size_t evec_len = 0;
  // A recorded hipGraph has the scratch pointer of its capture time baked into its kernel nodes.  When the scratch has
  // to grow while a graph of this Ceed is alive (or is being recorded), the old buffer is PARKED, not freed: replays of
  // the older graphs keep a valid scratch of the size they were recorded with.  Parked buffers go when the last graph goes.
  std::vector<double *> evec_parked;
  int live_graphs = 0;
  bool atomic_scatter = false;  // CEED_MI355X_SCATTER=atomic: f64 atomics instead of E-vector + assembly
  int fused_variant = 1;        // CEED_MI355X_FUSED=rows: the first-generation row kernel (A/B); default pencil
  bool recompute_geo = true;    // fused pencil kernel recomputes SetupGeo's factors from the element maps (CEED_MI355X_GEO=0: reads qdata)
  bool even_odd = true;         // pencil kernel applies the 1-D tables in even-odd form (CEED_MI355X_EO=0: plain products)
  bool direct_interior = true;  // pencil kernel: element-interior nodes go straight to y (CEED_MI355X_DIRECT=0: all via the E-vector)
  unsigned *queue = nullptr;    // per-XCD ticket counters of the pencil kernel's dynamic schedule (8 x QUEUE_STRIDE)
  bool pair_merge = false;      // pencil kernel, two elements per wave: shared nodes summed in LDS before the store (CEED_MI355X_PAIR=1;
                                // measured: k_assemble -10 %, fused kernel +3.7 %, net 0 -- off by default)
  bool dynamic_sched = false;   // pencil kernel: groups taken dynamically per XCD (CEED_MI355X_SCHED=dynamic; the gated and folded assembly
                                // forms always do); default: round 1's static striding -- equal on large meshes, faster on small ones
  int asm_overlap = 0;          // EXPERIMENT CEED_MI355X_ASM_OVERLAP=1: k_assemble on a second stream beside the fused kernel (ungated: timing only)
  // Restriction transpose of the fused residual / Jacobian apply (CEED_MI355X_ASSEMBLE): "serial" (default) = k_assemble after
  // the fused kernel; "gated" = k_assemble_gated beside the fused kernel on a second stream + k_assemble_tail; "folded" =
  // summed by the pencil kernel's own waves, one item per element group, + k_assemble_tail.  Measured in DESIGN.md 8.
  bool gated_assembly = false;   // folded or gated
  bool folded_assembly = false;
  int gated_waves = 4;          // persistent assembler waves per CU (CEED_MI355X_ASM_WAVES)
  int gated_spins = 1 << 19;    // the gated kernel's bounded wait for one bucket, in ~2 us polls (CEED_MI355X_ASM_SPINS)
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int pipe_segments = 0;        // CEED_MI355X_ASSEMBLE=pipelined: segments per apply (CEED_MI355X_PIPE_SEGMENTS, default 4)
  int pipe_blocks = 0;          // workgroups of the k_assemble launches that run beside a fused kernel (0: one per 256 rows)
  int pipe_chains = 0;          // CEED_MI355X_PIPE_CHAINS=1: segments alternate between two streams (fused kernel + its rows per stream)
  int pipe_last_rounds = 4;     // rounds of the persistent waves in the LAST segment (CEED_MI355X_PIPE_LAST; 0: equal segments)
  int pipe_min_total_rounds = 20;   // rounds a whole apply must have to be pipelined (CEED_MI355X_PIPE_MIN_TOTAL)
  int pipe_min_rounds = 4;      // rounds of the persistent waves a segment must have (CEED_MI355X_PIPE_MIN_ROUNDS; 0: tests on small meshes)
  hipEvent_t ev_seg[16] = {nullptr};
  // RCCL communicator of the halo exchange (CeedXCommInit) and the stream its sends / receives run on
  void *comm = nullptr;
  int comm_rank = 0, comm_size = 1;
  hipStream_t comm_stream = nullptr;
  double *d_scalar = nullptr;   // device scalar for reductions
  double *h_scalar = nullptr;   // pinned host landing slot for it (pageable targets make the runtime stage + pin per copy)
  // hipGraph capture (CeedXGraphBeginCapture): device work is recorded on `capture_stream`
  hipStream_t capture_stream = nullptr, saved_stream = nullptr;
  bool capturing = false;
};
struct CeedXGraph_private {
  Ceed ceed = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  size_t nodes = 0;
};

struct CeedVector_private {
  Ceed ceed = nullptr;
  int refcount = 1;
  CeedInt length = 0;
  double *h = nullptr, *d = nullptr;   // current host / device storage
  bool h_owned = false, d_owned = false;
  bool h_valid = false, d_valid = false;
  // provenance of a qdata vector: written by the SetupGeo operator from trilinear elements whose map coefficients
  // are kept here ([nelem][GEO_NCOEF], device).  The fused kernels then recompute the geometric factors instead of
  // reading them (FusedGradArgs::geo).  Dropped by any other write to the vector.
  double *geo = nullptr;
  int geo_nelem = 0, geo_Q = 0;
  double geo_qref[MAXN1D] = {0}, geo_qwt[MAXN1D] = {0};
};
static void vec_drop_geo(CeedVector v) { if (v->geo) (void)hipFree(v->geo); v->geo = nullptr; v->geo_nelem = v->geo_Q = 0; }

// Transpose map of an offsets restriction: distinct node offsets and, per node, the E-vector
// positions (e*elemsize + n) of its contributors in element order.  Rows [0, nprio) are the
// "priority" nodes when the map was built with a priority mask (split-phase apply).
struct CsrMap {
  bool built = false, full_cover = false;
  int nnodes = 0, nprio = 0, nskipped = 0;
  std::vector<uint32_t> h_node_off;
  uint32_t *d_rowptr = nullptr, *d_cols = nullptr, *d_node_off = nullptr;
  void release() {
    if (d_rowptr) (void)hipFree(d_rowptr);
    if (d_cols) (void)hipFree(d_cols);
    if (d_node_off) (void)hipFree(d_node_off);
    d_rowptr = d_cols = d_node_off = nullptr; built = false;
  }
};

// The transpose map re-ordered for the gated assembly (kernels.hpp, GatedAsmArgs): rows by (chunk, bucket of the last
// contributor), cut rows last; built for one group size E (elements per wave of the pencil kernel) on the host.
struct GatedMap {
  bool built = false;
  int E = 0, skipP = 0, nb = 0, bucket_shift = 0, nitems = 0, nrows_local = 0, nrows = 0, nskipped = 0;
  int evec_stride = 0;   // doubles per element block of the E-vector: whole 128-byte lines
  int item_rows = 0, max_contrib = 0;   // rows per item; rows with more contributors are cut rows (0: no limit)
  bool full_cover = false;
  int item_begin[9] = {0};
  std::vector<uint32_t> h_node_off;   // re-ordered (for the per-operator Dirichlet flags)
  uint32_t *d_rowptr = nullptr, *d_cols = nullptr, *d_node_off = nullptr, *d_item_row = nullptr, *d_item_bucket = nullptr,
           *d_bucket_groups = nullptr, *d_bucket_items = nullptr;
  unsigned *d_ctrl = nullptr;
  void release() {
    for (uint32_t *p : {d_rowptr, d_cols, d_node_off, d_item_row, d_item_bucket, d_bucket_groups, d_bucket_items}) if (p) (void)hipFree(p);
    if (d_ctrl) (void)hipFree(d_ctrl);
    d_rowptr = d_cols = d_node_off = d_item_row = d_item_bucket = d_bucket_groups = d_bucket_items = nullptr; d_ctrl = nullptr; built = false;
  }
};

// The transpose map re-ordered for the PIPELINED assembly: the apply is cut into segments of consecutive elements, one
// launch of the fused kernel each; a row (node) belongs to the segment of its LAST contributor, rows are sorted by segment,
// and the rows of segment k are summed by their own k_assemble launch beside the fused kernel of segment k + 1.
struct PipeMap {
  bool built = false;
  int nseg = 0, req_seg = 0, E = 0, waves = 0, nrows = 0;
  int build_id = 0;                        // bumped by every (re)build: operators cache per-row flags in this row order
  const void *base = nullptr;              // the CsrMap it was derived from
  std::vector<int> elem_bound, row_bound;  // nseg + 1 each
  std::vector<uint32_t> h_node_off;        // re-ordered (for the per-operator Dirichlet flags)
  uint32_t *d_rowptr = nullptr, *d_cols = nullptr, *d_node_off = nullptr;
  void release() {
    for (uint32_t *p : {d_rowptr, d_cols, d_node_off}) if (p) (void)hipFree(p);
    d_rowptr = d_cols = d_node_off = nullptr; built = false;
  }
};

struct CeedElemRestriction_private {
  Ceed ceed = nullptr;
  int refcount = 1;
  CeedInt nelem = 0, elemsize = 0, ncomp = 0, compstride = 0, lsize = 0;
  bool strided = false, backend_strides = true;
  CeedInt strides[3] = {0, 0, 0};
  std::vector<CeedInt> h_offsets;
  uint32_t *d_offsets = nullptr;  // plain (unflagged)
  // transpose map for the atomic-free scatter: distinct node offsets (ascending), their contributors
  // (E-vector positions e*elemsize + n, in element order) -- built on first use
  CsrMap csr;   // default map (nodes in ascending offset order)
  CsrMap csr_shell;        // the same without the element-interior nodes (FusedGradArgs::direct)
  GatedMap gated;          // the map of the fused residual / Jacobian apply, re-ordered for the gated assembly
  PipeMap pipe;            // ... re-ordered for the pipelined assembly
  // pair merge (FusedGradArgs::pairs): built once per restriction for groups of two elements
  int pair_state = 0;      // 0 not built, 1 built, -1 not applicable
  uint16_t *d_pairs = nullptr;
  std::vector<unsigned char> h_nflag;   // per (element, local node): 1 = merged into the group's first element, 2 = complete there
  long pair_nodes = 0, pair_complete = 0;
  int interior_private = 0;  // 0: not checked yet; 1: every element-interior node has one contributor; -1: not so
};

struct CeedBasis_private {
  Ceed ceed = nullptr;
  int refcount = 1;
  CeedInt dim = 3, ncomp = 0, P1d = 0, Q1d = 0;
  CeedQuadMode qmode = CEED_GAUSS;
  std::vector<double> interp1d, grad1d, qref1d, qweight1d, colo1d;
};

struct QFField { std::string name; CeedInt size; CeedEvalMode emode; };

struct CeedQFunction_private {
  Ceed ceed = nullptr;
  int refcount = 1;
  CeedQFunctionUser f = nullptr;  // kept, never called: device functors do the work
  std::string source, name;
  int kind = QF_NONE;
  void *ctx = nullptr;
  size_t ctxsize = 0;
  CeedInt identity_size = 0;
  std::vector<QFField> in, out;
};

struct OpField { bool set = false; CeedElemRestriction rstr = nullptr; CeedBasis basis = nullptr; CeedVector vec = nullptr; };

enum PlanKind { PLAN_NONE = 0, PLAN_FUSED_GRAD, PLAN_SETUP_GEO, PLAN_PROLONG, PLAN_RESTRICT, PLAN_COORD, PLAN_ENERGY };

struct CeedOperator_private {
  Ceed ceed = nullptr;
  int refcount = 1;
  CeedQFunction qf = nullptr;
  std::vector<OpField> in, out;
  bool composite = false;
  std::vector<CeedOperator> sub;
  // lowering
  int plan = PLAN_NONE;
  int i_active = -1, i_qdata = -1, i_state = -1, i_weight = -1, o_active = -1, o_state = -1, o_qdata = -1;
  BasisTables tables;
  std::string kernel_name;
  // Dirichlet flags
  uint32_t *d_off_flagged_in = nullptr, *d_off_flagged_out = nullptr;  // same array unless transfer
  unsigned char *d_node_flags = nullptr;      // per node of the restriction's transpose map
  unsigned char *d_node_flags_ovl = nullptr;  // per node of the operator's own (priority-first) map
  unsigned char *d_node_flags_shell = nullptr;  // per node of the restriction's shell map (direct-store mode)
  unsigned char *d_node_flags_gated = nullptr;  // per row of the restriction's gated map
  unsigned char *d_node_flags_pipe = nullptr;   // per row of the restriction's pipelined map
  int pipe_flags_id = 0;                        // PipeMap::build_id the flags were made for
  uint32_t *d_off_paired = nullptr;             // offsets with the Dirichlet flags AND the pair-merge bits (whole applies of the pencil kernel)
  std::vector<unsigned char> h_mask;      // copy of the output mask (node flags are derived lazily)
  int mask_mode = 0;
  // optional fine-side scale for transfers
  CeedVector scale = nullptr;
  // split-phase apply (communication overlap): the first `ovl_lead` elements are the only
  // contributors of the priority nodes, which come first in the operator's own transpose map
  int ovl_lead = 0;
  CsrMap ovl_csr;
  unsigned long long *stamps = nullptr;  // diagnostic builds only
  // timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  double ms_accum = 0.;
  int64_t launches = 0;
  int launch_info[4] = {0, 0, 0, 0};   // CeedXOperatorGetLaunchInfo
};

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
  const char *sc = getenv("CEED_MI355X_SCATTER");
  c->atomic_scatter = sc && !strcmp(sc, "atomic");
  const char *fv = getenv("CEED_MI355X_FUSED");
  c->fused_variant = (fv && !strcmp(fv, "rows")) ? 0 : 1;
  const char *rg = getenv("CEED_MI355X_GEO");
  c->recompute_geo = c->fused_variant == 1 && !(rg && !strcmp(rg, "0"));
  const char *eo = getenv("CEED_MI355X_EO");
  c->even_odd = c->fused_variant == 1 && !(eo && !strcmp(eo, "0"));
  const char *di = getenv("CEED_MI355X_DIRECT");
  c->direct_interior = c->fused_variant == 1 && !c->atomic_scatter && !(di && !strcmp(di, "0"));
  const char *sd = getenv("CEED_MI355X_SCHED");
  c->dynamic_sched = c->fused_variant == 1 && sd && !strcmp(sd, "dynamic");
  const char *ao = getenv("CEED_MI355X_ASM_OVERLAP");
  c->asm_overlap = ao ? atoi(ao) : 0;
  const char *pm = getenv("CEED_MI355X_PAIR");
  c->pair_merge = c->fused_variant == 1 && pm && !strcmp(pm, "1");
  const char *ga = getenv("CEED_MI355X_ASSEMBLE");
  c->gated_assembly = c->fused_variant == 1 && !c->atomic_scatter && ga && (!strcmp(ga, "gated") || !strcmp(ga, "folded"));
  c->folded_assembly = c->gated_assembly && !strcmp(ga, "folded");
  // DEFAULT form of the restriction transpose on large launches: pipelined (CEED_MI355X_ASSEMBLE=serial switches it off)
  if (c->fused_variant == 1 && !c->atomic_scatter && !c->gated_assembly && !(ga && !strcmp(ga, "serial"))) {
    const char *ps = getenv("CEED_MI355X_PIPE_SEGMENTS"), *pb = getenv("CEED_MI355X_PIPE_BLOCKS");
    c->pipe_segments = ps && atoi(ps) >= 2 ? std::min(atoi(ps), 16) : -1;     // -1: chosen per launch (build_pipe)
    c->pipe_blocks = pb ? atoi(pb) : 0;
    const char *pc = getenv("CEED_MI355X_PIPE_CHAINS");
    c->pipe_chains = pc ? atoi(pc) : 1;
    const char *pl = getenv("CEED_MI355X_PIPE_LAST");
    if (pl) c->pipe_last_rounds = std::max(0, atoi(pl));
    const char *pt = getenv("CEED_MI355X_PIPE_MIN_TOTAL");
    if (pt) c->pipe_min_total_rounds = std::max(0, atoi(pt));
    const char *pr = getenv("CEED_MI355X_PIPE_MIN_ROUNDS");
    if (pr) c->pipe_min_rounds = std::max(0, atoi(pr));
  }
  const char *gw = getenv("CEED_MI355X_ASM_WAVES");
  if (gw && atoi(gw) > 0) c->gated_waves = atoi(gw);
  const char *gs = getenv("CEED_MI355X_ASM_SPINS");
  if (gs && atoi(gs) > 0) c->gated_spins = atoi(gs);
  *ceed = c;
  return 0;
}
static void ceed_ref(Ceed c) { c->refcount++; }
static int (*g_rccl_destroy)(void *) = nullptr;   // set when RCCL is bound (CeedXCommInit)
static void ceed_free_parked(Ceed c) { for (double *p : c->evec_parked) (void)hipFree(p); c->evec_parked.clear(); }
static void ceed_unref(Ceed c) { if (--c->refcount == 0) { if (c->capture_stream) (void)hipStreamDestroy(c->capture_stream); if (c->evec) (void)hipFree(c->evec); ceed_free_parked(c); if (c->queue) (void)hipFree(c->queue); if (c->side_stream) (void)hipStreamDestroy(c->side_stream); if (c->comm && g_rccl_destroy) g_rccl_destroy(c->comm); if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream); if (c->ev_fork) (void)hipEventDestroy(c->ev_fork); if (c->ev_join) (void)hipEventDestroy(c->ev_join); for (hipEvent_t e : c->ev_seg) if (e) (void)hipEventDestroy(e); if (c->d_scalar) (void)hipFree(c->d_scalar); if (c->h_scalar) (void)hipHostFree(c->h_scalar); delete c; } }
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
extern "C" int CeedXGraphBeginCapture(Ceed ceed) {
  if (ceed->capturing) return ceed_error("graph capture already in progress");
  HIPCHK(hipStreamSynchronize(ceed->stream));
  if (!ceed->capture_stream) HIPCHK(hipStreamCreateWithFlags(&ceed->capture_stream, hipStreamNonBlocking));
  ceed->saved_stream = ceed->stream;
  ceed->stream = ceed->capture_stream;
  HIPCHK(hipStreamBeginCapture(ceed->stream, hipStreamCaptureModeRelaxed));
  ceed->capturing = true;
  return 0;
}
extern "C" int CeedXGraphEndCapture(Ceed ceed, CeedXGraph *graph) {
  if (!ceed->capturing) return ceed_error("no graph capture in progress");
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(ceed->stream, &g);
  ceed->stream = ceed->saved_stream;
  ceed->capturing = false;
  if (e != hipSuccess || !g) return ceed_error("graph capture failed: %s", hipGetErrorString(e));
  CeedXGraph G = new CeedXGraph_private;
  G->ceed = ceed; G->graph = g;
  (void)hipGraphGetNodes(g, nullptr, &G->nodes);
  e = hipGraphInstantiate(&G->exec, g, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g); delete G; return ceed_error("hipGraphInstantiate: %s", hipGetErrorString(e)); }
  ceed_ref(ceed);
  ceed->live_graphs++;
  *graph = G;
  return 0;
}
extern "C" int CeedXGraphLaunch(CeedXGraph G) {
  if (G->ceed->capturing) return ceed_error("CeedXGraphLaunch during graph capture");
  HIPCHK(hipGraphLaunch(G->exec, G->ceed->stream));
  return 0;
}
extern "C" int CeedXGraphDestroy(CeedXGraph *graph) {
  if (!graph || !*graph) return 0;
  CeedXGraph G = *graph;
  (void)hipStreamSynchronize(G->ceed->stream);
  if (G->exec) (void)hipGraphExecDestroy(G->exec);
  if (G->graph) (void)hipGraphDestroy(G->graph);
  if (--G->ceed->live_graphs == 0 && !G->ceed->capturing) ceed_free_parked(G->ceed);   // the stream was drained above
  ceed_unref(G->ceed);
  delete G;
  *graph = nullptr;
  return 0;
}

// ---------------------------------------------------------------------------
// CeedVector: host and device mirrors with validity flags
// ---------------------------------------------------------------------------
static size_t vbytes(CeedVector v) {
  static const bool pad = getenv("CEED_MI355X_QPAD_ALLOC") != nullptr;   // experiment: room for padded q-point runs (tools/variants)
  return sizeof(double) * (size_t)(v->length > 0 ? v->length : 1) * (pad ? 132 : 128) / 128;
}
// Zero `n` doubles on the Ceed's stream.  While a hipGraph is being recorded this is a fill KERNEL rather than a memset node
// (same cost).  Round 1 had blamed a wrong replay on recorded memset nodes losing their order; a library-free reproducer
// (tools/microbench/graph_memset_repro.hip) and this library with CEED_MI355X_GRAPH_MEMSET=1 both replay correctly: the
// cause was the scratch E-vector being re-allocated under recorded nodes (see ceed_need_evec).
static int dev_zero(Ceed c, double *p, size_t n) {
  if (!n) return 0;
  static const bool memset_nodes = getenv("CEED_MI355X_GRAPH_MEMSET") && atoi(getenv("CEED_MI355X_GRAPH_MEMSET"));   // A/B: tools/graph_replay_check.py
  if (c->capturing && !memset_nodes) HIPCHK(launch_set_value(p, n, 0.0, c->stream));
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
static int vec_dev(CeedVector v, bool write, double **p) {
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
  ceed_unref(v->ceed);
  delete v;
  return 0;
}

// ---------------------------------------------------------------------------
// CeedElemRestriction
// ---------------------------------------------------------------------------
extern "C" int CeedElemRestrictionCreate(Ceed ceed, CeedInt nelem, CeedInt elemsize, CeedInt ncomp,
                                         CeedInt compstride, CeedInt lsize, CeedMemType mtype,
                                         CeedCopyMode, const CeedInt *offsets, CeedElemRestriction *rstr) {
  if (mtype != CEED_MEM_HOST) return ceed_error("restriction offsets are expected in host memory (setuplibceed.c:235)");
  if ((uint32_t)lsize > OFF_MASK) return ceed_error("L-vector of %d entries exceeds the 2^29 offset range of this backend", lsize);
  const size_t n = (size_t)nelem * elemsize;
  for (size_t i = 0; i < n; i++) {
    const long last = (long)offsets[i] + (long)(ncomp - 1) * compstride;
    if (offsets[i] < 0 || last >= lsize)
      return ceed_error("restriction offset %zu = %d out of range [0,%d)", i, offsets[i], lsize);
  }
  CeedElemRestriction r = new CeedElemRestriction_private;
  r->ceed = ceed; ceed_ref(ceed);
  r->nelem = nelem; r->elemsize = elemsize; r->ncomp = ncomp; r->compstride = compstride; r->lsize = lsize;
  r->h_offsets.assign(offsets, offsets + n);
  HIPCHK(hipMalloc((void **)&r->d_offsets, sizeof(uint32_t) * (n ? n : 1)));
  HIPCHK(hipMemcpy(r->d_offsets, offsets, sizeof(uint32_t) * n, hipMemcpyHostToDevice));
  *rstr = r;
  return 0;
}
extern "C" int CeedElemRestrictionCreateStrided(Ceed ceed, CeedInt nelem, CeedInt elemsize, CeedInt ncomp,
                                                CeedInt lsize, const CeedInt strides[3], CeedElemRestriction *rstr) {
  if ((long)nelem * elemsize * ncomp > lsize) return ceed_error("strided restriction larger than its L-vector");
  CeedElemRestriction r = new CeedElemRestriction_private;
  r->ceed = ceed; ceed_ref(ceed);
  r->nelem = nelem; r->elemsize = elemsize; r->ncomp = ncomp; r->lsize = lsize;
  r->strided = true;
  // CEED_STRIDES_BACKEND (setuplibceed.c:304-318): this backend lays q-point data out as
  // [element][component][point]: one contiguous run per wave-instruction in the fused kernels.
  r->backend_strides = strides[0] < 0;
  if (r->backend_strides) { r->strides[0] = 1; r->strides[1] = elemsize; r->strides[2] = elemsize * ncomp; }
  else {
    memcpy(r->strides, strides, sizeof r->strides);
    if (!(strides[0] == 1 && strides[1] == elemsize && strides[2] == elemsize * ncomp))
      return ceed_error("only the [elem][comp][node] strided layout is supported on /gpu/hip/mi355x");
  }
  *rstr = r;
  return 0;
}
extern "C" int CeedElemRestrictionCreateVector(CeedElemRestriction r, CeedVector *lvec, CeedVector *evec) {
  if (lvec) CHK(CeedVectorCreate(r->ceed, r->lsize, lvec));
  if (evec) CHK(CeedVectorCreate(r->ceed, r->nelem * r->elemsize * r->ncomp, evec));
  return 0;
}
extern "C" int CeedElemRestrictionApply(CeedElemRestriction r, CeedTransposeMode tmode, CeedVector u,
                                        CeedVector ru, CeedRequest *) {
  hipStream_t s = r->ceed->stream;
  double *pu, *pv;
  CHK(vec_dev(u, false, &pu));
  CHK(vec_dev(ru, true, &pv));
  if (r->strided) {  // identity layout: E == L
    const size_t n = (size_t)r->nelem * r->elemsize * r->ncomp;
    if (tmode == CEED_NOTRANSPOSE) HIPCHK(hipMemcpyAsync(pv, pu, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    else HIPCHK(launch_axpby(pv, 1., pu, 1., n, s));
    return 0;
  }
  if (tmode == CEED_NOTRANSPOSE) HIPCHK(launch_rstr_gather(r->d_offsets, r->nelem, r->elemsize, r->ncomp, r->compstride, pu, pv, s));
  else HIPCHK(launch_rstr_scatter_add(r->d_offsets, r->nelem, r->elemsize, r->ncomp, r->compstride, pu, pv, s));
  return 0;
}
extern "C" int CeedElemRestrictionGetMultiplicity(CeedElemRestriction r, CeedVector mult) {
  if (r->strided) return CeedVectorSetValue(mult, 1.);
  CHK(CeedVectorSetValue(mult, 0.));
  HIPCHK(launch_multiplicity(r->d_offsets, r->nelem, r->elemsize, r->ncomp, r->compstride, mult->d, r->ceed->stream));
  return 0;
}
extern "C" int CeedElemRestrictionDestroy(CeedElemRestriction *rstr) {
  if (!rstr || !*rstr) return 0;
  CeedElemRestriction r = *rstr;
  *rstr = nullptr;
  if (r == CEED_ELEMRESTRICTION_NONE) return 0;
  if (--r->refcount > 0) return 0;
  if (r->d_offsets) (void)hipFree(r->d_offsets);
  r->csr.release();
  r->csr_shell.release();
  r->gated.release();
  r->pipe.release();
  if (r->d_pairs) (void)hipFree(r->d_pairs);
  ceed_unref(r->ceed);
  delete r;
  return 0;
}

// ---------------------------------------------------------------------------
// 1-D rules and tables; host side, set-up time only
// ---------------------------------------------------------------------------
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
static void lagrange_at(int P, const double *nodes, double x, double *val, double *der) {
  for (int j = 0; j < P; j++) {
    double v = 1., d = 0.;
    for (int m = 0; m < P; m++) {
      if (m == j) continue;
      const double inv = 1. / (nodes[j] - nodes[m]);
      d = d * (x - nodes[m]) * inv + v * inv;
      v *= (x - nodes[m]) * inv;
    }
    val[j] = v; der[j] = d;
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
    lagrange_at(P, nodes.data(), b->qref1d[q], &b->interp1d[(size_t)q * P], &b->grad1d[(size_t)q * P]);
    // collocated derivative: Lagrange basis ON the quadrature points, differentiated there
    if (Q > 1) lagrange_at(Q, b->qref1d.data(), b->qref1d[q], tmp.data(), &b->colo1d[(size_t)q * Q]);
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
  if (o->d_node_flags_gated) (void)hipFree(o->d_node_flags_gated);
  if (o->d_node_flags_pipe) (void)hipFree(o->d_node_flags_pipe);
  o->d_node_flags_pipe = nullptr;
  if (o->d_off_paired) (void)hipFree(o->d_off_paired);
  o->d_off_paired = nullptr;
  o->d_node_flags = o->d_node_flags_ovl = o->d_node_flags_shell = o->d_node_flags_gated = nullptr;
  o->h_mask.clear();
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
  o->ovl_csr.release();
  CeedVectorDestroy(&o->scale);
  for (auto &ev : o->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  ceed_unref(o->ceed);
  delete o;
  return 0;
}

// Build a transpose map (setup time, host): counting sort over the L-vector.  With `prio`
// (one byte per L-vector entry, tested at each node's component-0 offset) the flagged nodes
// come first.
// `skipP` > 0 (elemsize == skipP^3): nodes interior to an element are left out of the map -- the fused kernel
// stores them itself (FusedGradArgs::direct); the caller has checked rstr_interior_private().
static int build_csr(CeedElemRestriction r, CsrMap &M, const unsigned char *prio, int skipP = 0, const unsigned char *nflag = nullptr) {
  if (M.built) return 0;
  if (r->ceed->capturing)
    return ceed_error("first apply of an operator during graph capture: its restriction's transpose map is built on the host; "
                      "apply the operator once before recording");
  const size_t n = r->h_offsets.size();
  std::vector<uint32_t> cnt((size_t)r->lsize + 1, 0u);
  for (size_t i = 0; i < n; i++) cnt[(size_t)r->h_offsets[i]]++;
  M.nskipped = 0;
  if (nflag)   // pair merge: the second element's copy of a shared node is no contributor
    for (size_t i = 0; i < n; i++)
      if (nflag[i] & 1) cnt[(size_t)r->h_offsets[i]]--;
  if (skipP > 0)
    for (size_t i = 0; i < n; i++)
      if (node_is_element_interior((int)(i % (size_t)r->elemsize), skipP) || (nflag && (nflag[i] & 2))) { cnt[(size_t)r->h_offsets[i]] = 0; M.nskipped++; }   // stored by the fused kernel itself
  std::vector<uint32_t> slot((size_t)r->lsize, 0xFFFFFFFFu), rowptr;
  M.h_node_off.clear();
  rowptr.push_back(0u);
  M.nprio = 0;
  for (int pass = prio ? 0 : 1; pass < 2; pass++)
    for (CeedInt o = 0; o < r->lsize; o++) {
      if (!cnt[o]) continue;
      if (prio && ((prio[o] != 0) != (pass == 0))) continue;
      slot[o] = (uint32_t)M.h_node_off.size();
      M.h_node_off.push_back((uint32_t)o);
      rowptr.push_back(rowptr.back() + cnt[o]);
      if (prio && pass == 0) M.nprio++;
    }
  const int nn = (int)M.h_node_off.size();
  std::vector<uint32_t> cursor(rowptr.begin(), rowptr.end() - 1), cols(n ? n : 1);
  for (size_t i = 0; i < n; i++) {  // element order => each node's contributors are sorted by element
    const uint32_t sl = slot[(size_t)r->h_offsets[i]];
    if (sl == 0xFFFFFFFFu || (nflag && (nflag[i] & 1))) continue;
    // E position: e * elemsize + n, or in the shell-only E-vector of the direct-store mode e * shell size + shell rank
    const size_t e = i / (size_t)r->elemsize; const int ln = (int)(i % (size_t)r->elemsize);
    cols[cursor[sl]++] = skipP > 0 ? (uint32_t)(e * (size_t)element_shell_size(skipP) + (size_t)node_shell_rank(ln, skipP)) : (uint32_t)i;
  }
  M.nnodes = nn;
  // every L-vector entry is written by the assembly (or, for the skipped nodes, by the fused kernel)
  M.full_cover = ((size_t)nn + (size_t)M.nskipped) * (size_t)r->ncomp == (size_t)r->lsize;
  HIPCHK(hipMalloc((void **)&M.d_rowptr, sizeof(uint32_t) * (nn + 1)));
  HIPCHK(hipMalloc((void **)&M.d_cols, sizeof(uint32_t) * cols.size()));
  HIPCHK(hipMalloc((void **)&M.d_node_off, sizeof(uint32_t) * (nn ? nn : 1)));
  HIPCHK(hipMemcpy(M.d_rowptr, rowptr.data(), sizeof(uint32_t) * (nn + 1), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(M.d_cols, cols.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(M.d_node_off, M.h_node_off.data(), sizeof(uint32_t) * nn, hipMemcpyHostToDevice));
  M.built = true;
  return 0;
}
// Are the element-interior nodes (local index 0 < i,j,k < P-1) of an offsets restriction private to their
// element?  True for every conforming mesh; checked because offsets are caller data.
static bool rstr_interior_private(CeedElemRestriction r, int P) {
  if (r->interior_private) return r->interior_private > 0;
  r->interior_private = -1;
  if (P < 3 || (size_t)P * P * P != (size_t)r->elemsize || r->ncomp != 3 || r->compstride != 1) return false;
  std::vector<unsigned char> cnt((size_t)r->lsize, 0);
  for (size_t i = 0; i < r->h_offsets.size(); i++) {
    unsigned char &c = cnt[(size_t)r->h_offsets[i]];
    if (c < 2) c++;
  }
  for (size_t i = 0; i < r->h_offsets.size(); i++)
    if (node_is_element_interior((int)(i % (size_t)r->elemsize), P) && cnt[(size_t)r->h_offsets[i]] != 1) return false;
  r->interior_private = 1;
  return true;
}
// Pair merge: for every group of two consecutive elements, the shell nodes they share (FusedGradArgs::pairs).
static int build_pairs(CeedElemRestriction r, int P) {
  if (r->pair_state) return 0;
  r->pair_state = -1;
  if (!r->ceed->pair_merge) return 0;
  const int P3 = P * P * P;
  if (P < 3 || P3 != r->elemsize || P3 > 255 || (uint32_t)r->lsize > PAIR_OFF_MASK || r->nelem < 2) return 0;
  const size_t n = r->h_offsets.size();
  std::vector<unsigned char> mult((size_t)r->lsize, 0);
  for (size_t i = 0; i < n; i++) { unsigned char &m = mult[(size_t)r->h_offsets[i]]; if (m < 255) m++; }
  const int ngroups = (r->nelem + 1) / 2;
  std::vector<uint16_t> pairs((size_t)ngroups * PAIR_MAX, (uint16_t)0xFFFF);
  r->h_nflag.assign(n, 0);
  std::vector<int32_t> stamp((size_t)r->lsize, -1);   // (group << 8 | local node of the first element) + 1 ... as int64 would be safer; groups < 2^23
  if (ngroups >= (1 << 22)) return 0;
  for (int g = 0; g < ngroups; g++) {
    const size_t e0 = (size_t)2 * g, e1 = e0 + 1;
    if (e1 >= (size_t)r->nelem) break;
    for (int k = 0; k < P3; k++)
      if (!node_is_element_interior(k, P)) stamp[(size_t)r->h_offsets[e0 * P3 + k]] = (int32_t)((g << 8) | k);
    int np = 0;
    for (int k = 0; k < P3 && np < PAIR_MAX; k++) {
      if (node_is_element_interior(k, P)) continue;
      const size_t o = (size_t)r->h_offsets[e1 * P3 + k];
      const int32_t st = stamp[o];
      if (st < 0 || (st >> 8) != g) continue;
      const int n0 = st & 0xFF;
      if (r->h_nflag[e0 * P3 + n0] & 4) continue;   // (a node the second element holds twice: keep the first match only)
      pairs[(size_t)g * PAIR_MAX + np++] = (uint16_t)(n0 | (k << 8));
      r->h_nflag[e1 * P3 + k] |= 1;
      r->h_nflag[e0 * P3 + n0] |= 4;
      r->pair_nodes++;
      if (mult[o] == 2) { r->h_nflag[e0 * P3 + n0] |= 2; r->pair_complete++; }
    }
  }
  HIPCHK(hipMalloc((void **)&r->d_pairs, sizeof(uint16_t) * pairs.size()));
  HIPCHK(hipMemcpy(r->d_pairs, pairs.data(), sizeof(uint16_t) * pairs.size(), hipMemcpyHostToDevice));
  r->pair_state = 1;
  return 0;
}

// Re-order the transpose map `M` (shell or full) of restriction r for the gated assembly with groups of E elements.
static int build_gated(CeedElemRestriction r, const CsrMap &M, int E, int skipP, int item_rows, int max_contrib, GatedMap &G) {
  if (G.built && G.E == E && G.skipP == skipP && G.item_rows == item_rows && G.max_contrib == max_contrib) return 0;
  if (r->ceed->capturing)
    return ceed_error("first apply of an operator during graph capture: its restriction's transpose map is built on the host; "
                      "apply the operator once before recording");
  G.release();
  G.E = E; G.skipP = skipP; G.nskipped = M.nskipped; G.full_cover = M.full_cover; G.item_rows = item_rows; G.max_contrib = max_contrib;
  const int per_elem = skipP > 0 ? element_shell_size(skipP) : r->elemsize;
  G.evec_stride = ((3 * per_elem * 8 + 127) / 128) * 128 / 8;
  if ((size_t)r->nelem * (size_t)G.evec_stride > 0xFFFFFFFFull) return ceed_error("E-vector of %d elements exceeds the 32-bit index of the gated transpose map", r->nelem);
  const int ngroups = (r->nelem + E - 1) / E, chunk = (ngroups + 7) / 8;
  G.bucket_shift = 5;   // 32 groups per bucket
  const int BG = 1 << G.bucket_shift;
  G.nb = std::max(1, (chunk + BG - 1) / BG);
  // download-free: the host copies of rowptr / cols are rebuilt from the restriction (same counting sort as build_csr)
  const int nn = M.nnodes;
  std::vector<uint32_t> rowptr((size_t)nn + 1), cols;
  {
    HIPCHK(hipMemcpy(rowptr.data(), M.d_rowptr, sizeof(uint32_t) * ((size_t)nn + 1), hipMemcpyDeviceToHost));
    cols.resize(rowptr[nn] ? rowptr[nn] : 1);
    HIPCHK(hipMemcpy(cols.data(), M.d_cols, sizeof(uint32_t) * rowptr[nn], hipMemcpyDeviceToHost));
  }
  // key of a row: chunk * nb + bucket of its last contributor, or the cut key 8 * nb
  const uint32_t cutkey = 8u * (uint32_t)G.nb;
  std::vector<uint32_t> key((size_t)nn), cnt((size_t)cutkey + 2, 0u);
  for (int i = 0; i < nn; i++) {
    const uint32_t efirst = cols[rowptr[i]] / (uint32_t)per_elem, elast = cols[rowptr[i + 1] - 1] / (uint32_t)per_elem;   // element order
    const uint32_t gf = efirst / (uint32_t)E, gl = elast / (uint32_t)E, cf = gf / (uint32_t)chunk, cl = gl / (uint32_t)chunk;
    key[i] = cf == cl ? cl * (uint32_t)G.nb + ((gl - cl * (uint32_t)chunk) >> G.bucket_shift) : cutkey;
    if (max_contrib > 0 && rowptr[i + 1] - rowptr[i] > (uint32_t)max_contrib) key[i] = cutkey;   // folded form: more than four contributors (vertices, irregular nodes) are the tail kernel's
    cnt[key[i] + 1]++;
  }
  for (size_t k = 0; k + 1 < cnt.size(); k++) cnt[k + 1] += cnt[k];   // cnt[k] = first new row of key k
  std::vector<uint32_t> order((size_t)nn), cursor(cnt.begin(), cnt.end() - 1);
  for (int i = 0; i < nn; i++) order[cursor[key[i]]++] = (uint32_t)i;    // stable: ascending node offset within a bucket
  std::vector<uint32_t> rp2((size_t)nn + 1, 0u), cols2(cols.size()), no2((size_t)(nn ? nn : 1));
  G.h_node_off.resize((size_t)nn);
  for (int j = 0; j < nn; j++) {
    const uint32_t i = order[j];
    const uint32_t len = rowptr[i + 1] - rowptr[i];
    for (uint32_t k = 0; k < len; k++) {   // position e * per_elem + rank  ->  double index in the line-padded E-vector
      const uint32_t pos = cols[rowptr[i] + k];
      cols2[rp2[j] + k] = (pos / (uint32_t)per_elem) * (uint32_t)G.evec_stride + (pos % (uint32_t)per_elem) * 3u;
    }
    rp2[j + 1] = rp2[j] + len;
    no2[j] = G.h_node_off[j] = M.h_node_off[i];
  }
  G.nrows = nn; G.nrows_local = (int)cnt[cutkey];
  // items: pieces of <= GATED_ITEM_ROWS rows of one bucket, per chunk in bucket order
  std::vector<uint32_t> item_row, item_bucket, bgroups((size_t)cutkey, 0u), bitems((size_t)cutkey, 0u);
  for (int c = 0; c < 8; c++) {
    G.item_begin[c] = (int)item_bucket.size();
    const int cg = std::max(0, std::min(ngroups, (c + 1) * chunk) - c * chunk);   // groups of this chunk
    for (int b = 0; b < G.nb; b++) {
      bgroups[(size_t)c * G.nb + b] = (uint32_t)std::max(0, std::min(BG, cg - b * BG));
      const uint32_t k = (uint32_t)c * (uint32_t)G.nb + (uint32_t)b;
      for (uint32_t r0 = cnt[k]; r0 < cnt[k + 1]; r0 += (uint32_t)item_rows) { item_row.push_back(r0); item_bucket.push_back((uint32_t)b); }
      bitems[(size_t)c * G.nb + b] = (uint32_t)((int)item_bucket.size() - G.item_begin[c]);   // items waiting for bucket <= b
    }
  }
  G.item_begin[8] = (int)item_bucket.size();
  G.nitems = (int)item_bucket.size();
  // an item ends where the next begins -- also across bucket and chunk boundaries, since rows are contiguous in key order
  item_row.push_back((uint32_t)G.nrows_local);
  if (item_bucket.empty()) item_bucket.push_back(0u);
  auto up = [](uint32_t **dst, const std::vector<uint32_t> &v) -> int {
    HIPCHK(hipMalloc((void **)dst, sizeof(uint32_t) * (v.size() ? v.size() : 1)));
    if (!v.empty()) HIPCHK(hipMemcpy(*dst, v.data(), sizeof(uint32_t) * v.size(), hipMemcpyHostToDevice));
    return 0;
  };
  CHK(up(&G.d_rowptr, rp2)); CHK(up(&G.d_cols, cols2)); CHK(up(&G.d_node_off, no2)); CHK(up(&G.d_item_row, item_row));
  CHK(up(&G.d_item_bucket, item_bucket)); CHK(up(&G.d_bucket_groups, bgroups)); CHK(up(&G.d_bucket_items, bitems));
  const size_t nctrl = (size_t)GatedCtrl::size(G.nb, G.nitems);
  HIPCHK(hipMalloc((void **)&G.d_ctrl, sizeof(unsigned) * nctrl));
  HIPCHK(hipMemset(G.d_ctrl, 0, sizeof(unsigned) * nctrl));
  G.built = true;
  return 0;
}
// Segments of the pipelined assembly: element ranges whose group counts are whole rounds of the fused kernel's persistent
// waves (`waves` per launch) where the mesh is large enough for that -- a launch then ends with every wave finishing its
// last group at about the same time -- and the rows of the map sorted by the segment of their last contributor.
static int build_pipe(CeedElemRestriction r, const CsrMap &M, int E, int per_elem, int req_seg, int waves, PipeMap &G) {
  if (G.built && G.E == E && G.req_seg == req_seg && G.waves == waves && G.base == (const void *)&M) return 0;
  if (getenv("CEED_MI355X_PIPE_DEBUG")) fprintf(stderr, "build_pipe: %d elements, E %d, %d segments asked, %d waves\n", r->nelem, E, req_seg, waves);
  if (r->ceed->capturing)
    return ceed_error("first apply of an operator during graph capture: its restriction's transpose map is built on the host; "
                      "apply the operator once before recording");
  G.release();
  G.E = E; G.req_seg = req_seg; G.waves = waves; G.base = (const void *)&M;
  const int ngroups = (r->nelem + E - 1) / E;
  // at least `min_rounds` rounds per segment, else fewer segments (down to one: the caller then takes the serial path)
  const int min_rounds = r->ceed->pipe_min_rounds;
  // Below ~20 rounds of the persistent waves the fixed cost of the form (fork and join of the second stream, the summing
  // kernels competing with the fused kernel for memory: ~40 us at p = 4) exceeds what is hidden: measured -3 % at 24 rounds
  // (99 000 hexes, p = 4), +7 % at 11 rounds (44 928 hexes) -- such launches keep the serial form.
  if (min_rounds > 0 && ngroups < r->ceed->pipe_min_total_rounds * std::max(waves, 1)) req_seg = 1;
  // Segments asked for = 0: one per ~90 MB of E-vector (3 for config 4's 233 MB, 5 for twice that mesh, 15 for the whole of
  // config 5) -- a segment boundary costs ~10 us, and the smaller a segment the more of its E-vector is still in the 256 MB
  // last-level cache when its rows are summed (config 5, 1.4 GB of E-vector: 4.27 ms serial, 4.00 with 3 segments, 3.57
  // with 8, 3.42 with 12-16; config 4: 3 segments best, 4 already slower).
  else if (req_seg == 0) req_seg = std::max(2, std::min(16, (int)((double)r->nelem * per_elem * 24. / 90e6 + 0.5)));
  int nseg = min_rounds > 0 ? std::max(1, std::min(req_seg, ngroups / (min_rounds * std::max(waves, 1)))) : std::min(req_seg, std::max(1, ngroups));
  // Boundaries are laid out FROM THE END in whole rounds of the waves: the last segment (whose rows are summed with nothing
  // to hide behind) is `last_rounds` rounds, the others share the rest equally in whole rounds, and the odd remainder of the
  // mesh lands in the FIRST segment, where the next fused kernel fills the chip behind its ragged last round.
  G.elem_bound.assign(1, 0);
  const int last_rounds = r->ceed->pipe_last_rounds;
  const long total_rounds = ngroups / std::max(waves, 1);
  std::vector<long> gb;        // group boundaries, descending
  if (min_rounds > 0 && last_rounds > 0 && nseg >= 2 && total_rounds >= last_rounds + (long)(nseg - 1) * min_rounds) {
    long g = (long)ngroups - (long)last_rounds * waves;
    gb.push_back(g);
    const long per = (total_rounds - last_rounds) / (nseg - 1);       // rounds of the middle segments
    for (int k = nseg - 2; k >= 1; k--) { g -= per * waves; gb.push_back(g); }
  } else {
    for (int k = nseg - 1; k >= 1; k--) {
      long g = (long)ngroups * k / nseg;
      const long up = (long)ngroups - (((long)ngroups - g) / waves) * waves;           // whole rounds behind it, if that moves it sensibly
      gb.push_back(min_rounds > 0 && up > 0 && up < ngroups ? up : g);
    }
  }
  for (auto it = gb.rbegin(); it != gb.rend(); ++it) {
    const int e = (int)std::min<long>((long)r->nelem, *it * E);
    if (e > G.elem_bound.back() && e < r->nelem) G.elem_bound.push_back(e);
  }
  G.elem_bound.push_back(r->nelem);
  nseg = (int)G.elem_bound.size() - 1;
  G.nseg = nseg;
  const int nn = M.nnodes;
  std::vector<uint32_t> rowptr((size_t)nn + 1), cols;
  HIPCHK(hipMemcpy(rowptr.data(), M.d_rowptr, sizeof(uint32_t) * ((size_t)nn + 1), hipMemcpyDeviceToHost));
  cols.resize(rowptr[nn] ? rowptr[nn] : 1);
  HIPCHK(hipMemcpy(cols.data(), M.d_cols, sizeof(uint32_t) * rowptr[nn], hipMemcpyDeviceToHost));
  std::vector<int> seg((size_t)nn);
  std::vector<uint32_t> cnt((size_t)nseg + 1, 0u);
  for (int i = 0; i < nn; i++) {
    const int elast = (int)(cols[rowptr[i + 1] - 1] / (uint32_t)per_elem);      // contributors are in element order
    const int k = (int)(std::upper_bound(G.elem_bound.begin(), G.elem_bound.end(), elast) - G.elem_bound.begin()) - 1;
    seg[i] = k; cnt[(size_t)k + 1]++;
  }
  for (int k = 0; k < nseg; k++) cnt[k + 1] += cnt[k];
  G.row_bound.assign(cnt.begin(), cnt.end());
  std::vector<uint32_t> cursor(cnt.begin(), cnt.end() - 1), order((size_t)nn);
  for (int i = 0; i < nn; i++) order[cursor[seg[i]]++] = (uint32_t)i;   // stable: ascending node offset within a segment
  std::vector<uint32_t> rp2((size_t)nn + 1, 0u), cols2(cols.size()), no2((size_t)(nn ? nn : 1));
  G.h_node_off.resize((size_t)nn);
  for (int j = 0; j < nn; j++) {
    const uint32_t i = order[j], len = rowptr[i + 1] - rowptr[i];
    for (uint32_t k = 0; k < len; k++) cols2[rp2[j] + k] = cols[rowptr[i] + k];
    rp2[j + 1] = rp2[j] + len;
    no2[j] = G.h_node_off[j] = M.h_node_off[i];
  }
  G.nrows = nn;
  if (getenv("CEED_MI355X_PIPE_DEBUG"))
    for (int k = 0; k < nseg; k++)
      fprintf(stderr, "  segment %d: elements %d..%d (%.2f rounds), rows %d..%d\n", k, G.elem_bound[k], G.elem_bound[k + 1],
              (double)(G.elem_bound[k + 1] - G.elem_bound[k]) / E / waves, G.row_bound[k], G.row_bound[k + 1]);
  auto up = [](uint32_t **dst, const std::vector<uint32_t> &v) -> int {
    HIPCHK(hipMalloc((void **)dst, sizeof(uint32_t) * (v.size() ? v.size() : 1)));
    if (!v.empty()) HIPCHK(hipMemcpy(*dst, v.data(), sizeof(uint32_t) * v.size(), hipMemcpyHostToDevice));
    return 0;
  };
  CHK(up(&G.d_rowptr, rp2)); CHK(up(&G.d_cols, cols2)); CHK(up(&G.d_node_off, no2));
  G.built = true;
  G.build_id++;
  return 0;
}
static int ceed_need_evec(Ceed c, size_t len) {
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
static bool is_offsets(CeedElemRestriction r) { return r && r != CEED_ELEMRESTRICTION_NONE && !r->strided; }
static bool is_strided(CeedElemRestriction r) { return r && r != CEED_ELEMRESTRICTION_NONE && r->strided; }

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
    if (b->qmode != CEED_GAUSS && P == Q) { /* fine: any rule works, the tables carry it */ }
    fill_tables(op->tables, b);
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

struct TimerScope {
  CeedOperator op; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
  TimerScope(CeedOperator o, hipStream_t st) : op(o), s(st) {
    if (op->timing && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, s);
  }
  ~TimerScope() {
    if (a && b) { (void)hipEventRecord(b, s); op->events.emplace_back(a, b); }
  }
};

// The residual / Jacobian operator: k_fused_grad (+ k_assemble).  phase -1: whole apply; phase 0 / 1:
// the two halves of a split-phase apply (CeedXOperatorApplyPhase).
static int apply_fused_grad(CeedOperator op, CeedVector in, CeedVector out, bool add, int phase, const char **kname) {
  CeedQFunction qf = op->qf;
  hipStream_t s = op->ceed->stream;
  OpField &ai = op->in[op->i_active];
  CeedElemRestriction r = ai.rstr;
  if (!in || in == CEED_VECTOR_NONE || !out || out == CEED_VECTOR_NONE) return ceed_error("active vectors required");
  if (in->length < r->lsize || out->length < r->lsize) return ceed_error("active vector shorter than the restriction's L-size");
  if (in == out) return ceed_error("in-place operator apply is not supported");
  FusedGradArgs a{};
  double *px, *py, *pq, *ps = nullptr;
  CHK(vec_dev(in, false, &px));
  CHK(vec_dev(out, true, &py));
  CHK(vec_dev(op->in[op->i_qdata].vec, false, &pq));
  a.offsets = op->d_off_flagged_in ? op->d_off_flagged_in : r->d_offsets;
  a.x = px; a.y = py; a.qdata = pq;
  if (op->i_state >= 0) { CHK(vec_dev(op->in[op->i_state].vec, false, &ps)); a.state_in = ps; }
  if (op->o_state >= 0) {
    CeedVector sv = op->out[op->o_state].vec;
    if (!sv || sv == CEED_VECTOR_NONE || sv == CEED_VECTOR_ACTIVE) return ceed_error("state output needs a passive vector");
    CHK(vec_dev(sv, true, &ps)); a.state_out = ps;  // every point is overwritten
  }
  a.mask_in = (op->mask_mode & 1) ? 1 : 0; a.mask_out = (op->mask_mode & 2) ? 1 : 0;
  if (op->ceed->even_odd && ai.basis->Q1d >= 4) {   // even-odd tables, if all six are (anti)symmetric and small enough
    // (below 4 x 4 the additions cost what the halved products save: measured -1.6 % at Q = 3)
    const int Pn = ai.basis->P1d, Qn = ai.basis->Q1d;
    const BasisTables &t = op->tables;
    a.eo_ok = build_eo_table(t.interp, Qn, Pn, Pn, false, +1, a.eo[0]) && build_eo_table(t.interp, Pn, Qn, Pn, true, +1, a.eo[1]) &&
              build_eo_table(t.colo, Qn, Qn, Qn, false, -1, a.eo[2]) && build_eo_table(t.colo, Qn, Qn, Qn, true, -1, a.eo[3]) &&
              build_eo_table(t.grad, Qn, Pn, Pn, false, -1, a.eo[4]) && build_eo_table(t.grad, Pn, Qn, Pn, true, -1, a.eo[5]);
  }
  {  // geometric factors recomputed in the kernel if the qdata vector still is what SetupGeo wrote on these elements
    CeedVector qv = op->in[op->i_qdata].vec;
    bool same_rule = qv->geo && qv->geo_nelem == r->nelem && qv->geo_Q == ai.basis->Q1d;
    for (int i = 0; same_rule && i < ai.basis->Q1d; i++)
      same_rule = qv->geo_qref[i] == ai.basis->qref1d[i] && qv->geo_qwt[i] == ai.basis->qweight1d[i];
    if (same_rule && op->ceed->recompute_geo) {
      a.geo = qv->geo;
      for (int i = 0; i < ai.basis->Q1d; i++) { a.qref[i] = qv->geo_qref[i]; a.qwt[i] = qv->geo_qwt[i]; }
    }
  }
  CHK(read_phys(qf, &a.nu, &a.E));
  lame_constants(a.nu, a.E, &a.lambda, &a.TwoMu);
  a.stamps = op->stamps;
  a.variant = op->ceed->fused_variant;
  const bool use_evec = !op->ceed->atomic_scatter;
  const bool split = phase >= 0;
  if (split && (!use_evec || add || op->ovl_lead <= 0 || !op->ovl_csr.built))
    return ceed_error("split-phase apply needs CeedXOperatorSetOverlapSplit, the E-vector scatter and overwrite mode");
  // element range and transpose-map rows of this launch
  const CsrMap *M = nullptr;
  int row0 = 0, nrows = 0;
  a.elem_begin = 0; a.nelem = r->nelem;
  // element-interior nodes straight to y: overwrite mode only (split maps are built to match, see SetOverlapSplit)
  const bool direct = use_evec && !add && op->ceed->direct_interior && rstr_interior_private(r, ai.basis->P1d);
  a.direct = direct ? 1 : 0;
  if (use_evec) {  // atomic-free, deterministic scatter: element results -> E-vector -> per-node sums
    if (split) {
      M = &op->ovl_csr;
      if ((M->nskipped > 0) != direct) return ceed_error("split-phase map and direct-store mode disagree");
    } else if (direct) {
      if (a.variant == 1 && pencil_group_elems(ai.basis->Q1d) == 2 && !op->ceed->capturing) CHK(build_pairs(r, ai.basis->P1d));
      const bool paired = r->pair_state == 1 && a.variant == 1 && pencil_group_elems(ai.basis->Q1d) == 2;
      CHK(build_csr(r, r->csr_shell, nullptr, ai.basis->P1d, paired ? r->h_nflag.data() : nullptr));
      M = &r->csr_shell;
      if (paired) {
        if (!op->d_off_paired) {   // offsets + Dirichlet flags (bits 29-31) + pair bits (27, 28)
          std::vector<uint32_t> fl(r->h_offsets.size());
          for (size_t i = 0; i < fl.size(); i++) {
            uint32_t o = (uint32_t)r->h_offsets[i], f = 0;
            if (!op->h_mask.empty())
              for (int cc = 0; cc < 3; cc++) if (op->h_mask[(size_t)o + (size_t)cc * r->compstride]) f |= 1u << cc;
            fl[i] = o | (f << OFF_FLAG_SHIFT) | ((r->h_nflag[i] & 1) ? PAIR_SKIP : 0u) | ((r->h_nflag[i] & 2) ? PAIR_DIRECT : 0u);
          }
          HIPCHK(hipMalloc((void **)&op->d_off_paired, sizeof(uint32_t) * (fl.size() ? fl.size() : 1)));
          HIPCHK(hipMemcpy(op->d_off_paired, fl.data(), sizeof(uint32_t) * fl.size(), hipMemcpyHostToDevice));
        }
        a.offsets = op->d_off_paired;
        a.pairs = r->d_pairs;
      }
    }
    else { CHK(build_csr(r, r->csr, nullptr)); M = &r->csr; }
    unsigned char **flagsp = split ? &op->d_node_flags_ovl : (direct ? &op->d_node_flags_shell : &op->d_node_flags);
    if (!*flagsp && !op->h_mask.empty()) {
      std::vector<unsigned char> fl((size_t)M->nnodes, 0);
      for (int i = 0; i < M->nnodes; i++)
        for (int c = 0; c < r->ncomp && c < 3; c++)
          if (op->h_mask[(size_t)M->h_node_off[i] + (size_t)c * r->compstride]) fl[i] |= (unsigned char)(1u << c);
      HIPCHK(hipMalloc((void **)flagsp, fl.size() ? fl.size() : 1));
      HIPCHK(hipMemcpy(*flagsp, fl.data(), fl.size(), hipMemcpyHostToDevice));
    }
    CHK(ceed_need_evec(op->ceed, (size_t)r->nelem * (((size_t)3 * r->elemsize * 8 + 127) / 128 * 16)));   // line-padded blocks at most
    a.evec = op->ceed->evec;
    a.evec_stride = 3 * (direct ? element_shell_size(ai.basis->P1d) : r->elemsize);
    row0 = 0; nrows = M->nnodes;
    if (split) {
      if (phase == 0) { a.nelem = op->ovl_lead; nrows = M->nprio; }
      else { a.elem_begin = op->ovl_lead; a.nelem = r->nelem - op->ovl_lead; row0 = M->nprio; nrows = M->nnodes - M->nprio; }
    }
    if (!add && !M->full_cover && phase <= 0) CHK(dev_zero(op->ceed, py, (size_t)out->length));
  } else if (!add) {
    CHK(dev_zero(op->ceed, py, (size_t)out->length));
  }
  {
    TimerScope ts(op, s);
    Ceed c = op->ceed;
    // gated assembly: whole applies in overwrite mode through the pencil kernel
    // folded assembly is compiled into the pencil kernel up to Q = 5; beyond, the serial assembly unless "gated" was asked for
    const bool gated = c->gated_assembly && use_evec && !add && !split && a.variant == 1 && (!c->folded_assembly || ai.basis->Q1d <= 5);
    GatedAsmArgs ga{};
    if (gated) {
      GatedMap &G = r->gated;
      const bool fold_form = c->folded_assembly;
      CHK(build_gated(r, *M, pencil_group_elems(ai.basis->Q1d), direct ? ai.basis->P1d : 0, fold_form ? GATED_ITEM_ROWS : GATED_ITEM_ROWS_BESIDE, fold_form ? 4 : 0, G));
      if (!op->d_node_flags_gated && op->h_mask.empty()) {   // no Dirichlet mask: all-zero flags (the folded stages read them unconditionally)
        HIPCHK(hipMalloc((void **)&op->d_node_flags_gated, (size_t)(G.nrows ? G.nrows : 1)));
        HIPCHK(hipMemset(op->d_node_flags_gated, 0, (size_t)(G.nrows ? G.nrows : 1)));
      }
      if (!op->d_node_flags_gated && !op->h_mask.empty()) {
        std::vector<unsigned char> fl((size_t)G.nrows, 0);
        for (int i = 0; i < G.nrows; i++)
          for (int cc = 0; cc < r->ncomp && cc < 3; cc++)
            if (op->h_mask[(size_t)G.h_node_off[i] + (size_t)cc * r->compstride]) fl[i] |= (unsigned char)(1u << cc);
        HIPCHK(hipMalloc((void **)&op->d_node_flags_gated, fl.size() ? fl.size() : 1));
        HIPCHK(hipMemcpy(op->d_node_flags_gated, fl.data(), fl.size(), hipMemcpyHostToDevice));
      }
      a.queue = G.d_ctrl + GatedCtrl::QUEUE; a.done = G.d_ctrl + GatedCtrl::DONE; a.nb = G.nb; a.bucket_shift = G.bucket_shift;
      a.evec_stride = G.evec_stride;
      if (getenv("CEED_MI355X_GATED_DEBUG") && (atoi(getenv("CEED_MI355X_GATED_DEBUG")) & 4)) a.done = nullptr;
      if (getenv("CEED_MI355X_GATED_DEBUG") && (atoi(getenv("CEED_MI355X_GATED_DEBUG")) & 8)) a.queue = nullptr;
      ga.rowptr = G.d_rowptr; ga.cols = G.d_cols; ga.node_off = G.d_node_off;
      ga.flags = (op->mask_mode & 2) ? op->d_node_flags_gated : nullptr;
      ga.evec = a.evec; ga.y = py; ga.ctrl = G.d_ctrl; ga.item_row = G.d_item_row; ga.item_bucket = G.d_item_bucket;
      ga.bucket_groups = G.d_bucket_groups; ga.bucket_items = G.d_bucket_items;
      for (int i = 0; i < 9; i++) ga.item_begin[i] = G.item_begin[i];
      ga.nb = G.nb; ga.nitems = G.nitems; ga.nrows_local = G.nrows_local; ga.nrows = G.nrows;
      ga.max_spins = c->gated_spins; ga.item_rows = G.item_rows;
      if (c->folded_assembly) {
        a.as_rowptr = G.d_rowptr; a.as_cols = G.d_cols; a.as_node_off = G.d_node_off; a.as_item_row = G.d_item_row;
        a.as_bucket_items = G.d_bucket_items; a.as_flags = op->d_node_flags_gated; a.as_max_spins = c->gated_spins;
        for (int i = 0; i < 9; i++) a.as_item_begin[i] = G.item_begin[i];
        a.as_dbg = getenv("CEED_MI355X_FOLD_DBG") ? atoi(getenv("CEED_MI355X_FOLD_DBG")) : 0;
      }
      ga.dbg = getenv("CEED_MI355X_GATED_KDBG") ? atoi(getenv("CEED_MI355X_GATED_KDBG")) : 0;
    } else if (c->dynamic_sched && a.variant == 1 && use_evec) {   // tickets zeroed at allocation and by every k_assemble behind the fused kernel
      if (!c->queue) {
        HIPCHK(hipMalloc((void **)&c->queue, sizeof(unsigned) * 8 * QUEUE_STRIDE));
        HIPCHK(hipMemset(c->queue, 0, sizeof(unsigned) * 8 * QUEUE_STRIDE));
      }
      a.queue = c->queue;
      static const int qd = getenv("CEED_MI355X_DBG") ? atoi(getenv("CEED_MI355X_DBG")) : 0;   // bring-up
      if (qd & 1) { static unsigned *dummy = nullptr; if (!dummy) { HIPCHK(hipMalloc((void **)&dummy, 4096)); HIPCHK(hipMemset(dummy, 0, 4096)); } a.done = dummy; a.nb = 1; a.bucket_shift = 30; }
      if (qd & 2) CHK(dev_zero(c, (double *)c->queue, 8 * QUEUE_STRIDE / 2));
    }
    // pipelined assembly: whole applies in overwrite mode through the pencil kernel, large enough for two segments
    PipeMap *PM = nullptr;
    if (c->pipe_segments != 0 && !gated && use_evec && !add && !split && a.variant == 1 && !a.pairs && !a.queue) {
      int waves = 0;            // persistent waves of a full launch of THIS kernel (LDS-limited from Q = 6 on)
      {
        FusedGradArgs aq = a;
        aq.query_waves = &waves;
        const char *nm = "";
        HIPCHK(launch_fused_grad(ai.basis->P1d, ai.basis->Q1d, qf->kind, op->tables, aq, s, &nm));
        if (waves <= 0) return ceed_error("pipelined assembly: no persistent-wave count for P=%d Q=%d", ai.basis->P1d, ai.basis->Q1d);
      }
      const int per_elem = direct ? element_shell_size(ai.basis->P1d) : r->elemsize;
      if (!r->pipe.built && c->capturing) { /* cold map while recording: the serial path (its map exists) */ }
      else {
        CHK(build_pipe(r, *M, pencil_group_elems(ai.basis->Q1d), per_elem, std::max(c->pipe_segments, 0), waves, r->pipe));
        if (r->pipe.nseg >= 2) PM = &r->pipe;
      }
      if (PM && op->d_node_flags_pipe && op->pipe_flags_id != PM->build_id) {     // the map was rebuilt (another operator's wave count): new row order
        HIPCHK(hipStreamSynchronize(s));
        (void)hipFree(op->d_node_flags_pipe);
        op->d_node_flags_pipe = nullptr;
      }
      if (PM && !op->d_node_flags_pipe && !op->h_mask.empty()) {
        op->pipe_flags_id = PM->build_id;
        std::vector<unsigned char> fl((size_t)PM->nrows, 0);
        for (int i = 0; i < PM->nrows; i++)
          for (int cc = 0; cc < r->ncomp && cc < 3; cc++)
            if (op->h_mask[(size_t)PM->h_node_off[i] + (size_t)cc * r->compstride]) fl[i] |= (unsigned char)(1u << cc);
        HIPCHK(hipMalloc((void **)&op->d_node_flags_pipe, fl.size() ? fl.size() : 1));
        HIPCHK(hipMemcpy(op->d_node_flags_pipe, fl.data(), fl.size(), hipMemcpyHostToDevice));
      }
    }
    if (PM) {
      // segment k: fused kernel on the Ceed's stream, then its rows on the side stream beside the fused kernel of segment
      // k + 1; the last segment's rows on the Ceed's stream again, which then waits for the side stream.  Every row is summed
      // in contributor order by one thread, whatever the segment: bitwise the serial result.
      if (!c->side_stream) {
        HIPCHK(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
      }
      const unsigned char *fl = (op->mask_mode & 2) ? op->d_node_flags_pipe : nullptr;
      const int nseg = PM->nseg;
      op->launch_info[0] = nseg; op->launch_info[1] = c->pipe_chains ? 2 : 1; op->launch_info[2] = nseg;
      op->launch_info[3] = PM->elem_bound[nseg] - PM->elem_bound[nseg - 1];
      if (c->pipe_chains) {
        // two chains: segment k's fused kernel AND its rows on stream k % 2 -- the fused kernel of segment k + 1 sits in the
        // other queue and fills the chip as the waves of segment k retire (no kernel boundary between fused kernels)
        HIPCHK(hipEventRecord(c->ev_fork, s));
        HIPCHK(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
        for (int k = 0; k < nseg; k++) {
          hipStream_t sk = (k & 1) ? c->side_stream : s;     // (the FIRST segment on the operator's own stream; putting the last one there instead, so that the join is never waited for, measured 6-9 % slower at even segment counts)
          FusedGradArgs ak = a;
          ak.elem_begin = PM->elem_bound[k]; ak.nelem = PM->elem_bound[k + 1] - PM->elem_bound[k];
          hipError_t e = launch_fused_grad(ai.basis->P1d, ai.basis->Q1d, qf->kind, op->tables, ak, sk, kname);
          if (e == hipErrorInvalidValue && !**kname)
            return ceed_error("no fused kernel instantiated for P=%d Q=%d QFunction %s", ai.basis->P1d, ai.basis->Q1d, qf->name.c_str());
          HIPCHK(e);
          // the rows of segment k have contributors in EARLIER segments too (the nodes on the cut between two segments):
          // segment k - 1's fused kernel runs on the other stream, the ones before it precede one of the two in stream order
          if (!c->ev_seg[k]) HIPCHK(hipEventCreateWithFlags(&c->ev_seg[k], hipEventDisableTiming));
          HIPCHK(hipEventRecord(c->ev_seg[k], sk));
          if (k >= 1) HIPCHK(hipStreamWaitEvent(sk, c->ev_seg[k - 1], 0));
          const int r0 = PM->row_bound[k], nr = PM->row_bound[k + 1] - r0;
          HIPCHK(launch_assemble(PM->d_rowptr + r0, PM->d_cols, PM->d_node_off + r0, fl ? fl + r0 : nullptr, a.evec, py, nr, r->elemsize, 0,
                                 sk, nullptr, k + 1 < nseg ? c->pipe_blocks : 0));
        }
        HIPCHK(hipEventRecord(c->ev_join, c->side_stream));
        HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
        op->launches++;
        return 0;
      }
      for (int k = 0; k < nseg; k++) {
        FusedGradArgs ak = a;
        ak.elem_begin = PM->elem_bound[k]; ak.nelem = PM->elem_bound[k + 1] - PM->elem_bound[k];
        hipError_t e = launch_fused_grad(ai.basis->P1d, ai.basis->Q1d, qf->kind, op->tables, ak, s, kname);
        if (e == hipErrorInvalidValue && !**kname)
          return ceed_error("no fused kernel instantiated for P=%d Q=%d QFunction %s", ai.basis->P1d, ai.basis->Q1d, qf->name.c_str());
        HIPCHK(e);
        const int r0 = PM->row_bound[k], nr = PM->row_bound[k + 1] - r0;
        if (k + 1 < nseg) {
          if (!c->ev_seg[k]) HIPCHK(hipEventCreateWithFlags(&c->ev_seg[k], hipEventDisableTiming));
          HIPCHK(hipEventRecord(c->ev_seg[k], s));
          HIPCHK(hipStreamWaitEvent(c->side_stream, c->ev_seg[k], 0));
          HIPCHK(launch_assemble(PM->d_rowptr + r0, PM->d_cols, PM->d_node_off + r0, fl ? fl + r0 : nullptr, a.evec, py, nr, r->elemsize, 0,
                                 c->side_stream, nullptr, c->pipe_blocks));
        } else {
          HIPCHK(hipEventRecord(c->ev_join, c->side_stream));
          HIPCHK(launch_assemble(PM->d_rowptr + r0, PM->d_cols, PM->d_node_off + r0, fl ? fl + r0 : nullptr, a.evec, py, nr, r->elemsize, 0, s, nullptr, 0));
          HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
        }
      }
      op->launches++;
      return 0;
    }
    op->launch_info[0] = 1; op->launch_info[1] = 1; op->launch_info[2] = use_evec ? 1 : 0; op->launch_info[3] = a.nelem;
    const bool folded = gated && c->folded_assembly;
    const bool side = ((gated && !folded) || c->asm_overlap) && use_evec;
    if (side) {   // fork: the assembly runs on a second stream beside the fused kernel (also while a graph is recorded)
      if (!c->side_stream) {
        HIPCHK(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
      }
      HIPCHK(hipEventRecord(c->ev_fork, s));
      HIPCHK(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
    }
    // Gated: the fused kernel FIRST and on the Ceed's stream -- its eight waves per CU must be resident before the assembler's
    // arrive (launched the other way round, the assembler's waves took registers first and the fused kernel ran at
    // reduced occupancy: 431-508 us instead of 406-416) -- the assembler on the side stream, the tail behind the join.
    hipError_t e = launch_fused_grad(ai.basis->P1d, ai.basis->Q1d, qf->kind, op->tables, a, s, kname);
    if (e == hipErrorInvalidValue && !**kname)
      return ceed_error("no fused kernel instantiated for P=%d Q=%d QFunction %s", ai.basis->P1d, ai.basis->Q1d, qf->name.c_str());
    HIPCHK(e);
    if (folded) {
      HIPCHK(launch_assemble_tail(ga, s));   // cut rows, abandoned items, control block reset
    } else if (gated) {
      // launched AFTER the fused kernel, which never waits for it: whatever the queues do, the fused kernel completes;
      // the gated kernel's waits are bounded and the tail kernel finishes whatever it left
      static const int dbg = getenv("CEED_MI355X_GATED_DEBUG") ? atoi(getenv("CEED_MI355X_GATED_DEBUG")) : 0;   // bring-up only
      if (!(dbg & 1)) HIPCHK(launch_assemble_gated(ga, c->gated_waves, c->side_stream));
      HIPCHK(hipEventRecord(c->ev_join, c->side_stream));
      HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
      if (!(dbg & 2)) HIPCHK(launch_assemble_tail(ga, s));
    } else if (use_evec) {  // timed together with the fused kernel: the launches ARE the operator apply
      const unsigned char *fl = (op->mask_mode & 2) ? (split ? op->d_node_flags_ovl : (direct ? op->d_node_flags_shell : op->d_node_flags)) : nullptr;
      HIPCHK(launch_assemble(M->d_rowptr + row0, M->d_cols, M->d_node_off + row0, fl ? fl + row0 : nullptr, a.evec, py,
                             nrows, r->elemsize, add ? 1 : 0, side ? c->side_stream : s, (!gated && a.queue) ? a.queue : nullptr));
      if (side) {
        HIPCHK(hipEventRecord(c->ev_join, c->side_stream));
        HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
      }
    }
  }
  op->launches++;
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
    if (op->ceed->recompute_geo && !op->ceed->capturing && x.basis->P1d == 2 && x.rstr->elemsize == 8 && x.rstr->ncomp == 3 && x.rstr->compstride == 1) {
      HIPCHK(hipMalloc((void **)&out->geo, sizeof(double) * GEO_NCOEF * (size_t)a.nelem));
      HIPCHK(launch_geo_coeffs(a.off_x, px, out->geo, a.nelem, s));
      out->geo_nelem = a.nelem; out->geo_Q = x.basis->Q1d;
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
    double *px, *py, *psc = nullptr;
    CHK(vec_dev(in, false, &px));
    CHK(vec_dev(out, true, &py));
    if (op->scale) { CHK(vec_dev(op->scale, false, &psc)); if (op->scale->length < rf->lsize) return ceed_error("scale vector too short"); }
    // flagged arrays: *_in belongs to the input side's restriction, *_out to the output side's
    const uint32_t *fin = op->d_off_flagged_in, *fout = op->d_off_flagged_out;
    a.off_c = pro ? (fin ? fin : rc->d_offsets) : (fout ? fout : rc->d_offsets);
    a.off_f = pro ? (fout ? fout : rf->d_offsets) : (fin ? fin : rf->d_offsets);
    a.x = px; a.y = py; a.scale_f = psc; a.nelem = rc->nelem;
    a.mask_in = (op->mask_mode & 1) ? 1 : 0; a.mask_out = (op->mask_mode & 2) ? 1 : 0;
    // deterministic scatter, as for the residual / Jacobian: element results -> E-vector -> per-node sums in
    // element order over the OUTPUT restriction's transpose map (masked entries travel as zeros)
    CeedElemRestriction ro = pro ? rf : rc;
    const bool use_evec = !op->ceed->atomic_scatter;
    if (use_evec) {
      CHK(build_csr(ro, ro->csr, nullptr));
      CHK(ceed_need_evec(op->ceed, (size_t)ro->nelem * ro->ncomp * ro->elemsize));
      a.evec = op->ceed->evec;
    }
    if (!add && !(use_evec && ro->csr.full_cover)) CHK(dev_zero(op->ceed, py, (size_t)out->length));
    TimerScope ts(op, s);
    hipError_t e = launch_transfer(b->P1d, b->Q1d, pro, op->tables, a, s, &kname);
    if (e == hipErrorInvalidValue && !*kname) return ceed_error("no transfer kernel for Pc=%d Pf=%d", b->P1d, b->Q1d);
    HIPCHK(e);
    if (use_evec)
      HIPCHK(launch_assemble(ro->csr.d_rowptr, ro->csr.d_cols, ro->csr.d_node_off, nullptr, a.evec, py, ro->csr.nnodes,
                             ro->elemsize, add ? 1 : 0, s));
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
  const bool use_evec = !op->ceed->atomic_scatter;
  if (use_evec) {  // deterministic: element contributions -> E-vector -> per-node sums in element order
    CHK(build_csr(ai.rstr, ai.rstr->csr, nullptr));
    CHK(ceed_need_evec(op->ceed, (size_t)ai.rstr->nelem * ai.rstr->ncomp * ai.rstr->elemsize));
    a.evec = op->ceed->evec;
  }
  const char *kname = "";
  hipError_t e = launch_diag(ai.basis->P1d, ai.basis->Q1d, qf->kind, op->tables, a, s, &kname);
  if (e == hipErrorInvalidValue && !*kname) return ceed_error("no diagonal kernel for P=%d Q=%d %s", ai.basis->P1d, ai.basis->Q1d, qf->name.c_str());
  HIPCHK(e);
  if (use_evec)
    HIPCHK(launch_assemble(ai.rstr->csr.d_rowptr, ai.rstr->csr.d_cols, ai.rstr->csr.d_node_off, nullptr, a.evec, pd,
                           ai.rstr->csr.nnodes, ai.rstr->elemsize, 0, s));
  return 0;
}

// ---------------------------------------------------------------------------
// extensions
// ---------------------------------------------------------------------------
extern "C" int CeedXOperatorGetKernelName(CeedOperator op, const char **name) { *name = op->kernel_name.c_str(); return 0; }

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
    CHK(make_flagged(op->in[0].rstr, mask, lsize, &op->d_off_flagged_in));
    CHK(make_flagged(op->out[0].rstr, mask_out, lsize_out, &op->d_off_flagged_out));
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
  CHK(build_csr(r, op->ovl_csr, priority, (op->ceed->direct_interior && rstr_interior_private(r, P1)) ? P1 : 0));
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
// Gated assembly of the operator's active restriction: out[0] = items of the transpose map, out[1] = rows in them,
// out[2] = cut rows, out[3] = items the tail kernel had to sum (the gated kernel had not), out[4] = applies -- [3], [4]
// since the last call.  Diagnostic only (placement and co-residency decide [3], never the result).
extern "C" int CeedXOperatorGetGatedStats(CeedOperator op, long long out[5]) {
  for (int i = 0; i < 5; i++) out[i] = 0;
  if (op->composite || op->plan != PLAN_FUSED_GRAD) return 0;
  GatedMap &G = op->in[op->i_active].rstr->gated;
  if (!G.built) return 0;
  unsigned st[4] = {0, 0, 0, 0};
  HIPCHK(hipStreamSynchronize(op->ceed->stream));
  unsigned *p = G.d_ctrl + GatedCtrl::stats(G.nb, G.nitems);
  HIPCHK(hipMemcpy(st, p, sizeof st, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(p, 0, sizeof st));
  out[0] = G.nitems; out[1] = G.nrows_local; out[2] = G.nrows - G.nrows_local; out[3] = st[0]; out[4] = st[1];
  return 0;
}

// Diagnostic builds (-DCPS_STAMPS) write 8 s_memtime stamps per wave here; ignored otherwise.
extern "C" int CeedXOperatorSetStampBuffer(CeedOperator op, void *dev) { op->stamps = (unsigned long long *)dev; return 0; }

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

// ---------------------------------------------------------------------------
// Halo exchange over RCCL (the L-vector interface sum of src/matops.c:57 across the GPUs of one node)
// ---------------------------------------------------------------------------
// RCCL is bound at first use with dlopen: a C host gets /opt/rocm's librccl, a Python host the copy torch has already
// loaded (one RCCL per process, like the HIP runtime: see ceed.py).  No link-time dependency for single-GPU users.
namespace {
typedef struct { char internal[128]; } rccl_unique_id;
struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(rccl_unique_id *) = nullptr;
  int (*CommInitRank)(void **, int, rccl_unique_id, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
const int RCCL_FLOAT64 = 8;   // ncclFloat64 (rccl.h)
int rccl_load() {
  if (g_rccl.h) return 0;
  const char *names[] = {"librccl.so.1", "librccl.so"};
  for (int pass = 0; pass < 2 && !g_rccl.h; pass++)      // an already loaded copy first (RTLD_NOLOAD)
    for (const char *n : names)
      if (!g_rccl.h) g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
  if (!g_rccl.h) return ceed_error("the halo exchange needs RCCL (librccl.so.1): %s", dlerror());
  auto sym = [](const char *n) { return dlsym(g_rccl.h, n); };
  g_rccl.GetUniqueId = (int (*)(rccl_unique_id *))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void **, int, rccl_unique_id, int))sym("ncclCommInitRank");
  g_rccl.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
  g_rccl_destroy = g_rccl.CommDestroy;
  g_rccl.GroupStart = (int (*)())sym("ncclGroupStart");
  g_rccl.GroupEnd = (int (*)())sym("ncclGroupEnd");
  g_rccl.Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))sym("ncclSend");
  g_rccl.Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))sym("ncclRecv");
  g_rccl.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.Send ||
      !g_rccl.Recv || !g_rccl.GetErrorString) { g_rccl.h = nullptr; return ceed_error("librccl lacks a point-to-point entry point"); }
  return 0;
}
}  // namespace
#define RCCLCHK(x) do { int r_ = (x); if (r_ != 0) return ceed_error("%s failed: %s", #x, g_rccl.GetErrorString(r_)); } while (0)

extern "C" int CeedXCommGetUniqueId(Ceed, char id[128]) {
  CHK(rccl_load());
  rccl_unique_id u;
  RCCLCHK(g_rccl.GetUniqueId(&u));
  memcpy(id, u.internal, 128);
  return 0;
}
extern "C" int CeedXCommInit(Ceed ceed, int nranks, int rank, const char id[128]) {
  if (ceed->comm) return ceed_error("this Ceed already has a communicator");
  if (nranks < 1 || rank < 0 || rank >= nranks) return ceed_error("CeedXCommInit: rank %d of %d", rank, nranks);
  CHK(rccl_load());
  rccl_unique_id u;
  memcpy(u.internal, id, 128);
  HIPCHK(hipSetDevice(ceed->device));
  RCCLCHK(g_rccl.CommInitRank(&ceed->comm, nranks, u, rank));
  ceed->comm_rank = rank; ceed->comm_size = nranks;
  if (!ceed->comm_stream) HIPCHK(hipStreamCreateWithFlags(&ceed->comm_stream, hipStreamNonBlocking));
  return 0;
}
extern "C" int CeedXCommDestroy(Ceed ceed) {
  if (ceed->comm) { (void)hipStreamSynchronize(ceed->comm_stream); (void)g_rccl.CommDestroy(ceed->comm); ceed->comm = nullptr; }
  return 0;
}

struct HaloNeighbour { int rank = 0, n = 0; uint32_t *d_idx = nullptr; double *send = nullptr, *recv = nullptr; };
struct CeedXHalo_private {
  Ceed ceed = nullptr;
  std::vector<HaloNeighbour> nb;
  hipEvent_t packed = nullptr, arrived = nullptr;
  bool in_flight = false;
  CeedInt lsize_min = 0;
};
// Neighbour lists: `index[k]` holds the `count[k]` L-vector entries shared with rank `neigh_rank[k]`, in an order both
// sides agree on (halo.py sorts them by partition-independent node keys).  Entries are unique within one list.
extern "C" int CeedXHaloCreate(Ceed ceed, CeedInt nneigh, const int *neigh_rank, const CeedInt *count,
                               const CeedInt *const *index, CeedXHalo *halo) {
  if (nneigh > 0 && !ceed->comm) return ceed_error("CeedXHaloCreate: call CeedXCommInit first");
  CeedXHalo H = new CeedXHalo_private;
  H->ceed = ceed; ceed_ref(ceed);
  for (int k = 0; k < nneigh; k++) {
    if (neigh_rank[k] < 0 || neigh_rank[k] >= ceed->comm_size || count[k] < 0) return ceed_error("CeedXHaloCreate: bad neighbour %d", k);
    HaloNeighbour nb;
    nb.rank = neigh_rank[k]; nb.n = count[k];
    std::vector<uint32_t> idx((size_t)nb.n);
    for (int i = 0; i < nb.n; i++) {
      if (index[k][i] < 0) return ceed_error("CeedXHaloCreate: negative index");
      idx[i] = (uint32_t)index[k][i];
      H->lsize_min = std::max(H->lsize_min, index[k][i] + 1);
    }
    HIPCHK(hipMalloc((void **)&nb.d_idx, sizeof(uint32_t) * (idx.size() ? idx.size() : 1)));
    HIPCHK(hipMemcpy(nb.d_idx, idx.data(), sizeof(uint32_t) * idx.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&nb.send, sizeof(double) * (nb.n ? nb.n : 1)));
    HIPCHK(hipMalloc((void **)&nb.recv, sizeof(double) * (nb.n ? nb.n : 1)));
    H->nb.push_back(nb);
  }
  HIPCHK(hipEventCreateWithFlags(&H->packed, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&H->arrived, hipEventDisableTiming));
  *halo = H;
  return 0;
}
// Start: pack on the Ceed's stream, then all sends and receives of this rank as ONE RCCL group on the communicator's
// stream -- the Ceed's stream is free for the interior elements meanwhile (CeedXOperatorApplyPhase 1).
extern "C" int CeedXHaloStart(CeedXHalo H, CeedVector y) {
  if (H->in_flight) return ceed_error("CeedXHaloStart: an exchange is already in flight");
  if (H->nb.empty()) return 0;
  Ceed c = H->ceed;
  if (c->capturing) return ceed_error("CeedXHaloStart during graph capture");
  if (y->length < H->lsize_min) return ceed_error("CeedXHaloStart: vector shorter than the halo's indices");
  double *py;
  CHK(vec_dev(y, false, &py));
  for (HaloNeighbour &nb : H->nb) HIPCHK(launch_halo_pack(nb.d_idx, nb.n, py, nb.send, c->stream));
  HIPCHK(hipEventRecord(H->packed, c->stream));
  HIPCHK(hipStreamWaitEvent(c->comm_stream, H->packed, 0));
  RCCLCHK(g_rccl.GroupStart());
  for (HaloNeighbour &nb : H->nb) {
    RCCLCHK(g_rccl.Send(nb.send, (size_t)nb.n, RCCL_FLOAT64, nb.rank, c->comm, c->comm_stream));
    RCCLCHK(g_rccl.Recv(nb.recv, (size_t)nb.n, RCCL_FLOAT64, nb.rank, c->comm, c->comm_stream));
  }
  RCCLCHK(g_rccl.GroupEnd());
  HIPCHK(hipEventRecord(H->arrived, c->comm_stream));
  H->in_flight = true;
  return 0;
}
// Finish: the Ceed's stream waits for the arrivals and adds them, neighbour by neighbour in list order (a node shared by
// three ranks gets its two additions in the same order every time: the sum is reproducible).
extern "C" int CeedXHaloFinish(CeedXHalo H, CeedVector y) {
  if (H->nb.empty()) return 0;
  if (!H->in_flight) return ceed_error("CeedXHaloFinish without CeedXHaloStart");
  Ceed c = H->ceed;
  double *py;
  CHK(vec_dev(y, true, &py));
  HIPCHK(hipStreamWaitEvent(c->stream, H->arrived, 0));
  for (HaloNeighbour &nb : H->nb) HIPCHK(launch_halo_unpack_add(nb.d_idx, nb.n, nb.recv, py, c->stream));
  H->in_flight = false;
  return 0;
}
extern "C" int CeedXHaloDestroy(CeedXHalo *halo) {
  if (!halo || !*halo) return 0;
  CeedXHalo H = *halo;
  (void)hipStreamSynchronize(H->ceed->stream);
  if (H->ceed->comm_stream) (void)hipStreamSynchronize(H->ceed->comm_stream);
  for (HaloNeighbour &nb : H->nb) { (void)hipFree(nb.d_idx); (void)hipFree(nb.send); (void)hipFree(nb.recv); }
  if (H->packed) (void)hipEventDestroy(H->packed);
  if (H->arrived) (void)hipEventDestroy(H->arrived);
  ceed_unref(H->ceed);
  delete H;
  *halo = nullptr;
  return 0;
}

// ---------------------------------------------------------------------------
// Assembled sparse operator (coarse multigrid level; misc.c:151-183, elasticity.c:457-483)
// ---------------------------------------------------------------------------
struct CeedXCsr_private {
  Ceed ceed = nullptr;
  int nrows = 0, ncols = 0, nnz = 0, ncoo = 0, n_unit = 0;
  uint32_t *d_rowptr = nullptr, *d_cols = nullptr, *d_slotptr = nullptr, *d_perm = nullptr, *d_unit_slot = nullptr,
           *d_diag_slot = nullptr;
  double *d_vals = nullptr;
  // values as a fixed linear combination of another matrix's values (CeedXCsrSetSource / CeedXCsrUpdate)
  CeedXCsr src = nullptr;
  uint32_t *d_termptr = nullptr, *d_term_slot = nullptr;
  double *d_term_w = nullptr, *d_gj = nullptr;
  int *d_info = nullptr;
  bool dense = false;       // full pattern, columns ascending: vals is a row-major nrows x nrows matrix
  int refs = 1;             // a source is kept alive by the matrices combined from it
  std::vector<int> h_rowptr, h_cols;      // host copy of the pattern (operand of CeedXCsrCreateProduct)
  std::vector<double> h_vals;             // host copy of FIXED values (CeedXCsrCreateRect with values), else empty
};
template <class T>
static int csr_upload(uint32_t **dst, const std::vector<T> &v) {
  HIPCHK(hipMalloc((void **)dst, sizeof(uint32_t) * (v.size() ? v.size() : 1)));
  if (!v.empty()) HIPCHK(hipMemcpy(*dst, v.data(), sizeof(uint32_t) * v.size(), hipMemcpyHostToDevice));
  return 0;
}
extern "C" int CeedXCsrCreate(Ceed ceed, CeedInt nrows, const CeedInt *rowptr, const CeedInt *cols, CeedInt ncoo,
                              const CeedInt *coo_slot, CeedInt n_unit, const CeedInt *unit_rows, CeedXCsr *csr) {
  if (nrows < 0 || ncoo < 0 || !rowptr || (rowptr[nrows] > 0 && !cols)) return ceed_error("CeedXCsrCreate: bad pattern");
  const int nnz = rowptr[nrows];
  std::vector<uint32_t> rp(rowptr, rowptr + nrows + 1), cl(cols, cols + nnz), diag((size_t)nrows, 0xFFFFFFFFu);
  for (int r = 0; r < nrows; r++) {
    if (rowptr[r + 1] < rowptr[r]) return ceed_error("CeedXCsrCreate: rowptr not monotone");
    for (int k = rowptr[r]; k < rowptr[r + 1]; k++) {
      if (cols[k] < 0 || cols[k] >= nrows) return ceed_error("CeedXCsrCreate: column %d out of range in row %d", cols[k], r);
      if (cols[k] == r) diag[r] = (uint32_t)k;
    }
  }
  // transpose of coo_slot: for every CSR slot the COO entries it sums, ascending (counting sort keeps the order)
  std::vector<uint32_t> slotptr((size_t)nnz + 1, 0u), perm;
  size_t kept = 0;
  for (int k = 0; k < ncoo; k++) {
    if (coo_slot[k] >= nnz) return ceed_error("CeedXCsrCreate: COO entry %d maps to slot %d of %d", k, coo_slot[k], nnz);
    if (coo_slot[k] >= 0) { slotptr[(size_t)coo_slot[k] + 1]++; kept++; }
  }
  for (int s = 0; s < nnz; s++) slotptr[s + 1] += slotptr[s];
  perm.resize(kept ? kept : 1);
  {
    std::vector<uint32_t> cur(slotptr.begin(), slotptr.end() - 1);
    for (int k = 0; k < ncoo; k++) if (coo_slot[k] >= 0) perm[cur[coo_slot[k]]++] = (uint32_t)k;
  }
  std::vector<uint32_t> unit;
  for (int i = 0; i < n_unit; i++) {
    if (unit_rows[i] < 0 || unit_rows[i] >= nrows || diag[unit_rows[i]] == 0xFFFFFFFFu)
      return ceed_error("CeedXCsrCreate: unit row %d has no diagonal entry in the pattern", unit_rows[i]);
    unit.push_back(diag[unit_rows[i]]);
  }
  CeedXCsr A = new CeedXCsr_private;
  A->ceed = ceed; ceed_ref(ceed);
  A->nrows = nrows; A->ncols = nrows; A->nnz = nnz; A->ncoo = ncoo; A->n_unit = n_unit;
  A->h_rowptr.assign(rowptr, rowptr + nrows + 1); A->h_cols.assign(cols, cols + nnz);
  CHK(csr_upload(&A->d_rowptr, rp)); CHK(csr_upload(&A->d_cols, cl)); CHK(csr_upload(&A->d_slotptr, slotptr));
  CHK(csr_upload(&A->d_perm, perm)); CHK(csr_upload(&A->d_unit_slot, unit)); CHK(csr_upload(&A->d_diag_slot, diag));
  HIPCHK(hipMalloc((void **)&A->d_vals, sizeof(double) * (nnz ? nnz : 1)));
  HIPCHK(hipMemset(A->d_vals, 0, sizeof(double) * (nnz ? nnz : 1)));
  *csr = A;
  return 0;
}
extern "C" int CeedXCsrAssemble(CeedXCsr A, CeedVector coo_values) {
  if (coo_values->length < A->ncoo) return ceed_error("CeedXCsrAssemble: %d COO values, %d expected", coo_values->length, A->ncoo);
  double *pc;
  CHK(vec_dev(coo_values, false, &pc));
  HIPCHK(launch_csr_sum(A->d_slotptr, A->d_perm, pc, A->d_vals, A->nnz, A->d_unit_slot, A->n_unit, A->ceed->stream));
  return 0;
}
extern "C" int CeedXCsrApply(CeedXCsr A, CeedVector x, CeedVector y) {
  if (x == y) return ceed_error("CeedXCsrApply: in-place apply is not supported");
  if (x->length < A->ncols || y->length < A->nrows) return ceed_error("CeedXCsrApply: vector shorter than the matrix");
  double *px, *py;
  CHK(vec_dev(x, false, &px)); CHK(vec_dev(y, true, &py));
  HIPCHK(launch_csr_spmv(A->d_rowptr, A->d_cols, A->d_vals, px, py, A->nrows, A->ceed->stream));
  return 0;
}
extern "C" int CeedXCsrGetDiagonal(CeedXCsr A, CeedVector d) {
  if (d->length < A->nrows) return ceed_error("CeedXCsrGetDiagonal: vector shorter than the matrix");
  double *pd;
  CHK(vec_dev(d, true, &pd));
  HIPCHK(launch_csr_diag(A->d_diag_slot, A->d_vals, pd, A->nrows, A->ceed->stream));
  return 0;
}
// Rectangular matrix with fixed values (prolongation / restriction of the aggregation hierarchy), or a pattern whose
// values come from CeedXCsrUpdate.
extern "C" int CeedXCsrCreateRect(Ceed ceed, CeedInt nrows, CeedInt ncols, const CeedInt *rowptr, const CeedInt *cols,
                                  const CeedScalar *vals, CeedXCsr *csr) {
  if (nrows < 0 || ncols < 0 || !rowptr || (rowptr[nrows] > 0 && !cols)) return ceed_error("CeedXCsrCreateRect: bad pattern");
  const int nnz = rowptr[nrows];
  bool dense = nrows == ncols && (long long)nnz == (long long)nrows * nrows;
  for (int r = 0; r < nrows; r++) {
    if (rowptr[r + 1] < rowptr[r]) return ceed_error("CeedXCsrCreateRect: rowptr not monotone");
    for (int k = rowptr[r]; k < rowptr[r + 1]; k++) {
      if (cols[k] < 0 || cols[k] >= ncols) return ceed_error("CeedXCsrCreateRect: column %d out of range in row %d", cols[k], r);
      if (dense && cols[k] != k - rowptr[r]) dense = false;
    }
  }
  std::vector<uint32_t> rp(rowptr, rowptr + nrows + 1), cl(cols, cols + nnz), diag((size_t)nrows, 0xFFFFFFFFu);
  for (int r = 0; r < nrows; r++)
    for (int k = rowptr[r]; k < rowptr[r + 1]; k++) if (cols[k] == r) diag[r] = (uint32_t)k;
  CeedXCsr A = new CeedXCsr_private;
  A->ceed = ceed; ceed_ref(ceed);
  A->nrows = nrows; A->ncols = ncols; A->nnz = nnz; A->dense = dense;
  A->h_rowptr.assign(rowptr, rowptr + nrows + 1); A->h_cols.assign(cols, cols + nnz);
  if (vals) A->h_vals.assign(vals, vals + nnz);
  CHK(csr_upload(&A->d_rowptr, rp)); CHK(csr_upload(&A->d_cols, cl)); CHK(csr_upload(&A->d_diag_slot, diag));
  HIPCHK(hipMalloc((void **)&A->d_vals, sizeof(double) * (nnz ? nnz : 1)));
  if (vals && nnz) HIPCHK(hipMemcpy(A->d_vals, vals, sizeof(double) * nnz, hipMemcpyHostToDevice));
  else HIPCHK(hipMemset(A->d_vals, 0, sizeof(double) * (nnz ? nnz : 1)));
  *csr = A;
  return 0;
}
// C = L R where one operand has FIXED values (given to CeedXCsrCreateRect) and the other is `variable`: its values are
// read by every CeedXCsrUpdate(C).  The pattern of C and, per entry, the list of (slot of the variable operand, weight)
// terms are worked out here once, row by row (Gustavson), terms of an entry ordered by slot: vals[s] = sum_k w[k] * V[slot[k]].
// dense != 0: C gets the full pattern (entries without a term stay zero), for CeedXCsrInvertDenseSPD.
extern "C" int CeedXCsrCreateProduct(CeedXCsr Lm, CeedXCsr Rm, int variable, int dense, CeedXCsr *csr) {
  if (!Lm || !Rm || Lm == Rm || (variable != 0 && variable != 1)) return ceed_error("CeedXCsrCreateProduct: bad operands");
  if (Lm->ncols != Rm->nrows) return ceed_error("CeedXCsrCreateProduct: %d columns times %d rows", Lm->ncols, Rm->nrows);
  CeedXCsr V = variable == 0 ? Lm : Rm, F = variable == 0 ? Rm : Lm;
  if ((int)F->h_vals.size() != F->nnz) return ceed_error("CeedXCsrCreateProduct: the fixed operand must carry values from CeedXCsrCreateRect");
  const int nrows = Lm->nrows, ncols = Rm->ncols;
  if (dense && nrows != ncols) return ceed_error("CeedXCsrCreateProduct: a dense result must be square");
  struct Term { int col, slot; double w; };
  // row blocks in parallel on the host (the lists of a 150 000-row level are ~3e8 terms: 17 s on one thread), stitched in row order
  struct Part { std::vector<uint32_t> len, cl, tcount, ts; std::vector<double> tw; };
  const int nthreads = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, nrows / 256 + 1}));
  std::vector<Part> parts((size_t)nthreads);
  auto work = [&](int t) {
    Part &pt = parts[(size_t)t];
    const int r0 = (int)((long long)nrows * t / nthreads), r1 = (int)((long long)nrows * (t + 1) / nthreads);
    std::vector<Term> row;
    for (int i = r0; i < r1; i++) {
      row.clear();
      for (int a = Lm->h_rowptr[i]; a < Lm->h_rowptr[i + 1]; a++) {
        const int j = Lm->h_cols[a];
        for (int b = Rm->h_rowptr[j]; b < Rm->h_rowptr[j + 1]; b++)
          row.push_back(variable == 0 ? Term{Rm->h_cols[b], a, F->h_vals[b]} : Term{Rm->h_cols[b], b, F->h_vals[a]});
      }
      std::sort(row.begin(), row.end(), [](const Term &x, const Term &y) { return x.col != y.col ? x.col < y.col : x.slot < y.slot; });
      size_t k = 0;
      uint32_t n_in_row = 0;
      for (int c = 0; dense ? c < ncols : k < row.size(); c++) {
        if (!dense) c = row[k].col;
        uint32_t cnt = 0;
        while (k < row.size() && row[k].col == c) { pt.ts.push_back((uint32_t)row[k].slot); pt.tw.push_back(row[k].w); k++; cnt++; }
        pt.cl.push_back((uint32_t)c); pt.tcount.push_back(cnt); n_in_row++;
      }
      pt.len.push_back(n_in_row);
    }
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
  }
  size_t tot_e = 0, tot_t = 0;
  for (const Part &pt : parts) { tot_e += pt.cl.size(); tot_t += pt.ts.size(); }
  if (tot_t >= 0x7FFFFFFFull || tot_e >= 0x7FFFFFFFull) return ceed_error("CeedXCsrCreateProduct: more than 2^31 product terms");
  std::vector<uint32_t> rp((size_t)nrows + 1, 0u), cl, tp(1, 0u), ts;
  std::vector<double> tw;
  std::vector<int> h_cols;
  cl.reserve(tot_e); h_cols.reserve(tot_e); tp.reserve(tot_e + 1); ts.reserve(tot_t); tw.reserve(tot_t);
  {
    int i = 0;
    for (Part &pt : parts) {
      for (uint32_t L : pt.len) { rp[(size_t)i + 1] = rp[(size_t)i] + L; i++; }
      for (size_t e = 0; e < pt.cl.size(); e++) { cl.push_back(pt.cl[e]); h_cols.push_back((int)pt.cl[e]); tp.push_back(tp.back() + pt.tcount[e]); }
      ts.insert(ts.end(), pt.ts.begin(), pt.ts.end());
      tw.insert(tw.end(), pt.tw.begin(), pt.tw.end());
      Part().len.swap(pt.len); std::vector<uint32_t>().swap(pt.ts); std::vector<double>().swap(pt.tw);
      std::vector<uint32_t>().swap(pt.cl); std::vector<uint32_t>().swap(pt.tcount);
    }
  }
  const int nnz = (int)cl.size();
  std::vector<uint32_t> diag((size_t)nrows, 0xFFFFFFFFu);
  for (int r = 0; r < nrows; r++)
    for (uint32_t k = rp[r]; k < rp[r + 1]; k++) if ((int)cl[k] == r) diag[r] = k;
  CeedXCsr A = new CeedXCsr_private;
  A->ceed = V->ceed; ceed_ref(A->ceed);
  A->nrows = nrows; A->ncols = ncols; A->nnz = nnz; A->dense = dense != 0;
  A->h_rowptr.assign(rp.begin(), rp.end()); A->h_cols.swap(h_cols);
  CHK(csr_upload(&A->d_rowptr, rp)); CHK(csr_upload(&A->d_cols, cl)); CHK(csr_upload(&A->d_diag_slot, diag));
  CHK(csr_upload(&A->d_termptr, tp)); CHK(csr_upload(&A->d_term_slot, ts));
  HIPCHK(hipMalloc((void **)&A->d_term_w, sizeof(double) * (tw.size() ? tw.size() : 1)));
  if (!tw.empty()) HIPCHK(hipMemcpy(A->d_term_w, tw.data(), sizeof(double) * tw.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc((void **)&A->d_vals, sizeof(double) * (nnz ? nnz : 1)));
  HIPCHK(hipMemset(A->d_vals, 0, sizeof(double) * (nnz ? nnz : 1)));
  A->src = V; V->refs++;
  *csr = A;
  return 0;
}
extern "C" int CeedXCsrGetPattern(CeedXCsr A, CeedInt *nrows, CeedInt *ncols, CeedInt *nnz, const CeedInt **rowptr, const CeedInt **cols) {
  if (nrows) *nrows = A->nrows;
  if (ncols) *ncols = A->ncols;
  if (nnz) *nnz = A->nnz;
  if (rowptr) *rowptr = A->h_rowptr.data();
  if (cols) *cols = A->h_cols.data();
  return 0;
}
extern "C" int CeedXCsrUpdate(CeedXCsr A) {
  if (!A->src) return ceed_error("CeedXCsrUpdate: not a product (CeedXCsrCreateProduct)");
  HIPCHK(launch_csr_combine(A->d_termptr, A->d_term_slot, A->d_term_w, A->src->d_vals, A->d_vals, A->nnz, A->ceed->stream));
  return 0;
}
extern "C" int CeedXCsrGetValues(CeedXCsr A, CeedVector v) {
  if (v->length < A->nnz) return ceed_error("CeedXCsrGetValues: vector of %d for %d entries", v->length, A->nnz);
  double *pv;
  CHK(vec_dev(v, true, &pv));
  if (A->nnz) HIPCHK(hipMemcpyAsync(pv, A->d_vals, sizeof(double) * A->nnz, hipMemcpyDeviceToDevice, A->ceed->stream));
  return 0;
}
// In-place inverse of a matrix with a FULL pattern (every row holds columns 0..n-1 in order) and symmetric positive
// definite values: the coarsest level of the aggregation hierarchy, applied afterwards with CeedXCsrApply.
extern "C" int CeedXCsrInvertDenseSPD(CeedXCsr A) {
  if (!A->dense) return ceed_error("CeedXCsrInvertDenseSPD: the pattern is not a full square one with ascending columns");
  if (A->ceed->capturing) return ceed_error("CeedXCsrInvertDenseSPD cannot be recorded into a graph (it reports a status to the host)");
  if (!A->d_gj) {
    HIPCHK(hipMalloc((void **)&A->d_gj, sizeof(double) * 32 * 32));
    HIPCHK(hipMalloc((void **)&A->d_info, sizeof(int)));
  }
  HIPCHK(hipMemsetAsync(A->d_info, 0, sizeof(int), A->ceed->stream));
  HIPCHK(launch_dense_spd_inverse(A->d_vals, A->nrows, A->d_gj, A->d_info, A->ceed->stream));
  int info = 0;
  HIPCHK(hipMemcpyAsync(&info, A->d_info, sizeof(int), hipMemcpyDeviceToHost, A->ceed->stream));
  HIPCHK(hipStreamSynchronize(A->ceed->stream));
  if (info) return ceed_error("CeedXCsrInvertDenseSPD: pivot %d is not positive: the matrix is not positive definite", info - 1);
  return 0;
}
extern "C" int CeedXCsrDestroy(CeedXCsr *csr) {
  if (!csr || !*csr) return 0;
  CeedXCsr A = *csr;
  *csr = nullptr;
  if (--A->refs > 0) return 0;        // still the source of another matrix: freed with the last of those
  (void)hipStreamSynchronize(A->ceed->stream);
  for (uint32_t *p : {A->d_rowptr, A->d_cols, A->d_slotptr, A->d_perm, A->d_unit_slot, A->d_diag_slot, A->d_termptr, A->d_term_slot})
    if (p) (void)hipFree(p);
  for (double *p : {A->d_vals, A->d_term_w, A->d_gj}) if (p) (void)hipFree(p);
  if (A->d_info) (void)hipFree(A->d_info);
  CeedXCsr src = A->src;
  ceed_unref(A->ceed);
  delete A;
  if (src) (void)CeedXCsrDestroy(&src);
  return 0;
}
