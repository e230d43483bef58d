// ceed_impl.hpp -- private object layouts and helpers shared by the host-side sources of the MI355X backend
// (ceed_core.cpp: Ceed, vectors, graphs; ceed_basis.cpp; ceed_restriction.cpp; ceed_operator.cpp; ceed_halo.cpp;
// ceed_csr.cpp).  Nothing here is part of the ABI: include/ceed.h is.
#pragma once
#include <ceed.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"

// ---------------------------------------------------------------------------
// errors: every entry point returns 0 or ceed_error(...).  The reference ignores return codes, so
// errors abort by default; CeedXSetErrorReturn(1) makes them return (tests).
// ---------------------------------------------------------------------------
int ceed_error(const char *fmt, ...);
#define CHK(x) do { int ierr_ = (x); if (ierr_) return ierr_; } while (0)
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
  return ceed_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// ---------------------------------------------------------------------------
// Run-time options, read ONCE from the environment by CeedInit (never on an apply path).  A/B switches of shipped
// features only; the experiments of rounds 1-2 (row kernel, atomics, dynamic / gated / folded / pair forms) are gone.
// ---------------------------------------------------------------------------
struct CeedOptions {
  bool recompute_geo = true;     // CEED_MI355X_GEO=0: the fused kernels read qdata instead of recomputing it from the element maps
  bool direct_interior = true;   // CEED_MI355X_DIRECT=0: element-interior nodes go through the E-vector like the shared ones
  bool derived_state = true;     // CEED_MI355X_DERIVED=0: HyperFSdF forms F^-1 and ln J from the stored grad u at every point
  bool affine_geo = true;        // CEED_MI355X_AFFINE=0: all-affine meshes take the general per-point recompute too
  bool swept_geo = true;         // CEED_MI355X_SWEPT=0: meshes of swept (extruded) elements take the general per-point recompute too
  // restriction transpose of large whole applies: pipelined in segments over two streams
  int pipe_segments = -1;        // 0: never (CEED_MI355X_ASSEMBLE=serial); -1: chosen per launch; >= 2: CEED_MI355X_PIPE_SEGMENTS
  int pipe_mb = 0;               // CEED_MI355X_PIPE_MB: one segment per this many MB of E-vector when the count is chosen per launch
                                 // (0: 160 for the finite-strain kernels, 90 for the cheaper ones -- see get_pipe)
  int pipe_blocks = 0;           // CEED_MI355X_PIPE_BLOCKS: cap on the workgroups of a k_assemble that runs beside a fused kernel
  int pipe_last_rounds = 4;      // CEED_MI355X_PIPE_LAST: rounds of the persistent waves in the LAST segment
  int pipe_min_total_rounds = 20;   // CEED_MI355X_PIPE_MIN_TOTAL: rounds a whole apply must have to be pipelined
  int pipe_min_rounds = 4;       // CEED_MI355X_PIPE_MIN_ROUNDS: rounds a segment must have (0: tests on small meshes)
  bool pipe_debug = false;       // CEED_MI355X_PIPE_DEBUG
  bool graph_memset = false;     // CEED_MI355X_GRAPH_MEMSET=1: recorded zero-fills as memset nodes instead of fill kernels
  int pencil_waves = 0;          // CEED_MI355X_PENCIL_WAVES: persistent waves per CU of the fused kernel (tuning hook)
  // split-phase apply with the halo exchange (CeedXOperatorApplyWithHalo)
  int ovl_mode = 0;              // CEED_MI355X_OVL_MODE: 0 (default) the whole apply, then the exchange, in order on one stream;
                                 // 1 split-phase on one stream (interface elements, exchange started, interior elements: round 2's
                                 // sequence); 2 split-phase on two streams (both phases' fused kernels side by side)
  int ovl_groups0 = 1, ovl_groups1 = 0;   // CEED_MI355X_OVL_G0 / _G1: groups per wave of the two phases (0: persistent grid)
  bool epi_pipelined = false;    // CEED_MI355X_EPI_PIPELINED=1: the apply fused with its consumer in the pipelined form too (default: serial, measured faster)
  bool spmv_stream = true;       // CEED_MI355X_SPMV=vector: CeedXCsrApply a wave per row (rounds 2-4's kernel, A/B) instead of the CSR-stream form
  bool spgemm_row = true;        // CEED_MI355X_SPGEMM=entry: Galerkin products an entry of C per lane with binary searches in global memory (round 3's kernel, A/B)
  int fold_pack = 1;             // CEED_MI355X_FOLD_PACK=0: the exchange's pack as a launch of its own (A/B)
  int comm_inline = 1;           // CEED_MI355X_COMM_INLINE=0: the exchange's sends / receives on a stream of their own (see halo_pack_and_send)
  int comm_priority = 0;         // CEED_MI355X_COMM_PRIO=1: the exchange on a highest-priority stream -- measured 4x SLOWER (see CeedXCommInit)
};

// ---------------------------------------------------------------------------
// object layouts
// ---------------------------------------------------------------------------
// What a recorded operator apply read BESIDE its vectors' arrays: the provenance buffers kept with a qdata vector (geometry
// coefficients) or a stored-state vector (derived state of the tangent).  Any other write to such a vector drops its
// provenance, and an eager apply then reads the array itself; a recorded apply would go on reading the old buffer -- so a graph
// remembers what it depends on and CeedXGraphLaunch refuses to replay once it no longer holds.
struct GraphDep { CeedVector v; const double *geo; const double *derived; };
struct Ceed_private {
  int refcount = 1;
  std::string resource;
  hipStream_t stream = nullptr;
  int device = 0;
  CeedOptions opt;
  // scratch E-vector shared by the operators of this Ceed (applies are serialised on `stream`)
  double *evec = nullptr;
  size_t evec_len = 0;
  // A recorded hipGraph has the scratch pointer of its capture time baked into its kernel nodes.  When the scratch has
  // to grow while a graph of this Ceed is alive (or is being recorded), the old buffer is PARKED, not freed: replays of
  // the older graphs keep a valid scratch of the size they were recorded with.  Parked buffers go when the last graph
  // goes.  The same holds for re-ordered transpose maps and flag arrays that are replaced (parked_misc).
  std::vector<double *> evec_parked;
  std::vector<void *> parked_misc;
  int live_graphs = 0;
  hipStream_t side_stream = nullptr;          // second chain of a pipelined apply; phase 1 of a split-phase apply
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_seg[16] = {nullptr};
  // RCCL communicator of the halo exchange (CeedXCommInit) and the stream its sends / receives run on
  void *comm = nullptr;
  int comm_rank = 0, comm_size = 1;
  hipStream_t comm_stream = nullptr;
  double *d_scalar = nullptr;   // device scalar for reductions (1 + 2048 doubles)
  double *h_scalar = nullptr;   // pinned host landing slot for it (pageable targets make the runtime stage + pin per copy)
  // hipGraph capture (CeedXGraphBeginCapture): device work is recorded on `capture_stream`
  hipStream_t capture_stream = nullptr, saved_stream = nullptr;
  bool capturing = false;
  std::vector<GraphDep> capture_deps;   // collected while recording (fused_prepare), handed to the graph at EndCapture
};
struct CeedXGraph_private {
  std::vector<GraphDep> deps;
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
  uint64_t version = 0;                // bumped by every write access through the API (caches keyed by a vector's contents check it)
  bool h_owned = false, d_owned = false;
  bool h_valid = false, d_valid = false;
  // provenance of a qdata vector: written by the SetupGeo operator from trilinear elements whose map coefficients
  // are kept here ([nelem][GEO_NCOEF], device).  The fused kernels then recompute the geometric factors instead of
  // reading them (FusedGradArgs::geo).  Dropped by any other write to the vector.
  double *geo = nullptr;
  double *geo_aff = nullptr;   // set when EVERY element is affine: [nelem][GEO_NAFF] constant factors (FusedGradArgs::geo_aff)
  double *geo_swept = nullptr; // set when every element is swept along the reference direction geo_axis (FusedGradArgs::geo_swept)
  int geo_axis = 0;
  int geo_nelem = 0, geo_Q = 0;
  double geo_qref[cps::MAXN1D] = {0}, geo_qwt[cps::MAXN1D] = {0};
  // provenance of a stored-state vector (grad u): written by this backend's HyperFSF kernel, which left the DERIVED state of
  // the tangent ([nelem][10][Q^3]: F^-1, lambda ln J - mu) beside it; HyperFSdF then reads that instead (QF_HYPERFS_DF_DS).
  // Any other write to the vector invalidates it (the buffer is kept for the next residual evaluation).
  double *derived = nullptr;
  size_t derived_len = 0;
  bool derived_valid = false;
  int derived_nelem = 0, derived_Q3 = 0;
  double derived_nu = 0., derived_E = 0.;   // the material the derived state was formed with
};

// Transpose map of an offsets restriction: distinct node offsets and, per node, the E-vector
// positions (e*elemsize + n) of its contributors in element order.  Rows [0, nprio) are the
// "priority" nodes when the map was built with a priority mask (split-phase apply).
struct CsrMap {
  bool built = false, full_cover = false;
  int nnodes = 0, nprio = 0, nskipped = 0;
  std::vector<uint32_t> h_node_off, h_rowptr, h_cols;   // host copies (the re-ordered maps are derived from them)
  uint32_t *d_rowptr = nullptr, *d_cols = nullptr, *d_node_off = nullptr;
  void release() {
    if (d_rowptr) (void)hipFree(d_rowptr);
    if (d_cols) (void)hipFree(d_cols);
    if (d_node_off) (void)hipFree(d_node_off);
    d_rowptr = d_cols = d_node_off = nullptr; built = false;
  }
};

// The transpose map re-ordered for the PIPELINED assembly: the apply is cut into segments of consecutive elements, one
// launch of the fused kernel each; a row (node) belongs to the segment of its LAST contributor, rows are sorted by segment,
// and the rows of segment k are summed by their own k_assemble launch beside the fused kernel of segment k + 1.
struct PipeMap {
  bool built = false;
  int nseg = 0, req_seg = 0, E = 0, waves = 0, nrows = 0, mb = 0;
  const void *base = nullptr;              // the CsrMap it was derived from
  std::vector<int> elem_bound, row_bound;  // nseg + 1 each
  std::vector<uint32_t> h_node_off;        // re-ordered (for the per-operator Dirichlet flags)
  uint32_t *d_rowptr = nullptr, *d_cols = nullptr, *d_node_off = nullptr;
};

struct CeedElemRestriction_private {
  Ceed ceed = nullptr;
  int refcount = 1;
  CeedInt nelem = 0, elemsize = 0, ncomp = 0, compstride = 0, lsize = 0;
  bool strided = false, backend_strides = true;
  CeedInt strides[3] = {0, 0, 0};
  std::vector<CeedInt> h_offsets;
  uint32_t *d_offsets = nullptr;  // plain (unflagged)
  CsrMap csr;              // default map (nodes in ascending offset order), built on first use
  CsrMap csr_shell;        // the same without the element-interior nodes (FusedGradArgs::direct)
  // pipelined maps, one per (base map, group size, persistent waves, segments asked for): never replaced once built, so
  // recorded graphs and alternating operators (different Q on one restriction) keep valid pointers
  std::vector<PipeMap *> pipes;
  int interior_private = 0;  // 0: not checked yet; 1: every element-interior node has one contributor; -1: not so
  uint32_t *d_int_off = nullptr;   // node offsets of the element-interior nodes, elements in order, int_per_elem each (direct-store mode:
  int int_per_elem = 0;            // the epilogue kernels read those nodes' values from y -- build_interior_list)
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
  int kind = cps::QF_NONE;
  void *ctx = nullptr;
  size_t ctxsize = 0;
  CeedInt identity_size = 0;
  std::vector<QFField> in, out;
};

struct OpField { bool set = false; CeedElemRestriction rstr = nullptr; CeedBasis basis = nullptr; CeedVector vec = nullptr; };

enum PlanKind { PLAN_NONE = 0, PLAN_FUSED_GRAD, PLAN_SETUP_GEO, PLAN_PROLONG, PLAN_RESTRICT, PLAN_COORD, PLAN_ENERGY };

struct CeedXHalo_private;

struct CeedOperator_private {
  Ceed ceed = nullptr;
  int refcount = 1;
  CeedQFunction qf = nullptr;
  std::vector<OpField> in, out;
  bool composite = false;
  std::vector<CeedOperator> sub;
  // lowering (op_plan)
  int plan = PLAN_NONE;
  int i_active = -1, i_qdata = -1, i_state = -1, i_weight = -1, o_active = -1, o_state = -1, o_qdata = -1;
  cps::BasisTables tables;
  double eo[6][cps::EO_TAB];          // even-odd forms of the six 1-D products (fused operators with pencil_even_odd(Q))
  std::string kernel_name;
  int geo_mode = 0;                   // last fused launch: 0 qdata read, 1 recomputed per point, 2 affine elements
  // Dirichlet flags
  uint32_t *d_off_flagged_in = nullptr, *d_off_flagged_out = nullptr;  // same array unless transfer
  unsigned char *d_node_flags = nullptr;        // per node of the restriction's transpose map
  unsigned char *d_node_flags_ovl = nullptr;    // per node of the operator's own (priority-first) map
  unsigned char *d_node_flags_shell = nullptr;  // per node of the restriction's shell map (direct-store mode)
  std::vector<std::pair<const PipeMap *, unsigned char *>> pipe_flags;   // per row of a pipelined map of the restriction
  // pack of a halo exchange folded into the rows' launch: per (transpose map, halo) the rows' send slots (HaloPackFold)
  struct PackFold { const CsrMap *M; CeedXHalo H; long serial; uint32_t *d_ptr, *d_slot; bool ok; };
  std::vector<PackFold> pack_folds;
  std::vector<unsigned char> h_mask;      // copy of the output mask (node flags are derived lazily)
  int mask_mode = 0;
  // optional fine-side scale for transfers
  CeedVector scale = nullptr;
  // transfer operators in OWNER form (ceed_operator.cpp: transfer_owner_map, transfer_weights)
  std::vector<unsigned char> h_mask_fine;  // the fine side's Dirichlet mask (empty: none)
  uint32_t *d_own_f = nullptr;             // [nelem][Pf^3] offset | flags of the fine nodes each element owns
  bool own_full_cover = false;             // every entry of the fine L-vector has an owner
  double *d_w = nullptr;                   // scale x local multiplicity per fine dof
  size_t w_len = 0;
  bool w_ready = false, w_unit = false;    // w_unit: every covered weight is 1 -- the kernels read none
  CeedVector w_scale = nullptr;            // the scale vector and its version the weights were formed from
  uint64_t w_version = 0;
  // split-phase apply (communication overlap): the first `ovl_lead` elements are the only
  // contributors of the priority nodes, which come first in the operator's own transpose map
  int ovl_lead = 0;
  CsrMap ovl_csr;
  long ovl_halo_checked = 0;     // serial of the halo whose entries were last checked to lie on priority rows of ovl_csr
  // timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  double ms_accum = 0.;
  int64_t launches = 0;
  int launch_info[4] = {0, 0, 0, 0};   // CeedXOperatorGetLaunchInfo
};

// ---------------------------------------------------------------------------
// helpers shared across the sources
// ---------------------------------------------------------------------------
void ceed_ref(Ceed c);
void ceed_unref(Ceed c);
// park a device allocation that recorded graph nodes may still read; freed with the last graph (or at once if none)
void ceed_retire(Ceed c, void *p);
int ceed_need_evec(Ceed c, size_t len);
int ceed_need_side_stream(Ceed c);
int dev_zero(Ceed c, double *p, size_t n);
// device pointer of a vector for kernels (synchronised from the host mirror if needed); write=true invalidates the
// host mirror and drops the qdata provenance
int vec_dev(CeedVector v, bool write, double **p);
void vec_drop_geo(CeedVector v);

// restriction maps (ceed_restriction.cpp)
int build_csr(CeedElemRestriction r, CsrMap &M, const unsigned char *prio, int skipP = 0);
bool rstr_interior_private(CeedElemRestriction r, int P);
int build_interior_list(CeedElemRestriction r, int P);
int get_pipe(CeedElemRestriction r, const CsrMap &M, int E, int per_elem, int req_seg, int waves, int mb_per_segment, PipeMap **out);

// halo internals the operator apply needs (ceed_halo.cpp)
struct HaloNeighbour { int rank = 0, n = 0, offset = 0; };
struct CeedXHalo_private {
  Ceed ceed = nullptr;
  long serial = 0;                   // unique per halo ever created (caches keyed by a halo check it, not just the address)
  std::vector<HaloNeighbour> nb;     // slices [offset, offset + n) of the send / receive buffers
  int total = 0;                     // entries over all neighbours
  uint32_t *d_idx = nullptr;         // [total] L-vector entry of every slot (pack)
  std::vector<uint32_t> h_idx;       // host copy (the operators fold the pack into their rows' launch from it)
  double *send = nullptr, *recv = nullptr;
  // arrivals by destination: distinct entries, and per entry its slots in neighbour-list order (unpack-add)
  int ndst = 0;
  uint32_t *d_dst = nullptr, *d_uptr = nullptr, *d_uslot = nullptr;
  hipEvent_t packed = nullptr, arrived = nullptr;
  hipStream_t arrived_on = nullptr;  // the stream the last exchange's receives were issued on (its `arrived` event too)
  bool in_flight = false;
  CeedInt lsize_min = 0;
};
int halo_pack_and_send(CeedXHalo H, const double *py, hipStream_t pack_stream);   // pack, then the RCCL group (same stream, or the comm stream)
int halo_send(CeedXHalo H, hipStream_t pack_stream);                              // the RCCL group alone (send buffer already filled)
int halo_wait_arrivals(CeedXHalo H, hipStream_t s);                               // make `s` wait for the last exchange's receives
cps::HaloUnpackArgs halo_unpack_args(CeedXHalo H);

static inline bool is_offsets(CeedElemRestriction r) { return r && r != CEED_ELEMRESTRICTION_NONE && !r->strided; }
static inline bool is_strided(CeedElemRestriction r) { return r && r != CEED_ELEMRESTRICTION_NONE && r->strided; }
