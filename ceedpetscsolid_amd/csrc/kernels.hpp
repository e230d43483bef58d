// kernels.hpp -- host-visible launch interface of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace cps {

enum QFKind : int {
  QF_NONE = 0,
  QF_SETUP_GEO,
  QF_LINELAS,     // LinElasF and LinElasdF: same linear map
  QF_HYPERSS_F,
  QF_HYPERSS_DF,
  QF_HYPERFS_F,
  QF_HYPERFS_DF,
  QF_IDENTITY,
  QF_CONST_FORCE,
  QF_MMS_FORCE,
  QF_MMS_TRUE,
  QF_ENERGY_LINELAS,
  QF_ENERGY_HYPERSS,
  QF_ENERGY_HYPERFS,
  QF_DIAG_LINELAS,
  QF_DIAG_HYPERSS,
  QF_DIAG_HYPERFS,
  QF_HYPERFS_DF_DS,   // HyperFSdF reading the DERIVED state (F^-1, lambda ln J - mu) the residual kernel wrote beside grad u
};

constexpr int MAXN1D = 8;  // largest P or Q supported by the kernel tables
constexpr int EO_TAB = 36;  // doubles per even-odd table (see FusedGradArgs::eo)

// 1-D tables handed to kernels by value: they live in the kernarg segment; the pencil kernel reads them from
// there as scalar operands, the other kernels stage them into LDS once per block.
struct BasisTables {
  double interp[MAXN1D * MAXN1D];  // B[q][p], Q x P row-major (CeedBasis interp1d)
  double colo[MAXN1D * MAXN1D];    // Dq[q][m], Q x Q: derivative of the Lagrange basis on the
                                   // quadrature points, evaluated there (collocated gradient)
  double grad[MAXN1D * MAXN1D];    // G[q][p], Q x P (CeedBasis grad1d; diagonal assembly)
  double qw[MAXN1D];               // 1-D quadrature weights
};

// Offsets carry the Dirichlet flags of the three components of a node in their
// top bits (set by CeedXOperatorSetDirichletMask); plain offsets have none.
constexpr uint32_t OFF_MASK = 0x1FFFFFFFu;
constexpr int OFF_FLAG_SHIFT = 29;

// Even-odd form of the 1-D products (FusedGradArgs::eo): used wherever the form exists -- the bases of this ABI are built
// on Gauss / Gauss-Lobatto points (CeedBasisCreateTensorH1Lagrange), whose tables are centro-(anti)symmetric.  Below 4 x 4
// the additions cost what the halved products save (measured -1.6 % at Q = 3); at Q = 8 an even-odd table (36
// coefficients) no longer fits the 60 SGPRs a pass has for its table.
constexpr bool pencil_even_odd(int Q) { return Q >= 4 && Q <= 7; }
// The derived state of the finite-strain tangent (QF_HYPERFS_DF_DS) is used -- and written by the residual kernel -- from Q = 6 on:
// measured -2.6 ... -3.1 % there, +-0 % at Q = 5 (profiles/r03_ab_experiments.txt item 3), for ten more doubles per point stored.
#ifndef CPS_DERIVED_MIN_Q
#define CPS_DERIVED_MIN_Q 6   // (tuning hook; the whole library must be built with the same value)
#endif
constexpr bool pencil_derived_state(int Q) { return Q >= CPS_DERIVED_MIN_Q; }

struct FusedGradArgs {
  const uint32_t *offsets;  // [nelem][P^3] (flagged)
  const double *x;          // active input L-vector, interlaced [node][3]
  double *y;                // active output L-vector: element-interior nodes are stored here directly (direct)
  const double *qdata;      // [nelem][10][Q^3]
  const double *state_in;   // [nelem][9][Q^3] (QF_HYPERFS_DF_DS: [nelem][10][Q^3], the derived state) or null
  double *state_out;        // [nelem][9][Q^3] or null
  double *state_out2;       // HyperFSF only, may be null: [nelem][10][Q^3] derived state of the tangent (qf_hyperfs_df_ds), written too
  int nelem;                // elements processed by this launch ...
  int elem_begin;           // ... starting at this element (segments of a pipelined apply, split-phase apply)
  int mask_in, mask_out;    // honour the Dirichlet flags on gather / scatter
  double nu, E, lambda, TwoMu;
  double *evec;               // element results ([elem][shell node or node][3], plain coalesced stores); launch_assemble()
                               // sums them into y per node in element order
  int evec_stride;            // doubles between the E-vector blocks of consecutive elements: 3 * nodes per block
  const double *geo;          // if set, [nelem][GEO_NCOEF] trilinear-map coefficients of the elements (launch_geo_coeffs);
                               // the kernel then RECOMPUTES qdata = SetupGeo(x) at every point (27 FMAs + adjugate) instead
                               // of streaming its 80 bytes per point from HBM
  const double *geo_aff;      // set (with geo) only when EVERY element is affine: [nelem][GEO_NAFF] = {det J, dXdx[9]} of the
                               // element (launch_geo_affine) -- the kernel then multiplies det J by the point's weight and
                               // reads the nine factors instead of forming J, its adjugate and a reciprocal at every point
  const double *geo_swept;    // set (with geo, without geo_aff) when EVERY element is SWEPT along the same reference direction geo_axis:
                               // x and y bilinear in the other two directions, z linear in that one (the prisms of an extruded mesh: the
                               // reference's cylinders).  [nelem][GEO_NSWEPT] (launch_geo_swept); J is then a 2 x 2 block and a constant,
                               // dXdx has five entries and the two products of the physics with it take 15 multiply-adds instead of 27
  int geo_axis;               // the sweep's reference direction (0, 1, 2), one for the whole mesh
  double qref[MAXN1D], qwt[MAXN1D];  // 1-D quadrature points / weights of the geometry (used with geo)
  // Even-odd form of the 1-D tables (pencil_even_odd(Q)).  The tables of symmetric point sets are
  // centro-symmetric (interp: M[N-1-i][K-1-j] = M[i][j]) or centro-antisymmetric (derivatives), so with
  // xe = x_j + x_{K-1-j}, xo = x_j - x_{K-1-j} an N x K product costs ~N K / 2 FMAs + N + K adds instead of N K FMAs.
  // eo[t]: t = 0 B, 1 B^T, 2 D, 3 D^T, 4 G, 5 G^T; per table Me[r][j] at r * (K/2) + j, Mo at 16 + r * (K/2) + j, the
  // middle column at 32 + r (r < (N+1)/2, j < K/2); built and checked by the host when the operator is planned.
  double eo[6][EO_TAB];
  int direct;                 // results at ELEMENT-INTERIOR nodes (0 < i,j,k < P-1; one contributor, verified on the
                               // host) are stored straight into y and skip the E-vector round trip; the E-vector then is
                               // [elem][shell node][3] and the transpose map handed to launch_assemble() holds the shell
                               // nodes only, its columns being positions e * element_shell_size(P) + rank
  // ---- host side only (behind everything the kernels read) ----
  int wave_groups;            // > 0: every wave is given at most this many groups (grid = groups / wave_groups) instead of
                               // the persistent grid (launch_fused_pencil_t)
  int waves_per_cu;           // > 0: persistent waves per CU (tuning hook CEED_MI355X_PENCIL_WAVES)
  int *query_waves;           // if set, the launcher stores the number of persistent waves a full launch has and launches nothing
};
hipError_t launch_clock_probe(long long *out /* device: shader cycles, 100 MHz ticks */, int spin_us, hipStream_t s);
// compute units of the current device (one device per process: cached)
int device_cu_count();
// element-interior test shared by the kernel and the host-side map builder
#ifdef __HIPCC__
__host__ __device__
#endif
static inline bool node_is_element_interior(int n, int P) {
  const int i = n % P, j = (n / P) % P, k = n / (P * P);
  return i > 0 && i < P - 1 && j > 0 && j < P - 1 && k > 0 && k < P - 1;
}
// rank of node n among the SHELL (non-interior) nodes of its element: the E-vector of the direct-store mode holds shell
// entries only, [element][shell rank][component].  FACE-MAJOR order (round 2): the two k-faces whole (edges and vertices
// included), then the two j-faces without the rows the k-faces hold, then the two i-faces' interiors -- so the nodes an
// element shares with ONE neighbour are contiguous in both elements' blocks, and k_assemble, whose consecutive rows are
// the nodes of one shared face, reads runs of whole faces instead of every fifth 24-byte record of an i-face (round 1's
// lexicographic order: 0.15 GB of line over-fetch per apply).  CPS_SHELL_LEX restores the lexicographic order (A/B).
#ifdef __HIPCC__
__host__ __device__
#endif
static inline int node_shell_rank(int n, int P) {
  const int m = P - 2, i = n % P, j = (n / P) % P, k = n / (P * P);
#ifdef CPS_SHELL_LEX
#define CPS_CLAMPM(v) ((v) < 0 ? 0 : ((v) > m ? m : (v)))
  int before = CPS_CLAMPM(k - 1) * m * m;                  // interior nodes in the planes below
  if (k > 0 && k < P - 1) {
    before += CPS_CLAMPM(j - 1) * m;                       // ... in the rows below of this plane
    if (j > 0 && j < P - 1) before += CPS_CLAMPM(i - 1);   // ... to the left in this row
  }
#undef CPS_CLAMPM
  return n - before;
#else
  if (k == 0) return j * P + i;
  if (k == P - 1) return P * P + j * P + i;
  if (j == 0) return 2 * P * P + (k - 1) * P + i;
  if (j == P - 1) return 2 * P * P + P * m + (k - 1) * P + i;
  if (i == 0) return 2 * P * P + 2 * P * m + (k - 1) * m + (j - 1);
  return 2 * P * P + 2 * P * m + m * m + (k - 1) * m + (j - 1);   // i == P - 1 (interior nodes have no rank)
#endif
}
#ifdef __HIPCC__
__host__ __device__
#endif
static inline int element_shell_size(int P) { return P * P * P - (P > 2 ? (P - 2) * (P - 2) * (P - 2) : 0); }
// 24-byte records between the shell E-vector blocks of consecutive elements (tuning hook, round 4: 16 makes every block a whole number
// of 128-byte lines -- measured, nothing: profiles/r04_ab_experiments.txt item 19)
#ifndef CPS_EVEC_ALIGN
#define CPS_EVEC_ALIGN 1
#endif
#ifdef __HIPCC__
__host__ __device__
#endif
static inline int evec_block_records(int P) { return (element_shell_size(P) + CPS_EVEC_ALIGN - 1) / CPS_EVEC_ALIGN * CPS_EVEC_ALIGN; }

// p-multigrid transfer in OWNER form (kernels_misc.hip, k_transfer): every fine node belongs to the first element that holds it.
struct TransferArgs {
  const uint32_t *off_c;  // coarse [nelem][Pc^3], the coarse side's Dirichlet flags in the top bits
  const uint32_t *own_f;  // fine   [nelem][Pf^3]: offset | Dirichlet flags of the nodes this element OWNS, 0xFFFFFFFF for the others
  const double *x;
  double *y;              // prolong: the fine L-vector, stored by the owners
  const double *w_f;      // per fine dof: (scale, e.g. multiplicity^-1 over all ranks) x (local multiplicity); null when every entry is 1
  int nelem;
  int mask_c, mask_f;     // honour the flags of the coarse / fine side
  int add;                // prolong: y += (restrict: launch_assemble() adds)
  double *evec;           // restrict: element results ([elem][Pc^3][3], masked entries as zeros); launch_assemble() sums them
                          // into y in element order
};
hipError_t launch_transfer_weights(double *w, const double *scale, size_t n, int *n_not_unit, hipStream_t s);

// x_c(xi) = a0 + a[c][0] xi + a[c][1] eta + a[c][2] zeta + a[c][3] xi eta + a[c][4] xi zeta + a[c][5] eta zeta + a[c][6] xi eta zeta
// on [-1,1]^3: the 7 coefficients per component that the Jacobian d x / d xi needs, [c][m] per element.
constexpr int GEO_NCOEF = 21;
hipError_t launch_geo_coeffs(const uint32_t *off_x, const double *xcoord, double *geo, int nelem, hipStream_t s);
// Affine elements (the 12 coefficients of the xi eta ... terms vanish to 1e-14 of the linear ones: parallelepipeds).  For
// every element {det J, dXdx[9]} -- SetupGeo's output at ANY of its points but for the quadrature weight -- goes to
// aff[e][GEO_NAFF]; *n_not_affine (device, zeroed by the caller) counts the elements that are NOT affine.
constexpr int GEO_NAFF = 10;
hipError_t launch_geo_affine(const double *geo, double *aff, int nelem, int *n_not_affine, hipStream_t s);
// Swept elements: z = z0 + zs xi_s for one reference direction s, x and y bilinear in the other two (a < b) and free of xi_s, all to
// 1e-13 of the largest linear coefficient.  sw[e][GEO_NSWEPT] = {x_a, x_b, x_ab, y_a, y_b, y_ab, sgn zs, 1 / zs} for the element's OWN
// sweep direction (sgn = -1 for s = 1: (a, b, s) is then an odd permutation of the reference directions).  Two passes: axis < 0 COUNTS --
// count[s] (device, zeroed by the caller) the elements swept along s, EVERY direction an element qualifies for (an axis-aligned brick
// qualifies for all three, so a mesh that mixes bricks with z-swept prisms still finds its common direction), count[3] the ones that are
// not swept at all (or along x or y in space); axis >= 0 FILLS sw[] for that direction.
constexpr int GEO_NSWEPT = 8;
hipError_t launch_geo_swept(const double *geo, double *sw, int nelem, int *count, int axis, hipStream_t s);

struct SetupGeoArgs {
  const uint32_t *off_x;  // [nelem][8]
  const double *xcoord;   // interlaced [vertex][3]
  double *qdata;          // [nelem][10][Q^3]
  int nelem;
};

struct DiagArgs {
  const uint32_t *offsets;
  double *diag;  // pre-zeroed L-vector
  const double *qdata, *state_in;
  int nelem, mask_out;
  double nu, E, lambda, TwoMu;
  double *evec;  // element contributions ([elem][P^3][3]); launch_assemble() sums them
};

// Each returns hipSuccess or the launch error; `name` receives a static string
// naming the instantiation, or the call returns hipErrorInvalidValue when the
// (P, Q, qf) combination is not instantiated.
hipError_t launch_fused_grad(int P, int Q, int qf, const BasisTables &t, const FusedGradArgs &a,
                             hipStream_t s, const char **name);
hipError_t launch_transfer(int Pc, int Pf, bool prolong, const BasisTables &t, const TransferArgs &a,
                           hipStream_t s, const char **name);
hipError_t launch_setup_geo(int Q, const BasisTables &t, const SetupGeoArgs &a, hipStream_t s,
                            const char **name);
hipError_t launch_diag(int P, int Q, int qf, const BasisTables &t, const DiagArgs &a, hipStream_t s,
                       const char **name);

// The instantiations of one quadrature size are compiled as pencil_inst_parts(Q) objects (kernels_fused_inst.hip, -DCPS_PART=<k>): the kernels
// with (Q - P) % parts == k -- the Q = 8 object alone took 72 s of a 90 s build.  csrc/Makefile lists the same parts.
constexpr int pencil_inst_parts(int Q) { return Q == 8 ? 4 : (Q == 7 ? 2 : 1); }
// elements per wave (= per group) of the pencil kernel, PencilGeom<P, Q>::E: 3 Q^2 pencils per element and pass against 64
// lanes and the 9 Q^3-double LDS slab per element
#ifndef CPS_PENCIL_E5
#define CPS_PENCIL_E5 2      // (tuning hook for variant builds: elements per wave at Q = 5)
#endif
constexpr int pencil_group_elems(int Q) { return Q <= 2 ? 8 : (Q <= 4 ? 4 : (Q == 5 ? CPS_PENCIL_E5 : 1)); }

// Deterministic, atomic-free E^T: y[node_off[r] + c] (+)= sum over the node's contributors, in element
// order, of E[3 * cols[k] + c] (cols[k] = e * P3 + n).  `flags` (one byte per node, bit c = component c constrained) may be null.
// `unpack` (may be null): the arrivals of a halo exchange, added to y by extra workgroups of the SAME launch (one launch
// less on the critical path of a split-phase apply): y[dst[u]] += recv[slot[k]], k in [ptr[u], ptr[u+1]), in list order.
struct HaloUnpackArgs { const uint32_t *dst, *ptr, *slot; const double *recv; int n; };
// `pack` (may be null): the pack of a halo exchange folded into the rows' launch -- row r's finished sums also go to
// send[slot[k] & 0x3FFFFFFF], component slot[k] >> 30, k in [ptr[r], ptr[r+1]) (ptr over the rows of THIS launch).
struct HaloPackFold { const uint32_t *ptr, *slot; double *send; };
hipError_t launch_assemble(const uint32_t *rowptr, const uint32_t *cols, const uint32_t *node_off,
                           const unsigned char *flags, const double *evec, double *y, int nnodes,
                           int add, hipStream_t s, int max_blocks = 0, const HaloUnpackArgs *unpack = nullptr,
                           const HaloPackFold *pack = nullptr);   // max_blocks: cap on the grid (pipelined assembly)

// The same sum with an epilogue in place of the store of y (kernels_misc.hip, k_assemble_epi): the output of a fused apply consumed
// where it is formed -- a Chebyshev step or the residual b - A v.
enum EpilogueKind : int { EPI_NONE = 0, EPI_CHEB = 1, EPI_RESID = 2 };
struct EpilogueArgs {
  int kind;
  const double *t;          // the apply's output vector: holds the ELEMENT-INTERIOR nodes' values (stored by the fused kernel) only
  const uint32_t *int_off;  // node offsets of the element-interior nodes handled by this launch, n_int of them (0: none)
  int n_int;
  // EPI_CHEB: r = (r0 ? r0 : r) - t (stored if r is given; r0 = the right-hand side b: the residual recomputed, not recurred);
  //           d = c1 dinv r + c2 d;  x = assign_x ? d : x + d
  double *x, *d, *r;
  const double *r0, *dinv;
  double c1, c2;
  int assign_x;
  // EPI_RESID: w = b - t
  double *w;
  const double *b;
};
hipError_t launch_assemble_epi(const uint32_t *rowptr, const uint32_t *cols, const uint32_t *node_off, const unsigned char *flags,
                               const double *evec, int nnodes, const EpilogueArgs &ep, hipStream_t s, int max_blocks = 0);

// Coordinate-driven set-up operators (kernels_coord.hip): opSetupForce and opTrue of setuplibceed.c:555-623.
struct CoordOpArgs {
  const uint32_t *off_x;   // [nelem][8] coordinate restriction
  const double *xcoord;    // interlaced [vertex][3]
  const uint32_t *off_u;   // [nelem][Pout^3] displacement restriction
  double *y;               // output L-vector, pre-zeroed, accumulated over the elements
  const double *qdata;     // [nelem][10][Q^3] (forcing) or null (true solution)
  int nelem, Q, Pout, mode;  // mode 0: SetupConstantForce, 1: SetupMMSForce, 2: MMSTrueSoln
  double ctx[3];           // direction | (nu, E)
  double bx[MAXN1D * 2];   // coordinate basis interp1d, Q x 2
  double bu[MAXN1D * MAXN1D];  // displacement basis interp1d, Q x Pout (forcing only)
};
hipError_t launch_coord_op(const CoordOpArgs &a, hipStream_t s);
// Strain-energy operator (kernels_coord.hip): opEnergy of setuplibceed.c:651-670.
struct EnergyOpArgs {
  const uint32_t *off_u;   // [nelem][P^3] displacement restriction (3 interlaced components)
  const double *u;         // displacement L-vector
  const uint32_t *off_e;   // [nelem][P^3] energy restriction (1 component)
  double *y;               // energy L-vector, pre-zeroed
  const double *qdata;     // [nelem][10][Q^3]
  int nelem, Q, P, model;  // model 0: LinElas, 1: HyperSS, 2: HyperFS
  int diag;                // 0: *Energy -> 1 component through INTERP^T; 1: *Diagnostic (opDiagnostic, setuplibceed.c:712-737)
                           // -> 8 components collocated with the points, off_e then is the [nelem][Q^3] diagnostic restriction
  double nu, E;
  double interp[MAXN1D * MAXN1D], grad[MAXN1D * MAXN1D];  // displacement basis, Q x P
  double interp_e[MAXN1D * MAXN1D];                        // energy basis, Q x P
};
hipError_t launch_energy_op(const EnergyOpArgs &a, hipStream_t s);

// Interface-dof halo exchange (CeedXHalo*, the L-vector sum of src/matops.c:57 across GPUs).  ONE launch packs the entries
// of all neighbour lists into the (contiguous) send buffer; ONE launch adds all arrivals: a thread per distinct destination
// entry adds that entry's arrivals in neighbour-list order (no atomics; the sum is reproducible).
hipError_t launch_halo_pack(const uint32_t *idx, int n, const double *y, double *buf, hipStream_t s);
hipError_t launch_halo_unpack_add(const HaloUnpackArgs &u, double *y, hipStream_t s);

// Assembled coarse-level operator (kernels_csr.hip).
hipError_t launch_csr_sum(const uint32_t *slotptr, const uint32_t *perm, const double *coo, double *vals, int nnz,
                          const uint32_t *unit_diag_slot, int n_unit, hipStream_t s);
hipError_t launch_csr_spmv(const uint32_t *rowptr, const uint32_t *cols, const double *vals, const double *x, double *y,
                           int nrows, hipStream_t s);
// CSR-stream form: row_block[b] .. row_block[b + 1] are consecutive rows with at most 2048 entries in all (a longer row is a run of its
// own), at most 256 rows per run; cut by csr_row_blocks() on the host
hipError_t launch_csr_spmv_stream(const uint32_t *row_block, int nblocks, const uint32_t *rowptr, const uint32_t *cols, const double *vals,
                                  const double *x, double *y, hipStream_t s);
hipError_t launch_csr_diag(const uint32_t *diag_slot_of_row, const double *vals, double *d, int nrows, hipStream_t s);
hipError_t launch_csr_spgemm(const uint32_t *l_rowptr, const uint32_t *l_cols, const double *l_vals, const uint32_t *r_rowptr,
                             const uint32_t *r_cols, const double *r_vals, const uint32_t *c_rowptr, const uint32_t *c_cols, double *c_vals,
                             int nrows, hipStream_t s, int dense_ncols = 0, int max_row_c = 0);   // C = L R on fixed patterns (R's columns sorted within each row);
                             // dense_ncols > 0: C is a dense row-major nrows x dense_ncols matrix (LDS row accumulation);
                             // max_row_c > 0: the longest row of C -- up to 4096 entries a wave per row with an LDS accumulator (k_csr_spgemm_row)
hipError_t launch_dense_spd_inverse(double *A, int n, double *scratch /* 1024 doubles */, int *info, hipStream_t s);

// Vector / restriction utilities.
hipError_t launch_set_value(double *v, size_t n, double val, hipStream_t s);
hipError_t launch_waxpby(double *w, double a, const double *x, double b, const double *y, size_t n, hipStream_t s);
hipError_t launch_cheb_update(double *x, double *d, double *r, const double *r0, const double *t, const double *dinv, double c1, double c2,
                              int assign_x, size_t n, hipStream_t s);
hipError_t launch_reciprocal(double *v, size_t n, hipStream_t s);
hipError_t launch_pointwise_mult(double *w, const double *x, const double *y, size_t n, hipStream_t s);
hipError_t launch_axpby(double *y, double a, const double *x, double b, size_t n, hipStream_t s);
hipError_t launch_masked_copy(double *dst, const double *src, const unsigned char *mask, size_t n,
                              hipStream_t s);  // dst = mask ? 0 : src
hipError_t launch_rstr_gather(const uint32_t *off, int nelem, int elemsize, int ncomp, int compstride,
                              const double *l, double *e, hipStream_t s);
hipError_t launch_rstr_scatter_add(const uint32_t *off, int nelem, int elemsize, int ncomp,
                                   int compstride, const double *e, double *l, hipStream_t s);
hipError_t launch_multiplicity(const uint32_t *off, int nelem, int elemsize, int ncomp, int compstride,
                               double *l, hipStream_t s);
hipError_t launch_dot(const double *x, const double *y, const double *w, size_t n, double *result_dev,
                      hipStream_t s, double *out = nullptr);  // result_dev[0] (and *out) = sum w_i x_i y_i (w may be null), reproducibly; result_dev: 1 + 2048 doubles
hipError_t launch_scalar_div(double *sc, int dst, int num, int den, double scale, hipStream_t s);
hipError_t launch_axpby_dev(double *y, const double *sc, int ia, double sa, const double *x, int ib, double sb, size_t n, hipStream_t s);

}  // namespace cps
