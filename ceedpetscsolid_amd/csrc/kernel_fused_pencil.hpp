// kernel_fused_pencil.hpp -- the fused operator kernel: PENCIL PER LANE.
//
// The whole CeedOperatorApply of the residual / Jacobian operators (setuplibceed.c:517-542, :817-839) in one launch:
// gather, interp, grad, QFunction, grad^T, interp^T, element results to y / the shell E-vector.
//
// Why pencils.  The first-generation kernel of round 1 (one OUTPUT point of a contraction per lane; removed from the
// tree in round 3, last present at commit 14cd5f8) gave every lane one output point, so the Q lanes
// that share an input row each read the whole row from LDS (Q-fold redundant LDS reads) and
// per-lane coefficient rows also come from LDS: ~71 % LDS-pipe busy, LDS bound.  Here a lane
// owns a whole PENCIL (one line of the element along the contraction direction, one
// component): it reads the NIN inputs once, produces all NOUT outputs in registers and writes
// them back IN PLACE.  Every value is read once and written once per pass, and the
// coefficients are wave-uniform, so they are scalar operands (kernarg -> SGPR), not LDS
// traffic.  Measured (PMC, config 4): LDS instructions per element 441 -> 220, LDS index unit ~30 % busy.
//
// Pipeline for a group of E elements owned by ONE wave64 (no s_barrier anywhere: a wave's LDS
// queue is executed in order).  Arrays A, BX, BZ: [c][k][j][i], strides (1, Q, Q^2, Q^3) doubles.
//   gather  x -> A (nodes)                          node-owner lanes
//   F1..F3  interpolate along i, j, k in place      pencil lanes;  F3 also writes dU/dz -> BZ (grad1d)
//   F4, F5  collocated d/dx: A -> BX, d/dy: A -> A  pencil lanes
//   QF      9 gradient entries in, 9 out, in place  point-owner lanes, one round of 64 points at a
//                                                   time; stored state prefetched two rounds ahead; the
//                                                   geometric factors RECOMPUTED from the element's
//                                                   trilinear map (GEO) or read with the state
//   B1..B5  transposes of F5..F1, accumulating      pencil lanes
//   final   A (nodes) -> y (element-interior nodes), shell E-vector (shared nodes) / atomics   node-owner lanes
// Elements per wave E: 8 (Q=2), 4 (Q=3,4), 2 (Q=5), 1 (Q>=6): a pass has 3*Q^2 pencils per element,
// E picks how well they fill 64 lanes against the LDS slab (9*Q^3 doubles per element).
#pragma once
#include "kernels_common.hpp"
#include "qfunctions_device.hpp"

namespace cps {

template <int P, int Q> struct PencilGeom {
  static constexpr int Q3 = Q * Q * Q, P3 = P * P * P;
  static constexpr int E = pencil_group_elems(Q);                            // elements per wave
  static constexpr int SJ = Q, SK = Q * Q, SC = Q3;                         // strides in doubles
  static constexpr int ARR = 3 * SC;                                        // one 3-component array
  static constexpr int PAD = Q == 5 ? 5 : 1;                                // tools/pencil_layout_search.py
#ifdef CPS_TIMING_ALIAS_BZ   // TIMING-ONLY build (WRONG results): BZ aliased onto BX, a 6 Q^3 slab -- the upper bound of what holding the
  static constexpr int SE = 2 * ARR + PAD;   // k-direction in registers could buy in occupancy at Q = 7 (profiles/r05_ab_experiments.txt item 1)
#else
  static constexpr int SE = 3 * ARR + PAD;                                  // element slab: A, BX, BZ
#endif
  static constexpr int RQ = (E * Q3 + 63) / 64;                             // point rounds per group
  static constexpr int RN = (E * P3 + 63) / 64;                             // node rounds per group
  static constexpr int GEO = E * GEO_NCOEF + 2 * Q;                           // element map coefficients + 1-D points / weights
  static constexpr int LDS_BYTES = (E * SE + GEO) * 8;
};

// ---- LDS accessors: VOLATILE 8-byte accesses through an LDS pointer + constant offset ----------
// Volatile does two jobs.  (1) hipcc keeps each access a separate ds_read_b64 / ds_write_b64 with an
// immediate offset (and still tracks them: it places counted s_waitcnt lgkmcnt(N) itself) instead of
// pairing the strided pencil reads into ds_read2_b64, which runs at half the byte rate of ds_read_b64
// on gfx950 (MI355X_MICROARCH.md, LDS table).  (2) Volatile accesses are never reordered against
// each other, which is all the ordering the passes need: they communicate through LDS across lanes
// of ONE wave, whose LDS queue the hardware executes in order.
// (Inline-asm ds_read_b64 + explicit waits were tried first: the compiler does not know that an asm
// output is still in flight, so it may copy or spill the register before the wait -- it did.)
typedef __attribute__((address_space(3))) double lds_double;
typedef volatile lds_double *ldsp_t;
typedef volatile __attribute__((address_space(3))) char *ldsb_t;
// Wave priority by phase (s_setprio, 0 ... 3).  The two waves of a SIMD share its vector pipe; the arbiter prefers the higher
// priority.  A wave in its pencil passes issues short bursts of FMAs between LDS round trips, a wave in its q-point rounds a long
// run of them: with the passes at 3 and the q-point rounds at 0 a pass never waits behind the other wave's physics, its LDS
// latency is hidden by that physics, and two waves that started together drift half a group apart by themselves.  Same-box A/B
// (profiles/r03_ab_experiments.txt item 13): config 4 -2.6 %, 13 200 hexes -8 %, config 5's block -3.6 %, p = 2 -6.5 %; a
// level for the requests / gather / final stores of their own, and for the loads and stores inside a q-point round: no
// further gain; the staggered start of round 2 (-1.9 % then) adds nothing beside it and left the kernel.
// (the levels are tuning hooks for variant builds, tools/mkvariant.sh; -1: never set)
#ifndef CPS_PRIO_TOP
#define CPS_PRIO_TOP -1    // requests for the next group, gather, final stores: stay at the passes' level
#endif
#ifndef CPS_PRIO_PASS
#define CPS_PRIO_PASS 3    // the twelve pencil passes
#endif
#ifndef CPS_PRIO_PASS_B
#define CPS_PRIO_PASS_B CPS_PRIO_PASS   // the transposed passes after the q-point rounds
#endif
#ifndef CPS_PRIO_PHYS
#define CPS_PRIO_PHYS 0    // the q-point rounds
#endif
template <int LEVEL> CPS_DEV void set_prio() {
  if constexpr (LEVEL >= 0) __builtin_amdgcn_s_setprio(LEVEL);
}
// (diagnostic build only, tools/phase_timing.py) -DCPS_PHASE_TIMING=<k>: every wave writes the shader-clock time stamps of the
// phase boundaries of its k-th group to the buffer whose address the environment gives (CEED_MI355X_PHASE_BUF), 32 per wave.
#ifdef CPS_PHASE_TIMING
#define CPS_PH(i) do { if (ph_iter == CPS_PHASE_TIMING && lane == 0 && ph_buf) ph_buf[(size_t)blockIdx.x * 32 + (i)] = clock64(); } while (0)
#else
#define CPS_PH(i) do { } while (0)
#endif
template <int OFF>
CPS_DEV double lds_rd(ldsp_t a) {
  static_assert(OFF >= 0 && OFF < 65536 && OFF % 8 == 0, "ds offset field is 16 bits");
  return a[OFF / 8];
}
template <int OFF>
CPS_DEV void lds_wr(ldsp_t a, double v) {
  static_assert(OFF >= 0 && OFF < 65536 && OFF % 8 == 0, "ds offset field is 16 bits");
  a[OFF / 8] = v;
}
// N values at stride SB bytes from byte offset OFF
template <int N, int SB, int OFF, int M = 0>
CPS_DEV void pencil_ld(ldsp_t a, double *r) {
  if constexpr (M < N) {
    r[M] = lds_rd<OFF + M * SB>(a);
    pencil_ld<N, SB, OFF, M + 1>(a, r);
  }
}
template <int N, int SB, int OFF, int M = 0>
CPS_DEV void pencil_st(ldsp_t a, const double *r) {
  if constexpr (M < N) {
    lds_wr<OFF + M * SB>(a, r[M]);
    pencil_st<N, SB, OFF, M + 1>(a, r);
  }
}

// Coefficient tables are read straight from the kernarg segment (constant address space ->
// s_load, SGPR operands of the FMAs).  Each pass takes a freshly "laundered" pointer, so the
// compiler loads that pass's coefficients inside the pass instead of hoisting all three tables
// out of the element loop (150 SGPRs: it then spilled them to VGPR lanes, 900 v_readlane per group).
typedef const __attribute__((address_space(4))) double *ktab_t;
CPS_DEV ktab_t ktab_fresh(ktab_t p) {
  // "memory": the previous pass's last stores (and the FMAs feeding them) stay above this point, so two passes'
  // tables are never live together (they do not fit the SGPR file: 16 of them were spilled to VGPR lanes)
  asm volatile("" : "+s"(p) : : "memory");
  return p;
}
// a fresh table pointer that the compiler cannot use before the N values at `dep` have been computed
template <int N>
CPS_DEV ktab_t ktab_fresh_after(ktab_t p, const double *dep) {
#pragma unroll
  for (int i = 0; i < N; i++) asm volatile("" : "+s"(p) : "v"(dep[i]));
  return p;
}
// The launch arguments are read the same way: a laundered pointer into the kernarg segment at every use site, so
// that pointers and scalars are re-read with s_load (no VALU) where they are needed instead of living in SGPRs
// through the whole element loop, where the register allocator spills them to VGPR lanes (v_readlane per use).
typedef const __attribute__((address_space(4))) FusedGradArgs *kargs_t;
template <bool LAUNDER>
CPS_DEV kargs_t kargs_fresh() {
  static_assert(sizeof(BasisTables) % 8 == 0, "second kernel argument follows the tables without padding");
  auto p = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(BasisTables);
  if constexpr (LAUNDER) asm volatile("" : "+s"(p));
  return (kargs_t)p;
}
// out[o] += sum_m M(o, m) in[m];  M(o, m) = TR ? tab[m * LD + o] : tab[o * LD + m]  (wave-uniform -> SGPR operands)
template <int NOUT, int NIN, int LD, bool TR>
CPS_DEV void pencil_mac(ktab_t tab, const double *in, double *out) {
#pragma unroll
  for (int o = 0; o < NOUT; o++) {
#pragma unroll
    for (int m = 0; m < NIN; m++) out[o] += (TR ? tab[m * LD + o] : tab[o * LD + m]) * in[m];
  }
}
// The same product with the table in even-odd form (FusedGradArgs::eo; SGN = +1 centro-symmetric, -1 antisymmetric):
// xe_j = x_j + x_{K-1-j}, xo_j = x_j - x_{K-1-j};  E_i = sum_j Me[i][j] xe_j + Mm[i] x_mid,  O_i = sum_j Mo[i][j] xo_j;
// y_i = E_i + O_i,  y_{N-1-i} = SGN (E_i - O_i);  the middle row keeps only its non-vanishing half.
template <int NOUT, int NIN, int SGN>
CPS_DEV void pencil_mac_eo(ktab_t T, const double *in, double *out) {
  constexpr int HIN = NIN / 2, HOUT = NOUT / 2;
  double xe[HIN > 0 ? HIN : 1], xo[HIN > 0 ? HIN : 1];
#pragma unroll
  for (int j = 0; j < HIN; j++) { xe[j] = in[j] + in[NIN - 1 - j]; xo[j] = in[j] - in[NIN - 1 - j]; }
#pragma unroll
  for (int i = 0; i < HOUT; i++) {
    double ev = (NIN & 1) ? T[32 + i] * in[HIN] : 0., od = 0.;
#pragma unroll
    for (int j = 0; j < HIN; j++) { ev += T[i * HIN + j] * xe[j]; od += T[16 + i * HIN + j] * xo[j]; }
    out[i] += ev + od;
    out[NOUT - 1 - i] += SGN > 0 ? ev - od : od - ev;
  }
  if constexpr (NOUT & 1) {
    double mid = (SGN > 0 && (NIN & 1)) ? T[32 + HOUT] * in[HIN] : 0.;
#pragma unroll
    for (int j = 0; j < HIN; j++) mid += SGN > 0 ? T[HOUT * HIN + j] * xe[j] : T[16 + HOUT * HIN + j] * xo[j];
    out[NOUT - 1 - HOUT] += mid;
  }
}
// plain or even-odd product; `tab` is the matching table (BasisTables entry or FusedGradArgs::eo[t])
template <int NOUT, int NIN, int LD, bool TR, int SGN, bool EO>
CPS_DEV void mac_sel(ktab_t tab, const double *in, double *out) {
  if constexpr (EO) pencil_mac_eo<NOUT, NIN, SGN>(tab, in, out);
  else pencil_mac<NOUT, NIN, LD, TR>(tab, in, out);
}
// round r of a pass with `ntask` tasks: is this lane's task t = lane + 64 r a real one?
CPS_DEV bool pencil_ok(int lane, int r, int ntask) { return ntask == 0 ? lane < 0 : ((r + 1) * 64 <= ntask ? true : lane + 64 * r < ntask); }
// Task -> lane mapping of a pass.  FLAT: task t = lane + 64 r over (element, component, b, a), a fastest.  BLOCKED (round 4): every
// (element, component) block of n = NA NB pencils starts on a 32- or 64-lane boundary, the lanes behind its last pencil idle --
// taken where that padding costs no extra round (Q = 5: 6 blocks of 25 in 6 half-waves = the same 3 rounds; Q = 7: 3 blocks of 49 in
// 3 waves).  A ds_read_b64 is served in the lane groups {0-31}, {32-63} (MI355X_MICROARCH.md, LDS): with the blocks aligned to them a
// group reads ONE component's pencils -- consecutive or evenly strided words, conflict-free along i and k -- instead of the tail of
// one component and the head of the next, 125 words further on, whose banks collide (tools/lds_conflict_model.py: LDS-array cycles
// per group of two elements 2 021 -> 1 896 at Q = 5).  Validity keeps the form of pencil_ok: the pass is handed a VIRTUAL lane
// (negative for the lanes of a real block, huge for idle ones) and ntask = 0, so that `vlane + 64 r < 0` holds exactly for the
// rounds in which the lane's block exists.
// Only the STORES of a blocked pass are predicated: an idle lane holds the address of its block's last pencil (identical addresses
// broadcast: no bank conflict) and loads and multiplies like its neighbours -- predicated loads would keep every round's registers
// live across the others' (256 VGPRs and scratch when tried).  Hence ntask = 0 means: every block exists (3 E blk = 64 rounds exactly).
#ifndef CPS_PENCIL_BLOCKED
#define CPS_PENCIL_BLOCKED 1
#endif
constexpr int pencil_blk(int n, int E) {   // lanes per block, or 0: flat
  if (!CPS_PENCIL_BLOCKED) return 0;
  const int flat = (3 * E * n + 63) / 64, b = n <= 32 ? 32 : (n <= 64 ? 64 : 0);
  return (b && b != n && (3 * E * b) % 64 == 0 && (3 * E * b) / 64 == flat) ? b : 0;
}
// loads and arithmetic of round r: real task, or any lane of a blocked pass
#ifndef CPS_BLOCKED_IDLE_MATH
#define CPS_BLOCKED_IDLE_MATH 0   // 1: the idle lanes of a blocked pass load and multiply too (only their stores are masked)
#endif
CPS_DEV bool pencil_ok_ld(int lane, int r, int ntask) { return (ntask == 0 && CPS_BLOCKED_IDLE_MATH) ? true : pencil_ok(lane, r, ntask); }

// One single-input pass: every task reads its pencil (NIN entries at stride SB bytes from array SRC),
// applies the NOUT x NIN matrix and writes NOUT entries to array DST (DST == SRC: in place).  All
// rounds' reads are issued first, so the waits are counted ones and the FMAs of one round overlap the
// reads of the next; tasks are disjoint pencils, so reads may pass the in-place writes of other rounds.
// A NOUT x NIN table is 2 NOUT NIN SGPRs: 50 at 5 x 5, but 98 at 7 x 7 and 128 at 8 x 8 -- more than the SGPR file, and
// the register allocator then spills coefficients to VGPR lanes (~950 v_readlane / v_writelane per element at Q = 7,
// a fifth of the VALU work).  Larger tables are therefore applied in SPLITS of whole output rows, each split loaded
// (s_load) only after the previous one's FMAs: all rounds of a pass keep their inputs and outputs in VGPRs meanwhile.
template <int NOUT, int NIN, bool EO = false> constexpr int table_splits() {
  return (EO || NOUT * NIN <= 30) ? 1 : (NOUT * NIN <= 56 ? 2 : 3);   // an even-odd table is at most 30 coefficients
}
// rows [O0, O1) of the product of pencil_mac
template <int O0, int O1, int NIN, int LD, bool TR>
CPS_DEV void pencil_mac_rows(ktab_t tab, const double *in, double *out) {
#pragma unroll
  for (int o = O0; o < O1; o++) {
#pragma unroll
    for (int m = 0; m < NIN; m++) out[o] += (TR ? tab[m * LD + o] : tab[o * LD + m]) * in[m];
  }
}
// out[r] += M in[r] for all rounds r of a pass, the table taken in table_splits() row blocks
// (DEP0: the first block, too, waits for the values already in `out` -- a previous product accumulated there)
template <int NOUT, int NIN, int LD, bool TR, int R, bool DEP0 = false, int S = 0>
CPS_DEV void mac_rounds(ktab_t table, const double (&in)[R][NIN], double (&out)[R][NOUT], int lane, int ntask) {
  constexpr int NS = table_splits<NOUT, NIN>(), H = (NOUT + NS - 1) / NS;
  if constexpr (S < NS) {
    const ktab_t t = (S > 0 || DEP0) ? ktab_fresh_after<R * NOUT>(table, &out[0][0]) : ktab_fresh(table);
#pragma unroll
    for (int r = 0; r < R; r++)
      if (pencil_ok_ld(lane, r, ntask)) pencil_mac_rows<S * H, ((S + 1) * H < NOUT ? (S + 1) * H : NOUT), NIN, LD, TR>(t, in[r], out[r]);
    mac_rounds<NOUT, NIN, LD, TR, R, DEP0, S + 1>(table, in, out, lane, ntask);
  }
}
template <int NIN, int NOUT, int LD, bool TR, int SB, int SRC, int DST, int SGN, bool EO, int R>
CPS_DEV void pencil_pass_impl(ktab_t table, const ldsp_t (&addr)[R], int lane, int ntask) {
  double in[R][NIN];
#pragma unroll
  for (int r = 0; r < R; r++)
    if (pencil_ok_ld(lane, r, ntask)) pencil_ld<NIN, SB, SRC>(addr[r], in[r]);
  if constexpr (table_splits<NOUT, NIN, EO>() == 1) {
    const ktab_t t = ktab_fresh(table);
#pragma unroll
    for (int r = 0; r < R; r++)
      if (pencil_ok_ld(lane, r, ntask)) {
        double out[NOUT] = {};
        mac_sel<NOUT, NIN, LD, TR, SGN, EO>(t, in[r], out);
        if (pencil_ok(lane, r, ntask)) pencil_st<NOUT, SB, DST>(addr[r], out);
      }
  } else {
    double out[R][NOUT];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
      for (int o = 0; o < NOUT; o++) out[r][o] = 0.;
    mac_rounds<NOUT, NIN, LD, TR, R>(table, in, out, lane, ntask);
#pragma unroll
    for (int r = 0; r < R; r++)
      if (pencil_ok(lane, r, ntask)) pencil_st<NOUT, SB, DST>(addr[r], out[r]);
  }
}

// blocked pass (ntask = 0, lane = the virtual lane): ONE exec region around the whole pass for the lanes that own a pencil, every
// round unconditional inside it (per-round predicates on loads and products cost 256 VGPRs and scratch when tried)
template <int NIN, int NOUT, int LD, bool TR, int SB, int SRC, int DST, int SGN, bool EO, int R>
CPS_DEV void pencil_pass(ktab_t table, const ldsp_t (&addr)[R], int lane, int ntask) {
  if (ntask == 0) {
    if (CPS_BLOCKED_IDLE_MATH || lane < 0) pencil_pass_impl<NIN, NOUT, LD, TR, SB, SRC, DST, SGN, EO, R>(table, addr, CPS_BLOCKED_IDLE_MATH ? lane : 0, CPS_BLOCKED_IDLE_MATH ? 0 : 64 * R);
  } else pencil_pass_impl<NIN, NOUT, LD, TR, SB, SRC, DST, SGN, EO, R>(table, addr, lane, ntask);
}

#ifndef CPS_PENCIL_MINW
#define CPS_PENCIL_MINW 2   // waves per SIMD the register allocation is held to (256 VGPRs)
#endif
#ifndef CPS_PENCIL_NSET
#define CPS_PENCIL_NSET 2   // q-point register sets: 2 = every round's data is requested two rounds ahead
#endif                      // (198 VGPRs with the hyperFS tangent; 1 set: 4-6 % slower; 3 sets: no gain)
#ifndef CPS_PENCIL_MINW5
#define CPS_PENCIL_MINW5 CPS_PENCIL_MINW   // (tuning hook for variant builds: Q = 5 only)
#endif
constexpr int pencil_minw(int Q) { return Q == 5 ? CPS_PENCIL_MINW5 : CPS_PENCIL_MINW; }
#ifndef CPS_PENCIL_NSET_BIGQ
#define CPS_PENCIL_NSET_BIGQ 2   // Q >= 6: two sets as well since round 4.  Rounds 1-3 had ONE (a second set pushed the hyperFS tangent past
#endif                           // 256 VGPRs: 26 spilled); the blocked task -> lane mapping freed ~18 registers at Q = 7 and the second set
                                 // fits (202-236 VGPRs at Q = 6, 7).  Same-box A/B (profiles/r04_ab_experiments.txt item 12): config 5's block
                                 // -3.3 %, the whole 64^3 box -4.5 %, Q = 6 -2.3 %, hyperSS at Q = 7 -7.1 %
// GEO = 1: the geometric factors are recomputed per point from the element's trilinear map (FusedGradArgs::geo) instead of
// read; GEO = 2: every element of the mesh is AFFINE (a parallelepiped: the box meshes of configs 1, 2 and 5), dXdx and
// det J are constants of the element (FusedGradArgs::geo_aff, ten doubles) and only the weight varies from point to point;
// GEO = 3: every element is SWEPT along the same reference direction (FusedGradArgs::geo_swept, geo_axis): x and y bilinear in the other two
// directions, z linear in that one -- the prisms of an extruded mesh, which the reference's cylinders are.  J is a 2 x 2 block and a
// constant: four FMAs, a 2 x 2 determinant and one reciprocal per point instead of 27 FMAs, an adjugate and a 3 x 3 determinant, and the
// physics multiplies with the five entries of dXdx that are left (qf_point<QF, true>): 15 multiply-adds for each of its two products
// instead of 27.  The q-point round hands the reference directions to the physics in the order (in-plane, in-plane, sweep) by choosing
// which of the three arrays it reads first -- a relabelling of a sum's terms, no arithmetic;
// GEO = 0: qdata is read.  The 1-D tables are applied in even-odd form wherever that form exists (pencil_even_odd(Q)).
template <int P, int Q, int QF, int GEO>
__global__ __launch_bounds__(64, pencil_minw(Q)) void k_fused_pencil(const BasisTables tab_, const FusedGradArgs a) {
  static_assert(offsetof(BasisTables, interp) == 0 && offsetof(BasisTables, colo) == 8 * MAXN1D * MAXN1D &&
                offsetof(BasisTables, grad) == 16 * MAXN1D * MAXN1D, "kernarg layout of the tables");
  (void)tab_;  // first kernel argument: lives at offset 0 of the kernarg segment, read through kt below
  constexpr bool EO = pencil_even_odd(Q);
  const ktab_t kt = (ktab_t)__builtin_amdgcn_kernarg_segment_ptr();
  const ktab_t ktB = kt, ktD = kt + MAXN1D * MAXN1D, ktG = kt + 2 * MAXN1D * MAXN1D;
  // the six products' tables: plain (B, D, G read forward or transposed) or their even-odd forms (FusedGradArgs::eo)
  const ktab_t eo0 = (ktab_t)((const __attribute__((address_space(4))) char *)kt + sizeof(BasisTables) + offsetof(FusedGradArgs, eo));
  const ktab_t tBf = EO ? eo0 : ktB, tBt = EO ? eo0 + EO_TAB : ktB, tDf = EO ? eo0 + 2 * EO_TAB : ktD,
               tDt = EO ? eo0 + 3 * EO_TAB : ktD, tGf = EO ? eo0 + 4 * EO_TAB : ktG, tGt = EO ? eo0 + 5 * EO_TAB : ktG;
  using G = PencilGeom<P, Q>;
  constexpr int Q3 = G::Q3, P3 = G::P3, E = G::E, RQ = G::RQ, RN = G::RN;
  constexpr int SJ = G::SJ, SK = G::SK, SC = G::SC, SE = G::SE;
  constexpr int BI = 8, BJ = 8 * SJ, BK = 8 * SK, BC = 8 * SC;        // byte strides
#ifdef CPS_TIMING_ALIAS_BZ
  constexpr int oA = 0, oBX = 8 * G::ARR, oBZ = 8 * G::ARR;
#else
  constexpr int oA = 0, oBX = 8 * G::ARR, oBZ = 16 * G::ARR;         // byte offsets of the arrays
#endif
  constexpr bool ST_IN = QFTraits<QF>::state_in, ST_OUT = QFTraits<QF>::state_out;
  constexpr int NST = QFTraits<QF>::nstate;
  constexpr int QS = Q3;
  static_assert(P <= Q, "interpolation to at least as many points as nodes");
  // re-read the launch arguments at every use (kargs_fresh) where that frees the SGPR file of spills: Q <= 5.  At
  // Q >= 6 one coefficient table alone (2 Q^2 SGPRs) overflows it and the extra scalar-load waits only cost (measured).
  constexpr bool KA = Q <= 5;

  __shared__ __attribute__((aligned(16))) double slab[E * SE + G::GEO];
  const ldsp_t lds0 = (lds_double *)slab;
  constexpr bool geo = GEO != 0;   // the geometric factors are not read from qdata
  constexpr int NCO = GEO == 2 ? GEO_NAFF : (GEO == 3 ? GEO_NSWEPT : GEO_NCOEF);   // doubles per element kept in LDS for them
  const int lane = threadIdx.x & 63;
  // ---- work list of this wave -----------------------------------------------------------------------
  // The groups are cut into 8 contiguous chunks, one per XCD (neighbouring elements share their nodes through one L2):
  // block b serves chunk b % 8 (blocks b and b + 8 share an XCD under the round-robin placement), group = chunk begin +
  // wave rank + k * waves per chunk.  (A dynamic per-XCD ticket schedule was built and measured in round 2: not faster on
  // large launches, slower on small ones; removed in round 3.)
  const int ngroups = (a.nelem + E - 1) / E;
  const int nxcd = gridDim.x % 8 == 0 ? 8 : 1;
  const int wrank = (int)blockIdx.x / nxcd, wper = (int)gridDim.x / nxcd;
  const int gend = min(ngroups, (int)(blockIdx.x % nxcd + 1) * ((ngroups + nxcd - 1) / nxcd));
  int grp = (int)(blockIdx.x % nxcd) * ((ngroups + nxcd - 1) / nxcd) + wrank;
  if (grp >= gend) return;

  // ---- loop-invariant lane -> work maps -----------------------------------------------------
  // pencil passes: task t = lane + 64 r over (element, component, b, a), a fastest; address of the
  // pencil's first entry.  Five families: direction i over nodal / quadrature (j,k), direction j over
  // (i', nodal / quadrature k), direction k over (i', j').
  auto pencil_addr = [&](int t, int NA, int NB, int sa, int sb, int blk) -> ldsp_t {
    const int n = NA * NB;
    const int bl = blk ? t / blk : t / n, pen = blk ? min(t % blk, n - 1) : t % n;      // (idle lanes of a block: any address in it)
    const int el = min(bl / 3, E - 1), c = bl % 3, ia = pen % NA, ib = pen / NA;
    return lds0 + (el * SE + c * SC + (ia * sa + ib * sb) / 8);
  };
  // virtual lane of pencil_ok for a blocked family (see pencil_blk): real block <=> vlane + 64 r < 0
  auto pencil_vlane = [&](int n, int blk) -> int { return blk == 0 ? lane : ((lane % blk) < n ? blk * (lane / blk) - blk * 3 * E : (1 << 24)); };
  constexpr int T_IP = 3 * P * P, T_IQ = 3 * Q * Q, T_JP = 3 * Q * P, T_JQ = 3 * Q * Q, T_K = 3 * Q * Q;
  constexpr int B_PP = pencil_blk(P * P, E), B_QP = pencil_blk(Q * P, E), B_QQ = pencil_blk(Q * Q, E);
  constexpr int R_IP = (E * T_IP + 63) / 64, R_IQ = (E * T_IQ + 63) / 64, R_JP = (E * T_JP + 63) / 64,
                R_JQ = (E * T_JQ + 63) / 64, R_K = (E * T_K + 63) / 64;
  // what the passes are handed as (lane, ntask): the real ones (flat) or (virtual lane, 0) (blocked)
  constexpr int N_IP = B_PP ? 0 : E * T_IP, N_JP = B_QP ? 0 : E * T_JP, N_QQ = B_QQ ? 0 : E * T_IQ;
  const int lPP = pencil_vlane(P * P, B_PP), lQP = pencil_vlane(Q * P, B_QP), lQQ = pencil_vlane(Q * Q, B_QQ);
  // the k passes and the two-input j pass are written out below: a blocked one sits in ONE exec region (lQQ < 0), all its rounds unconditional
  constexpr bool PIN_ALL = !B_QQ || CPS_BLOCKED_IDLE_MATH;
  const int lQQi = PIN_ALL ? lQQ : 0;
  constexpr int N_QQi = PIN_ALL ? N_QQ : 64 * R_K;
  ldsp_t aIP[R_IP], aIQ[R_IQ], aJP[R_JP], aJQ[R_JQ], aK[R_K];
#pragma unroll
  for (int r = 0; r < R_IP; r++) aIP[r] = pencil_addr(lane + 64 * r, P, P, BJ, BK, B_PP);
#pragma unroll
  for (int r = 0; r < R_IQ; r++) aIQ[r] = pencil_addr(lane + 64 * r, Q, Q, BJ, BK, B_QQ);
#pragma unroll
  for (int r = 0; r < R_JP; r++) aJP[r] = pencil_addr(lane + 64 * r, Q, P, BI, BK, B_QP);
#pragma unroll
  for (int r = 0; r < R_JQ; r++) aJQ[r] = pencil_addr(lane + 64 * r, Q, Q, BI, BK, B_QQ);
#pragma unroll
  for (int r = 0; r < R_K; r++) aK[r] = pencil_addr(lane + 64 * r, Q, Q, BI, BJ, B_QQ);
  // point owners: q = lane + 64 r over (element, k, j, i); node owners likewise over P^3
  ldsp_t aPt[RQ], aNd[RN];
#pragma unroll
  for (int r = 0; r < RQ; r++) {
    const int t = lane + 64 * r, el = t / Q3, q = t % Q3;
    aPt[r] = lds0 + (el * SE + (q / (Q * Q)) * SK + ((q / Q) % Q) * SJ + q % Q);
  }
  // geometry recompute: packed byte offsets of the point's 1-D indices into the LDS tables, and its element
  uint32_t pqi[RQ];
#pragma unroll
  for (int r = 0; r < RQ; r++) {
    const int t = min(lane + 64 * r, E * Q3 - 1), el = t / Q3, q = t % Q3;
    pqi[r] = (uint32_t)((q % Q) * 8) | ((uint32_t)(((q / Q) % Q) * 8) << 8) | ((uint32_t)((q / (Q * Q)) * 8) << 16) | ((uint32_t)el << 24);
  }
  constexpr int oGC = E * SE * 8, oGT = oGC + E * GEO_NCOEF * 8;   // byte offsets of the coefficient / table areas (sized for GEO = 1)
  if (geo && lane < 2 * Q) {   // 1-D points then weights (written once; the LDS queue orders it before any read)
    const auto kp = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(BasisTables);
    const double v = lane < Q ? ((kargs_t)kp)->qref[lane] : ((kargs_t)kp)->qwt[lane - Q];
    *(ldsp_t)((ldsb_t)lds0 + oGT + lane * 8) = v;
  }
  uint32_t nd_interior = 0;  // bit r: this lane's node of round r is interior to its element (direct store to y)
  uint32_t ev_idx[(RN + 1) / 2] = {};  // E-vector entry (in doubles) of this lane's node within the group's block, 16 bits each
  static_assert(E * (P3 * 3 + 15) < 65536, "16-bit E-vector entry index");
#pragma unroll
  for (int r = 0; r < RN; r++) {
    const int t = lane + 64 * r, el = t / P3, n = t % P3;
    aNd[r] = lds0 + (el * SE + (n / (P * P)) * SK + ((n / P) % P) * SJ + n % P);
    if (a.direct && node_is_element_interior(n, P)) nd_interior |= 1u << r;
    // [element][shell rank or node][3], the element blocks a.evec_stride doubles apart
    ev_idx[r / 2] |= (uint32_t)(min(el, E - 1) * a.evec_stride + (a.direct ? node_shell_rank(n, P) : n) * 3) << (16 * (r % 2));
  }
  // element-in-group of owner slot t, recomputed where needed (a few compares) instead of held in
  // registers through the physics
  auto el_of = [&](int t, int n3) {
    int el = 0;
#pragma unroll
    for (int e = 1; e < E; e++) el += (t >= e * n3) ? 1 : 0;
    return el;
  };

  // ---- global-memory side: clamped, unconditional loads ------------------------------------------
  // Addressing: a wave-uniform 64-bit base per group (SGPRs) plus a 32-bit per-lane index inside the
  // group's block.  Lanes of a dead element (only in the last, partial group) read the group's last
  // live element instead.
  auto nlive_of = [&](int nelem, int g) { const int n = nelem - g * E; return n < E ? n : E; };  // uniform, >= 1
  // (Q >= 6 with qdata READ -- ten more doubles per point and set -- keeps one set: the residual kernel at Q = 6 spilled with two)
  constexpr int NSET = RQ >= 2 ? (Q >= 6 ? (GEO == 0 ? 1 : CPS_PENCIL_NSET_BIGQ) : CPS_PENCIL_NSET) : 1;
  double qd[NSET][10], st[NSET][NST];
  auto load_point = [&](double *qdv, double *stv, int g, int r) {
    const kargs_t ka = kargs_fresh<KA>();  // one scalar load for the fields used here
    const int t = lane + 64 * r, el0 = el_of(t, Q3);
    const int q = min(t - el0 * Q3, Q3 - 1), el = min(el0, nlive_of(ka->nelem, g) - 1);
    const size_t e0 = (size_t)(ka->elem_begin + g * E);
    const double *qb = ka->qdata + e0 * (10 * Q3);
    const uint32_t vo = (uint32_t)(el * (10 * Q3) + q);
    if (!geo) {
#pragma unroll
      for (int c = 0; c < 10; c++) qdv[c] = (qb + c * Q3)[vo];
    }
    if constexpr (ST_IN) {
      const double *sb = ka->state_in + e0 * (NST * QS);
      const uint32_t vs = (uint32_t)(el * (NST * QS) + q);
#pragma unroll
      for (int c = 0; c < NST; c++) stv[c] = (sb + c * QS)[vs];
    }
  };
  auto load_offsets = [&](int g, uint32_t *o) {
    const kargs_t ka = kargs_fresh<KA>();
    const uint32_t *ob = ka->offsets + (size_t)(ka->elem_begin + g * E) * P3;
    const int nlive = nlive_of(ka->nelem, g);
#pragma unroll
    for (int r = 0; r < RN; r++) {
      const int t = lane + 64 * r, el0 = el_of(t, P3);
      const int n = min(t - el0 * P3, P3 - 1), el = min(el0, nlive - 1);
      o[r] = ob[(uint32_t)(el * P3 + n)];
    }
  };
  auto load_x = [&](const uint32_t *o, double (*xv)[3]) {
    const kargs_t kx = kargs_fresh<KA>();
    const double *xb = kx->x;
#pragma unroll
    for (int r = 0; r < RN; r++) {
      const uint32_t base = o[r] & OFF_MASK;
#pragma unroll
      for (int c = 0; c < 3; c++) xv[r][c] = xb[base + c];
    }
  };

  uint32_t off[RN], off_nx[RN];
  double xin[RN][3];
  load_offsets(grp, off);
  load_x(off, xin);
#pragma unroll
  for (int t = 0; t < NSET; t++) load_point(qd[t], st[t], grp, t);

#ifdef CPS_PHASE_TIMING
  int ph_iter = 0;
  long long *const ph_buf = (long long *)a.query_waves;
#endif
  for (;;) {
    CPS_PH(0);
#ifdef CPS_PHASE_TIMING
    if (ph_iter == CPS_PHASE_TIMING && lane == 0 && ph_buf) ph_buf[(size_t)blockIdx.x * 32 + 24] = wall_clock64();   // 100 MHz
#endif
    const int grp_nx = grp + wper;
    const bool more = grp_nx < gend;  // wave-uniform
    const int g_nx = more ? grp_nx : grp;
    load_offsets(g_nx, off_nx);
    constexpr int RG = (E * NCO + 63) / 64;
    double gcoef[RG];    // this group's element-map coefficients (or affine factors), lane + 64 i; into LDS right before the physics
    if (geo) {
      const kargs_t ka = kargs_fresh<KA>();
      const int nlive = nlive_of(ka->nelem, grp);
      const double *gb = (GEO == 2 ? ka->geo_aff : (GEO == 3 ? ka->geo_swept : ka->geo)) + (size_t)(ka->elem_begin + grp * E) * NCO;
#pragma unroll
      for (int i = 0; i < RG; i++) {
        const int t = min(lane + 64 * i, E * NCO - 1), el = min(t / NCO, nlive - 1);
        gcoef[i] = gb[(uint32_t)(el * NCO + t % NCO)];
      }
    }

    // ---- gather: x -> A at the nodes (Dirichlet flags applied; dead elements of the last group zero) ----
    const kargs_t kg = kargs_fresh<KA>();
    const int g_nelem = kg->nelem, g_mask_in = kg->mask_in;
#pragma unroll
    for (int r = 0; r < RN; r++) {
      if (pencil_ok(lane, r, E * P3)) {
        const bool live = grp * E + el_of(lane + 64 * r, P3) < g_nelem;
        const uint32_t fl = live ? (g_mask_in ? (off[r] >> OFF_FLAG_SHIFT) : 0u) : 7u;
        lds_wr<oA + 0 * BC>(aNd[r], (fl & 1u) ? 0. : xin[r][0]);
        lds_wr<oA + 1 * BC>(aNd[r], (fl & 2u) ? 0. : xin[r][1]);
        lds_wr<oA + 2 * BC>(aNd[r], (fl & 4u) ? 0. : xin[r][2]);
      }
    }

    // ---- B: nodes -> points, in place -----------------------------------------------------------------
    set_prio<CPS_PRIO_PASS>();
    CPS_PH(1);
    pencil_pass<P, Q, P, false, BI, oA, oA, +1, EO>(tBf, aIP, lPP, N_IP);   // F1: along i at nodal (j, k)
    CPS_PH(2);
    pencil_pass<P, Q, P, false, BJ, oA, oA, +1, EO>(tBf, aJP, lQP, N_JP);   // F2: along j at (i', nodal k)
    CPS_PH(3);
    if (PIN_ALL || lQQ < 0) {  // F3: along k at (i', j'): U -> A in place and dU/dz -> BZ (grad1d on the nodal values), one
       // table at a time (both = 100 SGPRs = SGPR spills)
      double in[R_K][P];
#pragma unroll
      for (int r = 0; r < R_K; r++)
        if (pencil_ok_ld(lQQi, r, N_QQi)) pencil_ld<P, BK, oA>(aK[r], in[r]);
      if constexpr (table_splits<Q, P, EO>() == 1) {
        const ktab_t tB = ktab_fresh(tBf);
#pragma unroll
        for (int r = 0; r < R_K; r++)
          if (pencil_ok_ld(lQQi, r, N_QQi)) {
            double out[Q] = {};
            mac_sel<Q, P, P, false, +1, EO>(tB, in[r], out);
            if (pencil_ok(lQQi, r, N_QQi)) pencil_st<Q, BK, oA>(aK[r], out);
          }
        const ktab_t tG = ktab_fresh(tGf);
#pragma unroll
        for (int r = 0; r < R_K; r++)
          if (pencil_ok_ld(lQQi, r, N_QQi)) {
            double dz[Q] = {};
            mac_sel<Q, P, P, false, -1, EO>(tG, in[r], dz);
            if (pencil_ok(lQQi, r, N_QQi)) pencil_st<Q, BK, oBZ>(aK[r], dz);
          }
      } else {  // large tables: row blocks (mac_rounds), one product after the other
        double out[R_K][Q];
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
#pragma unroll
          for (int r = 0; r < R_K; r++)
#pragma unroll
            for (int o = 0; o < Q; o++) out[r][o] = 0.;
          mac_rounds<Q, P, P, false, R_K>(pass == 0 ? ktB : ktG, in, out, lQQi, N_QQi);
#pragma unroll
          for (int r = 0; r < R_K; r++)
            if (pencil_ok(lQQi, r, N_QQi)) {
              if (pass == 0) pencil_st<Q, BK, oA>(aK[r], out[r]);
              else pencil_st<Q, BK, oBZ>(aK[r], out[r]);
            }
        }
      }
    }
    // ---- collocated gradient on the quadrature points -----------------------------------------------------
    CPS_PH(4);
    pencil_pass<Q, Q, Q, false, BI, oA, oBX, -1, EO>(tDf, aIQ, lQQ, N_QQ);  // F4: d/dx: A -> BX
    CPS_PH(5);
    pencil_pass<Q, Q, Q, false, BJ, oA, oA, -1, EO>(tDf, aJQ, lQQ, N_QQ);   // F5: d/dy: A -> A in place
    CPS_PH(6);

    set_prio<CPS_PRIO_PHYS>();
    if (geo) {
#pragma unroll
      for (int i = 0; i < RG; i++)
        if (lane + 64 * i < E * NCO)
          *(ldsp_t)((ldsb_t)lds0 + oGC + (lane + 64 * i) * 8) = gcoef[i];
    }
    // ---- physics: point owners, one round at a time; ug[d*3+c] from (BX, A, BZ), dv back in place ----
#pragma unroll
    for (int r = 0; r < RQ; r++) {
      CPS_PH(7 + (r < 8 ? r : 8));
      const kargs_t ka = kargs_fresh<KA>();
      const bool okp = pencil_ok(lane, r, E * Q3);
      const int pel = el_of(lane + 64 * r, Q3), pq = lane + 64 * r - pel * Q3;
      const bool live = okp && (grp * E + pel < ka->nelem);
      double ug[9], dv[9], sto[9];
      // swept elements: the arrays of the two in-plane reference directions first, the sweep direction's last (wave-uniform byte offsets)
      ldsp_t pd0 = aPt[r], pd1 = aPt[r], pd2 = aPt[r];
      int sw_a = 0, sw_b = 1;
      if constexpr (GEO == 3) {
        const int ax = ka->geo_axis;
        sw_a = ax == 0 ? 1 : 0; sw_b = ax == 2 ? 1 : 2;
        auto arr = [](int d) { return d == 0 ? oBX : (d == 1 ? oA : oBZ); };
        pd0 = (ldsp_t)((ldsb_t)aPt[r] + arr(sw_a)); pd1 = (ldsp_t)((ldsb_t)aPt[r] + arr(sw_b)); pd2 = (ldsp_t)((ldsb_t)aPt[r] + arr(ax));
      }
      if (okp) {
        if constexpr (GEO == 3) {
          ug[0] = lds_rd<0 * BC>(pd0); ug[1] = lds_rd<1 * BC>(pd0); ug[2] = lds_rd<2 * BC>(pd0);
          ug[3] = lds_rd<0 * BC>(pd1); ug[4] = lds_rd<1 * BC>(pd1); ug[5] = lds_rd<2 * BC>(pd1);
          ug[6] = lds_rd<0 * BC>(pd2); ug[7] = lds_rd<1 * BC>(pd2); ug[8] = lds_rd<2 * BC>(pd2);
        } else {
          ug[0] = lds_rd<oBX + 0 * BC>(aPt[r]); ug[1] = lds_rd<oBX + 1 * BC>(aPt[r]); ug[2] = lds_rd<oBX + 2 * BC>(aPt[r]);
          ug[3] = lds_rd<oA + 0 * BC>(aPt[r]);  ug[4] = lds_rd<oA + 1 * BC>(aPt[r]);  ug[5] = lds_rd<oA + 2 * BC>(aPt[r]);
          ug[6] = lds_rd<oBZ + 0 * BC>(aPt[r]); ug[7] = lds_rd<oBZ + 1 * BC>(aPt[r]); ug[8] = lds_rd<oBZ + 2 * BC>(aPt[r]);
        }
      }
      double qdl[10];
      if constexpr (GEO == 2) {  // affine element: dXdx and det J are the element's, the weight the point's (common.h:47-101)
        const uint32_t pk = pqi[r];
        const auto lb = (ldsb_t)lds0;
        const ldsp_t ti = (ldsp_t)(lb + oGT + (pk & 0xFFu)), tj = (ldsp_t)(lb + oGT + ((pk >> 8) & 0xFFu)),
                     tk = (ldsp_t)(lb + oGT + ((pk >> 16) & 0xFFu)), cf = (ldsp_t)(lb + oGC + (pk >> 24) * (GEO_NAFF * 8));
        qdl[0] = ti[Q] * tj[Q] * tk[Q] * cf[0];
#pragma unroll
        for (int c = 1; c < 10; c++) qdl[c] = cf[c];
      } else if constexpr (GEO == 3) {  // swept element: J = {2 x 2 block in the plane, zs along the sweep} (common.h:47-101 on that J)
        const uint32_t pk = pqi[r];
        const auto lb = (ldsb_t)lds0;
        const ldsp_t ti = (ldsp_t)(lb + oGT + (pk & 0xFFu)), tj = (ldsp_t)(lb + oGT + ((pk >> 8) & 0xFFu)),
                     tk = (ldsp_t)(lb + oGT + ((pk >> 16) & 0xFFu)), cf = (ldsp_t)(lb + oGC + (pk >> 24) * (GEO_NSWEPT * 8));
        const ldsp_t ta = (ldsp_t)(lb + oGT + ((pk >> (8 * sw_a)) & 0xFFu)), tb = (ldsp_t)(lb + oGT + ((pk >> (8 * sw_b)) & 0xFFu));
        const double xa = ta[0], xb = tb[0], w = ti[Q] * tj[Q] * tk[Q];
        const double xab = cf[2], yab = cf[5];
        const double J00 = __builtin_fma(xab, xb, cf[0]), J10 = __builtin_fma(xab, xa, cf[1]);   // d x / d xi_a, d x / d xi_b
        const double J01 = __builtin_fma(yab, xb, cf[3]), J11 = __builtin_fma(yab, xa, cf[4]);   // d y / d xi_a, d y / d xi_b
        const double d2 = __builtin_fma(J00, J11, -(J01 * J10));
        const double r2 = rcp_nr(d2), nr2 = -r2;
        qdl[0] = w * cf[6] * d2;                       // w det J (cf[6] = sgn zs: the sign of the permutation (a, b, sweep))
        qdl[1] = J11 * r2; qdl[2] = J10 * nr2; qdl[3] = 0.;
        qdl[4] = J01 * nr2; qdl[5] = J00 * r2; qdl[6] = 0.;
        qdl[7] = 0.; qdl[8] = 0.; qdl[9] = cf[7];      // 1 / zs
      } else if (geo) {  // SetupGeo (common.h:47-101) recomputed at this point from the element's trilinear map
        const uint32_t pk = pqi[r];
        const auto lb = (ldsb_t)lds0;
        const ldsp_t ti = (ldsp_t)(lb + oGT + (pk & 0xFFu)), tj = (ldsp_t)(lb + oGT + ((pk >> 8) & 0xFFu)),
                     tk = (ldsp_t)(lb + oGT + ((pk >> 16) & 0xFFu)), cf = (ldsp_t)(lb + oGC + (pk >> 24) * (GEO_NCOEF * 8));
        const double xi = ti[0], eta = tj[0], zeta = tk[0], w = ti[Q] * tj[Q] * tk[Q];
        const double xe = xi * eta, xz = xi * zeta, ez = eta * zeta;
        double Jg[9];
#pragma unroll
        for (int c = 0; c < 3; c++) {  // m: 0 xi, 1 eta, 2 zeta, 3 xi eta, 4 xi zeta, 5 eta zeta, 6 xi eta zeta
          const double a0 = cf[c * 7 + 0], a1 = cf[c * 7 + 1], a2 = cf[c * 7 + 2], a3 = cf[c * 7 + 3], a4 = cf[c * 7 + 4],
                       a5 = cf[c * 7 + 5], a6 = cf[c * 7 + 6];
          Jg[0 * 3 + c] = a0 + a3 * eta + a4 * zeta + a6 * ez;
          Jg[1 * 3 + c] = a1 + a3 * xi + a5 * zeta + a6 * xz;
          Jg[2 * 3 + c] = a2 + a4 * xi + a5 * eta + a6 * xe;
        }
        qf_setup_geo_rcp(Jg, w, qdl);
      } else {
#pragma unroll
        for (int c = 0; c < 10; c++) qdl[c] = qd[r % NSET][c];
      }
      if (live) {
        if constexpr (QF == QF_HYPERFS_F) {
          double dso[10];
          double *db = ka->state_out2;      // wave-uniform: also write the derived state of the tangent?
          qf_point<QF, GEO == 3>(Phys{ka->nu, ka->E, ka->lambda, ka->TwoMu}, ug, qdl, st[r % NSET], dv, sto, db ? dso : nullptr);
          if (db) {
            db += (size_t)(ka->elem_begin + grp * E) * (10 * QS);
            const uint32_t vd = (uint32_t)(pel * (10 * QS) + pq);
#pragma unroll
            for (int c = 0; c < 10; c++) (db + c * QS)[vd] = dso[c];
          }
        } else {
          qf_point<QF, GEO == 3>(Phys{ka->nu, ka->E, ka->lambda, ka->TwoMu}, ug, qdl, st[r % NSET], dv, sto);
        }
        if constexpr (ST_OUT) {
          double *sb = ka->state_out + (size_t)(ka->elem_begin + grp * E) * (9 * QS);
          const uint32_t vs = (uint32_t)(pel * (9 * QS) + pq);
#pragma unroll
          for (int c = 0; c < 9; c++) (sb + c * QS)[vs] = sto[c];
        }
      } else {
#pragma unroll
        for (int c = 0; c < 9; c++) dv[c] = 0.;
      }
      // refill this register set for the round NSET further down the stream (this group or the next)
      if (r + NSET < RQ) load_point(qd[r % NSET], st[r % NSET], grp, r + NSET);
      else load_point(qd[r % NSET], st[r % NSET], g_nx, r + NSET - RQ);
      if (okp) {
        if constexpr (GEO == 3) {
          lds_wr<0 * BC>(pd0, dv[0]); lds_wr<1 * BC>(pd0, dv[1]); lds_wr<2 * BC>(pd0, dv[2]);
          lds_wr<0 * BC>(pd1, dv[3]); lds_wr<1 * BC>(pd1, dv[4]); lds_wr<2 * BC>(pd1, dv[5]);
          lds_wr<0 * BC>(pd2, dv[6]); lds_wr<1 * BC>(pd2, dv[7]); lds_wr<2 * BC>(pd2, dv[8]);
        } else {
          lds_wr<oBX + 0 * BC>(aPt[r], dv[0]); lds_wr<oBX + 1 * BC>(aPt[r], dv[1]); lds_wr<oBX + 2 * BC>(aPt[r], dv[2]);
          lds_wr<oA + 0 * BC>(aPt[r], dv[3]);  lds_wr<oA + 1 * BC>(aPt[r], dv[4]);  lds_wr<oA + 2 * BC>(aPt[r], dv[5]);
          lds_wr<oBZ + 0 * BC>(aPt[r], dv[6]); lds_wr<oBZ + 1 * BC>(aPt[r], dv[7]); lds_wr<oBZ + 2 * BC>(aPt[r], dv[8]);
        }
      }
    }

    // ---- gradient^T --------------------------------------------------------------------------------------
    set_prio<CPS_PRIO_PASS_B>();
    CPS_PH(16);
    pencil_pass<Q, Q, Q, true, BI, oBX, oBX, -1, EO>(tDt, aIQ, lQQ, N_QQ);  // B1: W1 = Dx^T g0, BX in place
    CPS_PH(17);
    if (PIN_ALL || lQQ < 0) {  // B2: W2 = W1 + Dy^T g1, A in place.  Two inputs per task: software-pipelined over the rounds
       // (two rounds of inputs live instead of all)
      if constexpr (table_splits<Q, Q, EO>() == 1) {
        const ktab_t tD = ktab_fresh(tDt);
        double in[2][Q], acc[2][Q];
        if (pencil_ok_ld(lQQi, 0, N_QQi)) { pencil_ld<Q, BJ, oA>(aJQ[0], in[0]); pencil_ld<Q, BJ, oBX>(aJQ[0], acc[0]); }
#pragma unroll
        for (int r = 0; r < R_JQ; r++) {
          if (r + 1 < R_JQ && pencil_ok_ld(lQQi, r + 1, N_QQi)) {
            pencil_ld<Q, BJ, oA>(aJQ[r + 1], in[(r + 1) & 1]);
            pencil_ld<Q, BJ, oBX>(aJQ[r + 1], acc[(r + 1) & 1]);
          }
          if (pencil_ok_ld(lQQi, r, N_QQi)) {
            mac_sel<Q, Q, Q, true, -1, EO>(tD, in[r & 1], acc[r & 1]);
            if (pencil_ok(lQQi, r, N_QQi)) pencil_st<Q, BJ, oA>(aJQ[r], acc[r & 1]);
          }
        }
      } else {  // large tables: all rounds live, the table in row blocks
        double in[R_JQ][Q], acc[R_JQ][Q];
#pragma unroll
        for (int r = 0; r < R_JQ; r++) {
#pragma unroll
          for (int o = 0; o < Q; o++) acc[r][o] = 0.;
          if (pencil_ok_ld(lQQi, r, N_QQi)) { pencil_ld<Q, BJ, oA>(aJQ[r], in[r]); pencil_ld<Q, BJ, oBX>(aJQ[r], acc[r]); }
        }
        mac_rounds<Q, Q, Q, true, R_JQ>(ktD, in, acc, lQQi, N_QQi);
#pragma unroll
        for (int r = 0; r < R_JQ; r++)
          if (pencil_ok(lQQi, r, N_QQi)) pencil_st<Q, BJ, oA>(aJQ[r], acc[r]);
      }
    }
    CPS_PH(18);
    if (PIN_ALL || lQQ < 0) {  // B3: along k: A[k<P] = B^T W2 + G^T g2, in two sweeps so that one coefficient table is live at a time
      double out[R_K][P];
      if constexpr (table_splits<P, Q, EO>() > 1) {  // large tables: all rounds live, the tables in row blocks
        double in[R_K][Q];
#pragma unroll
        for (int r = 0; r < R_K; r++) {
#pragma unroll
          for (int m = 0; m < P; m++) out[r][m] = 0.;
          if (pencil_ok_ld(lQQi, r, N_QQi)) pencil_ld<Q, BK, oA>(aK[r], in[r]);
        }
        mac_rounds<P, Q, P, true, R_K>(ktB, in, out, lQQi, N_QQi);
#pragma unroll
        for (int r = 0; r < R_K; r++)
          if (pencil_ok_ld(lQQi, r, N_QQi)) pencil_ld<Q, BK, oBZ>(aK[r], in[r]);
        mac_rounds<P, Q, P, true, R_K, true>(ktG, in, out, lQQi, N_QQi);
#pragma unroll
        for (int r = 0; r < R_K; r++)
          if (pencil_ok(lQQi, r, N_QQi)) pencil_st<P, BK, oA>(aK[r], out[r]);
      } else {
      {
        const ktab_t tB = ktab_fresh(tBt);
        double in[2][Q];
        if (pencil_ok_ld(lQQi, 0, N_QQi)) pencil_ld<Q, BK, oA>(aK[0], in[0]);
#pragma unroll
        for (int r = 0; r < R_K; r++) {
          if (r + 1 < R_K && pencil_ok_ld(lQQi, r + 1, N_QQi)) pencil_ld<Q, BK, oA>(aK[r + 1], in[(r + 1) & 1]);
#pragma unroll
          for (int m = 0; m < P; m++) out[r][m] = 0.;
          if (pencil_ok_ld(lQQi, r, N_QQi)) mac_sel<P, Q, P, true, +1, EO>(tB, in[r & 1], out[r]);
        }
      }
      {
        // not before the first sweep has used its table: both at once do not fit the SGPR file (they were spilled)
        const ktab_t tG = ktab_fresh_after<R_K * P>(tGt, &out[0][0]);
        double in2[2][Q];
        if (pencil_ok_ld(lQQi, 0, N_QQi)) pencil_ld<Q, BK, oBZ>(aK[0], in2[0]);
#pragma unroll
        for (int r = 0; r < R_K; r++) {
          if (r + 1 < R_K && pencil_ok_ld(lQQi, r + 1, N_QQi)) pencil_ld<Q, BK, oBZ>(aK[r + 1], in2[(r + 1) & 1]);
          if (pencil_ok_ld(lQQi, r, N_QQi)) {
            mac_sel<P, Q, P, true, -1, EO>(tG, in2[r & 1], out[r]);
            if (pencil_ok(lQQi, r, N_QQi)) pencil_st<P, BK, oA>(aK[r], out[r]);
          }
        }
      }
      }
    }
    CPS_PH(19);
    load_x(off_nx, xin);  // next group's x (its offsets landed long ago): issued this late so its 6 RN registers
                          // are not live across the physics and the register-hungry passes; B4, B5, the
                          // final store and the next gather's address work hide most of its latency
    // ---- B^T: points -> nodes ---------------------------------------------------------------------------
    pencil_pass<Q, P, P, true, BJ, oA, oA, +1, EO>(tBt, aJP, lQP, N_JP);    // B4: along j
    CPS_PH(20);
    pencil_pass<Q, P, P, true, BI, oA, oA, +1, EO>(tBt, aIP, lPP, N_IP);    // B5: along i
    CPS_PH(21);
    // ---- final: node owners -> y (element-interior nodes) / shell E-vector (plain coalesced stores) ------------------
    set_prio<CPS_PRIO_TOP>();
    {
      double v[RN][3];
#pragma unroll
      for (int r = 0; r < RN; r++)
        if (pencil_ok(lane, r, E * P3)) {
          v[r][0] = lds_rd<oA + 0 * BC>(aNd[r]); v[r][1] = lds_rd<oA + 1 * BC>(aNd[r]); v[r][2] = lds_rd<oA + 2 * BC>(aNd[r]);
        }
      const kargs_t ka = kargs_fresh<KA>();
#pragma unroll
      for (int r = 0; r < RN; r++) {
        const int nel = el_of(lane + 64 * r, P3);
        if (pencil_ok(lane, r, E * P3) && grp * E + nel < ka->nelem) {
          if ((nd_interior >> r) & 1u) {  // no contributor outside this wave: the node's final value goes straight to y
            const uint32_t base = off[r] & OFF_MASK;
            const uint32_t fl = ka->mask_out ? (off[r] >> OFF_FLAG_SHIFT) : 0u;
            double *yb = ka->y;
            yb[base] = (fl & 1u) ? 0. : v[r][0]; (yb + 1)[base] = (fl & 2u) ? 0. : v[r][1]; (yb + 2)[base] = (fl & 4u) ? 0. : v[r][2];
          } else {
            double *eb = ka->evec + (size_t)(ka->elem_begin + grp * E) * ka->evec_stride;
            const uint32_t ve = (r % 2) ? (ev_idx[r / 2] >> 16) : (ev_idx[r / 2] & 0xFFFFu);
            eb[ve] = v[r][0]; (eb + 1)[ve] = v[r][1]; (eb + 2)[ve] = v[r][2];
          }
        }
      }
    }
    CPS_PH(22);
#ifdef CPS_PHASE_TIMING
    if (ph_iter == CPS_PHASE_TIMING && lane == 0 && ph_buf) ph_buf[(size_t)blockIdx.x * 32 + 25] = wall_clock64();
    ph_iter++;
#endif
    if (!more) break;
    grp = grp_nx;
#pragma unroll
    for (int r = 0; r < RN; r++) off[r] = off_nx[r];
  }
}

template <int P, int Q> constexpr int pencil_waves_per_cu() {
  constexpr int by_lds = (160 * 1024) / PencilGeom<P, Q>::LDS_BYTES;
  return by_lds < 1 ? 1 : (by_lds > 4 * pencil_minw(Q) ? 4 * pencil_minw(Q) : by_lds);  // 8 waves per CU = 2 per SIMD at <= 256 VGPRs
}

// Launch shape.  Default: a PERSISTENT grid -- as many one-wave workgroups as the device holds at once (CUs x waves per
// CU), each walking its strided list of groups with the next group's data requested a group ahead.  `a.wave_groups` > 0
// (host-side hint, small launches): every wave is given at most that many groups instead and the grid grows accordingly
// (rounded to whole XCD sets) -- workgroups beyond the resident set are dispatched as earlier ones retire, which balances
// a launch of a few rounds and gives co-scheduled kernels (the halo exchange's) a slot at every retirement.
template <int P, int Q, int QF>
hipError_t launch_fused_pencil_t(const BasisTables &t, const FusedGradArgs &a_in, hipStream_t s) {
  using G = PencilGeom<P, Q>;
#ifdef CPS_PHASE_TIMING   // (diagnostic build) the time-stamp buffer rides in the query pointer, which a real launch does not use
  FusedGradArgs a = a_in;
  const bool is_query = a_in.query_waves != nullptr;
#else
  const FusedGradArgs &a = a_in;
  constexpr bool is_query_always = true;
#endif
  if (a.nelem <= 0 && !a.query_waves) return hipSuccess;
  const int ngroups = (a.nelem + G::E - 1) / G::E;
  const int ncu = device_cu_count();
  if (ncu <= 0) return hipErrorUnknown;
  const int resident = ncu * (a.waves_per_cu > 0 ? a.waves_per_cu : pencil_waves_per_cu<P, Q>());
#ifdef CPS_PHASE_TIMING
  if (is_query) { *a.query_waves = resident; return hipSuccess; }
  if (const char *e = getenv("CEED_MI355X_PHASE_BUF")) a.query_waves = (int *)strtoull(e, nullptr, 0);
#else
  if (a.query_waves) { *a.query_waves = resident; return hipSuccess; }
  (void)is_query_always;
#endif
  int grid = resident;
  if (a.wave_groups > 0) grid = ((ngroups + a.wave_groups - 1) / a.wave_groups + 7) / 8 * 8;
  // (A persistent grid SHRUNK so that every wave gets the same number of groups -- 1 650 waves x 4 groups instead of 2 048 x 3.2 at
  // 13 200 hexes -- was measured in round 3: 3 ... 14 % slower at every size from 5 500 to 99 000 hexes.  More waves in flight beat an
  // even finish: profiles/r03_ab_experiments.txt item 11.)
  if (grid > ngroups) grid = ngroups;
  if (a.geo && a.geo_aff) hipLaunchKernelGGL((k_fused_pencil<P, Q, QF, 2>), dim3(grid), dim3(64), 0, s, t, a);
  else if (a.geo && a.geo_swept) hipLaunchKernelGGL((k_fused_pencil<P, Q, QF, 3>), dim3(grid), dim3(64), 0, s, t, a);
  else if (a.geo) hipLaunchKernelGGL((k_fused_pencil<P, Q, QF, 1>), dim3(grid), dim3(64), 0, s, t, a);
  else hipLaunchKernelGGL((k_fused_pencil<P, Q, QF, 0>), dim3(grid), dim3(64), 0, s, t, a);
  return hipGetLastError();
}

}  // namespace cps
