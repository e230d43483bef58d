// kernels_misc.hip -- geometry set-up, p-multigrid transfer, diagonal assembly,
// restriction and vector utilities, and the (P,Q,qf) dispatch tables.
#include <algorithm>
#include <cstring>
#include "kernels_common.hpp"
#include "qfunctions_device.hpp"

namespace cps {

// ===========================================================================
// SetupGeo operator (setuplibceed.c:370-389): coordinates (P=2 per direction,
// :279,339) -> d x / d xi at the quadrature points -> qdata[10].
// ===========================================================================
template <int Q>
__global__ __launch_bounds__(Geom<Q>::BLOCK) void k_setup_geo(const BasisTables tab,
                                                               const SetupGeoArgs a) {
  using G = Geom<Q>;
  constexpr int Q3 = G::Q3, TPE = G::TPE, EPB = G::EPB;
  __shared__ double sx[EPB][24];
  const int tid = threadIdx.x, el = tid / TPE, q = tid % TPE;
  const int e = blockIdx.x * EPB + el;
  const bool live = e < a.nelem;
  if (live && q < 8) {
    const uint32_t base = a.off_x[(size_t)e * 8 + q] & OFF_MASK;
#pragma unroll
    for (int c = 0; c < 3; c++) sx[el][c * 8 + q] = a.xcoord[base + c];
  }
  __syncthreads();
  if (!live || q >= Q3) return;
  const int i = q % Q, j = (q / Q) % Q, k = q / (Q * Q);
  // tables of the coordinate basis: B[q][p], G[q][p] with P = 2
  const double bi[2] = {tab.interp[i * 2], tab.interp[i * 2 + 1]}, gi[2] = {tab.grad[i * 2], tab.grad[i * 2 + 1]};
  const double bj[2] = {tab.interp[j * 2], tab.interp[j * 2 + 1]}, gj[2] = {tab.grad[j * 2], tab.grad[j * 2 + 1]};
  const double bk[2] = {tab.interp[k * 2], tab.interp[k * 2 + 1]}, gk[2] = {tab.grad[k * 2], tab.grad[k * 2 + 1]};
  double Jg[9];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    double s0 = 0., s1 = 0., s2 = 0.;
#pragma unroll
    for (int cc = 0; cc < 2; cc++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int aa = 0; aa < 2; aa++) {
          const double x = sx[el][c * 8 + aa + 2 * b + 4 * cc];
          s0 += gi[aa] * bj[b] * bk[cc] * x;
          s1 += bi[aa] * gj[b] * bk[cc] * x;
          s2 += bi[aa] * bj[b] * gk[cc] * x;
        }
    Jg[0 * 3 + c] = s0; Jg[1 * 3 + c] = s1; Jg[2 * 3 + c] = s2;
  }
  double qd[10];
  qf_setup_geo(Jg, tab.qw[i] * tab.qw[j] * tab.qw[k], qd);
  double *out = a.qdata + (size_t)e * 10 * Q3 + q;
#pragma unroll
  for (int c = 0; c < 10; c++) out[c * Q3] = qd[c];
}

// Trilinear-map coefficients of every element (vertices in tensor order v = i + 2 j + 4 k, xi_v = +-1): what the fused
// kernel needs to recompute SetupGeo's output at a point (FusedGradArgs::geo).
__global__ void k_geo_coeffs(const uint32_t *off_x, const double *xcoord, double *geo, int nelem) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, e = t / 3, c = t % 3;
  if (e >= nelem) return;
  double x[8];
#pragma unroll
  for (int v = 0; v < 8; v++) x[v] = xcoord[(off_x[(size_t)e * 8 + v] & OFF_MASK) + c];
  // monomial m of the bit set B (1: xi, 2: eta, 4: zeta): a = 1/8 sum_v x_v prod_{d in B} s_d(v)
  const int B[7] = {1, 2, 4, 3, 5, 6, 7};
  double *out = geo + (size_t)e * GEO_NCOEF + c * 7;
#pragma unroll
  for (int m = 0; m < 7; m++) {
    double s = 0.;
#pragma unroll
    for (int v = 0; v < 8; v++) s += (__popc((unsigned)(~v & B[m])) & 1) ? -x[v] : x[v];
    out[m] = 0.125 * s;
  }
}
hipError_t launch_geo_coeffs(const uint32_t *off_x, const double *xcoord, double *geo, int nelem, hipStream_t s) {
  if (nelem <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_geo_coeffs, dim3((unsigned)((3 * nelem + 255) / 256)), dim3(256), 0, s, off_x, xcoord, geo, nelem);
  return hipGetLastError();
}

__global__ void k_geo_affine(const double *geo, double *aff, int nelem, int *n_not_affine) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nelem) return;
  const double *g = geo + (size_t)e * GEO_NCOEF;
  double lin = 0., nonlin = 0., Jg[9];
#pragma unroll
  for (int c = 0; c < 3; c++) {
#pragma unroll
    for (int m = 0; m < 7; m++) {
      const double v = fabs(g[c * 7 + m]);
      if (m < 3) lin = fmax(lin, v); else nonlin = fmax(nonlin, v);
    }
#pragma unroll
    for (int d = 0; d < 3; d++) Jg[d * 3 + c] = g[c * 7 + d];   // J[d][c] = d x_c / d xi_d, constant on the element
  }
  if (nonlin > 1e-14 * lin) atomicAdd(n_not_affine, 1);
  double qd[10];
  qf_setup_geo_rcp(Jg, 1.0, qd);     // {det J, dXdx}: the same arithmetic as the per-point recompute
#pragma unroll
  for (int i = 0; i < GEO_NAFF; i++) aff[(size_t)e * GEO_NAFF + i] = qd[i];
}
hipError_t launch_geo_affine(const double *geo, double *aff, int nelem, int *n_not_affine, hipStream_t s) {
  if (nelem <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_geo_affine, dim3((unsigned)((nelem + 255) / 256)), dim3(256), 0, s, geo, aff, nelem, n_not_affine);
  return hipGetLastError();
}

// axis < 0: COUNT -- count[s]++ for EVERY direction s the element is swept along (an axis-aligned brick qualifies for all three), count[3]++
// if for none; axis >= 0: FILL sw[] for that direction (the host has found every element to qualify for it).
__global__ void k_geo_swept(const double *geo, double *sw, int nelem, int *count, int axis) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nelem) return;
  const double *g = geo + (size_t)e * GEO_NCOEF;   // [c][m], m: 0 xi, 1 eta, 2 zeta, 3 xi eta, 4 xi zeta, 5 eta zeta, 6 xi eta zeta
  double lin = 0.;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int m = 0; m < 3; m++) lin = fmax(lin, fabs(g[c * 7 + m]));
  const double tol = 1e-13 * lin;
  int found = 3;
#pragma unroll
  for (int s = 0; s < 3; s++) {
    // monomials that contain direction s: the linear one, the two pairs with it, the triple
    const int p0 = s == 0 ? 3 : (s == 1 ? 3 : 4), p1 = s == 0 ? 4 : (s == 1 ? 5 : 5);
    bool ok = fabs(g[2 * 7 + s]) > tol;
#pragma unroll
    for (int m = 0; m < 7; m++)
      if (m != s) ok = ok && fabs(g[2 * 7 + m]) <= tol;                 // z depends on xi_s alone
#pragma unroll
    for (int c = 0; c < 2; c++)
      ok = ok && fabs(g[c * 7 + s]) <= tol && fabs(g[c * 7 + p0]) <= tol && fabs(g[c * 7 + p1]) <= tol && fabs(g[c * 7 + 6]) <= tol;
    if (ok && axis < 0) atomicAdd(count + s, 1);
    if (ok && found == 3) found = s;
  }
  if (axis < 0) { if (found == 3) atomicAdd(count + 3, 1); return; }
  found = axis;
  const int a = found == 0 ? 1 : 0, b = found == 2 ? 1 : 2, ab = (a == 0 && b == 1) ? 3 : ((a == 0 && b == 2) ? 4 : 5);
  double *o = sw + (size_t)e * GEO_NSWEPT;
  o[0] = g[a]; o[1] = g[b]; o[2] = g[ab];
  o[3] = g[7 + a]; o[4] = g[7 + b]; o[5] = g[7 + ab];
  const double zs = g[14 + found];
  o[6] = found == 1 ? -zs : zs;
  o[7] = 1. / zs;
}
hipError_t launch_geo_swept(const double *geo, double *sw, int nelem, int *count, int axis, hipStream_t s) {
  if (nelem <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_geo_swept, dim3((unsigned)((nelem + 255) / 256)), dim3(256), 0, s, geo, sw, nelem, count, axis);
  return hipGetLastError();
}

template <int Q>
static hipError_t setup_geo_t(const BasisTables &t, const SetupGeoArgs &a, hipStream_t s) {
  using G = Geom<Q>;
  if (a.nelem <= 0) return hipSuccess;
  hipLaunchKernelGGL((k_setup_geo<Q>), dim3((a.nelem + G::EPB - 1) / G::EPB), dim3(G::BLOCK), 0, s, t, a);
  return hipGetLastError();
}
hipError_t launch_setup_geo(int Q, const BasisTables &t, const SetupGeoArgs &a, hipStream_t s,
                            const char **name) {
#define CPS_SG(Qv) case Qv: *name = "setup_geo<Q=" #Qv ">"; return setup_geo_t<Qv>(t, a, s);
  switch (Q) { CPS_SG(2) CPS_SG(3) CPS_SG(4) CPS_SG(5) CPS_SG(6) CPS_SG(7) CPS_SG(8) }
  return hipErrorInvalidValue;
}

// ===========================================================================
// p-multigrid transfer (setuplibceed.c:847-862; matops.c:115-203), round 5: pencil passes, OWNER form.
//
// A prolonged H1 field is single-valued at a fine node shared by several elements (the tensor-product interpolant on a
// face sees that face's coarse nodes only), so sum_e contribution_e / multiplicity (matops.c:149) IS any one element's
// contribution, to rounding.  Every fine node therefore has ONE owning element (the first that holds it, in element
// order: TransferArgs::own_f):
//   PROLONG : coarse gather -> interp Pc -> Pf in three pencil passes -> each element STORES the fine nodes it owns.
//             No fine E-vector, no scatter-add, no k_assemble, no multVec read on one rank.
//   RESTRICT: each element GATHERS the fine nodes it owns (the others read as zero) -> interp^T -> coarse E-vector
//             -> launch_assemble() (27 nodes per element at Pc = 3: small, and bit-reproducible) -- exactly the transpose.
// The per-dof weight w = (fine-side scale) x (local multiplicity) is 1 where the scale is 1 / multiplicity (one rank); it is
// read (w_f) only when the host found an entry that differs: the interface nodes of an element partition, whose scale holds
// the multiplicity over ALL ranks, or the extension-free form without a scale (plain libCEED semantics: w = multiplicity).
//
// One wave64 = one workgroup owns XferGeom::E elements.  In a pass a lane owns one line of an element along the contraction
// direction, one component; tables are wave-uniform (kernarg segment -> SGPR operands).  LDS arrays are laid out so that the
// lane-fastest index of the pass that reads them is contiguous: U0 [kc][jc][ic][c], U1 [kc][jc][if][c], U2 [kc][jf][if][c].
// ===========================================================================
constexpr uint32_t XFER_SKIP = 0xFFFFFFFFu;    // own_f entry of a fine node another element owns
#ifndef CPS_XFER_E5
#define CPS_XFER_E5 2
#endif
constexpr int xfer_group_elems(int PF) { return PF <= 3 ? 4 : (PF == 4 ? 4 : (PF == 5 ? CPS_XFER_E5 : 1)); }
template <int PC, int PF> struct XferGeom {
  static constexpr int C3 = PC * PC * PC, F2 = PF * PF, F3 = PF * PF * PF;
  static constexpr int E = xfer_group_elems(PF);
  static constexpr int N0 = 3 * C3, N1 = 3 * PC * PC * PF, N2 = 3 * PC * F2;   // doubles per element of U0, U1, U2
  static constexpr int NI = E * PC * PC * 3, NJ = E * PC * PF * 3, NK = E * F2 * 3;   // pencils of the i-, j-, k-pass
  static constexpr int KR = (NK + 63) / 64;                                    // rounds of the k-pass
};

template <int PC, int PF, bool PROLONG, bool WEIGHTED>
__global__ __launch_bounds__(64) void k_transfer(const BasisTables tab, const TransferArgs a) {
  using G = XferGeom<PC, PF>;
  constexpr int C3 = G::C3, F2 = G::F2, F3 = G::F3, E = G::E, N0 = G::N0, N1 = G::N1, N2 = G::N2, KR = G::KR;
  constexpr int SR = (E * N0 + 63) / 64;           // rounds of the coarse-side staging (a lane per coarse value)
  // U0 (the coarse values: read by the first pass of a prolongation, written by the last of a restriction) shares its storage with U2
  // (written by the j-pass / read by the j^T pass: never live together) -- 5.8 instead of 7.1 KB per wave at (3, 5): 27 instead of 22 waves per CU
  static_assert(N0 <= N2, "the coarse slab fits the widest intermediate");
  __shared__ double U1[E * N1], U2[E * N2];
  double *const U0 = U2;
  const int lane = threadIdx.x;
  // XCD-aware: the groups are cut into 8 contiguous chunks; block b serves chunk b % 8 (blocks b and b + 8 share an XCD under the
  // round-robin placement), so the elements that share coarse and fine nodes meet in one L2 (prolong p2 -> p4 at 99 000 hexes:
  // 175 -> 82 MB fetched).  One group per workgroup, NOT a persistent loop: a wave that ends never waits for its stores, a wave
  // that goes on to a next group does (its next loads count behind them in vmcnt) -- measured 47 -> 69 us for that prolongation.
  const int ngroups = (a.nelem + E - 1) / E, chunk = (ngroups + 7) / 8;
  const int grp = (int)(blockIdx.x % 8) * chunk + (int)(blockIdx.x / 8);
  if (grp >= min(ngroups, (int)(blockIdx.x % 8 + 1) * chunk)) return;
  // B[f][c] = tab.interp[f * PC + c]: value of coarse basis function c at fine node f (GLL points of the fine level).
  // Loads are written as straight-line rounds (no data-dependent control flow around them: all are in flight together); a lane
  // without work reads a valid entry of its group and discards it.
  auto load_own = [&](int g, uint32_t (&o)[KR][PF]) {      // the k-pass columns of this lane: the fine nodes they own
    const int e0 = g * E, ne = min(E, a.nelem - e0);
#pragma unroll
    for (int r = 0; r < KR; r++) {
      const int t = lane + 64 * r, el = t / (3 * F2), n2 = (t % (3 * F2)) / 3;
      const bool live = t < ne * 3 * F2;
#pragma unroll
      for (int k = 0; k < PF; k++) {
        const uint32_t v = a.own_f[(size_t)e0 * F3 + (live ? el * F3 + k * F2 + n2 : 0)];
        o[r][k] = live ? v : XFER_SKIP;
      }
    }
  };
  auto load_offc = [&](int g, uint32_t (&o)[SR]) {
    const int e0 = g * E, ne = min(E, a.nelem - e0);
#pragma unroll
    for (int r = 0; r < SR; r++) {
      const int t = lane + 64 * r;
      o[r] = a.off_c[(size_t)e0 * C3 + (t < ne * N0 ? t / 3 : 0)];
    }
  };
  uint32_t own[KR][PF], offc[SR];
  load_own(grp, own);
  load_offc(grp, offc);
  {
    const int e0 = grp * E, ne = min(E, a.nelem - e0);
    if constexpr (PROLONG) {
      double xin[SR];
#pragma unroll
      for (int r = 0; r < SR; r++) xin[r] = a.x[(offc[r] & OFF_MASK) + (lane + 64 * r) % 3];
      // ApplyAdd (the V-cycle's correction added in place): the old values of the owned nodes are requested NOW, behind the
      // owner list that has just landed, so that their latency passes under the three passes instead of in front of the stores
      double yold[KR][PF];
      if (a.add) {
#pragma unroll
        for (int r = 0; r < KR; r++) {
          const int c = ((lane + 64 * r) % (3 * F2)) % 3;
#pragma unroll
          for (int f = 0; f < PF; f++) yold[r][f] = own[r][f] == XFER_SKIP ? 0. : a.y[(own[r][f] & OFF_MASK) + c];
        }
      }
#pragma unroll
      for (int r = 0; r < SR; r++) {
        const int t = lane + 64 * r, c = t % 3;
        const bool dead = a.mask_c && ((offc[r] >> (OFF_FLAG_SHIFT + c)) & 1u);
        if (t < ne * N0) U0[t] = dead ? 0. : xin[r];
      }
      __syncthreads();
      for (int t = lane; t < G::NI; t += 64) {          // i: U0[kc][jc][ic][c] -> U1[kc][jc][if][c]
        const int el = t / (PC * PC * 3), r = t % (PC * PC * 3), m = r / 3, c = r % 3;
        double u[PC];
#pragma unroll
        for (int i = 0; i < PC; i++) u[i] = U0[el * N0 + (m * PC + i) * 3 + c];
#pragma unroll
        for (int f = 0; f < PF; f++) {
          double s = 0.;
#pragma unroll
          for (int i = 0; i < PC; i++) s += tab.interp[f * PC + i] * u[i];
          U1[el * N1 + (m * PF + f) * 3 + c] = s;
        }
      }
      __syncthreads();
      for (int t = lane; t < G::NJ; t += 64) {          // j: U1[kc][jc][if][c] -> U2[kc][jf][if][c]
        const int el = t / (PC * PF * 3), r = t % (PC * PF * 3), kc = r / (PF * 3), ic = r % (PF * 3);
        double u[PC];
#pragma unroll
        for (int j = 0; j < PC; j++) u[j] = U1[el * N1 + (kc * PC + j) * PF * 3 + ic];
#pragma unroll
        for (int f = 0; f < PF; f++) {
          double s = 0.;
#pragma unroll
          for (int j = 0; j < PC; j++) s += tab.interp[f * PC + j] * u[j];
          U2[el * N2 + (kc * PF + f) * PF * 3 + ic] = s;
        }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < KR; r++) {                    // k: U2[kc][jf][if][c] -> the owned fine nodes of the column, stored
        const int t = lane + 64 * r, el = t / (3 * F2), rem = t % (3 * F2), c = rem % 3;
        if (t >= ne * 3 * F2) break;
        double u[PC];
#pragma unroll
        for (int k = 0; k < PC; k++) u[k] = U2[el * N2 + k * F2 * 3 + rem];
#pragma unroll
        for (int f = 0; f < PF; f++) {
          const uint32_t off = own[r][f];
          if (off == XFER_SKIP) continue;
          double s = 0.;
#pragma unroll
          for (int k = 0; k < PC; k++) s += tab.interp[f * PC + k] * u[k];
          double *dst = a.y + (off & OFF_MASK) + c;
          if constexpr (WEIGHTED) s *= a.w_f[(off & OFF_MASK) + c];
          if (a.mask_f && ((off >> (OFF_FLAG_SHIFT + c)) & 1u)) s = 0.;
          *dst = a.add ? yold[r][f] + s : s;
        }
      }
    } else {
      // all gathers of the wave's columns are issued together (lanes without an owned node read entry 0 and discard it)
      double xv[KR][PF];
#pragma unroll
      for (int r = 0; r < KR; r++) {
        const int c = ((lane + 64 * r) % (3 * F2)) % 3;
#pragma unroll
        for (int f = 0; f < PF; f++) {
          const uint32_t off = own[r][f];
          const uint32_t idx = off == XFER_SKIP ? 0u : (off & OFF_MASK) + c;
          xv[r][f] = a.x[idx];
          if constexpr (WEIGHTED) xv[r][f] *= a.w_f[idx];
        }
      }
#pragma unroll
      for (int r = 0; r < KR; r++) {                    // k^T: the owned fine nodes of the column -> U2[kc][jf][if][c]
        const int t = lane + 64 * r, el = t / (3 * F2), rem = t % (3 * F2), c = rem % 3;
        double v[PF];
#pragma unroll
        for (int f = 0; f < PF; f++) {
          const uint32_t off = own[r][f];
          const bool dead = off == XFER_SKIP || (a.mask_f && ((off >> (OFF_FLAG_SHIFT + c)) & 1u));
          v[f] = dead ? 0. : xv[r][f];
        }
        if (t < E * 3 * F2) {
#pragma unroll
          for (int k = 0; k < PC; k++) {
            double s = 0.;
#pragma unroll
            for (int f = 0; f < PF; f++) s += tab.interp[f * PC + k] * v[f];
            U2[el * N2 + k * F2 * 3 + rem] = s;
          }
        }
      }
      __syncthreads();
      for (int t = lane; t < G::NJ; t += 64) {          // j^T: U2[kc][jf][if][c] -> U1[kc][jc][if][c]
        const int el = t / (PC * PF * 3), r = t % (PC * PF * 3), kc = r / (PF * 3), ic = r % (PF * 3);
        double v[PF];
#pragma unroll
        for (int f = 0; f < PF; f++) v[f] = U2[el * N2 + (kc * PF + f) * PF * 3 + ic];
#pragma unroll
        for (int j = 0; j < PC; j++) {
          double s = 0.;
#pragma unroll
          for (int f = 0; f < PF; f++) s += tab.interp[f * PC + j] * v[f];
          U1[el * N1 + (kc * PC + j) * PF * 3 + ic] = s;
        }
      }
      __syncthreads();
      for (int t = lane; t < G::NI; t += 64) {          // i^T: U1[kc][jc][if][c] -> U0[kc][jc][ic][c]
        const int el = t / (PC * PC * 3), r = t % (PC * PC * 3), m = r / 3, c = r % 3;
        double v[PF];
#pragma unroll
        for (int f = 0; f < PF; f++) v[f] = U1[el * N1 + (m * PF + f) * 3 + c];
#pragma unroll
        for (int i = 0; i < PC; i++) {
          double s = 0.;
#pragma unroll
          for (int f = 0; f < PF; f++) s += tab.interp[f * PC + i] * v[f];
          U0[el * N0 + (m * PC + i) * 3 + c] = s;
        }
      }
      __syncthreads();
      // the group's block of the coarse E-vector [elem][node][3] is contiguous: whole-line stores; masked entries travel as zeros
#pragma unroll
      for (int r = 0; r < SR; r++) {
        const int t = lane + 64 * r, c = t % 3;
        const bool dead = a.mask_c && ((offc[r] >> (OFF_FLAG_SHIFT + c)) & 1u);
        if (t < ne * N0) a.evec[(size_t)e0 * N0 + t] = dead ? 0. : U0[t];
      }
    }
  }
}
template <int PC, int PF>
static hipError_t transfer_t(bool prolong, const BasisTables &t, const TransferArgs &a, hipStream_t s) {
  using G = XferGeom<PC, PF>;
  if (a.nelem <= 0) return hipSuccess;
  const int ngroups = (a.nelem + G::E - 1) / G::E;
  const dim3 grid(8 * ((ngroups + 7) / 8)), block(64);
  const bool w = a.w_f != nullptr;
  if (prolong) { if (w) hipLaunchKernelGGL((k_transfer<PC, PF, true, true>), grid, block, 0, s, t, a); else hipLaunchKernelGGL((k_transfer<PC, PF, true, false>), grid, block, 0, s, t, a); }
  else { if (w) hipLaunchKernelGGL((k_transfer<PC, PF, false, true>), grid, block, 0, s, t, a); else hipLaunchKernelGGL((k_transfer<PC, PF, false, false>), grid, block, 0, s, t, a); }
  return hipGetLastError();
}
hipError_t launch_transfer(int Pc, int Pf, bool prolong, const BasisTables &t, const TransferArgs &a,
                           hipStream_t s, const char **name) {
#define CPS_TR(C, F)                                                                  \
  if (Pc == C && Pf == F) {                                                           \
    *name = prolong ? "prolong<Pc=" #C ",Pf=" #F ">" : "restrict<Pc=" #C ",Pf=" #F ">"; \
    return transfer_t<C, F>(prolong, t, a, s);                                        \
  }
  // adjacent level pairs of the logarithmic (1,2,4,..,p) and uniform ladders up to p = 7
  CPS_TR(2, 3) CPS_TR(3, 4) CPS_TR(3, 5) CPS_TR(4, 5) CPS_TR(5, 6) CPS_TR(5, 7) CPS_TR(6, 7) CPS_TR(5, 8)
  CPS_TR(7, 8) CPS_TR(2, 4) CPS_TR(2, 5)
  return hipErrorInvalidValue;
}
// w[i] = (local multiplicity, as counted into w by launch_multiplicity) * (scale ? scale[i] : 1); *n_not_unit counts the
// covered entries whose weight is not 1 (to 4 ulp: (1 / m) m rounds to 1 for the multiplicities of a mesh, not for every integer)
__global__ void k_xfer_weights(double *w, const double *scale, size_t n, int *n_not_unit) {
  int bad = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double m = w[i], v = scale ? m * scale[i] : m;
    w[i] = v;
    if (m != 0. && fabs(v - 1.) > 1e-15) bad = 1;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicAdd(n_not_unit, 1);
}
hipError_t launch_transfer_weights(double *w, const double *scale, size_t n, int *n_not_unit, hipStream_t s) {
  if (!n) return hipSuccess;
  size_t b = (n + 255) / 256;
  hipLaunchKernelGGL(k_xfer_weights, dim3((unsigned)(b > 2048 ? 2048 : b)), dim3(256), 0, s, w, scale, n, n_not_unit);
  return hipGetLastError();
}

// ===========================================================================
// Diagonal of B^T D B (matops.c:227).  D's (d,c),(d',c) entries come
// from the Jacobian physics applied to unit reference gradients.
// ===========================================================================
// diag_c(n) = sum_q sum_{d,d2} g_d(n,q) D^c_{d d2}(q) g_d2(n,q): the tangent D probed with nine unit gradients per point.
// (The unfactorised first version -- P^3 Q^3 60 FMAs per element, 5x slower -- left the tree in round 3.)
// Sum-factorised: g_d is a product of
// 1-D factors, so each of the 18 tensors S^c_pair(q) (pair = (d,d2), d <= d2; off-diagonal pairs hold D_{d d2} + D_{d2 d})
// is contracted direction by direction with the PRODUCT tables BB, BG, GG (table_x(i,a) = X_d(i,a) X_d2(i,a), X = G in its
// own direction, B otherwise): 18 * (P Q^2 + P^2 Q + P^3) * Q FMAs per element instead of P^3 * Q^3 * 60 (25x fewer at
// P = Q = 5); what remains is the nine physics evaluations per point that probe D.
template <int P, int Q, int QF>
__global__ __launch_bounds__(Geom<Q>::TPE) void k_diag_sf(const BasisTables tab, const DiagArgs a) {
  using G = Geom<Q>;
  constexpr int Q3 = G::Q3, P3 = P * P * P, TPE = G::TPE, NT = 18;
  constexpr int PQQ = P * Q * Q, PPQ = P * P * Q, S0 = Q3 > PPQ ? Q3 : PPQ;
  constexpr bool ST_IN = QFTraits<QF>::state_in;
  extern __shared__ double dyn[];
  double *sT = dyn;                  // [3][Q * P]: BB, BG, GG
  double *s0 = sT + 3 * Q * P;       // [NT][Q3] the tensors, later [NT][P * P * Q]
  double *s1 = s0 + NT * S0;         // [NT][P * Q * Q]
  const int q = threadIdx.x, e = blockIdx.x;
  for (int i = q; i < Q * P; i += TPE) {
    const double bb = tab.interp[i], gg = tab.grad[i];
    sT[i] = bb * bb; sT[Q * P + i] = bb * gg; sT[2 * Q * P + i] = gg * gg;
  }
  if (q < Q3) {
    double qd[10], st[9], dv[9], sto[9], ug[9], D[3][3][3];   // D[c][dout][din]
    const double *qp = a.qdata + (size_t)e * 10 * Q3 + q;
#pragma unroll
    for (int c = 0; c < 10; c++) qd[c] = qp[c * Q3];
    if constexpr (ST_IN) {
      const double *sp = a.state_in + (size_t)e * 9 * Q3 + q;
#pragma unroll
      for (int c = 0; c < 9; c++) st[c] = sp[c * Q3];
    }
#pragma unroll
    for (int din = 0; din < 3; din++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int s = 0; s < 9; s++) ug[s] = (s == din * 3 + c) ? 1. : 0.;
        qf_point<QF>(Phys{a.nu, a.E, a.lambda, a.TwoMu}, ug, qd, st, dv, sto);
#pragma unroll
        for (int dout = 0; dout < 3; dout++) D[c][dout][din] = dv[dout * 3 + c];
      }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      s0[(c * 6 + 0) * Q3 + q] = D[c][0][0];
      s0[(c * 6 + 1) * Q3 + q] = D[c][1][1];
      s0[(c * 6 + 2) * Q3 + q] = D[c][2][2];
      s0[(c * 6 + 3) * Q3 + q] = D[c][0][1] + D[c][1][0];
      s0[(c * 6 + 4) * Q3 + q] = D[c][0][2] + D[c][2][0];
      s0[(c * 6 + 5) * Q3 + q] = D[c][1][2] + D[c][2][1];
    }
  }
  __syncthreads();
  // table kind of pair p in direction dir: (d == dir) + (d2 == dir)  (0 BB, 1 BG, 2 GG)
  auto kind = [](int p, int dir) {
    const int d = p < 3 ? p : (p == 5 ? 1 : 0), d2 = p < 3 ? p : (p == 3 ? 1 : 2);
    return (d == dir) + (d2 == dir);
  };
  // x: U1[t][k][j][a] = sum_i T(i,a) S[t][k][j][i]
  for (int o = q; o < NT * PQQ; o += TPE) {
    const int t = o / PQQ, r = o % PQQ, aa = r % P, kj = r / P;
    const double *T = sT + kind(t % 6, 0) * Q * P, *src = s0 + t * Q3 + kj * Q;
    double v = 0.;
#pragma unroll
    for (int i = 0; i < Q; i++) v += T[i * P + aa] * src[i];
    s1[o] = v;
  }
  __syncthreads();
  // y: U2[t][k][b][a] = sum_j T(j,b) U1[t][k][j][a]
  for (int o = q; o < NT * PPQ; o += TPE) {
    const int t = o / PPQ, r = o % PPQ, aa = r % P, bb = (r / P) % P, k = r / (P * P);
    const double *T = sT + kind(t % 6, 1) * Q * P, *src = s1 + t * PQQ + k * Q * P + aa;
    double v = 0.;
#pragma unroll
    for (int j = 0; j < Q; j++) v += T[j * P + bb] * src[j * P];
    s0[o] = v;
  }
  __syncthreads();
  // z, and the sum over the pairs: one node per thread
  if (q < P3) {
    const int ab = q % (P * P), nc = q / (P * P);
    double acc[3] = {0., 0., 0.};
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int p = 0; p < 6; p++) {
        const double *T = sT + kind(p, 2) * Q * P, *src = s0 + (c * 6 + p) * PPQ + ab;
        double v = 0.;
#pragma unroll
        for (int k = 0; k < Q; k++) v += T[k * P + nc] * src[k * P * P];
        acc[c] += v;
      }
    const uint32_t off = a.offsets[(size_t)e * P3 + q];
    const uint32_t fl = a.mask_out ? (off >> OFF_FLAG_SHIFT) : 0u;
#pragma unroll
    for (int c = 0; c < 3; c++) a.evec[((size_t)e * P3 + q) * 3 + c] = ((fl >> c) & 1u) ? 0. : acc[c];  // summed by launch_assemble()
  }
}
template <int P, int Q, int QF>
static hipError_t diag_t(const BasisTables &t, const DiagArgs &a, hipStream_t s) {
  using G = Geom<Q>;
  if (a.nelem <= 0) return hipSuccess;
  constexpr int PQQ = P * Q * Q, PPQ = P * P * Q, S0 = G::Q3 > PPQ ? G::Q3 : PPQ;
  const size_t lds = sizeof(double) * (3 * Q * P + 18 * (S0 + PQQ));
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t er = hipFuncSetAttribute((const void *)k_diag_sf<P, Q, QF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (er != hipSuccess) return er;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_diag_sf<P, Q, QF>), dim3(a.nelem), dim3(G::TPE), lds, s, t, a);
  return hipGetLastError();
}
hipError_t launch_diag(int P, int Q, int qf, const BasisTables &t, const DiagArgs &a, hipStream_t s,
                       const char **name) {
#define CPS_DG(Pv, Qv, QFv, nm)                                   \
  if (P == Pv && Q == Qv && qf == QFv) {                          \
    *name = "diag<P=" #Pv ",Q=" #Qv "," nm ">";                   \
    return diag_t<Pv, Qv, QFv>(t, a, s);                          \
  }
#define CPS_DG3(Pv, Qv) CPS_DG(Pv, Qv, QF_LINELAS, "LinElas") CPS_DG(Pv, Qv, QF_HYPERSS_DF, "HyperSSdF") \
  CPS_DG(Pv, Qv, QF_HYPERFS_DF, "HyperFSdF")
  CPS_DG3(2, 2) CPS_DG3(2, 3) CPS_DG3(3, 3) CPS_DG3(2, 4) CPS_DG3(3, 4) CPS_DG3(4, 4)
  CPS_DG3(2, 5) CPS_DG3(3, 5) CPS_DG3(4, 5) CPS_DG3(5, 5) CPS_DG3(2, 7) CPS_DG3(3, 7) CPS_DG3(5, 7) CPS_DG3(7, 7)
  // degrees 5 and 7 (logarithmic ladders 1, 2, 4, p) and the uniform ladders of degrees 5 and 6
  CPS_DG3(2, 6) CPS_DG3(3, 6) CPS_DG3(4, 6) CPS_DG3(5, 6) CPS_DG3(6, 6) CPS_DG3(4, 7) CPS_DG3(6, 7)
  CPS_DG3(2, 8) CPS_DG3(3, 8) CPS_DG3(4, 8) CPS_DG3(5, 8) CPS_DG3(6, 8) CPS_DG3(7, 8) CPS_DG3(8, 8)
  return hipErrorInvalidValue;
}

// One wave that idles for `ticks` of the constant 100 MHz counter and reports how many SHADER clock cycles went by: the clock the
// chip runs at under whatever load the other streams put on it (a slow box and a slow build are then told apart: bench.py).
__global__ void k_clock_probe(long long *out, long long ticks) {
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64(), c0 = clock64();
  long long t1 = t0;
  while (t1 - t0 < ticks) { __builtin_amdgcn_s_sleep(32); t1 = wall_clock64(); }
  out[0] = clock64() - c0; out[1] = t1 - t0;
}
hipError_t launch_clock_probe(long long *out, int spin_us, hipStream_t s) {
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, out, (long long)spin_us * 100);
  return hipGetLastError();
}

int device_cu_count() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    ncu = prop.multiProcessorCount;
  }
  return ncu;
}

// ===========================================================================
// Fused-operator dispatch over the per-Q objects (kernels_fused_inst.hip).
// ===========================================================================
#define CPS_DECL_QP(Qv, Pt) hipError_t launch_fused_grad_q##Qv##p##Pt(int, int, const BasisTables &, const FusedGradArgs &, hipStream_t, const char **);
CPS_DECL_QP(2, 0) CPS_DECL_QP(3, 0) CPS_DECL_QP(4, 0) CPS_DECL_QP(5, 0) CPS_DECL_QP(6, 0) CPS_DECL_QP(7, 0) CPS_DECL_QP(7, 1)
CPS_DECL_QP(8, 0) CPS_DECL_QP(8, 1) CPS_DECL_QP(8, 2) CPS_DECL_QP(8, 3)
hipError_t launch_fused_grad(int P, int Q, int qf, const BasisTables &t, const FusedGradArgs &a,
                             hipStream_t s, const char **name) {
  static_assert(pencil_inst_parts(7) == 2 && pencil_inst_parts(8) == 4 && pencil_inst_parts(6) == 1, "the objects declared above");
  const int part = (Q >= 2 && Q <= MAXN1D && P <= Q) ? (Q - P) % pencil_inst_parts(Q) : 0;
  switch (Q) {
    case 2: return launch_fused_grad_q2p0(P, qf, t, a, s, name);
    case 3: return launch_fused_grad_q3p0(P, qf, t, a, s, name);
    case 4: return launch_fused_grad_q4p0(P, qf, t, a, s, name);
    case 5: return launch_fused_grad_q5p0(P, qf, t, a, s, name);
    case 6: return launch_fused_grad_q6p0(P, qf, t, a, s, name);
    case 7: return part == 0 ? launch_fused_grad_q7p0(P, qf, t, a, s, name) : launch_fused_grad_q7p1(P, qf, t, a, s, name);
    case 8: return part == 0 ? launch_fused_grad_q8p0(P, qf, t, a, s, name) : (part == 1 ? launch_fused_grad_q8p1(P, qf, t, a, s, name) :
                   (part == 2 ? launch_fused_grad_q8p2(P, qf, t, a, s, name) : launch_fused_grad_q8p3(P, qf, t, a, s, name)));
  }
  return hipErrorInvalidValue;
}

// ===========================================================================
// Vector and restriction utilities (HBM-bound, grid-stride, 2048-block cap).
// ===========================================================================
static inline dim3 stream_grid(size_t n) {
  size_t b = (n + 255) / 256;
  return dim3((unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b)));
}
__global__ void k_set_value(double *v, size_t n, double val) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) v[i] = val;
}
__global__ void k_reciprocal(double *v, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (fabs(v[i]) > 1e-300) v[i] = 1. / v[i];
}
__global__ void k_pointwise_mult(double *w, const double *x, const double *y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) w[i] = x[i] * y[i];
}
__global__ void k_axpby(double *y, double a, const double *x, double b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = a * x[i] + (b == 0. ? 0. : b * y[i]);
}
// One dof of a Chebyshev step: r = rbase - t (has_t), d = c1 dinv r + c2 d, x = d or x + d.  ONE definition for the stand-alone
// update (k_cheb_update) and the epilogue of the fused apply (k_assemble_epi): explicit operation order, no contraction left to
// the compiler, so that both forms give the same bits.
CPS_DEV void cheb_dof_regs(double rbase, bool has_t, double ti, bool store_r, size_t i, double dinv_i, double d_i, double x_i, double *x, double *d,
                           double *r, double c1, double c2, int assign_x) {
#pragma clang fp contract(off)
  const double ri = has_t ? rbase - ti : rbase;
  if (store_r) r[i] = ri;
  double di = (c1 * dinv_i) * ri;
  if (c2 != 0.) di = __builtin_fma(c2, d_i, di);
  d[i] = di;
  x[i] = assign_x ? di : x_i + di;
}
CPS_DEV void cheb_dof(double rbase, bool has_t, double ti, bool store_r, size_t i, double *x, double *d, double *r, const double *dinv,
                      double c1, double c2, int assign_x) {
  cheb_dof_regs(rbase, has_t, ti, store_r, i, dinv[i], c2 != 0. ? d[i] : 0., assign_x ? 0. : x[i], x, d, r, c1, c2, assign_x);
}
__global__ void k_cheb_update(double *x, double *d, double *r, const double *r0, const double *t, const double *dinv, double c1,
                              double c2, int assign_x, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    // r0: the right-hand side of a first step (r = b - t without a copy of b)
    cheb_dof(r0 ? r0[i] : r[i], t != nullptr, t ? t[i] : 0., (t || r0) && r, i, x, d, r, dinv, c1, c2, assign_x);
}
__global__ void k_masked_copy(double *dst, const double *src, const unsigned char *mask, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = mask[i] ? 0. : src[i];
}
// E-layout [e][c][n]
__global__ void k_rstr(const uint32_t *off, size_t total, int elemsize, int ncomp, int compstride,
                       const double *src, double *dst, int mode) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i % elemsize, ec = i / elemsize, c = ec % ncomp, e = ec / ncomp;
    const size_t li = (size_t)(off[e * elemsize + n] & OFF_MASK) + c * (size_t)compstride;
    if (mode == 0) dst[i] = src[li];
    else if (mode == 1) atomic_add_f64(dst + li, src[i]);
    else atomic_add_f64(dst + li, 1.0);
  }
}
// One lane per L-node.  The E-vector is interlaced [elem][node][3] like the L-vector: a contributor is
// 24 contiguous bytes, consecutive lanes (consecutively numbered nodes of one element) read and write
// consecutive 24-byte rows, so the three strided 8-byte accesses of a wave cover whole cache lines.
// Workgroups [nb_rows, gridDim.x) -- present only with `un.n` > 0 -- add the arrivals of a halo exchange instead
// (HaloUnpackArgs: different entries of y than any row of this launch).
__global__ void k_assemble(const uint32_t *rowptr, const uint32_t *cols, const uint32_t *node_off,
                           const unsigned char *flags, const double *evec, double *y, int nnodes, int add, int nb_rows,
                           const HaloUnpackArgs un, const HaloPackFold pk) {
#ifdef CPS_ASM_PRIO   // (tuning hook) wave priority of the row sums beside a fused kernel
  __builtin_amdgcn_s_setprio(CPS_ASM_PRIO);
#endif
  if ((int)blockIdx.x >= nb_rows) {
    for (int u = ((int)blockIdx.x - nb_rows) * blockDim.x + threadIdx.x; u < un.n; u += ((int)gridDim.x - nb_rows) * blockDim.x) {
      double v = y[un.dst[u]];
      for (uint32_t k = un.ptr[u]; k < un.ptr[u + 1]; k++) v += un.recv[un.slot[k]];
      y[un.dst[u]] = v;
    }
    return;
  }
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nnodes; r += nb_rows * blockDim.x) {
    const uint32_t k0 = rowptr[r], k1 = rowptr[r + 1];
    double a0 = 0., a1 = 0., a2 = 0.;
    // four contributors per trip: the index loads, then the twelve value loads, are issued together (the
    // chain rowptr -> cols -> E-vector is latency bound otherwise); lanes with fewer contributors re-read
    // their last one and discard it.  Sums are still formed in contributor (= element) order.
    for (uint32_t k = k0; k < k1; k += 4) {
      uint32_t c[4];
#pragma unroll
      for (int j = 0; j < 4; j++) c[j] = cols[k + j < k1 ? k + j : k1 - 1];
      double v[4][3];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const double *p = evec + (size_t)c[j] * 3;
        v[j][0] = p[0]; v[j][1] = p[1]; v[j][2] = p[2];
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (k + j < k1) { a0 += v[j][0]; a1 += v[j][1]; a2 += v[j][2]; }
    }
    const unsigned fl = flags ? flags[r] : 0u;
    double *dst = y + (node_off[r] & OFF_MASK);
    if (fl & 1u) a0 = 0.;
    if (fl & 2u) a1 = 0.;
    if (fl & 4u) a2 = 0.;
    if (add) { a0 += dst[0]; a1 += dst[1]; a2 += dst[2]; }
    dst[0] = a0; dst[1] = a1; dst[2] = a2;
    if (pk.ptr)   // interface node: its finished sums go straight into the exchange's send buffer (no pack launch)
      for (uint32_t k = pk.ptr[r]; k < pk.ptr[r + 1]; k++) {
        const uint32_t e = pk.slot[k], cmp = e >> 30;
        pk.send[e & 0x3FFFFFFFu] = cmp == 0 ? a0 : (cmp == 1 ? a1 : a2);
      }
  }
}

// k_assemble with an EPILOGUE instead of the store of y (round 5): the operator's output t = A v is consumed where it is formed.
//   EPI_CHEB : one step of the Chebyshev smoother (elasticity.c:539-552) -- r = (r0 or r) - t, d = c1 dinv r + c2 d, x = d or x + d
//   EPI_RESID: w = b - t (the residual between the smoother and the restriction of a V-cycle)
// Rows [0, nnodes): the shell nodes of the transpose map, summed in contributor order exactly as k_assemble does.  Workgroups
// [nb_rows, gridDim.x): the dofs of the ELEMENT-INTERIOR nodes (int_off: their node offsets, elements in order), whose t the fused
// kernel stored into `t` itself.  t at the shell nodes is never written.  The apply's input may be d (or x) itself: a row is
// summed only after the last element that holds its node has finished (pipelined form: rows belong to the segment of their LAST
// contributor), and no later element gathers it.
__global__ void k_assemble_epi(const uint32_t *rowptr, const uint32_t *cols, const uint32_t *node_off, const unsigned char *flags,
                               const double *evec, int nnodes, int nb_rows, const EpilogueArgs ep) {
  if ((int)blockIdx.x >= nb_rows) {
    const size_t n = (size_t)ep.n_int * 3;
    for (size_t u = ((size_t)blockIdx.x - nb_rows) * blockDim.x + threadIdx.x; u < n; u += ((size_t)gridDim.x - nb_rows) * blockDim.x) {
      const size_t i = (size_t)(ep.int_off[u / 3] & OFF_MASK) + u % 3;
      const double ti = ep.t[i];
      if (ep.kind == EPI_CHEB) cheb_dof(ep.r0 ? ep.r0[i] : ep.r[i], true, ti, ep.r != nullptr, i, ep.x, ep.d, ep.r, ep.dinv, ep.c1, ep.c2, ep.assign_x);
      else ep.w[i] = ep.b[i] - ti;
    }
    return;
  }
  // The SUM is formed a lane per row (node), as k_assemble forms it -- one walk of rowptr / cols per node, the same additions in the
  // same order: same bits.  The CONSUMER then runs a lane per DOF: the 64 rows of a wave are 192 dofs = three rounds of 64 lanes,
  // dof j = 64 q + lane belongs to the row of lane j / 3, component j % 3 (sums and node offsets fetched from that lane by
  // ds_bpermute), so every stream of the epilogue -- r, dinv, d, x, b, w -- is read and written 8 bytes per lane, contiguous over the
  // wave where the rows' nodes are numbered consecutively, instead of three 24-byte-strided accesses per lane (measured 164 -> 100 us
  // per launch over a 99 000-hex solve); the streams are requested BEFORE the dependent chain rowptr -> cols -> E-vector.
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)((nb_rows * blockDim.x) >> 6);
  for (int row0 = wave * 64; row0 < nnodes; row0 += nwaves * 64) {     // (wave-uniform trip count: every lane takes part in the shuffles)
    const int r = row0 + lane;
    const bool live = r < nnodes;
    const uint32_t off = live ? (node_off[r] & OFF_MASK) : 0u;
    size_t idx[3];
    bool ok[3];
    double s0[3], s1[3], s2[3], s3[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int j = 64 * q + lane, src = j / 3, c = j % 3;
      idx[q] = (size_t)__shfl(off, src, 64) + c;
      ok[q] = row0 + src < nnodes;
      s0[q] = s1[q] = s2[q] = s3[q] = 0.;
      if (ok[q]) {
        if (ep.kind == EPI_CHEB) {
          s0[q] = ep.r0 ? ep.r0[idx[q]] : ep.r[idx[q]]; s1[q] = ep.dinv[idx[q]];
          s2[q] = ep.c2 != 0. ? ep.d[idx[q]] : 0.; s3[q] = ep.assign_x ? 0. : ep.x[idx[q]];
        } else s0[q] = ep.b[idx[q]];
      }
    }
    double a0 = 0., a1 = 0., a2 = 0.;
    if (live) {
      const uint32_t k0 = rowptr[r], k1 = rowptr[r + 1];
      for (uint32_t k = k0; k < k1; k += 4) {
        uint32_t c[4];
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = cols[k + j < k1 ? k + j : k1 - 1];
        double v[4][3];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const double *p = evec + (size_t)c[j] * 3;
          v[j][0] = p[0]; v[j][1] = p[1]; v[j][2] = p[2];
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (k + j < k1) { a0 += v[j][0]; a1 += v[j][1]; a2 += v[j][2]; }
      }
      const unsigned fl = flags ? flags[r] : 0u;
      if (fl & 1u) a0 = 0.;
      if (fl & 2u) a1 = 0.;
      if (fl & 4u) a2 = 0.;
    }
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int j = 64 * q + lane, src = j / 3, c = j % 3;
      const double t0 = __shfl(a0, src, 64), t1 = __shfl(a1, src, 64), t2 = __shfl(a2, src, 64);
      const double ti = c == 0 ? t0 : (c == 1 ? t1 : t2);
      if (!ok[q]) continue;
      if (ep.kind == EPI_CHEB) cheb_dof_regs(s0[q], true, ti, ep.r != nullptr, idx[q], s1[q], s2[q], s3[q], ep.x, ep.d, ep.r, ep.c1, ep.c2, ep.assign_x);
      else ep.w[idx[q]] = s0[q] - ti;
    }
  }
}
hipError_t launch_assemble_epi(const uint32_t *rowptr, const uint32_t *cols, const uint32_t *node_off, const unsigned char *flags,
                               const double *evec, int nnodes, const EpilogueArgs &ep, hipStream_t s, int max_blocks) {
  if (nnodes <= 0 && ep.n_int <= 0) return hipSuccess;
  constexpr int AB = 256;
  unsigned nb_rows = (unsigned)((std::max(nnodes, 0) + AB - 1) / AB);
  if (max_blocks > 0 && nb_rows > (unsigned)max_blocks) nb_rows = (unsigned)max_blocks;
  unsigned nb_int = (unsigned)std::min<size_t>(((size_t)std::max(ep.n_int, 0) * 3 + AB - 1) / AB, 4096);
  if (max_blocks > 0 && nb_int > (unsigned)max_blocks) nb_int = (unsigned)max_blocks;
  hipLaunchKernelGGL(k_assemble_epi, dim3(nb_rows + nb_int), dim3(AB), 0, s, rowptr, cols, node_off, flags, evec, nnodes, (int)nb_rows, ep);
  return hipGetLastError();
}

__global__ void k_halo_pack(const uint32_t *idx, int n, const double *y, double *buf) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = y[idx[i]];
}
__global__ void k_halo_unpack_add(const HaloUnpackArgs un, double *y) {
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < un.n; u += gridDim.x * blockDim.x) {
    double v = y[un.dst[u]];
    for (uint32_t k = un.ptr[u]; k < un.ptr[u + 1]; k++) v += un.recv[un.slot[k]];   // neighbour-list order
    y[un.dst[u]] = v;
  }
}
hipError_t launch_halo_pack(const uint32_t *idx, int n, const double *y, double *buf, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)std::min((n + 255) / 256, 2048)), dim3(256), 0, s, idx, n, y, buf);
  return hipGetLastError();
}
hipError_t launch_halo_unpack_add(const HaloUnpackArgs &u, double *y, hipStream_t s) {
  if (u.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_halo_unpack_add, dim3((unsigned)std::min((u.n + 255) / 256, 2048)), dim3(256), 0, s, u, y);
  return hipGetLastError();
}

__global__ void k_dot(const double *x, const double *y, const double *w, size_t n, double *result) {
  double s = 0.;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s += (w ? w[i] : 1.) * x[i] * y[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  // per-block partial, summed in a fixed order by k_dot_final: the dot is reproducible run to run
  if (threadIdx.x == 0) result[1 + blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(256) void k_dot_final(double *result, int nparts, double *out) {
  __shared__ double sh[256];
  double s = 0.;
  for (int i = threadIdx.x; i < nparts; i += 256) s += result[1 + i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) { result[0] = sh[0]; if (out) *out = sh[0]; }
}
// scalars kept on the device (a Krylov recurrence without a host round trip per dot): s[dst] = scale * s[num] / s[den]
// (den < 0: no division); a non-positive denominator gives 0, which the host reads as "breakdown" afterwards
__global__ void k_scalar_div(double *s, int dst, int num, int den, double scale) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double d = den < 0 ? 1. : s[den];
    s[dst] = (den < 0 || d > 0.) ? scale * s[num] / d : 0.;
  }
}
// y = sa * (ia < 0 ? 1 : s[ia]) * x + sb * (ib < 0 ? 1 : s[ib]) * y
__global__ void k_axpby_dev(double *y, const double *s, int ia, double sa, const double *x, int ib, double sb, size_t n) {
  const double a = sa * (ia < 0 ? 1. : s[ia]), b = sb * (ib < 0 ? 1. : s[ib]);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = a * x[i] + b * y[i];
}

hipError_t launch_set_value(double *v, size_t n, double val, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_set_value, stream_grid(n), dim3(256), 0, s, v, n, val);
  return hipGetLastError();
}
hipError_t launch_reciprocal(double *v, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_reciprocal, stream_grid(n), dim3(256), 0, s, v, n);
  return hipGetLastError();
}
hipError_t launch_pointwise_mult(double *w, const double *x, const double *y, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_pointwise_mult, stream_grid(n), dim3(256), 0, s, w, x, y, n);
  return hipGetLastError();
}
__global__ void k_waxpby(double *w, double a, const double *x, double b, const double *y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) w[i] = a * x[i] + b * y[i];
}
hipError_t launch_waxpby(double *w, double a, const double *x, double b, const double *y, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_waxpby, stream_grid(n), dim3(256), 0, s, w, a, x, b, y, n);
  return hipGetLastError();
}
hipError_t launch_axpby(double *y, double a, const double *x, double b, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_axpby, stream_grid(n), dim3(256), 0, s, y, a, x, b, n);
  return hipGetLastError();
}
hipError_t launch_cheb_update(double *x, double *d, double *r, const double *r0, const double *t, const double *dinv, double c1, double c2,
                              int assign_x, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_cheb_update, stream_grid(n), dim3(256), 0, s, x, d, r, r0, t, dinv, c1, c2, assign_x, n);
  return hipGetLastError();
}
hipError_t launch_masked_copy(double *dst, const double *src, const unsigned char *mask, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_masked_copy, stream_grid(n), dim3(256), 0, s, dst, src, mask, n);
  return hipGetLastError();
}
static hipError_t rstr(const uint32_t *off, int nelem, int elemsize, int ncomp, int compstride,
                       const double *src, double *dst, int mode, hipStream_t s) {
  const size_t total = (size_t)nelem * elemsize * ncomp;
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(k_rstr, stream_grid(total), dim3(256), 0, s, off, total, elemsize, ncomp, compstride, src, dst, mode);
  return hipGetLastError();
}
hipError_t launch_rstr_gather(const uint32_t *off, int nelem, int elemsize, int ncomp, int compstride,
                              const double *l, double *e, hipStream_t s) {
  return rstr(off, nelem, elemsize, ncomp, compstride, l, e, 0, s);
}
hipError_t launch_rstr_scatter_add(const uint32_t *off, int nelem, int elemsize, int ncomp, int compstride,
                                   const double *e, double *l, hipStream_t s) {
  return rstr(off, nelem, elemsize, ncomp, compstride, e, l, 1, s);
}
hipError_t launch_multiplicity(const uint32_t *off, int nelem, int elemsize, int ncomp, int compstride,
                               double *l, hipStream_t s) {
  return rstr(off, nelem, elemsize, ncomp, compstride, nullptr, l, 2, s);
}
hipError_t launch_assemble(const uint32_t *rowptr, const uint32_t *cols, const uint32_t *node_off,
                           const unsigned char *flags, const double *evec, double *y, int nnodes,
                           int add, hipStream_t s, int max_blocks, const HaloUnpackArgs *unpack, const HaloPackFold *pack) {
  const int nun = unpack ? unpack->n : 0;
  if (nnodes <= 0 && nun <= 0) return hipSuccess;
#ifndef CPS_ASM_BLOCK
#define CPS_ASM_BLOCK 256    // (tuning hook) threads per workgroup of k_assemble: 128 and 512 measured in round 4, nothing
#endif
  constexpr int AB = CPS_ASM_BLOCK;
  unsigned nb_rows = (unsigned)((std::max(nnodes, 0) + AB - 1) / AB);
  if (max_blocks > 0 && nb_rows > (unsigned)max_blocks) nb_rows = (unsigned)max_blocks;     // (grid-stride loop over the rows)
  const unsigned nb_un = (unsigned)std::min((nun + AB - 1) / AB, 1024);
  hipLaunchKernelGGL(k_assemble, dim3(nb_rows + nb_un), dim3(AB), 0, s, rowptr, cols, node_off, flags, evec, y, nnodes, add,
                     (int)nb_rows, unpack ? *unpack : HaloUnpackArgs{nullptr, nullptr, nullptr, nullptr, 0},
                     pack ? *pack : HaloPackFold{nullptr, nullptr, nullptr});
  return hipGetLastError();
}
hipError_t launch_dot(const double *x, const double *y, const double *w, size_t n, double *result_dev, hipStream_t s, double *out) {
  // result_dev: 1 + 2048 doubles ([0] the result, then the per-block partials); out: a second, device-side destination
  const dim3 g = n ? stream_grid(n) : dim3(1);
  hipLaunchKernelGGL(k_dot, g, dim3(256), 0, s, x, y, w, n, result_dev);
  hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(256), 0, s, result_dev, (int)g.x, out);
  return hipGetLastError();
}
hipError_t launch_scalar_div(double *sc, int dst, int num, int den, double scale, hipStream_t s) {
  hipLaunchKernelGGL(k_scalar_div, dim3(1), dim3(64), 0, s, sc, dst, num, den, scale);
  return hipGetLastError();
}
hipError_t launch_axpby_dev(double *y, const double *sc, int ia, double sa, const double *x, int ib, double sb, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_axpby_dev, stream_grid(n), dim3(256), 0, s, y, sc, ia, sa, x, ib, sb, n);
  return hipGetLastError();
}

}  // namespace cps
