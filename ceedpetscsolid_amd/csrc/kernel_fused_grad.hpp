// kernel_fused_grad.hpp -- FIRST-GENERATION fused operator kernel (one output point per lane), kept
// for A/B against kernel_fused_pencil.hpp (CEED_MI355X_FUSED=rows).  The whole CeedOperatorApply of
// the residual / Jacobian operators (setuplibceed.c:517-542, :817-839) in ONE launch:
//   gather (offsets, Dirichlet flags) -> sum-factorised interpolation to the Gauss points ->
//   gradient (z-derivative for free inside the z-pass, x/y by the collocated derivative) ->
//   pointwise physics in registers (q-point data streamed once, coalesced, prefetched a full
//   element ahead) -> gradient^T -> interpolation^T -> element results to the E-vector (summed by
//   k_assemble) or, with CEED_MI355X_SCATTER=atomic, f64 atomic scatter-add.
// No quadrature-point intermediate is ever written out.
//
// Wave-per-element, barrier-free.  A workgroup is ONE wave64 and owns its
// element(s) outright, so every LDS write -> read dependency stays inside one
// wave, where the LDS queue is in-order: no cross-wave barrier is ever needed and
// the waves of a CU never wait for each other (v1, two waves per element around
// ten workgroup barriers, spent 54 % of its wave-cycles in SQ_WAIT_ANY).  Each lane
// owns SLOTS quadrature points of the element (q = lane + 64 s): Q=5 -> 125 points
// on 2 x 64 slots (97.7 % of lane-slots busy), Q=4 -> one point per lane, Q=7 -> 6
// slots; small elements share a wave (Q=3: two elements, Q=2: eight).
//
// Persistent waves (v4).  The grid is a fixed number of waves per CU; each wave walks a
// strided list of elements and software-pipelines the global traffic across elements:
// the NEXT element's offsets are requested at the top of an element, its x values after
// the interpolation and its first q-point slot after the last physics evaluation, so the
// dependent offset -> x round trips (10k cycles of a 42k-cycle wave lifetime in v3:
// tools/stamp_profile.py) and the tail wait for the atomics are off the critical path.
// Blocks b and b+8 share an XCD (round-robin dispatch), so wave b works on the b%8-th
// contiguous chunk of the element list: concurrently processed elements on one XCD are
// mesh neighbours and their shared nodes hit that XCD's L2.
//
// LDS layout (v3).  Every 1-D contraction reads a ROW that is contiguous in LDS:
// each pass writes its result with the NEXT pass's contraction index fastest, and
// rows are padded to an even length so they are 16-byte aligned.  A row of 5
// doubles is then 2 x ds_read_b128 + 1 x ds_read_b64 (10 LDS cycles) instead of
// the 2.5 x ds_read2_b64 (20 cycles: ds_read2_b64 runs at half the byte rate on
// gfx950, MI355X_MICROARCH.md LDS table) hipcc emits for strided scalar reads;
// v2 had the LDS pipe busy 41 % of the kernel on those.
#pragma once
#include "kernels_common.hpp"
#include "qfunctions_device.hpp"

namespace cps {

constexpr int pad2(int n) { return n + (n & 1); }

template <int P, int Q> struct WaveGeom {
  static constexpr int Q3 = Q * Q * Q;
  static constexpr int TPE = Q3 <= 32 ? next_pow2(Q3) : ((Q3 + 63) / 64) * 64;  // point slots per element
  static constexpr int EPW = TPE >= 64 ? 1 : 64 / TPE;                           // elements per wave
  // Waves per element.  One wave owns its element(s) outright up to Q = 5 (2 point slots per lane at
  // Q = 5).  From Q = 6 the LDS slab (47 KB at Q = 7) would leave 3 single-wave workgroups per CU, so
  // TPE/128 waves share an element -- always 2 point slots per lane -- with a raw s_barrier behind an
  // LDS-only wait at the phase boundaries: Q = 6: 2 waves, Q = 7: 3 (6 waves per CU), Q = 8: 4.
#ifndef CPS_WPE_128   // waves per element when the element has exactly 128 point slots (Q = 5)
#define CPS_WPE_128 1
#endif
  static constexpr int WPE = TPE >= 256 ? TPE / 128 : (TPE == 128 ? CPS_WPE_128 : 1);
  static constexpr int NT = 64 * WPE;                                            // lanes per workgroup
  static constexpr int SLOTS = TPE >= NT ? TPE / NT : 1;                         // point slots per lane
  static constexpr int LD = pad2(Q), LDP = pad2(P);                              // padded row lengths
  static constexpr int BLK = 3 * Q * Q * LD;                                     // one 3-component block
  static constexpr int SLAB = 5 * BLK;                                           // doubles of LDS per element
};

// Phase boundary inside ONE wave.  The hardware executes a wave's LDS instructions in
// order, so a ds_write followed by another lane's ds_read of that address needs no wait
// and no s_barrier -- only the COMPILER must not move LDS accesses across the boundary.
// A wavefront-scope fence does exactly that and emits no instruction; in particular it
// does not drain vmcnt, so the q-point loads issued at kernel entry stay in flight across
// the whole interpolation.
// With two waves per element (WPE = 2) the boundary is a raw s_barrier behind an LDS-only wait:
// __syncthreads() would also drain vmcnt and with it the q-point prefetch.
template <int WPE>
CPS_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  if constexpr (WPE > 1) {
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
    __builtin_amdgcn_s_barrier();
  } else {
    __builtin_amdgcn_wave_barrier();
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// dot product of a coefficient row (registers) with a 16-byte-aligned LDS row of N doubles
template <int N>
CPS_DEV double row_dot(const double *coef, const double *row) {
  double t = 0.;
#pragma unroll
  for (int m = 0; m + 1 < N; m += 2) {
    const double2 v = *reinterpret_cast<const double2 *>(row + m);
    t += coef[m] * v.x;
    t += coef[m + 1] * v.y;
  }
  if (N & 1) t += coef[N - 1] * row[N - 1];
  return t;
}
template <int N>
CPS_DEV void row_load(const double *row, double *r) {
#pragma unroll
  for (int m = 0; m + 1 < N; m += 2) {
    const double2 v = *reinterpret_cast<const double2 *>(row + m);
    r[m] = v.x; r[m + 1] = v.y;
  }
  if (N & 1) r[N - 1] = row[N - 1];
}

#ifndef CPS_FUSED_MINW
#define CPS_FUSED_MINW 1
#endif
template <int P, int Q, int QF>
__global__ __launch_bounds__((WaveGeom<P, Q>::NT), CPS_FUSED_MINW) void k_fused_grad(const BasisTables tab, const FusedGradArgs a) {
  using G = WaveGeom<P, Q>;
  constexpr int NT = G::NT, WPE = G::WPE;
  constexpr int Q3 = G::Q3, P3 = P * P * P, TPE = G::TPE, EPW = G::EPW, SLOTS = G::SLOTS;
  constexpr int LD = G::LD, LDP = G::LDP, BLK = G::BLK;
  constexpr bool ST_IN = QFTraits<QF>::state_in, ST_OUT = QFTraits<QF>::state_out;
  static_assert(P <= Q, "interpolation to at least as many points as nodes");

  // coefficient tables, rows padded to even length (16-byte aligned rows)
  __shared__ __attribute__((aligned(16))) double sB[Q * LDP];    // B[q][p]
  __shared__ __attribute__((aligned(16))) double sG[Q * LDP];    // G[q][p]  (grad1d)
  __shared__ __attribute__((aligned(16))) double sBt[P * LD];    // B^T[p][q]
  __shared__ __attribute__((aligned(16))) double sD[Q * LD];     // Dq[q][m] (collocated derivative)
  __shared__ __attribute__((aligned(16))) double sDt[Q * LD];    // Dq^T[q][m] = Dq[m][q]
  __shared__ __attribute__((aligned(16))) double slab[EPW][G::SLAB];

#ifdef CPS_STAMPS  // diagnostic build: where does a wave's lifetime go?  (never in the product build)
  unsigned long long stamp_[8]; int nst_ = 0;
#define CPS_STAMP() do { __builtin_amdgcn_sched_barrier(0); stamp_[nst_++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CPS_STAMP() do {} while (0)
#endif
  const int lane = threadIdx.x;
  const int el = EPW > 1 ? lane / TPE : 0;
  const int q0 = EPW > 1 ? lane % TPE : lane;  // slot s handles point q0 + NT s
  // five 3-component blocks, reused along the pipeline (who is dead when is noted at each phase)
  double *B0 = slab[el], *B1 = B0 + BLK, *B2 = B0 + 2 * BLK, *B3 = B0 + 3 * BLK, *B4 = B0 + 4 * BLK;

  // coefficient tables: staged once per wave, reused for every element it processes
  for (int i = lane; i < Q * P; i += NT) {
    const int qq = i / P, pp = i % P;
    sB[qq * LDP + pp] = tab.interp[i];
    sG[qq * LDP + pp] = tab.grad[i];
    sBt[pp * LD + qq] = tab.interp[i];
  }
  for (int i = lane; i < Q * Q; i += NT) {
    const int qq = i / Q, mm = i % Q;
    sD[qq * LD + mm] = tab.colo[i];
    sDt[mm * LD + qq] = tab.colo[i];
  }

  // ---- work list of this wave (XCD-aware) --------------------------------------------------
  const int ngroups = (a.nelem + EPW - 1) / EPW;          // a group = the EPW elements of one wave pass
  const int nxcd = (gridDim.x % 8 == 0) ? 8 : 1;
  const int xcd = blockIdx.x % nxcd, wrank = blockIdx.x / nxcd, wper = gridDim.x / nxcd;
  const int chunk = (ngroups + nxcd - 1) / nxcd;           // contiguous groups per XCD
  const int gbeg = xcd * chunk, gend = min(ngroups, gbeg + chunk);
  int grp = gbeg + wrank;
  if (grp >= gend) return;

  // All global loads are unconditional and in-bounds (indices clamped, results selected
  // afterwards): no exec-mask branches around loads, so a wave's loads issue back to back.
  auto elem_of = [&](int g) { const int ee = g * EPW + el; return a.elem_begin + (ee < a.nelem ? ee : a.nelem - 1); };
  // q-point data lives in NSET register sets used round-robin by the point slots; a set is
  // refilled (for the slot NSET positions further down the element/slot stream) as soon as the
  // physics of its current slot is done, so every q-point load has a full element of work to hide under.
  constexpr int NSET = SLOTS >= 2 ? 2 : 1;
  double qd[NSET][10], st[NSET][9];
  auto load_point = [&](double *qdv, double *stv, int ec, int q) {
#ifdef CPS_ABLATE_QDATA  // timing-only build: no q-point stream (WRONG results)
    for (int c = 0; c < 10; c++) qdv[c] = (c == 1 || c == 5 || c == 9 || c == 0) ? 1.0 + 1e-3 * q : 1e-3 * c;
    for (int c = 0; c < 9; c++) stv[c] = 1e-3 * (c + lane);
    return;
#endif
    const int qc = q < Q3 ? q : Q3 - 1;
    const double *qp = a.qdata + (size_t)ec * 10 * Q3 + qc;
#pragma unroll
    for (int c = 0; c < 10; c++) qdv[c] = qp[c * Q3];
    if constexpr (ST_IN) {
      const double *sp = a.state_in + (size_t)ec * 9 * Q3 + qc;
#pragma unroll
      for (int c = 0; c < 9; c++) stv[c] = sp[c * Q3];
    }
  };
  auto load_offsets = [&](int ec, uint32_t *o) {
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
      const int n = q0 + NT * s;
      o[s] = a.offsets[(size_t)ec * P3 + (n < P3 ? n : P3 - 1)];
    }
  };
  auto load_x = [&](const uint32_t *o, double (*xv)[3]) {
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
      const uint32_t base = o[s] & OFF_MASK;
#pragma unroll
      for (int c = 0; c < 3; c++) xv[s][c] = a.x[base + c];
    }
  };

  // ---- loop-invariant LDS indices of this lane's point slots (hoisted out of the element loop:
  // v4 spent ~a quarter of its VALU instructions re-deriving them per element) ----------------
  struct SlotIdx {
    int g_w;                 // gather: write position in XA
    int x_r, x_w, x_c;       // x pass: read row, write position, coefficient row
    int y_r, y_w, y_c;       // y pass
    int z_r, z_c;            // z pass (writes use rowX+i / rowY+j)
    int rowX, rowY, rowZ;    // point rows: (k*Q+j)*LD, (k*Q+i)*LD, (j*Q+i)*LD
    int qi, qj, qk;
    int zt_r, zt_w, zt_c;    // z^T
    int yt_r, yt_w, yt_c;    // y^T
    int xt_r, xt_c;          // x^T
  } ix[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = q0 + NT * s;
    SlotIdx &I = ix[s];
    { const int i = q % P, kj = q / P; I.g_w = kj * LDP + i; I.xt_r = kj * LD; I.xt_c = i * LD; }
    { const int i = q % Q, kj = q / Q, j = kj % P, k = kj / P;
      I.x_r = kj * LDP; I.x_w = (k * Q + i) * LDP + j; I.x_c = i * LDP;
      I.yt_r = (k * Q + i) * LD; I.yt_w = (k * P + j) * LD + i; I.yt_c = j * LD; }
    { const int i = q % Q, j = (q / Q) % Q, k = q / (Q * Q);
      I.y_r = (k * Q + i) * LDP; I.y_w = (j * Q + i) * LDP + k; I.y_c = j * LDP;
      I.z_r = (j * Q + i) * LDP; I.z_c = k * LDP;
      I.rowX = (k * Q + j) * LD; I.rowY = (k * Q + i) * LD; I.rowZ = (j * Q + i) * LD;
      I.qi = i; I.qj = j; I.qk = k;
      I.zt_r = (j * Q + i) * LD; I.zt_w = (k * Q + i) * LD + j; I.zt_c = k * LD; }
  }
  constexpr int CPP = P * P * LDP, CPQ = P * Q * LDP, CQQ = Q * Q * LDP;   // component strides, P-padded rows
  constexpr int DQQ = Q * Q * LD, DPQ = P * Q * LD, DPP = P * P * LD;      // component strides, Q-padded rows

  // ---- pipeline prologue: first element's offsets, x and slot-0 q-point data ----------------
  uint32_t off[SLOTS], off_nx[SLOTS];
  double xin[SLOTS][3];
  load_offsets(elem_of(grp), off);
  load_x(off, xin);
#pragma unroll
  for (int t = 0; t < NSET; t++) load_point(qd[t], st[t], elem_of(grp), q0 + NT * t);

  for (;; ) {
  CPS_STAMP();
  const bool live = grp * EPW + el < a.nelem;
  const int e = a.elem_begin + grp * EPW + el;
  const int ec = live ? e : a.elem_begin + a.nelem - 1;
  const int grp_nx = grp + wper;
  const bool more = grp_nx < gend;                           // wave-uniform
  const int ec_nx = elem_of(more ? grp_nx : grp);
  load_offsets(ec_nx, off_nx);                               // next element's offsets: a whole element ahead

  // XA = B0: [c][k][j][i], i fastest (row length LDP)
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int n = q0 + NT * s;
    if (n < P3) {
      const uint32_t fl = (a.mask_in && live) ? (off[s] >> OFF_FLAG_SHIFT) : (live ? 0u : 7u);
#pragma unroll
      for (int c = 0; c < 3; c++) B0[c * CPP + ix[s].g_w] = ((fl >> c) & 1u) ? 0. : xin[s][c];
    }
  }
  wave_sync<WPE>();
  CPS_STAMP();  // 1: gather landed in LDS

  // ---- B: nodes -> points ------------------------------------------------------------------
  // x: XA[c][k][j][:] -> RB = B1: [c][k][i'][j], j fastest
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = q0 + NT * s;
    if (q < P * P * Q) {
      double b[P];
      row_load<P>(sB + ix[s].x_c, b);
#pragma unroll
      for (int c = 0; c < 3; c++) B1[c * CPQ + ix[s].x_w] = row_dot<P>(b, B0 + c * CPP + ix[s].x_r);
    }
  }
  wave_sync<WPE>();
  // y: RB[c][k][i'][:] -> RC = B2: [c][j'][i'][k], k fastest
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = q0 + NT * s;
    if (q < P * Q * Q) {
      double b[P];
      row_load<P>(sB + ix[s].y_c, b);
#pragma unroll
      for (int c = 0; c < 3; c++) B2[c * CQQ + ix[s].y_w] = row_dot<P>(b, B1 + c * CPQ + ix[s].y_r);
    }
  }
  wave_sync<WPE>();
  // z: RC[c][j'][i'][:] -> U and, from the same row, dU/dz (grad1d row).  U is stored twice:
  // UX = B0 [c][k'][j'][i'] (i fastest) and UY = B1 [c][k'][i'][j'] (j fastest).
  double uz[SLOTS][3];
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = q0 + NT * s;
    if (q < Q3) {
      double b[P], g[P];
      row_load<P>(sB + ix[s].z_c, b);
      row_load<P>(sG + ix[s].z_c, g);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double r[P];
        row_load<P>(B2 + c * CQQ + ix[s].z_r, r);
        double t = 0., tz = 0.;
#pragma unroll
        for (int m = 0; m < P; m++) { t += b[m] * r[m]; tz += g[m] * r[m]; }
        uz[s][c] = tz;
        B0[c * DQQ + ix[s].rowX + ix[s].qi] = t;
        B1[c * DQQ + ix[s].rowY + ix[s].qj] = t;
      }
    }
  }
  wave_sync<WPE>();
  CPS_STAMP();  // 2: interpolated
  load_x(off_nx, xin);                                       // next element's x (its offsets have landed)

  // ---- x/y collocated gradient + physics, one point slot at a time -------------------------
  // reads UX = B0, UY = B1; writes GX = B2 (i fastest), GY = B3 (j fastest), GZ = B4 (k fastest)
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = q0 + NT * s;
    double ug[9], dv[9], sto[9];
    if (q < Q3) {
      double d0[Q], d1[Q];
      row_load<Q>(sD + ix[s].qi * LD, d0);
      row_load<Q>(sD + ix[s].qj * LD, d1);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        ug[0 * 3 + c] = row_dot<Q>(d0, B0 + c * DQQ + ix[s].rowX);
        ug[1 * 3 + c] = row_dot<Q>(d1, B1 + c * DQQ + ix[s].rowY);
        ug[2 * 3 + c] = uz[s][c];
      }
    }
    if (live && q < Q3) {
      qf_point<QF>(Phys{a.nu, a.E, a.lambda, a.TwoMu}, ug, qd[s % NSET], st[s % NSET], dv, sto);
      if constexpr (ST_OUT) {
        double *sp = a.state_out + (size_t)e * 9 * Q3 + q;
#pragma unroll
        for (int c = 0; c < 9; c++) sp[c * Q3] = sto[c];
      }
    } else {
#pragma unroll
      for (int c = 0; c < 9; c++) dv[c] = 0.;
    }
    if (s + NSET < SLOTS) load_point(qd[s % NSET], st[s % NSET], ec, q + NT * NSET);       // same element, NSET slots on
    else load_point(qd[s % NSET], st[s % NSET], ec_nx, q0 + NT * (s + NSET - SLOTS));      // next element
    if (q < Q3) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        B2[c * DQQ + ix[s].rowX + ix[s].qi] = dv[0 * 3 + c];
        B3[c * DQQ + ix[s].rowY + ix[s].qj] = dv[1 * 3 + c];
        B4[c * DQQ + ix[s].rowZ + ix[s].qk] = dv[2 * 3 + c];
      }
    }
  }
  wave_sync<WPE>();
  CPS_STAMP();  // 3: physics done

  // ---- gradient^T: GX, GY, GZ -> WZ = B0 [c][j][i][k], k fastest (UX is dead) -----------------
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = q0 + NT * s;
    if (q < Q3) {
      double d0[Q], d1[Q], d2[Q];
      row_load<Q>(sDt + ix[s].qi * LD, d0);
      row_load<Q>(sDt + ix[s].qj * LD, d1);
      row_load<Q>(sDt + ix[s].qk * LD, d2);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double t = row_dot<Q>(d0, B2 + c * DQQ + ix[s].rowX) + row_dot<Q>(d1, B3 + c * DQQ + ix[s].rowY) +
                         row_dot<Q>(d2, B4 + c * DQQ + ix[s].rowZ);
        B0[c * DQQ + ix[s].rowZ + ix[s].qk] = t;
      }
    }
  }
  wave_sync<WPE>();
  CPS_STAMP();  // 4: gradient^T done

  // ---- B^T: points -> nodes, then scatter-add -----------------------------------------------
  // z^T: WZ[c][j'][i'][:] -> TY = B1 [c][k][i'][j'], j fastest (UY is dead)
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = q0 + NT * s;
    if (q < P * Q * Q) {
      double b[Q];
      row_load<Q>(sBt + ix[s].zt_c, b);
#pragma unroll
      for (int c = 0; c < 3; c++) B1[c * DPQ + ix[s].zt_w] = row_dot<Q>(b, B0 + c * DQQ + ix[s].zt_r);
    }
  }
  wave_sync<WPE>();
  // y^T: TY[c][k][i'][:] -> TX = B2 [c][k][j][i'], i fastest (GX is dead)
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int q = q0 + NT * s;
    if (q < P * P * Q) {
      double b[Q];
      row_load<Q>(sBt + ix[s].yt_c, b);
#pragma unroll
      for (int c = 0; c < 3; c++) B2[c * DPP + ix[s].yt_w] = row_dot<Q>(b, B1 + c * DPQ + ix[s].yt_r);
    }
  }
  wave_sync<WPE>();
  // x^T + scatter
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int n = q0 + NT * s;
    if (live && n < P3) {
      double b[Q];
      row_load<Q>(sBt + ix[s].xt_c, b);
      const uint32_t base = off[s] & OFF_MASK;
      const uint32_t fl = a.mask_out ? (off[s] >> OFF_FLAG_SHIFT) : 0u;
      if (a.evec) {  // wave-uniform: element results as plain coalesced stores, summed by launch_assemble()
#pragma unroll
        for (int c = 0; c < 3; c++)
          a.evec[((size_t)e * P3 + n) * 3 + c] = row_dot<Q>(b, B2 + c * DPP + ix[s].xt_r);
        continue;
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double t = row_dot<Q>(b, B2 + c * DPP + ix[s].xt_r);
#if defined(CPS_ABLATE_ATOMICS)   // timing-only build: plain store instead of the atomic (WRONG results)
        if (!((fl >> c) & 1u)) a.y[base + c] = t;
#else
        if (!((fl >> c) & 1u)) atomic_add_f64(a.y + base + c, t);
#endif
      }
    }
  }
  CPS_STAMP();  // 5: atomics issued (never waited for: the wave moves on)
#ifdef CPS_STAMPS
  if (a.stamps && lane == 0) {
    for (int i = 0; i < nst_; i++) a.stamps[(size_t)grp * 8 + i] = stamp_[i];
  }
  nst_ = 0;
#endif
  if (!more) break;
  grp = grp_nx;
#pragma unroll
  for (int s = 0; s < SLOTS; s++) off[s] = off_nx[s];
  wave_sync<WPE>();  // WAR: the next element's gather overwrites B0, which x^T's source B2 does not alias
  }  // element loop
}

// waves per CU the persistent grid is sized for: LDS (5 blocks + tables per wave) and VGPRs
template <int P, int Q> constexpr int fused_waves_per_cu() {
  using G = WaveGeom<P, Q>;
  constexpr int lds = (G::EPW * G::SLAB + 2 * Q * G::LDP + P * G::LD + 2 * Q * G::LD) * 8;
  constexpr int by_lds = (160 * 1024) / lds;          // workgroups per CU by LDS
  constexpr int by_vgpr = 8 / G::WPE;                 // 8 waves per CU = 2 per SIMD at <= 256 VGPRs
  return by_lds < 1 ? 1 : (by_lds > by_vgpr ? by_vgpr : by_lds);
}

template <int P, int Q, int QF>
hipError_t launch_fused_grad_t(const BasisTables &t, const FusedGradArgs &a, hipStream_t s) {
  using G = WaveGeom<P, Q>;
  if (a.nelem <= 0) return hipSuccess;
  const int ngroups = (a.nelem + G::EPW - 1) / G::EPW;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    ncu = prop.multiProcessorCount;
  }
  int grid = ncu * fused_waves_per_cu<P, Q>();   // persistent: a multiple of 8 on MI355X (256 CUs)
  if (grid > ngroups) grid = ngroups;
  hipLaunchKernelGGL((k_fused_grad<P, Q, QF>), dim3(grid), dim3(G::NT), 0, s, t, a);
  return hipGetLastError();
}

}  // namespace cps
