// kernels_fused_inst.hip -- instantiations of the fused operator kernel for one
// quadrature size.  Compiled once per -DCPS_Q=<Q> -DCPS_PART=<k> (see Makefile) so the
// instantiations build in parallel; each object exports launch_fused_grad_q<Q>p<k> and holds
// the kernels with (Q - P) % pencil_inst_parts(Q) == k.
//
// Node counts P per Q follow the level-degree rules of the reference
// (cloptions.c:195-225): the fine level has P = Q (qextra = 0), coarse levels
// use degrees 1, 2, 4 (logarithmic) or every degree below the fine one (uniform) with the FINE quadrature (setuplibceed.c:757).
// Residual kernels (which write the stored state) only exist on the fine level (P = Q, and P = Q - 1, Q - 2 for -qextra 1, 2).
#include "kernel_fused_pencil.hpp"

#ifndef CPS_Q
#error "compile with -DCPS_Q=<points per direction>"
#endif
#ifndef CPS_PART
#define CPS_PART 0
#endif

namespace cps {

#define CPS_CAT_(a, b) a##b
#define CPS_CAT(a, b) CPS_CAT_(a, b)
#define CPS_STR_(x) #x
#define CPS_STR(x) CPS_STR_(x)

// (the part test depends on the template parameter, so the kernels of the other parts are not instantiated here)
#define CPS_CASE(Pv, QFv, QFname)                                                   \
  if constexpr ((CPS_Q - (Pv)) % pencil_inst_parts(CPS_Q) == PART) {                \
    if (P == Pv && qf == QFv) {                                                     \
      *name = "fused_grad<P=" #Pv ",Q=" CPS_STR(CPS_Q) "," QFname ">/pencil";       \
      return launch_fused_pencil_t<Pv, CPS_Q, QFv>(t, a, s);                        \
    }                                                                               \
  }
// The derived-state tangent is instantiated where it measured a gain: Q >= 6 (one element per wave; -2.6 ... -3.1 % on config 5's
// block, same box); at Q = 5 it removes 9 % of the VALU instructions and 0 % of the time (pencil_derived_state, kernels.hpp).
#if CPS_Q >= CPS_DERIVED_MIN_Q
#define CPS_DERIVED(Pv) CPS_CASE(Pv, QF_HYPERFS_DF_DS, "HyperFSdF+derived")
#else
#define CPS_DERIVED(Pv)
#endif
#define CPS_JACOBIANS(Pv)              \
  CPS_CASE(Pv, QF_LINELAS, "LinElas")  \
  CPS_CASE(Pv, QF_HYPERSS_DF, "HyperSSdF") \
  CPS_CASE(Pv, QF_HYPERFS_DF, "HyperFSdF") \
  CPS_DERIVED(Pv)

// Residual kernels (they write the stored state) run on the FINE level only: P = Q, and P = Q - 1, Q - 2 for -qextra 1, 2
// (src/cloptions.c:53-55: Q = degree + 1 + qextra, setuplibceed.c:252).  A larger qextra is a loud "no fused kernel instantiated".
#define CPS_RESIDUALS(Pv) CPS_CASE(Pv, QF_HYPERSS_F, "HyperSSF") CPS_CASE(Pv, QF_HYPERFS_F, "HyperFSF")
#define CPS_FINE_WITH_QEXTRA(Pv) ((CPS_Q) > (Pv) && (CPS_Q) - (Pv) <= 2)

template <int PART>
static hipError_t dispatch_part(int P, int qf, const BasisTables &t, const FusedGradArgs &a, hipStream_t s, const char **name) {
  CPS_JACOBIANS(CPS_Q)
  CPS_RESIDUALS(CPS_Q)
#if CPS_Q > 2
  CPS_JACOBIANS(2)
#if CPS_FINE_WITH_QEXTRA(2)
  CPS_RESIDUALS(2)
#endif
#endif
#if CPS_Q > 3
  CPS_JACOBIANS(3)
#if CPS_FINE_WITH_QEXTRA(3)
  CPS_RESIDUALS(3)
#endif
#endif
#if CPS_Q > 4
  CPS_JACOBIANS(4)
#if CPS_FINE_WITH_QEXTRA(4)
  CPS_RESIDUALS(4)
#endif
#endif
#if CPS_Q > 5
  CPS_JACOBIANS(5)
#if CPS_FINE_WITH_QEXTRA(5)
  CPS_RESIDUALS(5)
#endif
#endif
#if CPS_Q > 6      // (uniform ladders of degrees 6 and 7, cloptions.c:195-225: every degree below the fine one is a level)
  CPS_JACOBIANS(6)
#if CPS_FINE_WITH_QEXTRA(6)
  CPS_RESIDUALS(6)
#endif
#endif
#if CPS_Q > 7
  CPS_JACOBIANS(7)
#if CPS_FINE_WITH_QEXTRA(7)
  CPS_RESIDUALS(7)
#endif
#endif
  return hipErrorInvalidValue;
}

hipError_t CPS_CAT(CPS_CAT(CPS_CAT(launch_fused_grad_q, CPS_Q), p), CPS_PART)(int P, int qf, const BasisTables &t, const FusedGradArgs &a,
                                                                              hipStream_t s, const char **name) {
  static_assert(CPS_PART >= 0 && CPS_PART < pencil_inst_parts(CPS_Q), "part of this quadrature size");
  return dispatch_part<CPS_PART>(P, qf, t, a, s, name);
}

}  // namespace cps
