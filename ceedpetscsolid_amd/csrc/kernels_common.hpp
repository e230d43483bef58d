// kernels_common.hpp -- device building blocks shared by the gfx950 kernels.
//
// `Geom<Q>` is the point-per-lane mapping of the SET-UP kernels (kernels_misc.hip: k_setup_geo, k_diag_sf): an
// element owns TPE lanes (Q^3 rounded up to a whole number of waves, or to a power of two when several
// elements share a wave) and a workgroup owns EPB elements.  The operator-apply kernel (kernel_fused_pencil.hpp)
// and, since round 5, the transfer kernels (k_transfer) have their own wave-level pencil mappings.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.hpp"

namespace cps {

#define CPS_DEV static __device__ __forceinline__

constexpr int cpow3(int n) { return n * n * n; }
constexpr int next_pow2(int n) { int p = 1; while (p < n) p *= 2; return p; }

template <int Q> struct Geom {
  static constexpr int Q3 = cpow3(Q);
  static constexpr int TPE = Q3 <= 32 ? next_pow2(Q3) : ((Q3 + 63) / 64) * 64;
#ifndef CPS_BLOCK_TARGET
#define CPS_BLOCK_TARGET 256
#endif
  static constexpr int EPB = TPE >= CPS_BLOCK_TARGET ? 1 : CPS_BLOCK_TARGET / TPE;
  static constexpr int BLOCK = TPE * EPB;
};

CPS_DEV void atomic_add_f64(double *p, double v) {
  // hardware f64 atomic (global_atomic_add_f64); no CAS loop
  unsafeAtomicAdd(p, v);
}

}  // namespace cps
