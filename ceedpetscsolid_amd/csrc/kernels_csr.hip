// kernels_csr.hip -- assembled coarse-level operator: deterministic COO -> CSR summation and SpMV.
// The p=1 level of the multigrid (22 k rows, 81 entries per row at config 3) is the only consumer:
// everything here is launch-latency bound, so the kernels are simple and few.
#include "kernels_common.hpp"

namespace cps {

// vals[s] = sum over the COO entries mapped to slot s, in ascending entry order (slotptr / perm = transpose
// of coo_slot, built on the host); then the unit diagonals.
__global__ void k_csr_sum(const uint32_t *slotptr, const uint32_t *perm, const double *coo, double *vals, int nnz) {
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nnz; s += gridDim.x * blockDim.x) {
    double a = 0.;
    for (uint32_t k = slotptr[s]; k < slotptr[s + 1]; k++) a += coo[perm[k]];
    vals[s] = a;
  }
}
__global__ void k_csr_unit_diag(const uint32_t *diag_slot, double *vals, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) vals[diag_slot[i]] = 1.;
}
// one wave64 per row
__global__ __launch_bounds__(256) void k_csr_spmv(const uint32_t *rowptr, const uint32_t *cols, const double *vals,
                                                 const double *x, double *y, int nrows) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < nrows; r += nw) {
    double a = 0.;
    for (uint32_t k = rowptr[r] + lane; k < rowptr[r + 1]; k += 64) a += vals[k] * x[cols[k]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if (lane == 0) y[r] = a;
  }
}
// y = A x, "CSR-stream" form (round 5): a workgroup owns a run of CONSECUTIVE rows holding at most SPMV_CHUNK entries (row_block[b] ..
// row_block[b + 1], cut on the host once per pattern).  Its entries are fetched in ONE coalesced sweep that does not wait for rowptr
// (cols and vals of the run are contiguous), multiplied by the gathered x into LDS, and every row is then summed from LDS by T lanes (T a
// power of two, the partial sums joined by a butterfly): two dependent memory trips (cols -> x) with up to 24 KB in flight per workgroup,
// instead of three (rowptr -> cols, vals -> x) with 1.3 KB per wave in the wave-per-row form -- the p = 1 level of config 4 (330 k rows of
// 81) ran at 2.7 TB/s that way.  A row longer than the chunk is a run of its own, summed by the whole workgroup.  Fixed summation order.
constexpr int SPMV_CHUNK = 2048, SPMV_BLOCK = 256;
__global__ __launch_bounds__(SPMV_BLOCK) void k_csr_spmv_stream(const uint32_t *row_block, const uint32_t *rowptr, const uint32_t *cols,
                                                               const double *vals, const double *x, double *y) {
  __shared__ double prod[SPMV_CHUNK];
  const int tid = threadIdx.x;
  const uint32_t r0 = row_block[blockIdx.x], r1 = row_block[blockIdx.x + 1];
  const uint32_t k0 = rowptr[r0], k1 = rowptr[r1], n = k1 - k0;
  const int nrows = (int)(r1 - r0);
  if (n > (uint32_t)SPMV_CHUNK) {               // one long row: strided partial sums, then a fixed tree over the workgroup
    double a = 0.;
    for (uint32_t k = k0 + tid; k < k1; k += SPMV_BLOCK) a += vals[k] * x[cols[k]];
    prod[tid] = a;
    __syncthreads();
    for (int o = SPMV_BLOCK / 2; o > 0; o >>= 1) {
      if (tid < o) prod[tid] += prod[tid + o];
      __syncthreads();
    }
    if (tid == 0) y[r0] = prod[0];
    return;
  }
  // lanes per row: the largest power of two with nrows * T <= SPMV_BLOCK (at most 64: a butterfly inside one wave)
  int T = 1;
  while (T < 64 && nrows * (2 * T) <= SPMV_BLOCK) T *= 2;
  const int row = tid / T, sub = tid % T;
  uint32_t ra = 0, rb = 0;
  if (row < nrows) { ra = rowptr[r0 + row] - k0; rb = rowptr[r0 + row + 1] - k0; }     // (requested with the sweep, needed after it)
  constexpr int TRIPS = SPMV_CHUNK / SPMV_BLOCK;
  uint32_t cc[TRIPS];
  double vv[TRIPS];
#pragma unroll
  for (int j = 0; j < TRIPS; j++) {
    const uint32_t i = (uint32_t)tid + j * SPMV_BLOCK, k = k0 + (i < n ? i : 0);
    cc[j] = cols[k]; vv[j] = vals[k];
  }
#pragma unroll
  for (int j = 0; j < TRIPS; j++) {
    const uint32_t i = (uint32_t)tid + j * SPMV_BLOCK;
    const double xv = x[cc[j]];
    if (i < n) prod[i] = vv[j] * xv;
  }
  __syncthreads();
  double a = 0.;
  for (uint32_t i = ra + sub; i < rb; i += T) a += prod[i];
  for (int o = T / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (row < nrows && sub == 0) y[r0 + row] = a;
}
hipError_t launch_csr_spmv_stream(const uint32_t *row_block, int nblocks, const uint32_t *rowptr, const uint32_t *cols, const double *vals,
                                  const double *x, double *y, hipStream_t s) {
  if (nblocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_csr_spmv_stream, dim3((unsigned)nblocks), dim3(SPMV_BLOCK), 0, s, row_block, rowptr, cols, vals, x, y);
  return hipGetLastError();
}
__global__ void k_csr_diag(const uint32_t *diag_slot_of_row, const double *vals, double *d, int nrows) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) {
    const uint32_t s = diag_slot_of_row[r];
    d[r] = s == 0xFFFFFFFFu ? 0. : vals[s];
  }
}

// C = L R on FIXED patterns, without term lists: one wave per row r of C; lane l owns the entries s = rowptr[r] + l + 64 m of
// that row and forms  C[r, c] = sum_k L[r, k] R[k, c]  by walking L's row (wave-uniform loads) and looking c up in R's row
// k by binary search (R's columns are sorted within a row; the rows hit are small and shared by the 64 lanes, so the
// probes are cache hits).  The k order is the order of L's row: every entry is summed the same way every time.  Entries
// of C's pattern that the product does not reach (a full "dense" pattern) come out as zeros.
// Round 2 held, per entry of C, the list of (slot, weight) terms -- 12 bytes per scalar multiplication: 36 M terms on
// config 3's mesh, 4.7e8 (5.6 GB, 6 s of host set-up) on the 44 928-hex reference cylinder, out of reach at config 4's
// 99 000.  This form needs the patterns only.
__global__ __launch_bounds__(256) void k_csr_spgemm(const uint32_t *l_rowptr, const uint32_t *l_cols, const double *l_vals,
                                                   const uint32_t *r_rowptr, const uint32_t *r_cols, const double *r_vals,
                                                   const uint32_t *c_rowptr, const uint32_t *c_cols, double *c_vals, int nrows) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < nrows; r += nw) {
    const uint32_t k0 = l_rowptr[r], k1 = l_rowptr[r + 1];
    for (uint32_t s = c_rowptr[r] + lane; s < c_rowptr[r + 1]; s += 64) {
      const uint32_t c = c_cols[s];
      double acc = 0.;
      for (uint32_t k = k0; k < k1; k++) {
        const uint32_t j = l_cols[k];
        uint32_t lo = r_rowptr[j], hi = r_rowptr[j + 1];
        if (lo == hi || c < r_cols[lo] || c > r_cols[hi - 1]) continue;
        while (hi - lo > 1) {                       // invariant: r_cols[lo] <= c
          const uint32_t mid = (lo + hi) >> 1;
          if (r_cols[mid] <= c) lo = mid; else hi = mid;
        }
        if (r_cols[lo] == c) acc += l_vals[k] * r_vals[lo];
      }
      c_vals[s] = acc;
    }
  }
}

// ---- in-place inverse of a dense SPD matrix (row-major n x n), blocked Gauss-Jordan without pivoting ------------------
// Block step k with pivot block K and the rest R:  D = inv(A[K,K]);  A[K,R] = D A[K,R];  A[R,R] -= A[R,K] A[K,R];
// A[R,K] = -A[R,K] D;  A[K,K] = D.  Four small launches per step; the matrix (a few MB) stays in L2.  Ragged last block:
// loads outside the matrix read the identity, stores outside are dropped.
constexpr int GJ = 32;
__device__ inline double gj_load(const double *A, int n, int r, int c) { return (r < n && c < n) ? A[(size_t)r * n + c] : (r == c ? 1. : 0.); }
// ONE WAVE: unblocked Gauss-Jordan of the 32 x 32 pivot block in registers.  Lane l holds row l & 31, columns
// 16 (l >> 5) .. + 15 (sixteen doubles); per step the pivot row comes from lane p (+ 32 for the upper column half) by sixteen
// bpermutes, the pivot column entry of the own row from the lane pair's other half by one -- no LDS, no barrier.  The 32
// steps are one dependent chain (bpermute -> divide -> FMA -> next bpermute): 14.2 us measured, against 15.9 us for the
// 1 024-thread LDS version with three barriers per step; the inverse of an n x n matrix has n such steps in a row,
// which is what bounds it (0.5 of its 1.3 ms at n = 1 080).
__global__ __launch_bounds__(64) void k_gj_pivot(double *A, int n, int k0, double *D, int *info) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  double r[16];
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = gj_load(A, n, k0 + i, k0 + 16 * h + k);
  bool bad = false;
#pragma unroll
  for (int p = 0; p < GJ; p++) {
    const int hp = p >> 4, kp = p & 15;           // compile-time after unrolling
    double row[16];
#pragma unroll
    for (int k = 0; k < 16; k++) row[k] = __shfl(r[k], p + 32 * h, 64);          // M[p][16 h + k]
    const double mine = r[kp], other = __shfl_xor(r[kp], 32, 64);
    const double mip = h == hp ? mine : other;                                      // M[i][p]
    const double piv = __shfl(r[kp], p + 32 * hp, 64);                              // M[p][p]
    if (!(piv > 0.)) bad = true;
    const double rp = 1. / piv, f = mip * rp;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const bool pc = (h == hp) && (k == kp);     // this register is column p
      double v;
      if (i == p) v = pc ? rp : row[k] * rp;
      else v = pc ? -f : r[k] - f * row[k];
      r[k] = v;
    }
    if (bad && lane == 0 && *info == 0) *info = k0 + p + 1;      // not positive definite (reported, not repaired)
    bad = false;
  }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    D[i * GJ + 16 * h + k] = r[k];
    if (k0 + i < n && k0 + 16 * h + k < n) A[(size_t)(k0 + i) * n + k0 + 16 * h + k] = r[k];
  }
}
// A[K, tile] = D * A[K, tile] for every column tile but the pivot's (mode 0);  A[tile, K] = -A[tile, K] * D (mode 1)
__global__ __launch_bounds__(GJ * GJ) void k_gj_panel(double *A, int n, int k0, const double *D, int mode) {
  __shared__ double Ds[GJ][GJ + 1], T[GJ][GJ + 1];
  const int i = threadIdx.y, j = threadIdx.x, t0 = blockIdx.x * GJ;
  if (t0 == k0) return;
  Ds[i][j] = D[i * GJ + j];
  const int r = mode == 0 ? k0 + i : t0 + i, c = mode == 0 ? t0 + j : k0 + j;
  T[i][j] = (r < n && c < n) ? A[(size_t)r * n + c] : 0.;
  __syncthreads();
  double a = 0.;
  if (mode == 0) { for (int p = 0; p < GJ; p++) a += Ds[i][p] * T[p][j]; }
  else           { for (int p = 0; p < GJ; p++) a -= T[i][p] * Ds[p][j]; }
  if (r < n && c < n) A[(size_t)r * n + c] = a;
}
// A[ti, tj] -= A[ti, K] * A[K, tj] for all tiles with ti != K, tj != K
__global__ __launch_bounds__(GJ * GJ) void k_gj_update(double *A, int n, int k0) {
  __shared__ double C[GJ][GJ + 1], R[GJ][GJ + 1];
  const int i = threadIdx.y, j = threadIdx.x, r0 = blockIdx.y * GJ, c0 = blockIdx.x * GJ;
  if (r0 == k0 || c0 == k0) return;
  C[i][j] = (r0 + i < n && k0 + j < n) ? A[(size_t)(r0 + i) * n + k0 + j] : 0.;
  R[i][j] = (k0 + i < n && c0 + j < n) ? A[(size_t)(k0 + i) * n + c0 + j] : 0.;
  __syncthreads();
  double a = 0.;
  for (int p = 0; p < GJ; p++) a += C[i][p] * R[p][j];
  if (r0 + i < n && c0 + j < n) A[(size_t)(r0 + i) * n + c0 + j] -= a;
}
// the inverse of a symmetric matrix is symmetric; rounding is not: A = (A + A^T) / 2, each pair by one thread
__global__ void k_symmetrize(double *A, int n) {
  const size_t tot = (size_t)n * n;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < tot; t += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(t / n), c = (int)(t % n);
    if (c < r) {
      const double m = 0.5 * (A[t] + A[(size_t)c * n + r]);
      A[t] = m; A[(size_t)c * n + r] = m;
    }
  }
}

static inline dim3 grid_for(size_t n, int per_block) {
  size_t b = (n + per_block - 1) / per_block;
  return dim3((unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b)));
}
hipError_t launch_csr_sum(const uint32_t *slotptr, const uint32_t *perm, const double *coo, double *vals, int nnz,
                          const uint32_t *unit_diag_slot, int n_unit, hipStream_t s) {
  if (nnz > 0) hipLaunchKernelGGL(k_csr_sum, grid_for((size_t)nnz, 256), dim3(256), 0, s, slotptr, perm, coo, vals, nnz);
  if (n_unit > 0) hipLaunchKernelGGL(k_csr_unit_diag, grid_for((size_t)n_unit, 256), dim3(256), 0, s, unit_diag_slot, vals, n_unit);
  return hipGetLastError();
}
hipError_t launch_csr_spmv(const uint32_t *rowptr, const uint32_t *cols, const double *vals, const double *x, double *y,
                           int nrows, hipStream_t s) {
  if (nrows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_csr_spmv, grid_for((size_t)nrows, 4), dim3(256), 0, s, rowptr, cols, vals, x, y, nrows);
  return hipGetLastError();
}
hipError_t launch_csr_diag(const uint32_t *diag_slot_of_row, const double *vals, double *d, int nrows, hipStream_t s) {
  if (nrows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_csr_diag, grid_for((size_t)nrows, 256), dim3(256), 0, s, diag_slot_of_row, vals, d, nrows);
  return hipGetLastError();
}
// The same product for a DENSE result (row-major nrows x ncols, every entry present: the coarsest Galerkin matrix, up to
// CSR_DENSE_MAX_COLS columns).  Searching every one of the ncols columns in rows of R that hold ~50 of them wastes ~95 % of
// the probes; here one wave owns a row of C in LDS and ADDS: for k along L's row (in order), the lanes take the entries of
// R's row k and add  L[r, k] R[k, c]  to acc[c] -- distinct columns within a row of R, so no two lanes touch one word, and
// one wave's LDS operations execute in order: every entry is summed in the order of L's row, the same every time.
constexpr int CSR_DENSE_MAX_COLS = 4096;
__global__ __launch_bounds__(64) void k_csr_spgemm_dense(const uint32_t *l_rowptr, const uint32_t *l_cols, const double *l_vals,
                                                        const uint32_t *r_rowptr, const uint32_t *r_cols, const double *r_vals,
                                                        double *c_vals, int nrows, int ncols) {
  __shared__ double acc[CSR_DENSE_MAX_COLS];
  const int lane = threadIdx.x;
  for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
    for (int c = lane; c < ncols; c += 64) acc[c] = 0.;
    for (uint32_t k = l_rowptr[r]; k < l_rowptr[r + 1]; k++) {
      const uint32_t j = l_cols[k];
      const double a = l_vals[k];
      for (uint32_t m = r_rowptr[j] + lane; m < r_rowptr[j + 1]; m += 64) acc[r_cols[m]] += a * r_vals[m];
    }
    for (int c = lane; c < ncols; c += 64) c_vals[(size_t)r * ncols + c] = acc[c];
  }
}

// C = L R on fixed patterns, rows of C with at most `maxc` entries (round 5): ONE WAVE owns a row of C.  Its column list (sorted) and
// an accumulator per entry live in LDS; for k along L's row IN ORDER the lanes take the entries of R's row l_cols[k], find each entry's
// column in C's list (binary search in LDS: ~log2 of the row length probes, against ~80 x 5 probes of GLOBAL memory per entry of C in
// k_csr_spgemm) and add  L[r, k] R[k, c]  to its accumulator.  Columns within a row of R are distinct (no two lanes on one word), one
// wave's LDS operations execute in order and a barrier separates the k steps: every entry of C is summed in the order of L's row, the
// same every time -- the values k_csr_spgemm gives, term for term.  Config 4's p = 1 level (330 k rows, 81 per row): T = A P 8.0 -> ~1 ms.
__global__ __launch_bounds__(64) void k_csr_spgemm_row(const uint32_t *l_rowptr, const uint32_t *l_cols, const double *l_vals,
                                                      const uint32_t *r_rowptr, const uint32_t *r_cols, const double *r_vals,
                                                      const uint32_t *c_rowptr, const uint32_t *c_cols, double *c_vals, int nrows, int maxc) {
  extern __shared__ double sm[];
  double *acc = sm;                                   // [maxc]
  uint32_t *ccol = (uint32_t *)(sm + maxc);           // [maxc]
  const int lane = threadIdx.x;
  for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
    const uint32_t s0 = c_rowptr[r], nc = c_rowptr[r + 1] - s0;
    for (uint32_t s = lane; s < nc; s += 64) { ccol[s] = c_cols[s0 + s]; acc[s] = 0.; }
    __syncthreads();
    if (nc) {
      for (uint32_t k = l_rowptr[r]; k < l_rowptr[r + 1]; k++) {
        const uint32_t j = l_cols[k];
        const double a = l_vals[k];
        for (uint32_t m = r_rowptr[j] + lane; m < r_rowptr[j + 1]; m += 64) {
          const uint32_t c = r_cols[m];
          uint32_t lo = 0, hi = nc;                     // invariant: ccol[lo] <= c < ccol[hi] where the column is present
          while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (ccol[mid] <= c) lo = mid; else hi = mid;
          }
          if (ccol[lo] == c) acc[lo] += a * r_vals[m];  // (an entry of the product outside C's pattern is dropped, as in k_csr_spgemm)
        }
        __syncthreads();
      }
    }
    for (uint32_t s = lane; s < nc; s += 64) c_vals[s0 + s] = acc[s];
    __syncthreads();
  }
}

hipError_t launch_csr_spgemm(const uint32_t *l_rowptr, const uint32_t *l_cols, const double *l_vals, const uint32_t *r_rowptr,
                             const uint32_t *r_cols, const double *r_vals, const uint32_t *c_rowptr, const uint32_t *c_cols, double *c_vals,
                             int nrows, hipStream_t s, int dense_ncols, int max_row_c) {
  if (nrows <= 0) return hipSuccess;
  if (dense_ncols > 0 && dense_ncols <= CSR_DENSE_MAX_COLS) {
    hipLaunchKernelGGL(k_csr_spgemm_dense, dim3((unsigned)std::min(nrows, 8192)), dim3(64), 0, s, l_rowptr, l_cols, l_vals, r_rowptr, r_cols, r_vals,
                       c_vals, nrows, dense_ncols);
    return hipGetLastError();
  }
  if (max_row_c > 0 && max_row_c <= 4096) {     // rows of C fit an LDS accumulator: a wave per row
    int maxc = 64;
    while (maxc < max_row_c) maxc *= 2;
    const size_t lds = (size_t)maxc * 12;
    hipLaunchKernelGGL(k_csr_spgemm_row, dim3((unsigned)std::min(nrows, 32768)), dim3(64), lds, s, l_rowptr, l_cols, l_vals, r_rowptr, r_cols, r_vals,
                       c_rowptr, c_cols, c_vals, nrows, maxc);
    return hipGetLastError();
  }
  const unsigned blocks = (unsigned)std::min<size_t>(((size_t)nrows + 3) / 4, 16384);     // four rows (waves) per workgroup
  hipLaunchKernelGGL(k_csr_spgemm, dim3(blocks), dim3(256), 0, s, l_rowptr, l_cols, l_vals, r_rowptr, r_cols, r_vals, c_rowptr, c_cols,
                     c_vals, nrows);
  return hipGetLastError();
}
// scratch: GJ * GJ doubles; info: one int, zero on entry, 1-based index of the first non-positive pivot otherwise
hipError_t launch_dense_spd_inverse(double *A, int n, double *scratch, int *info, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int nt = (n + GJ - 1) / GJ;
  for (int k0 = 0; k0 < n; k0 += GJ) {
    hipLaunchKernelGGL(k_gj_pivot, dim3(1), dim3(64), 0, s, A, n, k0, scratch, info);
    if (nt > 1) {
      hipLaunchKernelGGL(k_gj_panel, dim3(nt), dim3(GJ, GJ), 0, s, A, n, k0, scratch, 0);
      hipLaunchKernelGGL(k_gj_update, dim3(nt, nt), dim3(GJ, GJ), 0, s, A, n, k0);
      hipLaunchKernelGGL(k_gj_panel, dim3(nt), dim3(GJ, GJ), 0, s, A, n, k0, scratch, 1);
    }
  }
  hipLaunchKernelGGL(k_symmetrize, grid_for((size_t)n * n, 256), dim3(256), 0, s, A, n);
  return hipGetLastError();
}

}  // namespace cps
