// ceed_csr.cpp -- assembled sparse operators (CeedXCsr*): the coarse multigrid level and the aggregation hierarchy
// under it (the reference's FD-coloured AIJ matrix + PCGAMG, src/misc.c:151-183, elasticity.c:457-483,568-585).
#include <thread>

#include "ceed_impl.hpp"

using namespace cps;

struct CeedXCsr_private {
  Ceed ceed = nullptr;
  int nrows = 0, ncols = 0, nnz = 0, ncoo = 0, n_unit = 0;
  uint32_t *d_rowptr = nullptr, *d_cols = nullptr, *d_slotptr = nullptr, *d_perm = nullptr, *d_unit_slot = nullptr,
           *d_diag_slot = nullptr;
  double *d_vals = nullptr;
  // values = product of two other matrices' values on this pattern (CeedXCsrCreateProduct / CeedXCsrUpdate)
  CeedXCsr src = nullptr, src2 = nullptr;
  double *d_gj = nullptr;
  int *d_info = nullptr;
  bool dense = false;       // full pattern, columns ascending: vals is a row-major nrows x nrows matrix
  int max_row = 0;          // longest row of the pattern (products: chooses the kernel of CeedXCsrUpdate)
  uint32_t *d_row_block = nullptr;   // CSR-stream runs of CeedXCsrApply (launch_csr_spmv_stream), cut at the first apply
  int n_row_blocks = 0;
  int refs = 1;             // an operand is kept alive by the products formed from it
  std::vector<int> h_rowptr, h_cols;      // host copy of the pattern (operand of CeedXCsrCreateProduct)
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
  {  // a full pattern with ascending columns may be inverted in place (CeedXCsrInvertDenseSPD): the summed coarsest level of a distributed hierarchy
    bool dense = (long long)nnz == (long long)nrows * nrows && nrows > 0;
    for (int r = 0; dense && r < nrows; r++)
      for (int k = rowptr[r]; k < rowptr[r + 1]; k++) if (cols[k] != k - rowptr[r]) { dense = false; break; }
    A->dense = dense;
  }
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
// runs of consecutive rows for the CSR-stream SpMV: at most 2048 entries and 256 rows each; a longer row alone
static int csr_row_blocks(CeedXCsr A) {
  if (A->d_row_block) return 0;
  if (A->ceed->capturing) return ceed_error("CeedXCsrApply: first apply of a matrix during graph capture; apply it once before recording");
  std::vector<uint32_t> rb(1, 0u);
  int r = 0;
  while (r < A->nrows) {
    int r1 = r + 1;
    long nz = (long)A->h_rowptr[r + 1] - A->h_rowptr[r];
    while (r1 < A->nrows && r1 - r < 256 && nz + ((long)A->h_rowptr[r1 + 1] - A->h_rowptr[r1]) <= 2048) { nz += (long)A->h_rowptr[r1 + 1] - A->h_rowptr[r1]; r1++; }
    rb.push_back((uint32_t)r1);
    r = r1;
  }
  A->n_row_blocks = (int)rb.size() - 1;
  CHK(csr_upload(&A->d_row_block, rb));
  return 0;
}
extern "C" int CeedXCsrApply(CeedXCsr A, CeedVector x, CeedVector y) {
  if (x == y) return ceed_error("CeedXCsrApply: in-place apply is not supported");
  if (x->length < A->ncols || y->length < A->nrows) return ceed_error("CeedXCsrApply: vector shorter than the matrix");
  double *px, *py;
  CHK(vec_dev(x, false, &px)); CHK(vec_dev(y, true, &py));
  if (A->ceed->opt.spmv_stream) {
    CHK(csr_row_blocks(A));
    HIPCHK(launch_csr_spmv_stream(A->d_row_block, A->n_row_blocks, A->d_rowptr, A->d_cols, A->d_vals, px, py, A->ceed->stream));
  } else
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
  CHK(csr_upload(&A->d_rowptr, rp)); CHK(csr_upload(&A->d_cols, cl)); CHK(csr_upload(&A->d_diag_slot, diag));
  HIPCHK(hipMalloc((void **)&A->d_vals, sizeof(double) * (nnz ? nnz : 1)));
  if (vals && nnz) HIPCHK(hipMemcpy(A->d_vals, vals, sizeof(double) * nnz, hipMemcpyHostToDevice));
  else HIPCHK(hipMemset(A->d_vals, 0, sizeof(double) * (nnz ? nnz : 1)));
  *csr = A;
  return 0;
}
// C = L R on fixed patterns.  The pattern of C is worked out here once, row by row (Gustavson, columns ascending; row
// blocks on host threads); CeedXCsrUpdate(C) then recomputes its VALUES on the device from the operands' current values
// (k_csr_spgemm: no term lists, nothing but the three patterns in memory).  `variable` names the operand whose values
// change between updates (0 left, 1 right) -- kept for the callers' documentation: both are read at every update.
// dense != 0: C gets the full pattern (entries the product does not reach stay zero), for CeedXCsrInvertDenseSPD.
// R's columns must be sorted within each row (they are for every matrix this library or scipy builds).
extern "C" int CeedXCsrCreateProduct(CeedXCsr Lm, CeedXCsr Rm, int variable, int dense, CeedXCsr *csr) {
  if (!Lm || !Rm || Lm == Rm || (variable != 0 && variable != 1)) return ceed_error("CeedXCsrCreateProduct: bad operands");
  if (Lm->ncols != Rm->nrows) return ceed_error("CeedXCsrCreateProduct: %d columns times %d rows", Lm->ncols, Rm->nrows);
  const int nrows = Lm->nrows, ncols = Rm->ncols;
  if (dense && nrows != ncols) return ceed_error("CeedXCsrCreateProduct: a dense result must be square");
  for (int j = 0; j < Rm->nrows; j++)
    for (int b = Rm->h_rowptr[j] + 1; b < Rm->h_rowptr[j + 1]; b++)
      if (Rm->h_cols[b] <= Rm->h_cols[b - 1]) return ceed_error("CeedXCsrCreateProduct: the right operand's columns are not sorted in row %d", j);
  struct Part { std::vector<uint32_t> len, cl; };
  const int nthreads = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, nrows / 256 + 1}));
  std::vector<Part> parts((size_t)nthreads);
  auto work = [&](int t) {
    Part &pt = parts[(size_t)t];
    const int r0 = (int)((long long)nrows * t / nthreads), r1 = (int)((long long)nrows * (t + 1) / nthreads);
    std::vector<int> mark((size_t)ncols, -1);
    std::vector<uint32_t> row;
    for (int i = r0; i < r1; i++) {
      row.clear();
      if (dense) { for (int c = 0; c < ncols; c++) row.push_back((uint32_t)c); }
      else {
        for (int a = Lm->h_rowptr[i]; a < Lm->h_rowptr[i + 1]; a++) {
          const int j = Lm->h_cols[a];
          for (int b = Rm->h_rowptr[j]; b < Rm->h_rowptr[j + 1]; b++) {
            const int c = Rm->h_cols[b];
            if (mark[(size_t)c] != i) { mark[(size_t)c] = i; row.push_back((uint32_t)c); }
          }
        }
        std::sort(row.begin(), row.end());
      }
      pt.cl.insert(pt.cl.end(), row.begin(), row.end());
      pt.len.push_back((uint32_t)row.size());
    }
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
  }
  size_t tot_e = 0;
  for (const Part &pt : parts) tot_e += pt.cl.size();
  if (tot_e >= 0x7FFFFFFFull) return ceed_error("CeedXCsrCreateProduct: more than 2^31 entries");
  std::vector<uint32_t> rp((size_t)nrows + 1, 0u), cl;
  cl.reserve(tot_e);
  {
    int i = 0;
    for (Part &pt : parts) {
      for (uint32_t len : pt.len) { rp[(size_t)i + 1] = rp[(size_t)i] + len; i++; }
      cl.insert(cl.end(), pt.cl.begin(), pt.cl.end());
      std::vector<uint32_t>().swap(pt.cl); std::vector<uint32_t>().swap(pt.len);
    }
  }
  const int nnz = (int)cl.size();
  std::vector<uint32_t> diag((size_t)nrows, 0xFFFFFFFFu);
  for (int r = 0; r < nrows; r++)
    for (uint32_t k = rp[r]; k < rp[r + 1]; k++) if ((int)cl[k] == r) diag[r] = k;
  CeedXCsr A = new CeedXCsr_private;
  A->ceed = Lm->ceed; ceed_ref(A->ceed);
  A->nrows = nrows; A->ncols = ncols; A->nnz = nnz; A->dense = dense != 0;
  A->h_rowptr.assign(rp.begin(), rp.end()); A->h_cols.assign(cl.begin(), cl.end());
  CHK(csr_upload(&A->d_rowptr, rp)); CHK(csr_upload(&A->d_cols, cl)); CHK(csr_upload(&A->d_diag_slot, diag));
  HIPCHK(hipMalloc((void **)&A->d_vals, sizeof(double) * (nnz ? nnz : 1)));
  HIPCHK(hipMemset(A->d_vals, 0, sizeof(double) * (nnz ? nnz : 1)));
  for (int r = 0; r < nrows; r++) A->max_row = std::max(A->max_row, (int)(rp[(size_t)r + 1] - rp[(size_t)r]));
  A->src = Lm; Lm->refs++;
  A->src2 = Rm; Rm->refs++;
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
  if (!A->src || !A->src2) return ceed_error("CeedXCsrUpdate: not a product (CeedXCsrCreateProduct)");
  CeedXCsr Lm = A->src, Rm = A->src2;
  HIPCHK(launch_csr_spgemm(Lm->d_rowptr, Lm->d_cols, Lm->d_vals, Rm->d_rowptr, Rm->d_cols, Rm->d_vals, A->d_rowptr, A->d_cols, A->d_vals,
                           A->nrows, A->ceed->stream, A->dense ? A->ncols : 0, A->ceed->opt.spgemm_row ? A->max_row : 0));
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
  for (uint32_t *p : {A->d_rowptr, A->d_cols, A->d_slotptr, A->d_perm, A->d_unit_slot, A->d_diag_slot, A->d_row_block})
    if (p) (void)hipFree(p);
  for (double *p : {A->d_vals, A->d_gj}) if (p) (void)hipFree(p);
  if (A->d_info) (void)hipFree(A->d_info);
  CeedXCsr src = A->src, src2 = A->src2;
  ceed_unref(A->ceed);
  delete A;
  if (src) (void)CeedXCsrDestroy(&src);
  if (src2) (void)CeedXCsrDestroy(&src2);
  return 0;
}
