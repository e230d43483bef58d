// ceed_restriction.cpp -- CeedElemRestriction (offsets and strided) and the transpose maps the deterministic, atomic-free
// scatter is built on (set-up time, host).  Reference: CreateRestrictionPlex -> CeedElemRestrictionCreate
// (src/setuplibceed.c:194-240), CeedElemRestrictionCreateStrided with CEED_STRIDES_BACKEND (:304-318).
#include "ceed_impl.hpp"

using namespace cps;

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
  if (r->d_int_off) (void)hipFree(r->d_int_off);
  r->csr.release();
  r->csr_shell.release();
  for (PipeMap *p : r->pipes) {
    for (uint32_t *q : {p->d_rowptr, p->d_cols, p->d_node_off}) if (q) (void)hipFree(q);
    delete p;
  }
  ceed_unref(r->ceed);
  delete r;
  return 0;
}

// Build a transpose map (setup time, host): counting sort over the L-vector.  With `prio`
// (one byte per L-vector entry, tested at each node's component-0 offset) the flagged nodes
// come first.
// `skipP` > 0 (elemsize == skipP^3): nodes interior to an element are left out of the map -- the fused kernel
// stores them itself (FusedGradArgs::direct); the caller has checked rstr_interior_private().
int build_csr(CeedElemRestriction r, CsrMap &M, const unsigned char *prio, int skipP) {
  if (M.built) return 0;
  if (r->ceed->capturing)
    return ceed_error("first apply of an operator during graph capture: its restriction's transpose map is built on the host; "
                      "apply the operator once before recording");
  const size_t n = r->h_offsets.size();
  std::vector<uint32_t> cnt((size_t)r->lsize + 1, 0u);
  for (size_t i = 0; i < n; i++) cnt[(size_t)r->h_offsets[i]]++;
  M.nskipped = 0;
  if (skipP > 0)
    for (size_t i = 0; i < n; i++)
      if (node_is_element_interior((int)(i % (size_t)r->elemsize), skipP)) { cnt[(size_t)r->h_offsets[i]] = 0; M.nskipped++; }   // stored by the fused kernel itself
  std::vector<uint32_t> slot((size_t)r->lsize, 0xFFFFFFFFu);
  std::vector<uint32_t> &rowptr = M.h_rowptr, &cols = M.h_cols;
  M.h_node_off.clear(); rowptr.clear();
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
  std::vector<uint32_t> cursor(rowptr.begin(), rowptr.end() - 1);
  cols.assign(rowptr.back() ? rowptr.back() : 1, 0u);
  for (size_t i = 0; i < n; i++) {  // element order => each node's contributors are sorted by element
    const uint32_t sl = slot[(size_t)r->h_offsets[i]];
    if (sl == 0xFFFFFFFFu) continue;
    // E position: e * elemsize + n, or in the shell-only E-vector of the direct-store mode e * shell size + shell rank
    const size_t e = i / (size_t)r->elemsize; const int ln = (int)(i % (size_t)r->elemsize);
    cols[cursor[sl]++] = skipP > 0 ? (uint32_t)(e * (size_t)evec_block_records(skipP) + (size_t)node_shell_rank(ln, skipP)) : (uint32_t)i;
  }
  M.nnodes = nn;
  // every L-vector entry is written by the assembly (or, for the skipped nodes, by the fused kernel)
  M.full_cover = ((size_t)nn + (size_t)M.nskipped) * (size_t)r->ncomp == (size_t)r->lsize;
  HIPCHK(hipMalloc((void **)&M.d_rowptr, sizeof(uint32_t) * (nn + 1)));
  HIPCHK(hipMalloc((void **)&M.d_cols, sizeof(uint32_t) * cols.size()));
  HIPCHK(hipMalloc((void **)&M.d_node_off, sizeof(uint32_t) * (nn ? nn : 1)));
  HIPCHK(hipMemcpy(M.d_rowptr, rowptr.data(), sizeof(uint32_t) * (nn + 1), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(M.d_cols, cols.data(), sizeof(uint32_t) * cols.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(M.d_node_off, M.h_node_off.data(), sizeof(uint32_t) * nn, hipMemcpyHostToDevice));
  M.built = true;
  return 0;
}
// Are the element-interior nodes (local index 0 < i,j,k < P-1) of an offsets restriction private to their
// element?  True for every conforming mesh; checked because offsets are caller data.
bool rstr_interior_private(CeedElemRestriction r, int P) {
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

// Node offsets of the element-interior nodes (the ones the fused kernel stores itself), [elem][(P-2)^3] in element-local order.
int build_interior_list(CeedElemRestriction r, int P) {
  if (r->d_int_off || P < 3) return 0;
  if (r->ceed->capturing) return ceed_error("first apply of an operator during graph capture: apply it once before recording");
  const int m = (P - 2) * (P - 2) * (P - 2);
  std::vector<uint32_t> lst((size_t)r->nelem * m);
  size_t k = 0;
  for (CeedInt e = 0; e < r->nelem; e++)
    for (int n = 0; n < r->elemsize; n++)
      if (node_is_element_interior(n, P)) lst[k++] = (uint32_t)r->h_offsets[(size_t)e * r->elemsize + n];
  HIPCHK(hipMalloc((void **)&r->d_int_off, sizeof(uint32_t) * (lst.size() ? lst.size() : 1)));
  HIPCHK(hipMemcpy(r->d_int_off, lst.data(), sizeof(uint32_t) * lst.size(), hipMemcpyHostToDevice));
  r->int_per_elem = m;
  return 0;
}

// Segments of the pipelined assembly: element ranges whose group counts are whole rounds of the fused kernel's persistent
// waves (`waves` per launch) where the mesh is large enough for that -- a launch then ends with every wave finishing its
// last group at about the same time -- and the rows of the map sorted by the segment of their last contributor.
// Maps are cached per restriction and never replaced: a recorded graph, or a second operator with another quadrature on
// the same restriction, keeps valid pointers.  A launch too small to pipeline gets a map with nseg = 1 and NO copies.
int get_pipe(CeedElemRestriction r, const CsrMap &M, int E, int per_elem, int req_seg_in, int waves, int mb, PipeMap **out) {
  for (PipeMap *p : r->pipes)
    if (p->E == E && p->req_seg == req_seg_in && p->waves == waves && p->mb == mb && p->base == (const void *)&M) { *out = p; return 0; }
  const CeedOptions &opt = r->ceed->opt;
  int req_seg = req_seg_in;
  const int ngroups = (r->nelem + E - 1) / E;
  // at least `min_rounds` rounds per segment, else fewer segments (down to one: the caller then takes the serial path)
  const int min_rounds = opt.pipe_min_rounds;
  // Below ~20 rounds of the persistent waves the fixed cost of the form (fork and join of the second stream, the summing
  // kernels competing with the fused kernel for memory: ~40 us at p = 4) exceeds what is hidden: measured -3 % at 24 rounds
  // (99 000 hexes, p = 4), +7 % at 11 rounds (44 928 hexes) -- such launches keep the serial form.
  if (min_rounds > 0 && ngroups < opt.pipe_min_total_rounds * std::max(waves, 1)) req_seg = 1;
  // Segments asked for = 0: one per `mb` MB of E-vector -- a segment boundary costs ~10 us, and the smaller a segment the more of
  // its E-vector is still in the 256 MB last-level cache when its rows are summed (config 5, 1.4 GB of E-vector: 4.27 ms serial,
  // 4.00 with 3 segments, 3.57 with 8, 3.42 with 12-16 in round 2).  Rounds 2-3 used ~90 MB for every kernel (3 segments at config
  // 4); with round 4's faster fused kernel the finite-strain applies measure best at ~160 MB (config 4: 2 segments, -1.5 %; twice its
  // mesh: 3, -2 %; the whole of config 5: 9, +-0), the cheaper kernels (hyperSS, linElas: a shorter fused kernel to hide the same
  // rows behind) still at ~90 (profiles/r04_ab_experiments.txt item 14).  The caller passes the figure (apply_fused_grad).
  else if (req_seg == 0) req_seg = std::max(2, std::min(16, (int)((double)r->nelem * per_elem * 24. / (1e6 * std::max(mb, 1)) + 0.5)));
  int nseg = min_rounds > 0 ? std::max(1, std::min(req_seg, ngroups / (min_rounds * std::max(waves, 1)))) : std::min(req_seg, std::max(1, ngroups));
  if (nseg >= 2 && r->ceed->capturing) { *out = nullptr; return 0; }   // cold map while recording: the caller takes the serial path (its map exists)
  PipeMap *Gp = new PipeMap;
  PipeMap &G = *Gp;
  G.E = E; G.req_seg = req_seg_in; G.waves = waves; G.mb = mb; G.base = (const void *)&M;
  if (opt.pipe_debug) fprintf(stderr, "get_pipe: %d elements, E %d, %d segments asked, %d waves -> %d\n", r->nelem, E, req_seg_in, waves, nseg);
  if (nseg < 2) { G.nseg = 1; G.built = true; r->pipes.push_back(Gp); *out = Gp; return 0; }
  // Boundaries are laid out FROM THE END in whole rounds of the waves: the last segment (whose rows are summed with nothing
  // to hide behind) is `last_rounds` rounds, the others share the rest equally in whole rounds, and the odd remainder of the
  // mesh lands in the FIRST segment, where the next fused kernel fills the chip behind its ragged last round.
  G.elem_bound.assign(1, 0);
  const int last_rounds = opt.pipe_last_rounds;
  const long total_rounds = ngroups / std::max(waves, 1);
  std::vector<long> gb;        // group boundaries, descending
  if (min_rounds > 0 && last_rounds > 0 && total_rounds >= last_rounds + (long)(nseg - 1) * min_rounds) {
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
  const std::vector<uint32_t> &rowptr = M.h_rowptr, &cols = M.h_cols;
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
  if (opt.pipe_debug)
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
  r->pipes.push_back(Gp);
  *out = Gp;
  return 0;
}
