"""Aggregation multigrid under the assembled p = 1 level: the coarse solve of the p-multigrid V-cycle.

The reference hands the assembled coarse Jacobian to PCGAMG and applies ONE cycle of it per outer iteration
(KSPPREONLY + PCGAMG, elasticity.c:568-585).  This module builds the same kind of hierarchy, two levels deep:

* aggregates of the node graph of the assembled matrix (a root and its neighbours, leftovers join the
  neighbouring aggregate they touch most): ~40 nodes per aggregate on a hexahedral mesh;
* tentative prolongation P0 from the six rigid-body modes of each aggregate (translations and rotations about
  the aggregate's centroid, rows of constrained dofs zeroed, orthonormalised per aggregate by an SVD that also
  drops dependent columns) -- PCGAMG's near-null space for elasticity;
* smoothed prolongation P = (I - w D^-1 A) P0, computed ONCE from the first Jacobian of the solve and kept
  (P defines the coarse space; the Galerkin matrix below uses the current Jacobian every Newton step, so the
  coarse correction stays an exact projection.  Outer iteration counts with the frozen and with a refreshed P
  are equal on BASELINE config 3's mesh: DESIGN.md);
* per Newton step on the device, through `CeedXCsr*` of include/ceed.h: T = A P and A_c = P^T T as fixed linear
  combinations of values (`CeedXCsrCreateProduct` / `CeedXCsrUpdate`: the library builds the term lists once), then
  the in-place inverse of the dense A_c (`CeedXCsrInvertDenseSPD`, ~10^3 rows);
* per cycle: r_c = P^T r,  x_c = A_c^-1 r_c,  x += P x_c  -- three `CeedXCsrApply` launches between the level's own
  Chebyshev pre- and post-smoothing (solver.py).

Everything the cycle launches is on the Ceed's stream and recordable into the V-cycle graph.  The host part
(this file) is numpy / scipy on patterns and runs once per solve.
"""
from __future__ import annotations

import time

import numpy as np

from . import ceed as cd


def aggregate_nodes(indptr: np.ndarray, indices: np.ndarray) -> tuple[np.ndarray, int]:
    """Greedy aggregation of a symmetric node graph (CSR, self-loops allowed): pass 1 makes a root and ALL its
    neighbours an aggregate when none of them is taken; pass 2 attaches every remaining node to the aggregate
    most of its neighbours belong to (decided on the pass-1 state, so the result does not depend on the order
    inside pass 2); nodes without any aggregated neighbour become aggregates of their own."""
    m = indptr.size - 1
    agg = -np.ones(m, dtype=np.int64)
    na = 0
    for i in range(m):
        if agg[i] >= 0:
            continue
        nb = indices[indptr[i]:indptr[i + 1]]
        if np.all(agg[nb] < 0):
            agg[nb] = na
            agg[i] = na
            na += 1
    out = agg.copy()
    for i in np.nonzero(agg < 0)[0]:
        c = agg[indices[indptr[i]:indptr[i + 1]]]
        c = c[c >= 0]
        if c.size:
            out[i] = np.bincount(c).argmax()
    for i in np.nonzero(out < 0)[0]:
        out[i] = na
        na += 1
    return out, na


def rigid_body_prolongation(agg: np.ndarray, na: int, coords: np.ndarray, constrained: np.ndarray):
    """Tentative prolongation (scipy CSR, 3 * nnodes rows): per aggregate the orthonormalised rigid-body modes."""
    import scipy.sparse as sp
    order = np.argsort(agg, kind="stable")
    start = np.searchsorted(agg[order], np.arange(na + 1))
    rows, cols, vals = [], [], []
    nc = 0
    for a in range(na):
        nodes = order[start[a]:start[a + 1]]
        if nodes.size == 0:
            continue
        d = coords[nodes] - coords[nodes].mean(axis=0)
        B = np.zeros((3 * nodes.size, 6))
        B[0::3, 0] = 1.0; B[1::3, 1] = 1.0; B[2::3, 2] = 1.0
        B[0::3, 4] = d[:, 2]; B[0::3, 5] = -d[:, 1]
        B[1::3, 3] = -d[:, 2]; B[1::3, 5] = d[:, 0]
        B[2::3, 3] = d[:, 1]; B[2::3, 4] = -d[:, 0]
        dofs = (3 * nodes[:, None] + np.arange(3)).reshape(-1)
        B[constrained[dofs]] = 0.0
        U, sv, _ = np.linalg.svd(B, full_matrices=False)
        k = int((sv > 1e-8 * sv[0]).sum()) if sv[0] > 0.0 else 0
        for j in range(k):
            rows.append(dofs); cols.append(np.full(dofs.size, nc + j)); vals.append(U[:, j])
        nc += k
    if nc == 0:
        raise ValueError("the aggregation left no coarse degree of freedom (is every dof constrained?)")
    P0 = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(constrained.size, nc))
    P0.eliminate_zeros()
    return P0


class AggregationAMG:
    """Two-level smoothed aggregation under an `AssembledLevel` (see the module docstring)."""

    def __init__(self, asm, prolongator_damping: float = 0.66, verbose: bool = False, max_coarse_dofs: int = 4096):
        self.asm, self.ceed = asm, asm.ceed
        self.damping, self.verbose, self.max_coarse_dofs = prolongator_damping, verbose, max_coarse_dofs
        self.P = self.Pt = self.T = self.Ac = None
        self.rc = self.xc = None
        self.nc = 0
        self.setup_seconds = 0.0
        self.info = {}

    # ---- once per solve: aggregates, prolongation, term lists ------------------------------------------------------
    def build(self):
        import scipy.sparse as sp
        t0 = time.perf_counter()
        asm, c = self.asm, self.ceed
        n = asm.nrows
        lv = asm.p.levels[asm.level]
        constrained = lv.mask != 0
        rowptr, cols = asm.rowptr, asm.cols
        A = sp.csr_matrix((asm.csr.values(c), cols, rowptr), shape=(n, n))
        a_row = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
        # node graph without the fully constrained nodes (their rows are identity rows)
        nn = n // 3
        node_free = ~constrained.reshape(nn, 3).all(axis=1)
        G = sp.csr_matrix((np.ones(a_row.size), (a_row // 3, cols // 3)), shape=(nn, nn)).tocsr()
        fn = np.nonzero(node_free)[0]
        Gf = G[fn][:, fn].tocsr()
        agg_f, na = aggregate_nodes(Gf.indptr, Gf.indices)
        agg = np.full(nn, na, dtype=np.int64)          # constrained nodes: a dummy aggregate without columns
        agg[fn] = agg_f
        P0 = rigid_body_prolongation(agg, na, lv.dofmap.node_coords, constrained)
        # smoothed prolongation from the current (first) Jacobian: P = (I - w D^-1 A) P0, w = damping * 4/3 / lambda_max
        dinv = 1.0 / A.diagonal()
        x = np.random.default_rng(77).uniform(-1.0, 1.0, n) * ~constrained
        lam = 1.0
        for _ in range(30):                      # (sums, not BLAS norms: a threaded BLAS call leaves its pool spinning)
            y = dinv * (A @ x)
            lam = float(np.sqrt(np.square(y).sum() / np.square(x).sum()))
            x = y / np.sqrt(np.square(y).sum())
        lam *= 1.05
        P = (P0 - (self.damping * 4.0 / 3.0 / lam) * (sp.diags(dinv) @ (A @ P0))).tocsr()
        P.sort_indices()
        nc = P.shape[1]
        if nc > self.max_coarse_dofs:
            # two levels only: the coarsest matrix is inverted densely every Newton step (n^3: 1.3 ms at 1 080 rows, ~0.1 s at
            # 4 096); a mesh this large needs a third level (not built) -- the Chebyshev coarse solve takes any size
            raise ValueError(f"the aggregation leaves {nc} coarse dofs (limit {self.max_coarse_dofs}: the coarsest level is inverted "
                             f"densely); use coarse='assembled' for this mesh or raise max_coarse_dofs")
        Pt = P.T.tocsr()
        Pt.sort_indices()
        self.P = cd.Csr.rect(c, n, nc, P.indptr, P.indices, P.data)
        self.Pt = cd.Csr.rect(c, nc, n, Pt.indptr, Pt.indices, Pt.data)
        # Galerkin product with fixed patterns: T = A P (A varies), A_c = P^T T (T varies), A_c stored dense
        self.T = cd.Csr.product(asm.csr, self.P, variable=0)
        self.Ac = cd.Csr.product(self.Pt, self.T, variable=1, dense=True)
        self.rc, self.xc = c.vector(nc).set_value(0.0), c.vector(nc).set_value(0.0)
        self.nc = nc
        self.info = dict(aggregates=int(na), coarse_dofs=int(nc), nodes_per_aggregate=float(fn.size) / max(na, 1),
                         prolongation_entries_per_row=float(P.nnz) / n, galerkin_entries=int(self.T.nnz),
                         lambda_max=lam, build_seconds=time.perf_counter() - t0)
        if self.verbose:
            print("AggregationAMG:", self.info, flush=True)

    # ---- every Newton step: the Galerkin matrix of the current Jacobian and its inverse ----------------------------
    def setup(self):
        if self.P is None:
            self.build()
        self.T.update()
        self.Ac.update()
        self.Ac.invert_dense_spd()

    # ---- the coarse correction of one cycle: x += P A_c^-1 P^T r ---------------------------------------------------
    def restrict(self, r: cd.Vector):
        self.Pt.apply(r, self.rc)

    def solve_coarsest(self):
        self.Ac.apply(self.rc, self.xc)

    def prolong(self, z: cd.Vector):
        self.P.apply(self.xc, z)

    def destroy(self):
        for o in (self.Ac, self.T, self.Pt, self.P, self.rc, self.xc):
            if o is not None:
                o.destroy()
        self.P = self.Pt = self.T = self.Ac = self.rc = self.xc = None

