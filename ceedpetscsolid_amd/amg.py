"""Aggregation multigrid under the assembled p = 1 level: the coarse solve of the p-multigrid V-cycle.

The reference hands the assembled coarse Jacobian to PCGAMG and applies ONE cycle of it per outer iteration
(KSPPREONLY + PCGAMG, elasticity.c:568-585).  This module builds the same kind of hierarchy:

* aggregates of the node graph of a level's matrix (a root and its neighbours, leftovers join the neighbouring
  aggregate they touch most): ~40 nodes per aggregate on a hexahedral mesh, ~25 aggregates per aggregate above;
* tentative prolongation P0 from the near-null space B (level 0: the six rigid-body modes, translations and
  rotations about the aggregate's centroid, rows of constrained dofs zeroed -- PCGAMG's near-null space for
  elasticity): per aggregate B restricted to it is orthonormalised by an SVD that also drops dependent columns;
  U spans the aggregate's columns of P0, S V^T is the aggregate's block of the next level's B;
* smoothed prolongation P = (I - w D^-1 A) P0, computed ONCE from the first Jacobian of the solve and kept
  (P defines the coarse space; the Galerkin matrices below use the current Jacobian every Newton step, so every
  coarse correction stays an exact projection.  Outer iteration counts with the frozen and with a refreshed P
  are equal on BASELINE config 3's mesh: DESIGN.md);
* levels are added until one has at most ``max_coarse_dofs`` rows (default 1 500): that one is stored dense and
  inverted.  BASELINE config 3 (7 198 coarse nodes) gets two levels, 21 594 -> 1 080;
* per Newton step on the device, through `CeedXCsr*` of include/ceed.h: T = A P and A_next = P^T T per level as
  fixed linear combinations of values (`CeedXCsrCreateProduct` / `CeedXCsrUpdate`: the library builds the term
  lists once), the diagonal and a 10-step Lanczos bound for the smoother of every intermediate level (scalars
  on the device), then the in-place inverse of the dense last level (`CeedXCsrInvertDenseSPD`);
* per cycle and level: Chebyshev pre-smoothing, r_c = P^T r, the next level's cycle (or x_c = A_c^-1 r_c),
  x += P x_c, Chebyshev post-smoothing.  Level 0's smoothing is the solver's own (solver.py).

Everything the cycle launches is on the Ceed's stream and recordable into the V-cycle graph.  The host part
(this file) is numpy / scipy on patterns and runs once per solve.
"""
from __future__ import annotations

import ctypes as C
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


def rigid_body_modes(coords: np.ndarray, constrained: np.ndarray) -> np.ndarray:
    """Near-null space of the elasticity operator on the nodes: translations and rotations (about the origin; the
    aggregates re-centre them), rows of constrained dofs zeroed.  Shape (3 * nnodes, 6)."""
    n = coords.shape[0]
    B = np.zeros((3 * n, 6))
    B[0::3, 0] = 1.0; B[1::3, 1] = 1.0; B[2::3, 2] = 1.0
    B[0::3, 4] = coords[:, 2]; B[0::3, 5] = -coords[:, 1]
    B[1::3, 3] = -coords[:, 2]; B[1::3, 5] = coords[:, 0]
    B[2::3, 3] = coords[:, 1]; B[2::3, 4] = -coords[:, 0]
    B[constrained] = 0.0
    return B


def tentative_prolongation(agg: np.ndarray, na: int, dof_ptr: np.ndarray, B: np.ndarray):
    """Per aggregate a (nodes with agg == a; ids >= na are left out): B restricted to the aggregate's dofs = U S V^T;
    the columns of U with a singular value above 1e-8 of the largest are the aggregate's columns of P0, S V^T its rows
    of the next level's near-null space.  ``dof_ptr[i] .. dof_ptr[i + 1]`` are the dofs of node i.
    Returns P0 (scipy CSR), the next B, and the column offsets of the aggregates (the next level's dof_ptr)."""
    import scipy.sparse as sp
    order = np.argsort(agg, kind="stable")
    start = np.searchsorted(agg[order], np.arange(na + 1))
    rows, cols, vals, Bc, col_ptr = [], [], [], [], [0]
    nc = 0
    for a in range(na):
        nodes = order[start[a]:start[a + 1]]
        dofs = np.concatenate([np.arange(dof_ptr[i], dof_ptr[i + 1]) for i in nodes]) if nodes.size else np.zeros(0, dtype=np.int64)
        if dofs.size:
            U, sv, Vt = np.linalg.svd(B[dofs], full_matrices=False)
            k = int((sv > 1e-8 * sv[0]).sum()) if sv[0] > 0.0 else 0
            for j in range(k):
                rows.append(dofs); cols.append(np.full(dofs.size, nc + j)); vals.append(U[:, j])
            if k:
                Bc.append(sv[:k, None] * Vt[:k])
            nc += k
        col_ptr.append(nc)
    if nc == 0:
        raise ValueError("the aggregation left no coarse degree of freedom (is every dof constrained?)")
    P0 = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(B.shape[0], nc))
    P0.eliminate_zeros()
    return P0, np.concatenate(Bc, axis=0), np.asarray(col_ptr, dtype=np.int64)


def rigid_body_prolongation(agg: np.ndarray, na: int, coords: np.ndarray, constrained: np.ndarray):
    """Tentative prolongation of level 0 (scipy CSR, 3 * nnodes rows): per aggregate the orthonormalised rigid-body modes."""
    return tentative_prolongation(agg, na, 3 * np.arange(coords.shape[0] + 1, dtype=np.int64), rigid_body_modes(coords, constrained))[0]


class _Level:
    """One transfer of the hierarchy: A (n x n, values change per Newton step) -> A_next = P^T A P (nc x nc)."""
    def __init__(self):
        self.A = self.P = self.Pt = self.T = self.Anext = None
        self.dd = None              # several ranks, first transfer: the rank's Galerkin contribution and its place in the summed matrix
        self.n = self.nc = 0
        self.dense_next = False
        self.info = {}
        # smoother data of THIS level's matrix (levels >= 1 only; level 0 is smoothed by the solver)
        self.v = {}
        self.x0 = None
        self.emax = 1.0


class AggregationAMG:
    """Smoothed-aggregation hierarchy under an `AssembledLevel` (see the module docstring)."""

    def __init__(self, asm, prolongator_damping: float = 0.66, verbose: bool = False, max_coarse_dofs: int = 1500,
                 max_levels: int = 6, smooth_its: int = 3, smooth_ratio: float = 10.0, coarse_cycles: int = 1, dist_halo=None):
        """``dist_halo`` (several ranks): the HaloExchange of the assembled level.  The level's matrix then is the rank's OWN
        (its elements' sum, interface rows partial -- the additive piece A_r of A = sum_r R_r^T A_r R_r), the first transfer of
        the hierarchy is DISTRIBUTED (`_first_transfer_distributed`) and everything below it is small and replicated."""
        self.dist = dist_halo if (dist_halo is not None and dist_halo.world > 1) else None
        self.asm, self.ceed, self.L = asm, asm.ceed, asm.ceed.L
        self.damping, self.verbose, self.max_coarse_dofs, self.max_levels = prolongator_damping, verbose, max_coarse_dofs, max_levels
        self.smooth_its, self.smooth_ratio, self.coarse_cycles = smooth_its, smooth_ratio, coarse_cycles
        self.levels: list[_Level] = []
        self.rc = self.xc = None
        self.nc = 0
        self._scal = None
        self.info = {}

    # two-level accessors (tests, older callers)
    @property
    def P(self): return self.levels[0].P if self.levels else None
    @property
    def Pt(self): return self.levels[0].Pt if self.levels else None
    @property
    def T(self): return self.levels[0].T if self.levels else None
    @property
    def Ac(self): return self.levels[0].Anext if self.levels else None

    # ---- once per solve: aggregates, prolongations, product patterns -------------------------------------------------
    def _transfer(self, A_csr, A_host, dof_ptr, B, free_node, dense_limit):
        """One level: aggregate the node graph of A (nodes = runs of dofs given by dof_ptr; nodes with free_node False are
        left out), tentative + smoothed prolongation, the two products.  Returns the _Level, the next B and dof_ptr."""
        import scipy.sparse as sp
        c = self.ceed
        n = A_host.shape[0]
        nn = dof_ptr.size - 1
        node_of = np.searchsorted(dof_ptr, np.arange(n), side="right") - 1
        a_row = np.repeat(np.arange(n, dtype=np.int64), np.diff(A_host.indptr))
        G = sp.csr_matrix((np.ones(a_row.size), (node_of[a_row], node_of[A_host.indices])), shape=(nn, nn)).tocsr()
        fn = np.nonzero(free_node)[0]
        Gf = G[fn][:, fn].tocsr()
        t_ag = time.perf_counter()
        agg_f, na = aggregate_nodes(Gf.indptr, Gf.indices)
        t_ag = time.perf_counter() - t_ag
        agg = np.full(nn, na, dtype=np.int64)          # left-out nodes: a dummy aggregate without columns
        agg[fn] = agg_f
        t_p0 = time.perf_counter()
        P0, Bn, col_ptr = tentative_prolongation(agg, na, dof_ptr, B)
        t_p0 = time.perf_counter() - t_p0
        t_sm = time.perf_counter()
        # smoothed prolongation from the current (first) Jacobian: P = (I - w D^-1 A) P0, w = damping * 4/3 / lambda_max
        diag = A_host.diagonal()
        dinv = np.where(diag != 0.0, 1.0 / np.where(diag != 0.0, diag, 1.0), 0.0)
        x = np.random.default_rng(77).uniform(-1.0, 1.0, n) * (np.abs(B).sum(axis=1) > 0.0)
        lam = 1.0
        for _ in range(30):                      # (sums, not BLAS norms: a threaded BLAS call leaves its pool spinning)
            y = dinv * (A_host @ x)
            lam = float(np.sqrt(np.square(y).sum() / np.square(x).sum()))
            x = y / np.sqrt(np.square(y).sum())
        lam *= 1.05
        P = (P0 - (self.damping * 4.0 / 3.0 / lam) * (sp.diags(dinv) @ (A_host @ P0))).tocsr()
        P.sort_indices()
        Pt = P.T.tocsr()
        Pt.sort_indices()
        nc = P.shape[1]
        lv = _Level()
        lv.A, lv.n, lv.nc = A_csr, n, nc
        lv.P = cd.Csr.rect(c, n, nc, P.indptr, P.indices, P.data)
        lv.Pt = cd.Csr.rect(c, nc, n, Pt.indptr, Pt.indices, Pt.data)
        lv.dense_next = nc <= dense_limit
        t_sm = time.perf_counter() - t_sm
        # Galerkin product on fixed patterns: T = A P (A varies), A_next = P^T T (T varies)
        t_pr = time.perf_counter()
        lv.T = cd.Csr.product(A_csr, lv.P, variable=0)
        lv.Anext = cd.Csr.product(lv.Pt, lv.T, variable=1, dense=lv.dense_next)
        t_pr = time.perf_counter() - t_pr
        lv.info = dict(rows=int(n), aggregates=int(na), coarse_dofs=int(nc), nodes_per_aggregate=float(fn.size) / max(na, 1),
                       prolongation_entries_per_row=float(P.nnz) / n, galerkin_entries=int(lv.T.nnz), lambda_max=lam,
                       next_is_dense=bool(lv.dense_next),
                       seconds=dict(aggregate=t_ag, tentative=t_p0, smooth_and_upload=t_sm, product_patterns=t_pr),
                       device_bytes=int(12 * (P.nnz + Pt.nnz) + 12 * lv.T.nnz + 12 * lv.Anext.nnz),
                       distributed_bytes=int(12 * (P.nnz + Pt.nnz) + 12 * lv.T.nnz), replicated_bytes=int(12 * lv.Anext.nnz))
        return lv, Bn, col_ptr

    # ---- several ranks: the first transfer, distributed -------------------------------------------------------------------
    def _update_level(self, lv):
        """The Galerkin matrix of transfer lv for the current Jacobian: T = A P, A_next = P^T T on the device; on several ranks the
        first transfer's product is this rank's CONTRIBUTION P_r^T A_r P_r, summed over the ranks into the replicated matrix."""
        lv.T.update()
        if getattr(lv, "dd", None) is None:
            lv.Anext.update()
            return
        d = lv.dd
        d["local"].update()
        # the rank's values into their places of the union pattern, summed over the ranks, assembled: all on the vectors' own memory
        # (device tensors on a GPU; RCCL through the library where it has a communicator, torch.distributed otherwise)
        self.L.chk(self.L.lib.CeedXCsrGetValues(d["local"].h, d["vals"].h))
        if d["vals"].t.device.type == "cuda":
            self.ceed.synchronize()
        d["coo"].t.zero_()
        d["coo"].t.index_copy_(0, d["slot_t"], d["vals"].t[:d["slot_t"].numel()])
        if d["coo"].t.device.type == "cuda":
            d["coo"].set_device_pointer(d["coo"].t.data_ptr())
        self._allreduce(d["coo"])
        lv.Anext.assemble(d["coo"])

    def _allreduce_np(self, a: np.ndarray) -> np.ndarray:
        import torch
        import torch.distributed as dist
        h = self.dist
        on_dev = dist.get_backend(h.group) == "nccl"
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
        if on_dev:
            t = t.to(h.device)
        dist.all_reduce(t, group=h.group)
        return t.cpu().numpy()

    def _allreduce(self, v: cd.Vector):
        """Sum of a (small, coarse) vector over the ranks, in place: RCCL on the Ceed's stream where the library has a communicator
        (recordable), else torch.distributed on the tensor behind the vector."""
        if self.ceed.comm_size()[0] > 1:
            n = C.c_int()
            self.L.chk(self.L.lib.CeedVectorGetLength(v.h, C.byref(n)))
            self.L.chk(self.L.lib.CeedXCommAllReduce(self.ceed.h, v.h, 0, n.value))
            return
        import torch.distributed as dist
        h = self.dist
        if v.t.device.type == "cuda":
            self.ceed.synchronize()
        if h.stage_host and v.t.device.type == "cuda":
            t = v.t.cpu()
            dist.all_reduce(t, group=h.group)
            v.t.copy_(t)
        else:
            dist.all_reduce(v.t, group=h.group)
        if v.t.device.type == "cuda":
            v.set_device_pointer(v.t.data_ptr())

    def _dist_vector(self, n: int) -> cd.Vector:
        """A Ceed vector over a torch tensor (several ranks: torch.distributed sums it where the library has no communicator)."""
        import torch
        v = self.ceed.vector(n)
        v.t = torch.zeros(max(n, 1), dtype=torch.float64, device=self.dist.device)[:n]
        if v.t.device.type == "cuda":
            v.set_device_pointer(v.t.data_ptr())
        else:
            v.set_array(v.t.numpy(), copy=False)
        return v

    def _first_transfer_distributed(self, A_host, B, free_node, dense_limit):
        """The first transfer of the hierarchy on an ELEMENT-PARTITIONED level (the reference's PCGAMG is parallel, elasticity.c:568-585).
        The level's operator is A = sum_r R_r^T A_r R_r (A_r: rank r's own assembled matrix on its local nodes, `A_host` here); vectors
        are consistent L-vectors.  Per rank, on the host, once per solve:
          * aggregates of the free nodes this rank OWNS (lowest sharing rank), rigid-body tentative prolongation P0 on them, coarse
            dofs numbered globally by rank;
          * P0's rows at the interface nodes the rank holds but does not own are fetched from their owners; T0 = A_r P0 is formed
            locally, its interface rows (and the diagonal's) are summed over the sharing ranks in rank order, identically on each;
          * P = (I - w D^-1 A) P0 on the rank's local nodes (consistent on the interface), lambda_max(D^-1 A) by a distributed power
            iteration;
        on the device: P, P^T, T = A_r P, and the rank's Galerkin CONTRIBUTION P^T T (fixed patterns).  The coarse matrix is the sum of
        the contributions over the ranks on the union of their patterns: ONE all-reduce of its values per Newton step (`_update_level`).
        Everything below is replicated and small.  Per V-cycle: one all-reduce of the restricted residual (`restrict`); the
        prolongation is local.  Per-rank memory of the level: the rank's share of A, P, T -- proportional to 1 / ranks."""
        import scipy.sparse as sp
        import torch.distributed as dist
        from .mesh import key_bytes
        c, h, asm = self.ceed, self.dist, self.asm
        dm = asm.p.levels[asm.level].dofmap
        n = A_host.shape[0]
        nn = n // 3
        kb = key_bytes(dm.node_keys)
        own_node = np.asarray(h.owner_weight).reshape(nn, 3)[:, 0] > 0
        shared = np.zeros(nn, dtype=bool)
        for nb in h.neigh:
            shared[nb.dof_idx.cpu().numpy()[::3] // 3] = True
        dof_ptr = 3 * np.arange(nn + 1, dtype=np.int64)

        def gather(obj):
            out = [None] * h.world
            dist.all_gather_object(out, obj, group=h.group)
            return out
        # --- aggregates and tentative prolongation on the owned free nodes -----------------------------------------------------
        node_of = np.arange(n) // 3
        a_row = np.repeat(np.arange(n, dtype=np.int64), np.diff(A_host.indptr))
        G = sp.csr_matrix((np.ones(a_row.size), (node_of[a_row], node_of[A_host.indices])), shape=(nn, nn)).tocsr()
        fn = np.nonzero(free_node & own_node)[0]
        t_ag = time.perf_counter()
        if fn.size:
            Gf = G[fn][:, fn].tocsr()
            agg_f, na = aggregate_nodes(Gf.indptr, Gf.indices)
        else:
            agg_f, na = np.zeros(0, dtype=np.int64), 0
        t_ag = time.perf_counter() - t_ag
        agg = np.full(nn, na, dtype=np.int64)
        agg[fn] = agg_f
        if na:
            P0, Bn, col_ptr = tentative_prolongation(agg, na, dof_ptr, B)
        else:
            P0, Bn, col_ptr = sp.csr_matrix((n, 0)), np.zeros((0, B.shape[1])), np.zeros(1, dtype=np.int64)
        ncr = P0.shape[1]
        info = gather({"nc": ncr, "B": Bn, "col_ptr": col_ptr})
        off = np.concatenate([[0], np.cumsum([g["nc"] for g in info])]).astype(np.int64)
        nc = int(off[-1])
        if nc == 0:
            raise ValueError("the aggregation left no coarse degree of freedom on any rank")
        Bnext = np.concatenate([g["B"] for g in info], axis=0)
        col_ptr_g = np.concatenate([[0]] + [g["col_ptr"][1:] + off[r] for r, g in enumerate(info)]).astype(np.int64)
        P0 = sp.csr_matrix((P0.data, P0.indices + off[h.rank], P0.indptr), shape=(n, nc))          # global coarse columns

        def rows_of(M, nodes):
            """(key bytes, CSR rows of the 3 dofs of every node) of a sparse matrix, for publishing."""
            d = (nodes[:, None] * 3 + np.arange(3)[None, :]).ravel()
            S = M[d].tocsr()
            return {"keys": kb[nodes].tobytes(), "n": int(nodes.size), "indptr": S.indptr, "indices": S.indices, "data": S.data}

        def place(pub, ncols):
            """The published rows that belong to nodes THIS rank holds, as an n x ncols matrix on its local dofs."""
            if pub["n"] == 0:
                return sp.csr_matrix((n, ncols))
            keys = np.frombuffer(pub["keys"], dtype=kb.dtype)
            order = np.argsort(kb, kind="stable")
            pos = np.searchsorted(kb[order], keys)
            pos = np.minimum(pos, nn - 1)
            hit = kb[order][pos] == keys
            loc = order[pos]                                     # local node of every published node (where hit)
            S = sp.csr_matrix((pub["data"], pub["indices"], pub["indptr"]), shape=(3 * pub["n"], ncols)).tocoo()
            pn, comp = S.row // 3, S.row % 3
            keep = hit[pn]
            return sp.csr_matrix((S.data[keep], (loc[pn[keep]] * 3 + comp[keep], S.col[keep])), shape=(n, ncols))
        # --- P0 on every local node: the rows of interface nodes owned elsewhere come from their owners ------------------------------
        pubs = gather(rows_of(P0, np.nonzero(shared & own_node)[0]))
        P0 = P0.tolil() if False else P0
        for r, pub in enumerate(pubs):
            if r != h.rank:
                M = place(pub, nc)
                # (only nodes this rank does NOT own can match a row published by another owner)
                P0 = P0 + M
        P0 = P0.tocsr()
        # --- T0 = A P0 and the diagonal: local products, interface rows summed over the sharing ranks in rank order ----------------
        T0 = (A_host @ P0).tocsr()
        diag = A_host.diagonal()
        sh_nodes = np.nonzero(shared)[0]
        sh_dofs = (sh_nodes[:, None] * 3 + np.arange(3)[None, :]).ravel()
        Dm = sp.csr_matrix((diag[sh_dofs], (sh_dofs, np.zeros(sh_dofs.size, dtype=np.int64))), shape=(n, 1))
        pubs = gather({"T": rows_of(T0, sh_nodes), "D": rows_of(Dm, sh_nodes)})
        keep_rows = np.ones(n); keep_rows[sh_dofs] = 0.0
        Tsum = sp.diags(keep_rows) @ T0                          # interior rows as they are; interface rows rebuilt below
        Dsum = sp.csr_matrix((n, 1))
        mine_T, mine_D = T0[sh_dofs], diag[sh_dofs]
        for r, pub in enumerate(pubs):                           # rank order: the same sums, in the same order, on every sharing rank
            if r == h.rank:
                Tsum = Tsum + sp.csr_matrix((mine_T.tocoo().data, (sh_dofs[mine_T.tocoo().row], mine_T.tocoo().col)), shape=(n, nc))
                Dsum = Dsum + sp.csr_matrix((mine_D, (sh_dofs, np.zeros(sh_dofs.size, dtype=np.int64))), shape=(n, 1))
            else:
                Tsum = Tsum + place(pub["T"], nc)
                Dsum = Dsum + place(pub["D"], 1)
        diag_g = diag.copy()
        diag_g[sh_dofs] = np.asarray(Dsum.todense()).ravel()[sh_dofs]
        dinv = np.where(diag_g != 0.0, 1.0 / np.where(diag_g != 0.0, diag_g, 1.0), 0.0)
        # --- lambda_max(D^-1 A): power iteration, the products halo-summed, norms over the owned dofs ----------------------------
        import torch
        w_own = np.asarray(h.owner_weight, dtype=np.float64)

        def halo_sum(y):
            t = torch.from_numpy(np.ascontiguousarray(y)).to(h.device)
            h.add(t)
            return t.cpu().numpy()
        xk = np.sin(dm.node_coords @ np.array([[12.9898, 78.233, 37.719], [93.989, 67.345, 24.113], [45.164, 11.135, 83.951]]).T * 437.5453).reshape(-1)
        xk = xk * (np.abs(B).sum(axis=1) > 0.0)                  # (a function of the coordinates: consistent on shared nodes)
        lam = 1.0
        for _ in range(30):
            y = dinv * halo_sum(A_host @ xk)
            ny, nx = self._allreduce_np(np.array([np.sum(w_own * y * y), np.sum(w_own * xk * xk)]))
            lam = float(np.sqrt(ny / nx))
            xk = y / np.sqrt(ny)
        lam *= 1.05
        P = (P0 - (self.damping * 4.0 / 3.0 / lam) * (sp.diags(dinv) @ Tsum)).tocsr()
        P.sort_indices()
        Pt = P.T.tocsr()
        Pt.sort_indices()
        # --- device objects: the rank's share ---------------------------------------------------------------------------------------
        lv = _Level()
        lv.A, lv.n, lv.nc = asm.csr, n, nc
        lv.P = cd.Csr.rect(c, n, nc, P.indptr, P.indices, P.data)
        lv.Pt = cd.Csr.rect(c, nc, n, Pt.indptr, Pt.indices, Pt.data)
        lv.dense_next = nc <= dense_limit
        lv.T = cd.Csr.product(asm.csr, lv.P, variable=0)
        local = cd.Csr.product(lv.Pt, lv.T, variable=1, dense=lv.dense_next)
        # --- the coarse matrix: the contributions summed on the union of their patterns (replicated) -------------------------------
        nr, ncol, nz, rp, cl = local.pattern()
        if lv.dense_next:
            rp_u = np.arange(nc + 1, dtype=np.int64) * nc
            cl_u = np.tile(np.arange(nc, dtype=np.int64), nc)
            slot = np.arange(nc * nc, dtype=np.int64)
        else:
            pats = gather({"rp": rp, "cl": cl})
            U = None
            for g in pats:
                M = sp.csr_matrix((np.ones(g["cl"].size), g["cl"], g["rp"]), shape=(nc, nc))
                U = M if U is None else U + M
            U = U.tocsr(); U.sort_indices()
            rp_u, cl_u = U.indptr.astype(np.int64), U.indices.astype(np.int64)
            key_u = np.repeat(np.arange(nc, dtype=np.int64), np.diff(rp_u)) * nc + cl_u
            key_l = np.repeat(np.arange(nc, dtype=np.int64), np.diff(rp)) * nc + cl
            slot = np.searchsorted(key_u, key_l)
            assert np.array_equal(key_u[slot], key_l)
        nnz_u = int(rp_u[-1])
        lv.Anext = cd.Csr(c, rp_u, cl_u, np.arange(nnz_u, dtype=np.int64), ())
        wv = self._dist_vector(n)
        wv.t.copy_(torch.from_numpy(w_own * (np.asarray(asm.mask) == 0)).to(wv.t.device))
        if wv.t.device.type == "cuda":
            wv.set_device_pointer(wv.t.data_ptr())
        lv.dd = {"local": local, "slot_t": torch.from_numpy(np.ascontiguousarray(slot, dtype=np.int64)).to(h.device), "nnz_union": nnz_u,
                 "coo": self._dist_vector(max(nnz_u, 1)), "vals": self._dist_vector(max(local.nnz, 1)), "w": wv, "rw": c.vector(n).set_value(0.0)}
        lv.info = dict(rows=int(n), aggregates=int(na), coarse_dofs=int(nc), coarse_dofs_of_this_rank=int(ncr), distributed_over=int(h.world),
                       nodes_per_aggregate=float(fn.size) / max(na, 1), prolongation_entries_per_row=float(P.nnz) / max(n, 1),
                       galerkin_entries=int(lv.T.nnz), lambda_max=lam, next_is_dense=bool(lv.dense_next), seconds=dict(aggregate=t_ag),
                       device_bytes=int(12 * (P.nnz + Pt.nnz) + 12 * lv.T.nnz + 12 * local.nnz + 12 * nnz_u),
                       # the rank's share of the level (P, P^T, T = A_r P: proportional to 1 / ranks) and what is replicated under it
                       distributed_bytes=int(12 * (P.nnz + Pt.nnz) + 12 * lv.T.nnz), replicated_bytes=int(12 * local.nnz + 12 * nnz_u))
        return lv, Bnext, col_ptr_g

    def build(self):
        import scipy.sparse as sp
        t0 = time.perf_counter()
        asm, c = self.asm, self.ceed
        n = asm.nrows
        constrained = np.asarray(asm.mask) != 0          # (of the matrix's rows: the global ones when the level is replicated)
        A_host = sp.csr_matrix((asm.csr.values(c), asm.cols, asm.rowptr), shape=(n, n))
        nn = n // 3
        dof_ptr = 3 * np.arange(nn + 1, dtype=np.int64)
        B = rigid_body_modes(np.asarray(asm.node_coords), constrained)
        free_node = ~constrained.reshape(nn, 3).all(axis=1)       # fully constrained nodes have identity rows
        A_csr = asm.csr
        self.levels = []
        while True:
            last_allowed = len(self.levels) + 2 >= self.max_levels
            if self.dist is not None and not self.levels:
                lv, B, dof_ptr = self._first_transfer_distributed(A_host, B, free_node, dense_limit=self.max_coarse_dofs)
            else:
                lv, B, dof_ptr = self._transfer(A_csr, A_host, dof_ptr, B, free_node,
                                                dense_limit=self.max_coarse_dofs if not last_allowed else 2 ** 30)
            self.levels.append(lv)
            if lv.dense_next:
                break
            if lv.nc > 0.7 * lv.n:
                raise ValueError(f"aggregation stalls ({lv.n} -> {lv.nc} rows): no hierarchy for this matrix")
            # the next level's matrix on the host (for its aggregates and its prolongator smoothing), from the device product
            self._update_level(lv)
            nr, ncol, nz, rp, cl = lv.Anext.pattern()
            A_host = sp.csr_matrix((lv.Anext.values(c), cl, rp), shape=(nr, ncol))
            A_csr = lv.Anext
            free_node = np.ones(dof_ptr.size - 1, dtype=bool)
            # work vectors and the start vector of the eigenvalue estimate of the new level
            for k in ("x", "b", "r", "d", "t", "z", "dinv"):
                lv.v[k] = self._dist_vector(lv.nc) if (self.dist is not None and k == "b" and len(self.levels) == 1) else c.vector(lv.nc).set_value(0.0)
            x0 = np.random.default_rng(4321 + len(self.levels)).uniform(-1.0, 1.0, lv.nc)
            lv.x0 = c.vector(lv.nc).set_array(x0 / np.sqrt(np.square(x0).sum()))
        last = self.levels[-1]
        self.rc, self.xc = c.vector(last.nc).set_value(0.0), c.vector(last.nc).set_value(0.0)
        if self.dist is not None and len(self.levels) == 1:      # the restricted residual is summed over the ranks: behind a tensor
            self.rc = self._dist_vector(last.nc)
        self.nc = self.levels[0].nc
        self.info = dict(levels=len(self.levels) + 1, rows=[self.levels[0].n] + [l.nc for l in self.levels],
                         aggregates=self.levels[0].info["aggregates"], coarse_dofs=self.levels[0].nc,
                         nodes_per_aggregate=self.levels[0].info["nodes_per_aggregate"],
                         per_level=[l.info for l in self.levels], build_seconds=time.perf_counter() - t0)
        if self.verbose:
            print("AggregationAMG:", self.info, flush=True)

    # ---- every Newton step: the Galerkin matrices of the current Jacobian, smoother bounds, the inverse -------------
    def setup(self):
        if not self.levels:
            self.build()
        for i, lv in enumerate(self.levels):
            self._update_level(lv)
            if lv.dense_next:
                lv.Anext.invert_dense_spd()
            else:
                lv.Anext.diagonal(lv.v["dinv"])
                lv.v["dinv"].reciprocal()
                lv.emax = self._estimate_emax(lv)

    def _estimate_emax(self, lv, steps: int = 10) -> float:
        """Largest eigenvalue of D^-1 A_next: the Lanczos tridiagonal of `steps` Jacobi-PCG steps, scalars on the device
        (the same recurrence as solver.NewtonPMG._lanczos_device)."""
        lib, chk, v = self.L.lib, self.L.chk, lv.v
        r, z, pv, Ap = v["r"], v["z"], v["d"], v["t"]
        if self._scal is None:
            self._scal = self.ceed.vector(8 + 2 * 16)
        sc = self._scal
        sc.set_value(0.0)
        one, neg = C.c_double(1.0), C.c_double(-1.0)
        chk(lib.CeedXVectorAXPBY(r.h, one, lv.x0.h, C.c_double(0.0)))
        chk(lib.CeedXVectorPointwiseMult(z.h, r.h, v["dinv"].h))
        chk(lib.CeedXVectorAXPBY(pv.h, one, z.h, C.c_double(0.0)))
        chk(lib.CeedXVectorDotTo(r.h, z.h, None, sc.h, 0))
        for j in range(steps):
            rz, rz_new, ja, jb = (0, 3, 8 + 2 * j, 9 + 2 * j) if j % 2 == 0 else (3, 0, 8 + 2 * j, 9 + 2 * j)
            lv.Anext.apply(pv, Ap)
            chk(lib.CeedXVectorDotTo(pv.h, Ap.h, None, sc.h, 1))
            chk(lib.CeedXScalarDivide(sc.h, ja, rz, 1, one))
            chk(lib.CeedXVectorAXPBYScalars(r.h, sc.h, ja, neg, Ap.h, -1, one))
            chk(lib.CeedXVectorPointwiseMult(z.h, r.h, v["dinv"].h))
            chk(lib.CeedXVectorDotTo(r.h, z.h, None, sc.h, rz_new))
            chk(lib.CeedXScalarDivide(sc.h, jb, rz_new, rz, one))
            chk(lib.CeedXVectorAXPBYScalars(pv.h, sc.h, -1, one, z.h, jb, one))
        s = sc.to_numpy()
        alphas, betas = [], []
        for j in range(steps):
            if not (s[8 + 2 * j] > 0.0) or not np.isfinite(s[9 + 2 * j]):
                break
            alphas.append(float(s[8 + 2 * j])); betas.append(float(s[9 + 2 * j]))
        k = len(alphas)
        if not k:
            return 1.0
        T = np.zeros((k, k))
        for j in range(k):
            T[j, j] = 1.0 / alphas[j] + (betas[j - 1] / alphas[j - 1] if j else 0.0)
            if j + 1 < k:
                T[j, j + 1] = T[j + 1, j] = np.sqrt(max(betas[j], 0.0)) / alphas[j]
        return float(np.linalg.eigvalsh(T).max())

    # ---- the cycle ------------------------------------------------------------------------------------------------------
    def _chebyshev(self, lv, b, x, zero_guess):
        """Chebyshev-Jacobi sweep on lv.Anext (the matrix of the level BELOW transfer lv), bounds [emax / ratio, 1.1 emax]."""
        lib, chk, v = self.L.lib, self.L.chk, lv.v
        lmin, lmax = lv.emax / self.smooth_ratio, 1.1 * lv.emax
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta / delta
        rho = 1.0 / sigma
        r, d, t = v["r"], v["d"], v["t"]
        if zero_guess:
            chk(lib.CeedXVectorChebyshevStart(x.h, d.h, r.h, b.h, None, v["dinv"].h, C.c_double(1.0 / theta), 1))
        else:
            lv.Anext.apply(x, t)
            chk(lib.CeedXVectorChebyshevStart(x.h, d.h, r.h, b.h, t.h, v["dinv"].h, C.c_double(1.0 / theta), 0))
        for _ in range(1, self.smooth_its):
            lv.Anext.apply(d, t)
            rho_new = 1.0 / (2.0 * sigma - rho)
            chk(lib.CeedXVectorChebyshevUpdate(x.h, d.h, r.h, t.h, v["dinv"].h, C.c_double(2.0 * rho_new / delta), C.c_double(rho_new * rho), 0))
            rho = rho_new

    def _cycle(self, i):
        """Solve approximately A_{i} x = b on the level below transfer i - 1 (its vectors live in levels[i - 1].v)."""
        up = self.levels[i - 1]             # the transfer that produced this level: its matrix is up.Anext
        b, x = up.v["b"], up.v["x"]
        lv = self.levels[i]                 # the transfer from this level to the next
        lib, chk = self.L.lib, self.L.chk
        for rep in range(self.coarse_cycles):      # coarse_cycles = 2: two cycles of this level per visit (a W-cycle of the hierarchy)
            self._chebyshev(up, b, x, rep == 0)
            up.Anext.apply(x, up.v["t"])
            chk(lib.CeedXVectorWAXPBY(up.v["z"].h, C.c_double(1.0), b.h, C.c_double(-1.0), up.v["t"].h))
            if lv.dense_next:
                lv.Pt.apply(up.v["z"], self.rc)
                lv.Anext.apply(self.rc, self.xc)
                lv.P.apply(self.xc, up.v["z"])
            else:
                lv.Pt.apply(up.v["z"], lv.v["b"])
                self._cycle(i + 1)
                lv.P.apply(lv.v["x"], up.v["z"])
            chk(lib.CeedXVectorAXPBY(x.h, C.c_double(1.0), up.v["z"].h, C.c_double(1.0)))
            self._chebyshev(up, b, x, False)

    # ---- the coarse correction of level 0 (called by the solver between its own smoothing sweeps): x += P (...) P^T r
    def restrict(self, r: cd.Vector):
        l0 = self.levels[0]
        dst = self.rc if l0.dense_next else l0.v["b"]
        if self.dist is None:
            l0.Pt.apply(r, dst)
            return
        # several ranks: r is a consistent L-vector -- every dof counts ONCE (owner weights), each rank restricts its share onto the
        # (global) coarse dofs and the shares are summed over the ranks: the coarse level below is replicated
        d = l0.dd
        self.L.chk(self.L.lib.CeedXVectorPointwiseMult(d["rw"].h, r.h, d["w"].h))
        l0.Pt.apply(d["rw"], dst)
        self._allreduce(dst)

    def solve_coarsest(self):
        l0 = self.levels[0]
        if l0.dense_next:
            l0.Anext.apply(self.rc, self.xc)
        else:
            self._cycle(1)

    def prolong(self, z: cd.Vector):
        l0 = self.levels[0]
        l0.P.apply(self.xc if l0.dense_next else l0.v["x"], z)

    def destroy(self):
        for lv in reversed(self.levels):
            dd = getattr(lv, "dd", None) or {}
            for o in [lv.Anext, dd.get("local"), lv.T, lv.Pt, lv.P, lv.x0, dd.get("coo"), dd.get("vals"), dd.get("w"), dd.get("rw")] + list(lv.v.values()):
                if o is not None:
                    o.destroy()
        for o in (self.rc, self.xc, self._scal):
            if o is not None:
                o.destroy()
        self.levels, self.rc, self.xc, self._scal = [], None, None, None
