"""Assembled coarse-level Jacobian.

The reference assembles the p=1 operator of the multigrid hierarchy by finite-difference colouring
of the coarse residual (`SNESComputeJacobianDefaultColor`, misc.c:151-183, elasticity.c:457-483) --
at least 81 coarse residual evaluations per Newton step -- and hands the AIJ matrix to GAMG.  Here
the matrix is assembled EXACTLY from element matrices: the level's own Jacobian operator graph
(setuplibceed.c:817-839: same basis, QFunction, q-point data and stored state) is instantiated once
more on an element-DISCONTINUOUS restriction (every element owns private copies of its nodes), and
applied to the 3*P^3 unit vectors "dof (n', c') of every element = 1".  Output j holds column j of
every element matrix, so the 3*P^3 applies (24 at p=1, each the ordinary fused kernel on the coarse
mesh) yield all element matrices in COO form; `CeedXCsrAssemble` sums them into CSR in a fixed order.
Dirichlet rows and columns are dropped and replaced by a unit diagonal, which is what the masked
matrix-free operator plus the solver's identity on constrained rows amounts to.

Everything goes through the C ABI (`include/ceed.h`), so the same code runs on the CPU oracle, where
the tests compare A x against the matrix-free J x.

Several ranks (round 3): ``replicate=<the level's HaloExchange>`` makes the matrix the GLOBAL one on every rank -- the ranks
all-gather their element matrices (24 x 24 doubles per element at p = 1) every Newton step, each assembles the same global CSR
in a global numbering made of the partition-independent node keys, and the coarse solve under it (the aggregation hierarchy of
amg.py, PCGAMG's role) runs REPLICATED on every GPU; `globalise` / `localise` move a coarse L-vector between the two numberings
(one all-gather of the owned entries per V-cycle).  The p = 1 level is 1.7 % of the fine level's dofs: replicating it costs little
and gives every rank the single-rank hierarchy, instead of aggregates that stop at partition boundaries.
"""
from __future__ import annotations

import numpy as np

from . import ceed as cd
from .solid import SolidProblem


class AssembledLevel:
    def __init__(self, prob: SolidProblem, level: int = 0, replicate=None):
        self.p, self.level = prob, level
        c = self.ceed = prob.ceed
        lv = prob.levels[level]
        self.rep = replicate if (replicate is not None and replicate.world > 1) else None
        P = lv.degree + 1
        ne, P3 = prob.mesh.nelem, P ** 3
        self.ne, self.nd = ne, 3 * P3                              # element matrices are nd x nd
        nloc = ne * self.nd
        # --- the level's Jacobian on an element-discontinuous restriction ---------------------------
        eoff = (np.arange(ne * P3, dtype=np.int64) * 3).astype(np.int32)
        self.rstr = c.elem_restriction(ne, P3, 3, 1, nloc, eoff)
        name = prob.info["jacob"]
        self.qf = c.qfunction(name, source=f"qfunctions/{prob.info['src']}:{name}")
        self.qf.add_input("deltadu", 9, cd.EVAL_GRAD).add_input("qdata", 10, cd.EVAL_NONE)
        if prob.info["state"]:
            self.qf.add_input("gradu", 9, cd.EVAL_NONE)
        self.qf.add_output("deltadv", 9, cd.EVAL_GRAD)
        self.qf.set_context(prob.phys, reported_size=8)
        self.op = c.operator(self.qf)
        self.op.set_field("deltadu", self.rstr, lv.basisu, "active")
        self.op.set_field("qdata", prob.Erestrictqdi, None, prob.qdata)
        self.op.set_field("deltadv", self.rstr, lv.basisu, "active")
        if prob.info["state"]:
            self.op.set_field("gradu", prob.ErestrictGradui, None, prob.gradu)
        # --- unit vectors and the COO value buffer: entry [j][e][n][c] = K_e[(n,c), j] -----------------
        self.units = []
        for j in range(self.nd):
            u = np.zeros((ne, self.nd))
            u[:, j] = 1.0
            self.units.append(c.vector(nloc).set_array(u.reshape(-1)))
        self.on_device = c.preferred_memtype == cd.MEM_DEVICE
        self._coo_host = None if self.on_device else np.zeros(self.nd * nloc)
        self.coo = c.vector(self.nd * nloc)
        self._coo_local_t = None
        if self.on_device and self.rep is not None:              # gathered across ranks every Newton step: behind a torch tensor
            import torch
            self._coo_local_t = torch.zeros(self.nd * nloc, dtype=torch.float64, device=self.rep.device)
            self.coo.set_device_pointer(self._coo_local_t.data_ptr())
        elif self.on_device:
            self.coo.set_value(0.0)
        else:
            self.coo.set_array(self._coo_host, copy=False)       # the oracle works on host memory only
        self._cols_out = None
        # --- sparsity pattern and COO -> CSR map (host, once) ----------------------------------------
        off = np.asarray(lv.dofmap.offsets(), dtype=np.int64).reshape(ne, P3)
        dof = (off[:, :, None] + np.arange(3)[None, None, :]).reshape(ne, self.nd)       # [e][(n,c)]
        n = lv.dofmap.lsize
        constrained = lv.mask != 0
        self.mask, self.node_coords = lv.mask, lv.dofmap.node_coords                      # of the matrix's rows (amg.py)
        self.coo_local = self.coo
        if self.rep is not None:          # the GLOBAL matrix on every rank: global numbering, all ranks' elements
            dof, n, constrained = self._replicate_setup(lv, dof, constrained)
            ne = dof.shape[0]
        rows = np.broadcast_to(dof[None, :, :], (self.nd, ne, self.nd)).reshape(-1)        # [j][e][(n,c)]
        cols = np.broadcast_to(dof.T[:, :, None], (self.nd, ne, self.nd)).reshape(-1)      # column = dof (e, j)
        keep = ~(constrained[rows] | constrained[cols])
        key = rows.astype(np.int64) * n + cols
        uniq, inv = np.unique(np.concatenate([key[keep], np.nonzero(constrained)[0] * (n + 1)]), return_inverse=True)
        r_u, c_u = uniq // n, uniq % n
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.add.at(rowptr, r_u + 1, 1)
        rowptr = np.cumsum(rowptr)
        slot = np.full(rows.size, -1, dtype=np.int64)
        slot[keep] = inv[:int(keep.sum())]
        self.nrows, self.nnz = n, int(uniq.size)
        self.rowptr, self.cols = rowptr, c_u            # the pattern, for the hierarchy under this level (amg.py)
        self.csr = cd.Csr(c, rowptr, c_u, slot, np.nonzero(constrained)[0])

    # ---- several ranks: the replicated global matrix ----------------------------------------------------------------------
    def _replicate_setup(self, lv, dof_local, constrained_local):
        """Global numbering of this level's nodes from their partition-independent keys; every rank's element -> global dof
        lists and constrained set; the (torch) buffers of the per-Newton-step all-gather of the element matrices."""
        import torch
        import torch.distributed as dist
        from .mesh import key_bytes
        h, dm = self.rep, lv.dofmap
        kb = key_bytes(dm.node_keys)
        gathered = [None] * h.world
        dist.all_gather_object(gathered, {"keys": kb.tobytes(), "ne": int(dof_local.shape[0])}, group=h.group)
        allk = np.unique(np.concatenate([np.frombuffer(g["keys"], dtype=kb.dtype) for g in gathered]))
        gid = np.searchsorted(allk, kb)                                    # global node id of every local node
        self.n_global_nodes = int(allk.size)
        self.gdof_of_local = (3 * gid[:, None] + np.arange(3)[None, :]).reshape(-1)         # local L-vector entry -> global entry
        owned = np.nonzero(h.owner_weight > 0)[0]                          # every global entry is owned by exactly one rank
        info = [None] * h.world
        dist.all_gather_object(info, {"gdof": self.gdof_of_local[dof_local], "owned_g": self.gdof_of_local[owned],
                                      "con_g": self.gdof_of_local[np.nonzero(constrained_local)[0]],
                                      "xyz": (gid, np.asarray(dm.node_coords))}, group=h.group)
        n = 3 * self.n_global_nodes
        constrained = np.zeros(n, dtype=bool)
        coords = np.zeros((self.n_global_nodes, 3))
        for g in info:
            constrained[g["con_g"]] = True
            coords[g["xyz"][0]] = g["xyz"][1]
        self.mask, self.node_coords = constrained.astype(np.uint8), coords
        self.ne_rank = [int(g["gdof"].shape[0]) for g in info]
        self.ne_max = max(self.ne_rank)
        # owned entries: rank r's slice of the gathered (world x max_owned) array -> global positions
        self.owned_local = torch.from_numpy(owned)
        self.n_owned_rank = [int(g["owned_g"].size) for g in info]
        self.n_owned_max = max(self.n_owned_rank)
        pos = np.full(n, -1, dtype=np.int64)
        for r, g in enumerate(info):
            pos[g["owned_g"]] = r * self.n_owned_max + np.arange(g["owned_g"].size)
        assert (pos >= 0).all()
        self.global_from_gather = torch.from_numpy(pos)
        self.local_from_global = torch.from_numpy(self.gdof_of_local)
        # the global COO buffer [j][e over all ranks][(n,c)] behind the CeedVector the CSR sums from
        ne_tot = sum(self.ne_rank)
        dev = h.device
        self.coo_t = torch.zeros(self.nd * ne_tot * self.nd, dtype=torch.float64, device=dev)
        self.coo = self.ceed.vector(self.coo_t.numel())
        if dev.type == "cuda":
            self.coo.set_device_pointer(self.coo_t.data_ptr())
        else:
            self.coo.set_array(self.coo_t.numpy(), copy=False)
        self._pad = torch.zeros(self.nd, self.ne_max, self.nd, dtype=torch.float64, device="cpu" if h.stage_host else dev)
        self._all = [torch.zeros_like(self._pad) for _ in range(h.world)]
        return np.concatenate([g["gdof"] for g in info], axis=0), n, constrained

    def _gather_coo(self):
        """All ranks' element matrices into the global COO buffer (every Newton step)."""
        import torch
        import torch.distributed as dist
        h = self.rep
        if self.on_device:
            self.ceed.synchronize()
            loc = self._coo_local_t
        else:
            loc = torch.from_numpy(self._coo_host)
        self._pad[:, :self.ne, :] = loc.view(self.nd, self.ne, self.nd).to(self._pad.device)
        dist.all_gather(self._all, self._pad, group=h.group)
        g = self.coo_t.view(self.nd, -1, self.nd)
        e0 = 0
        for r, ner in enumerate(self.ne_rank):
            g[:, e0:e0 + ner, :] = self._all[r][:, :ner, :].to(g.device)
            e0 += ner
        if self.on_device:
            self.coo.set_device_pointer(self.coo_t.data_ptr())     # (the tensor was written outside the Ceed)

    def globalise(self, x_local_t, x_global_t):
        """Coarse L-vector (consistent on shared nodes) -> the global vector, on every rank: all-gather of the owned entries."""
        import torch
        import torch.distributed as dist
        h = self.rep
        buf = torch.zeros(self.n_owned_max, dtype=torch.float64, device="cpu" if h.stage_host else x_local_t.device)
        buf[:self.owned_local.numel()] = x_local_t[self.owned_local.to(x_local_t.device)].to(buf.device)
        out = [torch.zeros_like(buf) for _ in range(h.world)]
        dist.all_gather(out, buf, group=h.group)
        x_global_t.copy_(torch.cat(out)[self.global_from_gather.to(buf.device)].to(x_global_t.device))

    def localise(self, x_global_t, x_local_t):
        x_local_t.copy_(x_global_t[self.local_from_global.to(x_global_t.device)])

    def assemble(self):
        """Refresh the matrix for the current stored state: 3*P^3 operator applies + one summation."""
        nloc = self.ne * self.nd
        if self._cols_out is None:          # output j aliases slice j of the COO buffer (CEED_USE_POINTER borrow)
            if self.on_device:
                base = self.coo_local.device_pointer()
                self._cols_out = [self.ceed.vector(nloc).set_device_pointer(base + 8 * j * nloc) for j in range(self.nd)]
            else:
                self._cols_out = [self.ceed.vector(nloc).set_array(self._coo_host[j * nloc:(j + 1) * nloc], copy=False)
                                  for j in range(self.nd)]
        for j in range(self.nd):
            self.op.apply(self.units[j], self._cols_out[j])
        if self.rep is not None:
            self._gather_coo()
        self.csr.assemble(self.coo)

    def apply(self, x: cd.Vector, y: cd.Vector):
        self.csr.apply(x, y)

    def diagonal(self, d: cd.Vector):
        self.csr.diagonal(d)

    def destroy(self):
        for o in [self.csr, self.op, self.qf, self.rstr, self.coo] + ([self.coo_local] if self.coo_local is not self.coo else []) + self.units + (self._cols_out or []):
            o.destroy()
