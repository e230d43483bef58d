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

Several ranks: every rank assembles the matrix of ITS elements (interface rows hold the rank's partial sums); the level's operator is
"local product, then the interface sum", like the matrix-free levels.  The aggregation hierarchy under it is distributed in its first
transfer (amg.py, round 5); rounds 3-4 all-gathered every element matrix and replicated the whole level.
"""
from __future__ import annotations

import numpy as np

from . import ceed as cd
from .solid import SolidProblem


class AssembledLevel:
    def __init__(self, prob: SolidProblem, level: int = 0):
        self.p, self.level = prob, level
        c = self.ceed = prob.ceed
        lv = prob.levels[level]
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
        if self.on_device:
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
        self.csr.assemble(self.coo)

    def apply(self, x: cd.Vector, y: cd.Vector):
        self.csr.apply(x, y)

    def diagonal(self, d: cd.Vector):
        self.csr.diagonal(d)

    def destroy(self):
        for o in [self.csr, self.op, self.qf, self.rstr, self.coo] + ([self.coo_local] if self.coo_local is not self.coo else []) + self.units + (self._cols_out or []):
            o.destroy()
