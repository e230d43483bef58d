"""Post-processing operators on top of the path.

`StrainEnergy` mirrors `ComputeStrainEnergy` (matops.c:247-296) with the operator graph of
setuplibceed.c:651-670: du --GRAD(basisu)--> {LinElas,HyperSS,HyperFS}Energy with qdata
--INTERP^T(basisEnergy, 1 component)--> energy L-vector, whose sum is the strain energy.
"""
from __future__ import annotations

import numpy as np

from . import ceed as cd
from .solid import SolidProblem

ENERGY_QF = {"linElas": ("linElas.h", "LinElasEnergy"), "hyperSS": ("hyperSS.h", "HyperSSEnergy"),
             "hyperFS": ("hyperFS.h", "HyperFSEnergy")}


class StrainEnergy:
    def __init__(self, prob: SolidProblem, problem: str):
        self.p = prob
        c = prob.ceed
        lv = prob.levels[prob.fine]
        P, Q = lv.degree + 1, prob.Q
        ne = prob.mesh.nelem
        src, name = ENERGY_QF[problem]
        self.nnodes = lv.dofmap.nnodes
        off_e = (np.asarray(lv.dofmap.offsets(), dtype=np.int64) // 3).astype(np.int32)       # one dof per node
        self.rstr = c.elem_restriction(ne, P ** 3, 1, 1, self.nnodes, off_e)                   # ErestrictEnergy
        self.basis = c.basis_lagrange(3, 1, P, Q, cd.GAUSS)                                   # basisEnergy, :343-345
        self.qf = c.qfunction(name, source=f"qfunctions/{src}:{name}")
        self.qf.add_input("du", 9, cd.EVAL_GRAD).add_input("qdata", 10, cd.EVAL_NONE).add_output("energy", 1, cd.EVAL_INTERP)
        self.qf.set_context(prob.phys)
        self.op = c.operator(self.qf)
        self.op.set_field("du", lv.Erestrictu, lv.basisu, "active")
        self.op.set_field("qdata", prob.Erestrictqdi, None, prob.qdata)
        self.op.set_field("energy", self.rstr, self.basis, "active")
        self.eloc = c.vector(self.nnodes)

    def compute(self, xloc: cd.Vector) -> float:
        """xloc: the local displacement vector with the boundary values inserted (matops.c:256-261)."""
        self.op.apply(xloc, self.eloc)
        return float(self.eloc.to_numpy().sum())

    def destroy(self):
        for o in (self.op, self.qf, self.basis, self.rstr, self.eloc):
            o.destroy()
