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


DIAG_QF = {"linElas": ("linElas.h", "LinElasDiagnostic"), "hyperSS": ("hyperSS.h", "HyperSSDiagnostic"),
           "hyperFS": ("hyperFS.h", "HyperFSDiagnostic")}
DIAG_FIELDS = ("displacement_x", "displacement_y", "displacement_z", "pressure", "trace_E", "trace_E2", "det_J",
               "strain_energy_density")          # elasticity.c:181-188


class Diagnostics:
    """opDiagnostic (setuplibceed.c:679-737) + the multiplicity division of ViewDiagnosticQuantities
    (misc.c:217-311): the 8 nodal diagnostic fields of the fine level, evaluated on the GLL points
    (= the nodes) with a second SetupGeo on those points."""

    def __init__(self, prob: SolidProblem, problem: str):
        self.p = prob
        c = prob.ceed
        lv = prob.levels[prob.fine]
        P = lv.degree + 1
        ne, P3 = prob.mesh.nelem, P ** 3
        self.nnodes = lv.dofmap.nnodes
        # geometry on the GLL points (:679-708)
        self.basisx = c.basis_lagrange(3, 3, 2, P, cd.GAUSS_LOBATTO)
        self.rstr_qd = c.strided_restriction(ne, P3, 10, 10 * ne * P3)
        self.qdata = c.vector(10 * ne * P3)
        qfg = c.qfunction("SetupGeo", source="qfunctions/common.h:SetupGeo")
        qfg.add_input("dx", 9, cd.EVAL_GRAD).add_input("weight", 1, cd.EVAL_WEIGHT).add_output("qdata", 10, cd.EVAL_NONE)
        opg = c.operator(qfg)
        opg.set_field("dx", prob.Erestrictx, self.basisx, "active")
        opg.set_field("weight", None, self.basisx, None)
        opg.set_field("qdata", self.rstr_qd, None, "active")
        opg.apply(prob.xcoord, self.qdata)
        opg.destroy(); qfg.destroy()
        # the diagnostic operator (:712-737)
        self.basis = c.basis_lagrange(3, 3, P, P, cd.GAUSS_LOBATTO)                           # basisDiagnostic, :347-348
        off_d = (np.asarray(lv.dofmap.offsets(), dtype=np.int64) // 3 * 8).astype(np.int32)
        self.rstr = c.elem_restriction(ne, P3, 8, 1, 8 * self.nnodes, off_d)                  # ErestrictDiagnostic
        src, name = DIAG_QF[problem]
        self.qf = c.qfunction(name, source=f"qfunctions/{src}:{name}")
        self.qf.add_input("u", 3, cd.EVAL_INTERP).add_input("du", 9, cd.EVAL_GRAD).add_input("qdata", 10, cd.EVAL_NONE)
        self.qf.add_output("diagnostic", 8, cd.EVAL_NONE)
        self.qf.set_context(prob.phys)
        self.op = c.operator(self.qf)
        self.op.set_field("u", lv.Erestrictu, self.basis, "active")
        self.op.set_field("du", lv.Erestrictu, self.basis, "active")
        self.op.set_field("qdata", self.rstr_qd, None, self.qdata)
        self.op.set_field("diagnostic", self.rstr, None, "active")
        self.dloc = c.vector(8 * self.nnodes)
        m = c.vector(prob.lsize())
        lv.Erestrictu.multiplicity(m)
        self.mult = m.to_numpy().reshape(-1, 3)[:, 0].copy()

    def compute(self, xloc: cd.Vector) -> np.ndarray:
        """[nnodes][8] nodal fields (DIAG_FIELDS), averaged over the elements sharing each node."""
        self.op.apply(xloc, self.dloc)
        return self.dloc.to_numpy().reshape(self.nnodes, 8) / self.mult[:, None]
