"""Operator graphs and MatShell-style wrappers of the elasticity mini-app, over the
C-ABI boundary (``include/ceed.h``).

Mirrors, call for call, what the reference does above libCEED:

* ``SolidProblem._setup_fine_level`` ~ ``SetupLibceedFineLevel`` (src/setuplibceed.c:243-745):
  restrictions, bases, qdata via opSetupGeo, residual operator ``opApply``.
* ``SolidProblem._setup_level``      ~ ``SetupLibceedLevel`` (:748-939): per-level Jacobian
  operator (coarse P, FINE quadrature and q-data), prolong / restrict operators.
* ``apply_jacobian`` / ``form_residual`` / ``prolong`` / ``restrict`` / ``get_diag``
  ~ ``ApplyJacobian_Ceed`` / ``FormResidual_Ceed`` / ``Prolong_Ceed`` / ``Restrict_Ceed`` /
  ``GetDiag_Ceed`` (src/matops.c:98,63,115,160,206) incl. ``ApplyLocalCeedOp`` (:26-60).

PETSc's DM is replaced by the L-vector convention of this build: a
"global" vector is an L-vector whose constrained (Dirichlet) entries are zero and, on
several GPUs, whose interface entries are replicated and consistent.  G->L with zeroed
``Xloc`` (matops.c:33,106) and L->G dropping constrained rows (matops.c:57) therefore
reduce to the Dirichlet flags folded into the operator's offsets
(``CeedXOperatorSetDirichletMaskMode``), plus the neighbour halo sum in
``ceedpetscsolid_amd.halo`` when there is more than one rank.

The same code drives the product library and (in tests only) the CPU oracle.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import ctypes as C
import numpy as np

from . import ceed as cd
from .mesh import DofMap, HexMesh, build_dofmap, dirichlet_mask, side_set_nodes, boundary_nodes

# problemOptions[] of setuplibceed.c:41-107 (hyperFSIncomp is dead code upstream)
PROBLEMS = {
    "linElas": dict(apply="LinElasF", jacob="LinElasdF", state=False, src="linElas.h"),
    "hyperSS": dict(apply="HyperSSF", jacob="HyperSSdF", state=True, src="hyperSS.h"),
    "hyperFS": dict(apply="HyperFSF", jacob="HyperFSdF", state=True, src="hyperFS.h"),
}


def level_degrees(degree: int, multigrid: str = "logarithmic") -> List[int]:
    """cloptions.c:195-225."""
    if multigrid == "none" or degree == 1:
        return [degree]
    if multigrid == "uniform":
        return list(range(1, degree + 1))
    n = int(np.ceil(np.log2(degree))) + 1
    return [2 ** i for i in range(n - 1)] + [degree]


def smooth_displacement(X: np.ndarray, amplitude: float = 0.1, origin=None, span=None) -> np.ndarray:
    """A smooth displacement field with |grad u| ~ amplitude ("clamp-translate
    profile + MMS-shaped perturbation") at node coordinates X; host array in L layout.  With
    ``origin`` / ``span`` given it is a function of the absolute coordinates only, i.e. identical on
    every rank that shares a node."""
    origin = X.min(axis=0) if origin is None else np.asarray(origin, dtype=np.float64)
    span = np.maximum(X.max(axis=0) - X.min(axis=0), 1e-12) if span is None else np.asarray(span, dtype=np.float64)
    s = (X - origin) / span
    k = 2.0 * np.pi
    u = np.empty_like(X)
    u[:, 0] = amplitude * span[0] / k * np.sin(k * s[:, 1]) * np.cos(k * s[:, 2]) + 0.02 * amplitude * span[0] * s[:, 2]
    u[:, 1] = amplitude * span[1] / k * np.sin(k * s[:, 2]) * np.cos(k * s[:, 0]) - 0.05 * amplitude * span[1] * s[:, 2]
    u[:, 2] = amplitude * span[2] / k * np.sin(k * s[:, 0]) * np.cos(k * s[:, 1]) * 0.2 + 0.03 * amplitude * span[2] * s[:, 2]
    return u.reshape(-1)


@dataclass
class LevelData:  # CeedData, elasticity.h:218-240
    degree: int
    dofmap: DofMap
    mask: np.ndarray                      # uint8 Dirichlet mask over the L-vector
    Erestrictu: cd.ElemRestriction = None
    basisu: cd.Basis = None
    basisCtoF: cd.Basis = None
    qfJacob: cd.QFunction = None
    opJacob: cd.Operator = None
    opProlong: cd.Operator = None
    opRestrict: cd.Operator = None
    xceed: cd.Vector = None
    yceed: cd.Vector = None
    multinv: cd.Vector = None             # 1/multiplicity on this level (misc.c:115-143)


class SolidProblem:
    def __init__(self, ceed: cd.Ceed, mesh: HexMesh, degree: int, problem: str = "hyperFS",
                 nu: float = 0.3, E: float = 1.0, multigrid: str = "logarithmic", qextra: int = 0,
                 bc_sides: Optional[Sequence[int]] = None, bc_all_boundary: bool = False,
                 fused_bc: bool = True, shared_multiplicity=None, qf_callbacks=None):
        """``qf_callbacks``: name -> user callback pointer handed to CeedQFunctionCreateInterior, as the reference hands
        ``problemOptions[...].apply`` / ``.jacob`` (setuplibceed.c:470-474,822-824); a host backend calls it, this device
        backend resolves the name after ':' in the source string to its precompiled functor and ignores the pointer.
        ``fused_bc=False``: build the operator graphs exactly as the reference does and call NO extension of this
        backend (no Dirichlet flags folded into the offsets, no fused multiplicity scale): the drop-in form, in which the
        caller does what src/matops.c does around CeedOperatorApply.
        ``bc_sides``: side-set ids clamped (all three components; -bc_clamp, setupdm.c:171-190);
        ``bc_all_boundary``: the "marker" label of -test mode (setupdm.c:160-170)."""
        if problem not in PROBLEMS:
            raise ValueError(f"unknown problem {problem!r} (hyperFSIncomp is not implemented: dead code upstream)")
        self.ceed, self.mesh, self.problem, self.info = ceed, mesh, problem, PROBLEMS[problem]
        self.phys = np.array([nu, E], dtype=np.float64)  # Physics {nu, E}
        self.degrees = level_degrees(degree, multigrid)
        self.fine = len(self.degrees) - 1
        self.Q = degree + 1 + qextra
        self.fused_bc = fused_bc
        self._qf_cb = qf_callbacks or (lambda name: None)
        self.levels: List[LevelData] = []
        for p in self.degrees:
            dm = build_dofmap(mesh, p)
            if bc_all_boundary:
                nodes = boundary_nodes(mesh, dm)
            elif bc_sides:
                nodes = side_set_nodes(mesh, dm, bc_sides)
            else:
                nodes = np.zeros(0, dtype=np.int64)
            self.levels.append(LevelData(p, dm, dirichlet_mask(dm, nodes)))
        self._shared_multiplicity = shared_multiplicity
        self._setup_fine_level()
        for lv in range(len(self.levels)):
            self._setup_level(lv)

    # ------------------------------------------------------------------ setup
    def _setup_fine_level(self):
        c, mesh, Q = self.ceed, self.mesh, self.Q
        lv = self.levels[self.fine]
        P = lv.degree + 1
        ne = mesh.nelem
        nq = Q ** 3
        # restrictions (setuplibceed.c:279-318)
        self.Erestrictx = c.elem_restriction(ne, 8, 3, 1, 3 * mesh.nvert, (mesh.cells * 3).astype(np.int32))
        lv.Erestrictu = c.elem_restriction(ne, P ** 3, 3, 1, lv.dofmap.lsize, lv.dofmap.offsets())
        self.Erestrictqdi = c.strided_restriction(ne, nq, 10, 10 * ne * nq)
        self.ErestrictGradui = c.strided_restriction(ne, nq, 9, 9 * ne * nq) if self.info["state"] else None
        # coordinates (:323-329)
        self.xcoord = self.Erestrictx.create_lvector()
        self.xcoord.set_array(mesh.coords.reshape(-1), copy=True)
        # bases (:335-341)
        lv.basisu = c.basis_lagrange(3, 3, P, Q, cd.GAUSS)
        self.basisx = c.basis_lagrange(3, 3, 2, Q, cd.GAUSS)
        # persistent vectors (:353-361)
        assert lv.basisu.num_qpts == nq
        self.qdata = c.vector(10 * ne * nq)
        self.gradu = c.vector(9 * ne * nq) if self.info["state"] else None
        if self.gradu is not None:
            self.gradu.set_value(0.0)
        # geometric factors (:370-393)
        qf = c.qfunction("SetupGeo", f=self._qf_cb("SetupGeo"), source="qfunctions/common.h:SetupGeo")
        qf.add_input("dx", 9, cd.EVAL_GRAD).add_input("weight", 1, cd.EVAL_WEIGHT).add_output("qdata", 10, cd.EVAL_NONE)
        op = c.operator(qf)
        op.set_field("dx", self.Erestrictx, self.basisx, "active")
        op.set_field("weight", None, self.basisx, None)
        op.set_field("qdata", self.Erestrictqdi, None, "active")
        op.apply(self.xcoord, self.qdata)
        self.setupgeo_kernel = op.kernel_name
        op.destroy(); qf.destroy()
        # residual operator (:518-542)
        name = self.info["apply"]
        self.qfApply = c.qfunction(name, f=self._qf_cb(name), source=f"qfunctions/{self.info['src']}:{name}")
        self.qfApply.add_input("du", 9, cd.EVAL_GRAD).add_input("qdata", 10, cd.EVAL_NONE).add_output("dv", 9, cd.EVAL_GRAD)
        if self.info["state"]:
            self.qfApply.add_output("gradu", 9, cd.EVAL_NONE)
        self.qfApply.set_context(self.phys)
        self.opApply = c.operator(self.qfApply)
        self.opApply.set_field("du", lv.Erestrictu, lv.basisu, "active")
        self.opApply.set_field("qdata", self.Erestrictqdi, None, self.qdata)
        self.opApply.set_field("dv", lv.Erestrictu, lv.basisu, "active")
        if self.info["state"]:
            # the reference hands basisu to this EVAL_NONE field (:538-539); it is ignored
            self.opApply.set_field("gradu", self.ErestrictGradui, lv.basisu, self.gradu)
        if self.fused_bc:
            # residual: BC values stay in the input (matops.c:70-71), constrained rows dropped (:57)
            self._set_mask(self.opApply, lv.mask, mode=2)

    def _set_mask(self, op: cd.Operator, mask_in, mask_out=None, mode=3):
        L = self.ceed.L
        mi = np.ascontiguousarray(mask_in, dtype=np.uint8)
        pmi = mi.ctypes.data_as(C.POINTER(C.c_ubyte))
        if mask_out is None:
            L.chk(L.lib.CeedXOperatorSetDirichletMaskMode(op.h, cd.MEM_HOST, pmi, cd.c_int(mi.size), None, cd.c_int(0), mode))
        else:
            mo = np.ascontiguousarray(mask_out, dtype=np.uint8)
            L.chk(L.lib.CeedXOperatorSetDirichletMaskMode(
                op.h, cd.MEM_HOST, pmi, cd.c_int(mi.size), mo.ctypes.data_as(C.POINTER(C.c_ubyte)), cd.c_int(mo.size), mode))

    def _setup_level(self, level: int):
        c, Q = self.ceed, self.Q
        lv = self.levels[level]
        P = lv.degree + 1
        ne = self.mesh.nelem
        fine = self.levels[self.fine]
        if level != self.fine:  # (:771-784)
            lv.Erestrictu = c.elem_restriction(ne, P ** 3, 3, 1, lv.dofmap.lsize, lv.dofmap.offsets())
            lv.basisu = c.basis_lagrange(3, 3, P, Q, cd.GAUSS)
        if level != 0:          # (:799-803)
            lv.basisCtoF = c.basis_lagrange(3, 3, self.levels[level - 1].degree + 1, P, cd.GAUSS_LOBATTO)
        lv.xceed = c.vector(lv.dofmap.lsize)  # (:808-809)
        lv.yceed = c.vector(lv.dofmap.lsize)
        # Jacobian (:818-839)
        name = self.info["jacob"]
        lv.qfJacob = c.qfunction(name, f=self._qf_cb(name), source=f"qfunctions/{self.info['src']}:{name}")
        lv.qfJacob.add_input("deltadu", 9, cd.EVAL_GRAD).add_input("qdata", 10, cd.EVAL_NONE)
        if self.info["state"]:
            lv.qfJacob.add_input("gradu", 9, cd.EVAL_NONE)
        lv.qfJacob.add_output("deltadv", 9, cd.EVAL_GRAD)
        lv.qfJacob.set_context(self.phys, reported_size=8)  # sizeof(phys) quirk, :826
        lv.opJacob = c.operator(lv.qfJacob)
        lv.opJacob.set_field("deltadu", lv.Erestrictu, lv.basisu, "active")
        lv.opJacob.set_field("qdata", self.Erestrictqdi, None, self.qdata)
        lv.opJacob.set_field("deltadv", lv.Erestrictu, lv.basisu, "active")
        if self.info["state"]:
            lv.opJacob.set_field("gradu", self.ErestrictGradui, None, self.gradu)
        if self.fused_bc:
            self._set_mask(lv.opJacob, lv.mask, mode=3)
        # multiplicity (SetupProlongRestrictCtx, misc.c:115-143)
        lv.multinv = lv.Erestrictu.create_lvector()
        lv.Erestrictu.multiplicity(lv.multinv)
        if self._shared_multiplicity is not None:
            self._shared_multiplicity(level, lv.multinv)  # halo sum of the multiplicity (L2G add, G2L)
        lv.multinv.reciprocal()
        # transfer operators (:847-862)
        if level != 0:
            co = self.levels[level - 1]
            qfR = c.qfunction_identity(3, cd.EVAL_NONE, cd.EVAL_INTERP)   # elasticity.c:249-250
            qfP = c.qfunction_identity(3, cd.EVAL_INTERP, cd.EVAL_NONE)   # elasticity.c:251-252
            lv.opRestrict = c.operator(qfR)
            lv.opRestrict.set_field("input", lv.Erestrictu, None, "active")
            lv.opRestrict.set_field("output", co.Erestrictu, lv.basisCtoF, "active")
            lv.opProlong = c.operator(qfP)
            lv.opProlong.set_field("input", co.Erestrictu, lv.basisCtoF, "active")
            lv.opProlong.set_field("output", lv.Erestrictu, None, "active")
            L = self.ceed.L
            if self.fused_bc:   # fused_bc=False is the EXTENSION-FREE form: no CeedX* call at all; the caller applies multVec
                for op in (lv.opRestrict, lv.opProlong):   # (matops.c:149,176) and the Dirichlet handling (:33,57,106) itself
                    L.chk(L.lib.CeedXOperatorSetFineScale(op.h, lv.multinv.h))
            if self.fused_bc:
                self._set_mask(lv.opProlong, co.mask, lv.mask, mode=3)
                self._set_mask(lv.opRestrict, lv.mask, co.mask, mode=3)

    # --------------------------------------------------------------- sizes
    def lsize(self, level=None) -> int:
        return self.levels[self.fine if level is None else level].dofmap.lsize

    def n_free(self, level=None) -> int:
        """Unconstrained dofs of this rank's L-vector (Ugsz on one rank, elasticity.c:205-206)."""
        lv = self.levels[self.fine if level is None else level]
        return int(lv.mask.size - lv.mask.sum())

    # ------------------------------------------------- matops.c wrappers (on CeedVectors)
    def apply_jacobian(self, level: int, x: cd.Vector, y: cd.Vector):
        """ApplyJacobian_Ceed (matops.c:98-112): y = J(u) x, homogeneous Dirichlet."""
        self.levels[level].opJacob.apply(x, y)

    def form_residual(self, x: cd.Vector, y: cd.Vector):
        """FormResidual_Ceed (matops.c:63-79) after boundary values have been inserted into x."""
        self.opApply.apply(x, y)

    def prolong(self, level: int, xc: cd.Vector, yf: cd.Vector):
        """Prolong_Ceed (matops.c:115-157), coarse level-1 -> level."""
        self.levels[level].opProlong.apply(xc, yf)

    def prolong_add(self, level: int, xc: cd.Vector, yf: cd.Vector):
        """yf += Prolong_Ceed(xc): CeedOperatorApplyAdd of opProlong -- the correction of a V-cycle added in place (one rank: no interface
        sum between the prolongation and the addition).  Every fine node is written by its one owning element: the same bits as
        prolong into a scratch vector followed by an axpy."""
        L = self.ceed.L
        L.chk(L.lib.CeedOperatorApplyAdd(self.levels[level].opProlong.h, xc.h, yf.h, cd.C.c_void_p(L.REQUEST_IMMEDIATE)))

    def restrict(self, level: int, xf: cd.Vector, yc: cd.Vector):
        """Restrict_Ceed (matops.c:160-203), level -> level-1."""
        self.levels[level].opRestrict.apply(xf, yc)

    def get_diag(self, level: int, d: cd.Vector):
        """GetDiag_Ceed (matops.c:206-244)."""
        self.levels[level].opJacob.assemble_diagonal(d)

    # --------------------------------------------------------------- helpers
    def smooth_state(self, amplitude: float = 0.1, origin=None, span=None) -> np.ndarray:
        return smooth_displacement(self.levels[self.fine].dofmap.node_coords, amplitude, origin, span)

    def destroy(self):
        for lv in self.levels:
            for o in (lv.opJacob, lv.opProlong, lv.opRestrict, lv.qfJacob, lv.Erestrictu, lv.basisu, lv.basisCtoF,
                      lv.xceed, lv.yceed, lv.multinv):
                if o is not None:
                    o.destroy()
        for o in (self.opApply, self.qfApply, self.Erestrictx, self.Erestrictqdi, self.ErestrictGradui, self.basisx,
                  self.xcoord, self.qdata, self.gradu):
            if o is not None:
                o.destroy()
