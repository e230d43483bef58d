"""ctypes binding of the C++ host harness (include/solid_harness.h, csrc/solid_harness.cpp).

The harness restates the reference's ``SetupLibceedFineLevel`` / ``SetupLibceedLevel`` /
``src/matops.c`` callbacks in C++ over ``include/ceed.h`` only; this module just feeds it the arrays
DMPlex would provide (built by ``mesh.py``) and wraps the handles it returns.  ``bench.py`` times the
Jacobian apply through this path (``ApplyJacobian_Ceed``), so the measured host path is the native one.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from . import ceed as cd
from .mesh import HexMesh, boundary_nodes, build_dofmap, dirichlet_mask, side_set_nodes
from .solid import level_degrees

PROBLEM_TYPES = {"linElas": 0, "hyperSS": 1, "hyperFS": 2}
PRODUCT_HARNESS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libsolid_harness_mi355x.so")


def _wrap_vector(ceed: cd.Ceed, handle: int, n: int) -> cd.Vector:
    v = cd.Vector.__new__(cd.Vector)
    v.ceed, v.L, v.n, v._keep = ceed, ceed.L, n, None
    v.h = C.c_void_p(handle)
    return v


def _wrap_operator(ceed: cd.Ceed, handle: int) -> cd.Operator:
    o = cd.Operator.__new__(cd.Operator)
    o.ceed, o.L, o.qf, o._keep = ceed, ceed.L, None, []
    o.h = C.c_void_p(handle)
    return o


class SolidApp:
    """One problem instance of the C++ harness on a given Ceed (product or, in tests, the oracle)."""

    def __init__(self, ceed: cd.Ceed, mesh: HexMesh, degree: int, problem: str = "hyperFS", nu: float = 0.3,
                 E: float = 1.0, multigrid: str = "logarithmic", qextra: int = 0,
                 bc_sides: Optional[Sequence[int]] = None, bc_all_boundary: bool = False,
                 harness_lib: str = PRODUCT_HARNESS):
        if not os.path.exists(harness_lib):
            raise cd.CeedError(f"harness library not found: {harness_lib} (run __graft_entry__.build())")
        # the harness library NEEDs its Ceed backend; load the backend first through CeedLib so the
        # one-HIP-runtime rule of ceed.py applies
        self.ceed, self.mesh, self.problem = ceed, mesh, problem
        self.H = C.CDLL(harness_lib, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
        self.degrees = level_degrees(degree, multigrid)
        self.fine = len(self.degrees) - 1
        self.Q = degree + 1 + qextra
        self.dofmaps, self.masks = [], []
        for p in self.degrees:
            dm = build_dofmap(mesh, p)
            if bc_all_boundary:
                nodes = boundary_nodes(mesh, dm)
            elif bc_sides:
                nodes = side_set_nodes(mesh, dm, bc_sides)
            else:
                nodes = np.zeros(0, dtype=np.int64)
            self.dofmaps.append(dm)
            self.masks.append(np.ascontiguousarray(dirichlet_mask(dm, nodes), dtype=np.uint8))
        nl = len(self.degrees)
        degs = (C.c_int32 * nl)(*self.degrees)
        self._offs = [np.ascontiguousarray(dm.offsets(), dtype=np.int32) for dm in self.dofmaps]
        offs = (C.POINTER(C.c_int32) * nl)(*[o.ctypes.data_as(C.POINTER(C.c_int32)) for o in self._offs])
        lsz = (C.c_int32 * nl)(*[dm.lsize for dm in self.dofmaps])
        msk = (C.POINTER(C.c_ubyte) * nl)(*[m.ctypes.data_as(C.POINTER(C.c_ubyte)) for m in self.masks])
        coords = np.ascontiguousarray(mesh.coords, dtype=np.float64)
        cells = np.ascontiguousarray(mesh.cells, dtype=np.int32)
        self.h = C.c_void_p()
        rc = self.H.SolidAppCreate(ceed.h, C.c_int(PROBLEM_TYPES[problem]), C.c_double(nu), C.c_double(E),
                                   C.c_int32(nl), degs, C.c_int32(qextra), C.c_int32(mesh.nelem), C.c_int32(mesh.nvert),
                                   coords.ctypes.data_as(C.POINTER(C.c_double)), cells.ctypes.data_as(C.POINTER(C.c_int32)),
                                   offs, lsz, msk, C.byref(self.h))
        ceed.L.chk(rc)
        # handles
        q, g = C.c_void_p(), C.c_void_p()
        self.H.SolidAppGetVectors(self.h, C.byref(q), C.byref(g))
        nq = self.Q ** 3
        self.qdata = _wrap_vector(ceed, q.value, 10 * mesh.nelem * nq)
        self.gradu = _wrap_vector(ceed, g.value, 9 * mesh.nelem * nq) if g.value else None
        self.opJacob: List[cd.Operator] = []
        for l in range(nl):
            oj = C.c_void_p()
            self.H.SolidAppGetLevelOperators(self.h, C.c_int32(l), C.byref(oj), None, None)
            self.opJacob.append(_wrap_operator(ceed, oj.value))
        oa = C.c_void_p()
        self.H.SolidAppGetResidualOperator(self.h, C.byref(oa))
        self.opApply = _wrap_operator(ceed, oa.value)

    def lsize(self, level=None) -> int:
        return self.dofmaps[self.fine if level is None else level].lsize

    def n_free(self, level=None) -> int:
        m = self.masks[self.fine if level is None else level]
        return int(m.size - m.sum())

    def multinv(self, level) -> cd.Vector:
        v = C.c_void_p()
        self.H.SolidAppGetMultiplicityInverse(self.h, C.c_int32(level), C.byref(v))
        return _wrap_vector(self.ceed, v.value, self.lsize(level))

    # src/matops.c
    def apply_jacobian(self, level: int, x: cd.Vector, y: cd.Vector):
        self.ceed.L.chk(self.H.ApplyJacobian_Ceed(self.h, C.c_int32(level), x.h, y.h))

    def form_residual(self, x: cd.Vector, y: cd.Vector):
        self.ceed.L.chk(self.H.FormResidual_Ceed(self.h, x.h, y.h))

    def prolong(self, level: int, xc: cd.Vector, yf: cd.Vector):
        self.ceed.L.chk(self.H.Prolong_Ceed(self.h, C.c_int32(level), xc.h, yf.h))

    def restrict(self, level: int, xf: cd.Vector, yc: cd.Vector):
        self.ceed.L.chk(self.H.Restrict_Ceed(self.h, C.c_int32(level), xf.h, yc.h))

    def get_diag(self, level: int, d: cd.Vector):
        self.ceed.L.chk(self.H.GetDiag_Ceed(self.h, C.c_int32(level), d.h))

    def set_halo(self, level: int, halo):
        """Attach the interface sum of one level (a halo.RcclHalo or a raw CeedXHalo handle; None clears): every matops
        function then ends with it, as the reference's end with DMLocalToGlobal(ADD_VALUES)."""
        h = None if halo is None else getattr(halo, "h", halo)
        self.ceed.L.chk(self.H.SolidAppSetHalo(self.h, C.c_int32(level), h))

    def set_smoother_nu(self, nu: float):
        self.ceed.L.chk(self.H.SolidAppSetSmootherNu(self.h, C.c_double(nu)))

    def destroy(self):
        if self.h:
            self.H.SolidAppDestroy(C.byref(self.h))
