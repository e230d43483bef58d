"""ctypes binding of the C-ABI boundary declared in ``include/ceed.h``.

The binding is library-agnostic: it is handed the path of a shared object that
exports the ``Ceed*`` entry points.  The package itself only ever passes the
product library (``csrc/libceed_mi355x.so``, resource ``/gpu/hip/mi355x``);
tests, ``smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may additionally
bind the CPU oracle (``oracle/liboracle_ceed.so``) through the same class so the
parity tests drive both backends with identical call sequences -- the sequences
of the reference's ``src/setuplibceed.c`` / ``src/matops.c``.

Raw pointers only cross the boundary: numpy arrays on the host side, integer
device addresses (e.g. ``torch.Tensor.data_ptr()``) on the device side.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

c_int = C.c_int32
c_scalar_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int32)

MEM_HOST, MEM_DEVICE = 0, 1
COPY_VALUES, USE_POINTER, OWN_POINTER = 0, 1, 2
NOTRANSPOSE, TRANSPOSE = 0, 1
EVAL_NONE, EVAL_INTERP, EVAL_GRAD, EVAL_WEIGHT = 0, 1, 2, 16
GAUSS, GAUSS_LOBATTO = 0, 1

QFUNCTION_USER = C.CFUNCTYPE(C.c_int, C.c_void_p, c_int, C.POINTER(c_scalar_p), C.POINTER(c_scalar_p))

PRODUCT_LIB = os.environ.get(  # override only for A/B-ing kernel builds (tools/); must still be a *mi355x* build
    "CEEDPETSCSOLID_MI355X_LIB",
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libceed_mi355x.so"))


class CeedError(RuntimeError):
    pass


class CeedLib:
    """One loaded shared object exporting the ``include/ceed.h`` ABI."""

    # every symbol include/ceed.h declares (functions); checked by tests
    FUNCTIONS = [
        "CeedInit", "CeedDestroy", "CeedGetResource", "CeedGetPreferredMemType",
        "CeedVectorCreate", "CeedVectorSetArray", "CeedVectorTakeArray", "CeedVectorSetValue",
        "CeedVectorSyncArray", "CeedVectorGetArray", "CeedVectorGetArrayRead",
        "CeedVectorRestoreArray", "CeedVectorRestoreArrayRead", "CeedVectorGetLength",
        "CeedVectorReciprocal", "CeedVectorDestroy",
        "CeedElemRestrictionCreate", "CeedElemRestrictionCreateStrided",
        "CeedElemRestrictionCreateVector", "CeedElemRestrictionApply",
        "CeedElemRestrictionGetMultiplicity", "CeedElemRestrictionDestroy",
        "CeedBasisCreateTensorH1Lagrange", "CeedBasisGetNumQuadraturePoints", "CeedBasisGetNumNodes",
        "CeedBasisApply", "CeedBasisDestroy", "CeedGaussQuadrature", "CeedLobattoQuadrature",
        "CeedBasisGetInterp1D", "CeedBasisGetGrad1D", "CeedBasisGetQWeights1D",
        "CeedQFunctionCreateInterior", "CeedQFunctionCreateIdentity", "CeedQFunctionAddInput",
        "CeedQFunctionAddOutput", "CeedQFunctionSetContext", "CeedQFunctionDestroy",
        "CeedOperatorCreate", "CeedCompositeOperatorCreate", "CeedCompositeOperatorAddSub",
        "CeedOperatorSetField", "CeedOperatorApply", "CeedOperatorApplyAdd",
        "CeedOperatorLinearAssembleDiagonal", "CeedOperatorDestroy",
        "CeedXSetErrorReturn", "CeedXLastError", "CeedXSetStream", "CeedXSynchronize",
        "CeedXOperatorGetKernelName", "CeedXOperatorSetDirichletMask",
        "CeedXOperatorSetTiming", "CeedXOperatorGetTiming", "CeedXOperatorSetDirichletMaskMode", "CeedXOperatorGetLaunchInfo", "CeedXOperatorApplyWithHalo", "CeedXCommAllReduce",
        "CeedXCommGetUniqueId", "CeedXCommInit", "CeedXCommDestroy", "CeedXCommGetSize", "CeedXHaloCreate", "CeedXHaloStart", "CeedXHaloFinish", "CeedXHaloDestroy",
        "CeedXOperatorSetFineScale", "CeedXOperatorSetOverlapSplit", "CeedXOperatorApplyPhase",
        "CeedXVectorPointwiseMult", "CeedXVectorAXPBY", "CeedXVectorDot", "CeedXVectorChebyshevUpdate",
        "CeedXGraphBeginCapture", "CeedXGraphEndCapture", "CeedXGraphLaunch", "CeedXGraphDestroy", "CeedXGraphIsStale",
        "CeedXOperatorApplyChebyshev", "CeedXOperatorApplyResidual", "CeedXClockProbe",
        "CeedXVectorChebyshevStart", "CeedXVectorChebyshevStep", "CeedXVectorWAXPBY", "CeedXVectorDotTo", "CeedXScalarDivide", "CeedXVectorAXPBYScalars",
        "CeedXCsrCreate", "CeedXCsrAssemble", "CeedXCsrApply", "CeedXCsrGetDiagonal", "CeedXCsrDestroy",
        "CeedXCsrCreateRect", "CeedXCsrCreateProduct", "CeedXCsrGetPattern", "CeedXCsrUpdate", "CeedXCsrGetValues", "CeedXCsrInvertDenseSPD",
    ]
    DATA = [
        "CeedMemTypes", "CEED_VECTOR_ACTIVE", "CEED_VECTOR_NONE", "CEED_ELEMRESTRICTION_NONE",
        "CEED_BASIS_COLLOCATED", "CEED_QFUNCTION_NONE", "CEED_REQUEST_IMMEDIATE",
        "CEED_REQUEST_ORDERED", "CEED_STRIDES_BACKEND",
    ]

    def __init__(self, path: str = PRODUCT_LIB):
        if not os.path.exists(path):
            raise CeedError(
                f"Ceed backend library not found: {path} -- build it first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
        self.path = path
        if "mi355x" in os.path.basename(path):
            # One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 /
            # libhsa-runtime64 (soname libamdhip64.so.7, same as /opt/rocm's); two copies in
            # one process cannot both open the device.  Importing torch first makes the
            # loader satisfy this library's NEEDED entry with torch's already-loaded copy
            # (measured on the GPU box: tools/diag_hip_runtime.py).  A plain C host without
            # torch gets /opt/rocm's runtime through the usual search path.
            import torch  # noqa: F401
        # RTLD_LOCAL: the oracle and the product export the same Ceed* names
        self.lib = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
        L = self.lib
        L.CeedXLastError.restype = C.c_char_p
        L.CeedXSetErrorReturn(1)
        vp = C.c_void_p
        self.VECTOR_ACTIVE = vp.in_dll(L, "CEED_VECTOR_ACTIVE").value
        self.VECTOR_NONE = vp.in_dll(L, "CEED_VECTOR_NONE").value
        self.ELEMRESTRICTION_NONE = vp.in_dll(L, "CEED_ELEMRESTRICTION_NONE").value
        self.BASIS_COLLOCATED = vp.in_dll(L, "CEED_BASIS_COLLOCATED").value
        self.QFUNCTION_NONE = vp.in_dll(L, "CEED_QFUNCTION_NONE").value
        self.REQUEST_IMMEDIATE = vp.in_dll(L, "CEED_REQUEST_IMMEDIATE").value
        self.STRIDES_BACKEND = (c_int * 3).in_dll(L, "CEED_STRIDES_BACKEND")

    def chk(self, rc: int):
        if rc:
            msg = self.lib.CeedXLastError()
            raise CeedError(msg.decode() if msg else f"Ceed error {rc}")

    def missing_symbols(self):
        return [s for s in self.FUNCTIONS + self.DATA if not hasattr(self.lib, s)]


def _np_f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


class Csr:
    """Assembled sparse operator on L-vectors (CeedXCsr*): the coarse multigrid level."""

    def __init__(self, ceed: "Ceed", rowptr, cols, coo_slot, unit_rows=()):
        self.L, self._ceed = ceed.L, ceed
        self.h = C.c_void_p()
        rp = np.ascontiguousarray(rowptr, dtype=np.int32); cl = np.ascontiguousarray(cols, dtype=np.int32)
        sl = np.ascontiguousarray(coo_slot, dtype=np.int32); ur = np.ascontiguousarray(unit_rows, dtype=np.int32)
        self.nrows, self.nnz, self.ncoo = rp.size - 1, int(rp[-1]), sl.size
        self.L.chk(self.L.lib.CeedXCsrCreate(ceed.h, c_int(self.nrows), rp.ctypes.data_as(c_int_p), cl.ctypes.data_as(c_int_p),
                                             c_int(sl.size), sl.ctypes.data_as(c_int_p), c_int(ur.size),
                                             ur.ctypes.data_as(c_int_p), C.byref(self.h)))

    def assemble(self, coo_values: "Vector"):
        self.L.chk(self.L.lib.CeedXCsrAssemble(self.h, coo_values.h))

    def apply(self, x: "Vector", y: "Vector"):
        self.L.chk(self.L.lib.CeedXCsrApply(self.h, x.h, y.h))

    def diagonal(self, d: "Vector"):
        self.L.chk(self.L.lib.CeedXCsrGetDiagonal(self.h, d.h))

    # ---- pieces of the aggregation hierarchy (amg.py) ------------------------------------------------------
    @classmethod
    def rect(cls, ceed: "Ceed", nrows: int, ncols: int, rowptr, cols, vals=None) -> "Csr":
        """nrows x ncols matrix with fixed values (``vals``) or values computed by ``update()`` (``vals=None``)."""
        self = cls.__new__(cls)
        self.L, self.h, self._ceed = ceed.L, C.c_void_p(), ceed
        rp = np.ascontiguousarray(rowptr, dtype=np.int32); cl = np.ascontiguousarray(cols, dtype=np.int32)
        if rp.size != nrows + 1 or cl.size != int(rp[-1]):
            raise ValueError("rowptr / cols do not describe an nrows-row pattern")
        self.nrows, self.ncols, self.nnz, self.ncoo = nrows, ncols, int(rp[-1]), 0
        vp = None
        if vals is not None:
            va = np.ascontiguousarray(vals, dtype=np.float64)
            if va.size != self.nnz:
                raise ValueError("one value per pattern entry expected")
            vp = va.ctypes.data_as(C.POINTER(C.c_double))
        self.L.chk(self.L.lib.CeedXCsrCreateRect(ceed.h, c_int(nrows), c_int(ncols), rp.ctypes.data_as(c_int_p),
                                                 cl.ctypes.data_as(c_int_p), vp, C.byref(self.h)))
        return self

    @classmethod
    def product(cls, left: "Csr", right: "Csr", variable: int, dense: bool = False) -> "Csr":
        """left * right with one operand of fixed values and the other (``variable``: 0 left, 1 right) read at every
        ``update()``; pattern and term lists are worked out by the library (CeedXCsrCreateProduct)."""
        self = cls.__new__(cls)
        self.L, self.h, self._ceed = left.L, C.c_void_p(), left._ceed
        self.L.chk(self.L.lib.CeedXCsrCreateProduct(left.h, right.h, c_int(variable), c_int(1 if dense else 0), C.byref(self.h)))
        self.nrows, self.ncols, self.nnz = self.pattern()[:3]
        self.ncoo = 0
        return self

    def pattern(self):
        """(nrows, ncols, nnz, rowptr, cols): copies of the library's host pattern."""
        nr, nc, nz = c_int(), c_int(), c_int()
        rp, cl = c_int_p(), c_int_p()
        self.L.chk(self.L.lib.CeedXCsrGetPattern(self.h, C.byref(nr), C.byref(nc), C.byref(nz), C.byref(rp), C.byref(cl)))
        rowptr = np.ctypeslib.as_array(rp, shape=(nr.value + 1,)).copy()
        cols = np.ctypeslib.as_array(cl, shape=(max(nz.value, 1),))[:nz.value].copy() if nz.value else np.zeros(0, np.int32)
        return nr.value, nc.value, nz.value, rowptr, cols

    def update(self):
        self.L.chk(self.L.lib.CeedXCsrUpdate(self.h))

    def values(self, ceed: "Ceed" = None) -> np.ndarray:
        v = (ceed or self._ceed).vector(max(self.nnz, 1))
        self.L.chk(self.L.lib.CeedXCsrGetValues(self.h, v.h))
        out = v.to_numpy()[:self.nnz].copy()
        v.destroy()
        return out

    def invert_dense_spd(self):
        self.L.chk(self.L.lib.CeedXCsrInvertDenseSPD(self.h))

    def destroy(self):
        if self.h:
            self.L.lib.CeedXCsrDestroy(C.byref(self.h))
            self.h = None


class Graph:
    def __init__(self, L, h):
        self.L, self.h = L, h

    def launch(self):
        self.L.chk(self.L.lib.CeedXGraphLaunch(self.h))

    def stale(self) -> int:
        """What CeedXGraphLaunch would refuse for (0: nothing) -- local to this rank."""
        st = c_int()
        self.L.chk(self.L.lib.CeedXGraphIsStale(self.h, C.byref(st)))
        return st.value

    def destroy(self):
        if self.h:
            self.L.lib.CeedXGraphDestroy(C.byref(self.h))
            self.h = None


class Ceed:
    def __init__(self, lib: CeedLib, resource: str):
        self.L = lib
        self.h = C.c_void_p()
        lib.chk(lib.lib.CeedInit(resource.encode(), C.byref(self.h)))

    @property
    def resource(self) -> str:
        s = C.c_char_p()
        self.L.chk(self.L.lib.CeedGetResource(self.h, C.byref(s)))
        return s.value.decode()

    @property
    def preferred_memtype(self) -> int:
        m = C.c_int()
        self.L.chk(self.L.lib.CeedGetPreferredMemType(self.h, C.byref(m)))
        return m.value

    def set_stream(self, hip_stream: int):
        self.L.chk(self.L.lib.CeedXSetStream(self.h, C.c_void_p(hip_stream)))

    def synchronize(self):
        self.L.chk(self.L.lib.CeedXSynchronize(self.h))

    def clock_probe(self, spin_us: int = 2000) -> float:
        """Shader clock (GHz) while the work queued on this Ceed's stream runs (CeedXClockProbe)."""
        g = C.c_double()
        self.L.chk(self.L.lib.CeedXClockProbe(self.h, C.c_int(spin_us), C.byref(g)))
        return g.value

    def comm_size(self):
        """(ranks, this rank) of the Ceed's RCCL communicator as RCCL reports them (CeedXCommGetSize); (0, -1) without one."""
        n, r = C.c_int(), C.c_int()
        self.L.chk(self.L.lib.CeedXCommGetSize(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def capture(self, fn) -> "Graph":
        """Record the device work `fn()` queues on this Ceed into a hipGraph (CeedXGraph*)."""
        self.L.chk(self.L.lib.CeedXGraphBeginCapture(self.h))
        g, ok = C.c_void_p(), False
        try:
            fn()
            ok = True
        finally:
            rc = self.L.lib.CeedXGraphEndCapture(self.h, C.byref(g))
            if not ok and rc == 0 and g:           # fn() raised: the recording is ended and dropped, the error propagates
                self.L.lib.CeedXGraphDestroy(C.byref(g))
        self.L.chk(rc)
        return Graph(self.L, g)

    def destroy(self):
        if self.h:
            self.L.lib.CeedDestroy(C.byref(self.h))

    # -- factories ---------------------------------------------------------
    def vector(self, n: int) -> "Vector":
        return Vector(self, n)

    def elem_restriction(self, nelem, elemsize, ncomp, compstride, lsize, offsets) -> "ElemRestriction":
        return ElemRestriction(self, nelem, elemsize, ncomp, compstride, lsize, offsets=offsets)

    def strided_restriction(self, nelem, elemsize, ncomp, lsize, strides=None) -> "ElemRestriction":
        return ElemRestriction(self, nelem, elemsize, ncomp, 0, lsize, strides=strides, strided=True)

    def basis_lagrange(self, dim, ncomp, P, Q, qmode) -> "Basis":
        return Basis(self, dim, ncomp, P, Q, qmode)

    def qfunction(self, name: str, f=None, source: Optional[str] = None) -> "QFunction":
        return QFunction(self, name, f, source)

    def qfunction_identity(self, size, inmode, outmode) -> "QFunction":
        return QFunction(self, "Identity", identity=(size, inmode, outmode))

    def operator(self, qf: "QFunction") -> "Operator":
        return Operator(self, qf)


class Vector:
    def __init__(self, ceed: Ceed, n: int):
        self.ceed, self.L, self.n = ceed, ceed.L, int(n)
        self.h = C.c_void_p()
        self._keep = None
        self.L.chk(self.L.lib.CeedVectorCreate(ceed.h, c_int(n), C.byref(self.h)))

    def set_array(self, arr: np.ndarray, copy=True):
        a = _np_f64(arr)
        assert a.size == self.n, (a.size, self.n)
        if not copy:
            self._keep = a
        self.L.chk(self.L.lib.CeedVectorSetArray(
            self.h, MEM_HOST, COPY_VALUES if copy else USE_POINTER, a.ctypes.data_as(c_scalar_p)))
        return self

    def set_device_pointer(self, ptr: int):
        """Borrow a device buffer (CEED_MEM_DEVICE, CEED_USE_POINTER), matops.c:40-41."""
        self.L.chk(self.L.lib.CeedVectorSetArray(self.h, MEM_DEVICE, USE_POINTER, C.cast(C.c_void_p(ptr), c_scalar_p)))
        return self

    def take_array(self, mtype=MEM_HOST):
        self.L.chk(self.L.lib.CeedVectorTakeArray(self.h, mtype, None))
        self._keep = None

    def set_value(self, v: float):
        self.L.chk(self.L.lib.CeedVectorSetValue(self.h, C.c_double(v)))
        return self

    def to_numpy(self) -> np.ndarray:
        p = c_scalar_p()
        self.L.chk(self.L.lib.CeedVectorGetArrayRead(self.h, MEM_HOST, C.byref(p)))
        out = np.ctypeslib.as_array(p, shape=(self.n,)).copy() if self.n else np.zeros(0)
        self.L.chk(self.L.lib.CeedVectorRestoreArrayRead(self.h, C.byref(p)))
        return out

    def device_pointer(self) -> int:
        """Device address of the vector's storage (valid until the next Set/Take)."""
        p = c_scalar_p()
        self.L.chk(self.L.lib.CeedVectorGetArray(self.h, MEM_DEVICE, C.byref(p)))
        addr = C.cast(p, C.c_void_p).value
        self.L.chk(self.L.lib.CeedVectorRestoreArray(self.h, C.byref(p)))
        return addr

    def reciprocal(self):
        self.L.chk(self.L.lib.CeedVectorReciprocal(self.h))

    def destroy(self):
        if self.h:
            self.L.lib.CeedVectorDestroy(C.byref(self.h))


class ElemRestriction:
    def __init__(self, ceed, nelem, elemsize, ncomp, compstride, lsize, offsets=None, strides=None, strided=False):
        self.ceed, self.L = ceed, ceed.L
        self.nelem, self.elemsize, self.ncomp, self.lsize = int(nelem), int(elemsize), int(ncomp), int(lsize)
        self.h = C.c_void_p()
        if strided:
            st = self.L.STRIDES_BACKEND if strides is None else (c_int * 3)(*strides)
            self.L.chk(self.L.lib.CeedElemRestrictionCreateStrided(
                ceed.h, c_int(nelem), c_int(elemsize), c_int(ncomp), c_int(lsize), st, C.byref(self.h)))
        else:
            off = np.ascontiguousarray(offsets, dtype=np.int32)
            assert off.size == nelem * elemsize
            self.L.chk(self.L.lib.CeedElemRestrictionCreate(
                ceed.h, c_int(nelem), c_int(elemsize), c_int(ncomp), c_int(compstride), c_int(lsize),
                MEM_HOST, COPY_VALUES, off.ctypes.data_as(c_int_p), C.byref(self.h)))

    def create_lvector(self) -> Vector:
        v = Vector.__new__(Vector)
        v.ceed, v.L, v.n, v._keep = self.ceed, self.L, self.lsize, None
        v.h = C.c_void_p()
        self.L.chk(self.L.lib.CeedElemRestrictionCreateVector(self.h, C.byref(v.h), None))
        return v

    def create_evector(self) -> Vector:
        v = Vector.__new__(Vector)
        v.ceed, v.L, v.n, v._keep = self.ceed, self.L, self.nelem * self.elemsize * self.ncomp, None
        v.h = C.c_void_p()
        self.L.chk(self.L.lib.CeedElemRestrictionCreateVector(self.h, None, C.byref(v.h)))
        return v

    def apply(self, tmode, u: Vector, v: Vector):
        self.L.chk(self.L.lib.CeedElemRestrictionApply(self.h, tmode, u.h, v.h, C.c_void_p(self.L.REQUEST_IMMEDIATE)))

    def multiplicity(self, mult: Vector):
        self.L.chk(self.L.lib.CeedElemRestrictionGetMultiplicity(self.h, mult.h))

    def destroy(self):
        if self.h:
            self.L.lib.CeedElemRestrictionDestroy(C.byref(self.h))


class Basis:
    def __init__(self, ceed, dim, ncomp, P, Q, qmode):
        self.ceed, self.L = ceed, ceed.L
        self.dim, self.ncomp, self.P, self.Q, self.qmode = dim, ncomp, P, Q, qmode
        self.h = C.c_void_p()
        self.L.chk(self.L.lib.CeedBasisCreateTensorH1Lagrange(
            ceed.h, c_int(dim), c_int(ncomp), c_int(P), c_int(Q), qmode, C.byref(self.h)))

    @property
    def num_qpts(self) -> int:
        q = c_int()
        self.L.chk(self.L.lib.CeedBasisGetNumQuadraturePoints(self.h, C.byref(q)))
        return q.value

    def _table(self, fn, shape):
        p = c_scalar_p()
        self.L.chk(fn(self.h, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=shape).copy()

    @property
    def interp1d(self):
        return self._table(self.L.lib.CeedBasisGetInterp1D, (self.Q, self.P))

    @property
    def grad1d(self):
        return self._table(self.L.lib.CeedBasisGetGrad1D, (self.Q, self.P))

    @property
    def qweight1d(self):
        return self._table(self.L.lib.CeedBasisGetQWeights1D, (self.Q,))

    def apply(self, nelem, tmode, emode, u: Vector, v: Vector):
        self.L.chk(self.L.lib.CeedBasisApply(self.h, c_int(nelem), tmode, emode, u.h, v.h))

    def destroy(self):
        if self.h:
            self.L.lib.CeedBasisDestroy(C.byref(self.h))


class QFunction:
    """``name`` is the reference QFunction name (e.g. ``HyperFSdF``); ``source`` the
    "file:name" locator the reference passes (setuplibceed.c:49-53)."""

    def __init__(self, ceed, name, f=None, source=None, identity=None):
        self.ceed, self.L, self.name = ceed, ceed.L, name
        self.h = C.c_void_p()
        self._ctx = None
        if identity is not None:
            size, inmode, outmode = identity
            self.L.chk(self.L.lib.CeedQFunctionCreateIdentity(ceed.h, c_int(size), inmode, outmode, C.byref(self.h)))
        else:
            src = (source or f"qfunctions/{name}.h:{name}").encode()
            fptr = C.cast(f, C.c_void_p) if f is not None else C.c_void_p(0)
            self.L.chk(self.L.lib.CeedQFunctionCreateInterior(ceed.h, c_int(1), fptr, src, C.byref(self.h)))

    def add_input(self, name, size, emode):
        self.L.chk(self.L.lib.CeedQFunctionAddInput(self.h, name.encode(), c_int(size), emode))
        return self

    def add_output(self, name, size, emode):
        self.L.chk(self.L.lib.CeedQFunctionAddOutput(self.h, name.encode(), c_int(size), emode))
        return self

    def set_context(self, values: Sequence[float], reported_size: Optional[int] = None):
        """Borrowed context of doubles (Physics {nu, E}: elasticity.h:33-36).
        ``reported_size`` lets tests reproduce the reference's sizeof(pointer) quirk."""
        self._ctx = _np_f64(values).copy()
        size = self._ctx.nbytes if reported_size is None else reported_size
        self.L.chk(self.L.lib.CeedQFunctionSetContext(self.h, self._ctx.ctypes.data_as(C.c_void_p), C.c_size_t(size)))
        return self

    def destroy(self):
        if self.h:
            self.L.lib.CeedQFunctionDestroy(C.byref(self.h))


class Operator:
    def __init__(self, ceed, qf: QFunction):
        self.ceed, self.L, self.qf = ceed, ceed.L, qf
        self.h = C.c_void_p()
        none = C.c_void_p(self.L.QFUNCTION_NONE)
        self.L.chk(self.L.lib.CeedOperatorCreate(ceed.h, qf.h, none, none, C.byref(self.h)))
        self._keep = []

    def set_field(self, name, rstr, basis, vec):
        """rstr/basis/vec: wrapper objects, or None for the NONE/COLLOCATED sentinels,
        or the string "active" for CEED_VECTOR_ACTIVE."""
        L = self.L
        r = C.c_void_p(L.ELEMRESTRICTION_NONE) if rstr is None else rstr.h
        b = C.c_void_p(L.BASIS_COLLOCATED) if basis is None else basis.h
        if isinstance(vec, str):
            assert vec == "active"
            v = C.c_void_p(L.VECTOR_ACTIVE)
        elif vec is None:
            v = C.c_void_p(L.VECTOR_NONE)
        else:
            v = vec.h
        L.chk(L.lib.CeedOperatorSetField(self.h, name.encode(), r, b, v))
        self._keep.append((rstr, basis, vec))
        return self

    def apply(self, vin: Optional[Vector], vout: Optional[Vector]):
        L = self.L
        i = vin.h if vin is not None else C.c_void_p(L.VECTOR_NONE)
        o = vout.h if vout is not None else C.c_void_p(L.VECTOR_NONE)
        L.chk(L.lib.CeedOperatorApply(self.h, i, o, C.c_void_p(L.REQUEST_IMMEDIATE)))

    def assemble_diagonal(self, vec: Vector):
        self.L.chk(self.L.lib.CeedOperatorLinearAssembleDiagonal(self.h, vec.h, C.c_void_p(self.L.REQUEST_IMMEDIATE)))

    @property
    def kernel_name(self) -> str:
        s = C.c_char_p()
        self.L.chk(self.L.lib.CeedXOperatorGetKernelName(self.h, C.byref(s)))
        return s.value.decode() if s.value else ""

    def set_dirichlet_mask(self, mask: Optional[np.ndarray]):
        if mask is None:
            self.L.chk(self.L.lib.CeedXOperatorSetDirichletMask(self.h, MEM_HOST, None, c_int(0)))
        else:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            self.L.chk(self.L.lib.CeedXOperatorSetDirichletMask(
                self.h, MEM_HOST, m.ctypes.data_as(C.POINTER(C.c_ubyte)), c_int(m.size)))

    def set_overlap_split(self, n_leading_elems: int, priority: Optional[np.ndarray]):
        """CeedXOperatorSetOverlapSplit: leading elements / priority L-vector entries of the split-phase apply."""
        if priority is None:
            self.L.chk(self.L.lib.CeedXOperatorSetOverlapSplit(self.h, c_int(0), None, c_int(0)))
        else:
            m = np.ascontiguousarray(priority, dtype=np.uint8)
            self.L.chk(self.L.lib.CeedXOperatorSetOverlapSplit(
                self.h, c_int(n_leading_elems), m.ctypes.data_as(C.POINTER(C.c_ubyte)), c_int(m.size)))

    def apply_phase(self, vin: "Vector", vout: "Vector", phase: int):
        self.L.chk(self.L.lib.CeedXOperatorApplyPhase(self.h, vin.h, vout.h, C.c_int(phase)))

    def set_timing(self, enable: bool):
        self.L.chk(self.L.lib.CeedXOperatorSetTiming(self.h, int(enable)))

    def get_timing(self):
        ms, n = C.c_double(), C.c_int64()
        self.L.chk(self.L.lib.CeedXOperatorGetTiming(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def launch_info(self) -> dict:
        """CeedXOperatorGetLaunchInfo: how the last apply was launched (segments of the pipelined restriction transpose)."""
        out = (C.c_int * 4)()
        self.L.chk(self.L.lib.CeedXOperatorGetLaunchInfo(self.h, out))
        return dict(segments=out[0], streams=out[1], assemble_launches=out[2], last_segment_elements=out[3])

    def apply_with_halo(self, vin: "Vector", vout: "Vector", halo):
        """CeedXOperatorApplyWithHalo: the (split-phase) apply and the interface sum of its output in one call; `halo` is a
        halo.RcclHalo or a raw CeedXHalo handle."""
        self.L.chk(self.L.lib.CeedXOperatorApplyWithHalo(self.h, vin.h, vout.h, getattr(halo, "h", halo)))

    def destroy(self):
        if self.h:
            self.L.lib.CeedOperatorDestroy(C.byref(self.h))
