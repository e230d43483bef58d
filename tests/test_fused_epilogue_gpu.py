"""The Jacobian apply fused with its consumer (CeedXOperatorApplyChebyshev / CeedXOperatorApplyResidual, round 5): the Chebyshev
step of the smoother (elasticity.c:539-552) and the residual of the V-cycle (:588-590) formed in the epilogue of the apply's
restriction transpose instead of a pass of their own.  The fused forms must give the SAME BITS as CeedOperatorApply followed by
CeedXVectorChebyshevUpdate / ChebyshevStart / WAXPBY -- serial and pipelined assembly, every level of a ladder (P < Q kernels,
P = 2 without element-interior nodes), in-place input (the recurrence applies the operator to its own direction d), recorded
into a hipGraph -- and match the oracle's restatement at the parity bar."""
import ctypes as C
import os

import numpy as np
import pytest

from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import box_mesh, hollow_cylinder_mesh
from ceedpetscsolid_amd.solid import SolidProblem
from conftest import rel_err

pytestmark = pytest.mark.gpu


def _ceed_with_env(product_lib, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return cd.Ceed(product_lib, "/gpu/hip/mi355x")      # the switches are read at CeedInit
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _vectors(c, n, arrs):
    return {k: c.vector(n).set_array(v) for k, v in arrs.items()}


def _cheb_pair(p, lv, arrs, first, in_place=True):
    """(unfused, fused) results {x, d, r} of one Chebyshev step on level lv from the same host arrays.  first: False = the recurrence
    step (r -= A d), True = a first step (r = b - A x, c2 = 0), "recomputed" = the solver's step (r = b - A x with c2 != 0, r not stored)."""
    c, L, op = p.ceed, p.ceed.L, p.levels[lv].opJacob
    n = p.lsize(lv)
    c1, c2 = C.c_double(0.37), C.c_double(0.0 if first is True else 0.21)
    out = []
    for fused in (False, True):
        v = _vectors(c, n, arrs)
        t = c.vector(n).set_value(-3.0)
        src = v["x"] if first else v["d"]
        if not in_place:
            src = c.vector(n).set_array(arrs["x"] if first else arrs["d"])
        r = None if first == "recomputed" else v["r"].h
        if fused:
            L.chk(L.lib.CeedXOperatorApplyChebyshev(op.h, src.h, t.h, v["x"].h, v["d"].h, r, v["b"].h if first else None, v["dinv"].h, c1, c2, 0))
        else:
            op.apply(src, t)
            if first == "recomputed":
                L.chk(L.lib.CeedXVectorChebyshevStep(v["x"].h, v["d"].h, None, v["b"].h, t.h, v["dinv"].h, c1, c2, 0))
            elif first:
                L.chk(L.lib.CeedXVectorChebyshevStart(v["x"].h, v["d"].h, v["r"].h, v["b"].h, t.h, v["dinv"].h, c1, 0))
            else:
                L.chk(L.lib.CeedXVectorChebyshevUpdate(v["x"].h, v["d"].h, v["r"].h, t.h, v["dinv"].h, c1, c2, 0))
        out.append({k: v[k].to_numpy() for k in ("x", "d", "r")})
    return out


def _arrays(p, lv, seed):
    rng = np.random.default_rng(seed)
    n = p.lsize(lv)
    free = (p.levels[lv].mask == 0).astype(np.float64)
    return {"x": rng.uniform(-1, 1, n) * free, "d": rng.uniform(-1, 1, n) * free, "r": rng.uniform(-1, 1, n) * free,
            "b": rng.uniform(-1, 1, n) * free, "dinv": rng.uniform(0.5, 2.0, n) * free}


CASES = [
    ("cyl p4 fs", lambda: hollow_cylinder_mesh(3, 12, 6), 4, "hyperFS", [998, 999]),
    ("box p2 ss", lambda: box_mesh(5, 4, 3), 2, "hyperSS", [1]),
    ("box p3 le", lambda: box_mesh(3, 3, 2), 3, "linElas", [1, 2]),
    ("box p6 fs", lambda: box_mesh(2, 2, 3), 6, "hyperFS", [1]),
]


@pytest.mark.parametrize("form", ["serial", "pipelined"])
@pytest.mark.parametrize("name,mk,degree,problem,bc", CASES, ids=[c[0] for c in CASES])
def test_fused_chebyshev_step_and_residual_equal_the_two_steps_bitwise(product_lib, oracle, form, name, mk, degree, problem, bc):
    mesh = mk()
    env = {"CEED_MI355X_ASSEMBLE": "serial"} if form == "serial" else {"CEED_MI355X_PIPE_MIN_ROUNDS": "0", "CEED_MI355X_PIPE_SEGMENTS": "3", "CEED_MI355X_EPI_PIPELINED": "1"}
    gpu = _ceed_with_env(product_lib, env)
    p = SolidProblem(gpu, mesh, degree, problem, nu=0.3, E=1.0, bc_sides=bc)
    po = SolidProblem(oracle, mesh, degree, problem, nu=0.3, E=1.0, bc_sides=bc)
    n = p.lsize()
    u = p.smooth_state(0.08)
    for q in (p, po):
        X, Y = q.ceed.vector(n).set_array(u), q.ceed.vector(n)
        q.form_residual(X, Y)                                    # the stored state of the tangent
    for lv in range(len(p.levels)):
        arrs = _arrays(p, lv, 100 + lv)
        for first in (False, True, "recomputed"):
            for in_place in (True, False):
                a, b = _cheb_pair(p, lv, arrs, first, in_place)
                for k in ("x", "d", "r"):
                    assert np.array_equal(a[k], b[k]), (lv, first, in_place, k, np.abs(a[k] - b[k]).max())
            oa, ob = _cheb_pair(po, lv, arrs, first)             # the oracle's restatement of the fused entry = its two steps
            for k in ("x", "d", "r"):
                assert np.array_equal(oa[k], ob[k]) and rel_err(b[k], ob[k]) < 1e-10, (lv, first, k)
        info = p.levels[lv].opJacob.launch_info()
        assert info["segments"] == (1 if form == "serial" else 3), info
        # residual w = b - A x
        c, L, op = p.ceed, p.ceed.L, p.levels[lv].opJacob
        nl = p.lsize(lv)
        X, B, T, W1, W2 = (c.vector(nl).set_array(arrs["x"]), c.vector(nl).set_array(arrs["b"]), c.vector(nl), c.vector(nl), c.vector(nl))
        op.apply(X, T)
        L.chk(L.lib.CeedXVectorWAXPBY(W1.h, C.c_double(1.0), B.h, C.c_double(-1.0), T.h))
        T.set_value(9.0)
        L.chk(L.lib.CeedXOperatorApplyResidual(op.h, X.h, T.h, B.h, W2.h))
        assert np.array_equal(W1.to_numpy(), W2.to_numpy()), lv
    p.destroy(); po.destroy()


def test_fused_chebyshev_sweep_recorded_and_replayed(product_lib):
    """Three fused steps recorded into a hipGraph (pipelined form: fork and join inside the capture) and replayed on new data
    against the eager two-step form."""
    gpu = _ceed_with_env(product_lib, {"CEED_MI355X_PIPE_MIN_ROUNDS": "0", "CEED_MI355X_PIPE_SEGMENTS": "2", "CEED_MI355X_EPI_PIPELINED": "1"})
    mesh = hollow_cylinder_mesh(4, 16, 8)
    p = SolidProblem(gpu, mesh, 4, "hyperFS", nu=0.3, E=1.0, bc_sides=[998], multigrid="none")
    n = p.lsize()
    X, Y = gpu.vector(n).set_array(p.smooth_state(0.05)), gpu.vector(n)
    p.form_residual(X, Y)
    L, op = gpu.L, p.levels[p.fine].opJacob
    arrs = _arrays(p, p.fine, 5)
    v = _vectors(gpu, n, arrs)
    t = gpu.vector(n)
    coef = [(0.4, 0.0), (0.33, 0.2), (0.31, 0.25)]

    def sweep_fused():
        L.chk(L.lib.CeedXOperatorApplyChebyshev(op.h, v["x"].h, t.h, v["x"].h, v["d"].h, v["r"].h, v["b"].h, v["dinv"].h, C.c_double(coef[0][0]), C.c_double(0.0), 0))
        for c1, c2 in coef[1:]:
            L.chk(L.lib.CeedXOperatorApplyChebyshev(op.h, v["d"].h, t.h, v["x"].h, v["d"].h, v["r"].h, None, v["dinv"].h, C.c_double(c1), C.c_double(c2), 0))
    sweep_fused()                                                # warm: maps, flags, scratch
    g = gpu.capture(sweep_fused)
    rng = np.random.default_rng(9)
    for trial in range(3):
        a2 = {k: (rng.uniform(-1, 1, n) * (arrs["dinv"] != 0) if k != "dinv" else arrs["dinv"]) for k in arrs}
        w = _vectors(gpu, n, a2)
        tt = gpu.vector(n)
        op.apply(w["x"], tt)
        L.chk(L.lib.CeedXVectorChebyshevStart(w["x"].h, w["d"].h, w["r"].h, w["b"].h, tt.h, w["dinv"].h, C.c_double(coef[0][0]), 0))
        for c1, c2 in coef[1:]:
            op.apply(w["d"], tt)
            L.chk(L.lib.CeedXVectorChebyshevUpdate(w["x"].h, w["d"].h, w["r"].h, tt.h, w["dinv"].h, C.c_double(c1), C.c_double(c2), 0))
        for k in a2:
            v[k].set_array(a2[k]); v[k].device_pointer()
        g.launch()
        for k in ("x", "d", "r"):
            assert np.array_equal(v[k].to_numpy(), w[k].to_numpy()), (trial, k)
    g.destroy(); p.destroy()
