"""Pin the oracle's restated pointwise physics against golden vectors produced by the
REFERENCE's own qfunctions/*.h (oracle/gen_golden.py -> tests/golden/qfunctions.npz).
Also, when the reference build oracle/_ref is present, re-run it live against the fixture."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ORACLE_LIB, REF_QF_LIB, rel_err, _ensure_oracle

QF = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double)))
TOL = 1e-13  # same arithmetic in another operation order: a few ulp


def table(path, getter):
    lib = C.CDLL(path)
    fn = getattr(lib, getter)
    fn.restype, fn.argtypes = C.c_void_p, [C.c_char_p]
    def get(name):
        p = fn(name.encode())
        assert p, f"{getter}({name}) is NULL"
        return QF(p)
    get._lib = lib
    return get


def call(f, ctx, Q, ins, out_sizes):
    ins = [np.ascontiguousarray(a, dtype=np.float64) for a in ins]
    outs = [np.zeros((s, Q)) for s in out_sizes]
    dp = C.POINTER(C.c_double)
    pin = (dp * len(ins))(*[a.ctypes.data_as(dp) for a in ins])
    pout = (dp * len(outs))(*[a.ctypes.data_as(dp) for a in outs])
    ctxa = np.ascontiguousarray(ctx, dtype=np.float64)
    assert f(ctxa.ctypes.data_as(C.c_void_p), Q, pin, pout) == 0
    return outs


CASES = [
    ("SetupGeo", lambda g: [g["J"], g["w"]], [10], ["SetupGeo.qdata"]),
    ("LinElasF", lambda g: [g["ug"], g["SetupGeo.qdata"]], [9], ["LinElasF.dv"]),
    ("LinElasdF", lambda g: [g["dug"], g["SetupGeo.qdata"]], [9], ["LinElasdF.dv"]),
    ("HyperSSF", lambda g: [g["ug"], g["SetupGeo.qdata"]], [9, 9], ["HyperSSF.dv", "HyperSSF.gradu"]),
    ("HyperSSdF", lambda g: [g["dug"], g["SetupGeo.qdata"], g["HyperSSF.gradu"]], [9], ["HyperSSdF.dv"]),
    ("HyperFSF", lambda g: [g["ug"], g["SetupGeo.qdata"]], [9, 9], ["HyperFSF.dv", "HyperFSF.gradu"]),
    ("HyperFSdF", lambda g: [g["dug"], g["SetupGeo.qdata"], g["HyperFSF.gradu"]], [9], ["HyperFSdF.dv"]),
    ("SetupMMSForce", lambda g: [g["x"], g["SetupGeo.qdata"]], [3], ["SetupMMSForce.force"]),
    ("MMSTrueSoln", lambda g: [g["x"]], [3], ["MMSTrueSoln.true_soln"]),
    ("LinElasEnergy", lambda g: [g["ug"], g["SetupGeo.qdata"]], [1], ["LinElasEnergy.energy"]),
    ("HyperSSEnergy", lambda g: [g["ug"], g["SetupGeo.qdata"]], [1], ["HyperSSEnergy.energy"]),
    ("HyperFSEnergy", lambda g: [g["ug"], g["SetupGeo.qdata"]], [1], ["HyperFSEnergy.energy"]),
    ("LinElasDiagnostic", lambda g: [g["x"], g["ug"], g["SetupGeo.qdata"]], [8], ["LinElasDiagnostic.diagnostic"]),
    ("HyperSSDiagnostic", lambda g: [g["x"], g["ug"], g["SetupGeo.qdata"]], [8], ["HyperSSDiagnostic.diagnostic"]),
    ("HyperFSDiagnostic", lambda g: [g["x"], g["ug"], g["SetupGeo.qdata"]], [8], ["HyperFSDiagnostic.diagnostic"]),
]


@pytest.mark.parametrize("name,ins,sizes,keys", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_reference_golden(golden_qf, name, ins, sizes, keys):
    get = table(_ensure_oracle(), "OracleGetQFunction")
    g = golden_qf
    outs = call(get(name), g["phys"], g["w"].shape[1], ins(g), sizes)
    for o, k in zip(outs, keys):
        assert rel_err(o, g[k]) < TOL, (name, k, rel_err(o, g[k]))


def test_constant_force(golden_qf):
    g = golden_qf
    get = table(_ensure_oracle(), "OracleGetQFunction")
    (f,) = call(get("SetupConstantForce"), g["force_dir"], g["w"].shape[1], [g["x"], g["SetupGeo.qdata"]], [3])
    assert rel_err(f, g["SetupConstantForce.force"]) < TOL


def test_golden_exercises_both_log1p_range_shifts(golden_qf):
    """hyperFS.h:49-55: the fixture must contain points on both sides of the shifted range."""
    gu = golden_qf["HyperFSF.gradu"]
    J = np.array([np.linalg.det(np.eye(3) + gu[:, i].reshape(3, 3)) for i in range(gu.shape[1])])
    detCm1 = J * J - 1
    assert (detCm1 < np.sqrt(2) / 2 - 1).any() and (detCm1 > np.sqrt(2) - 1).any()


@pytest.mark.skipif(not os.path.exists(REF_QF_LIB), reason="oracle/_ref not built (reference tree absent)")
@pytest.mark.parametrize("name,ins,sizes,keys", CASES, ids=[c[0] for c in CASES])
def test_live_reference_reproduces_fixture(golden_qf, name, ins, sizes, keys):
    get = table(REF_QF_LIB, "RefGetQFunction")
    g = golden_qf
    outs = call(get(name), g["phys"], g["w"].shape[1], ins(g), sizes)
    for o, k in zip(outs, keys):
        assert np.array_equal(o, g[k]), (name, k)


def test_log1p_series_vs_libm():
    """the series deviates from libm log1p by up to ~3e-8 on its documented range;
    the oracle must follow the series (so must the GPU), not libm."""
    get = table(_ensure_oracle(), "OracleGetQFunction")
    # hyperSS residual of a pure volumetric strain isolates lambda*log1p_series(tr e)
    Q = 5
    tr = np.array([-0.25, -0.05, 1e-9, 0.1, 0.4])
    ug = np.zeros((9, Q)); qd = np.zeros((10, Q))
    qd[0] = 1.0; qd[1] = qd[5] = qd[9] = 1.0
    for d in range(3):
        ug[d * 3 + d] = tr / 3
    nu, E = 0.3, 1.0
    dv, _ = call(get("HyperSSF"), [nu, E], Q, [ug, qd], [9, 9])
    TwoMu = E / (1 + nu); lam = (3 * E / (3 * (1 - 2 * nu)) - TwoMu) / 3
    series = (dv[0] - TwoMu * tr / 3) / lam
    assert np.abs(series - np.log1p(tr)).max() < 1e-4
    assert np.abs(series - np.log1p(tr)).max() > 1e-12  # it IS the series, not libm
