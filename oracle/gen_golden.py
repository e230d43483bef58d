#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/qfunctions.npz.

Runs the REFERENCE's own QFunctions (qfunctions/*.h compiled where they lie into
oracle/_ref/libref_qfunctions.so by oracle/Makefile) on seeded inputs and stores
inputs + outputs as plain data.  Needs /root/reference, so it only runs in the
build container; the fixture it writes is committed and travels to the GPU box.

    python oracle/gen_golden.py            # rewrites tests/golden/qfunctions.npz
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
QF = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double)))


def load_table(path, getter):
    lib = C.CDLL(path)
    fn = getattr(lib, getter)
    fn.restype = C.c_void_p
    fn.argtypes = [C.c_char_p]
    return lib, lambda name: QF(fn(name.encode()))


def call_qf(f, ctx, Q, ins, out_sizes):
    ins = [np.ascontiguousarray(a, dtype=np.float64) for a in ins]
    outs = [np.zeros((s, Q)) for s in out_sizes]
    pin = (C.POINTER(C.c_double) * len(ins))(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in ins])
    pout = (C.POINTER(C.c_double) * len(outs))(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in outs])
    ctxa = np.ascontiguousarray(ctx, dtype=np.float64)
    rc = f(ctxa.ctypes.data_as(C.c_void_p), Q, pin, pout)
    assert rc == 0
    return outs


def make_inputs(seed=20261003, Q=96):
    rng = np.random.default_rng(seed)
    # distorted element Jacobians J[d][c] = d x_c / d xi_d (common.h:51, 62-70)
    J = np.zeros((3, 3, Q))
    for i in range(Q):
        h = rng.uniform(0.05, 0.6, size=3)
        A = np.diag(h) + 0.15 * h.mean() * rng.uniform(-1, 1, size=(3, 3))
        J[:, :, i] = A.T
    w = rng.uniform(0.01, 0.3, size=(1, Q))
    # reference-space displacement gradients, |grad u| up to ~0.3 after mapping
    ug = rng.uniform(-1, 1, size=(9, Q)) * 0.02
    dug = rng.uniform(-1, 1, size=(9, Q))
    x = rng.uniform(-0.5, 1.0, size=(3, Q))
    return dict(J=J.reshape(9, Q), w=w, ug=ug, dug=dug, x=x)


def main():
    ref_path = os.path.join(HERE, "_ref", "libref_qfunctions.so")
    if not os.path.exists(ref_path):
        sys.exit("oracle/_ref/libref_qfunctions.so missing: run `make -C oracle` in the build container")
    _, ref = load_table(ref_path, "RefGetQFunction")
    d = make_inputs()
    Q = d["w"].shape[1]
    out = {k: v for k, v in d.items()}
    phys = np.array([0.3, 2.5])  # {nu, E}
    out["phys"] = phys
    force_dir = np.array([0.1, -1.0, 0.25])
    out["force_dir"] = force_dir

    (qdata,) = call_qf(ref("SetupGeo"), phys, Q, [d["J"], d["w"]], [10])
    out["SetupGeo.qdata"] = qdata
    # scale the reference gradient so the physical gradient is O(0.1..0.3)
    ug = d["ug"]
    # points that force both range-shift branches of log1p_series_shifted
    # (hyperFS.h:49-55): physical grad u = -0.2 I  and  +0.2 I
    for i, s in ((0, -0.2), (1, 0.2), (2, -0.28), (3, 0.41)):
        dXdx = qdata[1:, i].reshape(3, 3)
        du = s * np.linalg.inv(dXdx)  # gradu[j][k] = sum_m du[j][m] dXdx[m][k] = s*delta
        # ug[(d*3+c)] = du[c][d]
        ug[:, i] = du.T.reshape(9)
    out["ug"] = ug

    for name in ("LinElasF", "LinElasdF"):
        (dv,) = call_qf(ref(name), phys, Q, [ug if name.endswith("sF") else d["dug"], qdata], [9])
        out[f"{name}.dv"] = dv
    for fam in ("HyperSS", "HyperFS"):
        dv, gradu = call_qf(ref(fam + "F"), phys, Q, [ug, qdata], [9, 9])
        out[f"{fam}F.dv"], out[f"{fam}F.gradu"] = dv, gradu
        (ddv,) = call_qf(ref(fam + "dF"), phys, Q, [d["dug"], qdata, gradu], [9])
        out[f"{fam}dF.dv"] = ddv
    for fam in ("LinElas", "HyperSS", "HyperFS"):   # strain energy density x w detJ (post-processing operator opEnergy)
        (en,) = call_qf(ref(fam + "Energy"), phys, Q, [ug, qdata], [1])
        out[f"{fam}Energy.energy"] = en
        (dg,) = call_qf(ref(fam + "Diagnostic"), phys, Q, [d["x"], ug, qdata], [8])   # u := the sample coordinates
        out[f"{fam}Diagnostic.diagnostic"] = dg
    (f1,) = call_qf(ref("SetupConstantForce"), force_dir, Q, [d["x"], qdata], [3])
    out["SetupConstantForce.force"] = f1
    (f2,) = call_qf(ref("SetupMMSForce"), phys, Q, [d["x"], qdata], [3])
    out["SetupMMSForce.force"] = f2
    (ts,) = call_qf(ref("MMSTrueSoln"), phys, Q, [d["x"]], [3])
    out["MMSTrueSoln.true_soln"] = ts

    dst = os.path.join(ROOT, "tests", "golden", "qfunctions.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
