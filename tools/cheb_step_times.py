#!/usr/bin/env python3
"""One Chebyshev smoothing step (Jacobian apply + update of r, d, x) on every level: the apply fused with its consumer
(CeedXOperatorApplyChebyshev) against CeedOperatorApply + CeedXVectorChebyshevUpdate; wall time per step over --steps steps
between two stream synchronisations.   python3 tools/cheb_step_times.py --cylinder 10,110,90 --degree 4 --problem hyperFS"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh, box_mesh
from ceedpetscsolid_amd.solid import SolidProblem

ap = argparse.ArgumentParser()
ap.add_argument("--cylinder"); ap.add_argument("--box"); ap.add_argument("--degree", type=int, default=4)
ap.add_argument("--problem", default="hyperFS"); ap.add_argument("--steps", type=int, default=40); ap.add_argument("--out")
a = ap.parse_args()
mesh = hollow_cylinder_mesh(*map(int, a.cylinder.split(","))) if a.cylinder else box_mesh(*map(int, a.box.split(",")))
ceed = cd.Ceed(cd.CeedLib(cd.PRODUCT_LIB), "/gpu/hip/mi355x")
L = ceed.L
bc = [s for s in (998, 999, 1, 2) if s in mesh.side_sets and len(mesh.side_sets[s])][:2]
p = SolidProblem(ceed, mesh, a.degree, a.problem, nu=0.3, E=1.0, bc_sides=bc)
n = p.lsize()
X, Y = ceed.vector(n).set_array(p.smooth_state(0.05)), ceed.vector(n)
p.form_residual(X, Y)
rows = []
for lv in range(len(p.levels)):
    nl = p.lsize(lv)
    rng = np.random.default_rng(lv)
    free = (p.levels[lv].mask == 0).astype(np.float64)
    v = {k: ceed.vector(nl).set_array(rng.uniform(-1, 1, nl) * free * (1e-3 if k == "dinv" else 1.0)) for k in ("x", "d", "r", "dinv")}
    t = ceed.vector(nl)
    op = p.levels[lv].opJacob
    c1, c2 = C.c_double(0.3), C.c_double(0.2)

    def fused():
        L.chk(L.lib.CeedXOperatorApplyChebyshev(op.h, v["d"].h, t.h, v["x"].h, v["d"].h, v["r"].h, None, v["dinv"].h, c1, c2, 0))

    def two():
        op.apply(v["d"], t)
        L.chk(L.lib.CeedXVectorChebyshevUpdate(v["x"].h, v["d"].h, v["r"].h, t.h, v["dinv"].h, c1, c2, 0))

    def apply_only():
        op.apply(v["d"], t)
    res = {}
    for name, fn in (("apply_only", apply_only), ("two_pass", two), ("fused", fused), ("two_pass_again", two)):
        for _ in range(5):
            fn()
        ceed.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        ceed.synchronize()
        res[name] = 1e6 * (time.perf_counter() - t0) / a.steps
    rows.append({"level": lv, "P": p.degrees[lv] + 1, "dofs": nl, **res, "launch_info": op.launch_info()})
    print("# level %d P=%d %9d dofs: apply %7.1f us | apply + update %7.1f | fused %7.1f | apply + update again %7.1f" % (lv, p.degrees[lv] + 1, nl, res["apply_only"], res["two_pass"], res["fused"], res["two_pass_again"]), file=sys.stderr)
print(json.dumps({"elements": mesh.nelem, "problem": a.problem, "levels": rows}))
if a.out:
    json.dump({"elements": mesh.nelem, "problem": a.problem, "levels": rows}, open(a.out, "w"), indent=1)
