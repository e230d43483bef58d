"""Bounded diagnostic of the gated assembly: small and mid-size applies, timing per apply, tail statistics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import box_mesh, hollow_cylinder_mesh
from ceedpetscsolid_amd.solid import SolidProblem

lib = cd.CeedLib(cd.PRODUCT_LIB)
def ceed_env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    os.environ.update(kw)
    try:
        return cd.Ceed(lib, "/gpu/hip/mi355x")
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)

gated, serial = ceed_env(CEED_MI355X_ASSEMBLE="gated"), ceed_env(CEED_MI355X_ASSEMBLE="serial")
for name, mesh, deg, prob in [("box3x2x2 p1", box_mesh(3, 2, 2), 1, "hyperFS"), ("cyl 2x8x3 p4", hollow_cylinder_mesh(2, 8, 3), 4, "hyperFS"),
                              ("cyl 4x24x16 p4", hollow_cylinder_mesh(4, 24, 16), 4, "hyperFS"), ("cyl 10x40x20 p4", hollow_cylinder_mesh(10, 40, 20), 4, "hyperFS"),
                              ("cyl 10x110x30 p4", hollow_cylinder_mesh(10, 110, 30), 4, "hyperFS")]:
    res = []
    for c in (gated, serial):
        p = SolidProblem(c, mesh, deg, prob, nu=0.3, E=1.0, bc_sides=[sorted(mesh.side_sets)[0]], multigrid="none")
        n = p.lsize()
        X, Y = c.vector(n), c.vector(n)
        X.set_array(p.smooth_state(0.1)); p.form_residual(X, Y)
        rng = np.random.default_rng(3)
        outs, times = [], []
        for it in range(6):
            X.set_array(rng.uniform(-1, 1, n))
            c.synchronize(); t0 = time.perf_counter()
            p.apply_jacobian(p.fine, X, Y)
            c.synchronize(); times.append(1e3 * (time.perf_counter() - t0))
            outs.append(Y.to_numpy())
        res.append((outs, times, p.levels[p.fine].opJacob.gated_stats()))
    same = all(np.array_equal(a, b) for a, b in zip(res[0][0], res[1][0]))
    print(f"{name}: nelem {mesh.nelem}  bitwise equal {same}  gated ms {[round(t, 3) for t in res[0][1]]}  serial ms {[round(t, 3) for t in res[1][1]]}  stats {res[0][2]}", flush=True)
