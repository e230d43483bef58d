"""A/B of the two fused kernels (CEED_MI355X_FUSED=rows|pencil) on identical data."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import box_mesh, hollow_cylinder_mesh
from ceedpetscsolid_amd.solid import SolidProblem

L = cd.CeedLib(cd.PRODUCT_LIB)
os.environ["CEED_MI355X_FUSED"] = "rows"; crow = cd.Ceed(L, "/gpu/hip/mi355x")
os.environ["CEED_MI355X_FUSED"] = "pencil"; cpen = cd.Ceed(L, "/gpu/hip/mi355x")
cases = [("box 2x2x2 p1", box_mesh(2, 2, 2), 1), ("box 3x3x3 p1", box_mesh(3, 3, 3), 1), ("box 2x2x2 p2", box_mesh(2, 2, 2), 2),
         ("box 3x2x2 p3", box_mesh(3, 2, 2), 3), ("box 2x2x2 p4", box_mesh(2, 2, 2), 4), ("box 3x3x3 p4", box_mesh(3, 3, 3), 4),
         ("cyl p4", hollow_cylinder_mesh(2, 8, 3), 4), ("box 2x2x1 p5", box_mesh(2, 2, 1), 5), ("box 2x1x1 p6", box_mesh(2, 1, 1), 6)]
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
for name, mesh, deg in cases:
    for problem in ("linElas", "hyperFS"):
        out = {}
        for tag, c in (("rows", crow), ("pencil", cpen)):
            p = SolidProblem(c, mesh, deg, problem, nu=0.3, E=1e3, bc_sides=[])
            n = p.lsize()
            X, Y = c.vector(n), c.vector(n)
            X.set_array(p.smooth_state(0.05)); p.form_residual(X, Y)
            res = Y.to_numpy().copy()
            lv_out = []
            for lv in range(len(p.levels)):
                nl = p.lsize(lv)
                x = np.random.default_rng(lv).uniform(-1, 1, nl)
                xa, ya, yb = c.vector(nl).set_array(x), c.vector(nl), c.vector(nl)
                p.apply_jacobian(lv, xa, ya); p.apply_jacobian(lv, xa, yb)
                lv_out.append((ya.to_numpy().copy(), bool(np.array_equal(ya.to_numpy(), yb.to_numpy())), p.levels[lv].opJacob.kernel_name))
            out[tag] = (res, lv_out)
        msg = f"{name:14s} {problem:8s} nelem {mesh.nelem:4d} residual {rel(out['pencil'][0], out['rows'][0]):.1e}"
        for lv, (po, ro) in enumerate(zip(out["pencil"][1], out["rows"][1])):
            msg += f" | L{lv} {po[2][11:18]} diff {rel(po[0], ro[0]):.1e} repro {po[1]}"
        print(msg, flush=True)
