"""Step-by-step trace of one tiny gated apply (every step printed and flushed)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def say(*a):
    print(f"[{time.perf_counter():.3f}]", *a, flush=True)
say("import")
import numpy as np
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import box_mesh, hollow_cylinder_mesh
from ceedpetscsolid_amd.solid import SolidProblem
say("lib")
lib = cd.CeedLib(cd.PRODUCT_LIB)
c = cd.Ceed(lib, "/gpu/hip/mi355x")
say("ceed ok; mode", os.environ.get("CEED_MI355X_ASSEMBLE", "(default gated)"))
which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
mesh, deg = (box_mesh(3, 2, 2), 1) if which == "tiny" else (hollow_cylinder_mesh(4, 24, 16), 4)
p = SolidProblem(c, mesh, deg, "hyperFS", nu=0.3, E=1.0, bc_sides=[sorted(mesh.side_sets)[0]], multigrid="none")
say("problem built", mesh.nelem)
n = p.lsize()
X, Y = c.vector(n), c.vector(n)
X.set_array(p.smooth_state(0.1))
say("residual launch")
p.form_residual(X, Y)
say("residual launched; sync")
c.synchronize()
say("residual done")
r = Y.to_numpy()
say("residual norm", float(np.linalg.norm(r)))
for it in range(3):
    X.set_array(np.random.default_rng(it).uniform(-1, 1, n))
    t0 = time.perf_counter()
    p.apply_jacobian(p.fine, X, Y)
    c.synchronize()
    say("jacobian", it, "ms", 1e3 * (time.perf_counter() - t0), "norm", float(np.linalg.norm(Y.to_numpy())))
say("stats", p.levels[p.fine].opJacob.gated_stats())
