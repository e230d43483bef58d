"""hipGraph replay check for the multigrid V-cycle (CeedXGraph*): replay == eager, before and after
unrelated eager work on the same operators.  History: round 1 saw a wrong replay after an eager Jacobian
apply in between and blamed recorded memset nodes; round 2 could not reproduce that in isolation
(tools/microbench/graph_memset_repro.hip) and this check passes with memset nodes too
(CEED_MI355X_GRAPH_MEMSET=1): the cause was the scratch E-vector being re-allocated under recorded
nodes, fixed by parking it (ceed_need_evec)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh
from ceedpetscsolid_amd.solid import SolidProblem
from ceedpetscsolid_amd.solver import NewtonPMG

L = cd.CeedLib(cd.PRODUCT_LIB); c = cd.Ceed(L, "/gpu/hip/mi355x")
mesh = hollow_cylinder_mesh(2, 8, 4, z0=-1.0, z1=1.0)
p = SolidProblem(c, mesh, 4, "hyperSS", nu=0.3, E=1e3, bc_sides=[998, 999])
s = NewtonPMG(p, clamp={998: dict(translate=(0.0, -0.05, 0.1)), 999: dict()}, coarse="chebyshev", coarse_cheb_its=10, coarse_cheb_ratio=50.0)
s.bcv.set_array(s.bc_values(0.5)); s.residual(s.U, s.R); s.setup_preconditioner()
top = s.nlev - 1
rng = np.random.default_rng(0)
free = (p.levels[top].mask == 0)
r0 = rng.standard_normal(p.lsize()) * free
r, z = s.w[top]["b"], s.kz


def err(a, b): return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def trial(name, fn, disturb, out):
    def run(launch):
        r.set_array(r0); L.chk(L.lib.CeedVectorSyncArray(r.h, 1)); out.set_value(-7.0); launch(); return out.to_numpy().copy()
    e0 = run(fn)
    g = c.capture(fn)
    a = err(run(g.launch), e0)
    disturb(); c.synchronize()
    b = err(run(g.launch), e0)
    g.destroy()
    print(f"{name:40s} |ref| {np.abs(e0).max():.2e}  replay-vs-eager {a:.2e}  after eager work {b:.2e}", flush=True)
    return max(a, b)


worst = 0.0
worst = max(worst, trial("A(top)", lambda: s.A(top, r, z), lambda: s.A(top, s.kp, s.kAp), z))
worst = max(worst, trial("chebyshev(top, 3)", lambda: s.chebyshev(top, r, z, 3, True), lambda: s.A(top, s.kp, s.kAp), z))
worst = max(worst, trial("vcycle", lambda: s.vcycle(top, r, z), lambda: s.A(top, s.kp, s.kAp), z))
print("OK" if worst < 1e-12 else "MISMATCH")
sys.exit(0 if worst < 1e-12 else 1)
