import os, sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh
from ceedpetscsolid_amd.solid import SolidProblem
from ceedpetscsolid_amd.solver import NewtonPMG
c = cd.Ceed(cd.CeedLib(cd.PRODUCT_LIB), "/gpu/hip/mi355x")
mesh = hollow_cylinder_mesh(10, 110, 90)
p = SolidProblem(c, mesh, 4, "hyperFS", nu=0.3, E=1e3, bc_sides=[998, 999])
s = NewtonPMG(p, clamp={998: dict(translate=(0.0, -0.02, 0.05)), 999: dict()}, coarse="amg", graph="auto", fuse_epilogue="auto")
s.bcv.set_array(s.bc_values(0.1)); s.residual(s.U, s.R)
def T(name, fn, reps=3):
    fn(); c.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    c.synchronize()
    print("%-40s %9.2f ms" % (name, 1e3 * (time.perf_counter() - t0) / reps), flush=True)
T("setup_preconditioner (whole)", s.setup_preconditioner)
T("asm.assemble", s.asm.assemble)
T("amg.setup", s.amg.setup)
for lv in range(s.nlev):
    T(f"get_diag level {lv}", (lambda lv=lv: s.p.get_diag(lv, s.w[lv]["dinv"])) if not (lv == 0) else (lambda: s.asm.diagonal(s.w[0]["dinv"])))
    T(f"lanczos level {lv}", lambda lv=lv: s._lanczos_device(lv, 10))
T("record_preconditioner", lambda: s.record_preconditioner(s.w[s.nlev - 1]["b"], s.kz))
T("residual", lambda: s.residual(s.U, s.R))
T("dot (host sync)", lambda: s.dot(s.R, s.R, True), 20)
T("fine apply", lambda: s.A(s.nlev - 1, s.kz, s.kAp), 20)
T("axpby fine", lambda: s.axpby(s.kz, 1.0, s.kAp, 0.5), 20)
T("precondition", lambda: s.precondition(s.w[s.nlev - 1]["b"], s.kz), 10)
