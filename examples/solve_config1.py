#!/usr/bin/env python3
"""BASELINE config 1 on the MI355X path: linElas, unit box 4x4x4, degree 2, manufactured solution.

The reference's own acceptance test (elasticity.c:36 //TESTARGS ... -forcing mms; :790-810): solve with the
SetupMMSForce body force and BCMMS boundary values, compare with MMSTrueSoln, fail if the relative L2 error
exceeds 0.05.  Also prints the strain energy (ComputeStrainEnergy, matops.c:247-296).
    python examples/solve_config1.py [--n 4] [--degree 2] [--oracle]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import box_mesh
from ceedpetscsolid_amd.solid import SolidProblem
from ceedpetscsolid_amd.solver import NewtonPMG
from ceedpetscsolid_amd.postprocess import StrainEnergy

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4)
ap.add_argument("--degree", type=int, default=2)
ap.add_argument("--E", type=float, default=1e6)
ap.add_argument("--nu", type=float, default=0.3)
ap.add_argument("--oracle", action="store_true", help="TESTS ONLY: the same solve on the CPU oracle")
args = ap.parse_args()
if args.oracle:
    lib = cd.CeedLib(os.path.join(ROOT, "oracle", "liboracle_ceed.so")); c = cd.Ceed(lib, "/cpu/self/oracle")
else:
    lib = cd.CeedLib(cd.PRODUCT_LIB); c = cd.Ceed(lib, "/gpu/hip/mi355x")
p = SolidProblem(c, box_mesh(args.n, args.n, args.n), args.degree, "linElas", nu=args.nu, E=args.E, bc_all_boundary=True)
lv = p.levels[p.fine]
n = p.lsize()


def coord_op(qfname, src, out_basis, out_emode, with_qdata):
    qf = c.qfunction(qfname, source=f"qfunctions/{src}:{qfname}")
    qf.add_input("x", 3, cd.EVAL_INTERP)
    if with_qdata:
        qf.add_input("qdata", 10, cd.EVAL_NONE)
    qf.add_output("out", 3, out_emode)
    qf.set_context(p.phys)
    op = c.operator(qf)
    op.set_field("x", p.Erestrictx, p.basisx if with_qdata else c.basis_lagrange(3, 3, 2, lv.degree + 1, cd.GAUSS_LOBATTO), "active")
    if with_qdata:
        op.set_field("qdata", p.Erestrictqdi, None, p.qdata)
    op.set_field("out", lv.Erestrictu, out_basis, "active")
    v = c.vector(n)
    op.apply(p.xcoord, v)
    return v.to_numpy()


force = coord_op("SetupMMSForce", "manufacturedForce.h", lv.basisu, cd.EVAL_INTERP, True)       # setuplibceed.c:555-583
true = coord_op("MMSTrueSoln", "manufacturedTrue.h", None, cd.EVAL_NONE, False)                 # :608-623
mult = c.vector(n); lv.Erestrictu.multiplicity(mult)
true /= mult.to_numpy()                                                                        # :626-636
s = NewtonPMG(p, mms=True, forcing=force)
st = s.solve(1)
u = s.U.to_numpy() + s.bc_values(1.0)
err = float(np.linalg.norm(u - true) / np.linalg.norm(true))
energy = StrainEnergy(p, "linElas").compute(c.vector(n).set_array(u))
print(json.dumps({"resource": c.resource, "elements": p.mesh.nelem, "degree": args.degree, "dofs": n, "converged": st.converged,
                  "snes_its": st.newton_its, "ksp_its": st.ksp_its, "l2_error": err, "gate": "l2_error <= 0.05 (elasticity.c:807)",
                  "passed": bool(err <= 0.05), "strain_energy": energy}))
sys.exit(0 if err <= 0.05 else 1)
