#!/usr/bin/env python3
"""config 3 at the README's load (-bc_clamp_998_translate 0,-0.5,1, README.rst:63) -- which side fails?

For hyperSS and hyperFS on cylinder8_5580e_4ss_us at degree 4 and 2, the FIRST load increment is run for a ladder of increment
sizes (the README translation divided by 10, 20, 40, 100, 200, 400 and the tenth-load step of the pinned test, /100) with three
line searches: this build's critical-point search ("cp"), SNESLINESEARCHCP as PETSc runs it by default ("cp-petsc": one secant
step, clamped, never rejecting -- what elasticity.c:596-601 selects) and full Newton steps ("full").  After the first Newton step
of each run the stored state is read back: the smallest det F = det(I + grad u) and the smallest 1 + tr(grad u) over all quadrature
points -- an inverted element (det F <= 0) or tr eps <= -1 (the argument of hyperSS's log1p_series, hyperSS.h:43-55, at or beyond
its pole) pins a failure on the physics rather than on the solver.
    python tools/r4_config3_load.py > gpurun_out/r4/config3_load.txt
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # (tools/archive/)
sys.path.insert(0, ROOT)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import load_mesh_npz
from ceedpetscsolid_amd.solid import SolidProblem
from ceedpetscsolid_amd.solver import NewtonPMG

mesh = load_mesh_npz(os.path.join(ROOT, "tests", "golden", "mesh_cylinder8_5580e_4ss_us.npz"))
if os.environ.get("STUDY_ON_ORACLE"):   # (script check on the CPU only)
    ceed = cd.Ceed(cd.CeedLib(os.path.join(ROOT, "oracle", "liboracle_ceed.so")), "/cpu/self/oracle")
else:
    ceed = cd.Ceed(cd.CeedLib(cd.PRODUCT_LIB), "/gpu/hip/mi355x")
FULL = np.array([0.0, -0.5, 1.0])
degrees = [int(d) for d in os.environ.get("DEGREES", "4,2").split(",")]
divisors = [int(d) for d in os.environ.get("DIVISORS", "10,20,40,100,200,400").split(",")]
searches = os.environ.get("SEARCHES", "cp,cp-petsc,full").split(",")
for problem in os.environ.get("PROBLEMS", "hyperSS,hyperFS").split(","):
    for degree in degrees:
        prob = SolidProblem(ceed, mesh, degree, problem, nu=0.3, E=1e3, bc_sides=[998, 999])
        Q3 = (degree + 1) ** 3
        for ls in searches:
            for div in divisors:
                solver = NewtonPMG(prob, clamp={998: dict(translate=tuple(FULL)), 999: dict()}, coarse="amg", graph=False, line_search=ls, snes_maxit=30)
                t0 = time.perf_counter()
                err = None
                try:
                    st = solver.solve(div, stop_after=1)
                except cd.CeedError as e:
                    st, err = solver.stats, str(e)[:90]
                # state after the last residual evaluation: grad u at every point, [elem][9][Q^3], component (3 c + k)
                g = prob.gradu.to_numpy().reshape(mesh.nelem, 3, 3, Q3)
                F = g + np.eye(3)[None, :, :, None]
                detF = np.linalg.det(np.moveaxis(F, 3, 1))
                tr = g[:, 0, 0] + g[:, 1, 1] + g[:, 2, 2]
                hist = st.history
                print(json.dumps({"problem": problem, "degree": degree, "line_search": ls, "increment_translation": (FULL / div).tolist(),
                                  "divisor": div, "converged": bool(st.converged and err is None and len(hist) > 0 and np.isfinite(hist[-1][4])), "newton_its": st.newton_its, "ksp_its": st.ksp_its,
                                  "lambdas": [round(h[3], 4) for h in hist[:6]], "rnorms": [float("%.3e" % h[4]) for h in hist[:6]],
                                  "final_rnorm": (float("%.3e" % hist[-1][4]) if hist else None),
                                  "min_detF": float(np.nanmin(detF)), "min_1_plus_tr": float(np.nanmin(1.0 + tr)), "nan_points": int(np.isnan(detF).sum()),
                                  "error": err, "seconds": round(time.perf_counter() - t0, 2)}), flush=True)
        prob.destroy()
