#!/usr/bin/env python3
"""Device time of the Jacobian apply on EVERY level of a p-multigrid ladder (the coarse levels run on the
FINE quadrature, as the reference does -- setuplibceed.c:757-784 -- so a p = 1 apply costs almost what a p = 4 apply does).
Per level: P, Q, dofs, the reference-formulation bytes (q-point data at the fine Q, offsets at P, x and y once),
microseconds per apply (hipEvents around the operator's launches, mean of --steps), the effective rate and its share of 8 TB/s.

    python3 tools/level_apply_times.py --cylinder 10,110,90 --degree 4 --problem hyperFS
    python3 tools/level_apply_times.py --mesh tests/golden/mesh_cylinder8_5580e_4ss_us.npz --degree 4 --problem hyperSS
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh, load_mesh_npz, box_mesh
from ceedpetscsolid_amd.solid import SolidProblem

ap = argparse.ArgumentParser()
ap.add_argument("--mesh"); ap.add_argument("--cylinder"); ap.add_argument("--box")
ap.add_argument("--degree", type=int, default=4); ap.add_argument("--problem", default="hyperFS")
ap.add_argument("--multigrid", default="logarithmic"); ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
mesh = (hollow_cylinder_mesh(*map(int, a.cylinder.split(","))) if a.cylinder else box_mesh(*map(int, a.box.split(","))) if a.box else load_mesh_npz(a.mesh))
ceed = cd.Ceed(cd.CeedLib(cd.PRODUCT_LIB), "/gpu/hip/mi355x")
bc = [s for s in (998, 999, 1, 2) if s in mesh.side_sets and len(mesh.side_sets[s])][:2]
p = SolidProblem(ceed, mesh, a.degree, a.problem, nu=0.3, E=1.0, bc_sides=bc, multigrid=a.multigrid)
n = p.lsize()
X, Y = ceed.vector(n), ceed.vector(n)
X.set_array(p.smooth_state(0.05)); p.form_residual(X, Y)
Q = a.degree + 1
state = a.problem != "linElas"
rows = []
for lv, deg in enumerate(p.degrees):
    nl, P = p.lsize(lv), deg + 1
    x = ceed.vector(nl).set_array(np.random.default_rng(lv).uniform(-1, 1, nl) * (p.levels[lv].mask == 0)); y = ceed.vector(nl)
    op = p.levels[lv].opJacob
    for _ in range(20):
        p.apply_jacobian(lv, x, y)
    ceed.synchronize()
    op.set_timing(True)
    for _ in range(a.steps):
        p.apply_jacobian(lv, x, y)
    ceed.synchronize()
    ms, launches = op.get_timing(); op.set_timing(False)
    us = 1e3 * ms / a.steps
    abytes = mesh.nelem * (8 * (19 if state else 10) * Q ** 3 + 4 * P ** 3) + 16 * nl
    rows.append({"level": lv, "degree": deg, "P": P, "Q": Q, "dofs": int(p.n_free(lv)), "us_per_apply": us, "algorithmic_bytes": abytes,
                 "GBs": abytes / us / 1e3, "frac_of_8TBs": abytes / us / 1e3 / 8000.0, "MDoFs": p.n_free(lv) / us, "kernel": op.kernel_name})
print(json.dumps({"mesh": getattr(mesh, "name", ""), "elements": mesh.nelem, "problem": a.problem, "levels": rows}))
for r in rows:
    print("# level %d  P=%d Q=%d  %9d dofs  %8.1f us  %7.0f GB/s (%.2f of 8 TB/s)  %7.0f MDoF/s  %s" % (r["level"], r["P"], r["Q"], r["dofs"], r["us_per_apply"], r["GBs"], r["frac_of_8TBs"], r["MDoFs"], r["kernel"]), file=sys.stderr)
