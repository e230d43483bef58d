#!/usr/bin/env python3
"""Device time of Prolong_Ceed / Restrict_Ceed (src/matops.c:115-203) on every level pair of a p-multigrid ladder.
Per pair and direction: microseconds per apply (hipEvents around the operator's launches: the transfer kernel and, for a
restriction, the coarse-side k_assemble), the algorithmic bytes of DESIGN.md 4 (16 B per coarse and per fine dof -- value and
multVec on the fine side, x and y on the coarse -- plus both offset arrays), the effective rate and its share of 8 TB/s.

    python3 tools/transfer_times.py --cylinder 10,110,90 --degree 4
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
ap.add_argument("--degree", type=int, default=4); ap.add_argument("--multigrid", default="logarithmic")
ap.add_argument("--steps", type=int, default=50); ap.add_argument("--out")
a = ap.parse_args()
mesh = (hollow_cylinder_mesh(*map(int, a.cylinder.split(","))) if a.cylinder else box_mesh(*map(int, a.box.split(","))) if a.box else load_mesh_npz(a.mesh))
ceed = cd.Ceed(cd.CeedLib(cd.PRODUCT_LIB), "/gpu/hip/mi355x")
bc = [s for s in (998, 999, 1, 2) if s in mesh.side_sets and len(mesh.side_sets[s])][:2]
p = SolidProblem(ceed, mesh, a.degree, "linElas", nu=0.3, E=1.0, bc_sides=bc, multigrid=a.multigrid)
rows = []
for lv in range(1, len(p.levels)):
    nf, nc = p.lsize(lv), p.lsize(lv - 1)
    Pf, Pc = p.degrees[lv] + 1, p.degrees[lv - 1] + 1
    rng = np.random.default_rng(lv)
    xc, xf = ceed.vector(nc).set_array(rng.uniform(-1, 1, nc)), ceed.vector(nf).set_array(rng.uniform(-1, 1, nf))
    yc, yf = ceed.vector(nc), ceed.vector(nf)
    abytes = 16 * (nf + nc) + 4 * mesh.nelem * (Pf ** 3 + Pc ** 3)     # the reference formulation: value + multVec per dof
    # what the OWNER form has to move: every fine and coarse dof once, both index arrays; a restriction also writes and re-reads
    # the coarse E-vector (24 B per coarse element node) and reads its transpose map (4 B per entry)
    floor = {"prolong": 8 * (nf + nc) + 4 * mesh.nelem * (Pf ** 3 + Pc ** 3),
             "restrict": 8 * (nf + nc) + 4 * mesh.nelem * (Pf ** 3 + Pc ** 3) + mesh.nelem * Pc ** 3 * (48 + 4)}
    for name, op, fn in (("prolong", p.levels[lv].opProlong, lambda: p.prolong(lv, xc, yf)), ("restrict", p.levels[lv].opRestrict, lambda: p.restrict(lv, xf, yc))):
        for _ in range(10):
            fn()
        ceed.synchronize()
        op.set_timing(True)
        for _ in range(a.steps):
            fn()
        ceed.synchronize()
        ms, _ = op.get_timing(); op.set_timing(False)
        us = 1e3 * ms / a.steps
        rows.append({"op": name, "level": lv, "Pc": Pc, "Pf": Pf, "fine_dofs": nf, "coarse_dofs": nc, "us_per_apply": us, "algorithmic_bytes": abytes,
                     "GBs": abytes / us / 1e3, "frac_of_8TBs": abytes / us / 1e3 / 8000.0, "own_floor_bytes": floor[name],
                     "own_floor_GBs": floor[name] / us / 1e3, "own_floor_frac_of_8TBs": floor[name] / us / 1e3 / 8000.0, "kernel": op.kernel_name})
rec = {"mesh": getattr(mesh, "name", ""), "elements": mesh.nelem, "transfers": rows}
print(json.dumps(rec))
if a.out:
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
for r in rows:
    print("# %-8s level %d  Pc=%d Pf=%d  %9d fine dofs  %8.1f us  %7.0f GB/s (%.2f of 8 TB/s) on the reference bytes, %5.0f GB/s (%.2f) on its own floor  %s" % (r["op"], r["level"], r["Pc"], r["Pf"], r["fine_dofs"], r["us_per_apply"], r["GBs"], r["frac_of_8TBs"], r["own_floor_GBs"], r["own_floor_frac_of_8TBs"], r["kernel"]), file=sys.stderr)
