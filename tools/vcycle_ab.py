#!/usr/bin/env python3
"""V-cycle time on one box with the apply's consumers fused (CeedXOperatorApplyChebyshev / ApplyResidual) and as passes of their
own: eager and as a replayed hipGraph, plus the pieces (3 smoothing steps per level).
    python3 tools/vcycle_ab.py --cylinder 10,110,90 --problem hyperFS"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh, load_mesh_npz
from ceedpetscsolid_amd.solid import SolidProblem
from ceedpetscsolid_amd.solver import NewtonPMG

ap = argparse.ArgumentParser()
ap.add_argument("--cylinder"); ap.add_argument("--mesh"); ap.add_argument("--problem", default="hyperFS"); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--out")
a = ap.parse_args()
mesh = hollow_cylinder_mesh(*map(int, a.cylinder.split(","))) if a.cylinder else load_mesh_npz(a.mesh)
c = cd.Ceed(cd.CeedLib(cd.PRODUCT_LIB), "/gpu/hip/mi355x")
bc = [s for s in (998, 999) if s in mesh.side_sets and len(mesh.side_sets[s])]
p = SolidProblem(c, mesh, 4, a.problem, nu=0.3, E=1e3, bc_sides=bc)
rec = {"elements": mesh.nelem, "problem": a.problem, "runs": []}


def timeit(fn, reps, graph):
    fn(); c.synchronize()
    run = fn
    g = None
    if graph:
        g = c.capture(fn); run = g.launch
    run(); c.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    c.synchronize()
    dt = 1e6 * (time.perf_counter() - t0) / reps
    if g is not None:
        g.destroy()
    return dt


for tag, fuse in (("fused", True), ("two_pass", False), ("fused_again", True), ("two_pass_again", False)):
    s = NewtonPMG(p, clamp={998: dict(translate=(0.0, -0.02, 0.05)), 999: dict()}, coarse="amg", fuse_epilogue=fuse)
    s.bcv.set_array(s.bc_values(0.1)); s.residual(s.U, s.R); s.setup_preconditioner()
    top = s.nlev - 1
    r = {"form": tag}
    for lv in range(1, s.nlev):
        w = s.w[lv]
        r[f"cheb3_zero_guess_level{lv}_us"] = timeit(lambda: s.chebyshev(lv, w["b"], w["x"], 3, True), a.reps, True)
        r[f"cheb3_level{lv}_us"] = timeit(lambda: s.chebyshev(lv, w["b"], w["x"], 3, False), a.reps, True)
        r[f"residual_level{lv}_us"] = timeit(lambda: s.level_residual(lv, w["b"], w["x"], w["z"]), a.reps, True)
    r["vcycle_eager_us"] = timeit(lambda: s.vcycle(top, s.w[top]["b"], s.kz), a.reps, False)
    r["vcycle_graph_us"] = timeit(lambda: s.vcycle(top, s.w[top]["b"], s.kz), a.reps, True)
    rec["runs"].append(r)
    print("# " + "  ".join(f"{k}={v:.1f}" if isinstance(v, float) else f"{k}={v}" for k, v in r.items()), file=sys.stderr, flush=True)
print(json.dumps(rec))
if a.out:
    json.dump(rec, open(a.out, "w"), indent=1)
