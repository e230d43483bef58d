#!/usr/bin/env python3
"""BASELINE config 3: hyperSS, cylinder8_5580e_4ss_us, degree 4, full Newton-CG-pMG solve on one MI355X.

README-style boundary conditions (README.rst:63): -bc_clamp 998,999 -bc_clamp_998_translate 0,-0.5,1,
10 load increments.  Prints the reference's own summary quantities (elasticity.c:684-764).
    python examples/solve_config3.py [--problem hyperSS] [--degree 4] [--increments 10] [--mesh <npz>]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")   # see bench.py: a spinning host BLAS pool throttles the launching thread
os.environ.setdefault("MKL_NUM_THREADS", "1")
import numpy as np
from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import load_mesh_npz
from ceedpetscsolid_amd.solid import SolidProblem
from ceedpetscsolid_amd.solver import NewtonPMG

ap = argparse.ArgumentParser()
ap.add_argument("--problem", default="hyperSS")
ap.add_argument("--degree", type=int, default=4)
ap.add_argument("--increments", type=int, default=10)
ap.add_argument("--mesh", default=os.path.join(ROOT, "tests", "golden", "mesh_cylinder8_5580e_4ss_us.npz"))
ap.add_argument("--cylinder", default=None, metavar="NR,NTH,NZ",
                help="instead of --mesh: the structured hollow cylinder of bench.py (10,110,90 = the 99 000-hex stand-in of BASELINE config 4)")
ap.add_argument("--translate", default="0,-0.05,0.1",
                help="clamp 998 translation; README.rst:63 uses 0,-0.5,1 (for linElas): with degree-4 elements the jump of a tenth of that per load step already puts the first GLL layer at O(1) strain, outside the small-strain model")
ap.add_argument("--E", type=float, default=1e3)
ap.add_argument("--nu", type=float, default=0.3)
ap.add_argument("--oracle", action="store_true", help="TESTS ONLY: run the same solve on the CPU oracle")
ap.add_argument("--verbose", action="store_true")
ap.add_argument("--coarse", default="cg", choices=["cg", "chebyshev", "assembled", "amg"])
ap.add_argument("--graph", action="store_true", help="replay the V-cycle as a hipGraph")
ap.add_argument("--auto", action="store_true", help="time the V-cycle eager / replayed, fused / two-pass at the first Newton step and keep the fastest")
ap.add_argument("--coarse-cheb-its", type=int, default=40)
ap.add_argument("--coarse-cheb-ratio", type=float, default=100.0)
ap.add_argument("--amg-smooth-its", type=int, default=3)
ap.add_argument("--amg-smooth-ratio", type=float, default=10.0)
ap.add_argument("--amg-coarse-cycles", type=int, default=1, help="cycles of an intermediate level of the aggregation hierarchy per visit (2: a W-cycle)")
ap.add_argument("--amg-max-coarse", type=int, default=1500, help="rows up to which a level of the aggregation hierarchy is stored dense and inverted")
ap.add_argument("--coarse-maxit", type=int, default=200)
ap.add_argument("--coarse-rtol", type=float, default=1e-3)
ap.add_argument("--no-fuse", action="store_true", help="A/B: the smoother's Chebyshev step and the V-cycle's residual as passes of their own")
args = ap.parse_args()

# several GPUs: `python -m torch.distributed.run --nproc-per-node N examples/solve_config3.py ...` -- one element
# partition (z-slabs) per rank, halo sums inside the solver (SOLVE_DIST_BACKEND=gloo rehearses it on one GPU)
world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
halos = None
if args.cylinder:
    from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh
    mesh = hollow_cylinder_mesh(*[int(v) for v in args.cylinder.split(",")])
    args.mesh = f"hollow_cylinder_{args.cylinder}"
else:
    mesh = load_mesh_npz(args.mesh)
if world > 1:
    import torch, torch.distributed as dist
    from ceedpetscsolid_amd.halo import HaloExchange
    from ceedpetscsolid_amd.mesh import partition_slabs, submesh
    backend = os.environ.get("SOLVE_DIST_BACKEND", "gloo" if args.oracle else "nccl")
    if not args.oracle:
        torch.cuda.set_device(local_rank if backend == "nccl" else 0)
    dist.init_process_group(backend)
    mesh = submesh(mesh, partition_slabs(mesh, world)[rank])
if args.oracle:
    lib = cd.CeedLib(os.path.join(ROOT, "oracle", "liboracle_ceed.so")); ceed = cd.Ceed(lib, "/cpu/self/oracle")
else:
    lib = cd.CeedLib(cd.PRODUCT_LIB); ceed = cd.Ceed(lib, "/gpu/hip/mi355x")
    if world > 1:
        ceed.set_stream(torch.cuda.current_stream().cuda_stream)
t0 = time.perf_counter()
bc_sides = [s for s in (998, 999) if s in mesh.side_sets and len(mesh.side_sets[s])]
prob = SolidProblem(ceed, mesh, args.degree, args.problem, nu=args.nu, E=args.E, bc_sides=bc_sides)
if world > 1:
    halos = [HaloExchange(mesh, lv.dofmap, device="cpu" if args.oracle else torch.device("cuda", torch.cuda.current_device()))
             for lv in prob.levels]
tr = tuple(float(t) for t in args.translate.split(","))
solver = NewtonPMG(prob, clamp={s: (dict(translate=tr) if s == 998 else dict()) for s in bc_sides}, halo=halos, verbose=args.verbose and rank == 0,
                   coarse_maxit=args.coarse_maxit, coarse_rtol=args.coarse_rtol, coarse=args.coarse, graph="auto" if args.auto else args.graph,
                   coarse_cheb_its=args.coarse_cheb_its, coarse_cheb_ratio=args.coarse_cheb_ratio,
                   amg_smooth_its=args.amg_smooth_its, amg_smooth_ratio=args.amg_smooth_ratio, amg_max_coarse_dofs=args.amg_max_coarse, amg_coarse_cycles=args.amg_coarse_cycles,
                   fuse_epilogue="auto" if args.auto else not args.no_fuse)
t_setup = time.perf_counter() - t0
st = solver.solve(args.increments)
u = solver.U.to_numpy().reshape(-1, 3)
out = {"resource": ceed.resource, "problem": args.problem, "mesh": os.path.basename(args.mesh), "elements": mesh.nelem,
       "level_degrees": prob.degrees, "global_dofs_per_level": [prob.n_free(l) for l in range(len(prob.levels))],
       "translate_998": list(tr), "coarse_solver": args.coarse, "vcycle_graph": solver.graph, "fused_epilogue": solver.fuse_epilogue, "vcycle_tuning": solver.tuning, "load_increments": st.increments, "converged": st.converged, "snes_its": st.newton_its, "ksp_its": st.ksp_its,
       "coarse_cg_its": st.coarse_its, "jacobian_applies": st.jacobian_applies, "residual_evals": st.residual_evals, "coarse_spmv": st.coarse_spmv,
       "setup_s": t_setup, "snes_solve_s": st.seconds,
       "amg": ({k: solver.amg.info.get(k) for k in ("levels", "rows", "build_seconds", "per_level")} if getattr(solver, "amg", None) is not None else None),
       "ranks": world,
       "MDoFs_per_s_in_SNES": 1e-6 * (halos[-1].global_count((prob.levels[prob.fine].mask == 0).astype(np.float64)) if halos else prob.n_free()) * st.ksp_its / st.seconds,   # elasticity.c:755-764
       "max_abs_displacement": np.abs(u).max(axis=0).tolist(), "final_residual_norm": st.history[-1][4] if st.history else None}
if rank == 0:
    print(json.dumps(out))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
