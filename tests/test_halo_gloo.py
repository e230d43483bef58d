"""N > 1 path on CPU: world_size 2 and 3 (gloo), element partition + interface halo sum (replaces
DMLocalToGlobal(ADD_VALUES), matops.c:57) against the single-rank result on the whole mesh."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh, key_bytes
from ceedpetscsolid_amd.solid import SolidProblem
from conftest import rel_err

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _halo_worker  # noqa: E402


@pytest.mark.parametrize("mode,world", [("slab", 2), ("partition", 2), ("slab", 3)],
                         ids=["slab-2", "partition-2", "slab-3 (an interior rank with two neighbours)"])
def test_two_rank_jacobian_matches_single_rank(oracle, mode, world):
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        mp.spawn(_halo_worker.run, args=(world, initfile, d, mode), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(world)]
    # single-rank reference on the whole mesh
    full = hollow_cylinder_mesh(2, 6, 2 * world, z0=-world, z1=world)
    p = SolidProblem(oracle, full, 3, "hyperFS", nu=0.3, E=1.0, bc_sides=[998, 999], multigrid="none")
    lv = p.levels[p.fine]
    n = p.lsize()
    xyz = lv.dofmap.node_coords
    u = 0.05 * np.stack([np.sin(xyz[:, 1]) * xyz[:, 2], np.cos(xyz[:, 0]) * 0.5 * xyz[:, 2], np.sin(xyz[:, 0] + xyz[:, 1])], axis=1).reshape(-1)
    X, Y = oracle.vector(n), oracle.vector(n)
    X.set_array(u); p.form_residual(X, Y)
    x = _halo_worker.coord_field(xyz, lv.mask)
    X.set_array(x); p.apply_jacobian(p.fine, X, Y)
    yref = Y.to_numpy().reshape(-1, 3)
    per_interface = 3 * (2 * 3 + 1) * (6 * 3)
    for r, part in enumerate(parts):
        assert part["nglob"] == p.n_free()
        assert part["nshared"] == per_interface * (1 if r in (0, world - 1) else 2)
    assert abs(parts[0]["dot"] - x @ Y.to_numpy()) < 1e-12 * abs(x @ Y.to_numpy())
    # match nodes by their partition-independent topological keys (interior-of-element keys excluded)
    kfull = key_bytes(lv.dofmap.node_keys)
    order = np.argsort(kfull)
    for part in parts:
        kp = key_bytes(part["keys"])
        shared_kind = part["keys"][:, 0] < 3           # vertices, edges, faces have global keys
        pos = np.searchsorted(kfull[order], kp[shared_kind])
        idx = order[pos]
        assert np.array_equal(kfull[idx], kp[shared_kind])
        got = part["y"].reshape(-1, 3)[shared_kind]
        assert rel_err(got, yref[idx]) < 1e-12


@pytest.mark.parametrize("mode,world", [("strong-cyl", 2), ("blocks", 4)], ids=["config 4 form: uneven layers of one cylinder, 2 ranks",
                                                                              "config 5 form: 2x2x1 blocks of one box, 4 ranks"])
def test_strong_scaling_partitions_match_single_rank(oracle, mode, world):
    """The partitions bench.py --scaling strong runs (halo.part_cylinder / part_box: ONE mesh over the ranks, as BASELINE
    configs 4 and 5 are stated) against the single-rank apply on the whole mesh; the block partition has nodes shared by
    four ranks (three additions per node, in neighbour order)."""
    from ceedpetscsolid_amd.mesh import box_mesh
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        mp.spawn(_halo_worker.run, args=(world, initfile, d, mode), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(world)]
    full, bc = (hollow_cylinder_mesh(2, 6, 5), [998, 999]) if mode == "strong-cyl" else (box_mesh(4, 4, 2, hi=(1.0, 1.0, 0.5)), [1, 2])
    p = SolidProblem(oracle, full, 3, "hyperFS", nu=0.3, E=1.0, bc_sides=bc, multigrid="none")
    lv = p.levels[p.fine]
    n = p.lsize()
    xyz = lv.dofmap.node_coords
    u = 0.05 * np.stack([np.sin(xyz[:, 1]) * xyz[:, 2], np.cos(xyz[:, 0]) * 0.5 * xyz[:, 2], np.sin(xyz[:, 0] + xyz[:, 1])], axis=1).reshape(-1)
    X, Y = oracle.vector(n), oracle.vector(n)
    X.set_array(u); p.form_residual(X, Y)
    x = _halo_worker.coord_field(xyz, lv.mask)
    X.set_array(x); p.apply_jacobian(p.fine, X, Y)
    yref = Y.to_numpy().reshape(-1, 3)
    assert sum(int(part["y"].size) for part in parts) > n            # the interfaces are replicated
    for part in parts:
        assert part["nglob"] == p.n_free()
    assert abs(parts[0]["dot"] - x @ Y.to_numpy()) < 1e-12 * abs(x @ Y.to_numpy())
    kfull = key_bytes(lv.dofmap.node_keys)
    order = np.argsort(kfull)
    for part in parts:
        kp = key_bytes(part["keys"])
        shared_kind = part["keys"][:, 0] < 3
        pos = np.searchsorted(kfull[order], kp[shared_kind])
        idx = order[pos]
        assert np.array_equal(kfull[idx], kp[shared_kind])
        # inputs agree (coordinates of shared vertices are computed per block) and so do the summed outputs
        assert rel_err(part["y"].reshape(-1, 3)[shared_kind], yref[idx]) < 1e-11


@pytest.mark.parametrize("mode,world", [("strong-cyl", 3), ("blocks", 4)])
def test_emulated_rank_has_the_neighbour_lists_of_the_real_job(mode, world):
    """bench.py --emulate-rank runs ONE rank of a partitioned job on a single GPU; its neighbour lists come from
    halo.virtual_world (the other ranks' boundary elements numbered locally) instead of the collectives.  They must be the
    lists the real job has: same neighbour ranks, same entries in the same order (compared through the partition-independent
    node keys), same interface-touching elements -- against an actual gloo run of every rank."""
    from ceedpetscsolid_amd.halo import HaloExchange, interface_elements, part_box, part_cylinder, virtual_world
    from ceedpetscsolid_amd.mesh import build_dofmap, reorder_elements_first
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_halo_worker.run, args=(world, os.path.join(d, "init"), d, mode), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(world)]
    part = (lambda r: part_cylinder(r, world, 2, 6, 5)) if mode == "strong-cyl" else (lambda r: part_box(r, world, 4, 4, 2))
    for K in range(world):
        mesh = part(K)
        vw = virtual_world(K, world, mesh, part, 3)
        lead = interface_elements(mesh, virtual=vw)
        assert int(lead.sum()) == int(parts[K]["nlead"])
        mesh = reorder_elements_first(mesh, lead)
        dm = build_dofmap(mesh, 3)
        h = HaloExchange(mesh, dm, device="cpu", virtual=vw)
        assert [nb.rank for nb in h.neigh] == list(parts[K]["neigh_rank"])
        assert [nb.dof_idx.numel() for nb in h.neigh] == list(parts[K]["neigh_n"])
        keys = np.concatenate([dm.node_keys[nb.dof_idx.numpy()[::3] // 3] for nb in h.neigh])
        assert np.array_equal(keys, parts[K]["neigh_keys"])


@pytest.mark.parametrize("coarse", ["chebyshev", "assembled", "amg"])
def test_two_rank_solve_matches_single_rank(oracle, coarse):
    """The whole Newton - PCG - pMG solve on two element partitions (halo sums after every operator, ownership-
    weighted dots, globally counted multiplicity) gives the single-rank solution.  "amg": the aggregation hierarchy under the p = 1
    level is distributed in its first transfer (round 5: test_distributed_aggregation_hierarchy_matches_single_rank looks at it more
    closely): the Krylov count of the single-rank solve within 5 %."""
    import _solver_worker
    from ceedpetscsolid_amd.solver import NewtonPMG
    world = 2
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        mp.spawn(_solver_worker.run, args=(world, initfile, d, coarse), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"solve_{r}.npz")) for r in range(world)]
    full = hollow_cylinder_mesh(1, 6, 2 * world, z0=-1.0, z1=1.0)
    p = SolidProblem(oracle, full, 2, "hyperSS", nu=0.3, E=10.0, bc_sides=[998, 999])
    s = NewtonPMG(p, clamp={998: dict(translate=(0.0, -0.05, 0.1)), 999: dict()}, coarse=coarse, coarse_cheb_its=20,
                  coarse_cheb_ratio=50.0)
    st = s.solve(1)
    assert st.converged
    X = p.levels[p.fine].dofmap.node_coords
    key = {tuple(np.round(x, 9)): i for i, x in enumerate(X)}
    U = s.U.to_numpy().reshape(-1, 3)
    for part in parts:
        assert bool(part["converged"]) and int(part["newton"]) == st.newton_its
        idx = np.array([key[tuple(np.round(x, 9))] for x in part["coords"]])
        assert rel_err(part["U"].reshape(-1, 3), U[idx]) < 1e-7
        if coarse == "amg":
            assert abs(int(part["ksp"]) - st.ksp_its) <= max(1, round(0.05 * st.ksp_its)), (int(part["ksp"]), st.ksp_its)


@pytest.mark.parametrize("world,max_coarse", [(2, 1500), (3, 1500), (2, 40), (3, 40)],
                         ids=["2 ranks, dense level below", "3 ranks, dense level below", "2 ranks, sparse level below", "3 ranks, sparse level below"])
def test_distributed_aggregation_hierarchy_matches_single_rank(oracle, world, max_coarse):
    """The coarse solve on several ranks (round 5; the reference's PCGAMG is parallel, elasticity.c:568-585): every rank assembles its
    OWN p = 1 matrix, aggregates the nodes it owns, the Galerkin product is formed per rank and summed over the ranks on the union of the
    patterns (one all-reduce of the coarse values per Newton step), the restricted residual is summed per V-cycle, everything below
    the first transfer is replicated.  No element matrix leaves its rank.  Against the single-rank solve: the same Newton count, the
    Krylov count within 2, the same solution; the level's device bytes per rank shrink with the number of ranks."""
    import _solver_worker
    from ceedpetscsolid_amd.solver import NewtonPMG
    margs = (2, 8, 8 * world, 2.0 * world)           # eight element layers per rank, layers 0.5 thick
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_solver_worker.run, args=(world, os.path.join(d, "init"), d, "amg", margs, max_coarse), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"solve_{r}.npz")) for r in range(world)]
    full = hollow_cylinder_mesh(*margs[:3], z0=-margs[3], z1=margs[3])
    p = SolidProblem(oracle, full, 2, "hyperSS", nu=0.3, E=10.0, bc_sides=[998, 999])
    s = NewtonPMG(p, clamp={998: dict(translate=(0.0, -0.05, 0.1)), 999: dict()}, coarse="amg", amg_max_coarse_dofs=max_coarse)
    st = s.solve(1)
    assert st.converged
    X = p.levels[p.fine].dofmap.node_coords
    key = {tuple(np.round(x, 9)): i for i, x in enumerate(X)}
    U = s.U.to_numpy().reshape(-1, 3)
    single_bytes = s.amg.info["per_level"][0]["distributed_bytes"]      # P, P^T and T = A P of the first transfer
    own = 0
    for part in parts:
        assert bool(part["converged"]) and int(part["newton"]) == st.newton_its
        assert abs(int(part["ksp"]) - st.ksp_its) <= 2, (int(part["ksp"]), st.ksp_its)
        idx = np.array([key[tuple(np.round(x, 9))] for x in part["coords"]])
        assert rel_err(part["U"].reshape(-1, 3), U[idx]) < 1e-7
        assert list(part["amg_rows"][1:]) == list(parts[0]["amg_rows"][1:])        # the levels below the first transfer are the same on every rank
        print("first transfer, bytes of P, P^T, A P on this rank / on one rank:", int(part["amg_level0_bytes"][0]), single_bytes)
        assert int(part["amg_level0_bytes"][0]) < (0.8 if world == 2 else 0.62) * single_bytes, (int(part["amg_level0_bytes"][0]), single_bytes)
        own += int(part["amg_level0_own_coarse"][0])
    assert own == int(parts[0]["amg_rows"][1])                                      # every coarse dof belongs to exactly one rank
    if max_coarse < 100:
        assert len(parts[0]["amg_rows"]) >= 3                                        # a sparse (summed) level and a dense one below it


@pytest.mark.parametrize("strict", [1, 0], ids=["strict: every rank raises", "not strict: every rank falls back and says so"])
def test_failed_bring_up_of_the_library_exchange_is_loud_and_collective(strict):
    """VERDICT r2 item 2 / ADVICE r2: a library exchange that cannot be brought up must be impossible to miss.  Two gloo
    ranks on the oracle (which has no communicator): rank 0's CeedXCommGetUniqueId fails -- the failure is broadcast, so
    rank 1 does not wait for ever -- and both ranks leave the same way, with the process group intact."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_halo_worker.run_bring_up, args=(2, os.path.join(d, "init"), d, strict), nprocs=2, join=True)
        outs = [np.load(os.path.join(d, f"bring{r}.npz")) for r in range(2)]
    for o in outs:
        assert str(o["kind"]) == ("raised" if strict else "fallback") and bool(o["ok"])
        assert "CeedXCommGetUniqueId failed" in str(o["note"])
        assert (not strict) == ("fell back to torch.distributed" in str(o["note"]))
        assert float(o["total"]) == 3.0 and float(o["changed"]) > 0.0
