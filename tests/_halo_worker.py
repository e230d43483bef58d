"""Worker for the world_size-2 gloo test: element-partitioned Jacobian apply with the halo sum."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def coord_field(X, mask):
    k = np.array([[12.9898, 78.233, 37.719], [93.989, 67.345, 24.113], [45.164, 11.135, 83.951]])
    v = np.sin(X @ k.T) * 437.5453
    return ((2.0 * (v - np.floor(v)) - 1.0).reshape(-1) * (mask == 0))


def run(rank, world, initfile, outdir, mode):
    from ceedpetscsolid_amd import ceed as cd
    from ceedpetscsolid_amd.halo import HaloExchange, interface_elements, part_box, part_cylinder, slab_cylinder
    from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh, partition_slabs, reorder_elements_first, submesh
    from ceedpetscsolid_amd.solid import SolidProblem
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    lib = cd.CeedLib(os.path.join(ROOT, "oracle", "liboracle_ceed.so"))   # tests only: the oracle as local operator
    ceed = cd.Ceed(lib, "/cpu/self/oracle")
    if mode == "slab":        # the weak-scaling generator of bench.py
        mesh = slab_cylinder(rank, world, 2, 6, 2, height_per_rank=2.0)
    elif mode == "strong-cyl":   # bench.py's strong-scaling form of config 4: uneven layers of ONE cylinder
        mesh = part_cylinder(rank, world, 2, 6, 5)
    elif mode == "blocks":       # bench.py's strong-scaling form of config 5: blocks of ONE box (edges shared by four ranks)
        mesh = part_box(rank, world, 4, 4, 2)
    else:                     # generic partition of one global mesh
        full = hollow_cylinder_mesh(2, 6, 2 * world, z0=-world, z1=world)
        mesh = submesh(full, partition_slabs(full, world)[rank])
    lead = interface_elements(mesh)
    mesh = reorder_elements_first(mesh, lead)          # interface-touching elements lead (split-phase apply)
    bc = [s for s in ((1, 2) if mode == "blocks" else (998, 999)) if s in mesh.side_sets and len(mesh.side_sets[s])]
    p = SolidProblem(ceed, mesh, 3, "hyperFS", nu=0.3, E=1.0, bc_sides=bc, multigrid="none")
    lv = p.levels[p.fine]
    n = p.lsize()
    halo = HaloExchange(mesh, lv.dofmap, device="cpu")
    X, Y = ceed.vector(n), ceed.vector(n)
    # state: a smooth function of the GLOBAL coordinates so both ranks agree on shared nodes
    xyz = lv.dofmap.node_coords
    u = 0.05 * np.stack([np.sin(xyz[:, 1]) * xyz[:, 2], np.cos(xyz[:, 0]) * 0.5 * xyz[:, 2], np.sin(xyz[:, 0] + xyz[:, 1])], axis=1).reshape(-1)
    X.set_array(u); p.form_residual(X, Y)
    x = coord_field(xyz, lv.mask)
    X.set_array(x); p.apply_jacobian(p.fine, X, Y)
    y = torch.from_numpy(Y.to_numpy().copy())
    halo.add(y)
    # the overlapped form: phase 0 (leading elements, interface nodes) -> start exchange -> phase 1 -> finish
    op = lv.opJacob
    op.set_overlap_split(int(lead.sum()), halo.interface_dof_mask())
    Y2 = ceed.vector(n)
    op.apply_phase(X, Y2, 0)
    y2 = torch.from_numpy(Y2.to_numpy().copy())
    assert np.all(y2.numpy()[halo.interface_dof_mask() == 0] == 0.0)     # phase 0 touched only interface nodes
    halo.start(y2)
    y2_sent = y2.clone()
    op.apply_phase(X, Y2, 1)
    y2 = torch.from_numpy(Y2.to_numpy().copy())
    assert np.array_equal(y2.numpy()[halo.interface_dof_mask() == 1], y2_sent.numpy()[halo.interface_dof_mask() == 1])
    halo.finish(y2)
    assert np.allclose(y2.numpy(), y.numpy(), rtol=0, atol=1e-13 * np.abs(y.numpy()).max())
    nglob = halo.global_count((lv.mask == 0).astype(np.float64))
    w = torch.from_numpy(halo.owner_weight.copy())
    dot = halo.dot(torch.from_numpy(x), y, w)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), y=y.numpy(), keys=lv.dofmap.node_keys, nglob=nglob, dot=dot,
             nshared=halo.n_shared_dofs, mask=lv.mask, x=x, nlead=int(lead.sum()),
             neigh_rank=np.array([nb.rank for nb in halo.neigh]), neigh_n=np.array([nb.dof_idx.numel() for nb in halo.neigh]),
             neigh_keys=np.concatenate([lv.dofmap.node_keys[(nb.dof_idx.numpy()[::3] // 3)] for nb in halo.neigh]) if halo.neigh else np.zeros((0, 7), dtype=np.int64))
    dist.destroy_process_group()


def run_bring_up(rank, world, initfile, outdir, strict):
    """The start-up check of the library's exchange (halo.checked_rccl_halo) where the library HAS no communicator (the
    oracle): every rank must take the same way out -- HaloBringUpError when strict, the torch exchange otherwise -- and the
    process group must still work afterwards (no thread left inside a collective)."""
    from ceedpetscsolid_amd import ceed as cd
    from ceedpetscsolid_amd.halo import HaloBringUpError, HaloExchange, checked_rccl_halo, part_cylinder
    from ceedpetscsolid_amd.mesh import build_dofmap
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    ceed = cd.Ceed(cd.CeedLib(os.path.join(ROOT, "oracle", "liboracle_ceed.so")), "/cpu/self/oracle")
    mesh = part_cylinder(rank, world, 2, 6, 4)
    dm = build_dofmap(mesh, 2)
    halo = HaloExchange(mesh, dm, device="cpu")
    probe = coord_field(dm.node_coords, np.zeros(dm.lsize))
    outcome = None
    try:
        got, note = checked_rccl_halo(ceed, halo, probe, "cpu", timeout_s=30.0, strict=bool(strict))
        outcome = ("fallback", got is None, note)
    except HaloBringUpError as e:
        outcome = ("raised", True, str(e))
    y = torch.from_numpy(probe.copy())
    halo.add(y)                                  # the group is still usable
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    np.savez(os.path.join(outdir, f"bring{rank}.npz"), kind=outcome[0], ok=outcome[1], note=outcome[2], total=float(t.item()),
             changed=float(np.abs(y.numpy() - probe).max()))
    dist.destroy_process_group()
