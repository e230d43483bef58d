"""Mesh -> element restriction pipeline (the step in front of the operator path): topology counts of
the reference's meshes, tensor closure order, Dirichlet node sets, partitions."""
import os

import numpy as np
import pytest

from ceedpetscsolid_amd.mesh import (box_mesh, boundary_nodes, build_dofmap, dirichlet_mask, gll_nodes,
                                     hollow_cylinder_mesh, key_bytes, load_mesh_npz, partition_slabs,
                                     side_set_nodes, submesh)
from conftest import GOLDEN


@pytest.mark.parametrize("p", [1, 2, 3, 4, 6])
def test_box_node_counts_and_closure_order(p):
    nx, ny, nz = 3, 2, 4
    m = box_mesh(nx, ny, nz)
    dm = build_dofmap(m, p)
    assert dm.nnodes == (nx * p + 1) * (ny * p + 1) * (nz * p + 1)
    assert dm.elem_nodes.shape == (m.nelem, (p + 1) ** 3)
    assert dm.elem_nodes.min() == 0 and dm.elem_nodes.max() == dm.nnodes - 1
    assert len(np.unique(dm.elem_nodes[0])) == (p + 1) ** 3
    # tensor (x fastest) closure order with GLL spacing: node coordinates of element 0
    g = 0.5 * (gll_nodes(p + 1) + 1.0)
    X = dm.node_coords[dm.elem_nodes[0]].reshape(p + 1, p + 1, p + 1, 3)   # [c][b][a]
    assert np.allclose(X[0, 0, :, 0], g / nx) and np.allclose(X[0, :, 0, 1], g / ny) and np.allclose(X[:, 0, 0, 2], g / nz)
    # shared nodes agree geometrically from every element that touches them
    P = p + 1
    xi = 0.5 * (gll_nodes(P) + 1.0)
    for e in (1, m.nelem - 1):
        v = m.coords[m.cells[e]]
        lo, hi = v.min(axis=0), v.max(axis=0)
        Z, Y, Xg = np.meshgrid(xi, xi, xi, indexing="ij")
        ref = np.stack([lo[0] + Xg * (hi[0] - lo[0]), lo[1] + Y * (hi[1] - lo[1]), lo[2] + Z * (hi[2] - lo[2])], axis=-1)
        assert np.allclose(dm.node_coords[dm.elem_nodes[e]].reshape(P, P, P, 3), ref, atol=1e-13)


@pytest.mark.parametrize("name,p,nodes", [("cube8_4096e_6ss_s", 3, 117649), ("cylinder8_5580e_4ss_us", 4, 386564),
                                          ("cylinder8_5580e_4ss_us", 2, 52030), ("cylinder8_5580e_4ss_us", 1, 7442)])
def test_reference_mesh_node_counts(name, p, nodes):
    """L dofs per level = 3 * nodes (cfg 2: 352 947; cfg 3: 1 159 692 / 156 090 / 22 326)."""
    m = load_mesh_npz(os.path.join(GOLDEN, f"mesh_{name}.npz"))
    dm = build_dofmap(m, p)
    assert dm.nnodes == nodes
    # every element is right-handed
    X = m.coords[m.cells]
    J = np.stack([X[:, 1] - X[:, 0], X[:, 2] - X[:, 0], X[:, 4] - X[:, 0]], axis=1)
    assert (np.linalg.det(J) > 0).all()


def test_cylinder_side_sets_match_geometry():
    m = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_5580e_4ss_us.npz"))
    dm = build_dofmap(m, 2)
    r = lambda nd: np.hypot(dm.node_coords[nd, 0], dm.node_coords[nd, 1])
    assert np.allclose(dm.node_coords[side_set_nodes(m, dm, [998]), 2], -5.0)
    assert np.allclose(dm.node_coords[side_set_nodes(m, dm, [999]), 2], 5.0)
    assert r(side_set_nodes(m, dm, [996])).max() < 0.5 + 1e-9 and r(side_set_nodes(m, dm, [997])).min() > 0.99


def test_synthetic_cylinder_matches_reference_family():
    m = hollow_cylinder_mesh(10, 110, 90)
    assert m.nelem == 99000 and set(m.side_sets) == {996, 997, 998, 999}
    rr = np.hypot(m.coords[:, 0], m.coords[:, 1])
    assert abs(rr.min() - 0.5) < 1e-12 and abs(rr.max() - 1.0) < 1e-12
    assert m.coords[:, 2].min() == -5.0 and m.coords[:, 2].max() == 5.0
    small = hollow_cylinder_mesh(2, 8, 2)
    dm = build_dofmap(small, 3)
    assert dm.nnodes == (2 * 3 + 1) * (8 * 3) * (2 * 3 + 1)   # periodic in theta


def test_boundary_nodes_and_masks():
    m = box_mesh(3, 3, 3)
    dm = build_dofmap(m, 2)
    b = boundary_nodes(m, dm)
    assert b.size == 7 ** 3 - 5 ** 3
    mask = dirichlet_mask(dm, side_set_nodes(m, dm, [1]))
    assert mask.sum() == 3 * 49 and mask.size == dm.lsize
    z0 = np.nonzero(mask.reshape(-1, 3)[:, 0])[0]
    assert np.allclose(dm.node_coords[z0, 2], 0.0)


def test_partition_keys_identify_shared_nodes():
    m = hollow_cylinder_mesh(2, 6, 4)
    parts = partition_slabs(m, 2)
    assert sum(len(p) for p in parts) == m.nelem
    subs = [submesh(m, p) for p in parts]
    dms = [build_dofmap(s, 3) for s in subs]
    k = [key_bytes(d.node_keys[boundary_nodes(s, d)]) for s, d in zip(subs, dms)]
    shared = np.intersect1d(k[0], k[1])
    assert shared.size == (2 * 3 + 1) * (6 * 3)     # one z-layer of nodes
    full = build_dofmap(m, 3)
    assert dms[0].nnodes + dms[1].nnodes - shared.size == full.nnodes


@pytest.mark.parametrize("p", [2, 3, 4])
def test_interior_nodes_are_numbered_last_one_contiguous_run_per_element(p):
    """Numbering contract the fused kernel's direct stores rely on for SPEED (not for correctness): the shell nodes
    come first in first-touch order, then every element's interior nodes as one run in local lexicographic order."""
    from ceedpetscsolid_amd.mesh import hollow_cylinder_mesh, build_dofmap
    mesh = hollow_cylinder_mesh(2, 6, 3)
    dm = build_dofmap(mesh, p)
    P, m = p + 1, p - 1
    en = dm.elem_nodes.reshape(mesh.nelem, P, P, P)
    interior = en[:, 1:-1, 1:-1, 1:-1].reshape(mesh.nelem, m ** 3)
    base = dm.nnodes - mesh.nelem * m ** 3
    assert np.array_equal(interior, base + np.arange(mesh.nelem * m ** 3).reshape(mesh.nelem, m ** 3))
    shell = np.setdiff1d(dm.elem_nodes.ravel(), interior.ravel())
    assert shell.size == base and shell.max() == base - 1
    # first-touch order of the shell nodes: the first appearance positions are increasing in the node id
    flat = dm.elem_nodes.ravel()
    first = np.full(dm.nnodes, flat.size, dtype=np.int64)
    np.minimum.at(first, flat, np.arange(flat.size))
    assert np.all(np.diff(first[:base]) > 0)


def _sorted_rows(a):
    a = np.round(np.asarray(a, dtype=np.float64), 9)
    return a[np.lexsort(a.T[::-1])]


def test_hex27_node_set_mesh_equals_its_hex8_side_set_twin():
    """`cylinder27_672e_4ns_us.exo` (HEX27 blocks, boundaries as NODE sets) read through its corner nodes with the
    node sets turned into side sets, against `cylinder8_672e_4ss_us.exo` (HEX8, SIDE sets): the same elements and the same
    boundary node sets at every degree -- so `-bc_clamp 998,999` clamps the same dofs whichever file form is used."""
    a = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder27_672e_4ns_us.npz"))
    b = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_672e_4ss_us.npz"))
    assert (a.nelem, a.nvert) == (b.nelem, b.nvert) == (672, 1000)            # the 5 664 mid-edge / face / centre nodes are dropped
    assert np.array_equal(_sorted_rows(a.coords[a.cells].mean(axis=1)), _sorted_rows(b.coords[b.cells].mean(axis=1)))
    assert {k: len(v) for k, v in a.side_sets.items()} == {k: len(v) for k, v in b.side_sets.items()} == {996: 192, 997: 384, 998: 28, 999: 28}
    for p in (1, 2, 4):
        da, db = build_dofmap(a, p), build_dofmap(b, p)
        assert da.nnodes == db.nnodes
        for sid in (996, 997, 998, 999):
            assert np.array_equal(_sorted_rows(da.node_coords[side_set_nodes(a, da, [sid])]), _sorted_rows(db.node_coords[side_set_nodes(b, db, [sid])]))


def test_node_sets_become_side_sets_in_the_exodus_reader(tmp_path):
    """The node-set -> side-set rule of mesh.read_exodus on a file written here (CDF-2 like the reference's): a face is in the
    set when its four corner nodes are."""
    from scipy.io import netcdf_file
    from ceedpetscsolid_amd.mesh import read_exodus
    m = box_mesh(2, 2, 1)
    # Exodus HEX8 order from the tensor order used here: inverse of the reader's permutation
    exo = m.cells[:, [0, 1, 3, 2, 4, 5, 7, 6]] + 1
    path = str(tmp_path / "box_ns.exo")
    f = netcdf_file(path, "w", version=2)
    f.createDimension("num_nodes", m.nvert); f.createDimension("num_elem", m.nelem); f.createDimension("num_nod_per_el1", 8)
    f.createDimension("num_node_sets", 2)
    for k, ax in enumerate(("coordx", "coordy", "coordz")):
        v = f.createVariable(ax, "d", ("num_nodes",)); v[:] = m.coords[:, k]
    c = f.createVariable("connect1", "i", ("num_elem", "num_nod_per_el1")); c[:] = exo.astype(np.int32); c.elem_type = b"HEX8"
    pr = f.createVariable("ns_prop1", "i", ("num_node_sets",)); pr[:] = np.array([11, 12], dtype=np.int32)
    zlo = np.nonzero(m.coords[:, 2] == 0.0)[0] + 1
    xhi = np.nonzero(m.coords[:, 0] == 1.0)[0] + 1
    f.createDimension("num_nod_ns1", zlo.size); f.createDimension("num_nod_ns2", xhi.size)
    n1 = f.createVariable("node_ns1", "i", ("num_nod_ns1",)); n1[:] = zlo.astype(np.int32)
    n2 = f.createVariable("node_ns2", "i", ("num_nod_ns2",)); n2[:] = xhi.astype(np.int32)
    f.close()
    r = read_exodus(path)
    assert r.nelem == 4 and sorted(r.side_sets) == [11, 12]
    dm = build_dofmap(r, 3)
    z0 = side_set_nodes(r, dm, [11]); x1 = side_set_nodes(r, dm, [12])
    assert z0.size == 7 * 7 and np.all(dm.node_coords[z0][:, 2] == 0.0)          # the whole z- face of the 2x2x1 box at degree 3
    assert x1.size == 7 * 4 and np.all(dm.node_coords[x1][:, 0] == 1.0)


def test_scrambled_mesh_is_the_same_discrete_problem(oracle):
    """mesh.scramble_mesh (bench.py --scramble): random element and vertex order, random rotation of every element's local axes.
    Same geometry, side sets and operator: with inputs that depend on the node coordinates only, the residual and the Jacobian action
    agree node for node (matched through the coordinates) with the structured mesh's."""
    from ceedpetscsolid_amd.mesh import hex_rotations, hollow_cylinder_mesh, scramble_mesh
    from ceedpetscsolid_amd.solid import SolidProblem, smooth_displacement
    assert len(hex_rotations()) == 24
    m0 = hollow_cylinder_mesh(2, 8, 3)
    outs = []
    from ceedpetscsolid_amd.mesh import reorder_elements_locality
    for m in (m0, scramble_mesh(m0, 3, order=True, orient=False), scramble_mesh(m0, 4, order=True, orient=True), scramble_mesh(m0, 5, order=False, orient=True),
              reorder_elements_locality(scramble_mesh(m0, 6, order=True, orient=True))):
        p = SolidProblem(oracle, m, 2, "hyperFS", nu=0.3, E=1.0, bc_sides=[998], multigrid="none")
        X = p.levels[p.fine].dofmap.node_coords
        n = p.lsize()
        u = smooth_displacement(X, 0.1, origin=(-1., -1., -5.), span=(2., 2., 10.))
        c = p.ceed
        U, Y = c.vector(n).set_array(u), c.vector(n)
        p.form_residual(U, Y)
        r = Y.to_numpy().reshape(-1, 3)
        x = (np.sin(X @ np.array([[1.3, 0.7, 2.1], [0.4, 1.9, 0.3], [2.2, 0.5, 1.1]])) * (p.levels[p.fine].mask.reshape(-1, 3) == 0)).reshape(-1)
        U.set_array(x); p.apply_jacobian(p.fine, U, Y)
        j = Y.to_numpy().reshape(-1, 3)
        key = np.lexsort(np.round(X, 9).T)
        outs.append((X[key], r[key], j[key]))
        p.destroy()
    for X, r, j in outs[1:]:
        assert np.allclose(X, outs[0][0], atol=1e-12)
        assert np.linalg.norm(r - outs[0][1]) < 1e-12 * np.linalg.norm(outs[0][1])
        assert np.linalg.norm(j - outs[0][2]) < 1e-12 * np.linalg.norm(outs[0][2])


def test_refined_swept_mesh_keeps_the_unstructured_cross_section():
    """mesh.refine_swept_mesh (bench.py --workload mesh --refine-layers 53: the reference's 44 928-hex cylinder's CUBIT-paved cross-section
    at config 4's size, 99 216 hexes): every quad of the cross-section becomes four, new boundary vertices stay on the circles, the
    result is a conforming right-handed HEX8 mesh with the two end caps as side sets."""
    from ceedpetscsolid_amd.mesh import load_mesh_npz, refine_swept_mesh
    m = load_mesh_npz(os.path.join(GOLDEN, "mesh_cylinder8_672e_4ss_us.npz"))
    z = np.unique(np.round(m.coords[:, 2], 6))
    nq = m.nelem // (len(z) - 1)
    r = refine_swept_mesh(m, 7)
    assert r.nelem == 4 * nq * 7
    assert sorted(len(v) for k, v in r.side_sets.items() if k in (998, 999)) == [4 * nq, 4 * nq]
    X = r.coords[r.cells]
    dx = (X[:, 1::2] - X[:, 0::2]).mean(axis=1); dy = (X[:, [2, 3, 6, 7]] - X[:, [0, 1, 4, 5]]).mean(axis=1); dz = (X[:, 4:] - X[:, :4]).mean(axis=1)
    det = np.einsum("ij,ij->i", np.cross(dx, dy), dz)
    assert det.min() > 0
    h = m.coords[:, 2].max() - m.coords[:, 2].min()
    assert abs(det.sum() / (np.pi * 0.75 * h) - 1.0) < 0.01          # the annulus, polygonal boundary
    rr = np.hypot(r.coords[:, 0], r.coords[:, 1])
    assert rr.min() > 0.5 - 1e-9 and rr.max() < 1.0 + 1e-9
    # conforming: every face belongs to one (boundary) or two elements, and the boundary faces are the caps and the two lateral surfaces
    faces = []
    for f in range(6):
        axis, side = f // 2, f % 2
        cs = [c for c in range(8) if ((c >> axis) & 1) == side]
        faces.append(np.sort(r.cells[:, cs], axis=1))
    faces = np.concatenate(faces)
    _, cnt = np.unique(faces, axis=0, return_counts=True)
    assert set(cnt.tolist()) <= {1, 2}
    on_bdry = int((cnt == 1).sum())
    edges_on_circles = on_bdry - 2 * 4 * nq                            # lateral faces = boundary edges of the cross-section x layers
    assert edges_on_circles % 7 == 0 and edges_on_circles > 0
