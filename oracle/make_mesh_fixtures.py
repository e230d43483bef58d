#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- convert the reference's config meshes to data fixtures.

/root/reference does not exist on the GPU box, so the two Exodus meshes BASELINE.json's
configs 2 and 3 name (meshes/cube8_4096e_6ss_s.exo, meshes/cylinder8_5580e_4ss_us.exo) are
stored as plain arrays (vertex coordinates, HEX8 connectivity in tensor order, side sets)
under tests/golden/.  Run in the build container only:  python oracle/make_mesh_fixtures.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ceedpetscsolid_amd.mesh import read_exodus, save_mesh_npz  # noqa: E402

REF = "/root/reference/meshes"
# cylinder8_44928e_2ss_us: the largest unstructured cylinder present (config 4's cylinder8_99Ke_4ss_us.exo is absent,
# SURVEY 8d): CUBIT element order and vertex numbering, 8.87 M dofs at p = 4 -- bench.py --workload mesh
# cylinder27_672e_4ns_us: the HEX27 / node-set form of cylinder8_672e_4ss_us (read through its corner nodes, node sets
# turned into side sets: mesh.read_exodus) -- tests check that the two fixtures describe the same mesh and boundaries
for name in ("cube8_4096e_6ss_s", "cylinder8_5580e_4ss_us", "cube8_8e_6ss_s", "cylinder8_672e_4ss_us", "cylinder8_44928e_2ss_us",
             "cylinder27_672e_4ns_us"):
    m = read_exodus(os.path.join(REF, name + ".exo"))
    dst = os.path.join(ROOT, "tests", "golden", f"mesh_{name}.npz")
    save_mesh_npz(m, dst)
    print(name, "verts", m.nvert, "elems", m.nelem, "side sets", {k: len(v) for k, v in m.side_sets.items()}, "->", dst)
