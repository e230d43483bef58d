#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/operators.npz.

Whole-operator vectors with the REFERENCE's own compiled callbacks inside: the oracle's CeedOperatorApply /
CeedOperatorLinearAssembleDiagonal run the reference's SetupGeo, *F and *dF (qfunctions/*.h compiled where they lie into
oracle/_ref/libref_qfunctions.so, handed over as the user-callback pointer of CeedQFunctionCreateInterior exactly as
setuplibceed.c:470-474,822-824 hands problemOptions[...].apply / .jacob) on a few small meshes, and the residual, stored
state, Jacobian action and diagonal are stored as plain data.  tests/test_gpu_parity.py::test_operators_match_reference_callbacks
then compares the HIP path with THESE arrays at 1e-10: the device result is tied to the object code of e.g.
hyperFS.h:286-464 end to end, not only to the restated physics.  (It does not pin the libCEED half -- basis, restriction and
the operator's orchestration are still the oracle's restatement -- and the header the callbacks are compiled against is this
repo's include/ceed.h: DESIGN.md section 2 says what that does and does not prove.)

Needs /root/reference (through oracle/_ref), so it runs in the build container only; the fixture is committed.

    python oracle/gen_operator_golden.py        # rewrites tests/golden/operators.npz
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from ceedpetscsolid_amd import ceed as cd  # noqa: E402
from ceedpetscsolid_amd.mesh import HexMesh, box_mesh, hollow_cylinder_mesh  # noqa: E402
from ceedpetscsolid_amd.solid import SolidProblem  # noqa: E402

# (name, mesh factory, degree, problem, SolidProblem keywords)
def distorted_box(nx, ny, nz, seed, amp=0.04):
    m = box_mesh(nx, ny, nz)
    m.coords += amp / max(nx, ny, nz) * np.random.default_rng(seed).uniform(-1, 1, m.coords.shape)
    return m


def sheared_cylinder():
    """A hollow cylinder whose cross-section is sheared with height: still extruded in the mesh's sense? No -- x and y now depend
    on z inside an element, so the elements are general hexahedra (the per-point geometry path)."""
    m = hollow_cylinder_mesh(2, 8, 3, z0=-1.0, z1=1.0)
    m.coords[:, 0] += 0.08 * m.coords[:, 2] ** 2
    return m


CASES = [
    ("box_p2_linElas", lambda: distorted_box(3, 2, 2, 11), 2, "linElas", dict(bc_sides=[1])),
    ("box_p2_hyperSS", lambda: distorted_box(3, 2, 2, 12), 2, "hyperSS", dict(bc_sides=[1, 2])),
    ("box_p2_hyperFS", lambda: distorted_box(3, 2, 2, 13), 2, "hyperFS", dict(bc_sides=[6])),
    ("box_p3_hyperFS_affine", lambda: box_mesh(2, 3, 2, lo=(0., 0., 0.), hi=(1., 1.5, 0.8)), 3, "hyperFS", dict(bc_sides=[1])),
    ("cyl_p4_hyperFS_swept", lambda: hollow_cylinder_mesh(2, 8, 3), 4, "hyperFS", dict(bc_sides=[998, 999])),
    ("cyl_p4_hyperSS_swept", lambda: hollow_cylinder_mesh(2, 8, 2), 4, "hyperSS", dict(bc_sides=[998])),
    ("cyl_p2_hyperFS_general", sheared_cylinder, 2, "hyperFS", dict(bc_sides=[998])),
    ("cyl_p4_linElas_swept", lambda: hollow_cylinder_mesh(1, 8, 2), 4, "linElas", dict(bc_sides=[999])),
    # -qextra > 0 (src/cloptions.c:53-55; Q = degree + 1 + qextra, setuplibceed.c:252,757): the fine level itself runs a P < Q kernel
    ("box_p2_hyperFS_qextra1", lambda: distorted_box(2, 2, 2, 21), 2, "hyperFS", dict(bc_sides=[1], qextra=1)),
    ("cyl_p3_hyperSS_qextra2", lambda: hollow_cylinder_mesh(1, 8, 2), 3, "hyperSS", dict(bc_sides=[998], qextra=2)),
]


def main():
    ref_path = os.path.join(HERE, "_ref", "libref_qfunctions.so")
    if not os.path.exists(ref_path):
        sys.exit("oracle/_ref/libref_qfunctions.so missing: run `make -C oracle` in the build container")
    ref = C.CDLL(ref_path)
    ref.RefGetQFunction.restype = C.c_void_p
    ref.RefGetQFunction.argtypes = [C.c_char_p]

    def callback(name):
        p = ref.RefGetQFunction(name.encode())
        assert p, f"the reference has no QFunction {name}"
        return C.c_void_p(p)

    orc = cd.Ceed(cd.CeedLib(os.path.join(HERE, "liboracle_ceed.so")), "/cpu/self/oracle")
    out = {"cases": np.array([c[0] for c in CASES])}
    for name, mk, degree, problem, kw in CASES:
        mesh = mk()
        nu, E = 0.3, 2.5
        p = SolidProblem(orc, mesh, degree, problem, nu=nu, E=E, qf_callbacks=callback, **kw)
        n = p.lsize()
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        u = p.smooth_state(0.15)
        X, Y = orc.vector(n).set_array(u), orc.vector(n)
        p.form_residual(X, Y)
        pre = name + "."
        out[pre + "coords"], out[pre + "cells"] = mesh.coords, mesh.cells
        out[pre + "side_ids"] = np.array(sorted(mesh.side_sets), dtype=np.int64)
        for sid in mesh.side_sets:
            out[pre + f"side_{sid}"] = np.asarray(mesh.side_sets[sid])
        out[pre + "meta"] = np.array([degree, nu, E, kw.get("qextra", 0)])
        out[pre + "problem"] = np.array(problem)
        out[pre + "bc_sides"] = np.array(kw.get("bc_sides", []), dtype=np.int64)
        out[pre + "offsets"] = p.levels[p.fine].dofmap.offsets()
        out[pre + "u"], out[pre + "residual"] = u, Y.to_numpy()
        out[pre + "qdata"] = p.qdata.to_numpy()
        if p.gradu is not None:
            out[pre + "gradu"] = p.gradu.to_numpy()
        for lv in range(len(p.levels)):
            nl = p.lsize(lv)
            x = rng.uniform(-1, 1, nl)
            Xl, Yl, D = orc.vector(nl).set_array(x), orc.vector(nl), orc.vector(nl)
            p.apply_jacobian(lv, Xl, Yl)
            p.get_diag(lv, D)
            out[pre + f"x{lv}"], out[pre + f"jacobian{lv}"], out[pre + f"diag{lv}"] = x, Yl.to_numpy(), D.to_numpy()
        out[pre + "nlevels"] = np.array(len(p.levels))
        print(name, "elements", mesh.nelem, "dofs", n, "levels", [lv.degree for lv in p.levels],
              "|residual|", float(np.linalg.norm(out[pre + "residual"])))
        p.destroy()
    dst = os.path.join(ROOT, "tests", "golden", "operators.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")


if __name__ == "__main__":
    main()
