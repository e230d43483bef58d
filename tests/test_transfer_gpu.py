"""Prolong_Ceed / Restrict_Ceed (src/matops.c:115-203; opProlong / opRestrict, src/setuplibceed.c:847-862) in the OWNER form
of round 5 (csrc/kernels_misc.hip k_transfer): every fine node is stored (prolong) or read (restrict) by ONE element.  The
result differs from the reference's sum over the sharing elements times 1 / multiplicity by rounding only; these tests pin
that on every ladder pair, on unstructured meshes, with the weighted path of an element partition (scale != 1 / local
multiplicity), in the extension-free form (no scale at all), under ApplyAdd, and after the scale vector is rewritten."""
import ctypes as C

import numpy as np
import pytest

from ceedpetscsolid_amd import ceed as cd
from ceedpetscsolid_amd.mesh import box_mesh, hollow_cylinder_mesh
from ceedpetscsolid_amd.solid import SolidProblem
from conftest import operator_golden_problem, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def distorted_box(nx, ny, nz, seed=0, amp=0.04):
    m = box_mesh(nx, ny, nz)
    m.coords += amp / max(nx, ny, nz) * np.random.default_rng(seed).uniform(-1, 1, m.coords.shape)
    return m


LADDERS = [  # every (Pc, Pf) pair the library instantiates shows up in one of these
    ("log p2", lambda: distorted_box(3, 2, 2), 2, {}),                                  # (2,3)
    ("log p3", lambda: distorted_box(2, 3, 2), 3, {}),                                  # (2,3) (3,4)
    ("log p4 cyl", lambda: hollow_cylinder_mesh(2, 8, 3), 4, {}),                       # (2,3) (3,5)
    ("log p5", lambda: distorted_box(2, 2, 1), 5, {}),                                  # (3,5) (5,6)
    ("log p6", lambda: distorted_box(2, 1, 2), 6, {}),                                  # (3,5) (5,7)
    ("log p7", lambda: distorted_box(1, 2, 1), 7, {}),                                  # (5,8)
    ("uniform p7", lambda: distorted_box(1, 1, 2), 7, dict(multigrid="uniform")),      # (2,3) (3,4) (4,5) (5,6) (6,7) (7,8)
    ("ragged p4", lambda: distorted_box(5, 1, 1), 4, {}),                               # element count not a multiple of the group
    ("single element p3", lambda: distorted_box(1, 1, 1), 3, {}),
]


def _pair(oracle, gpu, mesh, degree, **kw):
    kw.setdefault("bc_sides", [1])
    return (SolidProblem(oracle, mesh, degree, "linElas", nu=0.3, E=1.0, **kw),
            SolidProblem(gpu, mesh, degree, "linElas", nu=0.3, E=1.0, **kw))


@pytest.mark.parametrize("name,mk,degree,kw", LADDERS, ids=[c[0] for c in LADDERS])
def test_transfer_matches_oracle_and_is_adjoint(oracle, gpu, name, mk, degree, kw):
    mesh = mk()
    pa, pb = _pair(oracle, gpu, mesh, degree, bc_sides=[998] if "cyl" in name else [1], **kw)
    rng = np.random.default_rng(7)
    for lv in range(1, len(pa.levels)):
        nf, nc = pa.lsize(lv), pa.lsize(lv - 1)
        xc, xf = rng.uniform(-1, 1, nc), rng.uniform(-1, 1, nf)
        out = []
        for p in (pa, pb):
            c = p.ceed
            Xc, Yf, Xf, Yc = c.vector(nc).set_array(xc), c.vector(nf), c.vector(nf).set_array(xf), c.vector(nc)
            Yf.set_value(5.0); Yc.set_value(5.0)                    # overwrite semantics
            p.prolong(lv, Xc, Yf); p.restrict(lv, Xf, Yc)
            out.append((Yf.to_numpy(), Yc.to_numpy()))
        # the owner form differs from the reference's sum over the sharing elements x 1 / multiplicity by ROUNDING only: held at 1e-13,
        # three orders inside the parity bar
        assert rel_err(out[1][0], out[0][0]) < 1e-13, (lv, "prolong", rel_err(out[1][0], out[0][0]))
        assert rel_err(out[1][1], out[0][1]) < 1e-13, (lv, "restrict", rel_err(out[1][1], out[0][1]))
        assert "prolong<" in pb.levels[lv].opProlong.kernel_name and "restrict<" in pb.levels[lv].opRestrict.kernel_name
        # constrained entries: zero on both sides
        assert np.all(out[1][0][pa.levels[lv].mask != 0] == 0.0) and np.all(out[1][1][pa.levels[lv - 1].mask != 0] == 0.0)
        # Restrict = Prolong^T on the device alone, to rounding
        lhs, rhs = float(out[1][0] @ xf), float(xc @ out[1][1])
        assert abs(lhs - rhs) <= 1e-13 * max(abs(lhs), abs(rhs), 1.0), (lv, lhs, rhs)
    pa.destroy(); pb.destroy()


@pytest.mark.parametrize("name", ["cyl_p4_hyperFS_swept", "cyl_p2_hyperFS_general", "box_p3_hyperFS_affine"])
def test_transfer_on_the_reference_meshes(oracle, gpu, name):
    """Unstructured numbering (the fixtures' meshes come from the reference's .exo files and perturbed boxes)."""
    pa, _ = operator_golden_problem(oracle, name)
    pb, _ = operator_golden_problem(gpu, name)
    rng = np.random.default_rng(3)
    for lv in range(1, len(pa.levels)):
        nf, nc = pa.lsize(lv), pa.lsize(lv - 1)
        xc, xf = rng.uniform(-1, 1, nc), rng.uniform(-1, 1, nf)
        res = []
        for p in (pa, pb):
            c = p.ceed
            Xc, Xf, Yf, Yc = c.vector(nc).set_array(xc), c.vector(nf).set_array(xf), c.vector(nf), c.vector(nc)
            p.prolong(lv, Xc, Yf); p.restrict(lv, Xf, Yc)
            res.append((Yf.to_numpy(), Yc.to_numpy()))
        assert rel_err(res[1][0], res[0][0]) < TOL and rel_err(res[1][1], res[0][1]) < TOL, lv
    pa.destroy(); pb.destroy()


def test_weighted_owner_form_of_an_element_partition(oracle, gpu):
    """The scale of a rank of an element partition holds the multiplicity over ALL ranks: at interface nodes it is not the
    reciprocal of the local one, and the kernels read the per-dof weight scale x local multiplicity.  Emulated by a scale
    vector that counts phantom neighbours on one face; then the vector is REWRITTEN in place (as solver._refresh_multiplicity
    does) and the next apply must see the new values."""
    mesh = hollow_cylinder_mesh(2, 8, 4)
    pa, pb = _pair(oracle, gpu, mesh, 4, bc_sides=[998])
    rng = np.random.default_rng(11)
    lv = len(pa.levels) - 1
    nf, nc = pa.lsize(lv), pa.lsize(lv - 1)
    mult = 1.0 / pa.levels[lv].multinv.to_numpy()
    z = pa.levels[lv].dofmap.node_coords[:, 2]
    phantom = np.repeat(((np.abs(z - z.max()) < 1e-12) | (np.abs(z - z.min()) < 1e-12)).astype(np.float64), 3)   # both caps "shared with another rank" (one of them is clamped)
    xc, xf = rng.uniform(-1, 1, nc), rng.uniform(-1, 1, nf)
    for trial, scale in enumerate((1.0 / (mult * (1.0 + phantom)), 1.0 / (mult * (1.0 + 2.0 * phantom)), 1.0 / mult)):
        res = []
        for p in (pa, pb):
            c, L = p.ceed, p.ceed.L
            if trial == 0:
                p._scale = c.vector(nf).set_array(scale)
                for op in (p.levels[lv].opProlong, p.levels[lv].opRestrict):
                    L.chk(L.lib.CeedXOperatorSetFineScale(op.h, p._scale.h))
            else:
                p._scale.set_array(scale)                           # rewritten in place: no new SetFineScale
            Xc, Xf, Yf, Yc = c.vector(nc).set_array(xc), c.vector(nf).set_array(xf), c.vector(nf), c.vector(nc)
            p.prolong(lv, Xc, Yf); p.restrict(lv, Xf, Yc)
            res.append((Yf.to_numpy(), Yc.to_numpy()))
        assert rel_err(res[1][0], res[0][0]) < TOL and rel_err(res[1][1], res[0][1]) < TOL, trial
        if trial < 2:   # the phantom face really changed the answer (the weighted path ran)
            assert np.abs(res[1][0][phantom != 0]).max() > 0
    pa.destroy(); pb.destroy()


def test_apply_add_of_the_transfers(oracle, gpu):
    mesh = distorted_box(3, 2, 2)
    pa, pb = _pair(oracle, gpu, mesh, 4)
    rng = np.random.default_rng(5)
    lv = len(pa.levels) - 1
    nf, nc = pa.lsize(lv), pa.lsize(lv - 1)
    xc, xf, y0f, y0c = rng.uniform(-1, 1, nc), rng.uniform(-1, 1, nf), rng.uniform(-1, 1, nf), rng.uniform(-1, 1, nc)
    res = []
    for p in (pa, pb):
        c, L = p.ceed, p.ceed.L
        Yf, Yc, Xc, Xf = c.vector(nf).set_array(y0f), c.vector(nc).set_array(y0c), c.vector(nc).set_array(xc), c.vector(nf).set_array(xf)
        req = C.c_void_p(L.REQUEST_IMMEDIATE)
        L.chk(L.lib.CeedOperatorApplyAdd(p.levels[lv].opProlong.h, Xc.h, Yf.h, req))
        L.chk(L.lib.CeedOperatorApplyAdd(p.levels[lv].opRestrict.h, Xf.h, Yc.h, req))
        res.append((Yf.to_numpy(), Yc.to_numpy()))
    assert rel_err(res[1][0], res[0][0]) < TOL and rel_err(res[1][1], res[0][1]) < TOL
    pa.destroy(); pb.destroy()


def test_transfer_on_a_scrambled_mesh(oracle, gpu):
    """Elements and vertices in random order and every element's local axes rotated at random (mesh.scramble_mesh): neighbours no
    longer agree on which local direction is which, so a shared face's fine nodes are interpolated along DIFFERENT local directions by
    the element that owns them and by the ones the reference would also sum -- still one value to rounding."""
    from ceedpetscsolid_amd.mesh import scramble_mesh
    mesh = scramble_mesh(hollow_cylinder_mesh(3, 10, 5), 20261005, order=True, orient=True)
    pa, pb = _pair(oracle, gpu, mesh, 4, bc_sides=[998])
    rng = np.random.default_rng(2)
    for lv in range(1, len(pa.levels)):
        nf, nc = pa.lsize(lv), pa.lsize(lv - 1)
        xc, xf = rng.uniform(-1, 1, nc), rng.uniform(-1, 1, nf)
        res = []
        for p in (pa, pb):
            c = p.ceed
            Xc, Xf, Yf, Yc = c.vector(nc).set_array(xc), c.vector(nf).set_array(xf), c.vector(nf), c.vector(nc)
            p.prolong(lv, Xc, Yf); p.restrict(lv, Xf, Yc)
            res.append((Yf.to_numpy(), Yc.to_numpy()))
        assert rel_err(res[1][0], res[0][0]) < 1e-13 and rel_err(res[1][1], res[0][1]) < 1e-13, lv
    pa.destroy(); pb.destroy()
